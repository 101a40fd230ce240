"""Same probe at head dim 64 (the kernel needs 150 VGPRs there -> THREE workgroups per CU fit): does a third
co-resident workgroup keep filling the SIMDs?  S = 1024 keys, 128 queries per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]]
import tools.bench_ops as bo
for B in (16, 32, 48, 64, 96):
    bo.attn_case(f"attention d=64 B={B} (blocks={B*2*8})", B, 1024, 128, 2, 30)
