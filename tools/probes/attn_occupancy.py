"""How much do two co-resident attention workgroups overlap?  Same per-workgroup work (S = 1024 keys, 128
queries each), 256 workgroups (one per CU) vs 512 (two per CU) vs 768 / 1024."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]]
import tools.bench_ops as bo
for B in (16, 32, 48, 64):
    bo.attn_case(f"attention B={B} (blocks={B*2*8})", B, 1024, 256, 2, 30)
