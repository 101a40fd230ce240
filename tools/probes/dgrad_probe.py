"""Probe: the decoder conv1 data gradient (M = 49152, N = 256, K = 9 x 1024) one pass vs split over K, and the forward conv1 for scale."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]]
import tools.bench_ops as b
b.gemm_case("dec conv1 fwd k=9", 49152, 1024, 256, 9, 1536, 20, 0)
b.gemm_case("c2 dec conv1 dgrad", 49152, 256, 1024, 9, 1536, 20, 0)
for ks in (2, 3, 4, 8):
    b.splitk_case(f"c2 dec conv1 dgrad split {ks}", 49152, 256, 1024, 9, 1536, ks, 20)
