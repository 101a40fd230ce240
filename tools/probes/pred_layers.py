"""Cost of the single-launch predictor kernel by layer count (a proxy for what a resident K=256/768 GEMM+LN costs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]]
import tools.bench_ops as bo
for nl in (1, 2, 3, 5):
    bo.predictor_case(f"predictor {nl} layer(s)", 32, 1536, nl, 30)
bo.predictor_case("predictor 1 layer, B*S=8192", 32, 256, 1, 30)
