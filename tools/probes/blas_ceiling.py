"""Calibration only (not on the product path): what the vendor GEMM reaches on this box for the
forward's shapes, random bf16 data - the practical MFMA ceiling under the power-managed clock."""
import torch
dev = "cuda:0"
def t(M, N, K, reps=30):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    for _ in range(3): (a @ b.T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c = a @ b.T
    e1.record(); torch.cuda.synchronize()
    s = e0.elapsed_time(e1) / reps * 1e-3
    print(f"torch.matmul bf16 M={M} N={N} K={K}: {s*1e6:8.1f} us {2.0*M*N*K/s/1e12:7.1f} TF", flush=True)
for shp in [(49152, 1024, 2304), (4096, 4096, 4096), (8192, 8192, 8192), (49152, 256, 768), (49152, 256, 1024), (49152, 768, 256), (49152, 256, 256), (8192, 1024, 2304), (8192, 256, 1024)]:
    t(*shp)
