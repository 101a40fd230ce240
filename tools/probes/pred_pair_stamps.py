"""Segment stamps (s_memtime, 100 MHz) of workgroup 0 / wave 0 of predictor_pair_kernel, probe build -DFS2_PRED_PROBE: how long a
pure K loop (segment 0) takes against a K loop with the other tile's LayerNorm epilogue in its MFMA gaps (segments 1..2n-1) and
against the epilogue alone (the last one)."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = "/tmp/pred_probe.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DFS2_PRED_PROBE",
                f"-I{ROOT}/lightningfastspeech2_amd/csrc", f"-I{ROOT}/include", "-o", so,
                f"{ROOT}/lightningfastspeech2_amd/csrc/predictor_fused.hip"], check=True)
lib = C.CDLL(so)
B, S, nl = 32, 1536, 5
dev = "cuda:0"
x = torch.randn(B * S, 256, device=dev).to(torch.bfloat16)
w = (torch.randn(nl, 256, 768, device=dev) * 0.03).to(torch.bfloat16)
nb = 24 * 4 * 2 * 64 * 16  # PF_STEPS * PF_STEP_U4 * 16 is what the packer writes; over-allocate
wpk = torch.zeros(nl * 1 << 20, dtype=torch.uint8, device=dev)
lib.pack_for_probe.restype = C.c_size_t
per = lib.pack_for_probe(C.c_void_p(w.data_ptr()), C.c_void_p(wpk.data_ptr()), nl)
bias = torch.zeros(nl * 256, device=dev); g = torch.ones(nl * 256, device=dev); be = torch.zeros(nl * 256, device=dev)
hw = torch.randn(256, device=dev); pred = torch.empty(B * S, device=dev)
out = (C.c_ulonglong * 64)()
p = lambda t: C.c_void_p(t.data_ptr())
st = lib.pred_pair_stamps(p(x), p(wpk), p(bias), p(g), p(be), p(hw), p(pred), B, S, nl, out)
assert st == 0, st
v = list(out)
print("segment boundaries (x10 ns):", [v[i + 1] - v[i] for i in range(0, 2 * nl + 2)])
print("segment 3 taps: K+stage", [v[32 + 2 * t] - (v[33 + 2 * t - 2] if t else v[3]) for t in range(3)], "barrier", [v[33 + 2 * t] - v[32 + 2 * t] for t in range(3)])
