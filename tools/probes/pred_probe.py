"""Where does the single-launch predictor's time go?  Builds predictor_fused.hip with -DFS2_PRED_PROBE=<bits> into throw-away
libraries (parts of the loop compiled out; results are garbage) and times each on the C2 shape (32 x 1536 frames, 5 layers).
    python tools/probes/pred_probe.py [bits ...]     64 = the full kernel"""
import ctypes as C, os, subprocess, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(R, "lightningfastspeech2_amd", "csrc", "predictor_fused.hip")
probes = [int(x) for x in sys.argv[1:]] or [64, 1, 2, 3, 4, 7, 8, 9, 11, 15]
B, S, NL = 32, 1536, 5
x = torch.randn(B * S, 256, device="cuda").to(torch.bfloat16)
wpk = (torch.randn(NL * 24 * 1024 * 8, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.zeros(NL * 256, device="cuda"); g = torch.ones(NL * 256, device="cuda"); b = torch.zeros(NL * 256, device="cuda")
hw = torch.randn(256, device="cuda"); pred = torch.empty(B * S, device="cuda")
names = {1: "no weight loads", 2: "no activation reads", 4: "no epilogue", 8: "no MFMA", 64: "(full kernel)"}
for pb in probes:
    so = f"/tmp/pred_probe_{pb}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DFS2_PRED_PROBE={pb}",
                    f"-I{R}/lightningfastspeech2_amd/csrc", f"-I{R}/include", "-o", so, src], check=True, stderr=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.pred_probe.argtypes = [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.pred_probe(x.data_ptr(), wpk.data_ptr(), bias.data_ptr(), g.data_ptr(), b.data_ptr(), hw.data_ptr(), pred.data_ptr(), B, S, NL, st)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    print(f"probe {pb:3d}  {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  " + " + ".join(n for k, n in names.items() if pb & k), flush=True)
