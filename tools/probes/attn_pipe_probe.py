"""Where does the pipelined attention kernel's time go?  Builds attention_pipe.hip with -DFS2_ATTN_PROBE=<bits> (parts of the
loop body compiled out; results are then garbage) into throw-away libraries under /tmp and times each on the C2 decoder shape.
    python tools/probes/attn_pipe_probe.py [variant] [bits ...]"""
import ctypes as C, os, subprocess, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(R, "lightningfastspeech2_amd", "csrc", "attention_pipe.hip")
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
probes = [int(x) for x in sys.argv[2:]] or [64, 1, 2, 3, 4, 8, 12, 15, 16, 32, 48, 63]
B, S, H, heads = [int(x) for x in os.environ.get("PROBE_SHAPE", "32,1536,256,2").split(",")]
qkv = (torch.randn(B * S, 3 * H, device="cuda")).to(torch.bfloat16)
out = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
bits = torch.full((B, (S + 63) // 64), -1, dtype=torch.int64, device="cuda")
names = {1: "no exp/sum/pack", 2: "no overflow check", 4: "no fragment reads", 8: "no DMA", 16: "no QK MFMA", 32: "no PV MFMA", 64: "(full kernel)", 128: "32-query items only", 256: "no tile barrier", 512: "no mask test", 8192: "direct O stores (no LDS staging)"}
for pb in probes:
    so = f"/tmp/attn_probe_{pb}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DFS2_ATTN_PROBE={pb & 16383}", 
                    f"-I{R}/lightningfastspeech2_amd/csrc", f"-I{R}/include", "-o", so, src], check=True, stderr=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.attn_pipe_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.attn_pipe_probe(qkv.data_ptr(), bits.data_ptr(), out.data_ptr(), B, S, H, heads, variant, st)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    what = " + ".join(n for b, n in names.items() if pb & b) + (f" [FQ={(pb >> 12) & 15}]" if (pb >> 12) & 15 else "")
    print(f"probe {pb:3d}  {us:7.1f} us  {4.0 * B * S * S * H / us / 1e6:6.0f} TF  {what}", flush=True)
