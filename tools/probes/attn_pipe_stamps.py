"""Phase stamps (s_memtime) of workgroup 0 of the pipelined attention kernel, probe build -DFS2_ATTN_PROBE=4096: per item the
cycles from item start to [K landed + barrier, prologue done, tile loop done, epilogue issued, stores drained].
r03 (C2 decoder shape): 256-query item 1728 / 3620 / 61356 / 65532 / 66448, i.e. 37.6 cycles per MFMA in the loop (floor 32);
128-query item 1256 / 2612 / 39748 / 41440 / 42224 (48.4 per MFMA: twice the LDS reads per MFMA).  108.7 k cycles for a 74-78 us
launch: the kernel runs at ~1.45 GHz - power-limited - not at the 2.07 GHz an MFMA-only loop gets (tools/probes/mfma_filler_cost.hip)."""
import sys, torch, ctypes as C, subprocess
R = "/root/repo"
so = "/tmp/attn_stamp.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DFS2_ATTN_PROBE=4096",
                f"-I{R}/lightningfastspeech2_amd/csrc", f"-I{R}/include", "-o", so, f"{R}/lightningfastspeech2_amd/csrc/attention_pipe.hip"], check=True, stderr=subprocess.DEVNULL)
lib = C.CDLL(so)
B, S, H, heads = 32, 1536, 256, 2
qkv = torch.randn(B * S, 3 * H, device="cuda").to(torch.bfloat16)
out = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
bits = torch.full((B, 24), -1, dtype=torch.int64, device="cuda")
dbg = torch.zeros(B * heads * S + 1024, device="cuda")
lib.attn_pipe_probe_dbg.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
for it in range(3):
    dbg.zero_()
    lib.attn_pipe_probe_dbg(qkv.data_ptr(), bits.data_ptr(), out.data_ptr(), dbg.data_ptr(), B, S, H, heads, int(sys.argv[1]) if len(sys.argv) > 1 else 3, None)
    torch.cuda.synchronize()
d = dbg.cpu()[B * heads * S:B * heads * S + 12 * 8].view(12, 8)
print("per item of workgroup 0 (cycles since item start): [0, K landed + barrier, prologue done, loop done, epilogue issued, stores drained, NQB, t0 low bits]")
for r in d:
    if r[6] > 0: print([int(x) for x in r])
