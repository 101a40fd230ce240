#!/usr/bin/env python
"""Micro-benchmarks of single operators through the C ABI (device-resident, HIP-event timed).
    python tools/bench_ops.py [gemm|attn|all] [--variant V] [--reps R]
Shapes are the C2 (FS2-27M, B=32, L=256, T=1536) launches of the forward."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lightningfastspeech2_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda:0"
BF16 = _lib.FS2_BF16


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


FLUSH = [None]  # --flush MB: a torch pass over that many MB between reps (evicts the L2s, and the Infinity Cache from 256 MB on):
                 # what a launch costs inside a forward, where its weights were last touched a whole layer ago


def timeit(fn, reps):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        fn(st)
    if FLUSH[0] is not None:
        tot = 0.0
        for _ in range(reps):
            FLUSH[0].add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(st)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e-3
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def gemm_case(name, M, N, Cin, taps, S, reps, variant):
    lib.fs2_op_set_gemm_variant(variant)
    x = torch.randn(M, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, taps * Cin, device=DEV) * (taps * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=DEV)
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda st: lib.fs2_op_gemm(BF16, BF16, p(x), p(w), p(b), p(c), M, N, Cin, taps, S, 1, st), reps)
    fl = 2.0 * M * N * Cin * taps
    print(f"{name:28s} M={M:6d} N={N:5d} K={taps*Cin:5d} variant={variant}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF  ({fl/t/2.5e15*100:4.1f}% of 2.5 PF)")
    lib.fs2_op_set_gemm_variant(0)
    if taps == 1 and os.environ.get("FS2_BENCH_BLASLT"):  # the vendor library on the same operands: a yardstick, not a product path
        wt = w
        for _ in range(3):
            torch.nn.functional.linear(x, wt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.nn.functional.linear(x, wt)
        e1.record()
        torch.cuda.synchronize()
        tl = e0.elapsed_time(e1) / reps * 1e-3
        print(f"{'  (hipBLASLt via torch, no bias)':28s} {'':40s} {tl*1e6:8.1f} us  {fl/tl/1e12:7.1f} TF")


def splitk_case(name, M, N, Cin, taps, S, ks, reps):
    x = torch.randn(M, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, taps * Cin, device=DEV) * (taps * Cin) ** -0.5).to(torch.bfloat16)
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    part = torch.empty(ks, M, N, device=DEV)
    t = timeit(lambda st: lib.fs2_op_gemm_splitk(BF16, BF16, p(x), p(w), p(c), p(part), M, N, Cin, taps, S, ks, 1, st), reps)
    fl = 2.0 * M * N * Cin * taps
    print(f"{name:34s} M={M} N={N} K={taps}x{Cin}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF  (launch + plane sum, accumulating)")


def gemm_ln_case(name, M, N, Cin, taps, S, reps, variant, res=True, relu=False):
    """GEMM/conv + fused LayerNorm epilogue (the N=256 launches of the forward)."""
    lib.fs2_op_set_gemm_variant(variant)
    x = torch.randn(M, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, taps * Cin, device=DEV) * (taps * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=DEV)
    r = torch.randn(M, N, device=DEV).to(torch.bfloat16) if res else None
    g = torch.randn(N, device=DEV)
    be = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    tmp = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda st: lib.fs2_op_gemm_ln(BF16, p(x), p(w), p(b), p(r), p(g), p(be), None, C.c_float(0.0), None, None,
                                             p(y), p(tmp), M, N, Cin, taps, S, int(relu), st), reps)
    fl = 2.0 * M * N * Cin * taps
    by = 2.0 * (M * Cin + M * N * (2 if res else 1))
    print(f"{name:28s} M={M:6d} N={N:5d} K={taps*Cin:5d} variant={variant}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF  "
          f"({fl/t/2.5e15*100:4.1f}% of 2.5 PF)  {by/t/1e12:5.2f} TB/s algorithmic")
    lib.fs2_op_set_gemm_variant(0)


def bgemm_case(name, dims, A, Bm, Cout, reps, **kw):
    """One fs2_op_bgemm launch class of the training step's backward (bf16 operands unless the tensors are fp32)."""
    d = _lib.BGemmDescC()
    base = dict(nb1=1, nb2=1, alpha=1.0, beta=0.0, splitk=1, taps=1, c_dtype=_lib.FS2_F32 if Cout.dtype == torch.float32 else BF16)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    nb = lib.fs2_op_bgemm_ws_bytes(C.byref(d))
    ws = torch.empty(max(1, nb // 4), device=DEV)
    dt = _lib.FS2_F32 if A.dtype == torch.float32 else BF16
    t = timeit(lambda st: lib.fs2_op_bgemm(dt, C.byref(d), p(A), p(Bm), p(Cout), None, p(ws), st), reps)
    fl = 2.0 * dims
    print(f"{name:34s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF  ({fl/t/2.5e15*100:4.1f}% of 2.5 PF)")


def bwd_cases(reps):
    bf = torch.bfloat16
    M, H, F, k = 49152, 256, 1024, 9
    dy = torch.randn(M, F, device=DEV).to(bf)
    x = torch.randn(M, H, device=DEV).to(bf)
    w = torch.randn(F, k * H, device=DEV).to(bf)
    dx = torch.empty(M, H, device=DEV, dtype=bf)
    bgemm_case("conv1 k=9 dgrad (M,1024)->(M,256)", M * H * k * F, dy, w, dx, reps, M=M, N=H, K=k * F, sAm=F, sAk=1, sBk=k * H, sBn=1,
               ldc=H, seg=1536, taps=k, Kin=F, a_shift0=4, a_shift_step=-1, sBtap=H)
    dw = torch.zeros(F, k * H, device=DEV)
    # the 256 x 256 LDS-DMA kernel (knob 1001, default) next to the 128 x 128 one (1000) at each one's split counts
    for sk in (3, 7, 14):
        bgemm_case(f"conv1 k=9 wgrad tn256 splitk={sk}", M * H * k * F, dy, x, dw, reps, M=F, N=H, K=M, sAm=1, sAk=F, sBk=H, sBn=1, ldc=k * H,
                   nb2=k, sC2=H, seg=1536, b_shift0=-4, b_shift_step=1, splitk=sk, beta=1.0)
    dw2_ = torch.zeros(H, F, device=DEV)
    dyq = torch.randn(M, H, device=DEV).to(bf)
    for sk in (16, 32, 64):
        bgemm_case(f"conv2 1x1 wgrad tn256 splitk={sk}", M * H * F, dyq, dy, dw2_, reps, M=H, N=F, K=M, sAm=1, sAk=H, sBk=F, sBn=1, ldc=F, splitk=sk, beta=1.0)
    dw3_ = torch.zeros(3 * H, H, device=DEV)
    dy3_ = torch.randn(M, 3 * H, device=DEV).to(bf)
    for sk in (21, 42, 85):
        bgemm_case(f"in_proj wgrad tn256 splitk={sk}", M * H * 3 * H, dy3_, x, dw3_, reps, M=3 * H, N=H, K=M, sAm=1, sAk=3 * H, sBk=H, sBn=1, ldc=H, splitk=sk, beta=1.0)
    # C5 shapes (H 1536, F 6144, 8 x 1536 rows): conv2 1x1 and in-proj weight gradients
    M5, H5, F5 = 12288, 1536, 6144
    dy5 = torch.randn(M5, H5, device=DEV).to(bf)
    x5 = torch.randn(M5, F5, device=DEV).to(bf)
    dw5 = torch.zeros(H5, F5, device=DEV)
    for knob in (1001, 1000):
        lib.fs2_op_set_gemm_variant(knob)
        for sk in (1, 2, 4):
            bgemm_case(f"C5 conv2 1x1 wgrad knob={knob} splitk={sk}", M5 * H5 * F5, dy5, x5, dw5, reps, M=H5, N=F5, K=M5, sAm=1, sAk=H5, sBk=F5, sBn=1, ldc=F5, splitk=sk, beta=1.0)
    lib.fs2_op_set_gemm_variant(1001)
    del dy5, x5, dw5
    lib.fs2_op_set_gemm_variant(1000)
    for sk in (1, 4, 8, 16, 32):
        bgemm_case(f"conv1 k=9 wgrad_splitk={sk}", M * H * k * F, dy, x, dw, reps, M=F, N=H, K=M, sAm=1, sAk=F, sBk=H, sBn=1, ldc=k * H,
                   nb2=k, sC2=H, seg=1536, b_shift0=-4, b_shift_step=1, splitk=sk, beta=1.0)
    dy2 = torch.randn(M, H, device=DEV).to(bf)
    w2 = torch.randn(H, F, device=DEV).to(bf)
    dh = torch.empty(M, F, device=DEV, dtype=bf)
    bgemm_case("conv2 1x1 dgrad (M,256)->(M,1024)", M * H * F, dy2, w2, dh, reps, M=M, N=F, K=H, sAm=H, sAk=1, sBk=F, sBn=1, ldc=F)
    dw2 = torch.zeros(H, F, device=DEV)
    for sk in (16, 32, 64):
        bgemm_case(f"conv2 1x1 wgrad splitk={sk}", M * H * F, dy2, dy, dw2, reps, M=H, N=F, K=M, sAm=1, sAk=H, sBk=F, sBn=1, ldc=F, splitk=sk, beta=1.0)
    x3 = torch.randn(M, H, device=DEV).to(bf)
    dw3 = torch.zeros(3 * H, H, device=DEV)
    dy3 = torch.randn(M, 3 * H, device=DEV).to(bf)
    for sk in (8, 32, 64):
        bgemm_case(f"in_proj wgrad splitk={sk}", M * H * 3 * H, dy3, x3, dw3, reps, M=3 * H, N=H, K=M, sAm=1, sAk=3 * H, sBk=H, sBn=1, ldc=H, splitk=sk, beta=1.0)
    lib.fs2_op_set_gemm_variant(1001)
    B, S, heads, d = 32, 1536, 2, 128
    qkv = torch.randn(B * S, 3 * H, device=DEV).to(bf)
    sc = torch.empty(B, heads, S, S, device=DEV)
    bat = dict(nb1=B, nb2=heads)
    bgemm_case("attn scores QK^T -> fp32", B * heads * S * S * d, qkv, qkv[:, H:], sc, reps, M=S, N=S, K=d, sAm=3 * H, sAk=1, sBk=1, sBn=3 * H,
               ldc=S, sA1=S * 3 * H, sA2=d, sB1=S * 3 * H, sB2=d, sC1=heads * S * S, sC2=S * S, **bat)
    P = torch.randn(B, heads, S, S, device=DEV).to(bf)
    o = torch.empty(B * S, H, device=DEV, dtype=bf)
    bgemm_case("attn O = P V", B * heads * S * S * d, P, qkv[:, 2 * H:], o, reps, M=S, N=d, K=S, sAm=S, sAk=1, sBk=3 * H, sBn=1, ldc=H,
               sA1=heads * S * S, sA2=S * S, sB1=S * 3 * H, sB2=d, sC1=S * H, sC2=d, **bat)
    dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=bf)
    bgemm_case("attn dV = P^T dO", B * heads * S * S * d, P, o, dqkv[:, 2 * H:], reps, M=S, N=d, K=S, sAm=1, sAk=S, sBk=H, sBn=1, ldc=3 * H,
               sA1=heads * S * S, sA2=S * S, sB1=S * H, sB2=d, sC1=S * 3 * H, sC2=d, **bat)


def flash_bwd_case(name, B, S, H, heads, reps):
    bf = torch.bfloat16
    qkv = (0.5 * torch.randn(B * S, 3 * H, device=DEV)).to(bf)
    dout = torch.randn(B * S, H, device=DEV).to(bf)
    lse = torch.zeros(B, heads, S, device=DEV) + 5.0
    delta = torch.zeros(B, heads, S, device=DEV)
    dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=bf)
    for knobs in ((900, 902), (900, 903), (901, 902), (905, 903), (900, 907), (900, 908), (905, 907), (905, 908)):  # blocks per wave of the (dK dV, dQ) launches (905 / 907: 3, 908: 4)
        for knob in knobs:
            lib.fs2_op_set_gemm_variant(knob)
        t = timeit(lambda st: lib.fs2_op_attention_bwd(BF16, p(qkv), p(dout), p(lse), p(delta), None, p(dqkv), B, S, H, heads,
                                                       C.c_float(0.0), C.c_uint64(0), C.c_uint64(0), st), reps)
        fl = 5 * 2.0 * B * S * S * H  # dV, dP, dS-products: five S x S x d GEMMs are the useful work
        print(f"{name:30s} knobs={knobs[0]},{knobs[1]}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF useful ({fl/t/2.5e15*100:4.1f}% of 2.5 PF)")
    lib.fs2_op_set_gemm_variant(909)
    lib.fs2_op_set_gemm_variant(904)


def predictor_case(name, B, S, nl, reps):
    """Whole dense VariancePredictor (nl x conv k=3 + ReLU + LN, head) as one launch."""
    H, k = 256, 3
    x = torch.randn(B * S, H, device=DEV).to(torch.bfloat16)
    w = (torch.randn(nl, H, k * H, device=DEV) * (k * H) ** -0.5).to(torch.bfloat16)
    b, g, be = (torch.randn(nl, H, device=DEV) for _ in range(3))
    hw = torch.randn(H, device=DEV)
    pred = torch.empty(B * S, device=DEV)
    scratch = torch.empty(nl * H * k * H * 2, dtype=torch.uint8, device=DEV)
    t = timeit(lambda st: lib.fs2_op_predictor(BF16, p(x), p(w), p(b), p(g), p(be), p(hw), C.c_float(0.1), None, p(pred),
                                               p(scratch), B, S, H, nl, k, st), reps)
    fl = 2.0 * B * S * H * H * k * nl
    print(f"{name:28s} B={B} S={S} layers={nl}  {t*1e6:8.1f} us (incl. {nl} weight-pack launches)  {fl/t/1e12:7.1f} TF  "
          f"({fl/t/2.5e15*100:4.1f}% of 2.5 PF)")


def predictor_dw_case(name, B, S, nl, reps):
    """Whole depth-wise VariancePredictor (nl x [dw k=3 + pointwise + ReLU + LN], head) as one launch (r06)."""
    H = 256
    x = torch.randn(B * S, H, device=DEV).to(torch.bfloat16)
    w = (torch.randn(nl, H, H, device=DEV) * H ** -0.5).to(torch.bfloat16)
    dw = torch.randn(nl, 3, H, device=DEV) * 3 ** -0.5
    dwb, b, g, be = (torch.randn(nl, H, device=DEV) for _ in range(4))
    hw = torch.randn(H, device=DEV)
    pred = torch.empty(B * S, device=DEV)
    scratch = torch.empty(nl * H * H * 2, dtype=torch.uint8, device=DEV)
    t = timeit(lambda st: lib.fs2_op_predictor_dw(BF16, p(x), p(dw), p(dwb), p(w), p(b), p(g), p(be), p(hw), C.c_float(0.1), None, p(pred),
                                                  p(scratch), B, S, H, nl, st), reps)
    fl = 2.0 * B * S * H * (H + 3) * nl
    print(f"{name:28s} B={B} S={S} layers={nl}  {t*1e6:8.1f} us (incl. {nl} weight-pack launches)  {fl/t/1e12:7.1f} TF  "
          f"({fl/t/2.5e15*100:4.1f}% of 2.5 PF)")


def attn_case(name, B, S, H, heads, reps):
    qkv = torch.randn(B * S, 3 * H, device=DEV).to(torch.bfloat16)
    mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    bb = C.c_size_t()
    vb = lib.fs2_op_attention_scratch_bytes(BF16, B, S, H, heads, C.byref(bb))
    vt = torch.empty(vb, dtype=torch.uint8, device=DEV)
    bits = torch.empty(bb.value, dtype=torch.uint8, device=DEV)
    t = timeit(lambda st: lib.fs2_op_attention(BF16, p(qkv), p(mask), p(out), p(vt), p(bits), B, S, H, heads, st), reps)
    fl = 4.0 * B * S * S * H
    print(f"{name:28s} B={B} S={S} H={H} heads={heads}  {t*1e6:8.1f} us (incl. mask bits + V^T staging)  {fl/t/1e12:7.1f} TF")


def attn_out_case(name, B, S, reps):
    """Encoder block tail: attention + out-projection + residual + LayerNorm as one launch (r06) against the two launches it replaces."""
    H, heads = 256, 2
    qkv = torch.randn(B * S, 3 * H, device=DEV).to(torch.bfloat16)
    mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
    w = (torch.randn(H, H, device=DEV) * H ** -0.5).to(torch.bfloat16)
    bias, g, be = (torch.randn(H, device=DEV) for _ in range(3))
    res = torch.randn(B * S, H, device=DEV).to(torch.bfloat16)
    out = torch.empty_like(res)
    att = torch.empty_like(res)
    tmp = torch.empty_like(res)
    scratch = torch.empty(H * H * 2 + B * ((S + 63) // 64) * 8, dtype=torch.uint8, device=DEV)
    bb = C.c_size_t()
    vb = lib.fs2_op_attention_scratch_bytes(BF16, B, S, H, heads, C.byref(bb))
    vt = torch.empty(max(vb, 16), dtype=torch.uint8, device=DEV)
    bits = torch.empty(bb.value, dtype=torch.uint8, device=DEV)
    t1 = timeit(lambda st: lib.fs2_op_attn_out_ln(BF16, p(qkv), p(mask), p(w), p(bias), p(res), p(g), p(be), p(out), p(scratch), B, S, H, heads, st), reps)
    def two(st):
        lib.fs2_op_attention(BF16, p(qkv), p(mask), p(att), p(vt), p(bits), B, S, H, heads, st)
        lib.fs2_op_gemm_ln(BF16, p(att), p(w), p(bias), p(res), p(g), p(be), None, C.c_float(0.0), None, None, p(out), p(tmp), B * S, H, H, 1, B * S, 0, st)
    t2 = timeit(two, reps)
    fl = 4.0 * B * S * S * H + 2.0 * B * S * H * H
    print(f"{name:28s} B={B} S={S}  one launch {t1 * 1e6:7.1f} us (incl. mask bits + weight pack)   two launches {t2 * 1e6:7.1f} us (incl. mask bits)   "
          f"{fl / t1 / 1e12:6.1f} TF", flush=True)


def dwconv_case(name, B, S, Cc, k, reps):
    """Depth-wise conv launch (rowops.hip dwconv_kernel): HBM-bound, priced against 8 TB/s on read + write of the activations."""
    x = torch.randn(B * S, Cc, device=DEV).to(torch.bfloat16)
    w = torch.randn(Cc, k, device=DEV) * k ** -0.5
    b = torch.randn(Cc, device=DEV)
    y = torch.empty_like(x)
    t = timeit(lambda st: lib.fs2_op_dwconv(BF16, p(x), p(w), p(b), p(y), B, S, Cc, k, st), reps)
    by = 2.0 * x.numel() * 2
    print(f"{name:28s} B={B} S={S} C={Cc} k={k:2d}   {t * 1e6:8.1f} us   {by / t / 1e12:6.2f} TB/s  ({by / t / 8e12 * 100:4.1f}% of 8 TB/s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--shape", default="", help="attn: one extra case B,S,H,heads")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the case name")
    ap.add_argument("--flush", type=int, default=0, help="MB of unrelated memory touched between reps (cold caches); 0 = back-to-back reps")
    a = ap.parse_args()
    if a.flush:
        FLUSH[0] = torch.zeros(a.flush * 1024 * 1024 // 4, device=DEV)
    for k in [k for k in os.environ.get("FS2_GEMM_KNOBS", "").split(",") if k]:  # A/B knobs that are not kernel-family variants (210 / 211, 220 / 221 ...)
        if lib.fs2_op_set_gemm_variant(int(k)) != 0:
            raise SystemExit(f"knob {k} rejected")
    global gemm_case, attn_case, gemm_ln_case, bgemm_case
    if a.only:
        g0, a0, l0, b0 = gemm_case, attn_case, gemm_ln_case, bgemm_case
        bgemm_case = lambda name, *r, **k: b0(name, *r, **k) if a.only in name else None
        gemm_ln_case = lambda name, *r, **k: l0(name, *r, **k) if a.only in name else None
        gemm_case = lambda name, *r: g0(name, *r) if a.only in name else None
        attn_case = lambda name, *r: a0(name, *r) if a.only in name else None
    variants = [0, 3, 4, 5, 6, 7] if a.variant < 0 else [a.variant]
    if a.what in ("gemm", "all"):
        for v in variants:
            gemm_case("dec conv1 k=9", 49152, 1024, 256, 9, 1536, a.reps, v)
            gemm_case("same as plain GEMM", 49152, 1024, 2304, 1, 49152, a.reps, v)
            gemm_case("var-pred conv k=3", 49152, 256, 256, 3, 1536, a.reps, v)
            gemm_case("dec conv2 1x1", 49152, 256, 1024, 1, 49152, a.reps, v)
            gemm_case("dec in_proj", 49152, 768, 256, 1, 49152, a.reps, v)
            gemm_case("dec out_proj", 49152, 256, 256, 1, 49152, a.reps, v)
            gemm_case("enc conv1 k=9", 8192, 1024, 256, 9, 256, a.reps, v)
            gemm_case("enc conv2 1x1", 8192, 256, 1024, 1, 8192, a.reps, v)
            gemm_case("enc out_proj", 8192, 256, 256, 1, 8192, a.reps, v)
            gemm_case("enc in_proj", 8192, 768, 256, 1, 8192, a.reps, v)
            gemm_case("dur-pred conv k=3", 8192, 256, 256, 3, 256, a.reps, v)
            gemm_case("square 4096^3", 4096, 4096, 4096, 1, 4096, a.reps, v)
    if a.what == "splitk":  # the same data-gradient convs one pass vs split over K (fs2_op_gemm_splitk)
        for name, M, N, Cin, taps, S in (("c2 enc conv1 dgrad", 8192, 256, 1024, 9, 256), ("c5 enc conv1 dgrad", 2048, 1024, 4096, 9, 256)):
            gemm_case(name + " one pass", M, N, Cin, taps, S, a.reps, 0)
            for ks in (2, 4, 8):
                splitk_case(name + f" split {ks}", M, N, Cin, taps, S, ks, a.reps)
    if a.what == "dgrad":  # data-gradient convs of the training step: long reductions (K = taps x filter) at few rows
        for v in ([0, 6, 7, 3, 4, 5] if a.variant < 0 else [a.variant]):
            gemm_case("c2 enc conv1 dgrad", 8192, 256, 1024, 9, 256, a.reps, v)
            gemm_case("c2 dec conv1 dgrad", 49152, 256, 1024, 9, 1536, a.reps, v)
            gemm_case("c2 enc conv2 dgrad", 8192, 1024, 256, 1, 8192, a.reps, v)
            gemm_case("c5 enc conv1 dgrad", 2048, 1024, 4096, 9, 256, a.reps, v)
    if a.what in ("ln", "all"):
        for v in ([0, 6, 7, 3, 4] if a.variant < 0 else [a.variant]):
            gemm_ln_case("var-pred conv k=3 +LN", 49152, 256, 256, 3, 1536, a.reps, v, res=False, relu=True)
            gemm_ln_case("dec conv2 1x1 +res+LN", 49152, 256, 1024, 1, 49152, a.reps, v)
            gemm_ln_case("dec out_proj +res+LN", 49152, 256, 256, 1, 49152, a.reps, v)
            gemm_ln_case("enc conv2 1x1 +res+LN", 8192, 256, 1024, 1, 8192, a.reps, v)
            gemm_ln_case("enc out_proj +res+LN", 8192, 256, 256, 1, 8192, a.reps, v)
            gemm_ln_case("dur-pred conv k=3 +LN", 8192, 256, 256, 3, 256, a.reps, v, res=False, relu=True)
    if a.what in ("encmha",):
        attn_out_case("enc attention + out-proj + LN", 32, 256, a.reps)
        attn_out_case("enc attention + out-proj + LN", 8, 256, a.reps)
        attn_out_case("enc attention + out-proj + LN", 32, 128, a.reps)
        for S in (384, 512, 768, 1000):  # decoder-side lengths below the pipelined attention kernel's range (S < 1024)
            attn_out_case("attention + out-proj + LN", 32, S, a.reps)
        attn_out_case("attention + out-proj + LN", 4, 640, a.reps)
    if a.what in ("rows",):  # the LS-76M depth-wise convs: the variance predictors' (k = 3 / 5) and the decoder's
        for k in (3, 5, 9, 13, 17, 21, 25, 31):
            dwconv_case("c3 dwconv T rows", 32, 1536, 768, k, a.reps)
        for k in (5, 25, 13, 9):
            dwconv_case("c3 dwconv enc rows", 32, 256, 768, k, a.reps)
        for k in (3, 9, 13, 17, 21):  # the reference-default model (H = 256): 768 tiles of 256 rows = one round of three workgroups per CU
            dwconv_case("ref-default dwconv T rows", 32, 1536, 256, k, a.reps)
        for k in (5, 25, 13, 9):
            dwconv_case("ref-default dwconv enc rows", 32, 256, 256, k, a.reps)
    if a.what in ("c3gemm",):  # the LS-76M decoder's pointwise GEMMs, by tile height
        for v in ([3, 4, 5] if a.variant < 0 else [a.variant]):
            gemm_case("c3 in_proj", 49152, 2304, 768, 1, 49152, a.reps, v)
            gemm_case("c3 pw1", 49152, 3072, 768, 1, 49152, a.reps, v)
            gemm_case("c3 conv2", 49152, 768, 3072, 1, 49152, a.reps, v)
            gemm_case("c3 out_proj", 49152, 768, 768, 1, 49152, a.reps, v)
            gemm_case("c5 in_proj", 12288, 3072, 1024, 1, 12288, a.reps, v)
            gemm_case("c5 pw1", 12288, 4096, 1024, 1, 12288, a.reps, v)
            gemm_case("c5 conv2", 12288, 1024, 4096, 1, 12288, a.reps, v)
            gemm_case("c5 out_proj", 12288, 1024, 1024, 1, 12288, a.reps, v)
            gemm_case("c5 conv1 k=9", 12288, 4096, 1024, 9, 1536, a.reps, v)
            gemm_case("c5 conv1 as plain GEMM", 12288, 4096, 9216, 1, 12288, a.reps, v)
            gemm_case("square 4096^3", 4096, 4096, 4096, 1, 4096, a.reps, v)
    if a.what in ("flash",):
        flash_bwd_case("c2 decoder attention backward", 32, 1536, 256, 2, a.reps)
        flash_bwd_case("c3 decoder attention backward", 32, 1536, 768, 6, a.reps)
        flash_bwd_case("c2 encoder attention backward", 32, 256, 256, 2, a.reps)
    if a.what in ("bwd",):
        bwd_cases(a.reps)
    if a.what in ("pred", "all"):
        predictor_case("variance predictor fused", 32, 1536, 5, a.reps)
        predictor_case("duration predictor fused", 32, 256, 2, a.reps)
        predictor_dw_case("dw variance predictor fused", 32, 1536, 5, a.reps)
        predictor_dw_case("dw duration predictor fused", 32, 256, 2, a.reps)
    if a.what in ("attn", "all"):
        if a.variant >= 1200: lib.fs2_op_set_gemm_variant(a.variant)   # 1200 phase-serial kernel, 1201 / 1202 pipelined, 1203 by size
        if a.shape: attn_case("custom attention", *[int(x) for x in a.shape.split(",")], a.reps)
        attn_case("c3 decoder attention", 32, 1536, 768, 6, a.reps)
        attn_case("c2 decoder attention", 32, 1536, 256, 2, a.reps)
        attn_case("c5 decoder attention", 8, 1536, 1024, 8, a.reps)
        attn_case("c2 encoder attention", 32, 256, 256, 2, a.reps)
        lib.fs2_op_set_gemm_variant(1203)


if __name__ == "__main__":
    main()
