"""GPU: the backward operators of the training step (csrc/bgemm.hip, csrc/backward.hip) one by one through
the C ABI against plain fp32/fp64 torch CPU references of the same op (autograd where the op is a derivative).
Tolerances: GEMM-shaped ops 2e-5 relative to the result's scale (fp32 MFMA, different summation order); row ops 1e-5."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lightningfastspeech2_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32 = _lib.FS2_F32


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bgemm(desc_kw, A, B, Cout, bias=None, dtype=F32):
    lib = _lib.load()
    d = _lib.BGemmDescC()
    base = dict(nb1=1, nb2=1, alpha=1.0, beta=0.0, splitk=1, taps=1,
                c_dtype=_lib.FS2_BF16 if Cout.dtype == torch.bfloat16 else F32)
    base.update(desc_kw)
    for k, v in base.items():
        setattr(d, k, v)
    ws = None
    nbytes = lib.fs2_op_bgemm_ws_bytes(C.byref(d))
    if nbytes:
        ws = torch.empty(nbytes // 4, device=DEV)
    _lib.check(lib.fs2_op_bgemm(dtype, C.byref(d), p(A), p(B), p(Cout), p(bias), p(ws), st()), what="bgemm")
    return Cout


def close(got, want, rel=2e-5):
    want = want.float()
    scale = float(want.abs().max()) + 1e-30
    err = float((got.cpu().float() - want).abs().max())
    assert err <= rel * scale, (err, scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (200, 72, 50), (1, 1, 1), (513, 259, 131), (64, 300, 7)])
def test_bgemm_nt_nn_tn_layouts(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, b = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    want = a.double() @ b.double()
    bias = torch.randn(N, generator=g)
    # NT: A (M, K) k-contiguous, B stored (N, K)
    c = torch.empty(M, N, device=DEV)
    bgemm(dict(M=M, N=N, K=K, sAm=K, sAk=1, sBk=1, sBn=K, ldc=N), a.to(DEV), b.t().contiguous().to(DEV), c, bias.to(DEV))
    close(c, want + bias.double())
    # NN: B stored (K, N)
    c = torch.full((M, N), 3.0, device=DEV)
    bgemm(dict(M=M, N=N, K=K, sAm=K, sAk=1, sBk=N, sBn=1, ldc=N, alpha=0.5, beta=2.0), a.to(DEV), b.to(DEV), c)
    close(c, 0.5 * want + 6.0)
    # TN: A stored (K, M), B stored (K, N), split-K
    for sk in (1, 3):
        c = torch.empty(M, N, device=DEV)
        bgemm(dict(M=M, N=N, K=K, sAm=1, sAk=M, sBk=N, sBn=1, ldc=N, splitk=sk), a.t().contiguous().to(DEV), b.to(DEV), c)
        close(c, want)


def test_bgemm_batched_attention_shapes():
    """S = Q K^T and O = P V straight out of / into the (B*S, 3H) / (B*S, H) projections' layout, and the TN products of
    the attention backward (dV = P^T dO, dK = dS^T Q)."""
    B, S, H, heads = 3, 37, 64, 2
    d = H // heads
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * S, 3 * H, generator=g)
    q, k, v = (t.view(B, S, heads, d).transpose(1, 2) for t in qkv.view(B, S, 3 * H).split(H, dim=-1))
    scores = torch.empty(B, heads, S, S, device=DEV)
    dq = qkv.to(DEV)
    bgemm(dict(M=S, N=S, K=d, sAm=3 * H, sAk=1, sBk=1, sBn=3 * H, ldc=S, nb1=B, nb2=heads, sA1=S * 3 * H, sA2=d,
               sB1=S * 3 * H, sB2=d, sC1=heads * S * S, sC2=S * S, alpha=0.25), dq, dq[:, H:], scores)
    close(scores, 0.25 * (q.double() @ k.double().transpose(-1, -2)))
    P = torch.softmax(torch.randn(B, heads, S, S, generator=g), dim=-1)
    o = torch.empty(B * S, H, device=DEV)
    bgemm(dict(M=S, N=d, K=S, sAm=S, sAk=1, sBk=3 * H, sBn=1, ldc=H, nb1=B, nb2=heads, sA1=heads * S * S, sA2=S * S,
               sB1=S * 3 * H, sB2=d, sC1=S * H, sC2=d), P.to(DEV), dq[:, 2 * H:], o)
    close(o, (P.double() @ v.double()).transpose(1, 2).reshape(B * S, H))
    do = torch.randn(B * S, H, generator=g)
    dqkv = torch.zeros(B * S, 3 * H, device=DEV)
    bgemm(dict(M=S, N=d, K=S, sAm=1, sAk=S, sBk=H, sBn=1, ldc=3 * H, nb1=B, nb2=heads, sA1=heads * S * S, sA2=S * S,
               sB1=S * H, sB2=d, sC1=S * 3 * H, sC2=d), P.to(DEV), do.to(DEV), dqkv[:, 2 * H:])
    dov = do.view(B, S, heads, d).transpose(1, 2).double()
    want_dv = (P.double().transpose(-1, -2) @ dov).transpose(1, 2).reshape(B * S, H)
    close(dqkv[:, 2 * H:], want_dv)
    assert not dqkv[:, :2 * H].any()


@pytest.mark.parametrize("taps,Cin,N,S,B", [(9, 32, 48, 40, 3), (3, 20, 17, 11, 2), (1, 16, 16, 8, 2), (5, 64, 130, 150, 2)])
def test_bgemm_conv_dgrad_and_wgrad(taps, Cin, N, S, B):
    """The implicit-conv forms against autograd of F.conv1d(padding='same') over (B, C, S)."""
    g = torch.Generator().manual_seed(taps * 100 + Cin)
    x = torch.randn(B, S, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(N, Cin, taps, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(x.transpose(1, 2), w, padding="same").transpose(1, 2)  # (B, S, N)
    dy = torch.randn(B, S, N, generator=g, dtype=torch.float64)
    y.backward(dy)
    pad = (taps - 1) // 2
    w_tm = w.detach().permute(0, 2, 1).reshape(N, taps * Cin).float().contiguous().to(DEV)  # tap-major rows, as the engine holds them
    dyd = dy.float().reshape(B * S, N).contiguous().to(DEV)
    xd = x.detach().float().reshape(B * S, Cin).contiguous().to(DEV)
    dx = torch.empty(B * S, Cin, device=DEV)
    bgemm(dict(M=B * S, N=Cin, K=taps * N, sAm=N, sAk=1, sBk=taps * Cin, sBn=1, ldc=Cin, seg=S, taps=taps, Kin=N,
               a_shift0=pad, a_shift_step=-1, sBtap=Cin), dyd, w_tm, dx)
    close(dx, x.grad.reshape(B * S, Cin))
    for sk in (1, 4):
        dw = torch.zeros(N, taps * Cin, device=DEV)
        bgemm(dict(M=N, N=Cin, K=B * S, sAm=1, sAk=N, sBk=Cin, sBn=1, ldc=taps * Cin, nb2=taps, sC2=Cin, seg=S,
                   b_shift0=-pad, b_shift_step=1, splitk=sk), dyd, xd, dw)
        close(dw, w.grad.permute(0, 2, 1).reshape(N, taps * Cin))


@pytest.mark.parametrize("M,H,res", [(100, 256, True), (7, 48, False), (257, 768, True), (64, 1024, False), (33, 300, False)])
def test_layernorm_bwd(M, H, res):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + H)
    z = torch.randn(M, H, generator=g, dtype=torch.float64, requires_grad=True)
    r = torch.randn(M, H, generator=g, dtype=torch.float64) if res else None
    gam = torch.randn(H, generator=g, dtype=torch.float64, requires_grad=True)
    bet = torch.randn(H, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(M, H, generator=g, dtype=torch.float64)
    F.layer_norm(z + r if res else z, (H,), gam, bet, 1e-5).backward(dy)
    nparts = lib.fs2_op_layernorm_bwd_parts(M)
    dz = torch.empty(M, H, device=DEV)
    part = torch.empty(nparts, 3, H, device=DEV)
    zd, rd, dyd, gd = z.detach().float().to(DEV), r.float().to(DEV) if res else None, dy.float().to(DEV), gam.detach().float().to(DEV)
    _lib.check(lib.fs2_op_layernorm_bwd(F32, p(zd), p(rd), p(dyd), p(gd), p(dz), p(part), M, H, 0, st()))
    close(dz, z.grad, 2e-5)
    out = torch.full((1, 3 * H), 1.0, device=DEV)
    ws = torch.zeros(max(1, lib.fs2_op_col_sum_ws_bytes(nparts, 3 * H, 0) // 4), device=DEV)
    _lib.check(lib.fs2_op_col_sum(F32, p(part), p(out), p(ws), nparts, 3 * H, 3 * H, 0, 1, 1.0, st()))
    close(out[0, :H] - 1.0, gam.grad, 2e-5)
    close(out[0, H:2 * H] - 1.0, bet.grad, 2e-5)
    close(out[0, 2 * H:] - 1.0, z.grad.sum(0), 5e-5)
    if not res:  # relu_mask: z doubles as a ReLU output - dz is zeroed where z <= 0, and so is its column sum
        _lib.check(lib.fs2_op_layernorm_bwd(F32, p(zd), None, p(dyd), p(gd), p(dz), p(part), M, H, 1, st()))
        want = torch.where(z.detach() > 0, z.grad, torch.zeros((), dtype=torch.float64))
        close(dz, want, 2e-5)
        close(part.sum(0)[2], want.sum(0), 5e-5)


def test_col_sum_segments():
    lib = _lib.load()
    x = torch.randn(6 * 700, 100)
    out = torch.empty(6, 100, device=DEV)
    ws = torch.zeros(lib.fs2_op_col_sum_ws_bytes(6 * 700, 100, 700) // 4, device=DEV)
    xd = x.to(DEV)
    _lib.check(lib.fs2_op_col_sum(F32, p(xd), p(out), p(ws), 6 * 700, 100, 100, 700, 0, 0.5, st()))
    close(out, 0.5 * x.double().view(6, 700, 100).sum(1), 1e-5)


@pytest.mark.parametrize("B,heads,S", [(2, 2, 37), (1, 1, 1), (3, 2, 200)])
def test_softmax_fwd_bwd(B, heads, S):
    lib = _lib.load()
    g = torch.Generator().manual_seed(S)
    s = torch.randn(B, heads, S, S, generator=g, dtype=torch.float64, requires_grad=True)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    pad = torch.arange(S)[None, :] >= lens[:, None]
    pr = torch.softmax((0.3 * s).masked_fill(pad[:, None, None, :], float("-inf")), dim=-1)
    dp = torch.randn(B, heads, S, S, generator=g, dtype=torch.float64)
    pr.backward(dp)
    sd = s.detach().float().to(DEV)
    padd = pad.to(torch.uint8).to(DEV)
    _lib.check(lib.fs2_op_softmax_fwd(F32, p(sd), p(padd), p(sd), B, heads, S, 0.3, st()))
    close(sd, pr.detach(), 1e-5)
    dd = dp.float().to(DEV)
    _lib.check(lib.fs2_op_softmax_bwd(F32, p(dd), p(sd), p(dd), B, heads, S, 0.3, st()))
    close(dd, s.grad, 2e-5)


def test_scatter_rows_and_regulate_bwd():
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    R, H, V = 3000, 72, 40
    idx = torch.randint(0, V, (R,), generator=g)
    x = torch.randn(R, H, generator=g)
    for kind in ("i32", "i64"):
        table = torch.ones(V, H, device=DEV)
        i32 = idx.int().to(DEV) if kind == "i32" else None
        i64 = idx.to(DEV) if kind == "i64" else None
        xd = x.to(DEV)
        _lib.check(lib.fs2_op_scatter_rows(F32, p(xd), p(i32), p(i64), p(table), None, R, H, V, 0, st()))
        want = torch.ones(V, H, dtype=torch.float64).index_add_(0, idx, x.double())
        want[0] = 1.0
        close(table, want, 1e-5)
    B, L, T, H = 3, 9, 31, 40
    dur = torch.randint(0, 6, (B, L), generator=g)
    dur[0, :] = 5  # 45 frames > T: truncated
    cum = dur.cumsum(1).int()
    dy = torch.randn(B * T, H, generator=g)
    dx = torch.empty(B * L, H, device=DEV)
    dyd, cumd = dy.to(DEV), cum.to(DEV)
    _lib.check(lib.fs2_op_regulate_bwd(F32, p(dyd), p(cumd), p(dx), B, L, T, H, st()))
    want = torch.zeros(B, L, H, dtype=torch.float64)
    for b in range(B):
        t = 0
        for ph in range(L):
            for _ in range(int(dur[b, ph])):
                if t < T:
                    want[b, ph] += dy[b * T + t].double()
                t += 1
    close(dx, want.view(B * L, H), 1e-5)


@pytest.mark.parametrize("kind,inner", [(0, 80), (1, 1)])
def test_masked_loss_bwd(kind, inner):
    lib = _lib.load()
    g = torch.Generator().manual_seed(kind)
    rows = 500
    pred = torch.randn(rows, inner, generator=g, dtype=torch.float64, requires_grad=True)
    truth = torch.randn(rows, inner, generator=g, dtype=torch.float64)
    mask = torch.rand(rows, generator=g) < 0.3
    sel = ~mask
    loss = (pred[sel] - truth[sel]).abs().mean() if kind == 0 else ((pred[sel] - truth[sel]) ** 2).mean()
    (1.7 * loss).backward()
    ws = torch.zeros(lib.fs2_op_masked_loss_ws_bytes(), dtype=torch.uint8, device=DEV)
    stat = torch.empty(2, device=DEV)
    pd, td, md = pred.detach().float().to(DEV), truth.float().to(DEV), mask.to(torch.uint8).to(DEV)
    _lib.check(lib.fs2_op_masked_loss(p(pd), p(td), 0, p(md), rows, inner, kind, p(ws), p(stat), st()))
    dpred = torch.empty(rows, inner, device=DEV)
    _lib.check(lib.fs2_op_masked_loss_bwd(p(pd), p(td), 0, p(md), p(stat), p(dpred), rows, inner, kind, 1.7, st()))
    close(dpred, pred.grad, 1e-5)


def test_adamw_matches_torch_with_clipping():
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    n = 10003  # 16-byte body + element tail
    w = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(w.clone().double())
    opt = torch.optim.AdamW([ref], lr=2e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    wd, m, v = w.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ws = torch.empty(lib.fs2_op_sum_sq_ws_bytes(n) // 4, device=DEV)
    nsq = torch.empty(1, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.001)  # step 2 clips, the others do not
        ref.grad = grad.double().clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = grad.to(DEV)
        _lib.check(lib.fs2_op_sum_sq(p(gd), n, p(ws), p(nsq), st()))
        assert abs(float(nsq) - float((grad.double() ** 2).sum())) <= 1e-5 * float((grad.double() ** 2).sum())
        if step == 2:
            _lib.check(lib.fs2_op_adamw(p(wd), p(gd), p(m), p(v), n, 2e-4, 0.9, 0.98, 1e-8, 0.01, step, p(nsq), 1.0, 1.0, st()))
        else:  # the variant that also leaves the bf16 shadow of the new weights
            sh = torch.empty(n, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.fs2_op_adamw_shadow(p(wd), p(gd), p(m), p(v), p(sh), n, 2e-4, 0.9, 0.98, 1e-8, 0.01, step, p(nsq), 1.0, 1.0, st()))
            assert torch.equal(sh, wd.to(torch.bfloat16))
        close(wd, ref.detach(), 1e-6)


def test_ew_ops():
    lib = _lib.load()
    a, b = torch.randn(5003), torch.randn(5003)  # vector body + scalar tail
    a, b = a[:5000], b[:5000]
    out = torch.empty(5000, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    _lib.check(lib.fs2_op_ew(F32, 0, p(ad), p(bd), p(out), 5000, 2.0, -1.0, st()))
    assert torch.equal(out.cpu(), 2.0 * a - b)
    _lib.check(lib.fs2_op_ew(F32, 1, p(ad), p(bd), p(out), 5000, 0.0, 0.0, st()))
    assert torch.equal(out.cpu(), torch.where(b > 0, a, torch.zeros(())))
    n = 4099  # odd length: body of 4096 + tail of 3; and an unaligned view (scalar path only)
    ad, bd, out = torch.randn(n + 1, device=DEV), torch.randn(n + 1, device=DEV), torch.empty(n + 1, device=DEV)
    _lib.check(lib.fs2_op_ew(F32, 0, p(ad), p(bd), p(out), n, 1.5, 0.5, st()))
    assert torch.allclose(out[:n].cpu(), (1.5 * ad[:n] + 0.5 * bd[:n]).cpu(), rtol=1e-6, atol=1e-6)  # fma vs two roundings
    _lib.check(lib.fs2_op_ew(F32, 1, p(ad[1:]), p(bd[1:]), p(out[1:]), n, 0.0, 0.0, st()))
    assert torch.equal(out[1:].cpu(), torch.where(bd[1:] > 0, ad[1:], torch.zeros((), device=DEV)).cpu())


# ---- bf16 operands (the mixed-precision training path): same products, inputs rounded to bf16, fp32 accumulation ----
BF = _lib.FS2_BF16


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 72, 50), (1, 1, 1), (513, 264, 136), (64, 300, 7)])
def test_bgemm_bf16_layouts(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, b = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(K, N, generator=g))
    want = a.double() @ b.double()
    bias = torch.randn(N, generator=g)
    c = torch.empty(M, N, device=DEV)  # fp32 out
    bgemm(dict(M=M, N=N, K=K, sAm=K, sAk=1, sBk=1, sBn=K, ldc=N), a.to(DEV), b.t().contiguous().to(DEV), c, bias.to(DEV), dtype=BF)
    close(c, want + bias.double(), 1e-5)
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)  # bf16 out, NN
    bgemm(dict(M=M, N=N, K=K, sAm=K, sAk=1, sBk=N, sBn=1, ldc=N), a.to(DEV), b.to(DEV), c, dtype=BF)
    close(c, want, 5e-3)
    for sk in (1, 3):  # TN
        c = torch.full((M, N), 2.0, device=DEV)
        bgemm(dict(M=M, N=N, K=K, sAm=1, sAk=M, sBk=N, sBn=1, ldc=N, splitk=sk, beta=1.0), a.t().contiguous().to(DEV), b.to(DEV), c, dtype=BF)
        close(c, want + 2.0, 1e-5)


@pytest.mark.parametrize("taps,Cin,N,S,B", [(9, 32, 48, 40, 3), (3, 24, 16, 11, 2), (5, 64, 136, 150, 2), (3, 20, 17, 11, 2)])
def test_bgemm_bf16_conv_forms(taps, Cin, N, S, B):
    g = torch.Generator().manual_seed(taps * 100 + Cin)
    x = _bf(torch.randn(B, S, Cin, generator=g)).double().requires_grad_(True)
    w = _bf(torch.randn(N, Cin, taps, generator=g)).double().requires_grad_(True)
    y = F.conv1d(x.transpose(1, 2), w, padding="same").transpose(1, 2)
    dy = _bf(torch.randn(B, S, N, generator=g)).double()
    y.backward(dy)
    pad = (taps - 1) // 2
    w_tm = _bf(w.detach().permute(0, 2, 1).reshape(N, taps * Cin)).contiguous().to(DEV)
    dyd = _bf(dy.reshape(B * S, N)).contiguous().to(DEV)
    xd = _bf(x.detach().reshape(B * S, Cin)).contiguous().to(DEV)
    dx = torch.empty(B * S, Cin, device=DEV)
    bgemm(dict(M=B * S, N=Cin, K=taps * N, sAm=N, sAk=1, sBk=taps * Cin, sBn=1, ldc=Cin, seg=S, taps=taps, Kin=N,
               a_shift0=pad, a_shift_step=-1, sBtap=Cin), dyd, w_tm, dx, dtype=BF)
    close(dx, x.grad.reshape(B * S, Cin), 1e-5)
    for sk in (1, 4):
        dw = torch.zeros(N, taps * Cin, device=DEV)
        bgemm(dict(M=N, N=Cin, K=B * S, sAm=1, sAk=N, sBk=Cin, sBn=1, ldc=taps * Cin, nb2=taps, sC2=Cin, seg=S,
                   b_shift0=-pad, b_shift_step=1, splitk=sk), dyd, xd, dw, dtype=BF)
        close(dw, w.grad.permute(0, 2, 1).reshape(N, taps * Cin), 1e-5)


def test_row_ops_bf16():
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    M, H = 300, 256
    z, r, dy = (_bf(torch.randn(M, H, generator=g)) for _ in range(3))
    gam = torch.randn(H, generator=g)
    zz = (z.double() + r.double()).requires_grad_(True)
    gg = gam.double().requires_grad_(True)
    F.layer_norm(zz, (H,), gg, torch.zeros(H, dtype=torch.float64), 1e-5).backward(dy.double())
    nparts = lib.fs2_op_layernorm_bwd_parts(M)
    dz = torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
    part = torch.empty(nparts, 3, H, device=DEV)
    zd, rd, dyd, gd = z.to(DEV), r.to(DEV), dy.to(DEV), gam.to(DEV)
    _lib.check(lib.fs2_op_layernorm_bwd(BF, p(zd), p(rd), p(dyd), p(gd), p(dz), p(part), M, H, 0, st()))
    close(dz, zz.grad, 5e-3)
    close(part.sum(0)[0], gg.grad, 1e-4)
    # softmax: fp32 scores -> bf16 probabilities; fp32 dP + bf16 P -> bf16 dS
    B, heads, S = 2, 2, 70
    s = torch.randn(B, heads, S, S, generator=g)
    pr = torch.softmax(0.5 * s.double(), dim=-1)
    sd = s.to(DEV)
    pd = torch.empty(B, heads, S, S, device=DEV, dtype=torch.bfloat16)
    _lib.check(lib.fs2_op_softmax_fwd(BF, p(sd), None, p(pd), B, heads, S, 0.5, st()))
    close(pd, pr, 5e-3)
    dp = torch.randn(B, heads, S, S, generator=g)
    prb = pd.cpu().double()
    want = 0.5 * prb * (dp.double() - (dp.double() * prb).sum(-1, keepdim=True))
    dpd = dp.to(DEV)
    ds = torch.empty_like(pd)
    _lib.check(lib.fs2_op_softmax_bwd(BF, p(dpd), p(pd), p(ds), B, heads, S, 0.5, st()))
    close(ds, want, 5e-3)
    # column sums / scatter / regulate / relu on bf16 inputs
    x = _bf(torch.randn(1000, 72, generator=g))
    xd = x.to(DEV)
    out = torch.empty(1, 72, device=DEV)
    ws = torch.zeros(lib.fs2_op_col_sum_ws_bytes(1000, 72, 0) // 4, device=DEV)
    _lib.check(lib.fs2_op_col_sum(BF, p(xd), p(out), p(ws), 1000, 72, 72, 0, 0, 1.0, st()))
    close(out[0], x.double().sum(0), 1e-5)
    idx = torch.randint(0, 20, (1000,), generator=g).int()
    table = torch.zeros(20, 72, device=DEV)
    idxd = idx.to(DEV)
    _lib.check(lib.fs2_op_scatter_rows(BF, p(xd), p(idxd), None, p(table), None, 1000, 72, 20, -1, st()))
    close(table, torch.zeros(20, 72, dtype=torch.float64).index_add_(0, idx.long(), x.double()), 1e-5)
    y = _bf(torch.randn(1000, 72, generator=g)).to(DEV)
    o2 = torch.empty_like(xd)
    _lib.check(lib.fs2_op_ew(BF, 1, p(xd), p(y), p(o2), 1000 * 72, 0.0, 0.0, st()))
    assert torch.equal(o2.cpu(), torch.where(y.cpu() > 0, x, torch.zeros((), dtype=torch.bfloat16)))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,S,Cc,k", [(3, 40, 64, 9), (2, 150, 72, 3), (1, 7, 8, 25), (2, 300, 132, 17)])
def test_dwconv_backward(B, S, Cc, k, dtype):
    """depth-wise Conv1d: data gradient (the forward kernel with reversed taps) and weight / bias gradient against autograd."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * S + k)
    bf = dtype == "bf16"
    cast = (lambda t: _bf(t).double()) if bf else (lambda t: t.double())
    x = cast(torch.randn(B, S, Cc, generator=g)).requires_grad_(True)
    w = torch.randn(Cc, 1, k, generator=g).double().requires_grad_(True)
    bias = torch.randn(Cc, generator=g).double().requires_grad_(True)
    dy = cast(torch.randn(B, S, Cc, generator=g))
    F.conv1d(x.transpose(1, 2), w, bias, padding=(k - 1) // 2, groups=Cc).transpose(1, 2).backward(dy)
    tdt = torch.bfloat16 if bf else torch.float32
    dt = BF if bf else F32
    dyd, xd, wd = dy.to(tdt).to(DEV).contiguous(), x.detach().to(tdt).to(DEV).contiguous(), w.detach().float().view(Cc, k).to(DEV).contiguous()
    dx = torch.empty(B * S, Cc, device=DEV, dtype=tdt)
    _lib.check(lib.fs2_op_dwconv_dgrad(dt, p(dyd), p(wd), p(dx), B, S, Cc, k, st()))
    close(dx, x.grad.reshape(B * S, Cc), 5e-3 if bf else 2e-5)
    nparts = lib.fs2_op_dwconv_wgrad_parts(B, S)
    part = torch.empty(nparts, Cc * (k + 1), device=DEV)
    _lib.check(lib.fs2_op_dwconv_wgrad(dt, p(dyd), p(xd), p(part), B, S, Cc, k, st()))
    tot = part.sum(0).cpu().double()
    close(tot[:Cc * k].view(Cc, k), w.grad.view(Cc, k), 2e-5)
    close(tot[Cc * k:], bias.grad, 2e-5)


def test_fold_unfold_conv2():
    """conv2 of the depth-wise layer as one folded linear map, and the chain rule back to its four parameters."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    H, Fc, M = 16, 64, 50
    gs = Fc // H
    G = torch.randn(Fc, gs, 1, generator=g, dtype=torch.float64, requires_grad=True)
    bg = torch.randn(Fc, generator=g, dtype=torch.float64, requires_grad=True)
    W21 = torch.randn(H, Fc, 1, generator=g, dtype=torch.float64, requires_grad=True)
    b21 = torch.randn(H, generator=g, dtype=torch.float64, requires_grad=True)
    x = torch.randn(1, Fc, M, generator=g, dtype=torch.float64)
    y = F.conv1d(F.conv1d(x, G, bg, groups=H), W21, b21)  # (1, H, M)
    dy = torch.randn(1, H, M, generator=g, dtype=torch.float64)
    y.backward(dy)
    Gd, bgd, Wd, bd = (t.detach().float().reshape(t.shape[0], -1).squeeze(-1).contiguous().to(DEV) if t.dim() > 1 else t.detach().float().to(DEV)
                       for t in (G, bg, W21, b21))
    Wf, bf_ = torch.empty(H, Fc, device=DEV), torch.empty(H, device=DEV)
    _lib.check(lib.fs2_op_fold_conv2(F32, p(Gd), p(bgd), p(Wd), p(bd), p(Wf), p(bf_), H, Fc, st()))
    yf = Wf.cpu().double() @ x[0] + bf_.cpu().double()[:, None]
    close(yf, y[0].detach(), 1e-5)
    dWf = (dy[0] @ x[0].t()).float().to(DEV).contiguous()  # gradient of the folded map
    dbf = dy[0].sum(1).float().to(DEV)
    dG, dbg, dW, db = (torch.ones_like(t) for t in (Gd, bgd, Wd, bd))
    _lib.check(lib.fs2_op_unfold_conv2(p(dWf), p(dbf), p(Gd), p(bgd), p(Wd), p(dG), p(dbg), p(dW), p(db), H, Fc, st()))
    close(dG - 1, G.grad.view(Fc, gs), 1e-5)
    close(dbg - 1, bg.grad, 1e-5)
    close(dW - 1, W21.grad.view(H, Fc), 1e-5)
    close(db - 1, b21.grad, 1e-5)


@pytest.mark.parametrize("dtype,drop", [("fp32", 0.0), ("bf16", 0.0), ("fp32", 0.25)])
def test_fused_softmax_backward_epilogue(dtype, drop):
    """dS = P o (dropout(dO V^T) - delta) / sqrt(d) straight out of the product (no dP tensor), delta from fs2_op_attn_delta,
    against autograd of softmax(QK^T / sqrt(d)) -> dropout -> @ V with the mask the dropout op itself produces."""
    lib = _lib.load()
    B, S, H, heads = 2, 40, 64, 2
    d = H // heads
    bf = dtype == "bf16"
    tdt, dt = (torch.bfloat16, BF) if bf else (torch.float32, F32)
    g = torch.Generator().manual_seed(12)
    qkv = torch.randn(B * S, 3 * H, generator=g).to(tdt)
    q, k, v = (t.double().view(B, S, heads, d).transpose(1, 2) for t in qkv.view(B, S, 3 * H).split(H, dim=-1))
    scores = (q @ k.transpose(-1, -2) / d ** 0.5).requires_grad_(True)
    P = torch.softmax(scores, dim=-1)
    Pd = P.detach().to(tdt).to(DEV).contiguous()
    seed, key = 77, 5
    if drop > 0:  # the mask the op regenerates: run it on ones
        ones = torch.ones(B, heads, S, S, device=DEV)
        m = torch.empty_like(ones)
        _lib.check(lib.fs2_op_dropout(F32, p(ones), p(m), ones.numel(), C.c_float(drop), C.c_uint64(seed), C.c_uint64(key), st()))
        mask = m.cpu().double()
    else:
        mask = torch.ones(B, heads, S, S, dtype=torch.float64)
    out = ((P * mask) @ v).transpose(1, 2).reshape(B * S, H)
    dout = torch.randn(B * S, H, generator=g).to(tdt)
    out.backward(dout.double())
    od, dod, qkvd = out.detach().to(tdt).to(DEV).contiguous(), dout.to(DEV), qkv.to(DEV)
    delta = torch.empty(B, heads, S, device=DEV)
    _lib.check(lib.fs2_op_attn_delta(dt, p(dod), p(od), p(delta), B, S, H, heads, st()))
    ds = torch.empty(B, heads, S, S, device=DEV, dtype=tdt)
    dsc = _lib.BGemmDescC()
    for kk, vv in dict(M=S, N=S, K=d, sAm=H, sAk=1, sBk=1, sBn=3 * H, ldc=S, sA1=S * H, sA2=d, sB1=S * 3 * H, sB2=d, sC1=heads * S * S,
                       sC2=S * S, alpha=1.0 / d ** 0.5, beta=0.0, splitk=1, taps=1, nb1=B, nb2=heads, c_dtype=dt).items():
        setattr(dsc, kk, vv)
    _lib.check(lib.fs2_op_bgemm_softmax_bwd(dt, C.byref(dsc), p(dod), p(qkvd[:, 2 * H:]), p(ds), p(Pd), p(delta), C.c_float(drop),
                                            C.c_uint64(seed), C.c_uint64(key), st()))
    close(ds, scores.grad / d ** 0.5, 2e-2 if bf else 3e-5)  # gradient of the RAW product Q K^T (what dQ = dS K and dK = dS^T Q consume)


@pytest.fixture(params=[(), (905, 907), (905, 908), (901, 903)], ids=["by-size", "dkdv3-dq3", "dkdv3-dq4", "dkdv2-dq2"])
def attn_bwd_blocks(request):
    """16-row blocks per wave of the (dK dV, dQ) launches: the default (1 / by size) or forced (3 / 4: one wave per SIMD)"""
    lib = _lib.load()
    for k in request.param:
        lib.fs2_op_set_gemm_variant(k)
    yield request.param
    lib.fs2_op_set_gemm_variant(909)
    lib.fs2_op_set_gemm_variant(904)


@pytest.mark.parametrize("B,S,heads,drop", [(2, 150, 2, 0.0), (3, 64, 1, 0.0), (2, 200, 2, 0.2), (1, 37, 2, 0.0), (2, 700, 2, 0.0)])
def test_flash_attention_training_forward_and_backward(B, S, heads, drop, attn_bwd_blocks):
    """The fused attention on the training path: output + lse2 (+ attention-weight dropout) and the recomputing backward
    (dQ, dK, dV with neither P nor dP in HBM), against autograd of softmax(q k^T / sqrt(d) + key padding) -> dropout -> @ v
    in float64 on the same bf16 inputs, with the mask the dropout op regenerates."""
    lib = _lib.load()
    d = 128
    H = heads * d
    g = torch.Generator().manual_seed(S + heads)
    qkv = (0.5 * torch.randn(B * S, 3 * H, generator=g)).to(torch.bfloat16)
    lens = torch.randint(max(1, S // 2), S + 1, (B,), generator=g)
    lens[0] = S
    pad = torch.arange(S)[None, :] >= lens[:, None]
    q, k, v = (t.double().view(B, S, heads, d).transpose(1, 2) for t in qkv.view(B, S, 3 * H).split(H, dim=-1))
    q = q.clone().requires_grad_(True); k = k.clone().requires_grad_(True); v = v.clone().requires_grad_(True)
    scores = (q @ k.transpose(-1, -2) / d ** 0.5).masked_fill(pad[:, None, None, :], float("-inf"))
    P = torch.softmax(scores, dim=-1)
    seed, key = 123, 7
    if drop > 0:
        ones = torch.ones(B, heads, S, S, device=DEV)
        m = torch.empty_like(ones)
        _lib.check(lib.fs2_op_dropout(F32, p(ones), p(m), ones.numel(), C.c_float(drop), C.c_uint64(seed), C.c_uint64(key), st()))
        mask = m.cpu().double()
    else:
        mask = torch.ones(B, heads, S, S, dtype=torch.float64)
    out = ((P * mask) @ v).transpose(1, 2).reshape(B * S, H)
    dout = torch.randn(B * S, H, generator=g).to(torch.bfloat16)
    out.backward(dout.double())
    # forward
    qd, padd = qkv.to(DEV), pad.to(torch.uint8).to(DEV)
    od = torch.empty(B * S, H, device=DEV, dtype=torch.bfloat16)
    bb = C.c_size_t()
    vb = lib.fs2_op_attention_scratch_bytes(BF, B, S, H, heads, C.byref(bb))
    vt, bits = torch.empty(vb, dtype=torch.uint8, device=DEV), torch.empty(bb.value, dtype=torch.uint8, device=DEV)
    lse = torch.empty(B, heads, S, device=DEV)
    _lib.check(lib.fs2_op_attention_train(BF, p(qd), p(padd), p(od), p(vt), p(bits), p(lse), B, S, H, heads, C.c_float(drop),
                                          C.c_uint64(seed), C.c_uint64(key), st()))
    close(od, out.detach(), 2e-2)
    want_lse = torch.logsumexp(scores.detach(), dim=-1) / np.log(2.0)
    assert float((lse.cpu().double() - want_lse).abs().max()) <= 2e-2
    # backward
    assert lib.fs2_op_attention_bwd_supported(BF, H, heads) == 1
    dod = dout.to(DEV)
    delta = torch.empty(B, heads, S, device=DEV)
    _lib.check(lib.fs2_op_attn_delta(BF, p(dod), p(od), p(delta), B, S, H, heads, st()))
    dqkv = torch.full((B * S, 3 * H), float("nan"), device=DEV, dtype=torch.bfloat16)
    _lib.check(lib.fs2_op_attention_bwd(BF, p(qd), p(dod), p(lse), p(delta), p(padd), p(dqkv), B, S, H, heads, C.c_float(drop),
                                        C.c_uint64(seed), C.c_uint64(key), st()))
    want = torch.cat([t.grad.transpose(1, 2).reshape(B * S, H) for t in (q, k, v)], dim=1)
    assert bool(torch.isfinite(dqkv.float()).all())
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        close(dqkv[:, sl], want[:, sl], 3e-2)


@pytest.mark.parametrize("M,N,K,splitk", [(256, 128, 64, 1), (128, 256, 320, 3), (384, 128, 4096, 8)])
def test_bgemm_bf16_full_tile_instantiation(M, N, K, splitk):
    """Shapes that take the bounds-free instantiation (full 128 x 128 x 32 tiles, aligned operands): all four layouts, plus the
    weight-gradient form with utterance boundaries, against float64."""
    g = torch.Generator().manual_seed(M + N + K)
    a, b = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(K, N, generator=g))
    want = a.double() @ b.double()
    for akc in (True, False):
        for bkc in (True, False):
            A = (a if akc else a.t().contiguous()).to(DEV)
            Bm = (b.t().contiguous() if bkc else b).to(DEV)
            c = torch.empty(M, N, device=DEV)
            bgemm(dict(M=M, N=N, K=K, sAm=K if akc else 1, sAk=1 if akc else M, sBk=1 if bkc else N, sBn=K if bkc else 1, ldc=N,
                       splitk=splitk), A, Bm, c, dtype=BF)
            close(c, want, 1e-5)
    taps, Cin, Nn, S, B = 5, 128, 128, 64, 4  # wgrad form on full tiles: K = B*S = 256
    x = _bf(torch.randn(B, S, Cin, generator=g)).double().requires_grad_(True)
    w = _bf(torch.randn(Nn, Cin, taps, generator=g)).double().requires_grad_(True)
    dy = _bf(torch.randn(B, S, Nn, generator=g)).double()
    F.conv1d(x.transpose(1, 2), w, padding="same").transpose(1, 2).backward(dy)
    dyd, xd = _bf(dy.reshape(B * S, Nn)).contiguous().to(DEV), _bf(x.detach().reshape(B * S, Cin)).contiguous().to(DEV)
    dw = torch.zeros(Nn, taps * Cin, device=DEV)
    bgemm(dict(M=Nn, N=Cin, K=B * S, sAm=1, sAk=Nn, sBk=Cin, sBn=1, ldc=taps * Cin, nb2=taps, sC2=Cin, seg=S, b_shift0=-2, b_shift_step=1,
               splitk=2), dyd, xd, dw, dtype=BF)
    close(dw, w.grad.permute(0, 2, 1).reshape(Nn, taps * Cin), 1e-5)


def _tn256(desc_kw):
    lib = _lib.load()
    d = _lib.BGemmDescC()
    base = dict(nb1=1, nb2=1, alpha=1.0, beta=0.0, splitk=1, taps=1, c_dtype=F32)
    base.update(desc_kw)
    for k, v in base.items():
        setattr(d, k, v)
    return int(lib.fs2_op_bgemm_tn256(C.byref(d)))


@pytest.mark.parametrize("M,N,K,sk", [(256, 256, 32, 1), (256, 512, 96, 1), (512, 256, 160, 2), (256, 256, 1024, 7), (768, 256, 2048, 5),
                                       (256, 256, 64, 3)])
def test_bgemm_tn256_plain(M, N, K, sk):
    """The 256 x 256 LDS-DMA kernel of the bf16 TN products (csrc/bgemm.hip bgemm_tn256_kernel): plain form, direct and
    split-K (uneven splits, more splits than k-steps), alpha / beta; against fp64 of the bf16-rounded operands, and against
    the 128 x 128 kernel (knob 1000) on the same inputs."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K + sk)
    a = torch.randn(K, M, generator=g).to(torch.bfloat16)
    b = torch.randn(K, N, generator=g).to(torch.bfloat16)
    want = a.double().t() @ b.double()
    kw = dict(M=M, N=N, K=K, sAm=1, sAk=M, sBk=N, sBn=1, ldc=N, splitk=sk)
    assert _tn256(kw) == 1
    c = torch.full((M, N), 2.0, device=DEV)
    bgemm(dict(kw, alpha=0.5, beta=3.0), a.to(DEV), b.to(DEV), c, dtype=_lib.FS2_BF16)
    close(c, 0.5 * want + 6.0, rel=1e-5)
    try:
        lib.fs2_op_set_gemm_variant(1000)
        assert _tn256(kw) == 0
        c2 = torch.full((M, N), 2.0, device=DEV)
        bgemm(dict(kw, alpha=0.5, beta=3.0), a.to(DEV), b.to(DEV), c2, dtype=_lib.FS2_BF16)
    finally:
        lib.fs2_op_set_gemm_variant(1001)
    close(c, c2.cpu(), rel=1e-5)


@pytest.mark.parametrize("taps,Cin,N,S,B,sk", [(9, 256, 256, 96, 4, 1), (9, 256, 512, 64, 3, 5), (3, 512, 256, 32, 5, 2), (5, 256, 256, 160, 2, 10)])
def test_bgemm_tn256_wgrad_form(taps, Cin, N, S, B, sk):
    """... and the wgrad form (tap = batch index, shifted time rows that read as zero outside their utterance, utterance
    boundaries inside and at the edges of a k-step, splits that start inside an utterance) against autograd of F.conv1d."""
    g = torch.Generator().manual_seed(taps * 100 + Cin + S)
    x = torch.randn(B, S, Cin, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, S, N, generator=g).to(torch.bfloat16)
    xd64 = x.double().requires_grad_(True)
    w = torch.zeros(N, Cin, taps, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(xd64.transpose(1, 2), w, padding="same").transpose(1, 2)
    y.backward(dy.double())
    pad = (taps - 1) // 2
    kw = dict(M=N, N=Cin, K=B * S, sAm=1, sAk=N, sBk=Cin, sBn=1, ldc=taps * Cin, nb2=taps, sC2=Cin, seg=S,
              b_shift0=-pad, b_shift_step=1, splitk=sk)
    assert _tn256(kw) == 1
    dw = torch.zeros(N, taps * Cin, device=DEV)
    bgemm(kw, dy.reshape(B * S, N).contiguous().to(DEV), x.reshape(B * S, Cin).contiguous().to(DEV), dw, dtype=_lib.FS2_BF16)
    close(dw, w.grad.permute(0, 2, 1).reshape(N, taps * Cin), rel=1e-5)


def test_bgemm_tn256_eligibility():
    """Shapes the 256-tile kernel must leave to the general one: partial tiles, k-contiguous operands, bf16 output, utterances
    that are not whole k-steps."""
    ok = dict(M=256, N=256, K=64, sAm=1, sAk=256, sBk=256, sBn=1, ldc=256)
    assert _tn256(ok) == 1
    assert _tn256(dict(ok, M=128)) == 0
    assert _tn256(dict(ok, N=384, sBk=384, ldc=384)) == 0
    assert _tn256(dict(ok, K=48)) == 0
    assert _tn256(dict(ok, sAm=64, sAk=1)) == 0
    assert _tn256(dict(ok, c_dtype=_lib.FS2_BF16)) == 0
    assert _tn256(dict(ok, seg=48, K=96)) == 0
    assert _tn256(dict(ok, seg=32, K=64, nb2=3, b_shift0=-1, b_shift_step=1)) == 1
    assert _tn256(dict(ok, b_shift0=0, b_shift_step=1)) == 1   # no utterance length: the shifts are not applied


@pytest.mark.parametrize("fused", [0, 1])
def test_col_sum_two_destinations_and_one_launch_variant(fused):
    """Column sums with two destinations and their own accumulate flags, several chunks per column block, bit-equal reruns; in
    the default two-launch form and the one-launch variant (knob 1101: repeated launches on one workspace - the ticket
    counters must come back to zero - and the fallback for shapes with more (column block, segment) pairs than counters)."""
    lib = _lib.load()
    lib.fs2_op_set_gemm_variant(1100 + fused)
    try:
        _col_sum_cases(lib)
    finally:
        lib.fs2_op_set_gemm_variant(1100)


def _col_sum_cases(lib):
    g = torch.Generator().manual_seed(11)
    M, N, n1 = 1000, 3 * 200, 2 * 200
    x = torch.randn(M, N, generator=g)
    xd = x.to(DEV)
    ws = torch.zeros(lib.fs2_op_col_sum_ws_bytes(M, N, 0) // 4, device=DEV)
    want = x.double().sum(0)
    first = None
    for it in range(4):
        out = torch.ones(n1, device=DEV)
        out2 = torch.full((N - n1,), 5.0, device=DEV)
        _lib.check(lib.fs2_op_col_sum2(F32, p(xd), p(out), p(out2), n1, p(ws), M, N, N, 1, 0, 1.0, st()))
        close(out, want[:n1] + 1.0, 1e-5)
        close(out2, want[n1:], 1e-5)
        both = torch.cat([out, out2]).cpu()
        if first is None:
            first = both
        assert torch.equal(both, first)
        assert not ws[:8192].any()
        # a different shape on the same workspace in between
        o3 = torch.empty(72, device=DEV)
        _lib.check(lib.fs2_op_col_sum(F32, p(xd), p(o3), p(ws), M, 72, N, 0, 0, 2.0, st()))
        close(o3, 2.0 * want[:72], 1e-5)
    # more (column block, segment) pairs than counters: 70 column blocks x 120 segments
    nseg, seg, N2 = 120, 130, 64 * 70
    x2 = torch.randn(nseg * seg, N2, generator=g)
    ws2 = torch.zeros(lib.fs2_op_col_sum_ws_bytes(nseg * seg, N2, seg) // 4, device=DEV)
    out = torch.empty(nseg, N2, device=DEV)
    _lib.check(lib.fs2_op_col_sum(F32, p(x2.to(DEV)), p(out), p(ws2), nseg * seg, N2, N2, seg, 0, 1.0, st()))
    close(out, x2.double().view(nseg, seg, N2).sum(1), 1e-5)


@pytest.mark.parametrize("R,H,V,skip,dtype", [(5000, 72, 40, 0, "fp32"), (2048, 256, 256, -1, "bf16"), (9000, 200, 3, 2, "bf16"),
                                              (4097, 64, 255, -1, "fp32")])
def test_scatter_rows_chunked_form(R, H, V, skip, dtype):
    """The two-phase embedding backward (chunk x 64-column LDS tables, then a column sum over the chunks): ragged last chunk,
    column tail, skewed and out-of-range indices, padding row, int32 / int64 indices; bit-equal reruns; against index_add_."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(R + V)
    idx = torch.randint(0, V, (R,), generator=g)
    idx[::7] = V - 1                       # one heavy table row
    idx[5] = -1
    idx[6] = V + 3                         # ignored, as the one-launch kernel does (no workgroup owns them)
    x = torch.randn(R, H, generator=g)
    if dtype == "bf16":
        x = x.to(torch.bfloat16)
    xd = x.to(DEV)
    nbytes = lib.fs2_op_scatter_rows_ws_bytes(R, H, V)
    assert nbytes > 0
    ws = torch.zeros(nbytes // 4, device=DEV)
    keep = (idx >= 0) & (idx < V) & (idx != skip)
    want = torch.ones(V, H, dtype=torch.float64).index_add_(0, idx[keep], x.double()[keep])
    first = None
    for kind in ("i32", "i64", "i32"):
        table = torch.ones(V, H, device=DEV)
        i32 = idx.int().to(DEV) if kind == "i32" else None
        i64 = idx.to(DEV) if kind == "i64" else None
        _lib.check(lib.fs2_op_scatter_rows(_lib.FS2_BF16 if dtype == "bf16" else F32, p(xd), p(i32), p(i64), p(table), p(ws), R, H, V, skip, st()))
        close(table, want, 1e-5)
        if first is None:
            first = table.clone()
        assert torch.equal(table, first)
    assert lib.fs2_op_scatter_rows_ws_bytes(100, H, V) == 0 and lib.fs2_op_scatter_rows_ws_bytes(R, H, 300) == 0


def test_transpose_weight_batch_matches_single_launches():
    """All data-gradient weights in one launch (csrc/backward.hip transpose_weight_batch_kernel) against the per-weight
    kernel: tap counts 1 / 3 / 9, sizes that are not whole 64 x 64 tiles, odd widths (no pair accesses), bit-equal."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    shapes = [(256, 256, 1), (80, 256, 1), (1024, 256, 9), (256, 1024, 1), (72, 200, 3), (33, 65, 5), (64, 64, 1), (7, 130, 2)]
    rows, tiles, keep = [], 0, []
    for N, Cin, taps in shapes:
        src = torch.randn(N, taps * Cin, generator=g).to(torch.bfloat16).to(DEV)
        want = torch.empty(Cin, taps * N, device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.fs2_op_transpose_weight(_lib.FS2_BF16, p(src), p(want), N, Cin, taps, st()))
        got = torch.zeros_like(want)
        rows.append([src.data_ptr(), got.data_ptr(), N, Cin, taps, tiles])
        tiles += int(lib.fs2_op_transpose_weight_tiles(N, Cin, taps))
        keep.append((src, want, got))
    tab = torch.tensor(rows, dtype=torch.int64, device=DEV)
    _lib.check(lib.fs2_op_transpose_weight_batch(p(tab), len(rows), tiles, st()))
    for (src, want, got), (N, Cin, taps) in zip(keep, shapes):
        assert torch.equal(got, want), (N, Cin, taps)
        ref = src.view(N, taps, Cin).flip(1).permute(2, 1, 0).reshape(Cin, taps * N)
        assert torch.equal(got, ref)


def test_bgemm_tn256_two_batch_levels():
    """Both batch levels with their own operand / output strides (operands interleaved per head the way the attention tensors
    are), direct and split-K."""
    g = torch.Generator().manual_seed(77)
    nb1, nb2, K, M, N = 2, 3, 96, 256, 256
    a = torch.randn(nb1, K, nb2 * M, generator=g).to(torch.bfloat16)   # (b1, k, [b2][m])
    b = torch.randn(nb1, K, nb2 * N, generator=g).to(torch.bfloat16)
    want = torch.einsum("bkhm,bkhn->bhmn", a.double().view(nb1, K, nb2, M), b.double().view(nb1, K, nb2, N))
    for sk in (1, 2):
        kw = dict(M=M, N=N, K=K, sAm=1, sAk=nb2 * M, sBk=nb2 * N, sBn=1, ldc=N, nb1=nb1, nb2=nb2, sA1=K * nb2 * M, sA2=M,
                  sB1=K * nb2 * N, sB2=N, sC1=nb2 * M * N, sC2=M * N, splitk=sk)
        assert _tn256(kw) == 1
        c = torch.zeros(nb1, nb2, M, N, device=DEV)
        bgemm(kw, a.to(DEV), b.to(DEV), c, dtype=_lib.FS2_BF16)
        close(c, want, rel=1e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,Cin,taps,S", [(512, 1024, 256, 1, 512), (384, 200, 64, 1, 384), (3 * 96, 256, 128, 3, 96), (256, 320, 64, 1, 256)])
def test_gemm_gated_store(M, N, Cin, taps, S, dtype):
    """fs2_op_gemm_gated: the ReLU backward folded into the data-gradient product's store (c = gate > 0 ? x w^T : 0), full and
    partial column tiles, conv form; against the ungated launch + the elementwise mask (bit-equal) and fp64."""
    lib = _lib.load()
    bf = dtype == "bf16"
    dt, td = (_lib.FS2_BF16, torch.bfloat16) if bf else (F32, torch.float32)
    g = torch.Generator().manual_seed(M + N + taps)
    x = torch.randn(M, Cin, generator=g).to(td)
    w = (torch.randn(N, taps * Cin, generator=g) / (taps * Cin) ** 0.5).to(td)
    gate = torch.relu(torch.randn(M, N, generator=g)).to(td)   # a ReLU output: zeros and positives
    xd, wd, gd = x.to(DEV), w.to(DEV), gate.to(DEV)
    c = torch.empty(M, N, device=DEV, dtype=td)
    _lib.check(lib.fs2_op_gemm_gated(dt, p(xd), p(wd), None, p(gd), 1.0, p(c), M, N, Cin, taps, S, st()))
    c0 = torch.empty(M, N, device=DEV, dtype=td)
    _lib.check(lib.fs2_op_gemm(dt, dt, p(xd), p(wd), None, p(c0), M, N, Cin, taps, S, 0, st()))
    want = torch.where(gd > 0, c0, torch.zeros_like(c0))
    assert torch.equal(c, want)
    assert float((want != 0).float().mean()) > 0.3 and float((want == 0).float().mean()) > 0.3
    if taps == 1:
        ref = torch.where(gate.double() > 0, x.double() @ w.double().t(), torch.zeros(M, N, dtype=torch.float64))
        close(c, ref, rel=2e-2 if bf else 2e-5)
        _lib.check(lib.fs2_op_gemm_gated(dt, p(xd), p(wd), None, p(gd), 1.25, p(c), M, N, Cin, taps, S, st()))   # dropout's 1 / (1 - p)
        close(c, 1.25 * ref, rel=2e-2 if bf else 2e-5)


def test_gemm_gated_rejects_shapes_without_the_epilogue():
    lib = _lib.load()
    x = torch.zeros(256, 64, device=DEV)
    w = torch.zeros(128, 64, device=DEV)
    c = torch.zeros(256, 128, device=DEV)
    assert lib.fs2_op_gemm_gated(F32, p(x), p(w), None, p(c), 1.0, p(c), 256, 128, 64, 1, 256, st()) != 0   # N < 192: not the slab kernel
    assert lib.fs2_op_gemm_gated(F32, p(x), p(w), None, None, 1.0, p(c), 256, 128, 64, 1, 256, st()) != 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_col_sum_weighted(dtype):
    """out[n] += sum_r w[r] x[r][n] (the weight gradient of a Linear(H, 1) head) against fp64."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    M, N = 3000, 200
    x = torch.randn(M, N, generator=g)
    if dtype == "bf16":
        x = x.to(torch.bfloat16)
    w = torch.randn(M, generator=g)
    ws = torch.zeros(lib.fs2_op_col_sum_ws_bytes(M, N, 0) // 4, device=DEV)
    out = torch.ones(N, device=DEV)
    _lib.check(lib.fs2_op_col_sum_weighted(_lib.FS2_BF16 if dtype == "bf16" else F32, p(x.to(DEV)), p(w.to(DEV)), p(out), p(ws), M, N, N, 1,
                                           0.5, st()))
    close(out, 1.0 + 0.5 * (w.double()[:, None] * x.double()).sum(0), 1e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_layernorm_with_dropout_matches_the_two_pass_form(dtype):
    """fs2_op_layernorm_dropout == fs2_op_layernorm then fs2_op_dropout (same mask: the element index over (M, H)), and
    fs2_op_layernorm_bwd_dropout == fs2_op_dropout on dy then fs2_op_layernorm_bwd (fp32: bit-equal; bf16: one rounding fewer)."""
    lib = _lib.load()
    bf = dtype == "bf16"
    dt, td = (_lib.FS2_BF16, torch.bfloat16) if bf else (F32, torch.float32)
    g = torch.Generator().manual_seed(4)
    M, H, pdrop, seed, key = 300, 264, 0.3, 12345, 7
    x = torch.randn(M, H, generator=g).to(td).to(DEV)
    gam, bet = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV), (0.1 * torch.randn(H, generator=g)).to(DEV)
    y1 = torch.empty(M, H, device=DEV, dtype=td)
    _lib.check(lib.fs2_op_layernorm_dropout(dt, p(x), None, p(gam), p(bet), p(y1), M, H, pdrop, seed, key, st()))
    y0 = torch.empty(M, H, device=DEV, dtype=td)
    _lib.check(lib.fs2_op_layernorm(dt, p(x), None, p(gam), p(bet), p(y0), None, 0.0, None, None, M, H, st()))
    _lib.check(lib.fs2_op_dropout(dt, p(y0), p(y0), M * H, pdrop, seed, key, st()))
    assert torch.equal(y1 == 0, y0 == 0) and 0.25 < float((y1 == 0).float().mean()) < 0.35
    close(y1, y0.cpu(), rel=1e-2 if bf else 1e-6)
    dy = torch.randn(M, H, generator=g).to(td).to(DEV)
    nparts = lib.fs2_op_layernorm_bwd_parts(M)
    dz1, part1 = torch.empty(M, H, device=DEV, dtype=td), torch.empty(nparts, 3 * H, device=DEV)
    _lib.check(lib.fs2_op_layernorm_bwd_dropout(dt, p(x), None, p(dy), p(gam), p(dz1), p(part1), M, H, 0, pdrop, seed, key, st()))
    dyd = dy.clone()
    _lib.check(lib.fs2_op_dropout(dt, p(dyd), p(dyd), M * H, pdrop, seed, key, st()))
    dz0, part0 = torch.empty(M, H, device=DEV, dtype=td), torch.empty(nparts, 3 * H, device=DEV)
    _lib.check(lib.fs2_op_layernorm_bwd(dt, p(x), None, p(dyd), p(gam), p(dz0), p(part0), M, H, 0, st()))
    if bf:
        close(dz1, dz0.cpu(), rel=2e-2)
        close(part1, part0.cpu(), rel=1e-2)
    else:
        assert torch.equal(dz1, dz0) and torch.equal(part1, part0)
