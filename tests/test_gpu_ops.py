"""Per-kernel parity (`-m gpu`): every HIP operator, called through the C ABI, against the same op
of the CPU oracle / plain torch fp32 on seeded inputs.  fp32 mode is held to float-roundoff;
bf16 mode is compared with fp32 math on the bf16-rounded inputs (tolerance = bf16 output rounding).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _gpu as G
from oracle import oracle_cpu

pytestmark = pytest.mark.gpu

DTYPES = [pytest.param(G.F32, id="fp32"), pytest.param(G.BF16, id="bf16")]


def tol(dtype, ref, f32=2e-5, bf16=1.2e-2):
    scale = float(ref.abs().max()) + 1e-6
    return (bf16 if dtype == G.BF16 else f32) * scale


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(params=[1, 2, 3, 4, 5, 6, 7],
                ids=["gemm128x128", "gemm128x256dma", "slab128", "slab192", "slab256", "slab32", "slab64"])
def gemm_variant(request):
    """Run every GEMM/conv case on BOTH kernels (the engine picks by problem size)."""
    G.lib().fs2_op_set_gemm_variant(request.param)
    yield request.param
    G.lib().fs2_op_set_gemm_variant(0)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,relu", [(200, 80, 64, False), (128, 128, 128, True), (300, 260, 256, False),
                                         (1000, 768, 256, True), (37, 4, 64, False), (513, 1024, 1024, False)])
def test_gemm_plain(dtype, M, N, K, relu, gemm_variant):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    # asymmetric, transpose-detecting reference
    ref = G.rounded(x, dtype) @ G.rounded(w, dtype).T + b
    if relu:
        ref = torch.relu(ref)
    got = G.gemm(dtype, x, w, b, relu=relu)
    err = float((got - ref).abs().max())
    assert err <= tol(dtype, ref), (err, tol(dtype, ref))


@pytest.mark.parametrize("M,N,relu,bias", [(49152, 768, False, True), (8192, 768, False, True), (8192, 256, True, True), (1, 256, False, False),
                                         (95, 512, False, True), (97, 256, True, True), (96 * 9 + 5, 1024, False, True), (300, 768, True, False)])
def test_wres_gemm_is_bit_identical_to_the_slab_kernel(M, N, relu, bias):
    """bf16 GEMMs with K = 256 take the weight-resident kernel (gemm_wres.hip: a column tile's weights in registers, 96-row
    activation tiles streamed through LDS, one workgroup walking many tiles): bit-equal to the slab kernel it replaces (knob
    1400) on every work split - one row, ragged last tiles, more row groups than tiles, several tiles per workgroup - and
    within the bf16 tolerance of the fp32 reference."""
    x, w = rnd(M, 256, seed=3), rnd(N, 256, seed=4) / 16
    b = rnd(N, seed=5) if bias else None
    try:
        G.lib().fs2_op_set_gemm_variant(1400)
        old = G.gemm(G.BF16, x, w, b, relu=relu)
        G.lib().fs2_op_set_gemm_variant(1402)   # the weight-resident kernel at every size (1401, the default, picks it from three tiles per workgroup on)
        got = G.gemm(G.BF16, x, w, b, relu=relu)
        again = G.gemm(G.BF16, x, w, b, relu=relu)
    finally:
        G.lib().fs2_op_set_gemm_variant(1401)
    assert torch.equal(got, old) and torch.equal(got, again)
    ref = G.rounded(x, G.BF16) @ G.rounded(w, G.BF16).T + (0 if b is None else b)
    ref = ref.clamp_min(0) if relu else ref
    assert float((got - ref).abs().max()) <= tol(G.BF16, ref)


@pytest.mark.parametrize("M,N,K,relu,add", [
    (49152, 2304, 768, False, False),   # C3 decoder in-projection: 256-row tiles, 6.75 tiles per workgroup
    (49152, 3072, 768, True, False),    # C3 FFN pointwise conv1.1 (+ ReLU): 9 tiles per workgroup
    (49152, 768, 3072, False, True),    # C3 conv2 / out-projection shape with the residual in the accumulators (deferred epilogue, 192-row tiles)
    (49152, 768, 768, False, True),
    (36864, 768, 768, True, False),     # predictor-sized pointwise + ReLU
    (50000, 712, 384, False, False),    # ragged last row tile AND a column tail (N = 2 x 256 + 200): the uncounted-wait path
    (50000, 712, 384, False, True),
    (70001, 256, 128, True, False),     # two K steps per tile (the shortest stream the kernel takes), one column tile, odd row count
    (24576, 1024, 1024, False, False)])
def test_persistent_gemm_is_bit_identical_to_the_slab_kernel(M, N, K, relu, add):
    """gemm_persist.hip (knob 221, default): bf16 pointwise launches of more tiles than CUs on ONE workgroup per CU that walks its
    tiles - the next tile's first operands requested inside this tile's last K step, the epilogue's stores left in flight behind
    a counted wait - against the one-tile-per-workgroup slab kernel (knob 220): the same MFMA sequence per output element, so
    the same bits; plain (bias, ReLU) epilogue and the residual-in-the-accumulators epilogue; twice (tile hand-over races would
    show as run-to-run differences), and within the bf16 tolerance of the fp32 reference."""
    x, w = rnd(M, K, seed=31), rnd(N, K, seed=32) / math.sqrt(K)
    b = rnd(N, seed=33)
    addend = rnd(M, N, seed=34) if add else None
    run = (lambda: G.gemm_add(G.BF16, x, w, b, addend, in_place=False)) if add else (lambda: G.gemm(G.BF16, x, w, b, relu=relu))
    try:
        G.lib().fs2_op_set_gemm_variant(220)
        old = run()
        G.lib().fs2_op_set_gemm_variant(221)
        got = run()
        again = run()
    finally:
        G.lib().fs2_op_set_gemm_variant(221)
    assert torch.equal(got, old) and torch.equal(got, again)
    ref = G.rounded(x, G.BF16) @ G.rounded(w, G.BF16).T + b
    ref = ref.clamp_min(0) if relu else ref
    if add:
        ref = ref + G.rounded(addend, G.BF16)
    assert float((got - ref).abs().max()) <= tol(G.BF16, ref)


@pytest.mark.parametrize("B,S,Cin,N,taps", [(8, 768, 256, 1024, 9), (4, 512, 256, 2048, 3), (6, 1000, 1024, 1024, 1), (3, 700, 256, 1024, 9)])
def test_slab_tile_orders_are_bit_identical(B, S, Cin, N, taps):
    """Slab kernel tile orders (knobs 200 / 201 / 202: plain, XCD-contiguous, XCD-contiguous with column-tile pairs per XCD for weight
    panels wider than an L2 - the decoder FFN conv1): which workgroup computes which tile changes, nothing inside a tile; every tile
    computed exactly once (an order that skipped or doubled tiles would leave the poisoned output or differ).  The last case's row-tile
    count does not divide over the XCD groups: the launcher falls back to 201."""
    x, w = rnd(B * S, Cin, seed=35), rnd(N, taps * Cin, seed=36) / math.sqrt(taps * Cin)
    b = rnd(N, seed=37)
    outs = []
    try:
        for knob in (200, 201, 202):
            G.lib().fs2_op_set_gemm_variant(knob)
            outs.append(G.gemm(G.BF16, x, w, b, taps=taps, S=S, relu=True))
    finally:
        G.lib().fs2_op_set_gemm_variant(201)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("M,N,K", [(49152, 2304, 768), (8192, 2304, 768), (700, 2304, 768), (20000, 712, 384), (30000, 3072, 1024),
                                   (5000, 80, 768), (333, 100, 384)])
def test_gemm_rowscale_is_layernorm_then_gemm(M, N, K):
    """fs2_op_gemm_rowscale: LayerNorm(v) W^T + b evaluated on the PRE-norm rows v with gamma / beta folded into the operands and
    per-row (rstd, rstd * mean) applied in the epilogue - what the engine's in-projection does behind a deferred norm2
    (model.py:113-115).  Against LayerNorm-then-GEMM in fp32 on the same bf16-rounded v (tolerance: one bf16 rounding of W' and of the
    output, not of the normalised activations), and bit-equal between the one-tile-per-workgroup slab kernel (knob 220) and the
    persistent kernel (221), ragged last tiles and a column tail included."""
    g = torch.Generator().manual_seed(41)
    v = (torch.randn(M, K, generator=g) * 1.7 + 0.4 * torch.randn(M, 1, generator=g)).bfloat16().float()   # rows with their own mean
    w0 = torch.randn(N, K, generator=g) / math.sqrt(K)
    b0 = torch.randn(N, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    wf = (w0 * gamma[None, :]).bfloat16().float()                      # as stored
    bf = (b0.double() + w0.double() @ beta.double()).float()
    wg = wf.double().sum(1).float()
    parts = (K + 255) // 256
    stats = torch.zeros(M, parts, 2)
    for q in range(parts):                                            # what the deferred-LayerNorm epilogue leaves: per 256-column tile
        blk = v[:, q * 256:(q + 1) * 256].double()
        stats[:, q, 0] = blk.sum(1).float()
        stats[:, q, 1] = (blk * blk).sum(1).float()
    try:
        G.lib().fs2_op_set_gemm_variant(220)
        old = G.gemm_rowscale(v, wf, bf, stats, wg)
        G.lib().fs2_op_set_gemm_variant(221)
        got = G.gemm_rowscale(v, wf, bf, stats, wg)
        again = G.gemm_rowscale(v, wf, bf, stats, wg)
    finally:
        G.lib().fs2_op_set_gemm_variant(221)
    assert torch.equal(got, old) and torch.equal(got, again)
    ref = F.layer_norm(v, (K,), gamma, beta, 1e-5) @ w0.T + b0
    err = float((got - ref).abs().max())
    assert err <= 2.5 * tol(G.BF16, ref), (err, tol(G.BF16, ref))      # bf16 W' (2^-9 per weight) + bf16 output


def test_gemm_bf16_in_fp32_out(gemm_variant):
    x, w, b = rnd(150, 256, seed=4), rnd(80, 256, seed=5, scale=1 / 16), rnd(80, seed=6)
    ref = G.rounded(x, G.BF16) @ G.rounded(w, G.BF16).T + b
    got = G.gemm(G.BF16, x, w, b, out_dtype=G.F32)
    assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


def test_gemm_identity_layout(gemm_variant):
    # A = I picks W^T exactly: catches swapped row/col maps that symmetric data would hide
    K = 64
    x = torch.eye(K)
    w = torch.arange(K * K, dtype=torch.float32).reshape(K, K) / 64.0  # asymmetric
    got = G.gemm(G.F32, x, w, None)
    assert torch.equal(got, w.T.contiguous())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,Cin,N,k", [(3, 37, 64, 128, 3), (2, 130, 64, 64, 9), (4, 11, 128, 256, 5),
                                          (2, 300, 256, 1024, 9), (1, 5, 64, 64, 25)])
def test_gemm_conv_same_padding_per_utterance(dtype, B, S, Cin, N, k, gemm_variant):
    x = rnd(B, S, Cin, seed=7)
    w = rnd(N, Cin, k, seed=8, scale=(Cin * k) ** -0.5)
    b = rnd(N, seed=9)
    ref = F.conv1d(G.rounded(x, dtype).transpose(1, 2), G.rounded(w, dtype), b, padding="same").transpose(1, 2)
    got = G.gemm(dtype, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, taps=k, S=S).reshape(B, S, N)
    err = float((got - ref).abs().max())
    assert err <= tol(dtype, ref), (err, tol(dtype, ref))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", [0, 1, 3, 4, 6, 7], ids=["auto", "unfused128x128", "slab128", "slab192", "slab32", "slab64"])
@pytest.mark.parametrize("B,S,Cin,N,k,relu,use_res", [(3, 200, 256, 256, 3, True, False), (2, 333, 1024, 256, 1, False, True),
                                                     (2, 70, 64, 192, 5, True, True), (1, 1536, 256, 256, 9, False, True),
                                                     (2, 50, 768, 768, 1, False, True), (2, 333, 768, 768, 1, False, True),
                                                     (2, 300, 1024, 1024, 3, True, False), (1, 700, 3072, 768, 1, False, True),
                                                     (3, 77, 128, 320, 3, True, True)])
def test_gemm_fused_layernorm_epilogue(dtype, variant, B, S, Cin, N, k, relu, use_res):
    """conv/GEMM -> (+ReLU) -> (+residual) -> LayerNorm (-> predictor head): fused in the slab kernel's epilogue for rows of
    N <= 256 channels (one column tile); wider rows (768 / 1024: BASELINE configs C3 / C5) as a GEMM launch + the LayerNorm kernel."""
    x = rnd(B, S, Cin, seed=50)
    w = rnd(N, Cin, k, seed=51, scale=(Cin * k) ** -0.5)
    b, res = rnd(N, seed=52), rnd(B * S, N, seed=53)
    g, be = 1 + 0.2 * rnd(N, seed=54), 0.1 * rnd(N, seed=55)
    hw = rnd(N, seed=56, scale=N ** -0.5)
    mask = torch.zeros(B * S, dtype=torch.bool)
    mask[::5] = True
    z = F.conv1d(G.rounded(x, dtype).transpose(1, 2), G.rounded(w, dtype), b, padding="same").transpose(1, 2).reshape(B * S, N)
    if relu:
        z = torch.relu(z)
    if use_res:
        z = z + G.rounded(res, dtype)
    ref = F.layer_norm(z, (N,), g, be, 1e-5)
    pref = (ref @ hw + 0.3).masked_fill(mask, 0)
    G.lib().fs2_op_set_gemm_variant(variant)
    try:
        y, pred = G.gemm_ln(dtype, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, res if use_res else None, g, be,
                            taps=k, S=S, relu=relu, dot_w=hw, dot_b=0.3, mask=mask)
        _, pred2 = G.gemm_ln(dtype, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, res if use_res else None, g, be,
                             taps=k, S=S, relu=relu, dot_w=hw, dot_b=0.3, mask=mask, want_y=False)
    finally:
        G.lib().fs2_op_set_gemm_variant(0)
    assert float((y - ref).abs().max()) <= tol(dtype, ref, f32=5e-5, bf16=2.5e-2)
    ptol = 1e-4 if dtype == G.F32 else 3e-2
    assert float((pred - pref).abs().max()) <= ptol * (float(pref.abs().max()) + 1)
    assert torch.equal(pred, pred2)


@pytest.mark.parametrize("B,S,Cin,N,k,ln", [(2, 300, 256, 1024, 9, False), (1, 1536, 256, 256, 3, True), (3, 100, 1024, 256, 1, True),
                                             (2, 64, 256, 768, 1, False)])
def test_gemm_split_bf16x3_arithmetic(B, S, Cin, N, k, ln):
    """fp32 operands as bf16 head + tail, three bf16 MFMAs per product (knob 501; the front of FS2_MIXED_X3): against an
    fp64 reference the error must sit ~2 orders of magnitude under plain bf16 operands' and within ~1e-5 relative."""
    x, w, b = rnd(B, S, Cin, seed=70), rnd(N, Cin, k, seed=71, scale=(Cin * k) ** -0.5), rnd(N, seed=72)
    z = F.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding="same").transpose(1, 2).reshape(B * S, N)
    g, be = 1 + 0.2 * rnd(N, seed=74), 0.1 * rnd(N, seed=75)
    ref = F.layer_norm(torch.relu(z), (N,), g.double(), be.double(), 1e-5) if ln else z
    out = {}
    for knob in (500, 501):
        G.lib().fs2_op_set_gemm_variant(knob)
        try:
            if ln:
                out[knob] = G.gemm_ln(G.F32, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, None, g, be, taps=k, S=S, relu=True)[0]
            else:
                out[knob] = G.gemm(G.F32, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, taps=k, S=S)
        finally:
            G.lib().fs2_op_set_gemm_variant(500)
    scale = float(ref.abs().max())
    e32, e3 = float((out[500].double() - ref).abs().max()) / scale, float((out[501].double() - ref).abs().max()) / scale
    assert e32 <= 2e-6 and e3 <= 4e-5, (e32, e3)
    assert not torch.equal(out[500], out[501])  # the knob did select another arithmetic


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", [6, 7, 3], ids=["slab32", "slab64", "slab128"])
def test_gemm_fused_layernorm_residual_repeatable(dtype, variant):
    """The residual is fetched into the accumulators ahead of the DMA'd K loop; a missing DMA wait on
    the loop's back-edge barrier showed up here as sporadic stale-operand rows (about 1 launch in 15
    at fp32, 64-row tiles).  Many reruns must be bit-identical and correct."""
    B, S, Cin, N = 2, 333, 1024, 256
    x, w = rnd(B * S, Cin, seed=50), rnd(N, Cin, 1, seed=51, scale=Cin ** -0.5)
    b, res = rnd(N, seed=52), rnd(B * S, N, seed=53)
    g, be = 1 + 0.2 * rnd(N, seed=54), 0.1 * rnd(N, seed=55)
    z = G.rounded(x, dtype) @ G.rounded(w[:, :, 0], dtype).T + b + G.rounded(res, dtype)
    ref = F.layer_norm(z, (N,), g, be, 1e-5)
    G.lib().fs2_op_set_gemm_variant(variant)
    try:
        runs = [G.gemm_ln(dtype, x, G.pack_conv_weight(w), b, res, g, be)[0] for _ in range(25)]
    finally:
        G.lib().fs2_op_set_gemm_variant(0)
    assert float((runs[0] - ref).abs().max()) <= tol(dtype, ref, f32=5e-5, bf16=2.5e-2)
    for r in runs[1:]:
        assert torch.equal(r, runs[0])


def _predictor_ref(x, ws, bs, gs, bes, hw, hb, mask):
    """model.py:510-522 with the kernel's storage rounding: bf16 x / weights / inter-layer
    activations, fp32 arithmetic."""
    r = lambda t: t.to(torch.bfloat16).float()
    h = r(x)
    for j, (w, b, g, be) in enumerate(zip(ws, bs, gs, bes)):
        z = F.conv1d(h.transpose(1, 2), r(w), b, padding="same").transpose(1, 2)
        h = F.layer_norm(torch.relu(z), (x.shape[-1],), g, be, 1e-5)
        if j + 1 < len(ws):
            h = r(h)
    return (h @ hw + hb).masked_fill(mask, 0)


@pytest.mark.parametrize("B,S,nl", [(2, 300, 5), (3, 37, 2), (1, 1536, 5), (2, 217, 1), (4, 216, 5), (2, 450, 3), (5, 5, 5)])
def test_predictor_single_launch(B, S, nl):
    """n x [conv k=3 -> ReLU -> LN] -> Linear -> mask in ONE launch with the activations resident in
    LDS (predictor_fused.hip): tile seams (halo rows), utterance edges (zero padding at every
    layer, not only the first) and masking against a torch restatement."""
    H = 256
    x = rnd(B, S, H, seed=70)
    ws = [rnd(H, H, 3, seed=71 + j, scale=(3 * H) ** -0.5) for j in range(nl)]
    bs = [0.3 * rnd(H, seed=80 + j) for j in range(nl)]
    gs = [1 + 0.2 * rnd(H, seed=90 + j) for j in range(nl)]
    bes = [0.1 * rnd(H, seed=100 + j) for j in range(nl)]
    hw, hb = rnd(H, seed=110, scale=H ** -0.5), 0.25
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[:, S - S // 4:] = True
    ref = _predictor_ref(x, ws, bs, gs, bes, hw, hb, mask)
    got = G.predictor(x, ws, bs, gs, bes, hw, hb, mask, B, S)
    assert not torch.isnan(got).any()
    assert torch.equal(got[mask], torch.zeros_like(got[mask]))
    err = float((got - ref).abs().max())
    assert err <= 2e-2 * (float(ref.abs().max()) + 1), err
    again = G.predictor(x, ws, bs, gs, bes, hw, hb, mask, B, S)
    assert torch.equal(again, got)


def _predictor_dw_ref(x, dws, dwbs, ws, bs, gs, bes, hw, hb, mask):
    """model.py:510-522 with depth-wise layers (:541-558) and the kernels' storage rounding: bf16 x / pointwise weights / depth-wise
    outputs / inter-layer activations, fp32 arithmetic and fp32 depth-wise taps."""
    r = lambda t: t.to(torch.bfloat16).float()
    h = r(x)
    H = x.shape[-1]
    for j, (dw, dwb, w, b, g, be) in enumerate(zip(dws, dwbs, ws, bs, gs, bes)):
        u = r(F.conv1d(h.transpose(1, 2), dw, dwb, padding=1, groups=H))
        z = F.conv1d(u, r(w), b).transpose(1, 2)
        h = F.layer_norm(torch.relu(z), (H,), g, be, 1e-5)
        if j + 1 < len(ws):
            h = r(h)
    return (h @ hw + hb).masked_fill(mask, 0)


def _dw_case(B, S, nl, seed):
    H = 256
    x = rnd(B, S, H, seed=seed)
    dws = [rnd(H, 1, 3, seed=seed + 1 + j, scale=3 ** -0.5) for j in range(nl)]
    dwbs = [0.2 * rnd(H, seed=seed + 20 + j) for j in range(nl)]
    ws = [rnd(H, H, 1, seed=seed + 40 + j, scale=H ** -0.5) for j in range(nl)]
    bs = [0.3 * rnd(H, seed=seed + 60 + j) for j in range(nl)]
    gs = [1 + 0.2 * rnd(H, seed=seed + 80 + j) for j in range(nl)]
    bes = [0.1 * rnd(H, seed=seed + 100 + j) for j in range(nl)]
    hw, hb = rnd(H, seed=seed + 120, scale=H ** -0.5), 0.25
    return x, dws, dwbs, ws, bs, gs, bes, hw, hb


@pytest.mark.parametrize("B,S,nl", [(2, 300, 5), (3, 37, 2), (1, 1536, 5), (2, 217, 1), (4, 216, 5), (2, 450, 3), (5, 5, 5), (26, 1536, 5)])
def test_depthwise_predictor_single_launch(B, S, nl):
    """r06 (the reference's own predictor architecture, model.py:541-558): n x [dw conv k=3 -> pointwise -> ReLU -> LN] -> Linear -> mask
    in ONE launch (predictor_fused_kernel<..., DW>: the depth-wise pass over the LDS-resident slab in place, one tap of the K loop):
    tile seams (one stale row per layer and side), utterance edges (zero padding at every layer's depth-wise conv), masking, both tile
    heights, against a torch restatement; repeatable."""
    x, dws, dwbs, ws, bs, gs, bes, hw, hb = _dw_case(B, S, nl, 400)
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[:, S - S // 4:] = True
    ref = _predictor_dw_ref(x, dws, dwbs, ws, bs, gs, bes, hw, hb, mask)
    got = G.predictor_dw(x, dws, dwbs, ws, bs, gs, bes, hw, hb, mask, B, S)
    assert not torch.isnan(got).any()
    assert torch.equal(got[mask], torch.zeros_like(got[mask]))
    err = float((got - ref).abs().max())
    assert err <= 2e-2 * (float(ref.abs().max()) + 1), err
    again = G.predictor_dw(x, dws, dwbs, ws, bs, gs, bes, hw, hb, mask, B, S)
    assert torch.equal(again, got)


def test_depthwise_predictor_tile_heights_are_bit_identical():
    """As for the dense form: an utterance gives the same bits alone (64-row tiles) and inside a batch of 26 (112-row tiles)."""
    B, S, nl = 26, 1536, 5
    x, dws, dwbs, ws, bs, gs, bes, hw, hb = _dw_case(B, S, nl, 600)
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, S - 11 * b:] = True
    whole = G.predictor_dw(x, dws, dwbs, ws, bs, gs, bes, hw, hb, mask, B, S)
    alone = G.predictor_dw(x[3:5], dws, dwbs, ws, bs, gs, bes, hw, hb, mask[3:5], 2, S)
    assert torch.equal(alone, whole[3:5])


def test_predictor_tile_heights_are_bit_identical():
    """Which tile height runs (112-row tiles where they fill the chip, 64-row tiles for small launches) must not enter the
    arithmetic: same wave layout, same reduction tree - an utterance gives the same bits alone (64-row tiles) and inside a batch
    of 26 (112-row tiles).  (r03's 208-row and paired-tile forms, pinned here until r04, were measured slower and are gone.)"""
    B, S, nl, H = 26, 1536, 5, 256
    x = rnd(B, S, H, seed=170)
    ws = [rnd(H, H, 3, seed=171 + j, scale=(3 * H) ** -0.5) for j in range(nl)]
    bs = [0.3 * rnd(H, seed=180 + j) for j in range(nl)]
    gs = [1 + 0.2 * rnd(H, seed=190 + j) for j in range(nl)]
    bes = [0.1 * rnd(H, seed=200 + j) for j in range(nl)]
    hw, hb = rnd(H, seed=210, scale=H ** -0.5), 0.25
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, S - 11 * b:] = True
    whole = G.predictor(x, ws, bs, gs, bes, hw, hb, mask, B, S)          # 26 x 15 tiles of 112 rows
    alone = G.predictor(x[3:5], ws, bs, gs, bes, hw, hb, mask[3:5], 2, S)  # 2 x 27 tiles of 64 rows
    assert torch.equal(alone, whole[3:5])
    ref = _predictor_ref(x[:3], ws, bs, gs, bes, hw, hb, mask[:3])
    assert float((whole[:3] - ref).abs().max()) <= 2e-2 * (float(ref.abs().max()) + 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,Cin,N,k,ks", [(8, 256, 1024, 256, 9, 4), (3, 200, 512, 256, 3, 2), (1, 640, 512, 512, 1, 8), (5, 77, 256, 192, 5, 2)])
def test_gemm_split_k_matches_the_one_pass_kernel(dtype, B, S, Cin, N, k, ks):
    """Split-K on the slab kernel (the encoder-side data-gradient convs of the training step: few row tiles, a long reduction): K
    slices of the input channels as workgroups of one launch, fp32 planes, plane sum.  Against torch, against the one-pass kernel
    (same products, another summation order: fp32 rounding apart), accumulating into an existing tensor, repeatable."""
    x = rnd(B, S, Cin, seed=300)
    w = rnd(N, Cin, k, seed=301, scale=(Cin * k) ** -0.5)
    ref = F.conv1d(G.rounded(x, dtype).transpose(1, 2), G.rounded(w, dtype), None, padding="same").transpose(1, 2).reshape(B * S, N)
    xs, wp = x.reshape(B * S, Cin), G.pack_conv_weight(w)
    one = G.gemm(dtype, xs, wp, None, taps=k, S=S, out_dtype=G.F32)
    runs = [G.gemm_splitk(dtype, xs, wp, ks, taps=k, S=S, out_dtype=G.F32) for _ in range(2)]
    assert torch.equal(runs[0], runs[1])
    assert float((runs[0] - ref).abs().max()) <= tol(dtype, ref)
    assert float((runs[0] - one).abs().max()) <= 1e-4 * (float(ref.abs().max()) + 1)
    base = rnd(B * S, N, seed=302)
    acc = G.gemm_splitk(dtype, xs, wp, ks, taps=k, S=S, into=base)   # out dtype = the activation dtype, out += product
    want = G.rounded(base, dtype) + runs[0]
    assert float((acc - want).abs().max()) <= (2e-2 if dtype == G.BF16 else 1e-5) * (float(want.abs().max()) + 1)
    # what the launcher offers for this shape is a split the kernel accepts
    c = int(G.lib().fs2_op_gemm_splitk_choice(dtype, B * S, N, Cin, k, S))
    assert c >= 1 and (c == 1 or (Cin // (64 if dtype == G.BF16 else 32)) % c == 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,Cin,N,k", [(4, 300, 256, 768, 1), (3, 217, 1024, 256, 9), (2, 1536, 256, 256, 3), (5, 40, 128, 200, 3)])
def test_gemm_with_addend_in_place(dtype, B, S, Cin, N, k):
    """c = x w^T + bias + addend with addend == c (the accumulating data-gradient products of the training step)."""
    x = rnd(B, S, Cin, seed=310)
    w = rnd(N, Cin, k, seed=311, scale=(Cin * k) ** -0.5)
    b = rnd(N, seed=312)
    base = rnd(B * S, N, seed=313)
    ref = F.conv1d(G.rounded(x, dtype).transpose(1, 2), G.rounded(w, dtype), b, padding="same").transpose(1, 2).reshape(B * S, N)
    ref = ref + G.rounded(base, dtype)
    xs, wp = x.reshape(B * S, Cin), G.pack_conv_weight(w)
    got = G.gemm_add(dtype, xs, wp, b, base, taps=k, S=S)
    again = G.gemm_add(dtype, xs, wp, b, base, taps=k, S=S)
    other = G.gemm_add(dtype, xs, wp, b, base, taps=k, S=S, in_place=False)
    assert torch.equal(got, again) and torch.equal(got, other)
    assert float((got - ref).abs().max()) <= tol(dtype, ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,Cin,N,k", [(2, 300, 256, 1024, 9), (3, 77, 128, 320, 3), (1, 512, 256, 256, 1)])
def test_gemm_relu_dropout_store_uses_the_dropout_op_mask(dtype, B, S, Cin, N, k):
    """c = dropout(relu(x w^T + b)) in the GEMM's store (training forward, the FFN's hidden tensor): the kept / dropped pattern is
    fs2_op_dropout's for the same (p, seed, key) over the (M, N) tensor - so the backward's dropout call regenerates it - and kept
    values are relu(.) / (1 - p)."""
    import ctypes as C
    pdrop, seed, key = 0.3, 99, 5
    x = rnd(B, S, Cin, seed=320)
    w = rnd(N, Cin, k, seed=321, scale=(Cin * k) ** -0.5)
    b = rnd(N, seed=322) + 0.5
    M = B * S
    xd, wd = G.to_dev(x.reshape(M, Cin), dtype), G.to_dev(G.pack_conv_weight(w), dtype)
    bd = b.float().to(G.DEV)
    c = torch.empty(M, N, dtype=G.tdt(dtype), device=G.DEV)
    G.ok(G.lib().fs2_op_gemm_relu_dropout(dtype, G.p(xd), G.p(wd), G.p(bd), G.p(c), M, N, Cin, k, S, C.c_float(pdrop), C.c_uint64(seed),
                                          C.c_uint64(key), G.stream()), "gemm_relu_dropout")
    ones = torch.ones(M, N, dtype=torch.float32, device=G.DEV)
    mask = torch.empty_like(ones)
    G.ok(G.lib().fs2_op_dropout(G.F32, G.p(ones), G.p(mask), M * N, C.c_float(pdrop), C.c_uint64(seed), C.c_uint64(key), G.stream()), "dropout")
    torch.cuda.synchronize()
    plain = G.gemm(dtype, x.reshape(M, Cin), G.pack_conv_weight(w), b, taps=k, S=S, relu=True)
    got, mask = c.float().cpu(), mask.cpu()
    want = plain * mask                     # mask holds 0 or 1 / (1 - p)
    assert 0.2 < float((mask == 0).float().mean()) < 0.4
    assert torch.equal(got == 0, want == 0) or float(((got == 0) != (want == 0)).float().mean()) < 1e-4   # (relu zeros aside, a value that rounds to 0)
    assert float((got - want).abs().max()) <= (2e-2 if dtype == G.BF16 else 1e-5) * (float(want.abs().max()) + 1)


def test_gemm_split_k_rejects_what_it_cannot_run():
    x, w = rnd(256, 256, seed=1), rnd(256, 256, seed=2)
    xd, wd = G.to_dev(x, G.BF16), G.to_dev(w, G.BF16)
    c = torch.empty(256, 256, dtype=torch.bfloat16, device=G.DEV)
    part = torch.empty(3, 256, 256, dtype=torch.float32, device=G.DEV)
    # 256 channels = 4 blocks of 64: 3 does not divide them
    assert G.lib().fs2_op_gemm_splitk(G.BF16, G.BF16, G.p(xd), G.p(wd), G.p(c), G.p(part), 256, 256, 256, 1, 256, 3, 0, G.stream()) != 0
    assert G.lib().fs2_op_gemm_splitk(G.BF16, G.BF16, G.p(xd), G.p(wd), G.p(c), None, 256, 256, 256, 1, 256, 2, 0, G.stream()) != 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dma_pipeline_large_and_repeatable(dtype):
    """Full-size decoder conv tile stream (K = 9*256 -> 36 chunks through the 3-stage DMA ring),
    many workgroups per CU in flight: compare with torch and demand bit-identical reruns (a DMA /
    barrier race shows up as run-to-run differences)."""
    B, S, Cin, N, k = 4, 1536, 256, 1024, 9
    x = rnd(B, S, Cin, seed=40)
    w = rnd(N, Cin, k, seed=41, scale=(Cin * k) ** -0.5)
    b = rnd(N, seed=42)
    ref = F.conv1d(G.rounded(x, dtype).transpose(1, 2), G.rounded(w, dtype), b, padding="same").transpose(1, 2)
    ref = torch.relu(ref)
    for variant in (2, 3, 4, 5):
        G.lib().fs2_op_set_gemm_variant(variant)
        try:
            runs = [G.gemm(dtype, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, taps=k, S=S, relu=True) for _ in range(3)]
        finally:
            G.lib().fs2_op_set_gemm_variant(0)
        for r in runs[1:]:
            assert torch.equal(r, runs[0]), variant
        err = float((runs[0].reshape(B, S, N) - ref).abs().max())
        assert err <= tol(dtype, ref), (variant, err, tol(dtype, ref))


# ------------------------------------------------------------------------------------------------
def _attn_ref(qkv, mask, B, S, H, heads):
    d = H // heads
    q, k, v = qkv.view(B, S, 3 * H).split(H, dim=-1)
    q = q.view(B, S, heads, d).transpose(1, 2) * (1.0 / math.sqrt(d))
    k = k.view(B, S, heads, d).transpose(1, 2)
    v = v.view(B, S, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)).masked_fill(mask[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, H)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,heads,mask_kind", [
    (2, 40, 64, 2, "suffix"), (3, 130, 128, 2, "suffix"), (2, 200, 256, 2, "suffix"),
    (2, 64, 256, 2, "none"), (2, 257, 256, 2, "scatter"), (1, 700, 256, 2, "suffix"), (2, 33, 128, 4, "scatter")])
def test_attention(dtype, B, S, H, heads, mask_kind):
    qkv = rnd(B * S, 3 * H, seed=10)
    mask = torch.zeros(B, S, dtype=torch.bool)
    if mask_kind == "suffix":
        for b in range(B):
            mask[b, S - (7 + 13 * b) % S:] = True
        mask[0, :] = False
        mask[0, S - S // 2:] = True   # a long padded tail: whole tiles get skipped
    elif mask_kind == "scatter":
        g = torch.Generator().manual_seed(11)
        mask = torch.rand(B, S, generator=g) < 0.3
        mask[:, 0] = False
    ref = _attn_ref(G.rounded(qkv, dtype), mask, B, S, H, heads)
    got = G.attention(dtype, qkv, mask, B, S, H, heads)
    err = float((got - ref).abs().max())
    assert err <= tol(dtype, ref, f32=5e-5, bf16=2e-2), (err, tol(dtype, ref))


@pytest.mark.parametrize("B,S,mask_kind", [(2, 40, "suffix"), (3, 130, "suffix"), (2, 200, "suffix"), (2, 64, "none"), (2, 257, "scatter"),
                                            (1, 700, "suffix"), (9, 256, "suffix"), (32, 256, "none"), (3, 1, "none")])
def test_encoder_attention_out_projection_layernorm_one_launch(B, S, mask_kind):
    """r06, the block the north star names (nn.MultiheadAttention's core + out_proj + residual + norm1 of ConformerEncoderLayer.forward,
    model.py:108-116): self-attention of both heads + out-projection + residual + LayerNorm in ONE launch (attn_out_ln_kernel; bf16,
    H = 256, two heads) against (a) torch, (b) the two launches it replaces (fs2_op_attention, then fs2_op_gemm_ln): the attention rows are
    the same bits, so the outputs differ by the out-projection's summation order only (fp32 rounding before the bf16 store); key padding
    of every shape, skipped tiles, ragged last query block, more utterances than XCDs, a single row; repeatable."""
    H, heads = 256, 2
    qkv = rnd(B * S, 3 * H, seed=10)
    w, bias = rnd(H, H, seed=12, scale=H ** -0.5), 0.3 * rnd(H, seed=13)
    res = rnd(B * S, H, seed=14)
    g, be = 1 + 0.2 * rnd(H, seed=15), 0.1 * rnd(H, seed=16)
    mask = torch.zeros(B, S, dtype=torch.bool)
    if mask_kind == "suffix":
        for b in range(B):
            mask[b, S - (7 + 13 * b) % S:] = True
        mask[0, :] = False
        mask[0, S - S // 2:] = True
    elif mask_kind == "scatter":
        gen = torch.Generator().manual_seed(11)
        mask = torch.rand(B, S, generator=gen) < 0.3
        mask[:, 0] = False
    r = lambda t: G.rounded(t, G.BF16)
    att = r(_attn_ref(r(qkv), mask, B, S, H, heads))
    ref = F.layer_norm(r(res) + att @ r(w).T + bias, (H,), g, be, 1e-5)
    got = G.attn_out_ln(qkv, mask, w, bias, res, g, be, B, S, H, heads)
    assert not torch.isnan(got).any()
    assert float((got - ref).abs().max()) <= 4e-2 * (float(ref.abs().max()) + 1)
    two = G.gemm_ln(G.BF16, G.attention(G.BF16, qkv, mask, B, S, H, heads), w, bias, res, g, be)[0]
    assert float((got - two).abs().max()) <= 2e-2 * (float(two.abs().max()) + 1)    # one bf16 ulp of an O(1) value at most
    assert float((got - two).abs().mean()) <= 1e-3
    assert torch.equal(G.attn_out_ln(qkv, mask, w, bias, res, g, be, B, S, H, heads), got)


@pytest.mark.parametrize("B,S,H,heads,mask_kind", [
    (2, 40, 64, 2, "suffix"), (3, 130, 128, 2, "suffix"), (2, 200, 256, 2, "suffix"), (2, 64, 256, 2, "none"),
    (2, 257, 256, 2, "scatter"), (1, 700, 256, 2, "suffix"), (2, 33, 128, 4, "scatter"), (4, 1536, 256, 2, "none")])
def test_attention_split_bf16x3(B, S, H, heads, mask_kind):
    """attention_kernel<bf16, .., X3> (r04): q, k, v as bf16 head + tail of the fp32 values, three bf16 MFMAs per q.k and p.v
    product, fp32 softmax / accumulators / output - the attention of the fp32x3 and mixed3 modes.  Against the fp64 reference it
    must sit where the fp32-MFMA kernel sits (bar 5e-5 of the output scale, the fp32 kernel's own), 400x inside the bf16 kernel's."""
    qkv = rnd(B * S, 3 * H, seed=10)
    mask = torch.zeros(B, S, dtype=torch.bool)
    if mask_kind == "suffix":
        for b in range(B):
            mask[b, S - (7 + 13 * b) % S:] = True
        mask[0, :] = False
        mask[0, S - S // 2:] = True
    elif mask_kind == "scatter":
        g = torch.Generator().manual_seed(11)
        mask = torch.rand(B, S, generator=g) < 0.3
        mask[:, 0] = False
    ref = _attn_ref(qkv.double(), mask, B, S, H, heads).float()
    got = G.attention_x3(qkv, mask, B, S, H, heads)
    again = G.attention_x3(qkv, mask, B, S, H, heads)
    assert torch.equal(got, again)
    err = float((got - ref).abs().max())
    assert err <= 5e-5 * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))


def test_attention_split_bf16x3_spike():
    B, S, H, heads = 1, 256, 256, 2
    qkv = rnd(B * S, 3 * H, seed=12)
    qkv[200, H:H + 128] = 6.0 * qkv[5, :128]  # a late, dominant key: the deferred rescale fires
    mask = torch.zeros(B, S, dtype=torch.bool)
    ref = _attn_ref(qkv.double(), mask, B, S, H, heads).float()
    got = G.attention_x3(qkv, mask, B, S, H, heads)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(300, 768, 256), (1, 192, 64), (33, 200, 96), (8192, 768, 256)])
@pytest.mark.parametrize("split", [True, False])
def test_gemm_head_tail_store(M, N, K, split):
    """GemmArgs::C_lo (r04): the fp32 in-projection leaves as two bf16 tensors, head + tail, hi + lo == the fp32 result up to
    2^-17 - and hi is the RNE bf16 of it, so the pair is exactly what split_bf16x3 would make of the fp32 tensor."""
    x, w, b = rnd(M, K, seed=31), rnd(N, K, seed=32) / K ** 0.5, rnd(N, seed=33)
    hi, lo = G.gemm_split_out(x, w, b, split)
    try:
        if split:
            G.lib().fs2_op_set_gemm_variant(501)
        full = G.gemm(G.F32, x, w, b)
    finally:
        G.lib().fs2_op_set_gemm_variant(500)
    # (the fp32 launch may run on another tile shape / kernel than the head + tail store's, so the sums agree to fp32 rounding only)
    tot = hi + lo
    assert bool((lo.abs() <= 2.0 ** -8 * hi.abs() + 1e-38).all())   # the tail is at most half an ulp of the head: hi = RNE(value)
    assert float((tot - full).abs().max()) <= (2.0 ** -16 + 2e-6) * float(full.abs().max())


def test_attention_spike_forces_rescale():
    # one key dominates late in the sequence: the running max jumps at a late tile (rule 26)
    B, S, H, heads = 1, 256, 256, 2
    qkv = rnd(B * S, 3 * H, seed=12)
    qkv[200, H:H + 128] = 6.0 * qkv[5, :128]  # key 200 aligned with query 5 (head 0)
    mask = torch.zeros(B, S, dtype=torch.bool)
    ref = _attn_ref(qkv, mask, B, S, H, heads)
    got = G.attention(G.F32, qkv, mask, B, S, H, heads)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.fixture
def pipe_kernel(request):
    """Force the software-pipelined attention kernel (attention_pipe.hip) for shapes of any size; 1203 = by size again."""
    G.lib().fs2_op_set_gemm_variant(1200 + request.param)
    yield request.param
    G.lib().fs2_op_set_gemm_variant(1203)


def _mask(kind, B, S):
    mask = torch.zeros(B, S, dtype=torch.bool)
    if kind == "suffix":
        for b in range(B):
            mask[b, S - (7 + 13 * b) % S:] = True
        mask[0, :] = False
        mask[0, S - S // 2:] = True   # a long padded tail: whole tiles are never visited
    elif kind == "scatter":
        mask = torch.rand(B, S, generator=torch.Generator().manual_seed(11)) < 0.3
        mask[:, 0] = False
    elif kind == "prefix":            # valid keys only at the END: leading all-padded tiles, a first tile whose first half is padded
        mask[:, :S - min(S, 40)] = True
    elif kind == "holes":             # whole 64-key tiles padded in the middle of the valid range
        mask[:, 64:192] = True
        mask[1 % B, 200:S - 3] = True
    return mask


@pytest.mark.parametrize("pipe_kernel", [1, 2, 4], indirect=True)
@pytest.mark.parametrize("B,S,H,heads,mask_kind", [
    (2, 200, 256, 2, "suffix"), (2, 64, 256, 2, "none"), (2, 257, 256, 2, "scatter"), (1, 700, 256, 2, "suffix"),
    (3, 130, 128, 1, "suffix"), (2, 33, 384, 3, "scatter"), (2, 300, 256, 2, "prefix"), (2, 450, 256, 2, "holes"),
    (1, 1536, 256, 2, "none"), (5, 129, 256, 2, "suffix"), (1, 1, 128, 1, "none")])
def test_attention_pipelined(pipe_kernel, B, S, H, heads, mask_kind):
    """The MFMA-bound instance's kernel (bf16, head dim 128) against the fp32 reference of the same op: every work-split shape
    (one / several 128-query units per head, ragged last unit, more workgroup slots than units), every mask shape."""
    qkv = rnd(B * S, 3 * H, seed=10)
    mask = _mask(mask_kind, B, S)
    ref = _attn_ref(G.rounded(qkv, G.BF16), mask, B, S, H, heads)
    got = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err <= tol(G.BF16, ref, f32=5e-5, bf16=2e-2), (err, tol(G.BF16, ref))
    G.lib().fs2_op_set_gemm_variant(1200)
    old = G.attention(G.BF16, qkv, mask, B, S, H, heads)   # the phase-serial kernel: same arithmetic contract
    assert float((got - old).abs().max()) <= tol(G.BF16, ref, f32=5e-5, bf16=2e-2)
    # 32, 64 or 96 queries per wave: the same instruction sequence per query row - bit-identical, whatever the launch size picks
    G.lib().fs2_op_set_gemm_variant(1200 + {1: 2, 2: 4, 4: 1}[pipe_kernel])
    other = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    assert torch.equal(got, other)


@pytest.mark.parametrize("pipe_kernel", [1, 2, 4], indirect=True)
def test_attention_pipelined_spike_and_all_padded(pipe_kernel):
    # (i) one key dominates late in the sequence: the running max jumps at a late half tile, in one query block only (rule 26:
    # the deferred rescale must scale O, l and nothing else exactly once); (ii) an utterance whose every key is padded gives NaN
    # rows, as the reference's softmax over all -inf does, and leaves its neighbours alone
    B, S, H, heads = 2, 512, 256, 2
    qkv = rnd(B * S, 3 * H, seed=12)
    qkv[300, H:H + 128] = 8.0 * qkv[5, :128]          # key 300 aligned with query 5 (utterance 0, head 0)
    qkv[S + 450, H + 128:H + 256] = -9.0 * qkv[S + 77, 128:256]
    mask = torch.zeros(B, S, dtype=torch.bool)
    x = G.rounded(qkv, G.BF16)
    ref = _attn_ref(x, mask, B, S, H, heads)
    got = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    mask[1, :] = True
    got = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    assert torch.isnan(got[S:]).all()
    assert float((got[:S] - ref[:S]).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("B,S,H,heads,mask_kind", [
    (32, 1024, 256, 2, "suffix"), (32, 1536, 256, 2, "holes"), (11, 1536, 768, 6, "suffix"), (16, 1024, 512, 4, "holes"),
    (32, 1536, 256, 2, "ragged")])
def test_attention_pipelined_decoder_sized_key_padding(B, S, H, heads, mask_kind):
    """Key padding through the decoder-sized launches: S in {1024, 1536}, B x heads >= 64, so a workgroup's
    run is whole 384-query triples (`attention_pipe_kernel<3>`) with padded tails, fully padded 64-key tiles in the middle of
    the valid range and - "ragged" - every utterance its own valid length U{S/2..S}, as a ragged batch's `tgt_mask` is
    (model.py:358-361).  Against the fp32 reference of the same op on the bf16-rounded operands; padded QUERY rows are computed
    like any other (the reference does, and they feed the conv halo of valid rows); the by-size dispatch (96 queries per wave)
    bit-equal to the 32-query form."""
    qkv = rnd(B * S, 3 * H, seed=21)
    if mask_kind == "ragged":
        g = torch.Generator().manual_seed(5)
        lens = torch.randint(S // 2, S + 1, (B,), generator=g)
        lens[0] = S
        mask = torch.arange(S)[None, :] >= lens[:, None]
    else:
        mask = _mask(mask_kind, B, S)
    ref = _attn_ref(G.rounded(qkv, G.BF16), mask, B, S, H, heads)
    got = G.attention(G.BF16, qkv, mask, B, S, H, heads)   # knob 1203 (default): by size -> the pipelined kernel
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err <= tol(G.BF16, ref, f32=5e-5, bf16=2e-2), (err, tol(G.BF16, ref))
    G.lib().fs2_op_set_gemm_variant(1201)
    try:
        other = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    finally:
        G.lib().fs2_op_set_gemm_variant(1203)
    assert torch.equal(got, other)


def test_attention_by_size_picks_the_pipelined_kernel_and_agrees():
    # the dispatch the engine uses (knob 1203): a decoder-sized problem goes to the pipelined kernel, same answers as the
    # phase-serial one up to bf16 rounding of P
    B, S, H, heads = 8, 1536, 256, 2
    qkv = rnd(B * S, 3 * H, seed=19)
    mask = _mask("suffix", B, S)
    got = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    G.lib().fs2_op_set_gemm_variant(1200)
    try:
        old = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    finally:
        G.lib().fs2_op_set_gemm_variant(1203)
    assert torch.isfinite(got).all()
    assert float((got - old).abs().max()) <= 2e-2 * float(old.abs().max())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,H", [(37, 64), (100, 256), (9, 768), (130, 1024), (5, 128)])
def test_layernorm_residual_and_head(dtype, M, H):
    x, r = rnd(M, H, seed=13, scale=2.0), rnd(M, H, seed=14)
    g, b = 1 + 0.2 * rnd(H, seed=15), 0.1 * rnd(H, seed=16)
    w = rnd(H, seed=17, scale=H ** -0.5)
    mask = torch.zeros(M, dtype=torch.bool)
    mask[::3] = True
    ref = F.layer_norm(G.rounded(x, dtype) + G.rounded(r, dtype), (H,), g, b, 1e-5)
    y, pred = G.layernorm(dtype, x, r, g, b, dot_w=w, dot_b=0.25, mask=mask)
    assert float((y - ref).abs().max()) <= tol(dtype, ref)
    pref = (ref @ w + 0.25).masked_fill(mask, 0)
    assert float((pred - pref).abs().max()) <= 5e-5 * (float(pref.abs().max()) + 1)
    y2, _ = G.layernorm(dtype, x, None, g, b)
    ref2 = F.layer_norm(G.rounded(x, dtype), (H,), g, b, 1e-5)
    assert float((y2 - ref2).abs().max()) <= tol(dtype, ref2)
    _, pred2 = G.layernorm(dtype, x, r, g, b, dot_w=w, dot_b=0.25, mask=mask, want_y=False)
    assert torch.equal(pred, pred2)


@pytest.mark.parametrize("M,N,Cin", [(1000, 768, 768), (77, 200, 128), (49152 // 8, 768, 768)])
def test_gemm_head_sums_are_layernorm_then_linear_head(M, N, Cin):
    """fs2_op_gemm_head + fs2_op_head_finish: the last VariancePredictor layer's ReLU -> LayerNorm -> Linear(N, 1) -> masked_fill from
    the GEMM epilogue's row sums, against torch on the fp32 product of the same bf16 operands; partial column tiles and row tiles."""
    x, w = rnd(M, Cin, seed=71), rnd(N, Cin, seed=72, scale=Cin ** -0.5)
    b = rnd(N, seed=73)
    g, be = 1 + 0.2 * rnd(N, seed=74), 0.1 * rnd(N, seed=75)
    wh = rnd(N, seed=76, scale=N ** -0.5)
    mask = torch.zeros(M, dtype=torch.bool)
    mask[::5] = True
    pred, st, hd = G.gemm_head(x, w, b, g, be, wh, 0.3, mask=mask)
    v = torch.relu(G.rounded(x, G.BF16) @ G.rounded(w, G.BF16).T + b)
    ref = (F.layer_norm(v, (N,), g, be, 1e-5) @ wh + 0.3).masked_fill(mask, 0)
    assert torch.isfinite(st).all() and torch.isfinite(hd).all()
    assert float((st[:, :, 0].sum(1) - v.sum(1)).abs().max()) <= 2e-3 * (float(v.sum(1).abs().max()) + 1)
    assert float((hd.sum(1) - v @ (g * wh)).abs().max()) <= 1e-4 * (float((v @ (g * wh)).abs().max()) + 1)
    assert float((pred - ref).abs().max()) <= 2e-4 * (float(ref.abs().max()) + 1)
    assert bool((pred[mask] == 0).all())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,C,k", [(3, 37, 64, 3), (2, 130, 256, 25), (2, 70, 768, 17), (1, 4, 64, 9), (2, 64, 128, 1),
                                     (2, 300, 128, 5), (1, 517, 64, 7), (2, 260, 192, 13), (1, 700, 64, 21), (1, 280, 72, 31), (1, 33, 64, 32)])
def test_dwconv(dtype, B, S, C, k):
    x, w, b = rnd(B, S, C, seed=18), rnd(C, 1, k, seed=19, scale=k ** -0.5), rnd(C, seed=20)
    ref = F.conv1d(G.rounded(x, dtype).transpose(1, 2), w, b, padding="same", groups=C).transpose(1, 2)
    got = G.dwconv(dtype, x.reshape(B * S, C), w, b, B, S).reshape(B, S, C)
    assert float((got - ref).abs().max()) <= tol(dtype, ref)


@pytest.mark.parametrize("k", [3, 9, 21])
def test_dwconv_tile_heights_are_bit_identical(k):
    """r06: a bf16 depth-wise launch that cannot give every CU three workgroups runs 128-row tiles, a larger one 256-row tiles (a
    performance choice by launch size).  The taps of an output are added in the same order either way: an utterance gives the same
    bits alone (4 x 2 tiles of 128 rows) and inside a batch of 64 (64 x 2 x 4 tiles of 256 rows)."""
    B, S, C = 64, 300, 256
    x, w, b = rnd(B, S, C, seed=180), rnd(C, 1, k, seed=190, scale=k ** -0.5), rnd(C, seed=200)
    whole = G.dwconv(G.BF16, x.reshape(B * S, C), w, b, B, S).reshape(B, S, C)
    alone = G.dwconv(G.BF16, x[5:6].reshape(S, C), w, b, 1, S).reshape(1, S, C)
    assert torch.equal(alone, whole[5:6])
    ref = F.conv1d(G.rounded(x[:3], G.BF16).transpose(1, 2), w, b, padding="same", groups=C).transpose(1, 2)
    assert float((whole[:3] - ref).abs().max()) <= tol(G.BF16, ref)


# ------------------------------------------------------------------------------------------------
def test_durations_round_guard_prefix():
    B, L = 6, 300
    g = torch.Generator().manual_seed(21)
    p = torch.rand(B, L, generator=g) * 2.4 - 0.3
    p[1] = p[1] * 0.1 - 0.2          # rounds to all zeros -> guard
    p[2, :] = math.log(1.5)          # exp(p)-1 = .5 (to rounding) -> half-even
    p[3, :] = math.log(3.5)
    mask = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        mask[b, L - 17 * b:] = b > 0
    p = p.masked_fill(mask, 0)
    ref, guarded = oracle_cpu.round_durations(p.clone(), mask)
    dur, cum, tot, grd = G.durations(p, mask)
    exact = (torch.exp(p) - 1)
    risky = ((exact - torch.floor(exact)) - 0.5).abs() < 1e-5  # GPU/CPU expf may differ by an ulp here
    assert torch.equal(dur[~risky], ref[~risky])
    assert sorted(guarded) == sorted(torch.nonzero(grd).flatten().tolist())
    assert torch.equal(cum, torch.cumsum(dur, 1).int())
    assert torch.equal(tot, dur.sum(1).int())
    forced = torch.randint(0, 9, (B, L), generator=g).int()
    dur2, cum2, tot2, grd2 = G.durations(p, mask, forced)
    assert torch.equal(dur2, forced) and torch.equal(cum2, torch.cumsum(forced, 1).int()) and int(grd2.sum()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cap", [None, 50])
def test_length_regulator(dtype, cap):
    B, L, H = 4, 23, 64
    g = torch.Generator().manual_seed(22)
    x = rnd(B, L, H, seed=23)
    dur = torch.randint(0, 7, (B, L), generator=g).int()
    dur[3, 10:] = 0
    ref, rmask = oracle_cpu.length_regulator(G.rounded(x, dtype), dur, cap if cap else 1e9)
    T = ref.shape[1]
    y, mk = G.regulate(dtype, x.reshape(B * L, H), torch.cumsum(dur, 1), dur.sum(1), B, L, T, H)
    assert torch.equal(mk, rmask)
    assert torch.equal(y.reshape(B, T, H), ref)  # a pure gather: bit exact in either dtype


@pytest.mark.parametrize("dtype", DTYPES)
def test_bucketize_embed_add(dtype):
    B, T, H, nb = 3, 100, 128, 256
    bins = torch.linspace(-3, 3, nb - 1)
    pred = rnd(B * T, seed=24, scale=1.5)
    pred[:nb - 1] = bins                      # exactly on every edge
    pred[nb:nb + 5] = torch.tensor([-5.0, 3.0, 3.01, 0.0, -3.0])
    emb, x = rnd(nb, H, seed=25), rnd(B * T, H, seed=26)
    pe, spk = rnd(T, H, seed=27), rnd(B, H, seed=28)
    std, mean = 1.3, -0.2
    idx_ref = torch.bucketize(pred * std + mean, bins)
    xr = G.rounded(x, dtype)
    ref = (xr + emb[idx_ref]).reshape(B, T, H) + pe[None] + spk[:, None]
    y, idx = G.bucket_embed(dtype, x, pred, bins, emb, std, mean, pe, spk, B, T, H)
    assert torch.equal(idx.long(), idx_ref)
    assert float((y.reshape(B, T, H) - ref).abs().max()) <= tol(dtype, ref, f32=1e-6)
    y2, _ = G.bucket_embed(dtype, x, None, None, None, 0, 0, pe, spk, B, T, H)
    ref2 = xr.reshape(B, T, H) + pe[None] + spk[:, None]
    assert float((y2.reshape(B, T, H) - ref2).abs().max()) <= tol(dtype, ref2, f32=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_pe_speaker(dtype):
    B, L, H, V = 3, 19, 256, 40
    g = torch.Generator().manual_seed(29)
    phones = torch.randint(1, V, (B, L), generator=g)
    phones[1, 12:] = 0
    phones[2, 3:] = 0
    table = rnd(V, H, seed=30)
    table[0] = 0
    pe, spk = rnd(L, H, seed=31), rnd(B, H, seed=32)
    ref = (table[phones] + pe[None]) + spk[:, None]
    x, mk = G.embed(dtype, phones, table, pe, spk, V)
    assert torch.equal(mk, phones.eq(0))
    assert float((x.reshape(B, L, H) - ref).abs().max()) <= tol(dtype, ref, f32=1e-6)


def test_speaker_projection():
    B, H, D = 5, 256, 256
    dv, w, b = rnd(B, D, seed=33), rnd(H, D, seed=34, scale=1 / 16), rnd(H, seed=35)
    ref = torch.relu(dv @ w.T + b)
    got = G.spk_proj(dv, w, b)
    assert float((got - ref).abs().max()) <= 1e-5


def test_slab_gemm_bits_do_not_depend_on_tile_height():
    """The launcher picks the tile height (MI variant) from the row count; the same rows must give the same bits
    through every variant - the forward's shard == whole property rests on it.  (A packed mul + add that the
    compiler chose in ONE template instantiation of the LayerNorm epilogue broke this once: the fused
    multiply-adds there are explicit now.)"""
    torch.manual_seed(0)
    S = 1536
    for (K, N, taps, ln, relu, res) in ((256, 256, 1, True, False, True), (1024, 256, 1, True, False, True),
                                        (256, 256, 3, True, True, False), (256, 1024, 9, False, True, False),
                                        (256, 768, 1, False, False, False)):
        x = torch.randn(32 * S, K)
        w = torch.randn(N, K * taps) / (K * taps) ** 0.5
        b, g, be, hw = torch.randn(N), torch.randn(N), torch.randn(N), torch.randn(N)
        r = torch.randn(32 * S, N) if res else None
        outs, preds = [], []
        for nb in (32, 8, 1):
            if ln:
                y, pr = G.gemm_ln(G.BF16, x[:nb * S], w, b, None if r is None else r[:nb * S], g, be, taps=taps, S=S, relu=relu,
                                     dot_w=hw, dot_b=0.3)
                preds.append(pr[:S])
            else:
                y = G.gemm(G.BF16, x[:nb * S], w, b, taps=taps, S=S, relu=relu)
            outs.append(y[:S])
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (K, N, taps, ln, relu)
        for q in preds[1:]:
            assert torch.equal(preds[0], q), (K, N, taps, "head")


def test_attention_pipelined_rerun_pass_in_64_query_items():
    """The rare path of the pipelined kernel at the size where workgroups run 256-query items (64 queries per wave, O in the
    accumulator file): scores far above a row's reference (the max over its first 32 keys) - one row beyond 2^100 (its item
    runs a second pass for that row alone), several rows by 2^20 .. 2^60 (no second pass: bf16 P keeps its 8 bits at any scale) -
    next to ragged key masks; every row against the fp32 reference, and the other rows of the same item bit-equal to a launch
    without the spike (a row's bits may not depend on its neighbours)."""
    B, S, H, heads = 16, 1536, 256, 2
    qkv = rnd(B * S, 3 * H, seed=21)
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, S - 37 * b:] = True
    clean = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    spiked = qkv.clone()
    u = 3 * S                                               # utterance 3
    spiked[u + 700, H:H + 128] = 8.0 * qkv[u + 300, :128]   # key 700, head 0: ~130 log2 units above query 300's other scores
    spiked[u + 900, H + 128:H + 256] = 2.5 * qkv[u + 1000, 128:256]  # head 1: ~40 log2 units for query 1000
    ref = _attn_ref(G.rounded(spiked, G.BF16), mask, B, S, H, heads)
    got = G.attention(G.BF16, spiked, mask, B, S, H, heads)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    # utterances other than 3 see the same inputs: bit-equal; inside utterance 3 only rows whose own scores changed may differ
    other = torch.ones(B * S, dtype=torch.bool)
    other[u:u + S] = False
    assert torch.equal(got[other], clean[other])
    for knob in (1201, 1202, 1204):                         # 32 / 64 / 96 queries per wave: the same rows, bit for bit
        G.lib().fs2_op_set_gemm_variant(knob)
        try:
            other_shape = G.attention(G.BF16, spiked, mask, B, S, H, heads)
        finally:
            G.lib().fs2_op_set_gemm_variant(1203)
        assert torch.equal(got, other_shape), knob


@pytest.mark.parametrize("pipe_kernel", [1, 2, 4], indirect=True)
def test_attention_pipelined_rows_far_below_and_above_zero(pipe_kernel):
    """The first pass of the pipelined kernel exponentiates the scaled scores as they are (reference 0).  Rows whose scores all lie
    ~160 log2 units BELOW zero (denominator underflows) and rows ~160 above (overflows) rerun from the reference log2(denominator);
    rows in between do not.  All against the fp32 reference."""
    B, S, H, heads = 1, 512, 128, 1
    qkv = rnd(B * S, 3 * H, seed=31)
    q = qkv[:, :H].clone()
    qkv[:, H:2 * H] *= 0.3
    # every key gets a large component along -q7 (query 7 sees all its scores ~ -160 log2) and along +q9 (query 9: ~ +160)
    d7, d9 = q[7] / q[7].norm(), q[9] / q[9].norm()
    qkv[:, H:2 * H] += (-1250.0 / float(q[7].norm())) * d7 + (1250.0 / float(q[9].norm())) * d9
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[0, S - 50:] = True
    x = G.rounded(qkv, G.BF16)
    ref = _attn_ref(x, mask, B, S, H, heads)
    sc = (x[:, :H] @ x[:, H:2 * H].T) * (1.4426950408889634 / H ** 0.5)
    assert float(sc[7, :S - 50].max()) < -120 and float(sc[9, :S - 50].min()) > 120   # the test really is on both far sides
    got = G.attention(G.BF16, qkv, mask, B, S, H, heads)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
