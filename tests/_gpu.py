"""Helpers for the `-m gpu` parity tests: call single operators of libfs2_hip.so through the C ABI."""
import ctypes as C

import numpy as np
import torch

from lightningfastspeech2_amd import _lib

F32, BF16 = _lib.FS2_F32, _lib.FS2_BF16
DEV = "cuda:0"


def lib():
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def ok(status, what=""):
    assert status == 0, f"{what}: status {status} ({lib().fs2_status_string(status).decode()})"


def to_dev(x, dtype):
    """fp32 host/torch tensor -> device tensor in the engine dtype (bf16 kept as torch.bfloat16)."""
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x).float().to(DEV)
    return t.to(torch.bfloat16).contiguous() if dtype == BF16 else t.contiguous()


def tdt(dtype):
    return torch.bfloat16 if dtype == BF16 else torch.float32


def rounded(x, dtype):
    """What the kernel actually sees: fp32 value of the (possibly bf16-rounded) input."""
    x = torch.as_tensor(x).float()
    return x.to(torch.bfloat16).float() if dtype == BF16 else x


def gemm(dtype, x, w_packed, bias, taps=1, S=None, relu=False, out_dtype=None):
    M, Cin = x.shape
    N = w_packed.shape[0]
    out_dtype = dtype if out_dtype is None else out_dtype
    xd, wd = to_dev(x, dtype), to_dev(w_packed, dtype)
    bd = None if bias is None else torch.as_tensor(bias).float().to(DEV)
    c = torch.empty(M, N, dtype=tdt(out_dtype), device=DEV)
    ok(lib().fs2_op_gemm(dtype, out_dtype, p(xd), p(wd), p(bd), p(c), M, N, Cin, taps, S or M, int(relu), stream()), "gemm")
    torch.cuda.synchronize()
    return c.float().cpu()


def gemm_add(dtype, x, w_packed, bias, addend, taps=1, S=None, in_place=True):
    M, Cin = x.shape
    N = w_packed.shape[0]
    xd, wd, ad = to_dev(x, dtype), to_dev(w_packed, dtype), to_dev(addend, dtype)
    bd = None if bias is None else torch.as_tensor(bias).float().to(DEV)
    c = ad if in_place else torch.empty(M, N, dtype=tdt(dtype), device=DEV)
    ok(lib().fs2_op_gemm_add(dtype, p(xd), p(wd), p(bd), p(ad), p(c), M, N, Cin, taps, S or M, stream()), "gemm_add")
    torch.cuda.synchronize()
    return c.float().cpu()


def gemm_rowscale(x, w_folded, bias_folded, parts, wg, eps=1e-5):
    """fs2_op_rowstats_finish + fs2_op_gemm_rowscale (bf16): x = pre-norm rows, parts (M, nparts, 2) fp32 partial (sum, sum of
    squares) per row as the deferred-LayerNorm epilogue leaves them"""
    M, Cin = x.shape
    N = w_folded.shape[0]
    xd, wd = to_dev(x, BF16), to_dev(w_folded, BF16)
    bd = torch.as_tensor(bias_folded).float().to(DEV).contiguous()
    sd = torch.as_tensor(parts).float().to(DEV).contiguous()
    gd = torch.as_tensor(wg).float().to(DEV).contiguous()
    rs = torch.empty(M, 2, dtype=torch.float32, device=DEV)
    ok(lib().fs2_op_rowstats_finish(p(sd), int(parts.shape[1]), Cin, float(eps), p(rs), M, stream()), "rowstats_finish")
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ok(lib().fs2_op_gemm_rowscale(p(xd), p(wd), p(bd), p(rs), p(gd), p(c), M, N, Cin, stream()), "gemm_rowscale")
    torch.cuda.synchronize()
    return c.float().cpu()


def gemm_head(x, w, bias, gamma, beta, w_head, b_head, mask=None, relu=True, eps=1e-5):
    """fs2_op_gemm_head + fs2_op_head_finish (bf16): pred = masked(LayerNorm(act(x w^T + bias)) . w_head + b_head) from the epilogue's
    row sums; also returns the statistic parts (M, nparts, 2) and the head sums (M, nparts)"""
    M, Cin = x.shape
    N = w.shape[0]
    nparts = (N + 255) // 256
    xd, wd = to_dev(x, BF16), to_dev(w, BF16)
    bd = torch.as_tensor(bias).float().to(DEV).contiguous()
    gw = (torch.as_tensor(gamma).float() * torch.as_tensor(w_head).float())
    gd = gw.to(DEV).contiguous()
    st = torch.full((M, nparts, 2), float("nan"), dtype=torch.float32, device=DEV)
    hd = torch.full((M, nparts), float("nan"), dtype=torch.float32, device=DEV)
    ok(lib().fs2_op_gemm_head(p(xd), p(wd), p(bd), p(gd), p(st), p(hd), M, N, Cin, int(relu), stream()), "gemm_head")
    md = None if mask is None else torch.as_tensor(mask).to(torch.uint8).to(DEV).contiguous()
    pred = torch.empty(M, dtype=torch.float32, device=DEV)
    cst = float((torch.as_tensor(beta).double() * torch.as_tensor(w_head).double()).sum() + b_head)
    ok(lib().fs2_op_head_finish(p(st), p(hd), nparts, N, float(eps), float(gw.double().sum()), cst, p(md), p(pred), M, stream()), "head_finish")
    torch.cuda.synchronize()
    return pred.cpu(), st.cpu(), hd.cpu()


def gemm_splitk(dtype, x, w_packed, ksplit, taps=1, S=None, out_dtype=None, into=None):
    """fs2_op_gemm_splitk: K slices as workgroups of one launch into fp32 planes + the plane sum; into = a tensor the result is
    ADDED to (the accumulating data-gradient call of the training step)."""
    M, Cin = x.shape
    N = w_packed.shape[0]
    out_dtype = dtype if out_dtype is None else out_dtype
    xd, wd = to_dev(x, dtype), to_dev(w_packed, dtype)
    c = torch.empty(M, N, dtype=tdt(out_dtype), device=DEV) if into is None else to_dev(into, out_dtype).clone()
    part = torch.empty(ksplit, M, N, dtype=torch.float32, device=DEV)
    ok(lib().fs2_op_gemm_splitk(dtype, out_dtype, p(xd), p(wd), p(c), p(part), M, N, Cin, taps, S or M, ksplit, int(into is not None),
                                stream()), "gemm_splitk")
    torch.cuda.synchronize()
    return c.float().cpu()


def gemm_ln(dtype, x, w_packed, bias, res, g, b, taps=1, S=None, relu=False, dot_w=None, dot_b=0.0, mask=None, want_y=True):
    M, Cin = x.shape
    N = w_packed.shape[0]
    f = lambda a: None if a is None else torch.as_tensor(a).float().to(DEV).contiguous()
    xd, wd = to_dev(x, dtype), to_dev(w_packed, dtype)
    rd = None if res is None else to_dev(res, dtype)
    y = torch.empty(M, N, dtype=tdt(dtype), device=DEV) if want_y else None
    tmp = torch.empty(M, N, dtype=tdt(dtype), device=DEV)
    pred = torch.empty(M, dtype=torch.float32, device=DEV) if dot_w is not None else None
    mk = None if mask is None else torch.as_tensor(mask).to(torch.uint8).to(DEV)
    bd, gd, bed, dwd = f(bias), f(g), f(b), f(dot_w)
    ok(lib().fs2_op_gemm_ln(dtype, p(xd), p(wd), p(bd), p(rd), p(gd), p(bed), p(dwd), float(dot_b), p(mk), p(pred),
                            p(y), p(tmp), M, N, Cin, taps, S or M, int(relu), stream()), "gemm_ln")
    torch.cuda.synchronize()
    return (None if y is None else y.float().cpu()), (None if pred is None else pred.cpu())


def predictor(x, ws, biases, gammas, betas, head_w, head_b, mask, B, S):
    """Single-launch VariancePredictor (bf16): ws = list of (H, H, k) conv weights (torch layout)."""
    H, nl, k = x.shape[-1], len(ws), ws[0].shape[2]
    xd = to_dev(x.reshape(B * S, H), BF16)
    wd = to_dev(torch.stack([pack_conv_weight(w) for w in ws]), BF16)
    f = lambda a: torch.stack([torch.as_tensor(v).float() for v in a]).to(DEV).contiguous()
    bd, gd, bed = f(biases), f(gammas), f(betas)
    hw = torch.as_tensor(head_w).float().to(DEV)
    mk = None if mask is None else torch.as_tensor(mask).to(torch.uint8).to(DEV).contiguous()
    pred = torch.full((B * S,), float("nan"), dtype=torch.float32, device=DEV)
    scratch = torch.empty(nl * H * k * H * 2, dtype=torch.uint8, device=DEV)
    ok(lib().fs2_op_predictor(BF16, p(xd), p(wd), p(bd), p(gd), p(bed), p(hw), float(head_b), p(mk), p(pred), p(scratch),
                              B, S, H, nl, k, stream()), "predictor")
    torch.cuda.synchronize()
    return pred.cpu().reshape(B, S)


def predictor_dw(x, dws, dwbs, ws, biases, gammas, betas, head_w, head_b, mask, B, S):
    """Single-launch depth-wise VariancePredictor (bf16): dws = list of (H, 1, 3) depth-wise weights, ws = list of (H, H, 1) pointwise."""
    H, nl = x.shape[-1], len(ws)
    xd = to_dev(x.reshape(B * S, H), BF16)
    wd = to_dev(torch.stack([w.reshape(H, H) for w in ws]), BF16)
    f = lambda a: torch.stack([torch.as_tensor(v).float() for v in a]).to(DEV).contiguous()
    dwd = f([w.reshape(H, 3).t().contiguous() for w in dws])   # (nl, 3, H) tap-major
    dbd, bd, gd, bed = f(dwbs), f(biases), f(gammas), f(betas)
    hw = torch.as_tensor(head_w).float().to(DEV)
    mk = None if mask is None else torch.as_tensor(mask).to(torch.uint8).to(DEV).contiguous()
    pred = torch.full((B * S,), float("nan"), dtype=torch.float32, device=DEV)
    scratch = torch.empty(nl * H * H * 2, dtype=torch.uint8, device=DEV)
    ok(lib().fs2_op_predictor_dw(BF16, p(xd), p(dwd), p(dbd), p(wd), p(bd), p(gd), p(bed), p(hw), float(head_b), p(mk), p(pred), p(scratch),
                                 B, S, H, nl, stream()), "predictor_dw")
    torch.cuda.synchronize()
    return pred.cpu().reshape(B, S)


def pack_conv_weight(w):
    """torch (N, Cin, k) -> (N, k*Cin) tap-major (what the engine builds at fs2_finalize)."""
    w = torch.as_tensor(w).float()
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def attention(dtype, qkv, key_pad_mask, B, S, H, heads):
    qd = to_dev(qkv, dtype)
    md = torch.as_tensor(key_pad_mask).to(torch.uint8).to(DEV).contiguous()
    out = torch.empty(B * S, H, dtype=tdt(dtype), device=DEV)
    bits_bytes = C.c_size_t()
    vt_bytes = lib().fs2_op_attention_scratch_bytes(dtype, B, S, H, heads, C.byref(bits_bytes))
    vt = torch.empty(vt_bytes, dtype=torch.uint8, device=DEV)
    bits = torch.empty(bits_bytes.value, dtype=torch.uint8, device=DEV)
    ok(lib().fs2_op_attention(dtype, p(qd), p(md), p(out), p(vt), p(bits), B, S, H, heads, stream()), "attention")
    torch.cuda.synchronize()
    return out.float().cpu()


def attn_out_ln(qkv, key_pad_mask, w_out, bias, res, gamma, beta, B, S, H, heads):
    """fs2_op_attn_out_ln (bf16): LayerNorm(res + MHA-core(qkv) w_out^T + bias)."""
    qd, rd, wd = to_dev(qkv, BF16), to_dev(res, BF16), to_dev(w_out, BF16)
    md = torch.as_tensor(key_pad_mask).to(torch.uint8).to(DEV).contiguous()
    f = lambda v: torch.as_tensor(v).float().to(DEV).contiguous()
    bd, gd, bed = f(bias), f(gamma), f(beta)
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    scratch = torch.empty(H * H * 2 + B * ((S + 63) // 64) * 8, dtype=torch.uint8, device=DEV)
    ok(lib().fs2_op_attn_out_ln(BF16, p(qd), p(md), p(wd), p(bd), p(rd), p(gd), p(bed), p(out), p(scratch), B, S, H, heads, stream()), "attn_out_ln")
    torch.cuda.synchronize()
    return out.float().cpu()


def attention_x3(qkv, key_pad_mask, B, S, H, heads):
    """fs2_op_attention_x3: fp32 qkv, bf16 x 3 split products, fp32 out."""
    qd = torch.as_tensor(qkv).float().to(DEV).contiguous()
    md = torch.as_tensor(key_pad_mask).to(torch.uint8).to(DEV).contiguous()
    out = torch.empty(B * S, H, dtype=torch.float32, device=DEV)
    split = torch.empty(2 * B * S * 3 * H, dtype=torch.bfloat16, device=DEV)
    bits = torch.empty(B * ((S + 63) // 64) * 8, dtype=torch.uint8, device=DEV)
    ok(lib().fs2_op_attention_x3(p(qd), p(md), p(out), p(split), p(bits), B, S, H, heads, stream()), "attention_x3")
    torch.cuda.synchronize()
    return out.cpu()


def gemm_split_out(x, w, bias, split=True):
    M, K = x.shape
    N = w.shape[0]
    xd, wd = torch.as_tensor(x).float().to(DEV).contiguous(), torch.as_tensor(w).float().to(DEV).contiguous()
    bd = None if bias is None else torch.as_tensor(bias).float().to(DEV)
    hi = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    lo = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ok(lib().fs2_op_gemm_split_out(p(xd), p(wd), p(bd), p(hi), p(lo), M, N, K, int(split), stream()), "gemm_split_out")
    torch.cuda.synchronize()
    return hi.float().cpu(), lo.float().cpu()


def layernorm(dtype, x, res, gamma, beta, dot_w=None, dot_b=0.0, mask=None, want_y=True):
    M, H = x.shape
    xd = to_dev(x, dtype)
    rd = None if res is None else to_dev(res, dtype)
    g, b = torch.as_tensor(gamma).float().to(DEV), torch.as_tensor(beta).float().to(DEV)
    y = torch.empty(M, H, dtype=tdt(dtype), device=DEV) if want_y else None
    dw = None if dot_w is None else torch.as_tensor(dot_w).float().to(DEV)
    mk = None if mask is None else torch.as_tensor(mask).to(torch.uint8).to(DEV)
    pred = torch.empty(M, dtype=torch.float32, device=DEV) if dot_w is not None else None
    ok(lib().fs2_op_layernorm(dtype, p(xd), p(rd), p(g), p(b), p(y), p(dw), float(dot_b), p(mk), p(pred), M, H, stream()), "layernorm")
    torch.cuda.synchronize()
    return (None if y is None else y.float().cpu()), (None if pred is None else pred.cpu())


def dwconv(dtype, x, w, bias, B, S):
    C_ = x.shape[1]
    k = w.shape[-1]
    xd = to_dev(x, dtype)
    wd = torch.as_tensor(w).float().reshape(C_, k).to(DEV).contiguous()
    bd = torch.as_tensor(bias).float().to(DEV)
    y = torch.empty(B * S, C_, dtype=tdt(dtype), device=DEV)
    ok(lib().fs2_op_dwconv(dtype, p(xd), p(wd), p(bd), p(y), B, S, C_, k, stream()), "dwconv")
    torch.cuda.synchronize()
    return y.float().cpu()


def durations(dur_pred, src_mask, forced=None):
    B, L = dur_pred.shape
    dp = torch.as_tensor(dur_pred).float().to(DEV).contiguous()
    mk = torch.as_tensor(src_mask).to(torch.uint8).to(DEV).contiguous()
    fd = None if forced is None else torch.as_tensor(forced).to(torch.int32).to(DEV).contiguous()
    dur = torch.empty(B, L, dtype=torch.int32, device=DEV)
    cum = torch.empty(B, L, dtype=torch.int32, device=DEV)
    tot = torch.empty(B, dtype=torch.int32, device=DEV)
    grd = torch.empty(B, dtype=torch.int32, device=DEV)
    ok(lib().fs2_op_durations(p(dp), p(mk), p(fd), p(dur), p(cum), p(tot), p(grd), B, L, stream()), "durations")
    torch.cuda.synchronize()
    return dur.cpu(), cum.cpu(), tot.cpu(), grd.cpu()


def regulate(dtype, x, cum, totals, B, L, T, H):
    xd = to_dev(x, dtype)
    cd = torch.as_tensor(cum).to(torch.int32).to(DEV).contiguous()
    td = torch.as_tensor(totals).to(torch.int32).to(DEV).contiguous()
    y = torch.empty(B * T, H, dtype=tdt(dtype), device=DEV)
    mk = torch.empty(B, T, dtype=torch.uint8, device=DEV)
    ok(lib().fs2_op_regulate(dtype, p(xd), p(cd), p(td), p(y), p(mk), B, L, T, H, stream()), "regulate")
    torch.cuda.synchronize()
    return y.float().cpu(), mk.bool().cpu()


def bucket_embed(dtype, x, pred, bins, emb, std, mean, pe, spk, B, T, H):
    xd = to_dev(x, dtype)
    f = lambda a: None if a is None else torch.as_tensor(a).float().to(DEV).contiguous()
    pd, bd, ed, ped, sd = f(pred), f(bins), f(emb), f(pe), f(spk)
    nb = 0 if emb is None else emb.shape[0]
    y = torch.empty(B * T, H, dtype=tdt(dtype), device=DEV)
    idx = torch.empty(B * T, dtype=torch.int32, device=DEV)
    ok(lib().fs2_op_bucket_embed(dtype, p(xd), p(pd), p(bd), p(ed), nb, float(std), float(mean), p(ped), p(sd), p(y),
                                 p(idx), B, T, H, stream()), "bucket_embed")
    torch.cuda.synchronize()
    return y.float().cpu(), idx.cpu()


def embed(dtype, phones, table, pe, spk, n_phones):
    B, L = phones.shape
    H = table.shape[1]
    ph = torch.as_tensor(phones).long().to(DEV).contiguous()
    f = lambda a: torch.as_tensor(a).float().to(DEV).contiguous()
    x = torch.empty(B * L, H, dtype=tdt(dtype), device=DEV)
    mk = torch.empty(B, L, dtype=torch.uint8, device=DEV)
    td, ped, sd = f(table), f(pe), f(spk)
    ok(lib().fs2_op_embed(dtype, p(ph), p(td), p(ped), p(sd), p(x), p(mk), B, L, H, n_phones, stream()), "embed")
    torch.cuda.synchronize()
    return x.float().cpu(), mk.bool().cpu()


def spk_proj(dvec, w, b):
    B, Din = dvec.shape
    H = w.shape[0]
    f = lambda a: torch.as_tensor(a).float().to(DEV).contiguous()
    dd, wd, bd = f(dvec), f(w), f(b)
    out = torch.empty(B, H, dtype=torch.float32, device=DEV)
    ok(lib().fs2_op_spk_proj(p(dd), p(wd), p(bd), p(out), B, H, Din, stream()), "spk_proj")
    torch.cuda.synchronize()
    return out.cpu()
