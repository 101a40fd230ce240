"""Host mirror of the reference's training step:

    FastSpeech2.training_step (litfass/fastspeech2/fastspeech2.py:786-797):  result = self(batch); losses = self.loss(result, batch)
    -> loss.backward() (Lightning) -> clip_grad_norm_(gradient_clip_val, scripts/train.sh:16)
    -> torch.optim.AdamW(lr, betas=[0.9, 0.98], eps=1e-8, weight_decay=0.01) + NoamLR (fastspeech2.py:1166-1182, noam.py:4-25)

built on the HIP operators behind the C ABI (include/fs2.h, "Training step" section): the teacher-forced forward keeps
every tensor the backward needs in HBM (288 GB: nothing is recomputed except the attention probabilities where the fused
attention + recomputing backward applies - bf16, head dim 128; elsewhere they are materialised),
the backward is a hand-written tape over fs2_op_bgemm / fs2_op_layernorm_bwd / fs2_op_attention_bwd / ..., parameters,
gradients and both Adam moments live in ONE flat fp32 buffer each so the optimizer is a single launch and a data-parallel
job all-reduces one contiguous gradient buffer.  torch supplies device memory and streams only; there is no autograd and
no CPU path - without libfs2_hip.so the constructor raises.

Covered: dense and depth-wise convolution variants of every block (C1-C5 of BASELINE.json, the reference's class defaults
and the test-size configs; the depth-wise layer's conv2 pair is trained through its folded (H, F) map), frame-level
'none' variances, 'l1' / 'mse' losses; precision "fp32" (exact fp32 MFMA, the parity mode) or "bf16" (bf16 activations,
activation gradients and GEMM operands on the bf16 MFMA with fp32 accumulation; fp32 master weights, weight gradients,
Adam moments, LayerNorm / softmax statistics and losses - what the reference's `--precision 16` recipe does with fp16).
Dropout: every nn.Dropout site of the reference in training mode (encoder / decoder incl. the attention weights and both positional encodings, variance / duration predictors)
with a counter-based mask that is regenerated, not stored; 0 by default, which is what the parity fixtures pin (the
reference's masks come from torch's random stream and cannot be reproduced).  Priors (PriorEmbedding) are trained.  Rejected loudly:
phone-level / CWT variances, stochastic durations.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import Fs2Config
from .weights import state_dict_spec

F32, BF16 = _lib.FS2_F32, _lib.FS2_BF16
_KIND = {"l1": 0, "mse": 1}
_PRECISIONS = {"fp32": (F32, torch.float32), "bf16": (BF16, torch.bfloat16)}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class _Ops:
    """Thin typed wrappers: torch tensors in, C ABI launches on torch's current stream."""

    def __init__(self, device, precision="fp32"):
        self.lib = _lib.load()
        self.dt, self.tdt = _PRECISIONS[precision]   # activation dtype (C ABI enum, torch dtype)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("the training step runs on the GPU (no CPU path)")
        self._ws: Dict[str, torch.Tensor] = {}
        self.seed = 0       # dropout: set per micro-step by the trainer
        self._site = 0      # dropout site counter of the current forward
        self.fuse_ln = os.environ.get("FS2_TRAIN_FUSE_LN", "1") != "0"  # A/B switch: GEMM + LayerNorm as one launch where it applies
        self.fuse_ln_drop = os.environ.get("FS2_TRAIN_FUSE_LN_DROPOUT", "1") != "0"  # A/B switch: the residual-site dropout inside that launch
        self.fuse_add = os.environ.get("FS2_TRAIN_FUSE_ADD", "1") != "0"  # A/B switch: "dx +=" of a data-gradient product inside the GEMM launch
        self.splitk = os.environ.get("FS2_TRAIN_SPLITK", "1") != "0"    # A/B switch: split-K data-gradient convs where fs2_op_gemm_splitk_choice says so
        # Weight gradients are leaves of the backward: nothing downstream reads them before the optimizer step.  With
        # FS2_TRAIN_WGRAD_STREAM=1 they run on a second HIP stream (dW GEMM + split-K reduce + bias column sums: ~110 of a step's
        # ~510 launches) beside the data-gradient chain, ordered by events: the side stream waits for dy, the main stream waits
        # for the side stream before it overwrites a tensor a pending product still reads (_guard) and at the end of the step.
        # Measured r04 at C2, bf16: 10.95 -> 10.5 ms per step (12.30 -> 11.7 with the recipe's dropout); fp32 (long MFMA-bound launches,
        # nothing to fill) 63.2 -> 64.3, so the default follows the precision.  FS2_TRAIN_WGRAD_STREAM=0 / 1 overrides.
        side_on = os.environ.get("FS2_TRAIN_WGRAD_STREAM", "1" if precision == "bf16" else "0") == "1"
        self.side = torch.cuda.Stream(self.dev) if side_on else None
        self._ws_tag = ""
        self._pending: Dict[int, "torch.cuda.Event"] = {}

    def site(self):
        self._site += 1
        return self._site

    def st(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def act(self, *shape):
        return torch.empty(*shape, dtype=self.tdt, device=self.dev)

    def to_act(self, x):
        """fp32 -> activation dtype (loss gradients enter the backward chain through this)"""
        if self.dt == F32:
            return x
        y = self.act(*x.shape)
        self.ck(self.lib.fs2_op_convert(F32, self.dt, _p(x), _p(y), x.numel(), self.st()), "convert")
        return y

    @staticmethod
    def _base(t):
        return t.untyped_storage().data_ptr()

    def _guard(self, *tensors):
        """about to WRITE these tensors on the current stream: wait for side-stream products that still read them"""
        if not self._pending:
            return
        for t in tensors:
            ev = self._pending.pop(self._base(t), None) if t is not None else None
            if ev is not None:
                torch.cuda.current_stream(self.dev).wait_event(ev)

    def join_side(self):
        """the current stream waits for everything queued on the weight-gradient stream"""
        if self.side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self._pending.clear()

    def ws(self, key: str, nbytes: int) -> Optional[torch.Tensor]:
        if nbytes <= 0:
            return None
        key = key + self._ws_tag  # the side stream's launches have workspaces of their own
        t = self._ws.get(key)
        if t is None or t.numel() * 4 < nbytes:
            # zeroed: the column-sum workspace starts with ticket counters that every launch returns to zero (include/fs2.h)
            t = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=self.dev)
            self._ws[key] = t
        return t

    def ck(self, status, what):
        _lib.check(status, None, what)

    # ---- forward operators (the inference path's own launches, fp32) ----
    def gemm(self, x, w, bias, M, N, Cin, taps=1, S=None, relu=False, out_f32=False, gate=None, gate_scale=1.0):
        y = self.empty(M, N) if out_f32 else self.act(M, N)
        if gate is not None:  # y = gate > 0 ? gate_scale * x w^T : 0 in the store, where the kernel has that epilogue
            st_ = self.lib.fs2_op_gemm_gated(self.dt, _p(x), _p(w), _p(bias), _p(gate), C.c_float(gate_scale), _p(y), M, N, Cin, taps,
                                             S or M, self.st())
            if st_ == 0:
                return y
            if st_ != _lib.FS2_ERR_SHAPE:  # only "no such epilogue for this shape" falls back; a launch failure is raised
                self.ck(st_, "gemm_gated")
            self.ck(self.lib.fs2_op_gemm(self.dt, self.dt, _p(x), _p(w), _p(bias), _p(y), M, N, Cin, taps, S or M, 0, self.st()), "gemm")
            return self.gate_(y, gate, gate_scale)
        self.ck(self.lib.fs2_op_gemm(self.dt, F32 if out_f32 else self.dt, _p(x), _p(w), _p(bias), _p(y), M, N, Cin, taps, S or M,
                                     int(relu), self.st()), "gemm")
        return y

    def gemm_relu_dropout(self, x, w, bias, M, N, Cin, taps, S, p, key):
        """dropout(relu(x W^T + bias)) - the mask of ``dropout(., p, key)`` - in one launch where the slab kernel runs the shape."""
        if p > 0 and self.fuse_ln_drop:
            y = self.act(M, N)
            st_ = self.lib.fs2_op_gemm_relu_dropout(self.dt, _p(x), _p(w), _p(bias), _p(y), M, N, Cin, taps, S or M, C.c_float(p),
                                                    C.c_uint64(self.seed), C.c_uint64(key), self.st())
            if st_ == 0:
                return y
            if st_ != _lib.FS2_ERR_SHAPE:
                self.ck(st_, "gemm_relu_dropout")
        return self.dropout(self.gemm(x, w, bias, M, N, Cin, taps=taps, S=S, relu=True), p, key)

    def gemm_ln_tape(self, x, w, bias, res, g, b, M, N, Cin, taps=1, S=None, relu=False, drop=None):
        """y = LayerNorm(z), z = act(x W^T + bias) [+ res], both stored, in ONE launch (the inference engine's fused GEMM +
        LayerNorm epilogue with a pre-norm store); None where that epilogue does not apply (N > 256, odd shapes, knob off).
        drop = (p, key): z = dropout(act(.)) + res with the mask of ``dropout(., p, key)`` (the residual sites in training mode)."""
        if not self.fuse_ln or N > 256:
            return None
        y, z = self.act(M, N), self.act(M, N)
        if drop is not None and drop[0] > 0:
            if not self.fuse_ln_drop:
                return None
            st_ = self.lib.fs2_op_gemm_ln_tape_dropout(self.dt, _p(x), _p(w), _p(bias), _p(res), _p(g), _p(b), _p(y), _p(z), M, N, Cin, taps,
                                                       S or M, int(relu), C.c_float(drop[0]), C.c_uint64(self.seed), C.c_uint64(drop[1]),
                                                       self.st())
        else:
            st_ = self.lib.fs2_op_gemm_ln_tape(self.dt, _p(x), _p(w), _p(bias), _p(res), _p(g), _p(b), _p(y), _p(z), M, N, Cin, taps,
                                               S or M, int(relu), self.st())
        if st_ == _lib.FS2_ERR_SHAPE:
            return None
        self.ck(st_, "gemm_ln_tape")
        return y, z

    def layernorm(self, x, res, g, b, M, H, dot_w=None, dot_b=0.0, mask=None, want_y=True):
        y = self.act(M, H) if want_y else None
        pred = self.empty(M) if dot_w is not None else None
        self.ck(self.lib.fs2_op_layernorm(self.dt, _p(x), _p(res), _p(g), _p(b), _p(y), _p(dot_w), C.c_float(dot_b), _p(mask),
                                          _p(pred), M, H, self.st()), "layernorm")
        return y, pred

    # ---- backward operators ----
    def bgemm(self, A, B, Cout, bias=None, **kw):
        """operands in A's dtype (both fp32 or both bf16); C fp32 or the activation dtype, by Cout's dtype"""
        d = _lib.BGemmDescC()
        base = dict(nb1=1, nb2=1, alpha=1.0, beta=0.0, splitk=1, taps=1, c_dtype=F32 if Cout.dtype == torch.float32 else BF16)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        if A.dtype != B.dtype:
            raise TypeError(f"bgemm operands differ: {A.dtype} vs {B.dtype}")
        ws = self.ws("bgemm", int(self.lib.fs2_op_bgemm_ws_bytes(C.byref(d))))
        self.ck(self.lib.fs2_op_bgemm(F32 if A.dtype == torch.float32 else BF16, C.byref(d), _p(A), _p(B), _p(Cout), _p(bias), _p(ws),
                                      self.st()), "bgemm")
        return Cout

    def tn256(self, **kw):
        """does this bf16 product run on the 256 x 256 tile kernel (include/fs2.h fs2_op_bgemm_tn256)"""
        d = _lib.BGemmDescC()
        base = dict(nb1=1, nb2=1, alpha=1.0, beta=0.0, splitk=1, taps=1, c_dtype=F32)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        return bool(self.lib.fs2_op_bgemm_tn256(C.byref(d)))

    @staticmethod
    def _dt(t):
        return F32 if t.dtype == torch.float32 else BF16

    def col_sum(self, x, out, M, N, seg=0, accumulate=True, scale=1.0, ldx=None):
        ws = self.ws("colsum", int(self.lib.fs2_op_col_sum_ws_bytes(M, N, seg)))
        self.ck(self.lib.fs2_op_col_sum(self._dt(x), _p(x), _p(out), _p(ws), M, N, ldx or N, seg, int(accumulate), C.c_float(scale),
                                        self.st()), "col_sum")

    def col_sum2(self, x, out, out2, n1, M, N, ldx=None, accumulate=True, accumulate2=True):
        """columns [0, n1) of the sums to out, [n1, N) to out2"""
        ws = self.ws("colsum", int(self.lib.fs2_op_col_sum_ws_bytes(M, N, 0)))
        self.ck(self.lib.fs2_op_col_sum2(self._dt(x), _p(x), _p(out), _p(out2), n1, _p(ws), M, N, ldx or N, int(accumulate),
                                         int(accumulate2), C.c_float(1.0), self.st()), "col_sum2")

    def scatter_rows(self, x, idx32, idx64, table, R, H, V, skip_row):
        """embedding backward: table[idx[r]] += x[r]"""
        ws = self.ws("scatter", int(self.lib.fs2_op_scatter_rows_ws_bytes(R, H, V)))
        self.ck(self.lib.fs2_op_scatter_rows(self._dt(x), _p(x), _p(idx32), _p(idx64), _p(table), _p(ws), R, H, V, skip_row, self.st()),
                "scatter_rows")

    def relu_bwd(self, dy, y):
        """in place: dy *= (y > 0)"""
        self._guard(dy)
        self.ck(self.lib.fs2_op_ew(self._dt(dy), 1, _p(dy), _p(y), _p(dy), dy.numel(), C.c_float(0), C.c_float(0), self.st()), "relu_bwd")
        return dy

    def gate_(self, dy, y, scale=1.0):
        """in place: dy = y > 0 ? scale * dy : 0"""
        self.relu_bwd(dy, y)
        if scale != 1.0:
            self.ck(self.lib.fs2_op_ew(self._dt(dy), 2, _p(dy), None, _p(dy), dy.numel(), C.c_float(scale), C.c_float(0), self.st()), "scale")
        return dy

    def dwconv(self, x, w, bias, B, S, Cc, k):
        y = self.act(B * S, Cc)
        self.ck(self.lib.fs2_op_dwconv(self.dt, _p(x), _p(w), _p(bias), _p(y), B, S, Cc, k, self.st()), "dwconv")
        return y

    def dwconv_bwd(self, du, x, w, gw, gb, B, S, Cc, k):
        """depth-wise conv backward: (gw (C, k), gb (C)) += the weight / bias gradients; returns d/dx (B*S, C)."""
        nparts = int(self.lib.fs2_op_dwconv_wgrad_parts(B, S))

        def wg():  # parameter gradients: a leaf (side stream where there is one)
            part = self.empty(nparts, Cc * (k + 1))
            self.ck(self.lib.fs2_op_dwconv_wgrad(self.dt, _p(du), _p(x), _p(part), B, S, Cc, k, self.st()), "dwconv_wgrad")
            if gb.data_ptr() == gw.data_ptr() + 4 * Cc * k:
                self.col_sum(part, gw, nparts, Cc * (k + 1))
            else:
                self.col_sum(part, gw, nparts, Cc * k, ldx=Cc * (k + 1))
                self.col_sum(part[:, Cc * k:], gb, nparts, Cc, ldx=Cc * (k + 1))
        self.on_side((du, x), wg)
        dx = self.act(B * S, Cc)
        self.ck(self.lib.fs2_op_dwconv_dgrad(self.dt, _p(du), _p(w), _p(dx), B, S, Cc, k, self.st()), "dwconv_dgrad")
        return dx

    def dropout(self, x, p, key, out=None):
        """nn.Dropout (training): in place unless `out` is given; mask = f(self.seed, key, element index), so calling it again
        with the same key on a gradient is the backward."""
        if p <= 0.0:
            return x
        y = x if out is None else out
        self._guard(y)
        self.ck(self.lib.fs2_op_dropout(self._dt(x), _p(x), _p(y), x.numel(), C.c_float(p), C.c_uint64(self.seed), C.c_uint64(key),
                                        self.st()), "dropout")
        return y

    def add_(self, a, b):
        self._guard(a)
        self.ck(self.lib.fs2_op_ew(self._dt(a), 0, _p(a), _p(b), _p(a), a.numel(), C.c_float(1), C.c_float(1), self.st()), "add")
        return a

    # y = x W^T + b backward pieces.  w is (N, taps*Cin) tap-major; x (M, Cin); dy (M, N); rows in utterances of S.
    def dgrad(self, dy, w, M, N, Cin, taps=1, S=None, out=None, accumulate=False, wt=None, gate=None, gate_scale=1.0):
        """dX (M, Cin) = dY (M, N) . W: with the transposed / tap-flipped copy wt (Cin, taps*N) through the forward
        GEMM / slab-conv kernel (dX is a 'same' conv of dY with wt), else through the strided-batched GEMM."""
        if out is not None:
            self._guard(out)
        if wt is not None and N % 64 == 0 and Cin % 64 == 0 and dy.dtype == wt.dtype:
            # a long reduction over few row tiles (encoder conv1: M = B L, K = taps x filter): K slices as workgroups of one launch
            # into fp32 planes, the plane sum folds the "+=" of an accumulating call (no separate add launch)
            ks = int(self.lib.fs2_op_gemm_splitk_choice(self.dt, M, Cin, N, taps, S or M)) if (gate is None and self.splitk) else 1
            if ks > 1:
                dx = out if out is not None else self.act(M, Cin)
                part = self.empty(ks, M, Cin)
                st_ = self.lib.fs2_op_gemm_splitk(self.dt, self._dt(dx), _p(dy), _p(wt), _p(dx), _p(part), M, Cin, N, taps, S or M, ks,
                                                  int(bool(accumulate and out is not None)), self.st())
                if st_ == 0:
                    return dx
                if st_ != _lib.FS2_ERR_SHAPE:
                    self.ck(st_, "gemm_splitk")
            if out is not None and accumulate and gate is None and self.fuse_add:
                # out += dy . wt in the GEMM's own store (the accumulators start at out's tile)
                st_ = self.lib.fs2_op_gemm_add(self.dt, _p(dy), _p(wt), None, _p(out), _p(out), M, Cin, N, taps, S or M, self.st())
                if st_ == 0:
                    return out
                if st_ != _lib.FS2_ERR_SHAPE:
                    self.ck(st_, "gemm_add")
            dx = self.gemm(dy, wt, None, M, Cin, N, taps=taps, S=S, gate=gate, gate_scale=gate_scale)
            if out is None:
                return dx
            if accumulate:
                self.ck(self.lib.fs2_op_ew(self._dt(out), 0, _p(out), _p(dx), _p(out), out.numel(), C.c_float(1), C.c_float(1), self.st()), "add")
            else:
                out.copy_(dx)
            return out
        if gate is not None and (out is not None or accumulate):
            raise ValueError("dgrad: a gate goes with a fresh output only")
        dx = out if out is not None else self.act(M, Cin)
        beta = 1.0 if accumulate else 0.0
        if taps == 1:
            self.bgemm(dy, w, dx, M=M, N=Cin, K=N, sAm=N, sAk=1, sBk=Cin, sBn=1, ldc=Cin, beta=beta)
        else:
            pad = (taps - 1) // 2
            self.bgemm(dy, w, dx, M=M, N=Cin, K=taps * N, sAm=N, sAk=1, sBk=taps * Cin, sBn=1, ldc=Cin, seg=S or M,
                       taps=taps, Kin=N, a_shift0=pad, a_shift_step=-1, sBtap=Cin, beta=beta)
        return self.gate_(dx, gate, gate_scale) if gate is not None else dx

    def wgrad(self, dy, x, dw, db, M, N, Cin, taps=1, S=None, sync=False):
        """dw (N, taps*Cin) += dy^T x (per tap, rows shifted inside their utterance); db (N) += column sums of dy.
        ``sync``: the caller reads dw right away (the folded depth-wise conv2) - no side stream for this one."""
        if sync:
            return self._wgrad(dy, x, dw, db, M, N, Cin, taps, S)
        self.on_side((dy, x), lambda: self._wgrad(dy, x, dw, db, M, N, Cin, taps, S))

    def on_side(self, reads, fn):
        """Run the launches of fn() - LEAF work of the backward: parameter gradients nothing reads before the optimizer step - on the
        side stream, behind everything queued on the current stream so far; `reads` = the tensors they read."""
        if self.side is None:
            return fn()
        cur = torch.cuda.current_stream(self.dev)
        ready = torch.cuda.Event()
        ready.record(cur)                      # the inputs are complete on the main stream here
        self.side.wait_event(ready)
        self._ws_tag = "@side"
        try:
            with torch.cuda.stream(self.side):
                fn()
                done = torch.cuda.Event()
                done.record(self.side)
        finally:
            self._ws_tag = ""
        for t in reads:
            if t is not None:
                t.record_stream(self.side)     # the allocator must not hand the block out again before the side stream is through
                self._pending[self._base(t)] = done

    def _wgrad(self, dy, x, dw, db, M, N, Cin, taps=1, S=None):
        pad = (taps - 1) // 2
        kw = dict(M=N, N=Cin, K=M, sAm=1, sAk=N, sBk=Cin, sBn=1, ldc=taps * Cin, nb2=taps, sC2=Cin,
                  seg=(S or M) if taps > 1 else 0, b_shift0=-pad, b_shift_step=1, beta=1.0)
        if dy.dtype == torch.bfloat16 and self.tn256(**kw):
            # 256 x 256 tiles, one 512-thread workgroup per CU: fill the 256 CUs once (measured: conv1 7 splits = 252 workgroups,
            # 199 us against 335 on the 128 x 128 kernel; a second, part-filled round costs more than it brings)
            tiles = (N // 256) * (Cin // 256) * taps
            splitk = max(1, min(256 // tiles, 64, M // 256))
        else:
            tiles = ((N + 127) // 128) * ((Cin + 127) // 128) * taps
            splitk = max(1, min(32, -(-2304 // tiles), M // 1024))  # ~9 workgroups per CU (measured: conv1 16, conv2 / in-proj 32)
        self.bgemm(dy, x, dw, splitk=splitk, **kw)
        if db is not None:
            self.col_sum(dy, db, M, N)


class Trainer:
    """One process per GPU.  ``training_step(batch)`` = teacher-forced forward + losses + backward into the flat gradient
    buffer (accumulating, as Lightning's accumulate_grad_batches does); ``optimizer_step()`` = clip + AdamW + Noam."""

    def __init__(self, cfg: Fs2Config, state_dict, *, lr=2e-4, warmup_steps=4000, betas=(0.9, 0.98), eps=1e-8,
                 weight_decay=0.01, gradient_clip_val: Optional[float] = 1.0, variance_losses=None, mel_loss="l1",
                 duration_loss="mse", loss_alphas=None, precision="fp32", encoder_dropout=0.0, decoder_dropout=0.0,
                 variance_dropout=0.0, duration_dropout=0.0, seed=0, attention="auto", device="cuda:0",
                 soft_dtw_gamma=0.01, soft_dtw_chunk_size=256):
        if any(l != "frame" for l in cfg.variance_levels[:len(cfg.variances)]) or any(cfg.is_cwt(i) for i in range(len(cfg.variances))):
            # (the reference cannot train the CWT head either: loss.py:141,148 read self.mse_loss, which its __init__ never sets)
            raise NotImplementedError("training step: frame-level 'none' variances only")
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self.cfg, self.precision = cfg, precision
        self.ops = _Ops(device, precision)
        self.dev = self.ops.dev
        self.lr, self.warmup_steps, self.betas, self.eps, self.weight_decay = lr, warmup_steps, betas, eps, weight_decay
        # Lightning (>= 1.5, what the reference pins) reads gradient_clip_val None or 0 as "no clipping"; negative is an error there too
        if gradient_clip_val is not None and float(gradient_clip_val) < 0:
            raise ValueError(f"gradient_clip_val must be >= 0, got {gradient_clip_val}")
        self.gradient_clip_val = float(gradient_clip_val) if gradient_clip_val else None
        # 'same' padding of an even kernel is asymmetric (left (k-1)//2, right k-1-left): the data gradients run on the forward
        # kernels with tap-flipped weights, which is the transpose only for odd k (and the inference engine rejects even k too)
        for what, ks in (("encoder_kernel_sizes", cfg.encoder_kernel_sizes[:cfg.encoder_layers]),
                         ("decoder_kernel_sizes", cfg.decoder_kernel_sizes[:cfg.decoder_layers]),
                         ("variance_kernel_size", list(cfg.variance_kernel_size)), ("duration_kernel_size", [cfg.duration_kernel_size])):
            if any(int(k) % 2 == 0 for k in ks):
                raise ValueError(f"training step: {what} = {list(ks)} holds an even kernel size (odd sizes only, as in the inference engine)")
        # nn.Dropout sites of the reference in training mode (fastspeech2.py:94,103 encoder/decoder_dropout incl. the attention
        # weights and both PositionalEncoding calls; :65,74 variance / duration predictors).  Reference defaults 0.1 / 0.1 / 0.5 /
        # 0.5, the shipped recipe 0.1 everywhere (scripts/train.sh:12-13); 0 here unless asked, which is what parity pins.
        nv_ = len(cfg.variances)
        self.p_enc, self.p_dec, self.p_dur = float(encoder_dropout), float(decoder_dropout), float(duration_dropout)
        self.p_var = [float(v) for v in variance_dropout][:nv_] if isinstance(variance_dropout, (list, tuple)) else [float(variance_dropout)] * nv_
        self.seed, self._micro = int(seed), 0
        if attention not in ("auto", "materialized"):
            raise ValueError("attention must be 'auto' or 'materialized'")
        self.attention = attention  # auto: the fused forward + recomputing backward where built (bf16, head dim 128)
        self.variance_losses = list(variance_losses) if variance_losses is not None else ["mse"] * len(cfg.variances)
        self.mel_loss, self.duration_loss = mel_loss, duration_loss
        self.loss_alphas = dict(loss_alphas) if loss_alphas is not None else {
            "mel": 1.0, "pitch": 1e-1, "energy": 1e-1, "snr": 1e-1, "duration": 1e-4}
        missing = [k for k in ["mel", "duration"] + list(cfg.variances) if k not in self.loss_alphas]
        if missing:
            raise ValueError(f"loss_alphas has no entry for {missing} (the default covers mel / pitch / energy / snr / duration)")
        for k in self.variance_losses + [mel_loss, duration_loss]:
            if k not in _KIND and k != "soft_dtw":
                raise NotImplementedError(f"training step: loss kind {k!r} has no gradient kernel ('l1' / 'mse' / 'soft_dtw')")
        self.soft_dtw_gamma, self.soft_dtw_chunk_size = float(soft_dtw_gamma), int(soft_dtw_chunk_size)
        # ---- flat parameter / gradient / moment buffers, named views in KERNEL layout (conv weights tap-major) ----
        spec = state_dict_spec(cfg)
        self._layout: "OrderedDict[str, tuple]" = OrderedDict()   # name -> (offset, kernel shape, reference shape)
        off = 0
        for name, shape in spec.items():
            if name == "positional_encoding.pe" or name.endswith(".bins"):
                continue
            kshape = (shape[0], shape[2] * shape[1]) if len(shape) == 3 else tuple(shape)
            n = int(np.prod(shape))
            self._layout[name] = (off, kshape, tuple(shape))
            off += (n + 3) // 4 * 4  # every tensor starts on a 16-byte boundary
        self.n_flat = off
        self.flat_p = torch.zeros(off, device=self.dev)
        self.flat_g = torch.zeros(off, device=self.dev)
        self.flat_m = torch.zeros(off, device=self.dev)
        self.flat_v = torch.zeros(off, device=self.dev)
        self.P = {n: self.flat_p[o:o + int(np.prod(ks))].view(ks) for n, (o, ks, _) in self._layout.items()}
        self.G = {n: self.flat_g[o:o + int(np.prod(ks))].view(ks) for n, (o, ks, _) in self._layout.items()}
        # GEMM operands: the fp32 masters themselves, or their bf16 shadow (refreshed after every optimizer step)
        if self.ops.dt == F32:
            self.W = self.P
        else:
            self.flat_w = torch.zeros(off, device=self.dev, dtype=torch.bfloat16)
            self.W = {n: self.flat_w[o:o + int(np.prod(ks))].view(ks) for n, (o, ks, _) in self._layout.items()}
        # transposed, tap-flipped copies of every GEMM weight (activation dtype): the data gradients run on the forward kernels
        self.use_forward_dgrad = os.environ.get("FS2_TRAIN_DGRAD", "forward") == "forward"
        self.flat_wt = torch.zeros(off, device=self.dev, dtype=self.ops.tdt)
        self.WT = {}
        self._tw_table = None
        for n, (o, ks, rs) in self._layout.items():
            gemm_w = n.endswith("weight") and "embedding" not in n and (
                len(rs) == 2 or (len(rs) == 3 and rs[1] > 1 and ".conv2.0." not in n))  # not LayerNorm (1-D), depth-wise or grouped convs
            if gemm_w:
                taps = rs[2] if len(rs) == 3 else 1
                self.WT[n] = self.flat_wt[o:o + int(np.prod(ks))].view(rs[1], taps * rs[0])
        # depth-wise ConformerEncoderLayer: conv2 = grouped 1x1 + pointwise folded into one (H, F) map per optimizer step
        self.fold: Dict[str, dict] = {}
        H_ = cfg.hidden
        for side, nl, dw, F_ in (("encoder", cfg.encoder_layers, cfg.encoder_depthwise_conv, cfg.encoder_conv_filter_size),
                                 ("decoder", cfg.decoder_layers, cfg.decoder_depthwise_conv, cfg.decoder_conv_filter_size)):
            if dw:
                for i in range(nl):
                    self.fold[f"{side}.layers.{i}"] = {"Wf": self.ops.act(H_, F_), "bf": self.ops.empty(H_), "WfT": self.ops.act(F_, H_)}
        self.buffers: Dict[str, torch.Tensor] = {}
        self.load_state_dict(state_dict)
        self.steps = 0          # optimizer steps taken
        self._accum = 0         # micro-batches in the gradient buffer
        self._loss_ws = torch.zeros(int(self.ops.lib.fs2_op_masked_loss_ws_bytes()), dtype=torch.uint8, device=self.dev)
        self._nsq = torch.zeros(1, device=self.dev)
        self._nsq_ws = self.ops.empty(int(self.ops.lib.fs2_op_sum_sq_ws_bytes(off)) // 4 + 1)

    # ---- state ----
    def load_state_dict(self, sd):
        for name, (o, ks, rs) in self._layout.items():
            v = sd[name]
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(torch.float32)
            if tuple(t.shape) != rs:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {rs}")
            if len(rs) == 3:
                t = t.permute(0, 2, 1).reshape(ks)
            self.P[name].copy_(t.contiguous())
        for name in state_dict_spec(self.cfg):
            if name == "positional_encoding.pe" or name.endswith(".bins"):
                v = sd[name]
                t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(torch.float32)
                self.buffers[name] = t.reshape(-1, t.shape[-1]).contiguous().to(self.dev) if name.endswith(".pe") else t.contiguous().to(self.dev)
        self._refresh_shadow()

    def _refresh_shadow(self, converted=False):
        """bf16 weight shadow (unless the optimizer step has just written it), folded depth-wise conv2 maps, and the transposed
        tap-flipped weights the data gradients run on - all of those in one launch for bf16."""
        o = self.ops
        with torch.cuda.device(self.dev):
            if o.dt != F32 and not converted:
                o.ck(o.lib.fs2_op_convert(F32, o.dt, _p(self.flat_p), _p(self.flat_w), self.n_flat, o.st()), "convert")
            for pfx, f in self.fold.items():
                Hh, Ff = f["Wf"].shape
                o.ck(o.lib.fs2_op_fold_conv2(o.dt, _p(self.P[f"{pfx}.conv2.0.weight"]), _p(self.P[f"{pfx}.conv2.0.bias"]),
                                             _p(self.P[f"{pfx}.conv2.1.weight"]), _p(self.P[f"{pfx}.conv2.1.bias"]), _p(f["Wf"]), _p(f["bf"]),
                                             Hh, Ff, o.st()), "fold_conv2")
            jobs = [(f["Wf"], f["WfT"], f["Wf"].shape[0], f["Wf"].shape[1], 1) for f in self.fold.values()]
            if self.use_forward_dgrad:
                for n, wt in self.WT.items():
                    _, _, rs = self._layout[n]
                    jobs.append((self.W[n], wt, rs[0], rs[1], rs[2] if len(rs) == 3 else 1))
            if not jobs:
                return
            if o.dt == F32 or os.environ.get("FS2_TRAIN_TW_BATCH", "1") == "0":  # (A/B switch)
                for src, dst, N, Cin, taps in jobs:
                    o.ck(o.lib.fs2_op_transpose_weight(o.dt, _p(src), _p(dst), N, Cin, taps, o.st()), "transpose_weight")
                return
            if self._tw_table is None:  # the buffers never move: build the device table once
                rows, tiles = [], 0
                for src, dst, N, Cin, taps in jobs:
                    rows.append([src.data_ptr(), dst.data_ptr(), N, Cin, taps, tiles])
                    tiles += int(o.lib.fs2_op_transpose_weight_tiles(N, Cin, taps))
                self._tw_table = (torch.tensor(rows, dtype=torch.int64, device=self.dev), len(rows), tiles)
            tab, n, tiles = self._tw_table
            o.ck(o.lib.fs2_op_transpose_weight_batch(_p(tab), n, tiles, o.st()), "transpose_weight_batch")

    def _wt(self, name):
        return self.WT.get(name) if self.use_forward_dgrad else None

    def _to_ref_layout(self, name, t):
        _, ks, rs = self._layout[name]
        t = t.detach().cpu()
        if len(rs) == 3:
            t = t.view(rs[0], rs[2], rs[1]).permute(0, 2, 1)
        return t.reshape(rs).contiguous()

    def state_dict(self):
        """Reference key names and layouts (feeds FastSpeech2(...) of this package or the reference's load_state_dict)."""
        out = OrderedDict((n, self._to_ref_layout(n, self.P[n])) for n in self._layout)
        for n, b in self.buffers.items():
            out[n] = b.cpu().reshape(1, *b.shape) if n.endswith(".pe") else b.cpu()
        return out

    def optimizer_state(self):
        """Adam moments under the reference's parameter names and layouts + the step counter (what a Lightning checkpoint keeps
        as ``optimizer_states`` / ``lr_schedulers``): ``{"step": n, "exp_avg": {name: tensor}, "exp_avg_sq": {name: tensor}}``.
        Together with ``state_dict()`` this resumes a run exactly."""
        def views(flat):
            return OrderedDict((n, self._to_ref_layout(n, flat[o:o + int(np.prod(ks))].view(ks))) for n, (o, ks, _) in self._layout.items())
        return {"step": self.steps, "micro_step": self._micro, "exp_avg": views(self.flat_m), "exp_avg_sq": views(self.flat_v)}

    def load_optimizer_state(self, st):
        for key, flat in (("exp_avg", self.flat_m), ("exp_avg_sq", self.flat_v)):
            for n, (o, ks, rs) in self._layout.items():
                t = torch.as_tensor(st[key][n]).to(torch.float32)
                if tuple(t.shape) != rs:
                    raise ValueError(f"{key}[{n}]: shape {tuple(t.shape)} != {rs}")
                if len(rs) == 3:
                    t = t.permute(0, 2, 1).reshape(ks)
                flat[o:o + int(np.prod(ks))].view(ks).copy_(t.contiguous())
        self.steps = int(st["step"])
        self._micro = int(st.get("micro_step", 0))

    def lightning_optimizer_state(self):
        """``{"optimizer_states": [...], "lr_schedulers": [...]}`` in the layout Lightning writes for the reference's
        ``AdamW`` + ``NoamLR`` (fastspeech2.py:1166-1182): a checkpoint made of ``state_dict()`` + these two entries resumes in
        the reference, and ``load_lightning_optimizer_state`` resumes a reference checkpoint here."""
        from .checkpoint import to_lightning_optimizer_state
        out = to_lightning_optimizer_state(self.cfg, self.optimizer_state(), lr=self.lr, warmup_steps=self.warmup_steps,
                                           betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)
        out["fs2_micro_step"] = int(self._micro)  # the dropout-mask counter (Lightning keeps none): an extra key the reference ignores
        return out

    def load_lightning_optimizer_state(self, checkpoint, accumulate_grad_batches: int = 1):
        """``accumulate_grad_batches``: the value the run that wrote ``checkpoint`` was trained with (Lightning's Trainer flag,
        scripts/train.sh) - the dropout-mask counter resumes at optimizer steps x that many micro-batches, unless the checkpoint
        carries this framework's own ``fs2_micro_step`` (``lightning_optimizer_state()`` writes it)."""
        from .checkpoint import from_lightning_optimizer_state
        self.load_optimizer_state(from_lightning_optimizer_state(self.cfg, checkpoint, int(accumulate_grad_batches)))

    def gradients(self):
        return OrderedDict((n, self._to_ref_layout(n, self.G[n])) for n in self._layout)

    def zero_grad(self):
        self.flat_g.zero_()
        self._accum = 0

    def current_lr(self) -> float:
        """NoamLR.get_lr (noam.py:19-25) for the optimizer step about to be taken: last_epoch = steps taken so far."""
        e = max(1, self.steps)
        return self.lr * self.warmup_steps ** 0.5 * min(e ** -0.5, e * self.warmup_steps ** -1.5)

    # ---- forward / backward of one ConformerEncoderLayer (model.py:65-122, post-norm) ----
    def _layer_fwd(self, x, prefix, B, S, heads, F_, k, key_pad, pd=0.0):
        o, P, W, H = self.ops, self.P, self.W, self.cfg.hidden
        M, d = B * S, H // heads
        t = {"x": x, "pd": pd, "k_attn": o.site(), "k_sa": o.site(), "k_h": o.site(), "k_ff": o.site()}
        qkv = o.gemm(x, W[f"{prefix}.self_attn.in_proj_weight"], P[f"{prefix}.self_attn.in_proj_bias"], M, 3 * H, H)
        flash = self.attention == "auto" and bool(o.lib.fs2_op_attention_bwd_supported(o.dt, H, heads))
        scale = 1.0 / math.sqrt(d)
        attn = o.act(M, H)
        prob = prob_d = lse = None
        if flash:  # fused attention (scores never reach HBM) + lse2 for the recomputing backward
            bb = C.c_size_t()
            vb = int(o.lib.fs2_op_attention_scratch_bytes(o.dt, B, S, H, heads, C.byref(bb)))
            vt, bits = o.ws("attn_vt", vb), o.ws("attn_bits", int(bb.value))
            lse = o.empty(B, heads, S)
            o.ck(o.lib.fs2_op_attention_train(o.dt, _p(qkv), _p(key_pad), _p(attn), _p(vt), _p(bits), _p(lse), B, S, H, heads,
                                              C.c_float(pd), C.c_uint64(o.seed), C.c_uint64(t["k_attn"]), o.st()), "attention_train")
        else:
            scores = o.empty(B, heads, S, S)  # fp32 in either mode; the probabilities are kept in the activation dtype
            o.bgemm(qkv, qkv[:, H:], scores, M=S, N=S, K=d, sAm=3 * H, sAk=1, sBk=1, sBn=3 * H, ldc=S, nb1=B, nb2=heads,
                    sA1=S * 3 * H, sA2=d, sB1=S * 3 * H, sB2=d, sC1=heads * S * S, sC2=S * S)
            prob = scores if o.dt == F32 else o.act(B, heads, S, S)
            o.ck(o.lib.fs2_op_softmax_fwd(o.dt, _p(scores), _p(key_pad), _p(prob), B, heads, S, C.c_float(scale), o.st()), "softmax")
            del scores
            prob_d = o.dropout(prob, pd, t["k_attn"], out=o.act(B, heads, S, S)) if pd > 0 else prob  # MHA drops attention weights
            o.bgemm(prob_d, qkv[:, 2 * H:], attn, M=S, N=d, K=S, sAm=S, sAk=1, sBk=3 * H, sBn=1, ldc=H, nb1=B, nb2=heads,
                    sA1=heads * S * S, sA2=S * S, sB1=S * 3 * H, sB2=d, sC1=S * H, sC2=d)
        # without dropout the out-projection, the residual add and norm1 are ONE launch that also leaves the pre-norm sum for the
        # tape (t["proj"] then holds x + proj and the backward's LayerNorm takes no separate residual)
        f1 = o.gemm_ln_tape(attn, W[f"{prefix}.self_attn.out_proj.weight"], P[f"{prefix}.self_attn.out_proj.bias"], x,
                            P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], M, H, H, drop=(pd, t["k_sa"]))  # (dropout1 inside)
        if f1 is not None:
            x1, proj = f1
        else:
            proj = o.dropout(o.gemm(attn, W[f"{prefix}.self_attn.out_proj.weight"], P[f"{prefix}.self_attn.out_proj.bias"], M, H, H),
                             pd, t["k_sa"])  # dropout1
            x1, _ = o.layernorm(proj, x, P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], M, H)
        t["sum1"] = f1 is not None
        if prefix in self.fold:  # depth-wise FFN (model.py:73-93): dw(k) -> pw H->F -> ReLU -> [grouped 1x1 . pw F->H] folded
            t["u"] = o.dwconv(x1, P[f"{prefix}.conv1.0.weight"], P[f"{prefix}.conv1.0.bias"], B, S, H, k)
            h = o.gemm_relu_dropout(t["u"], W[f"{prefix}.conv1.1.weight"], P[f"{prefix}.conv1.1.bias"], M, F_, H, 1, None, pd, t["k_h"])
            c2 = o.dropout(o.gemm(h, self.fold[prefix]["Wf"], self.fold[prefix]["bf"], M, H, F_), pd, t["k_ff"])
        else:
            h = o.gemm_relu_dropout(x1, W[f"{prefix}.conv1.weight"], P[f"{prefix}.conv1.bias"], M, F_, H, k, S, pd, t["k_h"])
            f2 = o.gemm_ln_tape(h, W[f"{prefix}.conv2.weight"], P[f"{prefix}.conv2.bias"], x1, P[f"{prefix}.norm2.weight"],
                                P[f"{prefix}.norm2.bias"], M, H, F_, drop=(pd, t["k_ff"]))  # (dropout2 inside)
            if f2 is not None:
                x2, c2 = f2
            else:
                c2 = o.dropout(o.gemm(h, W[f"{prefix}.conv2.weight"], P[f"{prefix}.conv2.bias"], M, H, F_), pd, t["k_ff"])  # dropout2
        t["sum2"] = prefix not in self.fold and f2 is not None
        if not t["sum2"]:
            x2, _ = o.layernorm(c2, x1, P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], M, H)
        t.update(qkv=qkv, prob=prob, prob_d=prob_d, lse=lse, key_pad=key_pad, attn=attn, proj=proj, x1=x1, h=h, c2=c2, scale=scale)
        return x2, t

    def _ln_bwd(self, z, res, dy, gname, bname, M, H, bias_name=None, relu_mask=False, bias_out=None, drop=None, out_drop=None):
        """dz of y = LN(z [+ res]); dgamma / dbeta (adjacent in the flat buffer: one column-sum launch) and, when asked, the
        bias gradient of the layer that produced z (= column sums of dz; with relu_mask dz is the pre-activation gradient).
        out_drop = (p, key): z = res + dropout(u) - returns (dz, dzm) with dzm = u's gradient (the mask applied in the same launch)
        and the bias gradient asked for is u's."""
        o = self.ops
        nparts = int(o.lib.fs2_op_layernorm_bwd_parts(M))
        dz, part = o.act(M, H), o.empty(nparts, 3 * H)
        dzm = None
        if out_drop is not None and out_drop[0] > 0:
            dzm = o.act(M, H)
            o.ck(o.lib.fs2_op_layernorm_bwd_masked(o.dt, _p(z), _p(res), _p(dy), _p(self.P[gname]), _p(dz), _p(dzm), _p(part), M, H,
                                                   int(relu_mask), C.c_float(out_drop[0]), C.c_uint64(o.seed), C.c_uint64(out_drop[1]),
                                                   o.st()), "layernorm_bwd")
        elif drop is not None and drop[0] > 0:  # dy is the gradient of dropout(y): the mask is applied on load, no separate pass
            o.ck(o.lib.fs2_op_layernorm_bwd_dropout(o.dt, _p(z), _p(res), _p(dy), _p(self.P[gname]), _p(dz), _p(part), M, H, int(relu_mask),
                                                    C.c_float(drop[0]), C.c_uint64(o.seed), C.c_uint64(drop[1]), o.st()), "layernorm_bwd")
        else:
            o.ck(o.lib.fs2_op_layernorm_bwd(o.dt, _p(z), _p(res), _p(dy), _p(self.P[gname]), _p(dz), _p(part), M, H, int(relu_mask),
                                            o.st()), "layernorm_bwd")
        gw, gb = self.G[gname], self.G[bname]

        def sums():
            if gb.data_ptr() == gw.data_ptr() + 4 * H and (bias_name is None) != (bias_out is None):
                # dgamma | dbeta (adjacent in the flat buffer) and the bias gradient out of one launch
                o.col_sum2(part, gw, self.G[bias_name] if bias_name is not None else bias_out, 2 * H, nparts, 3 * H,
                           accumulate2=bias_name is not None)
                return
            if gb.data_ptr() == gw.data_ptr() + 4 * H:
                o.col_sum(part, gw, nparts, 2 * H, ldx=3 * H)
            else:
                o.col_sum(part, gw, nparts, H, ldx=3 * H)
                o.col_sum(part[:, H:], gb, nparts, H, ldx=3 * H)
            if bias_name is not None:
                o.col_sum(part[:, 2 * H:], self.G[bias_name], nparts, H, ldx=3 * H)
            if bias_out is not None:
                o.col_sum(part[:, 2 * H:], bias_out, nparts, H, ldx=3 * H, accumulate=False)
        # (these column sums are leaves too, but on the side stream they measured SLOWER - 10.6 -> 11.0 ms per step: ~125 launches of
        # ~6 us each cost more in event records / waits than they free on the main chain; only the weight-gradient GEMMs go there)
        sums()
        return dz if dzm is None else (dz, dzm)

    def _layer_bwd(self, dx2, t, prefix, B, S, heads, F_, k):
        o, P, W, G, H = self.ops, self.P, self.W, self.G, self.cfg.hidden
        M, d = B * S, H // heads
        pd, folded = t["pd"], prefix in self.fold
        dbf = o.empty(H) if folded else None
        # x2 = LN2(x1 + dropout2(c2)): dz2 is the gradient of x1 (residual) and, through dropout2, of c2
        fm = pd > 0 and o.fuse_ln_drop  # the dropout backward of the sub-layer's summand as a second output of the LayerNorm backward
        dx1 = self._ln_bwd(t["c2"], None if t["sum2"] else t["x1"], dx2, f"{prefix}.norm2.weight", f"{prefix}.norm2.bias", M, H,
                           bias_name=f"{prefix}.conv2.bias" if ((pd <= 0 or fm) and not folded) else None,
                           bias_out=dbf if ((pd <= 0 or fm) and folded) else None, out_drop=(pd, t["k_ff"]) if fm else None)
        dc2 = dx1
        if fm:
            dx1, dc2 = dx1
        elif pd > 0:
            dc2 = o.dropout(dx1, pd, t["k_ff"], out=o.act(M, H))
            o.col_sum(dc2, dbf if folded else G[f"{prefix}.conv2.bias"], M, H, accumulate=not folded)
        if folded:
            f = self.fold[prefix]
            dWf = torch.zeros(H, F_, device=self.dev)
            o.wgrad(dc2, t["h"], dWf, None, M, H, F_, sync=True)
            o.ck(o.lib.fs2_op_unfold_conv2(_p(dWf), _p(dbf), _p(P[f"{prefix}.conv2.0.weight"]), _p(P[f"{prefix}.conv2.0.bias"]),
                                           _p(P[f"{prefix}.conv2.1.weight"]), _p(G[f"{prefix}.conv2.0.weight"]), _p(G[f"{prefix}.conv2.0.bias"]),
                                           _p(G[f"{prefix}.conv2.1.weight"]), _p(G[f"{prefix}.conv2.1.bias"]), H, F_, o.st()), "unfold_conv2")
        # dh = (dc2 . W2) o [h > 0] / (1 - p): h = dropout(relu(.)) of the forward is > 0 exactly where the element was kept AND the
        # pre-activation was positive, so the ReLU backward and the dropout backward ride in the product's store together
        gate = t["h"] if os.environ.get("FS2_TRAIN_GATE", "1") != "0" else None  # (A/B switch)
        gs = 1.0 / (1.0 - pd) if pd > 0 else 1.0
        if folded:
            dh = o.dgrad(dc2, f["Wf"], M, H, F_, wt=f["WfT"] if self.use_forward_dgrad else None, gate=gate, gate_scale=gs)
        else:
            o.wgrad(dc2, t["h"], G[f"{prefix}.conv2.weight"], None, M, H, F_)
            dh = o.dgrad(dc2, W[f"{prefix}.conv2.weight"], M, H, F_, wt=self._wt(f"{prefix}.conv2.weight"), gate=gate, gate_scale=gs)
        if gate is None:
            dh = o.relu_bwd(o.dropout(dh, pd, t["k_h"]), t["h"])  # h (post-dropout) > 0  <=>  kept and pre-activation > 0
        if folded:
            o.wgrad(dh, t["u"], G[f"{prefix}.conv1.1.weight"], G[f"{prefix}.conv1.1.bias"], M, F_, H)
            du = o.dgrad(dh, W[f"{prefix}.conv1.1.weight"], M, F_, H, wt=self._wt(f"{prefix}.conv1.1.weight"))
            o.add_(dx1, o.dwconv_bwd(du, t["x1"], P[f"{prefix}.conv1.0.weight"], G[f"{prefix}.conv1.0.weight"], G[f"{prefix}.conv1.0.bias"],
                                     B, S, H, k))
        else:
            o.wgrad(dh, t["x1"], G[f"{prefix}.conv1.weight"], G[f"{prefix}.conv1.bias"], M, F_, H, taps=k, S=S)
            o.dgrad(dh, W[f"{prefix}.conv1.weight"], M, F_, H, taps=k, S=S, out=dx1, accumulate=True, wt=self._wt(f"{prefix}.conv1.weight"))
        # x1 = LN1(x + dropout1(proj))
        dx = self._ln_bwd(t["proj"], None if t["sum1"] else t["x"], dx1, f"{prefix}.norm1.weight", f"{prefix}.norm1.bias", M, H,
                          bias_name=f"{prefix}.self_attn.out_proj.bias" if (pd <= 0 or fm) else None,
                          out_drop=(pd, t["k_sa"]) if fm else None)
        dproj = dx
        if fm:
            dx, dproj = dx
        elif pd > 0:
            dproj = o.dropout(dx, pd, t["k_sa"], out=o.act(M, H))
            o.col_sum(dproj, G[f"{prefix}.self_attn.out_proj.bias"], M, H)
        o.wgrad(dproj, t["attn"], G[f"{prefix}.self_attn.out_proj.weight"], None, M, H, H)
        dattn = o.dgrad(dproj, W[f"{prefix}.self_attn.out_proj.weight"], M, H, H, wt=self._wt(f"{prefix}.self_attn.out_proj.weight"))
        qkv, prob = t["qkv"], t["prob"]
        dqkv = o.act(M, 3 * H)
        delta = o.empty(B, heads, S)
        o.ck(o.lib.fs2_op_attn_delta(o.dt, _p(dattn), _p(t["attn"]), _p(delta), B, S, H, heads, o.st()), "attn_delta")
        if t["lse"] is not None:  # recomputing backward: P, dP, dS live in registers only
        # (its two launches - (dK, dV) and dQ - side by side on two streams measured nothing: 10.4-10.5 ms either way, r04)
            o.ck(o.lib.fs2_op_attention_bwd(o.dt, _p(qkv), _p(dattn), _p(t["lse"]), _p(delta), _p(t["key_pad"]), _p(dqkv), B, S, H, heads,
                                            C.c_float(pd), C.c_uint64(o.seed), C.c_uint64(t["k_attn"]), o.st()), "attention_bwd")
        else:
            bat = dict(nb1=B, nb2=heads)
            sP = dict(sA1=heads * S * S, sA2=S * S)
            # dV = dropout(P)^T dO
            o.bgemm(t["prob_d"], dattn, dqkv[:, 2 * H:], M=S, N=d, K=S, sAm=1, sAk=S, sBk=H, sBn=1, ldc=3 * H, sB1=S * H, sB2=d,
                    sC1=S * 3 * H, sC2=d, **bat, **sP)
            # dS = P o (dropout(dO V^T) - delta) / sqrt(d) in the product's epilogue: no dP tensor, no softmax-backward pass
            dp = o.act(B, heads, S, S)
            dsc = _lib.BGemmDescC()
            for k_, v_ in dict(M=S, N=S, K=d, sAm=H, sAk=1, sBk=1, sBn=3 * H, ldc=S, sA1=S * H, sA2=d, sB1=S * 3 * H, sB2=d,
                               sC1=heads * S * S, sC2=S * S, alpha=t["scale"], beta=0.0, splitk=1, taps=1, nb1=B, nb2=heads,
                               c_dtype=o.dt).items():
                setattr(dsc, k_, v_)
            o.ck(o.lib.fs2_op_bgemm_softmax_bwd(o.dt, C.byref(dsc), _p(dattn), _p(qkv[:, 2 * H:]), _p(dp), _p(prob), _p(delta),
                                                C.c_float(pd), C.c_uint64(o.seed), C.c_uint64(t["k_attn"]), o.st()), "bgemm_softmax_bwd")
            # dQ = dS K ; dK = dS^T Q
            o.bgemm(dp, qkv[:, H:], dqkv, M=S, N=d, K=S, sAm=S, sAk=1, sBk=3 * H, sBn=1, ldc=3 * H, sB1=S * 3 * H, sB2=d,
                    sC1=S * 3 * H, sC2=d, **bat, **sP)
            o.bgemm(dp, qkv, dqkv[:, H:], M=S, N=d, K=S, sAm=1, sAk=S, sBk=3 * H, sBn=1, ldc=3 * H, sB1=S * 3 * H, sB2=d,
                    sC1=S * 3 * H, sC2=d, **bat, **sP)
        o.wgrad(dqkv, t["x"], G[f"{prefix}.self_attn.in_proj_weight"], G[f"{prefix}.self_attn.in_proj_bias"], M, 3 * H, H)
        o.dgrad(dqkv, W[f"{prefix}.self_attn.in_proj_weight"], M, 3 * H, H, out=dx, accumulate=True, wt=self._wt(f"{prefix}.self_attn.in_proj_weight"))
        return dx

    # ---- VariancePredictor (model.py:482-561, dense) ----
    def _predictor_fwd(self, x, prefix, nlayers, filt, k, B, S, mask, dw=False, pd=0.0):
        o, P, W, H = self.ops, self.P, self.W, self.cfg.hidden
        M = B * S
        tape, y, cin = [], x, H
        for j in range(nlayers):
            p = f"{prefix}.layers.{j}.layers"
            u = None
            if dw:  # VarianceConvolutionLayer, depth-wise form (model.py:541-558): dw(k) -> pw 1x1 -> ReLU -> LN
                u = o.dwconv(y, P[f"{p}.0.module.0.weight"], P[f"{p}.0.module.0.bias"], B, S, cin, k)
                c = o.gemm(u, W[f"{p}.0.module.1.weight"], P[f"{p}.0.module.1.bias"], M, filt, cin, relu=True)
            last = j == nlayers - 1
            fl = None
            if not dw and pd <= 0 and not last:  # conv -> ReLU -> LayerNorm in one launch, the ReLU output kept for the tape
                fl = o.gemm_ln_tape(y, W[f"{p}.0.module.weight"], P[f"{p}.0.module.bias"], None, P[f"{p}.2.weight"], P[f"{p}.2.bias"],
                                    M, filt, cin, taps=k, S=S, relu=True)
            if fl is not None:
                yn, c = fl
                tape.append({"xin": y, "c": c, "cin": cin, "u": None, "kd": o.site()})
                y, cin = yn, filt
                continue
            if not dw:
                c = o.gemm(y, W[f"{p}.0.module.weight"], P[f"{p}.0.module.bias"], M, filt, cin, taps=k, S=S, relu=True)
            fused_head = last and pd <= 0
            kd = None
            if pd > 0:  # VarianceConvolutionLayer ends in nn.Dropout (model.py:539,557): inside the LayerNorm launch
                kd = o.site()
                yn, pred = o.act(M, filt), None
                o.ck(o.lib.fs2_op_layernorm_dropout(o.dt, _p(c), None, _p(P[f"{p}.2.weight"]), _p(P[f"{p}.2.bias"]), _p(yn), M, filt,
                                                    C.c_float(pd), C.c_uint64(o.seed), C.c_uint64(kd), o.st()), "layernorm_dropout")
            else:
                if fused_head:  # LayerNorm + the Linear(filter, 1) head + mask in one launch, the head's bias read on the device
                    yn, pred = o.act(M, filt), o.empty(M)
                    o.ck(o.lib.fs2_op_layernorm_head(o.dt, _p(c), None, _p(P[f"{p}.2.weight"]), _p(P[f"{p}.2.bias"]), _p(yn),
                                                     _p(P[f"{prefix}.linear.weight"]), _p(P[f"{prefix}.linear.bias"]), _p(mask), _p(pred),
                                                     M, filt, o.st()), "layernorm_head")
                else:
                    yn, pred = o.layernorm(c, None, P[f"{p}.2.weight"], P[f"{p}.2.bias"], M, filt)
                kd = o.site()
            if last and pd > 0:
                pred = o.empty(M)
                o.ck(o.lib.fs2_op_row_dot(o.dt, _p(yn), _p(P[f"{prefix}.linear.weight"]), _p(P[f"{prefix}.linear.bias"]), _p(mask), _p(pred),
                                          M, filt, o.st()), "row_dot")
            tape.append({"xin": y, "c": c, "cin": cin, "u": u, "kd": kd})
            y, cin = yn, filt
        return pred, {"layers": tape, "y": y, "pd": pd}

    def _predictor_bwd(self, dpred, t, prefix, nlayers, filt, k, B, S, dx_out, dw=False):
        """dpred (M) -> gradients of the predictor's parameters, and dx_out (M, H) += d/dx."""
        o, P, W, G = self.ops, self.P, self.W, self.G
        M = B * S
        # pred = y . w + b  (masked rows carry dpred = 0 already): dw = sum_m dpred[m] y[m] as a row-weighted column sum (fp32
        # weights; as a 1 x filt x M product on the 128 x 128 GEMM tile it took 51 us per head)
        dpred32 = dpred

        def head_w():
            ws = o.ws("colsum", int(o.lib.fs2_op_col_sum_ws_bytes(M, filt, 0)))
            o.ck(o.lib.fs2_op_col_sum_weighted(o._dt(t["y"]), _p(t["y"]), _p(dpred32), _p(G[f"{prefix}.linear.weight"]), _p(ws), M, filt, filt, 1,
                                               C.c_float(1.0), o.st()), "col_sum_weighted")
        head_w()
        dpred = o.to_act(dpred)
        o.col_sum(dpred, G[f"{prefix}.linear.bias"], M, 1)
        dy = o.act(M, filt)
        o.bgemm(dpred, W[f"{prefix}.linear.weight"], dy, M=M, N=filt, K=1, sAm=1, sAk=1, sBk=filt, sBn=1, ldc=filt)
        for j in reversed(range(nlayers)):
            p = f"{prefix}.layers.{j}.layers"
            lt = t["layers"][j]
            drop = (t["pd"], lt["kd"])  # the layer's Dropout behind its LayerNorm: undone inside the LayerNorm backward launch
            if dw:
                dc = self._ln_bwd(lt["c"], None, dy, f"{p}.2.weight", f"{p}.2.bias", M, filt, bias_name=f"{p}.0.module.1.bias", relu_mask=True,
                                  drop=drop)
                o.wgrad(dc, lt["u"], G[f"{p}.0.module.1.weight"], None, M, filt, lt["cin"])
                du = o.dgrad(dc, W[f"{p}.0.module.1.weight"], M, filt, lt["cin"], wt=self._wt(f"{p}.0.module.1.weight"))
                dxin = o.dwconv_bwd(du, lt["xin"], P[f"{p}.0.module.0.weight"], G[f"{p}.0.module.0.weight"], G[f"{p}.0.module.0.bias"],
                                    B, S, lt["cin"], k)
                if j == 0:
                    o.add_(dx_out, dxin)
                else:
                    dy = dxin
                continue
            dc = self._ln_bwd(lt["c"], None, dy, f"{p}.2.weight", f"{p}.2.bias", M, filt, bias_name=f"{p}.0.module.bias", relu_mask=True,
                              drop=drop)
            o.wgrad(dc, lt["xin"], G[f"{p}.0.module.weight"], None, M, filt, lt["cin"], taps=k, S=S)
            if j == 0:
                o.dgrad(dc, W[f"{p}.0.module.weight"], M, filt, lt["cin"], taps=k, S=S, out=dx_out, accumulate=True, wt=self._wt(f"{p}.0.module.weight"))
            else:
                dy = o.dgrad(dc, W[f"{p}.0.module.weight"], M, filt, lt["cin"], taps=k, S=S, wt=self._wt(f"{p}.0.module.weight"))

    def _loss_soft_dtw(self, name, pred, truth, truth_kind, mask, rows, inner):
        """get_loss with loss == "soft_dtw" (loss.py:62-81): pads zero-filled in prediction and target, the time axis cut into
        chunks of soft_dtw_chunk_size frames, soft-DTW of every (prediction chunk, target chunk) pair summed over chunks and batch;
        the gradient reaches the prediction only (truth.requires_grad = False, :63) and not its zero-filled pads.  Value and
        gradient of a chunk: fs2_op_soft_dtw_grad (csrc/softdtw.hip), pinned on the reference's vendored module."""
        from .softdtw import soft_dtw_value_and_grad
        Bm, Tm = mask.shape
        valid = (mask.view(Bm, Tm) == 0).unsqueeze(-1).to(torch.float32)
        p3 = pred.view(Bm, Tm, inner).to(torch.float32) * valid
        t3 = truth.view(Bm, Tm, -1).to(torch.float32) if truth_kind == 0 else torch.log(truth.view(Bm, Tm, 1).to(torch.float32) + 1)
        t3 = t3 * valid
        total = None
        grads = []
        for pc, tc in zip(p3.split(self.soft_dtw_chunk_size, dim=1), t3.split(self.soft_dtw_chunk_size, dim=1)):
            v, g = soft_dtw_value_and_grad(pc.contiguous(), tc.contiguous(), self.soft_dtw_gamma)
            total = v.sum() if total is None else total + v.sum()
            grads.append(g)
        dpred = (torch.cat(grads, dim=1) * valid * float(self.loss_alphas[name])).reshape(rows, inner) if inner > 1 else \
            (torch.cat(grads, dim=1) * valid * float(self.loss_alphas[name])).reshape(rows)
        return torch.stack([total, valid.sum() * inner]), dpred.contiguous()

    def _loss(self, name, pred, truth, truth_kind, mask, rows, inner, kind, want_grad=True):
        o = self.ops
        if kind == "soft_dtw":
            return self._loss_soft_dtw(name, pred, truth, truth_kind, mask, rows, inner)
        stat = o.empty(2)
        o.ck(o.lib.fs2_op_masked_loss(_p(pred), _p(truth), truth_kind, _p(mask), rows, inner, _KIND[kind], _p(self._loss_ws),
                                      _p(stat), o.st()), "masked_loss")
        dpred = None
        if want_grad:
            dpred = o.empty(rows, inner) if inner > 1 else o.empty(rows)
            o.ck(o.lib.fs2_op_masked_loss_bwd(_p(pred), _p(truth), truth_kind, _p(mask), _p(stat), _p(dpred), rows, inner,
                                              _KIND[kind], C.c_float(self.loss_alphas[name]), o.st()), "masked_loss_bwd")
        return stat, dpred

    # ---- the step ----
    def training_step(self, batch: dict) -> Dict[str, torch.Tensor]:
        """forward(targets, inference=False) + FastSpeech2Loss + backward; gradients are ADDED to the flat buffer.
        Returns the losses as 0-dim device tensors (same keys as the reference's loss dict)."""
        cfg, o, P, W, G, dev = self.cfg, self.ops, self.P, self.W, self.G, self.dev
        H = cfg.hidden
        phones = batch["phones"].to(dev, torch.int64).contiguous()
        dvec = batch["speaker"].to(dev, torch.float32).contiguous()
        dur_t = batch["duration"].to(dev, torch.int64).contiguous()
        mel_t = batch["mel"].to(dev, torch.float32).contiguous()
        B, L = phones.shape
        T = int(mel_t.shape[1])
        if tuple(dur_t.shape) != (B, L):
            raise ValueError(f"duration must be {(B, L)}, got {tuple(dur_t.shape)}")
        var_t = {}
        for v in cfg.variances:
            t = batch[f"variances_{v}"].to(dev, torch.float32).contiguous()
            if tuple(t.shape) != (B, T):
                raise ValueError(f"variances_{v} must be {(B, T)}, got {tuple(t.shape)}")
            var_t[v] = t
        pe = self.buffers["positional_encoding.pe"]
        with torch.cuda.device(dev):
            # ---------------- forward ----------------
            o.seed, o._site = (self.seed * 1000003 + self._micro) & 0xFFFFFFFFFFFFFFFF, 0
            self._micro += 1
            pe_drop = self.p_enc > 0  # PositionalEncoding(dropout=encoder_dropout) serves both stacks (fastspeech2.py:296-298)
            k_pe_enc, k_pe_dec = o.site(), o.site()
            spk = o.empty(B, H)
            o.ck(o.lib.fs2_op_spk_proj(_p(dvec), _p(P["speaker_embedding.projection.weight"]), _p(P["speaker_embedding.projection.bias"]),
                                       _p(spk), B, H, dvec.shape[1], o.st()), "spk_proj")
            x = o.act(B * L, H)
            src_mask = o.empty(B, L, dtype=torch.uint8)
            zero_spk = torch.zeros(B, H, device=dev) if pe_drop else None
            o.ck(o.lib.fs2_op_embed(o.dt, _p(phones), _p(P["phone_embedding.weight"]), _p(pe), _p(zero_spk if pe_drop else spk), _p(x),
                                    _p(src_mask), B, L, H, cfg.n_phones, o.st()), "embed")
            if pe_drop:  # x = dropout(E[phones] + pe) + spk   (model.py:53-55, fastspeech2.py:655-660)
                o.dropout(x, self.p_enc, k_pe_enc)
                o.ck(o.lib.fs2_op_bucket_embed(o.dt, _p(x), None, None, None, 0, C.c_float(1), C.c_float(0), None, _p(spk), _p(x), None,
                                               B, L, H, o.st()), "add_spk")
            enc_t = []
            for i in range(cfg.encoder_layers):
                x, t = self._layer_fwd(x, f"encoder.layers.{i}", B, L, cfg.encoder_head, cfg.encoder_conv_filter_size,
                                       cfg.encoder_kernel_sizes[i], src_mask, pd=self.p_enc)
                enc_t.append(t)
            prior_idx = {}
            for pr in cfg.priors:  # output = output + relu(Emb[bucketize(prior)]) per utterance (fastspeech2.py:687-692, model.py:160-164)
                pfx = f"prior_embeddings.{pr}"
                vals = torch.as_tensor(np.asarray(batch[f"priors_{pr}"].cpu() if isinstance(batch[f"priors_{pr}"], torch.Tensor)
                                                  else batch[f"priors_{pr}"]), dtype=torch.float32).to(dev).contiguous()
                if vals.numel() != B:
                    raise ValueError(f"priors_{pr} must hold one value per utterance")
                remb = o.empty(cfg.variance_nbins, H)
                o.ck(o.lib.fs2_op_ew(F32, 1, _p(P[f"{pfx}.embedding.weight"]), _p(P[f"{pfx}.embedding.weight"]), _p(remb), remb.numel(),
                                     C.c_float(0), C.c_float(0), o.st()), "relu")
                idx = o.empty(B * L, dtype=torch.int32)
                xn = o.act(B * L, H)
                o.ck(o.lib.fs2_op_bucket_embed_utt(o.dt, _p(x), _p(vals), _p(self.buffers[f"{pfx}.bins"]), _p(remb), cfg.variance_nbins,
                                                   _p(xn), _p(idx), B, L, H, o.st()), "prior_embed")
                prior_idx[pr] = idx.view(B, L)[:, 0].contiguous()
                x = xn
            dur_pred, dur_tape = self._predictor_fwd(x, "variance_adaptor.duration_predictor", cfg.duration_nlayers,
                                                     cfg.duration_filter_size, cfg.duration_kernel_size, B, L, src_mask,
                                                     dw=cfg.duration_depthwise_conv, pd=self.p_dur)
            forced = dur_t.to(torch.int32)
            dur, cum, totals, guard = (o.empty(B, L, dtype=torch.int32), o.empty(B, L, dtype=torch.int32),
                                       o.empty(B, dtype=torch.int32), o.empty(B, dtype=torch.int32))
            o.ck(o.lib.fs2_op_durations(_p(dur_pred), _p(src_mask), _p(forced), _p(dur), _p(cum), _p(totals), _p(guard), B, L, o.st()), "durations")
            xr = o.act(B * T, H)
            tgt_mask = o.empty(B, T, dtype=torch.uint8)
            o.ck(o.lib.fs2_op_regulate(o.dt, _p(x), _p(cum), _p(totals), _p(xr), _p(tgt_mask), B, L, T, H, o.st()), "regulate")
            var_pred, var_tape, var_idx = {}, {}, {}
            xa = xr
            nv = len(cfg.variances)
            for vi, v in enumerate(cfg.variances):
                pfx = f"variance_adaptor.encoders.{v}"
                var_pred[v], var_tape[v] = self._predictor_fwd(xa, f"{pfx}.predictor", cfg.variance_nlayers[vi], cfg.variance_filter_size,
                                                               cfg.variance_kernel_size[vi], B, T, tgt_mask, dw=cfg.variance_depthwise_conv,
                                                               pd=self.p_var[vi])
                idx = o.empty(B * T, dtype=torch.int32)
                xn = o.act(B * T, H)
                last = vi == nv - 1
                st_ = cfg.stats[v]
                o.ck(o.lib.fs2_op_bucket_embed_target(o.dt, _p(xa), _p(var_t[v]), _p(self.buffers[f"{pfx}.bins"]), _p(P[f"{pfx}.embedding.weight"]),
                                                      cfg.variance_nbins, C.c_float(st_["std"]), C.c_float(st_["mean"]),
                                                      _p(pe) if last else None, _p(spk) if (last and not pe_drop) else None, _p(xn), _p(idx),
                                                      B, T, H, o.st()), "bucket_embed_target")
                var_idx[v] = idx
                xa = xn
            if nv == 0:
                xn = o.act(B * T, H)
                o.ck(o.lib.fs2_op_bucket_embed(o.dt, _p(xa), None, None, None, 0, C.c_float(1), C.c_float(0), _p(pe),
                                               None if pe_drop else _p(spk), _p(xn), None, B, T, H, o.st()), "pe_spk")
                xa = xn
            y = xa
            if pe_drop:  # y = dropout(x + pe) + spk   (fastspeech2.py:705-718)
                o.dropout(y, self.p_enc, k_pe_dec)
                o.ck(o.lib.fs2_op_bucket_embed(o.dt, _p(y), None, None, None, 0, C.c_float(1), C.c_float(0), None, _p(spk), _p(y), None,
                                               B, T, H, o.st()), "add_spk")
            dec_t = []
            for i in range(cfg.decoder_layers):
                y, t = self._layer_fwd(y, f"decoder.layers.{i}", B, T, cfg.decoder_head, cfg.decoder_conv_filter_size,
                                       cfg.decoder_kernel_sizes[i], tgt_mask, pd=self.p_dec)
                dec_t.append(t)
            mel = o.gemm(y, W["linear.weight"], P["linear.bias"], B * T, cfg.n_mels, H, out_f32=True)
            # ---------------- losses + their gradients (loss.py:83-213) ----------------
            losses = {}
            dvar = {}
            for v, kind in zip(cfg.variances, self.variance_losses):
                stat, dvar[v] = self._loss(v, var_pred[v], var_t[v], 0, tgt_mask, B * T, 1, kind)
                losses[v] = stat[0]
            stat, dmel = self._loss("mel", mel, mel_t, 0, tgt_mask, B * T, cfg.n_mels, self.mel_loss)
            losses["mel"] = stat[0]
            stat, ddur = self._loss("duration", dur_pred, dur_t, 1, src_mask, B * L, 1, self.duration_loss)
            losses["duration"] = stat[0]
            losses["total"] = sum(v * self.loss_alphas[k] for k, v in losses.items())
            # ---------------- backward ----------------
            dmel = o.to_act(dmel)
            o.wgrad(dmel, y, G["linear.weight"], G["linear.bias"], B * T, cfg.n_mels, H)
            dy = o.dgrad(dmel, W["linear.weight"], B * T, cfg.n_mels, H)
            for i in reversed(range(cfg.decoder_layers)):
                dy = self._layer_bwd(dy, dec_t[i], f"decoder.layers.{i}", B, T, cfg.decoder_head, cfg.decoder_conv_filter_size,
                                     cfg.decoder_kernel_sizes[i])
            dspk = torch.zeros(B, H, device=dev)
            o.col_sum(dy, dspk, B * T, H, seg=T)  # decoder input = dropout(adaptor out + pe) + spk (fastspeech2.py:705-718)
            dx = o.dropout(dy, self.p_enc, k_pe_dec)
            for vi in reversed(range(nv)):
                v = cfg.variances[vi]
                pfx = f"variance_adaptor.encoders.{v}"
                o.on_side((dx,), lambda: o.scatter_rows(dx, var_idx[v], None, G[f"{pfx}.embedding.weight"], B * T, H, cfg.variance_nbins, -1))
                self._predictor_bwd(dvar[v], var_tape[v], f"{pfx}.predictor", cfg.variance_nlayers[vi], cfg.variance_filter_size,
                                    cfg.variance_kernel_size[vi], B, T, dx, dw=cfg.variance_depthwise_conv)
            dxe = o.act(B * L, H)
            o.ck(o.lib.fs2_op_regulate_bwd(o.dt, _p(dx), _p(cum), _p(dxe), B, L, T, H, o.st()), "regulate_bwd")
            self._predictor_bwd(ddur, dur_tape, "variance_adaptor.duration_predictor", cfg.duration_nlayers, cfg.duration_filter_size,
                                cfg.duration_kernel_size, B, L, dxe, dw=cfg.duration_depthwise_conv)
            if cfg.priors:
                seg = o.empty(B, H)
                o.col_sum(dxe, seg, B * L, H, seg=L, accumulate=False)
                for pr in cfg.priors:
                    pfx = f"prior_embeddings.{pr}"
                    tmp = torch.zeros(cfg.variance_nbins, H, device=dev)
                    o.scatter_rows(seg, prior_idx[pr], None, tmp, B, H, cfg.variance_nbins, -1)
                    o.relu_bwd(tmp, P[f"{pfx}.embedding.weight"])
                    o.add_(G[f"{pfx}.embedding.weight"], tmp)
            for i in reversed(range(cfg.encoder_layers)):
                dxe = self._layer_bwd(dxe, enc_t[i], f"encoder.layers.{i}", B, L, cfg.encoder_head, cfg.encoder_conv_filter_size,
                                      cfg.encoder_kernel_sizes[i])
            o.col_sum(dxe, dspk, B * L, H, seg=L)
            o.dropout(dxe, self.p_enc, k_pe_enc)
            o.on_side((dxe,), lambda: o.scatter_rows(dxe, None, phones, G["phone_embedding.weight"], B * L, H, cfg.n_phones, 0))
            o.relu_bwd(dspk, spk)  # spk = relu(W dvec + b), model.py:137-143
            o.wgrad(dspk, dvec, G["speaker_embedding.projection.weight"], G["speaker_embedding.projection.bias"], B, H, dvec.shape[1])
        o.join_side()  # the weight gradients of this micro-step are in the flat buffer before anything reads it
        self._accum += 1
        self.last = {"mel": mel.view(B, T, cfg.n_mels), "duration_prediction": dur_pred.view(B, L), "tgt_mask": tgt_mask.bool(),
                     "src_mask": src_mask.bool(), **{f"variances_{v}": var_pred[v].view(B, T) for v in cfg.variances}}
        return losses

    def optimizer_step(self, group=None):
        """clip_grad_norm_(gradient_clip_val) on the mean of the accumulated micro-batch gradients, AdamW, NoamLR.  With an
        initialised torch.distributed process group (one process per GPU, "nccl" = RCCL) the flat gradient buffer is summed
        over the ranks first and the 1 / world factor joins the gradient scale - DDP's gradient averaging."""
        if self._accum == 0:
            raise RuntimeError("optimizer_step before any training_step")
        o = self.ops
        lr = self.current_lr()
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            from .dist import all_reduce_gradients
            world = torch.distributed.get_world_size(group)
            if world > 1:
                all_reduce_gradients(self.flat_g, group)
        with torch.cuda.device(self.dev):
            nsq = None
            if self.gradient_clip_val is not None:
                o.ck(o.lib.fs2_op_sum_sq(_p(self.flat_g), self.n_flat, _p(self._nsq_ws), _p(self._nsq), o.st()), "sum_sq")
                nsq = self._nsq
            shadow = self.flat_w if o.dt != F32 and os.environ.get("FS2_TRAIN_ADAMW_SHADOW", "1") != "0" else None  # (A/B switch)
            o.ck(o.lib.fs2_op_adamw_shadow(_p(self.flat_p), _p(self.flat_g), _p(self.flat_m), _p(self.flat_v), _p(shadow), self.n_flat,
                                           C.c_float(lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                           C.c_float(self.weight_decay), self.steps + 1, _p(nsq), C.c_float(self.gradient_clip_val or 0.0),
                                           C.c_float(1.0 / (self._accum * world)), o.st()), "adamw")
        self.steps += 1
        self.zero_grad()
        self._refresh_shadow(converted=shadow is not None)
        return lr
