"""Soft-DTW value on the GPU: drop-in for ``litfass.third_party.softdtw.SoftDTW`` as ``validation_epoch_end`` uses it
(fastspeech2.py:1149-1156: ``SoftDTW(normalize=True, gamma=...)(pred_mel, true_mel)``), forward value only.
All arithmetic in libfs2_hip.so (csrc/softdtw.hip) through ``fs2_op_soft_dtw``; no CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def soft_dtw_values(x: torch.Tensor, y: torch.Tensor, gamma: float) -> torch.Tensor:
    """(B, N, D), (B, M, D) fp32 on one GPU -> (B,) fp32: R[N, M] of every pair."""
    if x.device.type != "cuda" or x.device != y.device:
        raise RuntimeError("soft-DTW runs on the GPU its inputs live on (no CPU path)")
    x, y = x.to(torch.float32).contiguous(), y.to(torch.float32).contiguous()
    if x.dim() != 3 or y.dim() != 3 or x.shape[0] != y.shape[0] or x.shape[2] != y.shape[2]:
        raise ValueError(f"expected (B, N, D) and (B, M, D), got {tuple(x.shape)} and {tuple(y.shape)}")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = _lib.load().fs2_op_soft_dtw(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.shape[0], x.shape[1], y.shape[1],
                                         x.shape[2], float(gamma), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(st, None, "fs2_op_soft_dtw")
    return out


def soft_dtw_value_and_grad(x: torch.Tensor, y: torch.Tensor, gamma: float):
    """(B, N, D), (B, M, D) fp32 on one GPU -> ((B,) values, (B, N, D) d value / d x): the backward of the reference's
    ``_SoftDTW`` (third_party/softdtw/__init__.py:27-77) composed with ``calc_distance_matrix`` (:85-92), in one launch."""
    if x.device.type != "cuda" or x.device != y.device:
        raise RuntimeError("soft-DTW runs on the GPU its inputs live on (no CPU path)")
    x, y = x.to(torch.float32).contiguous(), y.to(torch.float32).contiguous()
    if x.dim() != 3 or y.dim() != 3 or x.shape[0] != y.shape[0] or x.shape[2] != y.shape[2]:
        raise ValueError(f"expected (B, N, D) and (B, M, D), got {tuple(x.shape)} and {tuple(y.shape)}")
    B, N, D = x.shape
    M = y.shape[1]
    lib = _lib.load()
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    nbytes = lib.fs2_op_soft_dtw_grad_scratch_bytes(B, N, M)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.fs2_op_soft_dtw_grad(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, N, M, D, float(gamma),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(gx.data_ptr()), C.c_void_p(scratch.data_ptr()), nbytes,
                                      C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(st, None, "fs2_op_soft_dtw_grad")
    return out, gx


class SoftDTW:
    """``SoftDTW(gamma=1.0, normalize=False)(x, y)`` - same arguments, same result shape (a 0-dim tensor for unbatched
    (N, D) inputs) as the reference's module (third_party/softdtw/__init__.py:101-139)."""

    def __init__(self, gamma: float = 1.0, normalize: bool = False, device="cuda:0"):
        self.gamma, self.normalize, self.device = gamma, normalize, torch.device(device)

    def __call__(self, x, y):
        return self.forward(x, y)

    def forward(self, x, y):
        x, y = torch.as_tensor(x), torch.as_tensor(y)
        assert x.dim() == y.dim()
        squeeze = x.dim() < 3
        if squeeze:
            x, y = x.unsqueeze(0), y.unsqueeze(0)
        dev = x.device if x.is_cuda else self.device
        x, y = x.to(dev), y.to(dev)
        out = soft_dtw_values(x, y, self.gamma)
        if self.normalize:  # out_xy - 1/2 (out_xx + out_yy), :126-134
            out = out - 0.5 * (soft_dtw_values(x, x, self.gamma) + soft_dtw_values(y, y, self.gamma))
        return out.squeeze(0) if squeeze else out
