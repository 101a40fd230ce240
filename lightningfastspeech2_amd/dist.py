"""Data-parallel sharding of utterance batches over the GPUs of one node.

The path shards naturally: utterances are independent (no cross-utterance math anywhere in
``FastSpeech2.forward`` — SURVEY.md §8e), weights are replicated, every rank runs the full forward
on its own contiguous shard *as its own padded batch* (what the reference does per rank under
Lightning DDP, or per ``--batch_size`` chunk in generate.py:186-195).  The only exchange is the
final mel tensors: one all-gather of the per-rank frame counts (so ranks with different T_r can be
padded to a common T) and one all-gather of the (B_r, T, n_mels) fp32 mels — RCCL over xGMI when
the process group is "nccl", gloo in the CPU tests.  No collective touches the forward itself.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.distributed as dist


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal split of B utterances: rank r gets [lo, hi)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: Dict[str, torch.Tensor], world: int, rank: int) -> Dict[str, torch.Tensor]:
    B = batch["phones"].shape[0]
    lo, hi = shard_bounds(B, world, rank)
    out = {}
    for k, v in batch.items():
        out[k] = v[lo:hi] if hasattr(v, "shape") and len(v.shape) > 0 and v.shape[0] == B else v
    # each shard is its own padded batch: trim the phone axis to the shard's longest utterance
    ph = out["phones"]
    if ph.numel():
        lens = (ph != 0).sum(dim=1)
        # pads are a suffix in the collate format (datasets.py:878-880)
        Lr = max(int(lens.max()), 1)
        out["phones"] = ph[:, :Lr].contiguous()
    return out


class MelGather:
    """An all-gather of the final mels in flight (``gather_mels_async``); ``wait()`` -> (mel_all, frames).
    The exchange runs on the collective library's own stream, so a caller that keeps launching the next
    forward overlaps it with compute; nothing here blocks the host except the tiny shape exchange."""

    def __init__(self, works, all_mel, all_fr, Bs, B_max):
        self._works, self._all_mel, self._all_fr, self._Bs, self._B_max = works, all_mel, all_fr, Bs, B_max

    def wait(self):
        for w in self._works:
            w.wait()   # nccl: the current stream waits for the collective's stream; gloo: host wait
        self._works = []
        if all(b == self._B_max for b in self._Bs):
            return self._all_mel, self._all_fr
        dev = self._all_mel.device
        keep = torch.cat([torch.arange(r * self._B_max, r * self._B_max + b, device=dev) for r, b in enumerate(self._Bs)])
        return self._all_mel[keep], self._all_fr[keep]


def gather_mels_async(mel: torch.Tensor, tgt_mask: torch.Tensor, group=None) -> MelGather:
    """Start the all-gather of every rank's final mels.

    mel (B_r, T_r, n_mels) fp32 and tgt_mask (B_r, T_r) bool (True = pad) of this rank; the result of
    ``wait()`` is ``(mel_all (B, T_max, n_mels), frames (B,) int64)`` on every rank, utterances in global
    batch order, rows beyond an utterance's frame count zeroed.  Two steps: the per-rank (B_r, T_r)
    pairs (16 bytes, read back on the host to size the buffers), then the padded mels and the frame
    counts as asynchronous collectives.
    """
    world = dist.get_world_size(group)
    dev = mel.device
    B_r, T_r, n_mels = mel.shape
    meta = torch.tensor([B_r, T_r], dtype=torch.int64, device=dev)
    metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.cpu().view(world, 2)
    Bs = [int(b) for b in metas[:, 0]]
    T_max, B_max = int(metas[:, 1].max()), max(Bs)
    frames_r = (~tgt_mask).sum(dim=1).to(torch.int64)
    if B_r == B_max and T_r == T_max:
        buf = mel * (~tgt_mask).unsqueeze(-1)
        fr = frames_r
    else:
        buf = torch.zeros(B_max, T_max, n_mels, dtype=mel.dtype, device=dev)
        buf[:B_r, :T_r] = mel * (~tgt_mask).unsqueeze(-1)
        fr = torch.zeros(B_max, dtype=torch.int64, device=dev)
        fr[:B_r] = frames_r
    all_mel = torch.empty(world * B_max, T_max, n_mels, dtype=mel.dtype, device=dev)
    all_fr = torch.empty(world * B_max, dtype=torch.int64, device=dev)
    works = [dist.all_gather_into_tensor(all_mel, buf, group=group, async_op=True),
             dist.all_gather_into_tensor(all_fr, fr, group=group, async_op=True)]
    return MelGather(works, all_mel, all_fr, Bs, B_max)


def gather_mels(mel: torch.Tensor, tgt_mask: torch.Tensor, group=None):
    """Blocking form of :func:`gather_mels_async`."""
    return gather_mels_async(mel, tgt_mask, group=group).wait()


def forward_sharded(forward_fn: Callable[[Dict[str, torch.Tensor]], Dict[str, torch.Tensor]],
                    batch: Dict[str, torch.Tensor], group=None):
    """Run ``forward_fn`` (e.g. ``lambda b: model(b, inference=True)``) on this rank's shard of
    ``batch`` and gather every rank's mels.  Returns (mel_all, frames, local_result)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local = forward_fn(shard_batch(batch, world, rank))
    mel_all, frames = gather_mels(local["mel"], local["tgt_mask"], group=group)
    return mel_all, frames, local
