"""Data-parallel sharding of utterance batches over the GPUs of one node.

The path shards naturally: utterances are independent (no cross-utterance math anywhere in
``FastSpeech2.forward`` — SURVEY.md §8e), weights are replicated, every rank runs the full forward
on its own contiguous shard.  Two padding modes:

* per-shard (default): a shard is *its own padded batch* — what the reference does per rank under
  Lightning DDP, or per ``--batch_size`` chunk in generate.py:186-195.  Because the reference's convs
  are unmasked, pad rows leak into valid frames, so the result equals the reference run
  on that shard alone, not the whole-batch run.
* global pad (``global_pad=True``): every shard keeps the whole batch's phone length and pads its frames
  to the whole batch's T — one 8-byte all-reduce(MAX) between the two phases of the forward — and the
  gathered result equals the whole-batch run exactly.

The only bulk exchange is the final mel tensors: an all-gather of the (B_r, T, n_mels) fp32 mels plus
the per-utterance frame counts — RCCL over xGMI when the process group is "nccl", gloo in the CPU
tests.  No collective touches the forward's arithmetic.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

PHONE_LEVEL_KEYS = ("phones", "duration")  # (B, L) entries of the collate format (datasets.py:852-882)


def _is_frame_level(key: str) -> bool:
    """(B, T, ...) entries of the collate format: the mel and the frame-level variance targets (datasets.py:866-877)."""
    return key == "mel" or key.startswith("variances_")


def collective_device(group=None, like=None) -> torch.device:
    """Where a small control tensor of a collective must live: the process group's backend decides (nccl = RCCL wants
    device memory on every rank - also on ranks that hold no utterance and therefore no device tensor to copy from)."""
    backend = str(dist.get_backend(group)).lower()
    if "nccl" in backend:
        # a rank whose model and batch live on cuda:{local_rank} without torch.cuda.set_device() must not build its control
        # tensors on cuda:0: take the device of the tensor the rank already holds, the current device only without one
        if isinstance(like, torch.Tensor) and like.is_cuda:
            return like.device
        return torch.device("cuda", torch.cuda.current_device())
    return like.device if isinstance(like, torch.Tensor) else torch.device("cpu")


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal split of B utterances: rank r gets [lo, hi)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: Dict[str, torch.Tensor], world: int, rank: int, *, trim: bool = True,
                phone_level_keys: Sequence[str] = PHONE_LEVEL_KEYS) -> Dict[str, torch.Tensor]:
    """This rank's contiguous rows of every per-utterance entry.  ``trim``: cut the phone axis of
    EVERY per-phone (B, L) entry (phones, duration, ...) to the shard's longest utterance, so that the
    shard is its own padded batch and teacher-forced durations stay aligned with the phones; when the batch carries
    target durations, the frame axis of the frame-level entries (``mel``, ``variances_*``) is cut to the shard's longest
    utterance in frames (max over its rows of sum(duration)) as well - a training shard is then padded to ITS T, which is
    what a DataLoader under DDP would have collated for that rank."""
    B = batch["phones"].shape[0]
    lo, hi = shard_bounds(B, world, rank)
    out = {}
    for k, v in batch.items():
        out[k] = v[lo:hi] if hasattr(v, "shape") and len(v.shape) > 0 and v.shape[0] == B else v
    ph = out["phones"]
    if trim and ph.numel():
        lens = (ph != 0).sum(dim=1)
        # pads are a suffix in the collate format (datasets.py:878-880)
        Lr = max(int(lens.max()), 1)
        L = ph.shape[1]
        for k in phone_level_keys:
            v = out.get(k)
            if v is None or not hasattr(v, "shape") or len(v.shape) < 2 or v.shape[1] != L:
                continue
            if k != "phones" and bool((torch.as_tensor(v)[:, Lr:] != 0).any()):
                raise ValueError(f"batch[{k!r}] has non-zero entries beyond the shard's longest utterance")
            out[k] = v[:, :Lr].contiguous()
        dur = out.get("duration")
        mel_t = out.get("mel")
        if dur is not None and hasattr(dur, "shape") and len(dur.shape) == 2:
            Tr = max(int(torch.as_tensor(dur).sum(dim=1).max()), 1)
            # the frame axis is the mel's: only entries that share it are frame-level (a phone-level (B, L) "variances_*"
            # entry with L > Tr - zero-length durations - is left alone); without a mel, the longest such entry names it
            T_b = mel_t.shape[1] if hasattr(mel_t, "shape") and len(mel_t.shape) >= 2 else max(
                [v.shape[1] for k, v in out.items() if _is_frame_level(k) and hasattr(v, "shape") and len(v.shape) >= 2
                 and v.shape[1] != L] or [0])
            for k, v in list(out.items()):
                if not (_is_frame_level(k) and hasattr(v, "shape") and len(v.shape) >= 2 and v.shape[1] == T_b and T_b > Tr):
                    continue
                if bool((torch.as_tensor(v)[:, Tr:] != 0).any()):
                    raise ValueError(f"batch[{k!r}] has non-zero frames beyond the shard's sum(duration) = {Tr}")
                out[k] = v[:, :Tr].contiguous()
    return out


class MelGather:
    """An all-gather of the final mels in flight (``gather_mels_async``); ``wait()`` -> (mel_all, frames).
    The exchange runs on the collective library's own stream, so a caller that keeps launching the next
    forward overlaps it with compute."""

    def __init__(self, works, all_mel, all_fr, Bs, B_max, keep_alive=None):
        self._works, self._all_mel, self._all_fr, self._Bs, self._B_max = works, all_mel, all_fr, Bs, B_max
        self._keep = keep_alive  # send buffers of a root-only gather: alive until the collective has finished

    def wait(self):
        for w in self._works:
            w.wait()   # nccl: the current stream waits for the collective's stream; gloo: host wait
        self._works = []
        self._keep = None
        if self._all_mel is None:  # root-only gather (dst=), and this rank is not the root
            return None, None
        if self._Bs is None or all(b == self._B_max for b in self._Bs):
            return self._all_mel, self._all_fr
        dev = self._all_mel.device
        keep = torch.cat([torch.arange(r * self._B_max, r * self._B_max + b, device=dev) for r, b in enumerate(self._Bs)])
        return self._all_mel[keep], self._all_fr[keep]


def gather_mels_async(mel: torch.Tensor, tgt_mask: torch.Tensor, group=None, *,
                      shapes: Optional[Tuple[Sequence[int], int]] = None, zeroed: bool = False,
                      dst: Optional[int] = None) -> MelGather:
    """Start the gather of every rank's final mels: onto every rank (``dst=None``: an all-gather) or onto rank ``dst`` only.

    ``dst`` (a global rank, as ``torch.distributed.gather`` takes it): the north star's literal "RCCL gather of the final mel
    tensors" - only the root allocates the (B, T_max, n_mels) result and receives (world - 1) x the shard; every other rank
    sends its shard once and ``wait()`` hands it ``(None, None)``.  Over xGMI that is one message per link into the root
    instead of every rank receiving from every other (at C4: 7 x 15.7 MB into one GPU instead of into each of eight); use it
    when one process consumes the batch (a vocoder / writer on rank 0), the all-gather when every rank needs it.

    mel (B_r, T_r, n_mels) fp32 and tgt_mask (B_r, T_r) bool (True = pad) of this rank; the result of
    ``wait()`` is ``(mel_all (B, T_max, n_mels), frames (B,) int64)`` on every rank, utterances in global
    batch order, rows beyond an utterance's frame count zero.

    ``shapes = (Bs, T_max)``: every rank's utterance count and the common frame capacity are already known
    to all ranks (a fixed batch split with global-pad mode or fixed-length inputs; ``forward_sharded``
    passes it in global-pad mode) — then nothing is exchanged or read back on the host before the
    collectives are queued.  Otherwise the per-rank (B_r, T_r) pairs are all-gathered and read back first
    (16 bytes, one host sync).
    ``zeroed``: the pad rows of ``mel`` are already zero (``Engine.set_zero_pad_mel``, fused into the mel
    GEMM's store) — skips the masking pass over the mel.
    """
    world = dist.get_world_size(group)
    dev = mel.device
    B_r, T_r, n_mels = mel.shape
    if shapes is None:
        meta = torch.tensor([B_r, T_r], dtype=torch.int64, device=dev)
        metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta, group=group)
        metas = metas.cpu().view(world, 2)
        Bs = [int(b) for b in metas[:, 0]]
        T_max = int(metas[:, 1].max())
    else:
        Bs, T_max = [int(b) for b in shapes[0]], int(shapes[1])
        if len(Bs) != world or B_r != Bs[dist.get_rank(group)] or T_r > T_max:
            raise ValueError(f"shapes={shapes} do not describe this rank's mel {tuple(mel.shape)}")
    B_max = max(Bs)
    frames_r = (~tgt_mask).sum(dim=1).to(torch.int64)
    src = mel if zeroed else mel * (~tgt_mask).unsqueeze(-1)
    if B_r == B_max and T_r == T_max:
        buf, fr = src.contiguous(), frames_r
    else:
        buf = torch.zeros(B_max, T_max, n_mels, dtype=mel.dtype, device=dev)
        buf[:B_r, :T_r] = src
        fr = torch.zeros(B_max, dtype=torch.int64, device=dev)
        fr[:B_r] = frames_r
    if dst is not None:
        if not 0 <= int(dst) < dist.get_world_size():
            raise ValueError(f"dst={dst} is not a rank of this job")
        if dist.get_rank() == int(dst):
            all_mel = torch.empty(world * B_max, T_max, n_mels, dtype=mel.dtype, device=dev)
            all_fr = torch.empty(world * B_max, dtype=torch.int64, device=dev)
            # the root's receive buffers are the rows of the result itself: contiguous (B_max, T_max, n_mels) views, no copy
            mel_list = list(all_mel.view(world, B_max, T_max, n_mels).unbind(0))
            fr_list = list(all_fr.view(world, B_max).unbind(0))
        else:
            all_mel = all_fr = mel_list = fr_list = None
        buf, fr = buf.contiguous(), fr.contiguous()
        works = [dist.gather(buf, mel_list, dst=int(dst), group=group, async_op=True),
                 dist.gather(fr, fr_list, dst=int(dst), group=group, async_op=True)]
        return MelGather(works, all_mel, all_fr, Bs, B_max, keep_alive=(buf, fr))
    all_mel = torch.empty(world * B_max, T_max, n_mels, dtype=mel.dtype, device=dev)
    all_fr = torch.empty(world * B_max, dtype=torch.int64, device=dev)
    works = [dist.all_gather_into_tensor(all_mel, buf, group=group, async_op=True),
             dist.all_gather_into_tensor(all_fr, fr, group=group, async_op=True)]
    return MelGather(works, all_mel, all_fr, Bs, B_max)


def gather_mels(mel: torch.Tensor, tgt_mask: torch.Tensor, group=None, **kw):
    """Blocking form of :func:`gather_mels_async`."""
    return gather_mels_async(mel, tgt_mask, group=group, **kw).wait()


def global_frames(T_local: int, device, group=None) -> int:
    """max over the ranks of the local frame count: the T the whole batch would be padded to."""
    t = torch.tensor([int(T_local)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def forward_sharded(forward_fn: Callable[..., Dict[str, torch.Tensor]], batch: Dict[str, torch.Tensor], group=None, *,
                    global_pad: bool = False, n_mels: Optional[int] = None, zeroed: bool = False,
                    device: Optional[torch.device] = None, dst: Optional[int] = None):
    """Run ``forward_fn`` on this rank's shard of ``batch`` and gather every rank's mels.

    per-shard mode: ``forward_fn(shard)`` (e.g. ``lambda b: model(b, inference=True)``).
    global-pad mode: ``forward_fn(shard, frames_hook)`` — the hook must be handed to the model's forward
    (``model.forward(b, True, frames_hook=hook)``); it all-reduces the frame count between the two phases.
    Ranks whose shard is empty (global batch smaller than the world) skip the forward and contribute an
    empty mel.  ``device``: where the control tensors of the collectives (and an empty shard's mel) live; default =
    what the group's backend needs (``collective_device``: the current GPU under nccl, also for a host-side batch).
    ``dst``: gather onto that rank only (``gather_mels_async``); the other ranks get ``(None, None, local)``.
    Returns (mel_all, frames, local_result or None)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = batch["phones"].shape[0]
    Bs = [hi - lo for lo, hi in (shard_bounds(B, world, r) for r in range(world))]
    local_batch = shard_batch(batch, world, rank, trim=not global_pad)
    sp_ = batch.get("speaker")
    dev = torch.device(device) if device is not None else collective_device(group, sp_ if isinstance(sp_, torch.Tensor) else None)
    t_glob = {}

    def hook(T_local):
        t_glob["T"] = global_frames(T_local, dev, group)
        return t_glob["T"]

    if Bs[rank] == 0:
        local = None
        if n_mels is None:
            raise ValueError("a world larger than the batch leaves empty shards: pass n_mels so that they can join the gather")
        if global_pad:
            hook(0)  # the ranks that do have utterances are waiting in the all-reduce
        mel = torch.zeros(0, 0, n_mels, dtype=torch.float32, device=dev)
        mask = torch.zeros(0, 0, dtype=torch.bool, device=dev)
    else:
        if global_pad:
            local = forward_fn(local_batch, hook)
        else:
            local = forward_fn(local_batch)
        mel, mask = local["mel"], local["tgt_mask"]
        if mel.device != mask.device:
            mask = mask.to(mel.device)
        if device is None and mel.is_cuda:
            dev = mel.device   # the gather's tensors follow the local mel
        if "nccl" in str(dist.get_backend(group)).lower() and mel.device != dev:
            raise ValueError(f"local mel on {mel.device}, collective control tensors on {dev}: RCCL needs one device per rank")
    shapes = (Bs, t_glob["T"]) if global_pad else None
    mel_all, frames = gather_mels(mel, mask, group=group, shapes=shapes, zeroed=zeroed, dst=dst)
    return mel_all, frames, local


# ---- training: gradient averaging over the data-parallel ranks (what Lightning's DDP strategy does, scripts/train.sh "--strategy ddp") ----
def all_reduce_gradients(flat_g: torch.Tensor, group=None, *, bucket_bytes: int = 256 << 20, async_op: bool = False):
    """SUM the flat fp32 gradient buffer of ``training.Trainer`` over the ranks, in place.

    All parameters' gradients live in one contiguous buffer, so this is a handful of large all-reduces (``bucket_bytes`` each;
    xGMI rings are per-link bound, big messages keep every link busy) instead of one per tensor.  The 1 / world factor is folded
    into the optimizer's gradient scale by the caller (``Trainer.optimizer_step``).  ``async_op``: returns the work handles;
    the collectives run on the collective library's stream (RCCL) and overlap whatever the caller launches next.
    """
    n = flat_g.numel()
    per = max(1, bucket_bytes // flat_g.element_size())
    works = []
    for lo in range(0, n, per):
        w = dist.all_reduce(flat_g[lo:min(n, lo + per)], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def rank_report(*, ms_per_step: float, ms_without_gather: float, ms_with_gather: float, frames_per_step: int, gather_bytes: int,
                local_rank: int, device_index: int, device, group=None) -> dict:
    """The `dist` object of a multi-rank bench line (bench.py): every rank's own numbers all-gathered over the SAME process
    group the mel gather uses, so that a line claiming N ranks can be checked - which ranks the collective library really
    connected, which device each drove, every rank's step time with and without the gather, the frames and the bytes a rank
    contributes.  Pure bookkeeping (one small all-gather); returns the same dict on every rank.

    Internal consistency a reader (and tests/test_dist_cpu.py) can hold it to: ``ranks_seen == list(range(world_size))``,
    ``sum(per_rank_frames_per_step)`` = the frames the headline value was computed from, ``gather_bytes_total_per_step ==
    world_size * gather_bytes_per_rank``, ``gather_ms_exposed == max(with) - max(without)``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([float(rank), float(local_rank), float(ms_per_step), float(ms_without_gather), float(frames_per_step),
                         float(device_index), float(ms_with_gather), float(gather_bytes)], dtype=torch.float64, device=device)
    allr = torch.empty(world * mine.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(allr, mine, group=group)
    allr = allr.cpu().view(world, -1)
    return {
        "backend": str(dist.get_backend(group)), "world_size": world, "ranks_seen": [int(r) for r in allr[:, 0]],
        "local_ranks": [int(r) for r in allr[:, 1]], "devices": [int(r) for r in allr[:, 5]],
        "per_rank_ms_per_step": [round(float(v), 4) for v in allr[:, 2]],
        "per_rank_ms_per_step_without_gather": [round(float(v), 4) for v in allr[:, 3]],
        "per_rank_frames_per_step": [int(v) for v in allr[:, 4]],
        "gather_bytes_per_rank": int(allr[:, 7].max()), "gather_bytes_total_per_step": int(allr[:, 7].sum()),
        "per_rank_gather_bytes": [int(v) for v in allr[:, 7]],
        "per_rank_ms_per_step_with_gather_ab": [round(float(v), 4) for v in allr[:, 6]],
        "gather_ms_exposed": round(float(allr[:, 6].max() - allr[:, 3].max()), 4),
    }
