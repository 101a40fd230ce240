"""world_size-2 gloo test of the data-parallel path (runs on CPU): shard -> per-rank forward ->
all-gather of mels only.  The per-rank forward here is the CPU oracle standing in for the HIP
engine (tests may use the oracle); what is under test is the sharding/gather logic of
lightningfastspeech2_amd/dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lightningfastspeech2_amd.config import Fs2Config
from lightningfastspeech2_amd.dist import forward_sharded, shard_batch, shard_bounds
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict


def _cfg():
    return Fs2Config(n_phones=30, encoder_hidden=32, decoder_hidden=32, encoder_head=2, decoder_head=2,
                     encoder_layers=1, decoder_layers=1, encoder_kernel_sizes=[3], decoder_kernel_sizes=[3],
                     encoder_conv_filter_size=64, decoder_conv_filter_size=64, encoder_depthwise_conv=False,
                     decoder_depthwise_conv=True, variance_filter_size=32, variance_nlayers=[1, 1, 1],
                     duration_filter_size=32, variance_nbins=8, n_mels=4)


def test_shard_bounds_cover_batch():
    for B in (1, 5, 8, 33):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(B, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_trims_to_its_own_padding():
    ph = torch.tensor([[1, 2, 3, 4], [5, 0, 0, 0], [6, 7, 0, 0]])
    sh = shard_batch({"phones": ph, "speaker": torch.zeros(3, 256)}, 2, 1)
    assert sh["phones"].tolist() == [[6, 7]] and sh["speaker"].shape == (1, 256)


def _batch(B):
    cfg = _cfg()
    sd = synth_state_dict(cfg, 5, randomize_norm=True, duration_bias=1.0)
    inp = synth_inputs(cfg, B, 9, seed=3, lengths=[9, 4, 7, 2, 6][:B])
    return cfg, sd, {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}


def _worker(rank, world, port, B, q, mode="shard"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import oracle_cpu
    cfg, sd, batch = _batch(B)
    if mode == "global":
        fwd = lambda b, hook: oracle_cpu.forward(sd, cfg, b["phones"], b["speaker"], frames_hook=hook)
        mel_all, frames, _ = forward_sharded(fwd, batch, global_pad=True, n_mels=cfg.n_mels)
    elif mode == "teacher":  # teacher-forced durations travel with their phones (shard_batch trims both)
        ref = oracle_cpu.forward(sd, cfg, batch["phones"], batch["speaker"])
        batch["duration"] = ref["duration_rounded"]

        def fwd(b):
            assert b["duration"].shape == b["phones"].shape
            return oracle_cpu.forward(sd, cfg, b["phones"], b["speaker"], force_durations=b["duration"])
        mel_all, frames, _ = forward_sharded(fwd, batch, n_mels=cfg.n_mels)
    elif mode.startswith("root"):  # root-only gather (dst=): rank 1 receives, the others only send and get (None, None)
        fwd = lambda b: oracle_cpu.forward(sd, cfg, b["phones"], b["speaker"])
        root = int(mode[4:])
        mel_all, frames, _ = forward_sharded(fwd, batch, n_mels=cfg.n_mels, dst=root)
        assert (mel_all is None) == (rank != root) and (frames is None) == (rank != root)
        if rank == root:
            q.put((mel_all.numpy(), frames.numpy()))
        dist.barrier()
        dist.destroy_process_group()
        return
    else:
        fwd = lambda b: oracle_cpu.forward(sd, cfg, b["phones"], b["speaker"])
        mel_all, frames, _ = forward_sharded(fwd, batch, n_mels=cfg.n_mels)
    if rank == 0:
        q.put((mel_all.numpy(), frames.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, B, mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_shard_trims_frame_level_targets_to_its_own_frames():
    """a training shard carries mel / variances_* padded to the WHOLE batch's T; with trim they are cut to the
    shard's longest utterance in frames (max over its rows of sum(duration)) - what that rank's DataLoader would have collated."""
    ph = torch.tensor([[3, 4, 5, 6], [7, 8, 0, 0], [9, 0, 0, 0]])
    dur = torch.tensor([[2, 3, 1, 4], [5, 1, 0, 0], [2, 0, 0, 0]])
    T = 10
    valid = (torch.arange(T)[None, :] < dur.sum(1)[:, None])   # the collate format zero-pads beyond an utterance's frames
    batch = {"phones": ph, "duration": dur, "speaker": torch.zeros(3, 256),
             "mel": (1 + torch.arange(3 * T * 2.0)).reshape(3, T, 2) * valid[..., None],
             "variances_pitch": (1 + torch.arange(3 * T * 1.0)).reshape(3, T) * valid, "priors_energy": torch.tensor([0.1, 0.2, 0.3])}
    sh = shard_batch(batch, 2, 1)   # rank 1: the last utterance alone (2 frames, 1 phone)
    assert sh["phones"].shape == (1, 1) and sh["duration"].shape == (1, 1)
    assert sh["mel"].shape == (1, 2, 2) and torch.equal(sh["mel"], batch["mel"][2:, :2])
    assert sh["variances_pitch"].shape == (1, 2) and sh["priors_energy"].shape == (1,)
    sh0 = shard_batch(batch, 2, 0)  # rank 0: utterances 0, 1 -> 10 and 6 frames
    assert sh0["mel"].shape == (2, 10, 2) and sh0["variances_pitch"].shape == (2, 10)
    keep = shard_batch(batch, 2, 1, trim=False)
    assert keep["mel"].shape == (1, T, 2) and keep["phones"].shape == (1, 4)
    # real frames beyond sum(duration) are an error, not silently dropped ...
    bad = dict(batch, mel=batch["mel"].clone())
    bad["duration"] = torch.tensor([[2, 3, 1, 4], [5, 1, 0, 0], [1, 0, 0, 0]])   # utterance 2 now claims 1 frame; its mel has 10
    with pytest.raises(ValueError, match="non-zero frames"):
        shard_batch(bad, 2, 1)
    # ... and a phone-level (B, L) variances_* entry is not cut on the frame axis (zero-length durations make L > Tr possible)
    ph2 = torch.tensor([[3, 4, 5, 6], [7, 8, 9, 1]])
    dur2 = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0]])
    b2 = {"phones": ph2, "duration": dur2, "speaker": torch.zeros(2, 256), "mel": torch.zeros(2, 6, 2),
          "variances_energy": torch.arange(8.0).reshape(2, 4)}
    s2 = shard_batch(b2, 1, 0)
    assert s2["variances_energy"].shape == (2, 4) and s2["mel"].shape == (2, 1, 2)


def test_collective_device_follows_the_backend():
    from lightningfastspeech2_amd import dist as D
    import torch.distributed as td
    td.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:29533")
    try:
        assert D.collective_device(None, None) == torch.device("cpu")          # host-side batch under gloo
        assert D.collective_device(None, torch.zeros(1)) == torch.device("cpu")
        # under nccl the control tensors follow the tensor the rank already holds (a rank that never called
        # torch.cuda.set_device), the current device only without one
        import unittest.mock as um
        like = um.MagicMock(spec=torch.Tensor)
        like.is_cuda, like.device = True, torch.device("cuda", 3)
        with um.patch.object(D.dist, "get_backend", return_value="nccl"), um.patch.object(torch.cuda, "current_device", return_value=0):
            assert D.collective_device(None, like) == torch.device("cuda", 3)
            assert D.collective_device(None, None) == torch.device("cuda", 0)
            assert D.collective_device(None, torch.zeros(1)) == torch.device("cuda", 0)
        # an explicit device wins in forward_sharded; an empty shard no longer guesses from batch["speaker"]
        out = D.forward_sharded(lambda b: {"mel": torch.zeros(1, 3, 4), "tgt_mask": torch.zeros(1, 3, dtype=torch.bool)},
                                {"phones": torch.ones(1, 2, dtype=torch.long), "speaker": [[0.0] * 256]}, n_mels=4, device="cpu")
        assert out[0].shape == (1, 3, 4)
    finally:
        td.destroy_process_group()


def test_global_pad_mode_equals_the_whole_batch():
    """SURVEY 8e global-pad mode on a ragged batch: shards keep the batch's phone length and pad their frames to the
    all-reduced maximum, so the gathered mels equal the single-process whole-batch run (pad leakage included)."""
    from oracle import oracle_cpu
    mel_all, frames = _run(2, 5, "global")
    cfg, sd, batch = _batch(5)
    ref = oracle_cpu.forward(sd, cfg, batch["phones"], batch["speaker"])
    assert mel_all.shape == tuple(ref["mel"].shape)
    for i in range(5):
        n = int((~ref["tgt_mask"][i]).sum())
        assert frames[i] == n
        np.testing.assert_allclose(mel_all[i, :n], ref["mel"][i, :n].numpy(), rtol=0, atol=1e-6)
        assert not mel_all[i, n:].any()


def test_world_larger_than_batch_does_not_hang():
    """B < world leaves empty shards; they must still join the gather."""
    from oracle import oracle_cpu
    mel_all, frames = _run(3, 2, "shard")
    cfg, sd, batch = _batch(2)
    assert mel_all.shape[0] == 2
    for r in range(2):
        ref = oracle_cpu.forward(sd, cfg, batch["phones"][r:r + 1, :int((batch["phones"][r] != 0).sum())], batch["speaker"][r:r + 1])
        n = int((~ref["tgt_mask"][0]).sum())
        assert frames[r] == n
        np.testing.assert_allclose(mel_all[r, :n], ref["mel"][0, :n].numpy(), rtol=0, atol=1e-6)
    mel_all, frames = _run(3, 2, "global")
    ref = oracle_cpu.forward(sd, cfg, batch["phones"], batch["speaker"])
    np.testing.assert_allclose(mel_all[0, :int(frames[0])], ref["mel"][0, :int(frames[0])].numpy(), rtol=0, atol=1e-6)


def test_teacher_forced_sharded_keeps_durations_aligned():
    from oracle import oracle_cpu
    mel_all, frames = _run(2, 5, "teacher")
    cfg, sd, batch = _batch(5)
    whole = oracle_cpu.forward(sd, cfg, batch["phones"], batch["speaker"])
    row = 0
    for r in range(2):
        sh = shard_batch({**batch, "duration": whole["duration_rounded"]}, 2, r)
        assert sh["duration"].shape == sh["phones"].shape
        ref = oracle_cpu.forward(sd, cfg, sh["phones"], sh["speaker"], force_durations=sh["duration"])
        for i in range(ref["mel"].shape[0]):
            n = int((~ref["tgt_mask"][i]).sum())
            np.testing.assert_allclose(mel_all[row, :n], ref["mel"][i, :n].numpy(), rtol=0, atol=1e-6)
            row += 1


@pytest.mark.parametrize("B", [4, 5])
def test_two_rank_gather_equals_per_shard_oracle(B):
    mel_all, frames = _run(2, B, "shard")
    # expected: the oracle on each shard separately (each shard is its own padded batch — pad
    # leakage makes that differ from the whole-batch result)
    from oracle import oracle_cpu
    cfg = _cfg()
    sd = synth_state_dict(cfg, 5, randomize_norm=True, duration_bias=1.0)
    inp = synth_inputs(cfg, B, 9, seed=3, lengths=[9, 4, 7, 2, 6][:B])
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    assert mel_all.shape[0] == B
    row = 0
    for r in range(2):
        sh = shard_batch(batch, 2, r)
        ref = oracle_cpu.forward(sd, cfg, sh["phones"], sh["speaker"])
        for i in range(ref["mel"].shape[0]):
            n = int((~ref["tgt_mask"][i]).sum())
            assert frames[row] == n
            np.testing.assert_allclose(mel_all[row, :n], ref["mel"][i, :n].numpy(), rtol=0, atol=1e-6)
            assert not mel_all[row, n:].any()
            row += 1


@pytest.mark.parametrize("world,B,root", [(2, 5, 1), (3, 2, 0), (2, 4, 0)])
def test_root_only_gather_equals_the_all_gather(world, B, root):
    """`dst=`: the north star's literal gather - only the root holds the (B, T_max, n_mels) result; ragged shards, an empty
    shard (world 3, batch 2) and a non-zero root; the same rows as the all-gather delivers to everyone."""
    mel_root, fr_root = _run(world, B, f"root{root}")
    mel_all, fr_all = _run(world, B, "shard")
    assert mel_root.shape == mel_all.shape
    np.testing.assert_array_equal(fr_root, fr_all)
    np.testing.assert_array_equal(mel_root, mel_all)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lightningfastspeech2_amd.dist import all_reduce_gradients
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    all_reduce_gradients(g, bucket_bytes=4 * 300)          # 4 buckets, the last one short
    h = torch.full((77,), float(rank))
    for w in all_reduce_gradients(h, async_op=True):
        w.wait()
    # plain lists: a torch tensor on an mp.Queue travels as a shared-memory fd the parent may open after this worker is gone
    q.put((rank, g.tolist(), h.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_two_ranks():
    """The training step's data-parallel exchange: the flat gradient buffer summed over the ranks in buckets."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in ps:
        p_.start()
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    for p_ in ps:
        p_.join(timeout=60)
    for _, g, h in res:
        assert g == (torch.arange(1000, dtype=torch.float32) * 3).tolist()
        assert h == [1.0] * 77


def _report_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lightningfastspeech2_amd.dist import gather_mels, rank_report
    # each rank "ran" its own shard: 3 + rank utterances of 10 + rank frames, 4 mel bins, ragged frame counts
    Br, Tr = 3 + rank, 10 + rank
    mel = torch.full((Br, Tr, 4), float(rank + 1))
    mask = torch.arange(Tr)[None, :] >= torch.tensor([Tr - i for i in range(Br)])[:, None]
    frames_r = int((~mask).sum())
    mel_all, frames = gather_mels(mel, mask)
    rep = rank_report(ms_per_step=2.0 + rank, ms_without_gather=1.5 + rank, ms_with_gather=1.75 + rank, frames_per_step=frames_r,
                      gather_bytes=mel.numel() * 4 + Br * 8, local_rank=rank, device_index=rank, device=torch.device("cpu"))
    q.put((rank, rep, int(frames.sum()), frames_r))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_dist_object_is_internally_consistent_at_world_two():
    """What a multi-rank bench line's `dist` object carries (bench.py -> dist.rank_report), built by two gloo ranks: every rank
    returns the same object, the ranks seen are exactly the job's, the per-rank frames add up to what the gathered frame counts
    say, the bytes add up, and `gather_ms_exposed` is the difference of the maxima."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_report_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rep0, gathered0, fr0), (_, rep1, gathered1, fr1) = got
    assert rep0 == rep1
    assert rep0["backend"] == "gloo" and rep0["world_size"] == 2
    assert rep0["ranks_seen"] == [0, 1] and rep0["local_ranks"] == [0, 1] and rep0["devices"] == [0, 1]
    assert rep0["per_rank_frames_per_step"] == [fr0, fr1] and sum(rep0["per_rank_frames_per_step"]) == gathered0 == gathered1
    assert rep0["per_rank_gather_bytes"] == [3 * 10 * 4 * 4 + 3 * 8, 4 * 11 * 4 * 4 + 4 * 8]
    assert rep0["gather_bytes_total_per_step"] == sum(rep0["per_rank_gather_bytes"])
    assert rep0["gather_bytes_per_rank"] == max(rep0["per_rank_gather_bytes"])
    assert rep0["per_rank_ms_per_step"] == [2.0, 3.0]
    assert rep0["gather_ms_exposed"] == round(max(rep0["per_rank_ms_per_step_with_gather_ab"]) - max(rep0["per_rank_ms_per_step_without_gather"]), 4) == 0.25
