"""Worker of tests/test_gpu_training.py::test_two_rank_data_parallel_step (one process per rank, both on this box's GPU,
gloo): shard the batch, one training step per rank, gradient all-reduce inside optimizer_step; rank 0 writes the weights."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(out_path):
    from lightningfastspeech2_amd.dist import shard_batch
    from lightningfastspeech2_amd.training import Trainer
    from test_gpu_training import _case
    dist.init_process_group(os.environ.get("FS2_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    cfg, sd, batch = _case(41, 4, 11, [11, 9, 6, 2])
    full = {k: torch.as_tensor(v) for k, v in batch.items()}
    # the collate format zero-pads every frame-level target beyond an utterance's own frames (datasets.py:866-877); the seeded
    # batch is random there, and shard_batch refuses to cut non-zero frames
    frames = full["duration"].sum(1)
    for k in list(full):
        if k == "mel" or k.startswith("variances_"):
            keep = torch.arange(full[k].shape[1])[None, :] < frames[:, None]
            full[k] = full[k] * (keep[..., None] if full[k].dim() == 3 else keep)
    mine = shard_batch(full, world, rank, trim=True)
    T = int(mine["duration"].sum(1).max())  # the shard is its own padded batch: frame-level targets cut to its longest utterance
    for k in list(mine):
        if k == "mel" or k.startswith("variances_"):
            mine[k] = mine[k][:, :T].contiguous()
    tr = Trainer(cfg, sd, lr=1e-3, warmup_steps=2, gradient_clip_val=1.0)
    tr.training_step({k: v.cuda() for k, v in mine.items()})
    tr.optimizer_step()
    after = tr.state_dict()
    flat = torch.cat([v.reshape(-1).float() for k, v in after.items() if not (k.endswith(".pe") or k.endswith(".bins"))])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        assert all(torch.equal(g, gathered[0]) for g in gathered), "ranks diverged"
        np.savez(out_path, **{k: v.numpy() for k, v in after.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
