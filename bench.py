#!/usr/bin/env python
"""Headline benchmark: mel-frames/s of the FastSpeech2 mel forward on synthetic 256-phoneme x
batch-32 inputs (BASELINE.json metric / configs[1]: FS2-27M, bf16, batch 32, 256 phonemes).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (fs2_encode + fs2_decode through the drop-in model object)
over one batch per rank, inputs already resident in HBM; with N > 1 every rank runs its own
batch-32 shard (weak scaling) and the final mels are all-gathered with RCCL.  Rank 0 prints ONE
JSON line.  ``roofline`` is for the dominant kernel (the implicit-GEMM dense Conv1d): algorithmic
FLOPs per launch / average launch duration measured with HIP events on the launch stream inside the
timed region.  ``cpu_baseline`` times the CPU oracle (a port, not the product) on a bounded sample
of the same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HOP, SR = 256, 22050


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", help="c2 (FS2-27M) | c3 (LS-76M) | c5 (FS2-1B) | ref-default")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--phones", type=int, default=256)
    ap.add_argument("--frames-per-phone", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-predictor", action="store_true", help="A/B: variance predictors layer by layer")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    return ap.parse_args()


def cpu_baseline(cfg, sd, args):
    """The CPU oracle on the first `cpu-sample-batch` utterances of the same synthetic workload.
    torch's intra-op pool stops scaling long before a 2 x 64-core host is full (the forward is many
    small ops), so the thread count is swept and the BEST one is reported (a strong baseline)."""
    from lightningfastspeech2_amd.weights import synth_inputs
    from oracle import oracle_cpu  # cpu_baseline leg only
    B = args.cpu_sample_batch
    inp = synth_inputs(cfg, args.batch, args.phones, seed=1234)
    ph, sp = inp["phones"][:B], inp["speaker"][:B]
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()

    def one_pass():
        t0 = time.perf_counter()
        out = oracle_cpu.forward(sd, cfg, ph, sp)
        return time.perf_counter() - t0, int((~out["tgt_mask"]).sum())

    best_nt, best_rate, spent = default_threads, 0.0, 0.0
    for nt in sorted({n for n in (8, 16, 32, 64, default_threads) if n <= ncpu}):
        torch.set_num_threads(nt)
        oracle_cpu.forward(sd, cfg, ph[:1], sp[:1])  # warm the pool at this size
        t, fr = one_pass()
        spent += t
        if fr / t > best_rate:
            best_nt, best_rate = nt, fr / t
    torch.set_num_threads(best_nt)
    reps, t_tot, frames = 3, 0.0, 0
    for _ in range(reps):
        t, fr = one_pass()
        t_tot += t
        frames += fr
    torch.set_num_threads(default_threads)
    return {"value": frames / t_tot, "unit": "mel-frames/s", "cores": best_nt, "kind": "port",
            "sample": f"oracle/oracle_cpu.py (torch {torch.__version__} CPU fp32 ops), {reps} passes over the "
                      f"first {B} of the {args.batch} x {args.phones}-phoneme utterances at the best of a "
                      f"thread sweep ({best_nt} threads; host has {ncpu} logical CPUs), {t_tot + spent:.1f} s of CPU work",
            "rtf": t_tot / (frames * HOP / SR)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FS2_BENCH_DEVICE / FS2_BENCH_BACKEND exist only to rehearse the multi-rank control flow on a
    # single-GPU box (all ranks on one device over gloo); the real run is one rank per GPU over RCCL.
    backend = os.environ.get("FS2_BENCH_BACKEND", "nccl")
    if "FS2_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["FS2_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    if world != args.gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")

    from lightningfastspeech2_amd import _lib
    from lightningfastspeech2_amd.config import preset
    from lightningfastspeech2_amd.dist import gather_mels_async
    from lightningfastspeech2_amd.model import FastSpeech2
    from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict

    cfg = preset(args.config)
    # duration head weight 0 / bias ln(1+f): every phone gets f frames -> T = f * phones
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(1.0 + args.frames_per_phone), duration_weight_scale=0.0)
    model = FastSpeech2(cfg, sd, precision=args.precision, device=dev)
    if args.no_fused_predictor:
        model.engine.set_fused_predictor(False)
    if os.environ.get("FS2_XCD_REMAP"):  # A/B: 0 = plain tile order in the slab GEMM
        _lib.load().fs2_op_set_gemm_variant(200 + int(os.environ["FS2_XCD_REMAP"]))
    inp = synth_inputs(cfg, args.batch, args.phones, seed=1234 + 17 * rank)
    batch = {"phones": torch.from_numpy(inp["phones"]).to(dev), "speaker": torch.from_numpy(inp["speaker"]).to(dev)}

    # N > 1: the all-gather of step i's mels rides the collective stream underneath the forward of
    # step i+1 (at most one in flight; the last one is waited for before the closing barrier + sync)
    pending = []

    def step():
        out = model(batch, inference=True)
        if world > 1:
            if pending:
                pending.pop().wait()
            if backend == "nccl":
                pending.append(gather_mels_async(out["mel"], out["tgt_mask"]))
            else:  # rehearsal path: gloo moves host tensors
                pending.append(gather_mels_async(out["mel"].cpu(), out["tgt_mask"].cpu()))
        return out

    def drain():
        if pending:
            mel_all, frames = pending.pop().wait()
            assert mel_all.shape[0] == frames.numel()

    for _ in range(args.warmup):
        out = step()
    drain()
    frames_rank = int((~out["tgt_mask"]).sum())
    T = int(out["mel"].shape[1])

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # dominant kernel: the decoder FFN's dense k-tap conv (implicit GEMM); depth-wise configs have no
    # dense conv, there the pointwise GEMM launches (aggregated) dominate
    kcls = _lib.K_DEC_FFN_CONV1 if not cfg.decoder_depthwise_conv else _lib.K_GEMM
    model.engine.profile_enable(kcls, True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    prof = model.engine.profile_read(kcls)
    model.engine.profile_enable(kcls, False)

    cdev = dev if backend == "nccl" else torch.device("cpu")
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    frames_all = torch.tensor([frames_rank], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(frames_all, op=dist.ReduceOp.SUM)
    elapsed = float(t_max.item())
    total_frames = int(frames_all.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = total_frames * args.steps / elapsed
        n = max(prof["launches"], 1)
        avg_s = prof["ms"] / n * 1e-3
        achieved = (prof["flops"] / n) / avg_s if avg_s > 0 else 0.0
        peak = MFMA_PEAK[args.precision]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and args.config == "c2" and args.precision == "bf16":
            try:
                traffic = json.load(open(tpath)).get("conv_gemm_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "mel-frames/sec (whole node), 256-phoneme batch-32",
            "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "rtf": elapsed / args.steps / (total_frames * HOP / SR),
            "config": {"workload": f"{args.config}: "
                       + {"c2": "FS2-27M dense k=9 H=256 F=1024 4+4 layers",
                          "c3": "LS-76M depth-wise H=768 F=3072 4+5 layers",
                          "c5": "FS2-1B dense k=9 H=1024 F=4096 12+12 layers"}.get(args.config, args.config)
                       + f", batch {args.batch}/GPU x {args.phones} phonemes, {args.frames_per_phone} frames/phone "
                         f"-> T={T}, random-init weights", "batch_per_gpu": args.batch, "phonemes": args.phones,
                       "frames_per_utterance": T, "global_batch": args.batch * world,
                       "parallelism": f"dp{world} (utterance shards, RCCL all-gather of mels only)" if world > 1 else "single GPU",
                       "params": cfg.param_count()},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "kernel": ("gemm_conv_slab_kernel: decoder FFN conv1, implicit-GEMM Conv1d "
                                    f"M={args.batch * T} N={cfg.decoder_conv_filter_size} K={cfg.decoder_kernel_sizes[0]}x{cfg.hidden}")
                         if kcls == _lib.K_DEC_FFN_CONV1 else "gemm_conv_slab_kernel (pointwise GEMM launches, aggregated)",
                         "launches_timed": prof["launches"], "avg_launch_us": avg_s * 1e6,
                         "flops_per_launch": prof["flops"] / n},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
