#!/usr/bin/env python
"""Headline benchmark: mel-frames/s of the FastSpeech2 mel forward on synthetic 256-phoneme x
batch-32 inputs (BASELINE.json metric / configs[1]: FS2-27M, bf16, batch 32, 256 phonemes).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (fs2_encode + fs2_decode through the drop-in model object)
over one batch per rank, inputs already resident in HBM; with N > 1 every rank runs its own
batch-32 shard (weak scaling) and the final mels are all-gathered with RCCL.  Rank 0 prints ONE
JSON line.
  roofline      the dominant kernel (the implicit-GEMM dense Conv1d): algorithmic FLOPs per launch /
                average launch duration measured with HIP events on the launch stream inside the timed
                region; traffic = HBM bytes per launch from this round's committed PMC passes.
  cpu_baseline  the CPU oracle (a port, not the product) on this box's host cores, rank 0 / N=1 only.
  parity        what tolerance the timed mode holds: the same engine in fp32 parity mode (ms/step, mel
                max-abs vs the oracle on a sample), and for the timed bf16 mode the free-running decision
                flips vs the oracle and the mel error under the oracle's decisions.
  value_incl_pcie  the same step with the inputs starting and the mels ending in pinned host memory.
"""
import argparse
import json
import math
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12  # HBM3E spec (same guide; ~6.3e12 measured-achievable)
HOP, SR = 256, 22050
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")  # regenerated every round: tools/pmc_traffic.sh


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", help="c2 (FS2-27M) | c3 (LS-76M) | c5 (FS2-1B) | ref-default")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "mixed", "mixed3", "fp32x3"])
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--phones", type=int, default=256)
    ap.add_argument("--frames-per-phone", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step side measurement")
    ap.add_argument("--no-fused-predictor", action="store_true", help="A/B: variance predictors layer by layer")
    ap.add_argument("--in-flight", type=int, default=2, help="forwards in flight (model.pipeline): 1 = one synchronous forward at a time; "
                    "the timed region takes the faster of 1 and this many, measured in the warm-up (FS2_BENCH_IN_FLIGHT pins it)")
    ap.add_argument("--parity-sample", type=int, default=0,
                    help="utterances of the workload the parity block checks against the oracle (0 = the whole batch, the default: ~3 s of CPU at C2)")
    return ap.parse_args()


def physical_cores():
    """(physical cores, model name) from lscpu; falls back to the logical count."""
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        return int(kv["Core(s) per socket"]) * int(kv["Socket(s)"]), kv.get("Model name", "?")
    except Exception:
        return os.cpu_count() or 1, "?"


def cpu_baseline(cfg, sd, args):
    """The CPU oracle on the same synthetic workload: a thread sweep on 8 utterances (torch's intra-op pool
    stops scaling long before a 2 x 64-core host is full, so the BEST count is the baseline of record and the
    physical-core count is one of the candidates), then the WHOLE batch at that count (1 warm-up + 3 passes,
    median), and a 1-thread figure on 2 utterances.  Also returns the oracle's outputs for the parity block."""
    from lightningfastspeech2_amd.weights import synth_inputs
    from oracle import oracle_cpu  # cpu_baseline leg only
    inp = synth_inputs(cfg, args.batch, args.phones, seed=1234)
    ph, sp = inp["phones"], inp["speaker"]
    ncpu = os.cpu_count() or 1
    phys, model = physical_cores()
    default_threads = torch.get_num_threads()
    spent = 0.0

    def one_pass(B):
        t0 = time.perf_counter()
        out = oracle_cpu.forward(sd, cfg, ph[:B], sp[:B])
        return time.perf_counter() - t0, int((~out["tgt_mask"]).sum())

    sweep = {}
    Bs = min(8, args.batch)
    for nt in sorted({n for n in (8, 16, 32, 64, phys, default_threads) if n <= ncpu}):
        torch.set_num_threads(nt)
        oracle_cpu.forward(sd, cfg, ph[:1], sp[:1])  # warm the pool at this size
        t, fr = one_pass(Bs)
        spent += t
        sweep[nt] = fr / t
    best_nt = max(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    t, _ = one_pass(args.batch)  # warm-up at the full batch
    spent += t
    times, frames = [], 0
    for _ in range(3):
        t, frames = one_pass(args.batch)
        times.append(t)
        spent += t
    med = statistics.median(times)
    # the same workload the way generate.py:186-195 feeds the reference: --batch_size chunks (8 utterances each), one forward per
    # chunk - torch's intra-op CPU kernels are at their best there (the sweep above), so this is the stronger CPU baseline
    ctimes, cframes = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        cframes = 0
        for lo in range(0, args.batch, Bs):
            o = oracle_cpu.forward(sd, cfg, ph[lo:lo + Bs], sp[lo:lo + Bs])
            cframes += int((~o["tgt_mask"]).sum())
        ctimes.append(time.perf_counter() - t0)
        spent += ctimes[-1]
    cmed = statistics.median(ctimes)
    whole, chunked = frames / med, cframes / cmed
    torch.set_num_threads(1)
    t1, f1 = one_pass(min(2, args.batch))
    spent += t1
    torch.set_num_threads(default_threads)
    best_t, best_f = (cmed, cframes) if chunked >= whole else (med, frames)
    return {"value": max(whole, chunked), "unit": "mel-frames/s", "cores": best_nt, "kind": "port",
            "sample": f"oracle/oracle_cpu.py (torch {torch.__version__} CPU fp32 ops) on all {args.batch} x {args.phones}-phoneme "
                      f"utterances of the workload, the better of (a) one forward over the whole batch and (b) {Bs}-utterance chunks as "
                      f"generate.py:186-195 feeds the reference; each 3 passes, median, at the best of a thread sweep over "
                      f"{sorted(sweep)} ({best_nt} threads); {spent:.1f} s of CPU work in total",
            "value_whole_batch_one_call": whole, "value_8_utterance_chunks": chunked,
            "rtf": best_t / (best_f * HOP / SR),
            "physical_cores": phys, "logical_cpus": ncpu, "cpu_model": model,
            "value_at_physical_cores": sweep.get(phys), "value_1_thread": f1 / t1,
            "thread_sweep_frames_per_s": {str(k): round(v) for k, v in sweep.items()}}


def parity_block(cfg, sd, args, dev, timed_model):
    """fp32 parity mode timed + checked against the oracle on a sample of the workload, and the timed mode's
    decision flips / forced-decision mel error on the same sample (rank 0, N=1)."""
    from lightningfastspeech2_amd import _lib
    from lightningfastspeech2_amd.model import FastSpeech2
    from lightningfastspeech2_amd.weights import synth_inputs
    from oracle import oracle_cpu  # checker only
    Bs = args.batch if args.parity_sample <= 0 else max(1, min(args.parity_sample, args.batch))
    inp = synth_inputs(cfg, args.batch, args.phones, seed=1234)
    ph, sp = inp["phones"][:Bs], inp["speaker"][:Bs]
    ref = oracle_cpu.forward(sd, cfg, ph, sp, return_intermediates=True)
    sample = {"phones": torch.from_numpy(ph).to(dev), "speaker": torch.from_numpy(sp).to(dev)}
    full = {"phones": torch.from_numpy(inp["phones"]).to(dev), "speaker": torch.from_numpy(inp["speaker"]).to(dev)}
    ref_b = {v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}

    def check(model):
        model.engine.set_debug(True)
        out = model(sample, inference=True)
        dfl = int((out["duration_rounded"].cpu() != ref["duration_rounded"]).sum())
        same_T = out["mel"].shape == ref["mel"].shape
        bfl = -1
        if dfl == 0 and same_T:
            bfl = sum(int((model.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref_b[v]).sum()) for v in cfg.variances)
        free = float((out["mel"].cpu() - ref["mel"]).abs().max()) if same_T else float("nan")
        out = model.forward(sample, force_durations=ref["duration_rounded"], force_buckets=ref_b)
        forced = float((out["mel"].cpu() - ref["mel"]).abs().max())
        model.engine.set_debug(False)
        return dfl, bfl, free, forced

    def timed(model, n):
        """ms per batch: one synchronous forward at a time, and with the timed region's number of forwards in flight"""
        for _ in range(2):
            model(full, inference=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(full, inference=True)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / n * 1e3
        nf = max(1, args.in_flight)
        if nf == 1:
            return one, one
        pipe = model.pipeline(nf)
        for _ in range(2 * nf):
            pipe.submit(full)
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2 * n):
            pipe.submit(full)
        pipe.drain()
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / (2 * n) * 1e3
        pipe.close()
        return one, piped

    res = {"sample": (f"all {args.batch} utterances of the timed workload (the timed batch itself)" if Bs == args.batch else
                      f"first {Bs} of the {args.batch} utterances of the timed workload (as their own batch: utterances "
                      f"are independent and unpadded here)") + ", oracle/oracle_cpu.py as the reference; every mode below is checked on this sample",
           "buckets_compared": int(sum(ref_b[v].numel() for v in cfg.variances)),
           "durations_compared": int(ref["duration_rounded"].numel()),
           "note": "the workload's duration head is weight 0 / bias ln 7 (T fixed at 6 frames/phone), so duration flips are "
                   "0 by construction here; tests/test_gpu_forward.py reports them on ragged random-head batches"}
    if args.precision != "fp32":
        m32 = FastSpeech2(cfg, sd, precision="fp32", device=dev)
        dfl, bfl, free, forced = check(m32)
        for _ in range(2):
            m32(full, inference=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n32 = 3
        for _ in range(n32):
            m32(full, inference=True)
        torch.cuda.synchronize()
        res.update({"fp32_ms_per_step": (time.perf_counter() - t0) / n32 * 1e3, "fp32_duration_flips": dfl,
                    "fp32_bucket_flips": bfl, "fp32_mel_maxabs_vs_oracle": forced if bfl else free,
                    "fp32_mel_maxabs_is": "under the oracle's decisions" if bfl else "free-running"})
        del m32
        # the parity mode's layout and row arithmetic with every GEMM / conv as bf16 x 3 split products (FS2_F32_X3)
        mx = FastSpeech2(cfg, sd, precision="fp32x3", device=dev)
        dfl, bfl, free, forced = check(mx)
        one, piped = timed(mx, 3)
        res["fp32x3"] = {"mode": "fp32 storage / softmax / LayerNorm / heads; every GEMM, conv and attention product as bf16 x 3 split products of the fp32 operands",
                         "ms_per_step": min(one, piped), "ms_per_step_one_in_flight": one, "in_flight": args.in_flight if piped < one else 1,
                         "duration_flips": dfl, "bucket_flips": bfl,
                         "mel_maxabs_vs_oracle": forced if bfl else free,
                         "mel_maxabs_is": "under the oracle's decisions" if bfl else "free-running"}
        del mx
    if args.precision == "bf16":  # the decision-safe throughput mode beside it: fp32-grade front, bf16 decoder
        m3 = FastSpeech2(cfg, sd, precision="mixed3", device=dev)
        dfl, bfl, free, forced = check(m3)
        one, piped = timed(m3, 5)
        res["decision_safe"] = {"mode": "mixed3 (front: fp32 storage, bf16 x 3 split products incl. attention; decoder: bf16)",
                                "ms_per_step": min(one, piped), "ms_per_step_one_in_flight": one, "in_flight": args.in_flight if piped < one else 1,
                                "value": float(args.batch * ref["mel"].shape[1] / (min(one, piped) * 1e-3)) if Bs == args.batch else None,
                                "unit": "mel-frames/s", "duration_flips": dfl, "bucket_flips": bfl,
                                "mel_maxabs_forced": forced}
        # its own roofline: the mode's dominant kernel is the single-launch variance predictor in the split
        # arithmetic - three bf16 MFMAs per product, so the dense peak for its ALGORITHMIC flops is a third of the bf16 peak
        try:
            m3.engine.profile_reserve(_lib.K_PREDICTOR, 64)
            m3.engine.profile_enable(_lib.K_PREDICTOR, True)
            for _ in range(3):
                m3(full, inference=True)
            torch.cuda.synchronize()
            pp = m3.engine.profile_read(_lib.K_PREDICTOR)
            m3.engine.profile_enable(_lib.K_PREDICTOR, False)
            if pp["launches"]:
                ach = pp["flops"] / (pp["ms"] * 1e-3)
                res["decision_safe"]["roofline"] = {
                    "bound": "mfma", "kernel": "predictor_fused_kernel<7, 4, 1, X3>: a whole dense VariancePredictor per launch, every product as "
                                               "three bf16 MFMAs on head / tail splits of the fp32 operands",
                    "achieved": ach / 1e12, "peak": 2500.0 / 3, "unit": "TFLOP/s", "frac": ach / (2.5e15 / 3),
                    "peak_is": "dense bf16 MFMA peak / 3 (three MFMAs per algorithmic product)",
                    "avg_launch_us": pp["ms"] / pp["launches"] * 1e3, "launches_timed": pp["launches"], "flops_per_launch": pp["flops"] / pp["launches"],
                    "measured": "HIP events on the launch stream around the decode phase's predictor launches, 3 eager forwards"}
        except Exception as ex:
            res["decision_safe"]["roofline"] = {"error": repr(ex)}
        del m3
    dfl, bfl, free, forced = check(timed_model)
    p = args.precision
    res.update({f"{p}_duration_flips": dfl, f"{p}_bucket_flips": bfl, f"{p}_mel_maxabs_free": free,
                f"{p}_mel_maxabs_forced": forced, "mel_scale": float(ref["mel"].abs().max())})
    return res


def training_block(cfg, sd, args, inp, T):
    """The same synthetic batch through `Trainer.training_step` + `optimizer_step` (teacher-forced forward, losses, backward,
    clip, AdamW): 2 warm-up + 5 timed steps.  Reported beside the forward metric, never as `value`."""
    from lightningfastspeech2_amd.training import Trainer
    del_model_mem = torch.cuda.memory_allocated()
    rs = np.random.RandomState(5)
    B, L = args.batch, args.phones
    batch = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda(),
             "duration": torch.full((B, L), args.frames_per_phone, dtype=torch.int64).cuda(),
             "mel": torch.from_numpy((rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)).cuda()}
    for v in cfg.variances:
        batch[f"variances_{v}"] = torch.from_numpy(rs.randn(B, T).astype(np.float32)).cuda()
    tr = Trainer(cfg, sd, precision=args.precision)
    for _ in range(2):
        tr.training_step(batch)
        tr.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 5
    for _ in range(k):
        losses = tr.training_step(batch)
        tr.optimizer_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    return {"what": "teacher-forced forward + FastSpeech2Loss + backward + clip + AdamW/Noam on the same batch (lightningfastspeech2_amd.training)",
            "precision": args.precision, "ms_per_step": dt * 1e3, "mel_frames_per_s": B * T / dt, "steps": k,
            "loss_total": float(losses["total"]), "extra_mem_GB": (torch.cuda.max_memory_allocated() - del_model_mem) / 2**30}


CU_INGEST_GBS = 64.0   # per-CU L2 -> LDS ceiling measured on this part (tools/probes/cu_ingest.hip: 61-69 GB/s, profiles/HISTORY.md §4)
N_CUS = 256


def encoder_mha_ingest_bound(cfg, B, L, block_s, fused_tail=False):
    """Roofline of the encoder self-attention block (in-projection, attention, out-projection + residual + LayerNorm) against the
    per-CU ingest ceiling: the bytes ONE CU has to pull through L2 into LDS / registers for its share of each launch, at the tile
    shapes the launchers pick for M = B L rows (gemm_mfma.hip launch_gemm_plain's cost model: 128-row x 256-column tiles for the
    in-projection, 32-row tiles with the whole row for the LayerNorm epilogue; attention.hip: 64 queries of one head per workgroup
    with all of that head's K and V).  bound_us = sum over the launches of rounds x bytes per workgroup / 64 GB/s."""
    H, heads, M, e = cfg.hidden, cfg.encoder_head, B * L, 2
    if H != 256:  # the tile shapes below are the H = 256 launchers' (C2); wider rows take the deferred-LayerNorm forms
        return None
    d = H // heads
    launches = []
    # in-projection: M x 3H, K = H: tiles of 128 rows x 256 columns, each pulls its weight tile (256 x H) and its rows (128 x H)
    t_in = -(-M // 128) * -(-3 * H // 256)
    launches.append(("in_proj", t_in, (256 * H + 128 * H) * e))
    if fused_tail:
        # r06, attn_out_ln_kernel: 64 queries x BOTH heads per workgroup: Q (64 x H) + K, V of both heads (2 x L x H) + the H x H out-projection
        # weights + 64 residual rows
        launches.append(("attention+out_proj+LN", B * -(-L // 64), (64 * H + 2 * L * H + H * H + 64 * H) * e))
    else:
        # attention: 64 queries x one head per workgroup: Q (64 x d) + K, V of the head (2 x L x d)
        t_at = B * heads * -(-L // 64)
        launches.append(("attention", t_at, (64 * d + 2 * L * d) * e))
        # out-projection + residual + LayerNorm: 32-row tiles over whole rows: the H x H weights, 32 rows of input, 32 rows of residual
        t_out = -(-M // 32)
        launches.append(("out_proj+LN", t_out, (H * H + 2 * 32 * H) * e))
    per = []
    bound = 0.0
    for name, wgs, by in launches:
        rounds = -(-wgs // N_CUS)
        us = rounds * by / (CU_INGEST_GBS * 1e3)
        bound += us
        per.append({"launch": name, "workgroups": wgs, "bytes_per_workgroup": by, "rounds_over_256_cus": rounds, "bound_us": round(us, 2)})
    return {"bound": "cu-ingest (L2 -> LDS)", "peak": CU_INGEST_GBS, "unit": "GB/s per CU", "bound_us": round(bound, 2),
            "measured_us": round(block_s * 1e6, 2), "frac": bound / (block_s * 1e6) if block_s > 0 else 0.0, "launches": per,
            "reading": "the launches stream for bound_us of the measured time; the rest is what a 10-us launch is made of besides its "
                       "stream - dispatch and ramp over 256 CUs, the first operand round trip, the LayerNorm epilogue's row exchange, the store "
                       "drain (profiles/HISTORY.md §4, 'the encoder launches').  r06: attention + out-projection + residual + LayerNorm are ONE "
                       "launch (attn_out_ln_kernel, 128 workgroups; 34 -> 31 us per block); folding the in-projection in as well would have every "
                       "workgroup re-project its utterance's K and V (4 x redundant, ~67 MFLOP each): sized at no gain, not built"}


def main():
    args = parse()
    # The forward issues ~150 launches around one host sync: on a loaded pool host (load average 30-55 seen) the launching thread
    # being descheduled shows up as GPU idle time (2.4 -> 4.7 ms measured with identical kernel times).  Ask the scheduler for
    # priority; harmless where it is not permitted.  Reported in the JSON line ("host": nice value, load average).
    try:
        os.nice(-20)
    except OSError:
        pass
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FS2_BENCH_DEVICE / FS2_BENCH_BACKEND exist only to rehearse the multi-rank control flow on a
    # single-GPU box (all ranks on one device over gloo); the real run is one rank per GPU over RCCL.
    backend = os.environ.get("FS2_BENCH_BACKEND", "nccl")
    if "FS2_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["FS2_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    # FS2_BENCH_FORCE_DIST=1: take the multi-rank path (process group, sync-free mel gathers on the collective stream, closing
    # barrier, max over ranks) also at WORLD_SIZE 1 - how the RCCL branch gets exercised on a one-GPU box (tests/test_gpu_boundary.py)
    multi = world > 1 or os.environ.get("FS2_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    if world != args.gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")

    from lightningfastspeech2_amd import _lib
    from lightningfastspeech2_amd.config import preset
    from lightningfastspeech2_amd.dist import gather_mels_async, global_frames
    from lightningfastspeech2_amd.model import FastSpeech2
    from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict

    cfg = preset(args.config)
    # duration head weight 0 / bias ln(1+f): every phone gets f frames -> T = f * phones
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(1.0 + args.frames_per_phone), duration_weight_scale=0.0)
    model = FastSpeech2(cfg, sd, precision=args.precision, device=dev)
    if args.no_fused_predictor:
        model.engine.set_fused_predictor(False)
    if os.environ.get("FS2_DEFER_LN") == "0":  # A/B: one launch per LayerNorm in the wide depth-wise blocks
        model.engine.set_deferred_layernorm(False)
    if os.environ.get("FS2_FOLD_LN") == "0":  # A/B: a normalise-only pass per wide depth-wise block instead of folding norm2 into the next in-projection
        model.engine.set_folded_layernorm(False)
    applied_knobs = []
    def set_knob(k):  # a mistyped knob must not yield an unlabeled default-config measurement: set_tuning raises
        # operator-level switches (the training step's kernels, the strided-batched GEMM, the vocoder run on THIS thread's switches):
        # applied to the bench thread, which is what training_block / the vocoder read; the engine refuses them
        op_only = 900 <= k <= 909 or k in (1000, 1001, 1100, 1101, 700, 701, 800, 801, 500, 501)
        if op_only:
            _lib.check(_lib.load().fs2_op_set_gemm_variant(int(k)), None, f"fs2_op_set_gemm_variant({k})")
        else:
            model.engine.set_tuning(int(k))  # this engine's (replicas made later inherit it); the switches are not process state
            _lib.load().fs2_op_set_gemm_variant(int(k))  # ... and the bench thread's, for the operator-level launches of the same kernels
        applied_knobs.append({"knob": int(k), "scope": "thread" if op_only else "engine+thread"})
    if os.environ.get("FS2_GEMM_KNOBS"):  # A/B: comma-separated fs2_op_set_gemm_variant values (include/fs2.h)
        for k in os.environ["FS2_GEMM_KNOBS"].split(","):
            set_knob(int(k))
    if os.environ.get("FS2_XCD_REMAP"):  # A/B: 0 = plain tile order in the slab GEMM
        set_knob(200 + int(os.environ["FS2_XCD_REMAP"]))
    inp = synth_inputs(cfg, args.batch, args.phones, seed=1234 + 17 * rank)
    batch = {"phones": torch.from_numpy(inp["phones"]).to(dev), "speaker": torch.from_numpy(inp["speaker"]).to(dev)}

    # N > 1: the all-gather of step i's mels rides the collective stream underneath the forward of step i+1
    # (at most one in flight; the last one is waited for before the closing barrier + sync).  Every rank's
    # (B_r, T) is fixed for this workload and agreed ONCE after the first warm-up step, so a step queues its
    # collectives without any exchange or host read-back; pad rows are zeroed by the mel GEMM's own store.
    pending = []
    shapes = [None]
    if multi:
        model.engine.set_zero_pad_mel(True)

    gather_bytes = [0]

    # Forwards in flight: model.pipeline(n) keeps n engine replicas, each on its own HIP stream and host thread - batch i + 1's
    # encoder is queued while the host reads batch i's frame count (the forward's one host sync) and launches its decoder, and the
    # two forwards' launches fill each other's ramps and tails.  A step is still one forward over one batch; results come back in
    # submission order a step or two later (drain() collects the rest before the closing sync).
    pipes = {}
    cur = {"in_flight": 1}

    def forward_outs():
        n = cur["in_flight"]
        if n <= 1:
            return [model(batch, inference=True)]
        return pipes[n].submit(batch)

    def on_result(out, gather):
        if multi and gather:
            if pending:
                pending.pop().wait()
            if shapes[0] is None:
                cdev0 = dev if backend == "nccl" else torch.device("cpu")
                shapes[0] = ([args.batch] * world, global_frames(out["mel"].shape[1], cdev0))
            gather_bytes[0] = out["mel"].numel() * 4 + out["mel"].shape[0] * 8  # this rank's contribution: fp32 mels + int64 frame counts
            if backend == "nccl":
                pending.append(gather_mels_async(out["mel"], out["tgt_mask"], shapes=shapes[0], zeroed=True))
            else:  # rehearsal path: gloo moves host tensors
                pending.append(gather_mels_async(out["mel"].cpu(), out["tgt_mask"].cpu(), shapes=shapes[0], zeroed=True))

    def step(gather=True):
        for out in forward_outs():
            on_result(out, gather)

    def drain(gather=True):
        if cur["in_flight"] > 1:
            for out in pipes[cur["in_flight"]].drain():
                on_result(out, gather)
        if pending:
            mel_all, frames = pending.pop().wait()
            assert mel_all.shape[0] == frames.numel()

    out = model(batch, inference=True)  # shapes of the workload (T, frames) from one plain forward
    for _ in range(max(args.warmup, 1) if multi else args.warmup):
        step()
    drain()
    frames_rank = int((~out["tgt_mask"]).sum())
    T = int(out["mel"].shape[1])

    # Launch mode of the timed region: eager launches or both phases replayed as hipGraphs (bit-identical outputs,
    # tests/test_gpu_forward.py).  Which is faster depends on the host: eager is GPU-bound on a quiet one, a busy one shows up
    # as GPU idle time between ~150 launches and graphs take the host out of the step.  A few untimed steps of each, the faster
    # one runs the timed region (FS2_BENCH_MODE=eager|graphs pins it); every rank takes rank 0's choice.
    def timed_steps(n):
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            step()
        drain()
        torch.cuda.synchronize()
        return (time.perf_counter() - t_) / n
    tune = {}
    pin = os.environ.get("FS2_BENCH_MODE", "")
    pin_n = int(os.environ.get("FS2_BENCH_IN_FLIGHT", "0"))
    flights = sorted({1, max(1, args.in_flight)}) if not pin_n else [pin_n]
    for n in flights:
        if n > 1 and n not in pipes:
            pipes[n] = model.pipeline(n)
            for m in pipes[n].models[1:]:  # the replicas take the main engine's per-engine settings
                if multi:
                    m.engine.set_zero_pad_mel(True)
                if args.no_fused_predictor:
                    m.engine.set_fused_predictor(False)
                if os.environ.get("FS2_DEFER_LN") == "0":
                    m.engine.set_deferred_layernorm(False)
                if os.environ.get("FS2_FOLD_LN") == "0":
                    m.engine.set_folded_layernorm(False)
    for n in flights:
        cur["in_flight"] = n
        for mode in ("eager", "graphs"):
            if pin and pin != mode:
                continue
            model.engine.set_graphs(mode == "graphs")
            if n > 1:
                pipes[n].set_graphs(mode == "graphs")
            timed_steps(3 * n)  # first sight, capture (per replica)
            tune[f"{mode}/{n}"] = min(timed_steps(6), timed_steps(6)) * 1e3
    pick_key = min(tune, key=tune.get)
    if multi:
        keys = sorted(tune)
        flag = torch.tensor([keys.index(pick_key)], device=dev if backend == "nccl" else torch.device("cpu"))
        dist.broadcast(flag, 0)
        pick_key = keys[int(flag.item())]
    pick, n_pick = pick_key.split("/")[0], int(pick_key.split("/")[1])
    cur["in_flight"] = n_pick
    model.engine.set_graphs(pick == "graphs")
    if n_pick > 1:
        pipes[n_pick].set_graphs(pick == "graphs")
    timed_steps(2 * n_pick)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # dominant kernel: the decoder FFN's first GEMM - the dense k-tap conv (implicit GEMM), or in a depth-wise
    # (LightSpeech) block the pointwise H -> F GEMM behind the depth-wise conv.  Its HIP events are recorded in a pass of
    # their own BEHIND the timed region (same workload, same launch shapes, same stream): the timed steps replay both phases as
    # hipGraphs (Engine default), which cannot hold event records, and the timed region carries no measurement traffic.
    kcls = _lib.K_DEC_FFN_CONV1
    # Python's cyclic collector is held off over the timed steps (as `timeit` does): what it would collect here is host garbage of the
    # warm-up - an engine replica among it means fs2_destroy -> hipFree -> a device-wide stall of tens of ms inside the region
    import gc
    gc.collect()
    gc.disable()
    sync()
    trace = [] if os.environ.get("FS2_BENCH_TRACE") else None  # host-side time of each step() call (diagnostics)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if trace is not None:
            ts = time.perf_counter()
        step()
        if trace is not None:
            trace.append(time.perf_counter() - ts)
    drain()
    t_sync = time.perf_counter()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if trace is not None and rank == 0:
        print(f"closing sync: {(time.perf_counter() - t_sync) * 1e3:.2f} ms", file=sys.stderr)
        print("host ms per step():", " ".join(f"{t * 1e3:.2f}" for t in trace), file=sys.stderr)
    graph_replays = model.engine.graph_replays()
    cur["in_flight"] = 1  # the event-bracketed passes below are plain synchronous forwards
    # multi-rank evidence (checkable the moment a node exists): which ranks the collective library really connected, every rank's own
    # step time, the bytes a rank contributes to the mel gather, and what the gather costs the step - the same K steps once more
    # WITHOUT the gather (forward only), same launch mode, same barrier + sync brackets; exposed = with - without
    dist_info = None
    if multi:
        cdev_i = dev if backend == "nccl" else torch.device("cpu")
        # alternating blocks (with / without the gather), medians: a single pass behind the timed region measured the clock
        # drift of a warm GPU rather than the gather (r04: "exposed" came out negative)
        import statistics as _st
        nblk = max(4, args.steps // 2)
        t_with, t_without = [], []
        for _ in range(3):
            for g_on, acc in ((True, t_with), (False, t_without)):
                sync()
                t1 = time.perf_counter()
                for _ in range(nblk):
                    step(gather=g_on)
                drain(gather=g_on)
                sync()
                acc.append((time.perf_counter() - t1) / nblk)
        el_nog = _st.median(t_without) * args.steps
        el_wg = _st.median(t_with) * args.steps
        from lightningfastspeech2_amd.dist import rank_report
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        dist_info = {
            **rank_report(ms_per_step=elapsed / args.steps * 1e3, ms_without_gather=el_nog / args.steps * 1e3,
                          ms_with_gather=el_wg / args.steps * 1e3, frames_per_step=frames_rank, gather_bytes=int(gather_bytes[0]),
                          local_rank=local_rank, device_index=torch.cuda.current_device(), device=cdev_i),
            "rccl_version": ver if backend == "nccl" else None, "device_name": torch.cuda.get_device_name(dev),
            "forced_single_rank": world == 1,
            "what": "all-gathered from every rank over the process group the mel gather uses; gather_ms_exposed = max-over-ranks ms per step "
                    "with the asynchronous mel all-gather minus without it, medians of three alternating blocks of steps behind the timed region (the collective runs on the collective "
                    "library's stream underneath the next forward)"}
    model.engine.set_graphs(False)
    # roofline pass: K eager steps with the events of the dominant launch class on; then the whole forward's kernel time from
    # events around each step of a third pass (graphs again): ms_per_step against it says what the host adds
    psteps = max(3, min(args.steps, 10))
    model.engine.profile_reserve(kcls, psteps * (cfg.encoder_layers + cfg.decoder_layers) * 2 + 16)
    model.engine.profile_enable(kcls, True)
    for _ in range(psteps):
        model(batch, inference=True)
    torch.cuda.synchronize()
    prof = model.engine.profile_read(kcls)
    model.engine.profile_enable(kcls, False)
    # the north star's attention figure (BASELINE.json: ">= 40 % of the bf16 MFMA roofline" on the attention GEMMs): the decoder
    # stack's self-attention launches - the MFMA-bound instance (4 B T^2 H flops, AI = T / 2) - timed the same way, own pass
    model.engine.profile_reserve(_lib.K_DEC_ATTENTION, psteps * cfg.decoder_layers + 16)
    model.engine.profile_enable(_lib.K_DEC_ATTENTION, True)
    for _ in range(psteps):
        model(batch, inference=True)
    torch.cuda.synchronize()
    prof_att = model.engine.profile_read(_lib.K_DEC_ATTENTION)
    model.engine.profile_enable(_lib.K_DEC_ATTENTION, False)
    # the literal north-star instance: the ENCODER's self-attention block (in-projection + attention + out-projection + LayerNorm
    # launches of ConformerEncoderLayer.forward, model.py:108-116) against SURVEY 8d's fused-MHA figure 8 B L H^2 + 4 B L^2 H
    model.engine.profile_reserve(_lib.K_ENC_MHA, psteps * cfg.encoder_layers + 16)
    model.engine.profile_enable(_lib.K_ENC_MHA, True)
    for _ in range(psteps):
        model(batch, inference=True)
    torch.cuda.synchronize()
    prof_mha = model.engine.profile_read(_lib.K_ENC_MHA)
    model.engine.profile_enable(_lib.K_ENC_MHA, False)
    # GPU time of one forward = the sum over every launch class (conv GEMMs, GEMMs, attention, row kernels) of the HIP-event
    # intervals around its launches, in a pass of its own (eager; the dominant class is a subset of the conv / GEMM class)
    all_cls = [_lib.K_CONV_GEMM, _lib.K_GEMM, _lib.K_ATTENTION, _lib.K_ROWOPS]
    for k in all_cls:
        model.engine.profile_reserve(k, psteps * 160 + 16)
        model.engine.profile_enable(k, True)
    for _ in range(psteps):
        model(batch, inference=True)
    torch.cuda.synchronize()
    gpu_ms = 0.0
    for k in all_cls:
        gpu_ms += model.engine.profile_read(k)["ms"]
        model.engine.profile_enable(k, False)
    gpu_ms /= psteps

    cdev = dev if backend == "nccl" else torch.device("cpu")
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    frames_all = torch.tensor([frames_rank], dtype=torch.int64, device=cdev)
    if multi:
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
        peak = MFMA_PEAK["fp32" if args.precision == "fp32" else "bf16"]
        traffic, traffic_src = None, None
        if os.path.exists(TRAFFIC_FILE) and args.config == "c2" and args.precision == "bf16" and args.batch == 32:
            try:
                tj = json.load(open(TRAFFIC_FILE))
                traffic = tj.get("conv_gemm_hbm_bytes_per_launch")
                import hashlib
                here = hashlib.sha256(open(os.path.join(ROOT, "lightningfastspeech2_amd", "csrc", "gemm_mfma.hip"), "rb").read()).hexdigest()[:16]
                sha_matches = tj.get("kernel_source_sha256") == here
                traffic_src = (f"profiles/{os.path.basename(TRAFFIC_FILE)}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + --pmc WRITE_SIZE, "
                               "separate passes over this launch shape (tools/pmc_traffic.sh); kernel source at commit "
                               f"{tj.get('kernel_commit', '?')} (git log -1 -- csrc/gemm_mfma.hip), measured at {tj.get('measured_at_commit', '?')}; "
                               f"sha256 of the measured csrc/gemm_mfma.hip {tj.get('kernel_source_sha256', '?')} "
                               + ("== this checkout's" if sha_matches else f"!= this checkout's {here}: the kernel source changed since the counters were read"))
            except Exception:
                traffic = None
        na = max(prof_att["launches"], 1)
        att_s = prof_att["ms"] / na * 1e-3
        att_tf = (prof_att["flops"] / na) / att_s / 1e12 if att_s > 0 else 0.0
        line = {
            "metric": "mel-frames/sec (whole node), 256-phoneme batch-32",
            "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "host": {"nice": os.nice(0), "loadavg_1min": os.getloadavg()[0], "cpus": os.cpu_count()},
            "dtype": args.precision, "data": "synthetic",
            "rtf": elapsed / args.steps / (total_frames * HOP / SR),
            "config": {"workload": f"{args.config}: "
                       + {"c2": "FS2-27M dense k=9 H=256 F=1024 4+4 layers",
                          "c3": "LS-76M depth-wise H=768 F=3072 4+5 layers",
                          "c5": "FS2-1B dense k=9 H=1024 F=4096 12+12 layers"}.get(args.config, args.config)
                       + f", batch {args.batch}/GPU x {args.phones} phonemes, {args.frames_per_phone} frames/phone "
                         f"-> T={T}, random-init weights", "batch_per_gpu": args.batch, "phonemes": args.phones,
                       "frames_per_utterance": T, "global_batch": args.batch * world,
                       "parallelism": f"dp{world} (utterance shards, RCCL all-gather of mels only)" if multi else "single GPU",
                       "params": cfg.param_count()},
            # which roof bounds the dominant launch: its algorithmic intensity (FLOPs / algorithmic bytes, both accounted by the engine)
            # against the machine balance 2.5 PF / 8 TB/s = 312 FLOP/B.  The dense conv (1.8 kFLOP/B) and the K = 768 pointwise GEMM
            # (607) are MFMA-bound; the K = 256 pointwise GEMM of the reference-default model (204: 25 MB in, 101 MB out) is HBM-bound
            "roofline": (lambda ai, bpl: {
                         **({"bound": "mfma", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak}
                            if ai >= peak / HBM_PEAK else
                            {"bound": "hbm", "achieved": bpl / avg_s / 1e9 if avg_s > 0 else 0.0, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": bpl / avg_s / HBM_PEAK if avg_s > 0 else 0.0, "mfma_frac": achieved / peak}),
                         "algorithmic_intensity_flop_per_byte": ai, "bytes_per_launch": bpl,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("gemm_conv_slab_kernel: decoder FFN conv1, implicit-GEMM Conv1d "
                                    f"M={args.batch * T} N={cfg.decoder_conv_filter_size} K={cfg.decoder_kernel_sizes[0]}x{cfg.hidden}")
                         if not cfg.decoder_depthwise_conv else
                         (f"{'gemm_persist_kernel' if cfg.hidden > 256 else 'gemm_wres_kernel'}: decoder FFN pointwise conv1.1 (behind the depth-wise conv) "
                          f"M={args.batch * T} N={cfg.decoder_conv_filter_size} K={cfg.hidden}")})(
                             prof["flops"] / max(prof["bytes"], 1.0), prof["bytes"] / n) | {
                         "launches_timed": prof["launches"], "avg_launch_us": avg_s * 1e6,
                         "flops_per_launch": prof["flops"] / n,
                         "measured": f"HIP events on the launch stream around every launch of this class in {psteps} eager steps of the same "
                                     "workload right behind the timed region (the timed steps replay hipGraphs, which hold no event records)"},
            "north_star_attention": {
                "kernel": f"decoder self-attention (QK^T, softmax, PV fused): {args.batch * cfg.decoder_head} x {T} queries x {T} keys, head dim "
                          f"{cfg.hidden // cfg.decoder_head}; attention_pipe_kernel where it applies (bf16, head dim 128)",
                "achieved": att_tf, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": att_tf * 1e12 / peak,
                "avg_launch_us": att_s * 1e6, "launches_timed": prof_att["launches"], "flops_per_launch": prof_att["flops"] / na,
                "what": "BASELINE.json north_star: >= 40 % of the bf16 MFMA roofline on the attention GEMMs; the encoder's instance "
                        "(256 keys) is ingest-bound (profiles/HISTORY.md §4), this is the MFMA-bound one; HIP events, own pass of eager steps"},
            "encoder_mha_block": (lambda n_, s_: {
                "kernel": "encoder self-attention block = two launches per layer (r06): in-projection GEMM; attn_out_ln_kernel = self-attention of both "
                          "heads + out-projection + residual + LayerNorm (three launches where that kernel does not apply: H != 256, fp32, knob 1340)",
                "achieved": (prof_mha["flops"] / n_) / s_ / 1e12 if s_ > 0 else 0.0, "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": (prof_mha["flops"] / n_) / s_ / peak if s_ > 0 else 0.0, "avg_block_us": s_ * 1e6, "blocks_timed": prof_mha["launches"],
                "flops_per_block": prof_mha["flops"] / n_,
                "ingest_roofline": encoder_mha_ingest_bound(cfg, args.batch, args.phones, s_,
                                                            fused_tail=args.precision == "bf16" and cfg.hidden == 256 and cfg.encoder_head == 2
                                                            and 1340 not in [k["knob"] for k in applied_knobs]),
                "what": "BASELINE.json north_star names the encoder attention; SURVEY 8d: only the fused block (8 B L H^2 + 4 B L^2 H, AI ~ 724) can "
                        "be MFMA-bound.  Two of its three launches are fused (r06, DESIGN.md §4.3): at B x L = 8192 rows the block is neither MFMA- nor "
                        "HBM-bound - `ingest_roofline` prices it against what bounds launches of this size, the bytes each CU pulls through "
                        "L2 -> LDS at the measured per-CU ceiling; `frac` against the MFMA peak is kept for the record; HIP events around the "
                        "block, one forward at a time"})(max(prof_mha["launches"], 1), prof_mha["ms"] / max(prof_mha["launches"], 1) * 1e-3),
            "gpu_ms_per_step": gpu_ms,
            "gpu_ms_per_step_is": f"sum of HIP-event intervals around every kernel launch of a forward, {psteps} eager steps behind the timed region",
            "in_flight": n_pick,
            "ms_per_step_one_in_flight": min((v for k, v in tune.items() if k.endswith("/1")), default=None),
            "in_flight_is": "forwards in flight in the timed region (lightningfastspeech2_amd.model.ForwardPipeline: one engine replica, HIP "
                            "stream and host thread each; a step = one forward over one batch either way); ms_per_step_one_in_flight = the best "
                            "warm-up figure with one synchronous forward at a time",
            "launch_mode": {"timed_region": pick, "warmup_ms_per_step": {k: round(v, 4) for k, v in tune.items()},
                            "graph_replays_total": int(graph_replays),
                            "what": "eager launches or both phases of the forward (encode ~100 launches, decode ~50) replayed as hipGraphs "
                                    "around the one host sync, with 1 or --in-flight forwards in flight; a few untimed steps of each "
                                    "combination (key mode/in_flight) after warm-up, the fastest on this host runs the timed region "
                                    "(FS2_BENCH_MODE / FS2_BENCH_IN_FLIGHT pin it)"},
        }
        if dist_info is not None:
            line["dist"] = dist_info
        if applied_knobs:
            line["knobs_applied"] = applied_knobs  # FS2_GEMM_KNOBS / FS2_XCD_REMAP: an A/B line says which switches it ran under
        if not multi:
            # the boundary takes device pointers; a host caller also pays H2D of phones + speaker and D2H of the
            # fp32 mels + mask per batch (SURVEY 8d's metric definition): timed separately, never `value`
            hp = torch.from_numpy(inp["phones"]).pin_memory()
            hs = torch.from_numpy(inp["speaker"]).pin_memory()
            hmel = torch.empty(args.batch, T, cfg.n_mels, dtype=torch.float32).pin_memory()
            hmask = torch.empty(args.batch, T, dtype=torch.bool).pin_memory()
            k2 = max(3, min(args.steps, 10))

            def pcie_step():
                b = {"phones": hp.to(dev, non_blocking=True), "speaker": hs.to(dev, non_blocking=True)}
                o = model(b, inference=True)
                hmel.copy_(o["mel"], non_blocking=True)
                hmask.copy_(o["tgt_mask"], non_blocking=True)
            pcie_step()
            torch.cuda.synchronize()
            reps1 = []
            for _ in range(3):  # (the fastest of three repeats, like the pipelined leg below)
                t0 = time.perf_counter()
                for _ in range(k2):
                    pcie_step()
                torch.cuda.synchronize()
                reps1.append(time.perf_counter() - t0)
            el2 = min(reps1)
            one = {"value": frames_rank * k2 / el2, "ms_per_step": el2 / k2 * 1e3, "steps": k2, "in_flight": 1,
                   "repeats_ms_per_step": [r / k2 * 1e3 for r in reps1]}
            # the same boundary inside the pipeline (r06): two forwards in flight, inputs copied on the forward's own stream, batch i's
            # mels + mask crossing PCIe on a copy stream under batch i + 1's forward (model.ForwardPipeline(host_outputs=...))
            hb = {"phones": hp, "speaker": hs}
            two = None
            if pin_n != 1:  # (FS2_BENCH_IN_FLIGHT=1 = the rocprof traces: one forward at a time, a launch's duration is its own - no pipelined leg)
                hpipe = model.pipeline(max(2, n_pick), host_outputs=("mel", "tgt_mask"))
                # eager launches here: a hipGraph signature holds every buffer address, and with host batches the device copies of the
                # inputs (and, with copies still in flight, the outputs) are fresh allocations every step - each would be captured anew
                hpipe.set_graphs(False)
                try:
                    for _ in range(4 * len(hpipe.models) + 1):  # every ring slot of every replica gets its pinned buffers (a 15.7 MB pinned allocation takes milliseconds)
                        hpipe.submit(hb)
                    hpipe.drain()
                    torch.cuda.synchronize()
                    # garbage collected INSIDE the timed loop can be an engine replica of an earlier pipeline: fs2_destroy -> hipFree -> a
                    # device-wide stall of tens of ms (measured r06, tools/probes/pcie_pipeline_probe3.py: one 40-90 ms submit in 40)
                    gc.collect()
                    torch.cuda.synchronize()
                    k3 = max(6, min(args.steps, 20))
                    n_dev_alloc = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
                    # three repeats of the k3-step loop, the fastest reported (all three printed): this leg runs five Python threads
                    # (two replicas, two copiers, the submitter) on a host that is not ours alone - one 15 ms scheduling hiccup in a
                    # 45 ms sample is +0.7 ms per step (measured r06: 2.05-2.25 on a quiet host, 2.75-2.79 in two of three suite runs)
                    reps = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        got = 0
                        for _ in range(k3):
                            got += len(hpipe.submit(hb))
                        got += len(hpipe.drain())  # every result's host copy has landed when drain returns
                        reps.append(time.perf_counter() - t0)
                        assert got == k3
                    el3 = min(reps)
                    two = {"value": frames_rank * k3 / el3, "ms_per_step": el3 / k3 * 1e3, "steps": k3, "in_flight": len(hpipe.models),
                           "repeats_ms_per_step": [r / k3 * 1e3 for r in reps],
                           "device_allocations_during": torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n_dev_alloc}
                finally:
                    hpipe.close()
            best = two if two is not None and two["ms_per_step"] <= one["ms_per_step"] else one
            line["value_incl_pcie"] = {**best, "one_at_a_time_ms_per_step": one["ms_per_step"], "pipelined_ms_per_step": two["ms_per_step"] if two else None,
                                       "what": "inputs from / mels + mask to pinned host memory every step (66 KB H2D, "
                                               f"{hmel.numel() * 4 / 1e6:.1f} MB D2H); in_flight > 1: the device-to-host copy of batch i on a "
                                               "copy stream under batch i + 1's forward; timed until the last host copy has landed; each leg = the fastest of "
                                               "three repeats of its loop (repeats_ms_per_step)"}
        if not multi and not args.no_parity:
            try:
                line["parity"] = parity_block(cfg, sd, args, dev, model)
            except Exception as ex:  # never lose the timed line to the checker
                line["parity"] = {"error": repr(ex)}
        if not multi and not args.no_train and args.precision in ("bf16", "fp32"):
            try:  # side measurement, never `value`: the same workload through the training step
                line["training_step"] = training_block(cfg, sd, args, inp, T)
            except Exception as ex:
                line["training_step"] = {"error": repr(ex)}
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, args)
        print(json.dumps(line), flush=True)
    for pp in pipes.values():
        pp.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
