#!/bin/bash
# tools/run_scale.sh [TAG] : the multi-GPU scaling sweep as a one-liner, for a node that has the GPUs (the builder's boxes have one).
#   N = 1, 2, 4, 8 ranks x { c2 (FS2-27M, 32 utterances / GPU = BASELINE configs[1]),
#                            c3 (LS-76M, 32 / GPU: at N = 8 this IS configs[3], batch 256 over 8 GPUs),
#                            c5 (FS2-1B, 8 / GPU: at N = 8 this IS configs[4], batch 64 over 8 GPUs) }
# exactly as the driver launches bench.py: one process per GPU under torch.distributed.run, backend nccl (= RCCL over xGMI), weak
# scaling.  Every line carries a "dist" object (ranks RCCL connected, per-rank ms_per_step with and without the mel all-gather,
# gather bytes per rank, gather_ms_exposed, RCCL version) so that a SCALE record can be checked, not just believed.
# -> gpurun_out/TAG_scale_<cfg>_n<N>.json ; N larger than the visible GPU count is skipped (and said so).
TAG=${1:-scale}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
PORT=29610
for cfg in c2 c3 c5; do
  B=32; [ $cfg = c5 ] && B=8
  for N in 1 2 4 8; do
    if [ $N -gt $NG ]; then echo "skip $cfg N=$N: $NG GPU(s) visible" >&2; continue; fi
    PORT=$((PORT + 1))
    EXTRA="--config $cfg --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-train"
    if [ $N -eq 1 ]; then  # the plain single-GPU line AND the forced multi-rank path at world size 1 (must agree within noise)
      python $R/bench.py --gpus 1 $EXTRA > $O/${TAG}_scale_${cfg}_n1.json 2> $O/${TAG}_scale_${cfg}_n1.err
      FS2_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT \
        $R/bench.py --gpus 1 $EXTRA > $O/${TAG}_scale_${cfg}_n1_forced_dist.json 2> $O/${TAG}_scale_${cfg}_n1_forced_dist.err
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        $R/bench.py --gpus $N $EXTRA > $O/${TAG}_scale_${cfg}_n$N.json 2> $O/${TAG}_scale_${cfg}_n$N.err
    fi
  done
done
python - "$O" "$TAG" <<'PY'
import glob, json, os, sys
O, TAG = sys.argv[1:3]
print("| config | N | ms/step | mel-frames/s (whole job) | vs N=1 x N | ranks seen | gather MB/rank | gather ms exposed |")
print("|---|---|---|---|---|---|---|---|")
for cfg in ("c2", "c3", "c5"):
    base = None
    for f in sorted(glob.glob(os.path.join(O, f"{TAG}_scale_{cfg}_n*.json")), key=lambda p: (len(p), p)):
        try:
            l = json.loads([x for x in open(f).read().splitlines() if x.startswith("{")][-1])
        except Exception:
            continue
        d = l.get("dist") or {}
        n = l["n_gpus"]
        if base is None:
            base = l["value"]
        tag = "1 (forced dist)" if "forced" in f else str(n)
        print(f"| {cfg} | {tag} | {l['ms_per_step']:.3f} | {l['value']:.3e} | {l['value'] / (base * n):.3f} | {d.get('ranks_seen', '-')} | "
              f"{d.get('gather_bytes_per_rank', 0) / 1e6:.1f} | {d.get('gather_ms_exposed', '-')} |")
PY
