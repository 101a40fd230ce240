#!/bin/bash
# tools/pmc_kernels.sh [config] : per-kernel MFMA utilisation and HBM-side traffic of one bench.py forward (north_star:
# "evidenced by rocprof HBM GB/s and MFMA utilisation against gfx950 peak").  Three separate rocprofv3 --pmc passes
# (SQ set, FETCH_SIZE, WRITE_SIZE - MI355X_MICROARCH.md: they do not fit one pass) + a plain --kernel-trace pass for the
# un-profiled durations.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B).  Writes gpurun_out/${ROUND:-r03}_pmc_kernels_<cfg>.md
CFG=${1:-c2}; B=32; [ $CFG = c5 ] && B=8
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_ks; mkdir -p $O
export FS2_BENCH_IN_FLIGHT=1 FS2_BENCH_MODE=eager   # one forward at a time, eager: a launch's counters and duration are its own
CMD="python $R/bench.py --config $CFG --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-train"
[ -n "$PMC_CMD" ] && CMD="$PMC_CMD"   # any other command (e.g. tools/bench_ops.py bwd --only ...); CFG then only names the output
cd /tmp && export TMPDIR=/tmp
rm -rf $O/*
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/s -o p -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > /dev/null 2>&1
python3 - "$O" "$CFG" <<'PY' > $R/gpurun_out/${ROUND:-r03}_pmc_kernels_$CFG.md
import csv, collections, glob, re, sys
O, cfg = sys.argv[1], sys.argv[2]
def rd(sub, name):
    f = glob.glob(f"{O}/{sub}/**/*{name}.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
short = lambda k: re.sub(r"fs2::", "", re.sub(r"^void ", "", k))[:74]
dur = collections.defaultdict(list)
for r in rd("t", "kernel_trace"):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for sub in ("s", "f", "w"):
    for r in rd(sub, "counter_collection"):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_INSTS_MFMA", "FETCH_SIZE", "WRITE_SIZE"): n[(short(r["Kernel_Name"]), r["Counter_Name"])] += 1
print(f"# per-kernel PMC summary, {cfg} bf16 (tools/pmc_kernels.sh)\n")
print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of SIMD cycles the matrix pipe is busy (in the profiled pass's own clock). "
      "waits = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / barriers), issue stalls = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES. "
      "HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; Infinity-Cache hits included), per launch; GB/s over the un-profiled launch duration.\n")
print("| kernel | launches | avg us | MFMA busy | VALU instr / MFMA instr | waits | issue stalls | fetch MB (x2) | write MB | GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
tot = sum(sum(v) for v in dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 0.004 * tot: continue
    d = cnt.get(k, {}); m = d.get("SQ_INSTS_MFMA", 0); va = d.get("SQ_INSTS_VALU", 0) - m
    mb, cu, wc = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), d.get("SQ_BUSY_CU_CYCLES", 0), d.get("SQ_WAVE_CYCLES", 0)
    nf, nw = max(n[(k, "FETCH_SIZE")], 1), max(n[(k, "WRITE_SIZE")], 1)
    fe, wr = 2 * d.get("FETCH_SIZE", 0) * 1024 / nf, d.get("WRITE_SIZE", 0) * 1024 / nw
    avg = sum(v) / len(v)
    gbs = (fe + wr) / (avg * 1e-6) / 1e9
    print(f"| `{k}` | {len(v)} | {avg:.1f} | {mb / max(4 * cu, 1):.2f} | {va / m if m else float('nan'):.2f} | {d.get('SQ_WAIT_ANY', 0) / max(wc, 1):.2f} | "
          f"{d.get('SQ_WAIT_INST_ANY', 0) / max(wc, 1):.2f} | {fe / 1e6:.1f} | {wr / 1e6:.1f} | {gbs:.0f} | {gbs / 80:.1f} |")
PY
rm -rf $O
