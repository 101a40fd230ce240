#!/bin/bash
# tools/prof_configs.sh TAG : bench.py + rocprofv3 --kernel-trace --stats for the BASELINE configs that fit one GPU
# (C2 FS2-27M B=32, C3 LS-76M B=32, C5 FS2-1B B=8/GPU).  Writes gpurun_out/TAG_{c2,c3,c5}_{bench.json,kernel_stats.md}.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c3 c5; do
  B=32; [ $cfg = c5 ] && B=8
  EXTRA="--no-cpu-baseline --no-parity --no-train"
  timeout 600 python $R/bench.py --config $cfg --batch $B --steps 10 --warmup 3 $EXTRA > $O/${TAG}_${cfg}_bench.json 2> $O/${TAG}_${cfg}_bench.err
  rm -rf $O/prof_$cfg
  # the trace runs with ONE forward in flight (FS2_BENCH_IN_FLIGHT=1): with two, launches of the two forwards share the CUs and a
  # launch's wall duration stops being its own; the bench line above it is the default (two in flight where that is faster)
  FS2_BENCH_IN_FLIGHT=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o p -- python $R/bench.py --config $cfg --batch $B --steps 10 --warmup 3 $EXTRA > $O/${TAG}_${cfg}_prof_bench.json 2> $O/${TAG}_${cfg}_prof.err
  DB=$(find $O/prof_$cfg -name '*results.db' | head -1)
  python $R/tools/rocpd_stats.py $DB "$TAG $cfg: rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg --batch $B --steps 10 --warmup 3, FS2_BENCH_IN_FLIGHT=1 (bf16)" > $O/${TAG}_${cfg}_kernel_stats.md
  find $O/prof_$cfg -name '*.db' -delete
done
