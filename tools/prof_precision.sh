#!/bin/bash
# tools/prof_precision.sh TAG [precisions...] : rocprofv3 --kernel-trace --stats of bench.py (C2) in the parity-grade precision modes
# -> gpurun_out/TAG_<precision>_kernel_stats.md
TAG=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pr in ${@:-fp32x3 mixed3}; do
  rm -rf $O/prof_$pr
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$pr -o p -- python $R/bench.py --precision $pr --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-train > $O/${TAG}_${pr}_prof_bench.json 2> $O/${TAG}_${pr}_prof.err
  DB=$(find $O/prof_$pr -name '*results.db' | head -1)
  python $R/tools/rocpd_stats.py $DB "$TAG c2 $pr: rocprofv3 --kernel-trace --stats -- python bench.py --precision $pr --steps 10 --warmup 3" > $O/${TAG}_${pr}_kernel_stats.md
  find $O/prof_$pr -name '*.db' -delete
done
