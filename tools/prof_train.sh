#!/bin/bash
# tools/prof_train.sh CFG BATCH TAG [extra bench_train args]: bench_train.py + rocprofv3 kernel trace, summarised per kernel
CFG=${1:-c2}; B=${2:-32}; TAG=${3:-r02}; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/tools/bench_train.py --config $CFG --batch $B "$@" > $O/${TAG}_train_${CFG}_bench.json 2> $O/${TAG}_train_${CFG}_bench.err
rm -rf $O/proft_$CFG
timeout 900 rocprofv3 --kernel-trace --stats -d $O/proft_$CFG -o p -- python $R/tools/bench_train.py --config $CFG --batch $B --steps 2 --warmup 1 "$@" > /dev/null 2> $O/${TAG}_train_${CFG}_prof.err
DB=$(find $O/proft_$CFG -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py $DB "$TAG training step $CFG B=$B $* (warm-up + timed steps of the command; forward + loss + backward + AdamW)" > $O/${TAG}_train_${CFG}_kernel_stats.md
find $O/proft_$CFG -name '*.db' -delete
