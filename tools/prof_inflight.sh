#!/bin/bash
# tools/prof_inflight.sh TAG [cfg] : GPU-busy fraction of bench.py with one forward at a time and with two in flight
# (rocprofv3 --kernel-trace, eager launches; tools/busy_union.py) -> gpurun_out/TAG_inflight_<cfg>.md
TAG=${1:-r04}; CFG=${2:-c2}; B=32; [ $CFG = c5 ] && B=8
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_inflight_$CFG.md
for n in 1 2; do
  rm -rf $O/prof_if
  FS2_BENCH_IN_FLIGHT=$n FS2_BENCH_MODE=eager timeout 900 rocprofv3 --kernel-trace -d $O/prof_if -o p -- python $R/bench.py --config $CFG --batch $B --steps 40 --warmup 3 --no-cpu-baseline --no-parity --no-train > $O/${TAG}_inflight_${CFG}_n$n.json 2> /dev/null
  DB=$(find $O/prof_if -name '*results.db' | head -1)
  python $R/tools/busy_union.py $DB --title "$CFG, $n forward(s) in flight (bench.py --steps 40, eager, under rocprofv3 --kernel-trace)" >> $O/${TAG}_inflight_$CFG.md
  find $O/prof_if -name '*.db' -delete
done
cat $O/${TAG}_inflight_$CFG.md
