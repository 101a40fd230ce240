#!/bin/bash
# tools/forward_timeline.sh [c2|c3|c5] : the launch sequence of one eager forward (kernel, grid, duration, idle gap before it)
# -> gpurun_out/timeline_<cfg>.md
cfg=${1:-c2}; B=32; [ $cfg = c5 ] && B=8
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl
FS2_BENCH_MODE=${FS2_TL_MODE:-eager} timeout 600 rocprofv3 --kernel-trace -d $O/prof_tl -o p -- python $R/bench.py --config $cfg --batch $B --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-train > /dev/null 2> $O/timeline_$cfg.err
DB=$(find $O/prof_tl -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py --timeline $DB "$cfg eager forward" > $O/timeline_${cfg}_${FS2_TL_MODE:-eager}.md
find $O/prof_tl -name '*.db' -delete
