#!/bin/bash
# tools/prof_classes.sh CFG BATCH TAG : rocprofv3 kernel trace of bench.py on one config, summarised per (kernel, grid) launch class
CFG=${1:-c3}; B=${2:-32}; TAG=${3:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/profc_$CFG
timeout 900 rocprofv3 --kernel-trace --stats -d $O/profc_$CFG -o p -- python $R/bench.py --config $CFG --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-train > $O/${TAG}_${CFG}_classes_bench.json 2> $O/${TAG}_${CFG}_classes.err
DB=$(find $O/profc_$CFG -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py --by-grid $DB "$TAG $CFG B=$B bf16, 13 forwards" > $O/${TAG}_${CFG}_launch_classes.md
sqlite3 $DB "pragma table_info(kernels)" > $O/kernels_cols.txt 2>/dev/null || python -c "
import sqlite3,sys; print([r[1] for r in sqlite3.connect('$DB').execute('pragma table_info(kernels)')])" > $O/kernels_cols.txt
find $O/profc_$CFG -name '*.db' -delete
