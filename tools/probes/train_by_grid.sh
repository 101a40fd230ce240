R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tbg
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tbg -o p -- python $R/tools/bench_train.py --precision bf16 --steps 2 --warmup 1 > /dev/null 2>&1
DB=$(find /tmp/tbg -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py --by-grid $DB "training step c2 bf16, launch classes" > $R/gpurun_out/r06_v24_train_c2_by_grid.md
python $R/tools/rocpd_stats.py --timeline $DB "training step c2 bf16" > $R/gpurun_out/r06_v24_train_c2_timeline.md 2>/dev/null
head -45 $R/gpurun_out/r06_v24_train_c2_by_grid.md | cut -c1-160
