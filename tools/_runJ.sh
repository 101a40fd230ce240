python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv" 2>&1 | tail -2
for tr in 256 128 64; do echo "== FS2_DW_TR=$tr"; FS2_DW_TR=$tr python tools/bench_ops.py rows 2>&1 | grep -E "dwconv"; done
