python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k predictor 2>&1 | tail -2
for rep in 1 2; do
for lib in variants/libfs2_pfp0.so libfs2_hip.so; do echo "== $lib"; FS2_LIB=$PWD/lightningfastspeech2_amd/$lib python tools/bench_ops.py pred 2>&1 | grep -E "predictor"; done
done
for rep in 1 2; do
for lib in variants/libfs2_pfp0.so libfs2_hip.so; do for cfg in c2 ref-default; do FS2_LIB=$PWD/lightningfastspeech2_amd/$lib python bench.py --config $cfg --no-train --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $cfg', round(d['ms_per_step'],4), round(d['ms_per_step_one_in_flight'],4))"; done; done
done
