python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv" 2>&1 | tail -1
for cfg in c2 ref-default c3; do python bench.py --config $cfg --no-train --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$cfg', round(d['ms_per_step'],4), r['bound'], round(r['achieved'],1), r['unit'], round(r['frac'],3), round(r['algorithmic_intensity_flop_per_byte']), r.get('mfma_frac'), r['kernel'][:40])"; done
