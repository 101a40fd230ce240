python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "encoder_attention_out or test_attention" 2>&1 | tail -4
python tools/bench_ops.py encmha 2>&1 | tail -3
python -m pytest tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -3
for k in 1340 1341 1340 1341; do for cfg in c2 ref-default; do FS2_GEMM_KNOBS=$k python bench.py --config $cfg --no-train --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['encoder_mha_block']; print('$k $cfg', round(d['ms_per_step'],4), round(d['ms_per_step_one_in_flight'],4), round(m['avg_block_us'],1), round(m['frac'],3), round(m['ingest_roofline']['frac'],3))"; done; done
