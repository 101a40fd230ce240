R=$PWD; O=$R/gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_boundary.py tests/test_gpu_parity_corners.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_ops.py encmha > $O/r06_v14_encmha_bench_ops.txt 2>&1; tail -3 $O/r06_v14_encmha_bench_ops.txt
for rep in 1 2 3; do for k in 1340 1341; do for cfg in c2 ref-default; do FS2_GEMM_KNOBS=$k python bench.py --config $cfg --no-train --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['encoder_mha_block']; print('knob $k $cfg ms_per_step', round(d['ms_per_step'],4), 'one in flight', round(d['ms_per_step_one_in_flight'],4), 'enc block us', round(m['avg_block_us'],1), 'mfma frac', round(m['frac'],3), 'ingest frac', round(m['ingest_roofline']['frac'],3))"; done; done; done | tee $O/r06_v14_enc_attn_out_ab.txt
