R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "predictor" > $O/r06_v1_pred_pytest.txt 2>&1; tail -5 $O/r06_v1_pred_pytest.txt
python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py -x -q -m gpu > $O/r06_v1_fwd_pytest.txt 2>&1; tail -5 $O/r06_v1_fwd_pytest.txt
python bench.py --config ref-default --no-train --no-cpu-baseline --no-parity > $O/r06_v1_refdefault_bench.json 2> $O/r06_v1_refdefault_bench.err
python bench.py --config ref-default --no-train --no-cpu-baseline --no-parity --no-fused-predictor > $O/r06_v1_refdefault_bench_unfused.json 2> $O/r06_v1_refdefault_bench_unfused.err
python tools/bench_ops.py pred > $O/r06_v1_bench_ops_pred.txt 2>&1; tail -5 $O/r06_v1_bench_ops_pred.txt
python -c "
import json
for f in ['r06_v1_refdefault_bench.json','r06_v1_refdefault_bench_unfused.json']:
    d=json.loads(open('gpurun_out/'+f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d.get('ms_per_step_one_in_flight'), {k:d['value_incl_pcie'][k] for k in ('ms_per_step','one_at_a_time_ms_per_step','pipelined_ms_per_step')})
"
