# tools/gpu_suite.sh : what the driver runs at round end - the full -m gpu suite, smoke(), the default bench line (profiles/r06_v15_*)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=6 > $O/r06_v12_pytest.txt 2>&1; tail -12 $O/r06_v12_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_v12_smoke.txt 2>&1; tail -1 $O/r06_v12_smoke.txt
python bench.py > $O/r06_v12_c2_bench_full.json 2> $O/r06_v12_c2_bench_full.err; python -c "
import json
d=json.loads(open('gpurun_out/r06_v12_c2_bench_full.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['value_incl_pcie']['ms_per_step'], d['value_incl_pcie']['in_flight'], d['cpu_baseline']['value'], d['training_step'].get('ms_per_step'))"
