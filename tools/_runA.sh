R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=15 > $O/r06_v6_pytest.txt 2>&1; tail -25 $O/r06_v6_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_v6_smoke.txt 2>&1; tail -2 $O/r06_v6_smoke.txt
python bench.py > $O/r06_v6_c2_bench_full.json 2> $O/r06_v6_c2_bench_full.err; python -c "
import json
d=json.loads(open('gpurun_out/r06_v6_c2_bench_full.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['value_incl_pcie'], d['cpu_baseline']['value'])"
