set -x
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_boundary.py tests/test_abi_cpu.py -x -q -m "gpu or not gpu" > $O/r06_v0_boundary_pytest.txt 2>&1; tail -3 $O/r06_v0_boundary_pytest.txt
python bench.py --no-train --no-cpu-baseline > $O/r06_v0_c2_bench.json 2> $O/r06_v0_c2_bench.err; tail -c 1500 $O/r06_v0_c2_bench.json
python bench.py --config ref-default --no-train --no-cpu-baseline --no-parity > $O/r06_v0_refdefault_bench.json 2> $O/r06_v0_refdefault_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_rd
FS2_BENCH_IN_FLIGHT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_rd -o p -- python $R/bench.py --config ref-default --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-train > $O/r06_v0_refdefault_prof_bench.json 2> $O/r06_v0_refdefault_prof.err
DB=$(find $O/prof_rd -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py $DB "r06 v0 ref-default: rocprofv3 --kernel-trace --stats -- python bench.py --config ref-default --steps 10 --warmup 3, FS2_BENCH_IN_FLIGHT=1 (bf16)" > $O/r06_v0_refdefault_kernel_stats.md
find $O/prof_rd -name '*.db' -delete
cd $R
bash tools/forward_timeline.sh ref-default; mv $O/timeline_ref-default_eager.md $O/r06_v0_timeline_refdefault_eager.md
python -c "
import json
for f in ['r06_v0_c2_bench.json','r06_v0_refdefault_bench.json']:
    d=json.loads(open('gpurun_out/'+f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d.get('value_incl_pcie'), d.get('ms_per_step_one_in_flight'))
"
