# tools/r06_evidence.sh : the gpurun command behind profiles/r06_v9_* (kernel stats, timelines, bench lines for c2 / c3 / c5 / ref-default)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/prof_configs.sh r06_v9 > /dev/null 2>&1
for c in c2 c3 ref-default; do bash tools/forward_timeline.sh $c; mv $O/timeline_${c}_eager.md $O/r06_v9_timeline_${c}_eager.md; done
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --config ref-default --no-train --no-cpu-baseline --no-parity > $O/r06_v9_refdefault_bench.json 2> /dev/null
rm -rf $O/prof_rd
FS2_BENCH_IN_FLIGHT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_rd -o p -- python $R/bench.py --config ref-default --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-train > $O/r06_v9_refdefault_prof_bench.json 2> /dev/null
DB=$(find $O/prof_rd -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py $DB "r06 v9 ref-default: rocprofv3 --kernel-trace --stats -- python bench.py --config ref-default --steps 10 --warmup 3, FS2_BENCH_IN_FLIGHT=1 (bf16)" > $O/r06_v9_refdefault_kernel_stats.md
find $O -name '*.db' -delete
cd $R; python bench.py --no-train > $O/r06_v9_c2_bench_full.json 2> $O/r06_v9_c2_bench_full.err
python -c "
import json
d=json.loads(open('gpurun_out/r06_v9_c2_bench_full.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['value_incl_pcie']['ms_per_step'], d['parity']['decision_safe'])"
head -9 $O/r06_v9_c2_kernel_stats.md | cut -c1-150
