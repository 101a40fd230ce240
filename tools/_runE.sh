R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/pmc_traffic.sh > $O/r06_pmc_traffic.log 2>&1; cp $O/pmc_traffic.json $O/r06_pmc_traffic.json; cat $O/r06_pmc_traffic.json | head -12
ROUND=r06 bash tools/pmc_kernels.sh c2 > /dev/null 2>&1
ROUND=r06 bash tools/pmc_kernels.sh ref-default > /dev/null 2>&1
bash tools/prof_configs.sh r06_v8 > /dev/null 2>&1
for c in c2 c3 ref-default; do bash tools/forward_timeline.sh $c; mv $O/timeline_${c}_eager.md $O/r06_v8_timeline_${c}_eager.md; done
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --config ref-default --no-train --no-cpu-baseline --no-parity > $O/r06_v8_refdefault_bench.json 2> /dev/null
rm -rf $O/prof_rd
FS2_BENCH_IN_FLIGHT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_rd -o p -- python $R/bench.py --config ref-default --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-train > $O/r06_v8_refdefault_prof_bench.json 2> /dev/null
DB=$(find $O/prof_rd -name '*results.db' | head -1)
python $R/tools/rocpd_stats.py $DB "r06 v8 ref-default: rocprofv3 --kernel-trace --stats -- python bench.py --config ref-default --steps 10 --warmup 3, FS2_BENCH_IN_FLIGHT=1 (bf16)" > $O/r06_v8_refdefault_kernel_stats.md
find $O -name '*.db' -delete
ls $O | grep r06_v8; head -12 $O/r06_v8_refdefault_kernel_stats.md
