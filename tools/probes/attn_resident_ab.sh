cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1210 1211; do
rm -rf /tmp/p_$v
rocprofv3 --kernel-trace --stats -d /tmp/p_$v -o p -- python $R/tools/bench_ops.py attn --only custom --shape ${SHAPE:-32,256,256,2} --variant $v --reps 30 > /dev/null 2>&1
DB=$(find /tmp/p_$v -name '*results.db' | head -1)
echo "variant $v"; python $R/tools/rocpd_stats.py $DB x | grep "attention_kernel"
done
