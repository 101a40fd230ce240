R=$PWD; OUT=$R/gpurun_out/pmc_t2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT; timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/bench_ops.py gemm --variant 0 --only "dec conv2" --reps 3 > /dev/null 2>&1
  python3 - "$c" "$OUT/pmc_counter_collection.csv" <<'PY'
import csv, sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[2])) if "gemm_conv_slab_kernel" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[1]]
print(sys.argv[1], "KiB per launch", sum(v[-3:])/3, "-> MB", sum(v[-3:])/3*1024/1e6*(2 if sys.argv[1]=="FETCH_SIZE" else 1))
PY
done
