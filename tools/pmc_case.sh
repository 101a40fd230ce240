#!/bin/bash
# tools/pmc_case.sh WHAT "CASE" KERNEL_SUBSTRING : FETCH_SIZE (x2, gfx950) and WRITE_SIZE per launch of one tools/bench_ops.py case
# (separate rocprofv3 --pmc passes, mean of the 3 timed launches) -> stdout.  e.g. tools/pmc_case.sh c3gemm "c3 pw1" gemm_persist_kernel
WHAT=$1; CASE=$2; KER=$3
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_c; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT; timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/bench_ops.py $WHAT --variant 0 --only "$CASE" --reps 3 > /dev/null 2>&1
  python3 - "$c" "$OUT/pmc_counter_collection.csv" "$KER" "$CASE" <<'PY'
import csv, sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[2])) if sys.argv[3] in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[1]]
x=sum(v[-3:])/3 if len(v)>=3 else (v[-1] if v else float("nan"))
print(f"{sys.argv[4]:16s} {sys.argv[1]:10s} {x*1024*(2 if sys.argv[1]=='FETCH_SIZE' else 1)/1e6:9.1f} MB per launch" + ("  (KiB counter x2: gfx950 counts 128-byte requests at 64)" if sys.argv[1]=='FETCH_SIZE' else ""))
PY
done
