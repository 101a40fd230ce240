#!/bin/bash
# A/B of the single-launch predictor's tile heights under rocprofv3 (kernel time only; bench_ops adds the weight-pack launches)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
for v in 1300 1302; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/tools/bench_ops.py pred --variant $v > /dev/null 2>&1
  echo "variant $v"; python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pp/**/p_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "predictor_" in r["Name"]: print("  ", r["Name"][:60], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
PY
done
