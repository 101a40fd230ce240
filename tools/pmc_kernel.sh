#!/bin/bash
# tools/pmc_kernel.sh "<kernel name substring>" -- <command...> : collect SQ counters for one kernel (2 passes)
PAT="$1"; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf $OUT; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc -- "$@" > /dev/null 2>&1
  python3 - "$PAT" <<PY
import csv, collections, sys
rows=[r for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv")) if sys.argv[1] in r["Kernel_Name"]]
acc=collections.defaultdict(list)
for r in rows: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:32s} {v[-1]:16.0f}   (n={len(v)})")
PY
done
