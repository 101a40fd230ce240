#!/bin/bash
# tools/probes/pmc_attn_pipe.sh [probe bits ...] : SQ counters of the pipelined attention kernel (probe builds of attention_pipe.hip)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_k
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_WAVE32_LDS SQ_INSTS_BRANCH"; do
  rm -rf $OUT; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/probes/attn_pipe_probe.py 2 "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, collections, glob
f = glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "attention_pipe" in r["Kernel_Name"]] if f else []
acc=collections.defaultdict(list)
for r in rows: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:32s} {v[-1]:16.0f}   (n={len(v)})")
PY
done
