#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_vocoder.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv")) if "vocoder" in r["Kernel_Name"]]
# last pass: group by dispatch id
byd=collections.OrderedDict()
for r in rows: byd.setdefault(r["Dispatch_Id"],{"k":r["Kernel_Name"][:60],"lds":r.get("LDS_Block_Size","")})[r["Counter_Name"]]=float(r["Counter_Value"])
ds=list(byd.values())[-40:]
for d in ds:
    ia=d.get("SQ_LDS_IDX_ACTIVE",0); bc=d.get("SQ_LDS_BANK_CONFLICT",0)
    print(f"{d['k'][15:50]:36s} lds={d['lds']:>7s} idx_active={ia:12.0f} bank_conflict={bc:12.0f} ({100*bc/max(ia,1):4.1f}%) insts_lds={d.get('SQ_INSTS_LDS',0):10.0f} mfma_busy={d.get('SQ_VALU_MFMA_BUSY_CYCLES',0):12.0f} busy={d.get('SQ_BUSY_CYCLES',0):12.0f} wait_lds={d.get('SQ_WAIT_INST_LDS',0):12.0f} wave_cyc={d.get('SQ_WAVE_CYCLES',0):12.0f}")
PY
