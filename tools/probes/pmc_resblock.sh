#!/bin/bash
# SQ stall counters of the 128-channel 4-wave resblock kernel, for a library variant
LIB=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rb
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS"; do
  rm -rf $OUT; FS2_LIB=$LIB timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_vocoder.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv")) if "resblock_kernel<fs2::bf16, 8, 128, 4, " in r["Kernel_Name"]]
byd=collections.OrderedDict()
for r in rows: byd.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
ds=list(byd.values())
d=ds[-2]  # a k=7/k=11 pair launch of the last pass
print("  ".join(f"{k}={v:.3g}" for k,v in d.items()), f"(n={len(ds)})")
PY
done
