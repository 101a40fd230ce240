#!/bin/bash
# VALU vs MFMA issue load per kernel of one bench.py step: on a SIMD, non-MFMA VALU instructions (x4 cycles per
# wave64 instruction) and MFMA passes mostly add up instead of overlapping, so VALU cycles next to
# MFMA-busy cycles show what the epilogues / address arithmetic cost.  $1 = bench command (default bench.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu
CMD=${1:-"python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT -o pmc -- $CMD > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv"))]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows: agg[r["Kernel_Name"][:64]][r["Counter_Name"]]+=float(r["Counter_Value"])
print(f"{'kernel':64s} {'valu/mfma instr':>16s} {'valu cyc / mfma cyc':>20s} {'mfma busy / cu busy*4':>22s}")
for k,d in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES",0))[:12]:
    m=d.get("SQ_INSTS_MFMA",0); v=d.get("SQ_INSTS_VALU",0)-m; mb=d.get("SQ_VALU_MFMA_BUSY_CYCLES",0); cu=d.get("SQ_BUSY_CU_CYCLES",0)
    if m<=0: continue
    print(f"{k:64s} {v/m:16.2f} {4*v/max(mb,1):20.2f} {mb/max(4*cu,1):22.2f}")
PY
