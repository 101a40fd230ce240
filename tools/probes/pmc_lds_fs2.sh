#!/bin/bash
# LDS bank-conflict share per kernel of one bench.py step (FS2 mel forward)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_fs2
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv"))]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_LDS_IDX_ACTIVE": cnt[k]+=1
for k,d in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES",0)):
    ia=d.get("SQ_LDS_IDX_ACTIVE",0); bc=d.get("SQ_LDS_BANK_CONFLICT",0)
    print(f"{k:70s} n={cnt[k]:4d} lds_active={ia:14.0f} conflict={100*bc/max(ia,1):5.1f}% mfma_busy/4={d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/4:14.0f} lds/mfma={ia/max(d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/4,1):5.2f}")
PY
