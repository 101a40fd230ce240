#!/bin/bash
# tools/pmc_lds.sh "<command>" TAG : LDS-side counters (bank conflicts, LDS instruction / wait shares) per kernel of a command
CMD="$1"; TAG=${2:-lds}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_lds; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/a -o p -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/b -o p -- $CMD > /dev/null 2>&1
python3 - "$O" <<'PY' > $R/gpurun_out/${TAG}_pmc_lds.md
import csv, collections, glob, sys
O = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("a", "b"):
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
print("| kernel | LDS bank conflict / LDS active | LDS insts / MFMA | VALU / MFMA | VMEM rd / MFMA | SALU / MFMA | wait LDS / wave cyc | wait any / wave cyc | MFMA busy |")
print("|---|---|---|---|---|---|---|---|---|")
for k, d in sorted(cnt.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    m = max(d.get("SQ_INSTS_MFMA", 0), 1)
    print(f"| `{k}` | {d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):.2f} | {d.get('SQ_INSTS_LDS',0)/m:.2f} | {(d.get('SQ_INSTS_VALU',0)-m)/m:.2f} | "
          f"{d.get('SQ_INSTS_VMEM_RD',0)/m:.2f} | {d.get('SQ_INSTS_SALU',0)/m:.2f} | {d.get('SQ_WAIT_INST_LDS',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.2f} | "
          f"{d.get('SQ_WAIT_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.2f} | {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(4*d.get('SQ_BUSY_CU_CYCLES',1),1):.2f} |")
PY
rm -rf $O
