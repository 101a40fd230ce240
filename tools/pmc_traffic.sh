#!/bin/bash
# tools/pmc_traffic.sh : HBM-side traffic of the decoder conv1 launch (bench.py's roofline.traffic).
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/bench_ops.py gemm --only "dec conv1";
# FETCH_SIZE is doubled (gfx950 counts 128-byte requests at 64 B, MI355X_MICROARCH.md HBM note); both are KiB.
# Writes gpurun_out/pmc_traffic.json (copy to profiles/rNN_pmc_traffic.json).  The numbers are keyed to the kernel source by
# COMPUTATION: the sha256 of csrc/gemm_mfma.hip as it lies on the measuring box, plus the commits __graft_entry__.build() recorded
# in lightningfastspeech2_amd/_build_info.json (the GPU box has no .git); bench.py compares the sha with its own copy of the source.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_t; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT; timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/bench_ops.py gemm --variant 0 --only "dec conv1" --reps 3 > /dev/null 2>&1
  python3 - "$c" "$OUT/pmc_counter_collection.csv" <<'PY' > $R/gpurun_out/pmc_$c.txt
import csv, sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[2])) if "gemm_conv_slab_kernel" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[1]]
print(sum(v[-3:])/3 if len(v)>=3 else v[-1])
PY
done
python3 - <<PY
import json, hashlib, os
src="$R/lightningfastspeech2_amd/csrc/gemm_mfma.hip"
sha=hashlib.sha256(open(src,"rb").read()).hexdigest()[:16]
try: bi=json.load(open("$R/lightningfastspeech2_amd/_build_info.json"))
except Exception: bi={}
f=float(open("$R/gpurun_out/pmc_FETCH_SIZE.txt").read()); w=float(open("$R/gpurun_out/pmc_WRITE_SIZE.txt").read())
d={"kernel":"gemm_conv_slab_kernel<bf16,bf16,8,false> decoder conv1 (M=49152,N=1024,K=2304)","FETCH_SIZE_KiB_per_launch":f,"WRITE_SIZE_KiB_per_launch":w,
   "fetch_bytes_corrected_x2":f*1024*2,"write_bytes":w*1024,"conv_gemm_hbm_bytes_per_launch":f*1024*2+w*1024,"algorithmic_bytes_per_launch":130547712,
   "kernel_source_sha256":sha,"kernel_commit":bi.get("kernel_commits",{}).get("gemm_mfma.hip","unknown"),"measured_at_commit":bi.get("head","unknown"),
   "source_dirty_at_build":bi.get("dirty"),
   "note":"tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/bench_ops.py gemm --only 'dec conv1' (mean of the 3 timed launches); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); L2<->fabric traffic, Infinity-Cache hits included"}
json.dump(d,open("$R/gpurun_out/pmc_traffic.json","w"),indent=1); print(d)
PY
