#!/bin/bash
# tools/voc_evidence.sh TAG : HiFi-GAN bench line (un-profiled), per-layer times and kernel stats of one profiled run
# -> gpurun_out/TAG_vocoder_{bench.json,layers.txt,kernel_stats.md}
TAG=${1:-r06_v20}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out; mkdir -p $O
python $ROOT/tools/bench_vocoder.py --steps 10 --warmup 3 > $O/${TAG}_vocoder_bench.json 2> $O/${TAG}_vocoder_bench.err
tail -c 600 $O/${TAG}_vocoder_bench.json
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/vev
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/vev -o t -- python $ROOT/tools/bench_vocoder.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_vocoder_prof.err
db=$(find /tmp/vev -name "*.db" | head -1)
python $ROOT/tools/voc_layer_times.py $db $((32*1536)) fused > $O/${TAG}_vocoder_layers.txt 2>&1
python $ROOT/tools/rocpd_stats.py $db "r06 vocoder: rocprofv3 --kernel-trace --stats -- python tools/bench_vocoder.py --steps 2 --warmup 1 --no-cpu-baseline (3 passes, HiFi-GAN V1, 32 x 1536 frames, bf16)" > $O/${TAG}_vocoder_kernel_stats.md 2>&1
head -3 $O/${TAG}_vocoder_layers.txt
