#!/bin/bash
# tools/voc_ablate.sh NAME[:extra bench args] ... : per-layer times of one vocoder pass for each library variant
# (NAME = lightningfastspeech2_amd/variants/libfs2_NAME.so, "base" = the in-tree library) -> gpurun_out/voc_abl/NAME.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/voc_abl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  n=${spec%%:*}; extra=""; [[ "$spec" == *:* ]] && extra=${spec#*:}
  lib=$ROOT/lightningfastspeech2_amd/variants/libfs2_$n.so; [ "$n" = base ] && lib=$ROOT/lightningfastspeech2_amd/libfs2_hip.so
  tag=$(echo "$spec" | tr ': -' '___')
  rm -rf /tmp/vabl_$tag
  FS2_LIB=$lib timeout 400 rocprofv3 --kernel-trace -d /tmp/vabl_$tag -o t -- python $ROOT/tools/bench_vocoder.py --no-cpu-baseline --steps 3 --warmup 1 $extra > $OUT/$tag.json 2>$OUT/$tag.err
  db=$(find /tmp/vabl_$tag -name "*.db" | head -1)
  python $ROOT/tools/voc_layer_times.py $db $((32*1536)) ${LAYERS_MODE:-fused} > $OUT/$tag.txt 2>&1
  echo "== $spec"; head -1 $OUT/$tag.txt
done
