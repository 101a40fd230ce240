#!/bin/bash
# A/B or ablation build of libfs2_hip.so with some translation units replaced:
#   EXTRA="-DFOO" tools/build_variant_files.sh NAME path/to/vocoder_conv.hip [path/to/other.hip ...]
# every given file stands in for the csrc/ source of the same base name; the rest link from csrc/*.o
# -> lightningfastspeech2_amd/variants/libfs2_NAME.so (git-ignored; select with FS2_LIB=...)
set -e
NAME=$1; shift
D=lightningfastspeech2_amd/csrc; V=lightningfastspeech2_amd/variants; O=$V/obj_$NAME; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$D $EXTRA"
ALL=$(for f in $D/*.hip; do basename $f .hip; done)
REPL=""
for f in "$@"; do b=$(basename $f .hip); REPL="$REPL $b"; /opt/rocm/bin/hipcc $FLAGS -c $f -o $O/$b.o & done
wait
OBJS=""
for b in $ALL; do
  if [[ " $REPL " == *" $b "* ]]; then OBJS="$OBJS $O/$b.o"; else
    [ -f $D/$b.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$D -c $D/$b.hip -o $D/$b.o
    OBJS="$OBJS $D/$b.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libfs2_$NAME.so $OBJS
echo built $V/libfs2_$NAME.so
