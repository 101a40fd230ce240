#!/bin/bash
# Build libfs2_hip.so variants for A/B runs: tools/build_variant.sh NAME path/to/attention.hip [path/to/gemm_mfma.hip]
# -> gpurun_out is not shipped; variants go to lightningfastspeech2_amd/variants/libfs2_NAME.so (git-ignored *.so)
set -e
NAME=$1; ATT=${2:-lightningfastspeech2_amd/csrc/attention.hip}; GEMM=${3:-lightningfastspeech2_amd/csrc/gemm_mfma.hip}
D=lightningfastspeech2_amd/csrc; V=lightningfastspeech2_amd/variants; mkdir -p $V/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$D $EXTRA"   # EXTRA="-DFOO" for ablation builds
/opt/rocm/bin/hipcc $FLAGS -c $ATT -o $V/obj_$NAME/attention.o &
/opt/rocm/bin/hipcc $FLAGS -c $GEMM -o $V/obj_$NAME/gemm_mfma.o &
wait
make -s -C $D -j6   # every other object as the in-tree library has it
REST=$(ls $D/*.o | grep -v -e /attention.o -e /gemm_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libfs2_$NAME.so $V/obj_$NAME/attention.o $V/obj_$NAME/gemm_mfma.o $REST
echo built $V/libfs2_$NAME.so
