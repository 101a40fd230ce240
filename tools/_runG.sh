for v in 0 1 2 4 6; do echo "== FS2_PF_PROBE=$v"; FS2_LIB=$PWD/lightningfastspeech2_amd/variants/libfs2_pfp$v.so python tools/bench_ops.py pred 2>&1 | grep -E "variance predictor|dw variance"; done
