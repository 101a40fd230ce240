for v in dw128w3 dw128w5; do echo "== $v"; FS2_LIB=$PWD/lightningfastspeech2_amd/variants/libfs2_$v.so python tools/bench_ops.py rows 2>&1 | grep -E "T rows"; done
