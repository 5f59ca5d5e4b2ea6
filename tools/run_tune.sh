for t in 0 4 8 12; do echo "--- SD_GEMM_TUNE=$t"; SD_GEMM_TUNE=$t timeout -k 10 200 python tools/bench_ops.py --only conv 2>&1 | grep -E "^conv res= 32  1920|^conv res= 64   320->  320 s1"; done
