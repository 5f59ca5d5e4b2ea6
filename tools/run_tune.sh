for t in 0 16 24; do echo "--- SD_GEMM_TUNE=$t"; SD_GEMM_TUNE=$t timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E "geglu"; done
