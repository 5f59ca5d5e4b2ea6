for t in 0 2; do echo "--- SD_GEMM_BIG=$t"; SD_GEMM_BIG=$t timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E "^gemm"; done
