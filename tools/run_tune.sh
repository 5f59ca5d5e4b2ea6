for t in 0 128 256 384; do echo "--- SD_GEMM_TUNE=$t"; SD_GEMM_TUNE=$t timeout -k 10 200 python tools/bench_ops.py --only conv0 2>&1 | grep -E "conv res"; done
