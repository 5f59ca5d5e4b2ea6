# (SD_GEMM_TUNE exists only in the SD_ABLATE build: the timing ablations run on libsdhip_ablate.so)
export SD_AMD_LIB=sonicdiffusionbayeslab_amd/lib/libsdhip_ablate.so
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -k "gemm or conv3x3 or unet" > gpurun_out/ab_t1.log 2>&1; tail -3 gpurun_out/ab_t1.log
SD_GEMM_BIG=1 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm or conv3x3" > gpurun_out/ab_t2.log 2>&1; tail -1 gpurun_out/ab_t2.log
echo "--- wide stores ON"; timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E "M= 65536|M= 16384|M=  4096"
echo "--- wide stores OFF (tune 64)"; SD_GEMM_TUNE=64 timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E "M= 65536|M= 16384"
