# (SD_GEMM_TUNE exists only in the SD_ABLATE build: the timing ablations run on libsdhip_ablate.so)
export SD_AMD_LIB=sonicdiffusionbayeslab_amd/lib/libsdhip_ablate.so
for t in 0 16 24; do echo "--- SD_GEMM_TUNE=$t"; SD_GEMM_TUNE=$t timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E "geglu"; done
