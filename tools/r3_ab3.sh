R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for rep in 1 2; do for lib in libsdhip_base.so libsdhip.so; do echo "== $lib: $(SD_AMD_LIB=$R/sonicdiffusionbayeslab_amd/lib/$lib timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E '^gemm' | head -12 | sed -E 's/.* ([0-9.]+) us .*/\1/' | tr '\n' ' ')"; done; done
S=$(date +%s); timeout -k 10 580 python bench.py > $O/r3_bench3.log 2> $O/r3_bench3.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"; tail -c 300 $O/r3_bench3.log
