R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_fp8_gpu.py -q -x -m gpu > $O/r3_ops4.log 2>&1; echo "tests rc=$?"; tail -2 $O/r3_ops4.log
for rep in 1 2; do for lib in libsdhip_base.so libsdhip.so; do echo "== $lib gemm: $(SD_AMD_LIB=$R/sonicdiffusionbayeslab_amd/lib/$lib timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -E '^gemm' | sed -E 's/.* ([0-9.]+) us .*/\1/' | tr '\n' ' ')"; done; done
for lib in libsdhip_base.so libsdhip.so; do echo "== $lib conv: $(SD_AMD_LIB=$R/sonicdiffusionbayeslab_amd/lib/$lib timeout -k 10 200 python tools/bench_ops.py --only conv,subpix 2>&1 | grep -E '^conv|^subpixel' | sed -E 's/.* ([0-9.]+) us .*/\1/' | tr '\n' ' ')"; done
for rep in 1 2; do for lib in libsdhip_base.so libsdhip.so; do SD_AMD_LIB=$R/sonicdiffusionbayeslab_amd/lib/$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-other-configs --no-e2e --no-cpu-baseline > $O/r3_b4_$lib.$rep.log 2>&1; echo "$lib: $(python - <<PY
import json
l=[x for x in open('$O/r3_b4_$lib.$rep.log') if x.startswith('{')][-1]; d=json.loads(l)
print(round(d['value'],3), {k:v['ms'] for k,v in d['kernel_breakdown'].items() if k in ('gemm','conv3x3','conv3x3_gemm','attention','groupnorm','xattn_fused')})
PY
)"; done; done
