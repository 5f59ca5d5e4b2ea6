for t in pipe old pipe old; do
  if [ $t = old ]; then export SD_ATTN_NO_PIPE=1; else unset SD_ATTN_NO_PIPE; fi
  echo "--- $t"; timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$t.log 2>&1; python - <<PY
import json
r=json.loads(open("gpurun_out/bench_$t.log").read().strip().splitlines()[-1])
print(r["value"], {k:v["ms"] for k,v in r["kernel_breakdown"].items() if k in ("attention","conv3x3","gemm")})
PY
done
