for t in 1 0 1; do echo "--- SD_CONV_HALO=$t"; SD_CONV_HALO=$t timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$t.log 2>&1; python - <<PY
import json
r=json.loads(open("gpurun_out/bench_$t.log").read().strip().splitlines()[-1])
print(r["value"], r["roofline"]["achieved"], {k:v["ms"] for k,v in r["kernel_breakdown"].items()})
PY
done
