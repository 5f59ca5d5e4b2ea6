#!/bin/bash
# the sub-pixel upsampler convs on the halo kernel's 4-tap mode vs the implicit-GEMM kernel, one box, interleaved
set -o pipefail
Q="--no-cpu-baseline --no-roofline --no-e2e --no-other-configs --steps 3 --warmup 1"
for rep in 1 2; do for f in 0 1; do
  echo "== SD_SUBPIX_HALO=$f"; SD_SUBPIX_HALO=$f timeout -k 10 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'images/s', round(d['ms_per_step'],1), 'ms')" || exit 1
done; done
for f in 0 1; do echo "== op times SD_SUBPIX_HALO=$f"; SD_SUBPIX_HALO=$f timeout -k 10 200 python tools/op_times.py 4 2>&1 | grep -E "per kind| 5120 | 2560 " ; done
