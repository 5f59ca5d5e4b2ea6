#!/bin/bash
# two builds of the library on one box, interleaved:  tools/lib_ab.sh base.so new.so   (SD_AMD_LIB selects the build)
set -o pipefail
Q="--no-cpu-baseline --no-roofline --no-e2e --no-other-configs --steps 3 --warmup 1"
for rep in 1 2; do for l in "$@"; do
  echo "== $l"; SD_AMD_LIB=$PWD/$l timeout -k 10 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'images/s', round(d['ms_per_step'],1), 'ms')" || exit 1
done; done
for l in "$@"; do echo "== op times $l"; SD_AMD_LIB=$PWD/$l timeout -k 10 200 python tools/op_times.py 4 2>&1 | grep -E "per kind" ; done
