#!/bin/bash
# one build, two environments, interleaved on one box:  tools/env_ab.sh "SD_X=0" "SD_X=1"   (each argument: VAR=value[,VAR=value])
set -o pipefail
Q="--no-cpu-baseline --no-roofline --no-e2e --no-other-configs --no-parity-full-length --steps 3 --warmup 1"
for rep in 1 2 3; do for e in "$@"; do
  echo -n "== $e: "; env $(echo $e | tr ',' ' ') timeout -k 10 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'images/s', round(d['ms_per_step'],1), 'ms')" || exit 1
done; done
