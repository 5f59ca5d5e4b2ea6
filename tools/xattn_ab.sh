#!/bin/bash
# variants of the fused cross-attention, one box:  tools/xattn_ab.sh "31 0"
set -o pipefail
O=gpurun_out; mkdir -p $O
for cfg in "64 4096 320" "16 4096 320" "16 1024 640"; do
  for rep in 1 2; do for v in ${1:-31 0}; do
    echo "== $cfg variant=$v"; SD_XATTN_VARIANT=$v timeout -k 10 120 python tools/xattn_stamps.py $cfg 2>&1 | grep -E "per launch|8-wave|total" || exit 1
  done; done
done
