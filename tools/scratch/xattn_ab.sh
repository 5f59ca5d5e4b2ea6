#!/bin/bash
# persistent vs per-block launch of the fused cross-attention, one box
set -o pipefail
O=gpurun_out; mkdir -p $O
for cfg in "64 4096 320" "16 4096 320" "16 1024 640" "32 4096 320"; do
  for pz in 0 1 0 1; do
    echo "== $cfg persist=$pz"; SD_XATTN_PERSIST=$pz timeout -k 10 120 python tools/xattn_stamps.py $cfg 2>&1 | grep -E "per launch|8-wave|total" || exit 1
  done
done
