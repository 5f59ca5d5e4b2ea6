#!/bin/bash
# A/B of the round-3 self-attention variants (SD_ATTN_VARIANT) and the xattn softmax variant (SD_XATTN_VARIANT) on one box
set -o pipefail
mkdir -p gpurun_out
for v in 0 1 3 7; do
  echo "== SD_ATTN_VARIANT=$v"
  SD_ATTN_VARIANT=$v timeout -k 10 200 python -m pytest tests/test_ops_gpu.py -q -x -k "attention" 2>&1 | tail -2 || exit 1
  SD_ATTN_VARIANT=$v timeout -k 10 120 python tools/bench_ops.py --only attn0 2>&1 | grep "attn N" || exit 1
done
for v in 15 31; do
  echo "== SD_XATTN_VARIANT=$v"
  SD_XATTN_VARIANT=$v timeout -k 10 200 python -m pytest tests/test_ops_gpu.py -q -x -k "xattn" 2>&1 | tail -2 || exit 1
  SD_XATTN_VARIANT=$v timeout -k 10 120 python tools/xattn_stamps.py 2>&1 | grep -E "us per launch|8-wave" || exit 1
  SD_XATTN_VARIANT=$v timeout -k 10 120 python tools/xattn_stamps.py 16 1024 640 2>&1 | grep -E "us per launch" || exit 1
done
