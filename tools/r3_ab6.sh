#!/bin/bash
# GEGLU epilogue variants (alternate libraries built by hand: lib/libsdhip_g{1,2,3}.so) on the GEMM op benchmark
for l in "" g1 g2 g3 ""; do
  if [ -n "$l" ]; then export SD_AMD_LIB=$PWD/sonicdiffusionbayeslab_amd/lib/libsdhip_$l.so; else unset SD_AMD_LIB; fi
  echo "== lib ${l:-default}"
  timeout -k 10 200 python tools/bench_ops.py --only gemm 2>&1 | grep -v "^==" | awk '{printf "%s | ", $0} END {print ""}' | cut -c1-3000
done
