for t in 0 2 4 6 8 12; do echo "--- SD_ATTN_STAGGER=$t"; SD_ATTN_STAGGER=$t timeout -k 10 200 python tools/bench_ops.py --only attn 2>&1 | grep -E "Nk= 4096|Nk= 1024"; done
