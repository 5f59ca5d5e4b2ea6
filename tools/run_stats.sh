# rocprofv3 kernel-trace stats of the HEADLINE workload only (the summary committed under profiles/): no other_configs,
# no end-to-end leg, no CPU baseline, so conv_halo_kernel<0> shows 44 launches per UNet forward at ~100 us -- the unit the
# bench line's roofline is quoted on (the program itself follows `--`: no env / shell hop under the profiler)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > $R/gpurun_out/stats_bench.log 2>&1 || { grep -v "^    @" $R/gpurun_out/stats_bench.log | tail -5; exit 1; }
tail -1 $R/gpurun_out/stats_bench.log | cut -c1-300
