# Round-4 profile passes (each its own rocprofv3 run: --pmc only with --kernel-trace; the program itself after `--`).
# usage: tools/run_profiles_r4.sh [xattn] [attn] [fwd] [stall] [traffic] [stats]     outputs under gpurun_out/, summaries copied to profiles/ by hand
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
fail() { grep -v "^    @" $1 | tail -5; exit 1; }
for step in "$@"; do case $step in
xattn)
  for v in 31; do export SD_XATTN_VARIANT=$v; rm -rf $O/pmc_xattn_v$v
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_xattn_v$v -- python $R/tools/xattn_stamps.py 64 4096 320 > $O/pmc_xattn_v$v.log 2>&1 || fail $O/pmc_xattn_v$v.log
    grep -E "per launch|8-wave" $O/pmc_xattn_v$v.log; done; unset SD_XATTN_VARIANT
  # the same counters on the BENCH's own launch (16 samples x 4096 tokens: 512 workgroups, ~68 us dispatches -- the clock
  # quotient reads high on dispatches this short, MI355X_MICROARCH.md "DVFS give-back")
  rm -rf $O/pmc_xattn_b16
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_xattn_b16 -- python $R/tools/xattn_stamps.py 16 4096 320 > $O/pmc_xattn_b16.log 2>&1 || fail $O/pmc_xattn_b16.log
  (cd $R && python tools/pmc_xattn_json.py gpurun_out/r4_xattn_pmc.json launch_64_samples=gpurun_out/pmc_xattn_v31 launch_16_samples_bench=gpurun_out/pmc_xattn_b16) ;;
fwd)
  rm -rf $O/pmc_fwd_r4
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_fwd_r4 -- python $R/tools/forward_once.py > $O/pmc_fwd_r4.log 2>&1 || fail $O/pmc_fwd_r4.log
  (cd $R && python tools/pmc_forward_summary.py gpurun_out/pmc_fwd_r4 gpurun_out/r4_pmc_forward.json) ;;
stall)   # where the waves' cycles go: parked (WAIT_ANY), issue-stalled (WAIT_INST_ANY, of which LDS), executing
  rm -rf $O/pmc_fwd_r4s
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fwd_r4s -- python $R/tools/forward_once.py > $O/pmc_fwd_r4s.log 2>&1 || fail $O/pmc_fwd_r4s.log
  (cd $R && python tools/pmc_forward_summary.py gpurun_out/pmc_fwd_r4s gpurun_out/r4_pmc_forward_stall.json | cut -c1-200) ;;
traffic)
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_traffic/$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_traffic/$c -- python $R/tools/forward_once.py > $O/pmc_traffic_$c.log 2>&1 || fail $O/pmc_traffic_$c.log; done
  (cd $R && python tools/pmc_traffic.py gpurun_out/pmc_traffic "conv_halo_kernel<0, 0, 8, 0>" gpurun_out/r4_conv_traffic.json) ;;
attn)    # 64x64 self-attention alone: round-2 kernel vs the round-3 one (SD_ATTN_VARIANT 0 / 7)
  bash $R/tools/r3_attn_pmc.sh 0 7 ;;
stats)
  rm -rf $O/stats
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > $O/stats_bench.log 2>&1 || fail $O/stats_bench.log
  tail -1 $O/stats_bench.log | cut -c1-300; find $O/stats -name "*kernel_stats.csv" | head -2 ;;
esac; done
