# Round-5 profile passes (each its own rocprofv3 run: --pmc only with --kernel-trace; the program itself after `--`).
# usage: tools/run_profiles_r5.sh [stats] [fwd] [stall] [traffic] [xa] [xaln] [attn]      outputs: gpurun_out/r5_*; copy to profiles/round5_*
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
fail() { grep -v "^    @" $1 | tail -5; exit 1; }
for step in "$@"; do case $step in
stats)   # HEADLINE workload only: no other_configs, no end-to-end, no CPU baseline, no batch-1 full-length parity loop
  rm -rf $O/r5_stats
  ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-parity-full-length"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_stats -- python $R/bench.py $ARGS > $O/r5_stats_bench.log 2>&1 || fail $O/r5_stats_bench.log
  F=$(find $O/r5_stats -name "*kernel_stats.csv"); N=$(echo "$F" | wc -l)
  [ "$N" = "1" ] || { echo "expected ONE kernel_stats.csv (one profiled process), found $N: $F"; exit 1; }
  (cd $R && SHA=$(python -c "from bench import kernel_sources_sha16; print(kernel_sources_sha16())") &&
   { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (headline workload only: 2 x 50 forwards at UNet batch 16 + 2 profiled forwards)"; echo "# kernel_sources_sha16=$SHA  source=$(basename $F)"; cat $F; } > $O/r5_bench_kernel_stats.csv)
  grep '^{"metric"' $O/r5_stats_bench.log | tail -1 > $O/r5_bench_line_under_rocprof.json; cut -c1-200 $O/r5_bench_line_under_rocprof.json; head -8 $O/r5_bench_kernel_stats.csv | cut -c1-160 ;;
fwd)
  rm -rf $O/r5_pmc_fwd
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/r5_pmc_fwd -- python $R/tools/forward_once.py > $O/r5_pmc_fwd.log 2>&1 || fail $O/r5_pmc_fwd.log
  (cd $R && python tools/pmc_forward_summary.py gpurun_out/r5_pmc_fwd gpurun_out/r5_pmc_forward.json | cut -c1-200) ;;
stall)
  rm -rf $O/r5_pmc_fwds
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/r5_pmc_fwds -- python $R/tools/forward_once.py > $O/r5_pmc_fwds.log 2>&1 || fail $O/r5_pmc_fwds.log
  (cd $R && python tools/pmc_forward_summary.py gpurun_out/r5_pmc_fwds gpurun_out/r5_pmc_forward_stall.json | cut -c1-200) ;;
traffic)   # HBM bytes of the dominant kernel AND of the HBM-bound ones (north_star: "rocprof HBM GB/s")
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/r5_pmc_traffic/$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r5_pmc_traffic/$c -- python $R/tools/forward_once.py > $O/r5_pmc_traffic_$c.log 2>&1 || fail $O/r5_pmc_traffic_$c.log; done
  (cd $R && python tools/pmc_traffic_multi.py gpurun_out/r5_pmc_traffic gpurun_out/r5_conv_traffic.json "conv_halo_kernel<0, 0, 8, 0>" "conv_halo_kernel<0, 0, 8, 1>" \
     "gn_apply_kernel" "gn_slab_kernel" "gn_finalize_stats_kernel" "gemm_lean_kernel<128, 160, 2, 2, 4, false>" "gemm_lean_kernel<256, 256, 4, 2, 2, true>" \
     "gemm_lean_kernel<128, 160, 2, 2, 0, false>" "attn_pipe40_kernel" "xattn_fused_kernel" "splitk_reduce_kernel" | cut -c1-260) ;;
xa)      # fused cross-attention: the 64-sample launch of rounds 2-4, the bench's own 16-sample launch, and LONG dispatches (512 / 2048
         # samples: 2.6 / 10 ms) -- GRBM_GUI_ACTIVE / 8 / time reads high on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS
         # give-back), and the busy fraction is a quotient over that clock
  for n in 16 64 512 2048; do rm -rf $O/r5_pmc_xattn_b$n
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/r5_pmc_xattn_b$n -- python $R/tools/xattn_stamps.py $n 4096 320 > $O/r5_pmc_xattn_b$n.log 2>&1 || fail $O/r5_pmc_xattn_b$n.log
    grep -E "per launch|8-wave|launch span" $O/r5_pmc_xattn_b$n.log; done
  (cd $R && python tools/pmc_xattn_json.py gpurun_out/r5_xattn_pmc.json launch_16_samples_bench=gpurun_out/r5_pmc_xattn_b16 launch_64_samples=gpurun_out/r5_pmc_xattn_b64 \
     launch_512_samples=gpurun_out/r5_pmc_xattn_b512 launch_2048_samples=gpurun_out/r5_pmc_xattn_b2048) ;;
xaln)    # the same launches of the PRODUCT variant of round 5 (norm2 folded in, X == R, row partials of Y written)
  for n in 16 64 512; do rm -rf $O/r5_pmc_xattnln_b$n
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/r5_pmc_xattnln_b$n -- python $R/tools/xattn_stamps.py $n 4096 320 ln > $O/r5_pmc_xattnln_b$n.log 2>&1 || fail $O/r5_pmc_xattnln_b$n.log
    grep -E "per launch" $O/r5_pmc_xattnln_b$n.log; done
  (cd $R && python tools/pmc_xattn_json.py gpurun_out/r5_xattn_ln_pmc.json launch_16_samples_bench=gpurun_out/r5_pmc_xattnln_b16 launch_64_samples=gpurun_out/r5_pmc_xattnln_b64 \
     launch_512_samples=gpurun_out/r5_pmc_xattnln_b512) ;;
attn)    # before / after rows: the round-4 work order (SD_ATTN_XCD=0: every (sample, head) read by all 8 L2s), then the product
  export SD_ATTN_XCD=0; bash $R/tools/r3_attn_pmc.sh 7 && cp $O/r3_attn_pmc.json $O/r5_attn_pmc_before.json || exit 1; unset SD_ATTN_XCD
  bash $R/tools/r3_attn_pmc.sh 7 || exit 1
  (cd $R && python -c "
import json
a=json.load(open('gpurun_out/r3_attn_pmc.json')); b=json.load(open('gpurun_out/r5_attn_pmc_before.json'))
a['command']='tools/run_profiles_r5.sh attn (tools/r3_attn_pmc.sh 7 with SD_ATTN_XCD=0, then without: two rocprofv3 --kernel-trace --pmc passes each over tools/bench_ops.py --only attn0)'
a['variants']={'round-4 work order (SD_ATTN_XCD=0)': list(b['variants'].values())[0], 'XCD-aware work order (product)': list(a['variants'].values())[0]}
json.dump(a, open('gpurun_out/r5_attn_pmc.json','w'), indent=1)
for k,r in a['variants'].items(): print(k, 'us', round(r['avg_us_profiled'],1), 'TFLOP/s', round(r['tflops_profiled'],1), 'mfma_busy', round(r['mfma_busy_frac'],3))
") ;;
esac; done
