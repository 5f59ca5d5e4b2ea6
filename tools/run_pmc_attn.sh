R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for mode in pipe old; do
  if [ $mode = old ]; then export SD_ATTN_NO_PIPE=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_attn_$mode -- python $R/tools/bench_ops.py --only attn0 > $R/gpurun_out/pmc_attn_$mode.log 2>&1
  echo "== $mode"; python $R/tools/pmc_summary.py $R/gpurun_out/pmc_attn_$mode attn_
done
