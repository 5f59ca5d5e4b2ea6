#!/bin/bash
# PMC passes over the 64x64 self-attention op alone (tools/bench_ops.py --only attn0), per SD_ATTN_VARIANT given as arguments
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
ST="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
for v in "$@"; do export SD_ATTN_VARIANT=$v
  rm -rf $O/pmc_attn_v$v $O/pmc_attn_s_v$v
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_attn_v$v -- python $R/tools/bench_ops.py --only attn0 > $O/pmc_attn_v$v.log 2>&1 || { tail -5 $O/pmc_attn_v$v.log; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ST --output-format csv -d $O/pmc_attn_s_v$v -- python $R/tools/bench_ops.py --only attn0 > $O/pmc_attn_s_v$v.log 2>&1 || { tail -5 $O/pmc_attn_s_v$v.log; exit 1; }
  echo "== variant $v"
  (cd $R && python tools/pmc_forward_summary.py gpurun_out/pmc_attn_v$v gpurun_out/r3_attn_pmc_v$v.json | grep -i "pipe40" | cut -c1-220
   python tools/pmc_forward_summary.py gpurun_out/pmc_attn_s_v$v gpurun_out/r3_attn_stall_v$v.json | grep -i "pipe40\|fractions" | cut -c1-260)
done
