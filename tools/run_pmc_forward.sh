# MFMA-busy / VALU / LDS counters of every kernel of a UNet forward (tools/forward_once.py: UNet batch 16, 64x64), one
# `rocprofv3 --pmc` pass (8 SQ slots + GRBM; no trace domains besides --kernel-trace), summarised per kernel.
# usage: tools/run_pmc_forward.sh <tag> [env assignments...]     -> gpurun_out/pmc_fwd_<tag>.json
tag=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $R/gpurun_out/pmc_fwd_$tag -- python $R/tools/forward_once.py > $R/gpurun_out/pmc_fwd_$tag.log 2>&1 \
  || { grep -v "^    @" $R/gpurun_out/pmc_fwd_$tag.log | tail -5; exit 1; }
cd $R
python tools/pmc_forward_summary.py gpurun_out/pmc_fwd_$tag gpurun_out/pmc_fwd_$tag.json
