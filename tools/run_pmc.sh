# usage: run_pmc.sh <tag> <bench_ops --only arg> [env assignments...]
tag=$1; only=$2; shift 2
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python $R/tools/bench_ops.py --only $only > $R/gpurun_out/pmc_${tag}_a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python $R/tools/bench_ops.py --only $only > $R/gpurun_out/pmc_${tag}_b.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_${tag}_c -- python $R/tools/bench_ops.py --only $only > $R/gpurun_out/pmc_${tag}_c.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_a $KPAT; python tools/pmc_summary.py gpurun_out/pmc_${tag}_b $KPAT; python tools/pmc_summary.py gpurun_out/pmc_${tag}_c $KPAT
