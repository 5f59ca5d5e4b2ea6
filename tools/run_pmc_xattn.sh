# L2 hit rate / HBM fetch of the fused cross-attention kernel at the 64x64 level (tools/xattn_stamps.py as the target)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_xattn_l2 -- python $R/tools/xattn_stamps.py 16 4096 320 > $R/gpurun_out/pmc_xattn_l2.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_xattn_fetch -- python $R/tools/xattn_stamps.py 16 4096 320 > $R/gpurun_out/pmc_xattn_fetch.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_xattn_l2 xattn_fused; python tools/pmc_summary.py gpurun_out/pmc_xattn_fetch xattn_fused
