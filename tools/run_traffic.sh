# HBM traffic of the dominant kernel (conv_halo_kernel, the 3x3 conv) per launch: one --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic/$c -- python $R/tools/forward_once.py > $R/gpurun_out/pmc_traffic_$c.log 2>&1 || { grep -v "^    @" $R/gpurun_out/pmc_traffic_$c.log | tail -5; exit 1; }
done
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_traffic "conv_halo_kernel" gpurun_out/conv_traffic.json
