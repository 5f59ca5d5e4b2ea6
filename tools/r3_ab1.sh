# round-3 A/B batch 1: correctness of the pipelined conv / xattn variants, then timings (one process per variant)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv or xattn" > $O/r3_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/r3_ops.log
timeout -k 10 300 python tools/conv_ab.py > $O/r3_conv_pipe.txt 2>&1 && SD_GEMM_TUNE=128 timeout -k 10 300 python tools/conv_ab.py > $O/r3_conv_base.txt 2>&1
paste $O/r3_conv_base.txt $O/r3_conv_pipe.txt | cut -f1,2,3,5,6
for v in 0 4 5 7; do for shp in "16 4096 320" "16 1024 640"; do echo "variant $v shape $shp: $(SD_XATTN_VARIANT=$v timeout -k 10 120 python tools/xattn_stamps.py $shp 2>&1 | grep -E 'per launch|phase 1' | tr '\n' ' ')"; done; done
timeout -k 10 600 python -m pytest tests/test_benchshapes_gpu.py tests/test_fp8_gpu.py -q -s -m gpu -k "batch_consistency or calibration or lcm_loop_fp8" > $O/r3_t2.log 2>&1; echo "t2 rc=$?"; grep -E "rel-L2|oracle|passed|failed|FAILED|Error|calibration" $O/r3_t2.log | tail -30
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-other-configs --no-e2e > $O/r3_bench2.log 2>&1; echo "bench rc=$?"; tail -c 3000 $O/r3_bench2.log
