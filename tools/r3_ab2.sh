# round-3 A/B batch 2: conv K-loop orders (SD_GEMM_TUNE 128 = round 2, 256 = both partners pipelined, 512 = staggered with
# in-loop branches, 0 = staggered, two straight-line loops) and the fused cross-attention variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv or xattn" > $O/r3_ops2.log 2>&1; echo "ops rc=$?"; tail -2 $O/r3_ops2.log
for t in 128 256 512 0; do SD_GEMM_TUNE=$t timeout -k 10 300 python tools/conv_ab.py > $O/r3_conv_t$t.txt 2>&1; done
paste $O/r3_conv_t128.txt $O/r3_conv_t256.txt $O/r3_conv_t512.txt $O/r3_conv_t0.txt | cut -f1,2,5,8,11,3,12 | grep -v amdgpu
for t in 128 0; do SD_GEMM_TUNE=$t timeout -k 10 300 python tools/conv_ab.py > $O/r3_conv_b_t$t.txt 2>&1; done
paste $O/r3_conv_b_t128.txt $O/r3_conv_b_t0.txt | cut -f1,2,5 | tail -1
for v in 0 7 15; do for shp in "16 4096 320" "16 1024 640"; do echo "variant $v shape $shp: $(SD_XATTN_VARIANT=$v timeout -k 10 120 python tools/xattn_stamps.py $shp 2>&1 | grep -E 'per launch|phase 1' | tr '\n' ' ')"; done; done
