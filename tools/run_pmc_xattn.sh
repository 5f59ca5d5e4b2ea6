# Counters of the fused cross-attention kernel on a dispatch long enough for a reliable clock estimate (B = 64 samples at the
# 64x64 level: ~0.3 ms; GRBM_GUI_ACTIVE / 8 / time reads high on shorter dispatches, MI355X_MICROARCH.md 'DVFS give-back'),
# plus its L2 hit rate and HBM fetch at the bench shape.  usage: tools/run_pmc_xattn.sh
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_xattn_sq -- python $R/tools/xattn_stamps.py 64 4096 320 > $R/gpurun_out/pmc_xattn_sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_xattn_l2 -- python $R/tools/xattn_stamps.py 16 4096 320 > $R/gpurun_out/pmc_xattn_l2.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_xattn_fetch -- python $R/tools/xattn_stamps.py 16 4096 320 > $R/gpurun_out/pmc_xattn_fetch.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_xattn_sq xattn_fused; python tools/pmc_summary.py gpurun_out/pmc_xattn_l2 xattn_fused; python tools/pmc_summary.py gpurun_out/pmc_xattn_fetch xattn_fused
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_xattn_sq/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "xattn_fused" in r["Kernel_Name"]]
print("xattn_fused_kernel durations under the SQ pass (us):", [round(x, 1) for x in d])
PY
grep "per launch\|8-wave" gpurun_out/pmc_xattn_sq.log
