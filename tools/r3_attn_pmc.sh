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
# before / after rows in one file (copied to profiles/round3_attn_pmc.json by hand)
cd $R && python - "$@" <<'PY'
import json, sys
from bench import kernel_sources_sha16
out = {"kernel": "attn_pipe40_kernel (64x64 self-attention: N = Nk = 4096, d = 40, 8 heads, UNet batch 16)",
       "command": "tools/r3_attn_pmc.sh 0 7 (two rocprofv3 --kernel-trace --pmc passes per variant over tools/bench_ops.py --only attn0)",
       "kernel_sources_sha16": kernel_sources_sha16(), "variants": {}}
for v in sys.argv[1:]:
    a = json.load(open(f"gpurun_out/r3_attn_pmc_v{v}.json")); s = json.load(open(f"gpurun_out/r3_attn_stall_v{v}.json"))
    ka = [k for k in a if "pipe40" in k][0]; ks = [k for k in s if "pipe40" in k][0]
    row = dict(a[ka]); row["stall_pass"] = s[ks]
    row["tflops_profiled"] = 4.0 * 16 * 8 * 4096 * 4096 * 40 / (row["avg_us_profiled"] * 1e-6) / 1e12
    out["variants"]["SD_ATTN_VARIANT=" + v] = row
json.dump(out, open("gpurun_out/r3_attn_pmc.json", "w"), indent=1)
for k, r in out["variants"].items():
    print(k, "us", round(r["avg_us_profiled"], 1), "TFLOP/s", round(r["tflops_profiled"], 1), "mfma_busy", round(r["mfma_busy_frac"], 3))
PY
