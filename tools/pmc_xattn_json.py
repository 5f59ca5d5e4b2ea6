"""profiles/roundN_xattn_pmc.json from the SQ counter passes of tools/run_profiles_r3.sh (one pass per kernel variant on the
SAME box: before / after rows): MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) on the 0.3 ms
dispatch (64 samples x 4096 tokens, C = 320).  usage: pmc_xattn_json.py out.json tag=dir [tag=dir ...]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16

out = {"kernel": "xattn_fused_kernel", "kernel_sources_sha16": kernel_sources_sha16(),
       "shape": "B=64 samples x 4096 tokens, C=320 (2048 workgroups; the 64x64-level launch of the bench has 512)",
       "command": "tools/run_profiles_r3.sh xattn (rocprofv3 --kernel-trace --pmc ..., target tools/xattn_stamps.py 64 4096 320)",
       "variants": {}}
for arg in sys.argv[2:]:
    tag, d = arg.split("=", 1)
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "xattn_fused" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "xattn_fused" in r["Kernel_Name"]]
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    row = dict(m)
    row.update({"launches": len(dur), "duration_us_profiled_mean": sum(dur) / max(len(dur), 1), "shader_cycles": cyc,
                "clock_GHz": cyc / (sum(dur) / max(len(dur), 1) * 1e3) if dur else None,
                "mfma_busy_frac": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else None})
    out["variants"][tag] = row
json.dump(out, open(sys.argv[1], "w"), indent=1)
for t, r in out["variants"].items():
    print(t, "mfma_busy", round(r["mfma_busy_frac"], 4), "us", round(r["duration_us_profiled_mean"], 1), "clock", round(r["clock_GHz"], 3))
