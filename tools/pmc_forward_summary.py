"""Per-kernel summary of a `rocprofv3 --kernel-trace --pmc ...` pass over tools/forward_once.py (second forward only:
the first one also pays first-touch costs).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the
dispatch), shader cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md 'DVFS give-back')."""
import collections, csv, glob, json, re, sys

d, out = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    e = disp.setdefault(k, {"name": r["Kernel_Name"], "c": {}})
    e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
starts = [i for i in ids if "sinusoid_kernel" in disp[i]["name"]]   # first launch of every forward
half = [i for i in ids if i >= starts[-1]] if starts else ids       # the LAST forward only
agg = collections.OrderedDict()
for i in half:
    e = disp[i]
    name = re.sub(r"\([^()]*\)$", "", e["name"]).replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    a = agg.setdefault(name, {"launches": 0, "c": collections.defaultdict(float)})
    a["launches"] += 1
    for k, v in e["c"].items():
        a["c"][k] += v
res = {}
for name, a in agg.items():
    c = a["c"]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    res[name] = {"launches": a["launches"], **{k: v / a["launches"] for k, v in c.items()},
                 "shader_cycles_per_launch": cyc / a["launches"],
                 "mfma_busy_frac": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)) if cyc else None,
                 "lds_conflict_frac": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None}
json.dump(res, open(out, "w"), indent=1)
import os
times = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        times[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name, a in agg.items():
    ds = [times[i] for i in half if i in times and re.sub(r"^void ", "", re.sub(r"\([^()]*\)$", "", disp[i]["name"]).replace("(anonymous namespace)::", "")) == name]
    if ds:
        res[name]["avg_us_profiled"] = sum(ds) / len(ds)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16
res["kernel_sources_sha16"] = kernel_sources_sha16()          # bench.py refuses this profile once the kernels change
json.dump(res, open(out, "w"), indent=1)
res.pop("kernel_sources_sha16")
for name, v in sorted(res.items(), key=lambda kv: -(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES") or 0) * kv[1]["launches"]):
    mb = v["mfma_busy_frac"]
    print(f"{name[:70]:70s} n={v['launches']:3d} us={v.get('avg_us_profiled', 0):8.1f} mfma_busy={mb if mb is None else round(mb, 3)} "
          f"valu_insts={v.get('SQ_INSTS_VALU', 0):.3g} lds_conf={v['lds_conflict_frac']}")

# stall pass (tools/run_profiles_r3.sh stall): where the waves' cycles go, as fractions of SQ_WAVE_CYCLES
if any("SQ_WAIT_INST_ANY" in v for v in res.values() if isinstance(v, dict)):
    print("\nfractions of SQ_WAVE_CYCLES: parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY, of which LDS) | issuing (ACTIVE_INST_ANY: LDS, VMEM, MISC)")
    for name, v in sorted(((n, x) for n, x in res.items() if isinstance(x, dict)), key=lambda kv: -(kv[1].get("SQ_WAVE_CYCLES") or 0) * kv[1]["launches"]):
        w = v.get("SQ_WAVE_CYCLES") or 0
        if not w:
            continue
        f = lambda k: round((v.get(k) or 0) / w, 3)
        print(f"{name[:64]:64s} n={v['launches']:3d} us={v.get('avg_us_profiled', 0):7.1f} parked={f('SQ_WAIT_ANY')} stalled={f('SQ_WAIT_INST_ANY')} "
              f"(lds {f('SQ_WAIT_INST_LDS')}) issuing={f('SQ_ACTIVE_INST_ANY')} (lds {f('SQ_ACTIVE_INST_LDS')} vmem {f('SQ_ACTIVE_INST_VMEM')} misc {f('SQ_ACTIVE_INST_MISC')})")
