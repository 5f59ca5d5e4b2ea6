"""HBM traffic per launch of several kernels from `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` passes over
tools/forward_once.py (one pass per counter: the two do not fit one pass on gfx950), LAST forward only.

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE reports half of the bytes of
16-B-per-lane reads on gfx950 (MI355X_MICROARCH.md, HBM section).  GB/s = those bytes / the kernel-trace duration of the
same dispatches (profiled passes run at a lower clock than the bench: a rate, not a record).
usage: pmc_traffic_multi.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <out.json> <kernel substring> [...]"""
import collections, csv, glob, json, os, re, sys
d, out, pats = sys.argv[1], sys.argv[2], sys.argv[3:]


def load(counter):
    vals, names, times = {}, {}, {}
    for f in glob.glob(f"{d}/{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                i = int(r["Dispatch_Id"])
                vals[i] = vals.get(i, 0.0) + float(r["Counter_Value"]); names[i] = r["Kernel_Name"]
    for f in glob.glob(f"{d}/{counter}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            times[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(vals)
    starts = [i for i in ids if "sinusoid_kernel" in names[i]]
    last = [i for i in ids if i >= starts[-1]] if starts else ids
    return vals, names, times, last


fv, fn, ft, fl = load("FETCH_SIZE")
wv, wn, wt, wl = load("WRITE_SIZE")
res = {"note": "per launch, last forward of tools/forward_once.py (UNet batch 16); hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 "
               "correction for 16-B-per-lane reads; us = kernel-trace duration under the counters", "kernels": {}}
clean = lambda n: re.sub(r"^void ", "", re.sub(r"\([^()]*\)$", "", n).replace("(anonymous namespace)::", ""))
for pat in pats:
    fi = [i for i in fl if pat in clean(fn[i])]
    wi = [i for i in wl if pat in clean(wn[i])]
    if not fi or len(fi) != len(wi):
        res["kernels"][pat] = {"error": f"{len(fi)} launches in the FETCH pass, {len(wi)} in the WRITE pass"}
        continue
    f = sum(fv[i] for i in fi) / len(fi); w = sum(wv[i] for i in wi) / len(wi)
    us = sum(ft[i] for i in fi if i in ft) / max(1, sum(1 for i in fi if i in ft))
    b = (2 * f + w) * 1024
    res["kernels"][pat] = {"launches": len(fi), "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "hbm_bytes_per_launch": b,
                           "avg_us_profiled": us, "hbm_GBps_profiled": b / (us * 1e-6) / 1e9 if us else None,
                           "frac_of_8TBps": b / (us * 1e-6) / 8e12 if us else None}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16
res["kernel_sources_sha16"] = kernel_sources_sha16()
# the keys bench.py reads for the dominant kernel (conv_traffic_bytes)
first = res["kernels"].get(pats[0], {})
res.update({"kernel": pats[0], "launches": first.get("launches"), "hbm_bytes_per_launch": first.get("hbm_bytes_per_launch"),
            "FETCH_SIZE_KiB_mean": first.get("FETCH_SIZE_KiB_mean"), "WRITE_SIZE_KiB_mean": first.get("WRITE_SIZE_KiB_mean")})
json.dump(res, open(out, "w"), indent=1)
for k, v in res["kernels"].items():
    print(k[:60], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
