"""Reduce `rocprofv3 --pmc FETCH_SIZE WRITE_SIZE` output to HBM bytes per launch for one kernel.

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE
reports half of the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section).  The first half of the
launches (the warm-up forward) is dropped.  Usage: pmc_traffic.py <dir> <kernel substring> <out.json>"""
import collections, csv, glob, json, sys
d, pat, out = sys.argv[1], sys.argv[2], sys.argv[3]
vals = collections.defaultdict(list)                  # counter -> values in dispatch order (one pass per counter)
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    for c, m in per.items():
        vals[c] = [m[i] for i in sorted(m)]
def mean2(v):
    v = v[len(v) // 2:]
    return sum(v) / len(v)
ids = vals["FETCH_SIZE"][len(vals["FETCH_SIZE"]) // 2:]
fetch, write = mean2(vals["FETCH_SIZE"]), mean2(vals["WRITE_SIZE"])
res = {"kernel": pat, "launches": len(ids), "FETCH_SIZE_KiB_mean": fetch, "WRITE_SIZE_KiB_mean": write,
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
       "note": "2x correction on FETCH_SIZE for 16-B-per-lane reads on gfx950; separate --pmc pass over tools/forward_once.py"}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16
res["kernel_sources_sha16"] = kernel_sources_sha16()          # bench.py refuses this profile once the kernels change
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
