"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel-name substring."""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
    for k, v in sorted(agg.items()):
        print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
    if agg:
        print("vgpr/agpr/lds/grid/wg:", meta)
