"""Per-dispatch listing of ONE steady forward out of a rocprofv3 --kernel-trace run of tools/forward_once.py (development).
usage: python tools/fw_trace.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sinusoid" in r["Kernel_Name"]]
fw = rows[idx[-1]:]                       # the last forward
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
tot = collections.OrderedDict()
for r in fw:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = short(r["Kernel_Name"])
    out.write(f"{k:64s} wgs={int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_X'])):6d} x{r['Workgroup_Size_X']:>4s} {us:8.1f} us\n")
    t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += us
span = (int(fw[-1]["End_Timestamp"]) - int(fw[0]["Start_Timestamp"])) / 1e3
print(f"{len(fw)} dispatches, span {span:.1f} us", file=sys.stderr)
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{us:9.1f} us {n:4d} x {us / n:7.1f}  {k}", file=sys.stderr)
