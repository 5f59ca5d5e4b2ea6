# per-dispatch kernel durations of one UNet forward (second of two), no event overhead
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python $R/tools/forward_once.py > $R/gpurun_out/trace.log 2>&1 || { grep -v "^    @" $R/gpurun_out/trace.log | tail -5; exit 1; }
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("conv_halo_kernel", "splitk_reduce", "gn_stats", "gn_finalize", "gn_apply", "layernorm", "attn_dma_kernel<40>", "attn_dma_kernel<80>", "attn_kernel<160", "gemv", "conv_in", "conv_out", "sinusoid"):
        if k in n: return k
    if "gemm_kernel" in n:
        return n[n.index("gemm_kernel"):n.index(">")+1]
    return n[:40]
# second forward = second half of our kernels
ours = [r for r in rows if "anonymous namespace" in r["Kernel_Name"] and "at::" not in r["Kernel_Name"]]
half = ours[len(ours)//2:]
t0, t1 = int(half[0]["Start_Timestamp"]), int(half[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in half:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print(f"forward wall {(t1-t0)/1e6:.3f} ms, sum of kernels {tot/1e3:.3f} ms, {len(half)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} n={v[0]:4d} total={v[1]/1e3:7.3f} ms avg={v[1]/v[0]:8.1f} us")
print("conv_halo dispatches in order (us):", [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1) for r in half if "conv_halo" in r["Kernel_Name"]])
PY
