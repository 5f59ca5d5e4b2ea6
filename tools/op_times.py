"""Per-launch times of one UNet forward at the bench shape (sd_unet_forward_op_times: one hipEvent between consecutive
launches), grouped by op kind and shape.  Development tool.   usage: python tools/op_times.py [kind ...]   (5 = gemm)"""
import collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sonicdiffusionbayeslab_amd import _lib
from sonicdiffusionbayeslab_amd.models import StableDiffusionModel

ub = int(os.environ.get("SD_UB", "16"))
kinds = [int(k) for k in sys.argv[1:]] or [5]
m = StableDiffusionModel.from_pretrained("synthetic:sd15").to("cuda")
u = m.unet
x = torch.randn(ub // 2, 4, 64, 64, device="cuda")
u.set_context(torch.randn(ub, 77, 768, device="cuda"))
for _ in range(2):
    u.forward_latents(x, ub, 501.0)
out = torch.empty(ub, 4, 64, 64, device="cuda")
ws = u._workspace(ub)
buf = C.create_string_buffer(1 << 20)
acc = collections.OrderedDict()
REP = 5
for _ in range(REP):
    n = u._lib.sd_unet_forward_op_times(u._handle, _lib.current_stream(), x.data_ptr(), x.shape[0], ub, 501.0, out.data_ptr(),
                                        u._ws_ptr(ws), ws.numel() - 256, 0, u.cache_branch_id, buf, len(buf))
    assert n > 0, u._lib.sd_last_error()
    for line in buf.value.decode().splitlines():
        i, kind, M, N, K, ms, gf, mb = line.split()
        a = acc.setdefault(int(i), dict(kind=int(kind), M=int(M), N=int(N), K=int(K), ms=0.0, gf=float(gf), mb=float(mb)))
        a["ms"] += float(ms) / REP
tot = collections.defaultdict(float)
for a in acc.values():
    tot[a["kind"]] += a["ms"]
print("per kind (ms):", {k: round(v, 3) for k, v in sorted(tot.items())})
for kind in kinds:
    grp = collections.OrderedDict()
    for i, a in acc.items():
        if a["kind"] == kind:
            key = (a["M"], a["N"], a["K"]) if a["M"] else (int(round(a["mb"] * 1e3)), 0, 0)     # norms carry no M N K: KB moved
            g = grp.setdefault(key, dict(n=0, ms=0.0, gf=a["gf"], mb=a["mb"]))
            g["n"] += 1; g["ms"] += a["ms"]
    print(f"kind {kind}: M N K | launches | us each | total ms | TFLOP/s | GB/s (algorithmic)")
    for (M, N, K), g in sorted(grp.items(), key=lambda kv: -kv[1]["ms"]):
        us = g["ms"] / g["n"] * 1e3
        print(f"  {M:6d} {N:5d} {K:5d} | {g['n']:3d} | {us:8.1f} | {g['ms']:6.3f} | {g['gf'] / us * 1e3 / 1e3:7.1f} | {g['mb'] / us * 1e3:7.0f}")
