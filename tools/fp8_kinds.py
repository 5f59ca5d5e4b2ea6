import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/sonicdiffusionbayeslab_amd") else os.getcwd())
import torch
from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
for wd in ("bf16", "fp8"):
    kw = {} if wd == "bf16" else {"weight_dtype": "fp8"}
    m = StableDiffusionModel.from_pretrained("synthetic:sd15", **kw).to("cuda")
    u = m.unet
    ub = 32
    x = torch.randn(ub, 4, 64, 64, device="cuda")
    u.set_context(torch.randn(ub, 77, 768, device="cuda"))
    for _ in range(2): u.forward_latents(x, ub, 501.0)
    acc = {}
    for _ in range(3):
        p = u.forward_profiled(x, ub, 501.0)
        for k, v in p.items(): acc[k] = acc.get(k, 0.0) + v["ms"] / 3
    print(wd, "total", round(sum(acc.values()), 2), {k: round(v, 2) for k, v in acc.items() if v > 0.05})
    del m, u
    torch.cuda.empty_cache()
