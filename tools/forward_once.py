"""Two UNet forwards at the bench shape (UNet batch 16), nothing else: the target of the
`rocprofv3 --pmc` passes (tools/run_traffic.sh).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sonicdiffusionbayeslab_amd.models import StableDiffusionModel

ub = int(os.environ.get("SD_UB", "16"))
m = StableDiffusionModel.from_pretrained("synthetic:sd15").to("cuda")
x = torch.randn(ub // 2, 4, 64, 64, device="cuda")
m.unet.set_context(torch.randn(ub, 77, 768, device="cuda"))
for _ in range(2):
    m.unet.forward_latents(x, ub, 501.0)
torch.cuda.synchronize()
print("ok")
