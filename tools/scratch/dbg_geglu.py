import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sonicdiffusionbayeslab_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for M in (256, 200, 512):
    C = 320; N = 8 * C; H = 4 * C
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, C, device="cuda") / math.sqrt(C)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    outs = []
    for lean in ("1", "0"):
        os.environ["SD_GEMM_LEAN"] = lean
        out = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
        _lib.check(lib.sd_op_gemm(st, x.data_ptr(), C, None, 0, C, w.data_ptr(), b.data_ptr(), None, None, 0, out.data_ptr(), H, M, N, C, 1))
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
    a, c = outs
    bad = ~(a == c)
    print(f"M={M}: lean NaNs {torch.isnan(a).sum().item()}, general NaNs {torch.isnan(c).sum().item()}, mismatches {bad.sum().item()} of {a.numel()}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  bad rows:", rows[:20].tolist(), "... count", len(rows))
        print("  bad cols:", cols[:40].tolist(), "... count", len(cols))
        r0, c0 = rows[0].item(), cols[0].item()
        print("  sample lean", a[r0, c0:c0 + 8].tolist()); print("  sample gen ", c[r0, c0:c0 + 8].tolist())
