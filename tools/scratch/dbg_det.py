import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sonicdiffusionbayeslab_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
def run(M, C, fold, epi, reps=6):
    N = 8 * C if epi else 3 * C
    H = N // 2 if epi else N
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.4).to(torch.bfloat16)
    w = (torch.randn(N, C, device="cuda") / math.sqrt(C)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); c1 = torch.randn(N, device="cuda")
    parts = 2 * (C // 160)
    xf = x.float().view(M, parts, C // parts)
    rs = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=2).permute(1, 0, 2).contiguous()
    outs = []
    for lean in ("1", "0"):
        os.environ["SD_GEMM_LEAN"] = lean
        res = []
        for r in range(reps):
            out = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
            if fold:
                _lib.check(lib.sd_op_gemm_ln(st, x.data_ptr(), C, w.data_ptr(), c1.data_ptr(), b.data_ptr(), rs.data_ptr(), parts, 1e-5, out.data_ptr(), H, M, N, C, epi))
            else:
                _lib.check(lib.sd_op_gemm(st, x.data_ptr(), C, None, 0, C, w.data_ptr(), b.data_ptr(), None, None, 0, out.data_ptr(), H, M, N, C, epi))
            torch.cuda.synchronize()
            res.append(out.float().cpu())
        nd = sum(int((res[0] != r).sum()) for r in res[1:])
        outs.append(res[0])
        print(f"M={M} C={C} fold={fold} epi={epi} lean={lean}: run-to-run mismatches {nd}, NaNs {int(torch.isnan(res[0]).sum())}")
    bad = outs[0] != outs[1]
    print(f"   lean vs general mismatches {int(bad.sum())}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   rows", rows[:24].tolist(), len(rows), "cols", cols[:24].tolist(), len(cols))
for (M, C, fold, epi) in ((256, 640, True, 1), (256, 640, True, 1), (4096, 640, True, 1), (16384, 640, True, 1), (512, 320, True, 1), (256, 1280, True, 1), (16384, 640, False, 1),
                          (16384, 640, True, 0), (16384, 640, False, 0), (65536, 320, True, 0)):
    run(M, C, fold, epi)
