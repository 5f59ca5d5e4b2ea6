import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sonicdiffusionbayeslab_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
def run(M, C, fold, epi, lean, reps=8, N=None):
    N = N or (8 * C if epi else 3 * C)
    H = N // 2 if epi else N
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.4).to(torch.bfloat16)
    w = (torch.randn(N, C, device="cuda") / math.sqrt(C)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); c1 = torch.randn(N, device="cuda")
    parts = 2 * (C // 160)
    xf = x.float().view(M, parts, C // parts)
    rs = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=2).permute(1, 0, 2).contiguous()
    os.environ["SD_GEMM_LEAN"] = lean
    outs = [torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(reps)]
    for out in outs:
        if fold:
            _lib.check(lib.sd_op_gemm_ln(st, x.data_ptr(), C, w.data_ptr(), c1.data_ptr(), b.data_ptr(), rs.data_ptr(), parts, 1e-5, out.data_ptr(), H, M, N, C, epi))
        else:
            _lib.check(lib.sd_op_gemm(st, x.data_ptr(), C, None, 0, C, w.data_ptr(), b.data_ptr(), None, None, 0, out.data_ptr(), H, M, N, C, epi))
    torch.cuda.synchronize()
    tot = 0
    for r in outs[1:]:
        bad = (outs[0] != r)
        n = int(bad.sum()); tot += n
        if n:
            idx = bad.nonzero()
            print(f"   mism {n}: rows {sorted(set(idx[:,0].tolist()))[:12]} cols {sorted(set(idx[:,1].tolist()))[:16]}")
            i, j = idx[0].tolist()
            print(f"   e.g. [{i},{j}] {outs[0][i, j].item()} vs {r[i, j].item()}; row {i} tile-row {i % 128}, col {j} tile-col {j % 160}")
    print(f"M={M} C={C} N={N} fold={fold} epi={epi} lean={lean}: total run-to-run mismatches {tot}")
for lean in ("0", "1"):
    run(65536, 320, True, 0, lean)
    run(65536, 320, False, 0, lean)
    run(65536, 320, False, 0, lean, N=320)
    run(16384, 640, True, 1, lean)
    run(16384, 640, False, 1, lean)
