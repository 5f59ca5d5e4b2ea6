import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sonicdiffusionbayeslab_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
def run(M, C, reps=6):
    N, H = 8 * C, 4 * C
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.4).to(torch.bfloat16)
    w = (torch.randn(N, C, device="cuda") / math.sqrt(C)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); c1 = torch.randn(N, device="cuda")
    parts = 2 * (C // 160)
    xf = x.float().view(M, parts, C // parts)
    rs = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=2).permute(1, 0, 2).contiguous()
    outs = [torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(reps)]
    for out in outs:
        _lib.check(lib.sd_op_gemm_ln(st, x.data_ptr(), C, w.data_ptr(), c1.data_ptr(), b.data_ptr(), rs.data_ptr(), parts, 1e-5, out.data_ptr(), H, M, N, C, 1))
    torch.cuda.synchronize()
    tot = [int((outs[-1] != r).sum()) for r in outs[:-1]]
    return tot
print(os.environ.get("SD_AMD_LIB", "default"), [run(16384, 640), run(65536, 320), run(4096, 1280), run(16384, 640)])
