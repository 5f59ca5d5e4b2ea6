import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
from sonicdiffusionbayeslab_amd import _lib
lib = C.CDLL(os.environ.get("SD_AMD_LIB", _lib.LIB_PATH))
for name in ("sd_op_gemm_ln", "sd_op_gemm"):
    fn = getattr(lib, name); fn.restype, fn.argtypes = _lib._SIGS[name]
class L:
    @staticmethod
    def check(rc):
        assert rc == 0, rc
_lib = L
st = torch.cuda.current_stream().cuda_stream
def run(M, C, epi, lean, fold=True, reps=6, seed=0):
    torch.manual_seed(seed)
    N = 8 * C if epi else 3 * C
    H = N // 2 if epi else N
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.4).to(torch.bfloat16)
    w = (torch.randn(N, C, device="cuda") / math.sqrt(C)).to(torch.bfloat16)
    c2 = torch.randn(N, device="cuda"); c1 = torch.randn(N, device="cuda")
    parts = 2 * (C // 160)
    xf = x.float().view(M, parts, C // parts)
    rs = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=2).permute(1, 0, 2).contiguous()
    os.environ["SD_GEMM_LEAN"] = lean
    outs = [torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(reps)]
    for out in outs:
        if fold:
            _lib.check(lib.sd_op_gemm_ln(st, x.data_ptr(), C, w.data_ptr(), c1.data_ptr(), c2.data_ptr(), rs.data_ptr(), parts, 1e-5, out.data_ptr(), H, M, N, C, epi))
        else:
            _lib.check(lib.sd_op_gemm(st, x.data_ptr(), C, None, 0, C, w.data_ptr(), c2.data_ptr(), None, None, 0, out.data_ptr(), H, M, N, C, epi))
    torch.cuda.synchronize()
    ref = outs[-1]
    for k, r in enumerate(outs[:-1]):
        bad = (ref != r)
        n = int(bad.sum())
        if n:
            idx = bad.nonzero()
            rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
            i, j = idx[0].tolist()
            print(f"  lean={lean} fold={fold} launch {k}: {n} mismatches rows {rows[:6]}..({len(rows)}) [tile {rows[0]//128}, local {rows[0]%128}] cols {cols[:8]}..({len(cols)}) [ntile {cols[0]//160}, local {cols[0]%160}] e.g. {ref[i,j].item()} vs {r[i,j].item()}")
for seed in range(4):
    for lean in ("1", "0"):
        for fold in (True, False):
            run(65536, 320, 0, lean, fold, seed=seed)
print("done")
