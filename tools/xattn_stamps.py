"""Where a workgroup of the fused cross-attention kernel spends its cycles: s_memtime stamps at the phase boundaries
(diagnostic; the stamp buffer is an argument of the debug entry point sd_op_xattn_fused_stamps).  Development tool."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sonicdiffusionbayeslab_amd import _lib

B, hw, C = (int(x) for x in (sys.argv[1:4] or (16, 4096, 320)))
LN = len(sys.argv) > 4 and sys.argv[4] == "ln"      # the product variant of round 5: norm2 folded in (no stamp entry point)
M = B * hw
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
x, r = torch.randn(M, C, device="cuda").to(bf), torch.randn(M, C, device="cuda").to(bf)
at = (torch.randn(B, C // 32, 640, 32, device="cuda") / math.sqrt(C)).to(bf)
bw = (torch.randn(B, C // 32, 20, 32, 32, device="cuda") / 25).to(bf)
bias = torch.randn(C, device="cuda")
y = torch.empty(M, C, device="cuda", dtype=bf)
nwg = M // 128 * 4
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
call = lambda: _lib.check(lib.sd_op_xattn_fused(st, x.data_ptr(), r.data_ptr(), y.data_ptr(), at.data_ptr(), bw.data_ptr(),
                                                bias.data_ptr(), M, C, hw, 77))
if LN:
    parts = 2 * ((C + 159) // 160)
    xs = x.float().view(M, C // 80, 80)
    rs = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).permute(1, 0, 2).contiguous()
    c2 = torch.randn(B, 640, device="cuda")
    oparts = lib.sd_op_ln_partials(1, M, C)
    ors = torch.empty(oparts, M, 2, device="cuda")
    call = lambda: _lib.check(lib.sd_op_xattn_fused_ln(st, x.data_ptr(), x.data_ptr(), y.data_ptr(), at.data_ptr(), bw.data_ptr(),
                                                       bias.data_ptr(), M, C, hw, 77, rs.data_ptr(), parts, M, c2.data_ptr(), 1e-5,
                                                       ors.data_ptr()))
for _ in range(3):
    call()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    call()
e.record(); torch.cuda.synchronize()
print(f"B={B} hw={hw} C={C}: {s.elapsed_time(e) / 10 * 1e3:.1f} us per launch (no stamps{', norm2 folded, X == R, row partials out' if LN else ''})")
if LN:
    sys.exit(0)
_lib.check(lib.sd_op_xattn_fused_stamps(st, x.data_ptr(), r.data_ptr(), y.data_ptr(), at.data_ptr(), bw.data_ptr(),
                                        bias.data_ptr(), M, C, hw, 77, stamps.data_ptr()))
torch.cuda.synchronize()
t = stamps.cpu().view(-1, 8)
t = t[t[:, 5] != 0]
names = ["phase1 pass0", "softmax0", "phase1 pass1", "softmax1", "phase2"]
d = (t[:, 1:6] - t[:, 0:5]).double()
print(f"{t.shape[0]} workgroups; cycles per workgroup (s_memtime ticks, mean / max):")
for i, n in enumerate(names):
    print(f"  {n:14s} {d[:, i].mean():10.0f} {d[:, i].max():10.0f}")
print(f"  total          {(t[:, 5] - t[:, 0]).double().mean():10.0f}; launch span {(t[:, 5].max() - t[:, 0].min()).item()} ticks")
print(f"  8-wave kernel: phase 1 (incl. first-tile latency) {(t[:, 1] - t[:, 0]).double().mean():.0f}; softmax "
      f"{(t[:, 2] - t[:, 1]).double().mean():.0f}; exchange {(t[:, 3] - t[:, 2]).double().mean():.0f}; phase 2 "
      f"{(t[:, 5] - t[:, 3]).double().mean():.0f}")
