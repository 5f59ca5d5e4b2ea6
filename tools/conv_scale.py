"""Halo conv 64x64 x 320 -> 320 and 32x32 x 640 -> 640: product kernel (mode 0) and its bare MFMA stream (mode 8) at several
batch sizes, from libsdhip_ablate.so: separates the per-launch / per-item fixed cost from the K loop's rate.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sonicdiffusionbayeslab_amd import _lib
from tools.bench_ops import timeit
lib = _lib.load_ablate()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
for (res, c) in ((64, 320), (32, 640), (16, 1280)):
    for UB in (4, 8, 16, 32, 64, 128):
        if UB * res * res * c * 2 > (3 << 30): continue
        x = torch.randn(UB, res, res, c, device="cuda").to(bf); w = torch.randn(c, c // 64, 9, 64, device="cuda").to(bf)
        out = torch.empty(UB, res, res, c, device="cuda", dtype=bf)
        row = []
        for mode in (0, 8):
            f = lambda: _lib.check(lib.sd_op_conv3x3_ablate(st, x.data_ptr(), w.data_ptr(), out.data_ptr(), UB, res, res, c, c, mode))
            ms = timeit(f, iters=20, warm=3)
            row.append((ms * 1e3, 2.0 * UB * res * res * c * 9 * c / ms / 1e9))
        items = (UB * res * res // 256) * ((c + 159) // 160)
        print(f"res={res:3d} C={c:5d} UB={UB:4d} items={items:5d} ({items / 256:5.1f} per CU): kernel {row[0][0]:8.1f} us {row[0][1]:7.1f} TF/s | bare MFMA stream {row[1][0]:8.1f} us {row[1][1]:7.1f} TF/s")
