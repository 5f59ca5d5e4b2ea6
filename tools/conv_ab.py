"""A/B of the halo conv at the UNet's shapes: times every shape and prints a SHA-256 of the output bytes, so that two
builds (SD_AMD_LIB=<other libsdhip.so>) can be compared for speed AND for bitwise identity:
    python tools/conv_ab.py > a.txt;  SD_AMD_LIB=/path/to/other/libsdhip.so python tools/conv_ab.py > b.txt;  diff <(cut -f1,3 a.txt) <(cut -f1,3 b.txt)
Development tool."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sonicdiffusionbayeslab_amd import _lib
from tools.bench_ops import timeit

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
UB = int(os.environ.get("SD_UB", "16"))
bf = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32, generator=g).to(bf)
shapes = ((64, 320, 320, 0), (64, 640, 320, 0), (64, 960, 320, 0), (32, 640, 640, 0), (32, 1280, 640, 0), (32, 1920, 640, 0),
          (16, 1280, 1280, 0), (16, 2560, 1280, 0), (8, 1280, 1280, 0), (8, 2560, 1280, 0), (32, 640, 640, 1))
tot = 0.0
for (res, cin, cout, up) in shapes:
    x, w = rnd(UB, res, res, cin), rnd(cout, cin // 64, 9, 64)
    bias = torch.randn(cout, device="cuda", generator=g)
    ho = res << up
    r = rnd(UB, ho, ho, cout)
    out = torch.empty(UB, ho, ho, cout, device="cuda", dtype=bf)
    f = lambda: _lib.check(lib.sd_op_conv3x3(st, x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, r.data_ptr(), out.data_ptr(), UB,
                                             res, res, cin, cout, 1, up))
    ms = timeit(f, iters=20, warm=3)
    torch.cuda.synchronize()
    h = hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    fl = 2.0 * UB * ho * ho * cout * 9 * cin
    tot += ms
    print(f"conv res={res:3d} {cin:5d}->{cout:5d} up{up}\t{ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s\t{h}")
print(f"total\t{tot * 1e3:8.1f} us\t-")
