"""Per-kernel micro-benchmark at the real SD-1.5 shapes (UNet batch 16 = 8 images with CFG).
Times each op through the C ABI with torch CUDA events on the current stream (the stream the
kernels are launched on) and prints TFLOP/s or GB/s.  Development tool, not part of the product."""
import argparse
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from sonicdiffusionbayeslab_amd import _lib


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ub", type=int, default=16)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    UB = args.ub
    bf = torch.bfloat16
    rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32).to(bf)

    if not args.only or "gemm" in args.only:
        print("== GEMM  M N K ==")
        for (hw, C) in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
            M = UB * hw
            for (N, K, tag) in ((C, C, "proj"), (3 * C, C, "qkv"), (C, 4 * C, "ff2"), (8 * C, C, "geglu")):
                x, w = rnd(M, K), rnd(N, K)
                epi = 1 if tag == "geglu" else 0
                out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=bf)
                f = lambda: _lib.check(lib.sd_op_gemm(st, x.data_ptr(), K, None, 0, K, w.data_ptr(), None, None, None, 0,
                                                      out.data_ptr(), out.shape[1], M, N, K, epi))
                ms = timeit(f)
                print(f"gemm {tag:6s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s")
    if "lnfold" in args.only:
        print("== LayerNorm fold: producer with row partials / consumer with the fold, against the plain GEMMs + LayerNorm ==")
        for (hw, C) in ((4096, 320), (1024, 640), (256, 1280)):
            M = UB * hw
            x, w, r = rnd(M, C), rnd(C, C), rnd(M, C)
            bias = torch.randn(C, device="cuda")
            h = torch.empty(M, C, device="cuda", dtype=bf)
            parts = lib.sd_op_ln_partials(0, M, C)
            rs = torch.zeros(parts, M, 2, device="cuda")
            f0 = lambda: _lib.check(lib.sd_op_gemm(st, x.data_ptr(), C, None, 0, C, w.data_ptr(), bias.data_ptr(), None, r.data_ptr(), C,
                                                   h.data_ptr(), C, M, C, C, 0))
            f1 = lambda: _lib.check(lib.sd_op_gemm_rowstats(st, x.data_ptr(), C, w.data_ptr(), bias.data_ptr(), r.data_ptr(), C,
                                                            h.data_ptr(), C, M, C, C, rs.data_ptr()))
            t0, t1 = timeit(f0), timeit(f1)
            print(f"producer proj M={M:6d} C={C:5d}: plain {t0*1e3:7.1f} us   + row partials {t1*1e3:7.1f} us")
            g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
            n = torch.empty(M, C, device="cuda", dtype=bf)
            tl = timeit(lambda: _lib.check(lib.sd_op_layernorm(st, h.data_ptr(), g.data_ptr(), b.data_ptr(), n.data_ptr(), M, C, 1e-5)))
            for (N, epi, tag) in ((3 * C, 0, "qkv"), (8 * C, 1, "geglu")):
                w2 = rnd(N, C)
                c1, c2 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
                out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=bf)
                f0 = lambda: _lib.check(lib.sd_op_gemm(st, n.data_ptr(), C, None, 0, C, w2.data_ptr(), c2.data_ptr() if epi else None, None, None, 0,
                                                       out.data_ptr(), out.shape[1], M, N, C, epi))
                f1 = lambda: _lib.check(lib.sd_op_gemm_ln(st, h.data_ptr(), C, w2.data_ptr(), c1.data_ptr(), c2.data_ptr(), rs.data_ptr(), parts,
                                                          1e-5, out.data_ptr(), out.shape[1], M, N, C, epi))
                t0, t1 = timeit(f0), timeit(f1)
                print(f"consumer {tag:5s} M={M:6d} N={N:5d}: plain {t0*1e3:7.1f} us (+ LayerNorm {tl*1e3:6.1f} us)   folded {t1*1e3:7.1f} us")
    if not args.only or "conv" in args.only:
        print("== conv3x3 ==")
        cshapes = ((64, 320, 320, 1, 0), (64, 640, 320, 1, 0), (64, 960, 320, 1, 0),
                   (32, 640, 640, 1, 0), (32, 1280, 640, 1, 0), (32, 1920, 640, 1, 0),
                   (16, 1280, 1280, 1, 0), (16, 2560, 1280, 1, 0), (8, 1280, 1280, 1, 0),
                   (8, 2560, 1280, 1, 0), (64, 320, 320, 2, 0), (32, 640, 640, 1, 1))
        if "conv0" in args.only:
            cshapes = cshapes[1:2]
        for (res, cin, cout, stride, up) in cshapes:
            x, w = rnd(UB, res, res, cin), rnd(cout, 3, 3, cin)
            ho = (res << up) // stride
            out = torch.empty(UB, ho, ho, cout, device="cuda", dtype=bf)
            f = lambda: _lib.check(lib.sd_op_conv3x3(st, x.data_ptr(), w.data_ptr(), None, None, None, out.data_ptr(), UB,
                                                     res, res, cin, cout, stride, up))
            ms = timeit(f)
            fl = 2.0 * UB * ho * ho * cout * 9 * cin
            print(f"conv res={res:3d} {cin:5d}->{cout:5d} s{stride} up{up}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
    if "subpix" in args.only:
        for (res, c) in ((32, 640), (16, 1280), (8, 1280)):
            x = rnd(UB, res, res, c); w4 = rnd(4, c, c // 64, 4, 64); b = torch.zeros(c, device="cuda")
            out = torch.empty(UB, 2 * res, 2 * res, c, device="cuda", dtype=bf)
            ms = timeit(lambda: _lib.check(lib.sd_op_conv3x3_upsample_subpixel(st, x.data_ptr(), w4.data_ptr(), b.data_ptr(), out.data_ptr(), UB, res, res, c, c)))
            fl = 2.0 * 4 * UB * res * res * c * 4 * c
            print(f"subpixel upsample conv {res}->{2*res} C={c}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s executed")
    if not args.only or "convio" in args.only:
        x = rnd(UB, 64, 64, 320); w = rnd(4, 9, 320); b = torch.zeros(4, device="cuda")
        out = torch.empty(UB, 4, 64, 64, device="cuda")
        ms = timeit(lambda: _lib.check(lib.sd_op_conv_out(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), UB, 64, 64, 320, 4)))
        print(f"conv_out 64x64 320->4  {ms*1e3:8.1f} us  {x.numel()*2/ms/1e6:7.1f} GB/s read")
    if not args.only or "attn" in args.only:
        print("== attention ==")
        shapes = ((4096, 320, 0), (1024, 640, 0), (256, 1280, 0), (64, 1280, 0), (4096, 320, 77), (1024, 640, 77), (256, 1280, 77))
        if "attn0" in args.only:
            shapes = shapes[:1]
        for (hw, C, nk) in shapes:
            D = C // 8
            Nk = nk or hw
            q = rnd(UB, hw, C); kv = rnd(UB, Nk, 2 * C)
            out = torch.empty(UB, hw, C, device="cuda", dtype=bf)
            f = lambda: _lib.check(lib.sd_op_attention(st, q.data_ptr(), C, kv.data_ptr(), 2 * C, kv.data_ptr() + 2 * C, 2 * C,
                                                       out.data_ptr(), C, UB, 8, hw, Nk, D, 1 / math.sqrt(D)))
            ms = timeit(f)
            fl = 4.0 * UB * 8 * hw * Nk * D
            print(f"attn N={hw:5d} Nk={Nk:5d} D={D:4d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
            if D == 40 and nk == 0:
                kh, vh = rnd(UB, 8, Nk, D), rnd(UB, 8, Nk, D)
                f = lambda: _lib.check(lib.sd_op_attention_headmajor(st, q.data_ptr(), C, kh.data_ptr(), vh.data_ptr(), out.data_ptr(), C,
                                                                     UB, 8, hw, Nk, D, 1 / math.sqrt(D)))
                ms = timeit(f)
                print(f"attn N={hw:5d} Nk={Nk:5d} D={D:4d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  (head-major K / V)")
    if not args.only or "norm" in args.only:
        print("== norms ==")
        for (hw, c1, c2) in ((4096, 320, 0), (4096, 640, 320), (4096, 320, 320), (1024, 640, 0), (1024, 1280, 640), (256, 1280, 0), (256, 1280, 1280), (64, 1280, 1280)):
            x1 = rnd(UB, hw, c1); x2 = rnd(UB, hw, c2) if c2 else None
            C = c1 + c2
            gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            out = torch.empty(UB, hw, C, device="cuda", dtype=bf)
            part = None
            f = lambda: _lib.check(lib.sd_op_groupnorm(st, x1.data_ptr(), c1, x2.data_ptr() if c2 else None, c2, gm.data_ptr(),
                                                       bt.data_ptr(), out.data_ptr(), UB, hw, 32, 1e-5, 1))
            ms = timeit(f, iters=5)
            by = 2.0 * 2 * UB * hw * C
            print(f"groupnorm HW={hw:5d} C={C:5d}  {ms*1e3:8.1f} us  {by/ms/1e6:7.1f} GB/s (r+w; incl. malloc+sync in sd_op)")
        for (hw, C) in ((4096, 320), (1024, 640), (256, 1280)):
            x = rnd(UB * hw, C)
            gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            out = torch.empty_like(x)
            f = lambda: _lib.check(lib.sd_op_layernorm(st, x.data_ptr(), gm.data_ptr(), bt.data_ptr(), out.data_ptr(), UB * hw, C, 1e-5))
            ms = timeit(f)
            print(f"layernorm rows={UB*hw:6d} C={C:5d}  {ms*1e3:8.1f} us  {2.0*2*UB*hw*C/ms/1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
