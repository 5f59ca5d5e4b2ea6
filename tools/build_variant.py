"""Build a DIAGNOSTIC variant of libsdhip.so: the listed translation units recompiled with extra flags, everything else from
the product objects.  The result goes to sonicdiffusionbayeslab_amd/lib/variants/libsdhip_<name>.so (+ the .s of the
recompiled units next to it) and is loaded with SD_AMD_LIB=<path>; the product library never contains these flags.

  python tools/build_variant.py pk --units gemm_lean.hip gemm_conv.hip --flags=-DSD_DIAG_LN_PACKED --drop=-fno-slp-vectorize
"""
import argparse, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicdiffusionbayeslab_amd import build as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--units", nargs="+", default=["gemm_lean.hip", "gemm_conv.hip"])
    ap.add_argument("--flags", default="", help="comma-separated, e.g. --flags=-DSD_DIAG_LN_PACKED,-DSD_DIAG_LN_NO_WAIT")
    ap.add_argument("--drop", default="", help="comma-separated product flags to leave out")
    a = ap.parse_args()
    a.flags = [f for f in a.flags.split(",") if f]
    a.drop = [f for f in a.drop.split(",") if f]
    B.build_library(verbose=False)
    out = os.path.join(B.LIB_DIR, "variants"); os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in B.SOURCES:
        if src in a.units:
            flags = [f for f in B.FLAGS + B.EXTRA_FLAGS.get(src, []) if f not in a.drop] + a.flags
            o = os.path.join(out, f"{src[:-4]}.{a.name}.o")
            s = os.path.join(B.CSRC, src)
            subprocess.run([hipcc, *flags, "-c", s, "-o", o], check=True)
            subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", s, "-o", o[:-2] + ".s"], check=True, stderr=subprocess.DEVNULL)
            objs.append(o)
        else:
            objs.append(os.path.join(B.LIB_DIR, src.replace(".hip", ".o")))
    lib = os.path.join(out, f"libsdhip_{a.name}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
