"""Build libsdhip.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

``python -m sonicdiffusionbayeslab_amd.build`` or ``build_library()``; hipcc cross-compiles
for gfx950 without a GPU.  The .so stays in the tree (git-ignored) so it travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsdhip.so")
ABLATE_LIB_PATH = os.path.join(LIB_DIR, "libsdhip_ablate.so")
ABLATE_SOURCES = ("gemm_conv.hip", "conv_halo.hip")     # the translation units with SD_ABLATE code
SOURCES = ["gemm_conv.hip", "gemm_lean.hip", "conv_halo.hip", "norm.hip", "attention.hip", "xattn.hip", "small.hip", "clip.hip", "unet.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "sd_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# attention is VALU-bound: keep MFMA results in arch VGPRs (no v_accvgpr_read/write copies)
# and let fmaxf lower to bare v_max/v_max3 (no canonicalising v_max x,x in front of each operand;
# the kernel masks with -1e30, never with inf/NaN)
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-ffinite-math-only"],
               # fused cross-attention: 160 accumulators + the softmax on them in ONE 256-register pool
               "xattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # the GEMM epilogues: scalar fp32 stays scalar (common.h::ln_fold: the SLP-packed v_pk_fma_f32 form of the
               # LayerNorm fold was the source of a rare wrong result; packed fp32 is also slower beside MFMAs)
               "gemm_conv.hip": ["-fno-slp-vectorize"], "gemm_lean.hip": ["-fno-slp-vectorize"]}


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def check_attention_asm(path: str) -> None:
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lds_waits", os.path.join(HERE, "..", "tools", "check_lds_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.main(path, ["attn_pipe40_kernel"]) != 0:
        os.remove(path)
        raise RuntimeError("attention.hip: hipcc's code around the inline-asm ds_read_b64_tr_b16 reads violates their hand-counted "
                           "lgkmcnt waits (see the VIOLATION lines above); build with the attention kernel's variant 0 "
                           "(SD_ATTN_VARIANT=0 uses no inline-asm reads) only after fixing the source")


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in hdrs):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    # libsdhip_ablate.so: the same library with the timing-ablation kernels compiled in (-DSD_ABLATE: DIAG instantiations
    # of conv_halo_kernel, the SD_GEMM_TUNE branches of gemm_kernel -- WRONG results by design).  The product library above
    # carries none of them; tools/, bench.py's MFMA-stream probe and one test load this one (_lib.load_ablate()).
    aobjs, ajobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if src in ABLATE_SOURCES:
            o = os.path.join(LIB_DIR, src.replace(".hip", ".ablate.o"))
            if force or _newer(s, o) or any(_newer(h, o) for h in hdrs):
                ajobs.append([hipcc, *FLAGS, "-DSD_ABLATE", *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])
        else:
            o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        aobjs.append(o)

    # attention.hip waits for its inline-asm V^T fragment reads with hand-counted `s_waitcnt lgkmcnt(N)`: the .s of the SAME
    # flags is checked on every (re)build (tools/check_lds_waits.py: no MFMA / copy / spill touches a fragment register
    # before its read is waited for) and a violation FAILS the build -- nothing else pins what hipcc does around them.
    attn_s = os.path.join(LIB_DIR, "attention.s")
    attn_src = os.path.join(CSRC, "attention.hip")
    check_attn = force or _newer(attn_src, attn_s) or any(_newer(h, attn_s) for h in hdrs)
    if check_attn:
        jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS["attention.hip"], "-S", "--cuda-device-only", attn_src, "-o", attn_s])

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs + ajobs))
    if check_attn:
        check_attention_asm(attn_s)
    if force or jobs or not os.path.exists(LIB_PATH):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    if force or jobs or ajobs or not os.path.exists(ABLATE_LIB_PATH):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *aobjs, "-o", ABLATE_LIB_PATH])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
