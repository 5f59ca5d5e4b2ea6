"""Build libsdhip.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

``python -m sonicdiffusionbayeslab_amd.build`` or ``build_library()``; hipcc cross-compiles
for gfx950 without a GPU.  The .so stays in the tree (git-ignored) so it travels to the GPU box.

Every translation unit is compiled with ``-save-temps`` in a scratch directory, so the device assembly of the SAME
compilation that produced the object is kept as ``lib/<unit>.s`` and linted (``asm_lint.py``: packed-fp32 ``op_sel`` forms
that MI355X mis-executes beside MFMAs, >8-byte buffer stores with an SGPR soffset, MFMA -> VALU distances, hand-counted LDS
waits).  A violation removes the object and FAILS the build.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile
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
# No packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in these units: hipcc broadcasts a scalar
# into a packed operand with op_sel, and the form that feeds a LO lane from the HI dword of a register pair loses that operand
# for lanes 48-63 on MI355X -- rarely, beside another wave's MFMAs (round 5: profiles/round5_notes.md; round 4 saw it as the
# LayerNorm fold's "c2 alone").  Which half of a pair a scalar lives in is the register allocator's choice, so these units
# (softmax chains of the attention kernels, the small elementwise kernels) are built without the instructions altogether;
# same-box A/B of the whole library: 10.74 -> 10.69 images/s.  The other units keep their explicit packed math (GEGLU
# polynomial, conv epilogues; conv_halo spills without it) and are held to "no lo-lane op_sel" by the lint below.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# attention is VALU-bound: keep MFMA results in arch VGPRs (no v_accvgpr_read/write copies)
# and let fmaxf lower to bare v_max/v_max3 (no canonicalising v_max x,x in front of each operand;
# the kernel masks with -1e30, never with inf/NaN)
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-ffinite-math-only", *NO_PACKED_FP32],
               # fused cross-attention: 160 accumulators + the softmax on them in ONE 256-register pool
               # (keeps its explicit packed softmax pairs -- plain operands, 2 % of the kernel; the one horizontal add hipcc turned
               # into the op_sel form is written as a scalar add in the source, and the lint holds the unit to that)
               "xattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
               "small.hip": NO_PACKED_FP32, "clip.hip": NO_PACKED_FP32, "norm.hip": NO_PACKED_FP32,      # (HBM-bound elementwise kernels)
               # the GEMM epilogues: scalar fp32 stays scalar (common.h::ln_fold: SLP re-packs the LayerNorm fold into
               # v_pk_fma_f32 with op_sel -- the failing form above; packed fp32 is also slower beside MFMAs)
               "gemm_conv.hip": ["-fno-slp-vectorize"], "gemm_lean.hip": ["-fno-slp-vectorize"]}
_HOST_NOISE = "is not a recognized feature for this target (ignoring feature)"      # the host pass sees the device-only feature


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def asm_path(src: str, ablate: bool = False) -> str:
    return os.path.join(LIB_DIR, src.replace(".hip", ".ablate.s" if ablate else ".s"))


def lint_asm(path: str, unit: str, verbose: bool = True) -> None:
    from . import asm_lint
    bad = asm_lint.lint_file(path, unit, verbose=verbose)
    if bad:
        raise RuntimeError(f"{unit}: the assembly hipcc produced violates {len(bad)} rule(s) of asm_lint.py (first: {bad[0]}); "
                           "see the VIOLATION lines above and sonicdiffusionbayeslab_amd/asm_lint.py for what each rule guards")


def compile_unit(hipcc: str, src: str, obj: str, flags, verbose: bool = True) -> None:
    """hipcc -c with -save-temps in a scratch directory; keeps <obj>.o and the device .s of the same compilation."""
    s = os.path.join(CSRC, src)
    tmp = tempfile.mkdtemp(prefix="sdhip_", dir=LIB_DIR)
    try:
        cmd = [hipcc, *flags, "-save-temps", "-c", s, "-o", os.path.join(tmp, "unit.o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        err = "\n".join(l for l in r.stderr.splitlines() if _HOST_NOISE not in l and "argument unused during compilation" not in l)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{err}")
        if verbose and err.strip():
            print(err, file=sys.stderr)
        dev = [f for f in os.listdir(tmp) if f.endswith("-gfx950.s")]
        if len(dev) != 1:
            raise RuntimeError(f"{src}: expected one device assembly file from -save-temps, found {dev}")
        asm = obj[:-2] + ".s"
        shutil.move(os.path.join(tmp, dev[0]), asm)
        try:
            lint_asm(asm, src.split(".")[0], verbose=verbose)
        except Exception:
            os.replace(asm, asm + ".rejected")
            raise
        shutil.move(os.path.join(tmp, "unit.o"), obj)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    me = [os.path.abspath(__file__), os.path.join(HERE, "asm_lint.py")]      # flags / lint rules are inputs of the objects too
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in hdrs + me) or not os.path.exists(o[:-2] + ".s"):
            jobs.append((src, o, [*FLAGS, *EXTRA_FLAGS.get(src, [])]))

    # libsdhip_ablate.so: the same library with the timing-ablation kernels compiled in (-DSD_ABLATE: DIAG instantiations
    # of conv_halo_kernel, the SD_GEMM_TUNE branches of gemm_kernel -- WRONG results by design).  The product library above
    # carries none of them; tools/, bench.py's MFMA-stream probe and one test load this one (_lib.load_ablate()).
    aobjs, ajobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if src in ABLATE_SOURCES:
            o = os.path.join(LIB_DIR, src.replace(".hip", ".ablate.o"))
            if force or _newer(s, o) or any(_newer(h, o) for h in hdrs + me) or not os.path.exists(o[:-2] + ".s"):
                ajobs.append((src, o, [*FLAGS, "-DSD_ABLATE", *EXTRA_FLAGS.get(src, [])]))
        else:
            o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        aobjs.append(o)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda j: compile_unit(hipcc, j[0], j[1], j[2], verbose), jobs + ajobs))
    if force or jobs or not os.path.exists(LIB_PATH):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    if force or jobs or ajobs or not os.path.exists(ABLATE_LIB_PATH):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *aobjs, "-o", ABLATE_LIB_PATH])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
