"""Build-time lint of the gfx950 assembly hipcc produced for libsdhip (``build.py`` runs it on the ``.s`` that the SAME
compilation left behind with ``-save-temps`` and FAILS the build on a violation; tests/test_host_cpu.py re-runs it).

Rules, each from a measured failure (profiles/round4_notes.md, profiles/round5_notes.md):

PK_OPSEL   a packed fp32 VALU instruction (``v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32`` ...) whose ``op_sel`` feeds a LO
           result lane from the HI dword of a source (any ``1`` in ``op_sel:[..]``).  On MI355X that form drops the selected
           operand for lanes 48-63 of the lo half -- rarely, and only while the other wave of the SIMD issues MFMAs
           (tools/probes/pk_fma_coexec.hip reproduces it stand-alone; in the LayerNorm-fold GEMMs it showed as 16 rows x 1
           column equal to the bias, ~1 launch in 35).  ``op_sel_hi`` selections and plain packed operands never failed.
STORE_SOFF a buffer store of more than 8 bytes with an SGPR soffset: hipcc assumes it needs no wait state before a VALU
           write of its data registers; gfx950 stored the NEW value in lanes 12-15 of every 16-lane row (round 4, hazard 1).
MFMA_DIST  a vector / LDS / memory instruction that touches the registers of an MFMA sooner than the hardware needs
           (tools/probes/mfma_hazard.hip, wait states by number of passes P: read of D  P + 4, write of D  P + 1, write of a
           SrcC that is not D  P - 4; hipcc pads exactly these for its own code -- the rule guards inline asm and regressions).
LDS_WAITS  hand-counted ``s_waitcnt lgkmcnt(N)`` around inline-asm LDS reads (csrc/attention.hip): no instruction may
           touch a register an in-flight read will write, no spills, no scalar-memory instructions in such a block.
"""
from __future__ import annotations

import re
import sys

_REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs(tok: str) -> set:
    """'v[12:15]' -> {v12..v15}; 'v7' -> {v7}."""
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            out |= {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add(f"{m.group(4)}{m.group(5)}")
    return out


def kernels(text: str, want: str = ""):
    """(name, body) of every function of the module; a body ends at its ``.Lfunc_end`` label (not at the first s_endpgm:
    a kernel with an early exit has several)."""
    for m in re.finditer(r'^(\w+):[ \t]*; @\1[^\n]*\n(.*?)^\.Lfunc_end\d+:', text, flags=re.S | re.M):
        if want in m.group(1):
            yield m.group(1), m.group(2)


def _instructions(body: str):
    """(label or None, text) per line that is an instruction or a label."""
    for line in body.split("\n"):
        mm = re.match(r'^(\.LBB\d+_\d+):', line)
        if mm:
            yield mm.group(1), None
            continue
        t = line.split(";")[0].strip()
        if t and not t.startswith("."):
            yield None, t


def _split_ops(rest: str):
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


_PASSES = (("v_mfma_f32_16x16x32_bf16", 4), ("v_mfma_f32_16x16x32_f16", 4), ("v_mfma_f32_16x16x32_fp8", 4), ("v_mfma_f32_16x16x32_bf8", 4),
           ("v_mfma_f32_32x32x16_bf16", 8), ("v_mfma_f32_32x32x16_f16", 8), ("v_mfma_f32_32x32x16_fp8", 8), ("v_mfma_f32_32x32x16_bf8", 8),
           ("v_mfma_scale_f32_16x16x128_f8f6f4", 8), ("v_mfma_f32_16x16x128_f8f6f4", 8),
           ("v_mfma_scale_f32_32x32x64_f8f6f4", 16), ("v_mfma_f32_32x32x64_f8f6f4", 16))


def _mfma_passes(op: str) -> int:
    for name, p in _PASSES:
        if op.startswith(name):
            return p
    return 16          # unknown shape: the longest


def check_pk_opsel(name: str, body: str):
    errs = []
    for _, t in _instructions(body):
        if t and re.match(r'v_pk_\w+_f32\b', t):
            m = re.search(r'op_sel:\[([01,]+)\]', t)
            if m and "1" in m.group(1):
                errs.append(f"{name}: PK_OPSEL `{t}`")
    return errs


def check_store_soffset(name: str, body: str):
    errs = []
    for _, t in _instructions(body):
        if t and re.match(r'buffer_store_(dwordx3|dwordx4|b96|b128|format_xyzw?)\b', t):
            ops = _split_ops(t.split(None, 1)[1])
            # vdata, vaddr (or off), srsrc, soffset [modifiers]
            soff = ops[3].split()[0] if len(ops) > 3 else "0"
            if re.match(r's\d+|m0|ttmp', soff):
                errs.append(f"{name}: STORE_SOFF `{t}`")
    return errs


def check_mfma_distance(name: str, body: str):
    """Linear walk in program order (fall-through paths; back edges are not followed)."""
    errs = []
    live = []           # [wait states since issue, passes, D registers, SrcC registers that are not D, text]
    for lab, t in _instructions(body):
        if t is None:
            continue
        op = t.split()[0]
        rest = t.split(None, 1)[1] if " " in t else ""
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):      # what follows is entered from elsewhere
            live = []
            continue
        if op == "s_nop":
            n = int(rest.strip()) + 1
            for e in live:
                e[0] += n
            continue
        ops = _split_ops(rest)
        is_mfma = op.startswith("v_mfma") or op.startswith("v_smfmac")
        reads, writes = set(), set()
        if op.startswith("v_") and not is_mfma:
            if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
                for o in ops[1:]:
                    reads |= regs(o)
            elif op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")):
                for o in ops:
                    reads |= regs(o); writes |= regs(o)
            else:
                writes |= regs(ops[0]) if ops else set()
                for o in ops[1:]:
                    reads |= regs(o)
                if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_accvgpr_write")) and ops:
                    reads |= regs(ops[0])
        elif op.startswith(("ds_write", "ds_store", "buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic")):
            for o in ops:
                reads |= regs(o)
        elif op.startswith(("ds_read", "ds_load", "buffer_load", "global_load", "flat_load", "scratch_load", "ds_bpermute", "ds_permute", "ds_swizzle")):
            if ops and " lds" not in (" " + rest):
                writes |= regs(ops[0])
            for o in ops[1:]:
                reads |= regs(o)
            if " lds" in (" " + rest) and ops:
                reads |= regs(ops[0])
        if not is_mfma:
            for ws, p, d, c, mt in live:
                if reads & d and ws < p + 4:
                    errs.append(f"{name}: MFMA_DIST `{t}` reads {sorted(reads & d)[:2]} {ws} wait states after `{mt}` (needs {p + 4})")
                if writes & d and ws < p + 1:
                    errs.append(f"{name}: MFMA_DIST `{t}` writes {sorted(writes & d)[:2]} {ws} wait states after `{mt}` (needs {p + 1})")
                if writes & c and ws < p - 4:
                    errs.append(f"{name}: MFMA_DIST `{t}` writes SrcC {sorted(writes & c)[:2]} {ws} wait states after `{mt}` (needs {p - 4})")
        for e in live:
            e[0] += 1
        live = [e for e in live if e[0] < 24]
        if is_mfma and len(ops) >= 4:
            d = regs(ops[0]); c = regs(ops[3].split()[0]) - d
            live.append([0, _mfma_passes(op), d, c, t])
    return errs


def check_lds_waits(name: str, body: str):
    """In-flight inline-asm LDS reads (see the module docstring).  Returns (errors, number of ds_read_b64_tr_b16)."""
    errors = []
    blocks, cur, lab = [], [], "entry"
    for l, t in _instructions(body):
        if l:
            blocks.append((lab, cur)); cur = []; lab = l
        else:
            cur.append(t)
    blocks.append((lab, cur))
    n_tr = 0
    for lab, ins in blocks:
        pending = []                       # in-flight LDS reads, oldest first: (mnemonic, destination registers)
        has_tr = any(i.startswith("ds_read_b64_tr_b16") for i in ins)
        for k, i in enumerate(ins):
            op = i.split()[0]
            if op.startswith("scratch_"):
                errors.append(f"{name} {lab}: spill instruction `{i}`")
                continue
            if op.startswith("ds_"):
                touched = regs(i.split(None, 1)[1]) if " " in i else set()
                for _, d in pending:
                    if d & touched:
                        errors.append(f"{name} {lab}: `{i}` names {sorted(d & touched)} while an LDS read into it is in flight")
                dest = regs(i.split(None, 1)[1].split(",")[0]) if op.startswith("ds_read") else set()
                pending.append((op, dest))
                n_tr += op == "ds_read_b64_tr_b16"
                continue
            if op.startswith("s_load") or op.startswith("s_buffer_load") or op == "s_memtime":
                if has_tr:
                    errors.append(f"{name} {lab}: scalar-memory instruction `{i}` in a block with counted LDS waits")
                continue
            if op == "s_waitcnt":
                mm = re.search(r'lgkmcnt\((\d+)\)', i)
                if mm:
                    n = int(mm.group(1))
                    pending = pending[len(pending) - n:] if n else []
                elif "vmcnt" not in i and "expcnt" not in i:
                    pending = []               # `s_waitcnt 0`-style full wait
                continue
            touched = regs(i.split(None, 1)[1]) if " " in i else set()
            for o, d in pending:
                if d & touched:
                    errors.append(f"{name} {lab}: `{i}` (instruction {k}) uses {sorted(d & touched)} before the {o} that writes it is waited for")
        if has_tr and any(o == "ds_read_b64_tr_b16" for o, _ in pending):
            errors.append(f"{name} {lab}: block ends with transposed reads still in flight")
    return errors, n_tr


# kernels whose inline-asm LDS reads are waited for with hand-counted lgkmcnt, per translation unit
LDS_WAIT_KERNELS = {"attention": ["attn_pipe40_kernel", "attn_pipe80_kernel"]}


def lint_file(path: str, unit: str | None = None, verbose: bool = True):
    """All rules on one .s; returns the list of violations."""
    text = open(path).read()
    unit = unit or path.rsplit("/", 1)[-1].split(".")[0]
    bad, nk = [], 0
    for name, body in kernels(text):
        nk += 1
        bad += check_pk_opsel(name, body)
        bad += check_store_soffset(name, body)
        bad += check_mfma_distance(name, body)
    for want in LDS_WAIT_KERNELS.get(unit, []):
        found = 0
        for name, body in kernels(text, want):
            found += 1
            errs, n_tr = check_lds_waits(name, body)
            if verbose:
                print(f"{name}: {n_tr} ds_read_b64_tr_b16, {len(errs)} violation(s)")
            bad += errs
        if not found:
            bad.append(f"no kernel matching {want} in {path}")
    if nk != text.count(".amdhsa_kernel "):      # (a host-only unit has none of either)
        bad.append(f"{path}: {text.count('.amdhsa_kernel ')} kernel descriptors but {nk} function bodies recognised")
    if verbose:
        print(f"asm_lint {path}: {nk} kernels, {len(bad)} violation(s)")
        for e in bad[:40]:
            print("  VIOLATION:", e)
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        rc |= 1 if lint_file(p) else 0
    sys.exit(rc)
