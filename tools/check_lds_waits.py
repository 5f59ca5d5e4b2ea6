"""Build-time check of hand-counted `s_waitcnt lgkmcnt(N)` around inline-asm LDS reads (csrc/attention.hip, SD_ATTN_VARIANT 7).

The 64x64 self-attention kernel issues its V^T fragment reads (`ds_read_b64_tr_b16`) from `asm volatile` statements and
waits for them, a whole group later, with literal `s_waitcnt lgkmcnt(6 / 4 / 2 / 0)` in SEPARATE asm statements.  hipcc knows
nothing about that contract: it may place one of its own `ds_read`s between a read and its wait (the literal count is then
one too small), copy or spill a destination register before the wait, or move an MFMA above it.  Today's code generation
is right; this script is what pins it -- `sonicdiffusionbayeslab_amd/build.py` runs it on the `.s` of every build of
attention.hip and FAILS the build otherwise, and tests/test_host_cpu.py runs it on the `.s` the build left behind.

Model (gfx950): LDS instructions return in issue order, so after `s_waitcnt lgkmcnt(N)` all but the wave's N youngest LDS
operations have completed.  Every basic block of the kernel is walked in program order with the list of LDS reads still
in flight (destination registers); any OTHER instruction that names a register an in-flight read will write -- as source
or as destination -- is a violation, as are scratch (spill) instructions anywhere in the kernel and scalar-memory loads in
a block that uses counted LDS waits (SMEM shares the counter and returns out of order).  A block is entered with nothing
of the asm reads in flight (checked: every block that issues a transposed read ends with all of them waited for).

usage: python tools/check_lds_waits.py file.s kernel-name-substring [...]      exit status 1 on a violation
"""
import re
import sys


def regs(tok):
    """'v[12:15]' -> {v12..v15}; 'v7' -> {v7}; anything else -> {}."""
    out = set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', tok):
        out |= {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r'\b([va])(\d+)\b', tok):
        out.add(f"{m.group(1)}{m.group(2)}")
    return out


def kernel_bodies(text, want):
    for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', text, flags=re.S):
        if want in m.group(1):
            yield m.group(1), m.group(2)


def check_kernel(name, body):
    errors = []
    blocks, cur, lab = [], [], "entry"
    for line in body.split("\n"):
        mm = re.match(r'^(\.LBB\d+_\d+):', line)
        if mm:
            blocks.append((lab, cur)); cur = []; lab = mm.group(1)
            continue
        t = line.split(";")[0].strip()
        if t and not t.startswith("."):
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


def main(path, wants):
    text = open(path).read()
    bad, found = [], 0
    for want in wants:
        for name, body in kernel_bodies(text, want):
            found += 1
            errs, n_tr = check_kernel(name, body)
            print(f"{name}: {n_tr} ds_read_b64_tr_b16, {len(errs)} violation(s)")
            bad += errs
    if not found:
        bad.append(f"no kernel matching {wants} in {path}")
    for e in bad[:40]:
        print("  VIOLATION:", e)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2:] or ["attn_pipe40_kernel"]))
