"""Per basic block of every kernel in a hipcc `-save-temps` .s file: instruction, MFMA, scratch (spill) load / store, LDS
read, LDS-DMA and wait counts -- to see whether spills or drained waits sit INSIDE the MFMA loops.  Development tool.
usage: python tools/asm_loops.py file.s [min_mfma_per_block]"""
import re, sys
s = open(sys.argv[1]).read()
minm = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel|\n(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, flags=re.S):
    name, body = (m.group(1), m.group(2)) if m.group(1) else (m.group(3), m.group(4))
    blocks, cur, lab = [], [], 'entry'
    for l in body.split('\n'):
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append((lab, cur)); cur = []; lab = l.split(':')[0]
        else:
            cur.append(l)
    blocks.append((lab, cur))
    tot_sl = sum('scratch_load' in x for _, b in blocks for x in b); tot_ss = sum('scratch_store' in x for _, b in blocks for x in b)
    print(f"{name}: {len(blocks)} blocks, scratch loads {tot_sl} stores {tot_ss}")
    for lab, b in blocks:
        nm = sum('v_mfma' in x for x in b)
        if nm >= minm:
            c = lambda pat: sum(bool(re.search(pat, x)) for x in b)
            v0, valu = c(r'vmcnt\(0\)'), c(r'^\s+v_(?!mfma)')
            print(f"  {lab:12s} insts={len(b):5d} mfma={nm:4d} scratch_ld={c('scratch_load'):3d} scratch_st={c('scratch_store'):3d} "
                  f"ds_read={c('ds_read'):3d} lds_dma={c('global_load_lds|buffer_load.*lds'):3d} vmcnt={c('s_waitcnt vmcnt')} "
                  f"vmcnt0={v0} lgkmcnt={c('lgkmcnt')} barrier={c('s_barrier')} valu={valu}")
