"""Instruction-class sequence of one basic block of a hipcc .s file, run-length encoded and cut at s_barrier: M = MFMA,
R = ds_read, W = ds_write, D = LDS-DMA, G = other global / buffer load, S = store, v / s = other VALU / SALU, <vN> / <lN> = s_waitcnt
vmcnt / lgkmcnt.  usage: python tools/asm_seq.py file.s .LBB2_82 [max segments]"""
import re, sys
s = open(sys.argv[1]).read()
lab = sys.argv[2]
i = s.index(lab + ':')
m = re.search(r'\n\.LBB\d+_\d+:|\n\.Lfunc_end', s[i + len(lab) + 1:])
body = s[i:i + len(lab) + 1 + (m.start() if m else 400000)]
def cls(l):
    if 'v_mfma' in l: return 'M'
    if 'ds_read' in l: return 'R'
    if 'ds_write' in l: return 'W'
    if re.search(r'global_load_lds|buffer_load\S* .*lds', l): return 'D'
    if re.search(r'global_load|buffer_load|scratch_load', l): return 'G'
    if re.search(r'global_store|buffer_store|scratch_store', l): return 'S'
    if 's_barrier' in l: return '|'
    if 's_waitcnt' in l:
        out = ''
        for k, t in (('vmcnt', 'v'), ('lgkmcnt', 'l')):
            mm = re.search(k + r'\((\d+)\)', l)
            if mm: out += f'<{t}{mm.group(1)}>'
        return out or '<w>'
    if l.startswith('v_'): return 'v'
    if l.startswith('s_'): return 's'
    return '?'
lines = [l.strip() for l in body.split('\n')[1:] if l.strip() and not l.strip().startswith(';')]
toks = []
for l in lines:
    toks += re.findall(r'<[^>]+>|.', cls(l))
out, prev, n = [], None, 0
for t in toks + [None]:
    if t == prev: n += 1
    else:
        if prev: out.append(prev + (str(n) if n > 1 else ''))
        prev, n = t, 1
segs = ' '.join(out).split('|')
for p in segs[:int(sys.argv[3]) if len(sys.argv) > 3 else len(segs)]:
    print(p.strip(), '\n  ---- barrier')
