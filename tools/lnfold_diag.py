"""Discriminating experiment for the intermittent LayerNorm-fold wrong result of round 4 (profiles/round5_notes.md).

For every library in LIBS (diagnostic variants from tools/build_variant.py; the product library is always included) the
LayerNorm-folded q|k|v projection of the 64x64 level (M 65536, N 960, K 320) and the folded GEGLU of the 32x32 level
(M 16384, N 5120, K 640) are launched ROUNDS x 7 times into fresh tensors; a launch whose bits differ from the majority is
decoded down to (work item, wave, accumulator tile a/b, register j, lane group) and printed with the raw bits of the wrong
and the right value and of the neighbouring operands, so that the event can be read against the .s of the same build
(sonicdiffusionbayeslab_amd/lib/variants/*.s).

  LIBS=pk,pk_drain,pk_late ROUNDS=60 python tools/lnfold_diag.py
"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicdiffusionbayeslab_amd import _lib as L

ROUNDS = int(os.environ.get("ROUNDS", "60"))
VAR_DIR = os.path.join(os.path.dirname(L.LIB_PATH), "variants")


def load(name):
    path = L.LIB_PATH if name == "product" else os.path.join(VAR_DIR, f"libsdhip_{name}.so")
    lib = C.CDLL(path)
    fn = lib.sd_op_gemm_ln
    fn.restype, fn.argtypes = L._SIGS["sd_op_gemm_ln"]
    return fn


def bits(x):
    return x.view(torch.int16).item() & 0xffff


def decode(i, j, epi):
    if epi == 0:      # 128 x 160 tile, waves 2 (M) x 2 (N), wave tile 64 x 80
        r, cl = i % 128, j % 160
        wm, b, lrow = r // 64, (r % 64) // 16, r % 16
        wn, a, cc = cl // 80, (cl % 80) // 16, cl % 16
        return f"item(m {i // 128}, n {j // 160}) wave {wm + 2 * wn} (wm {wm} wn {wn}) acc[a={a}][b={b}] reg j={cc % 4} lane group lq={cc // 4} lrow={lrow}"
    r, cl = i % 256, j % 128     # 256 x 256 tile -> 128 output columns; waves 4 (M) x 2 (N); output tile = accumulator pair (a, a+1)
    wm, b, lrow = r // 64, (r % 64) // 16, r % 16
    wn, ot, cc = cl // 64, (cl % 64) // 16, cl % 16
    # after the permlane16 pairing lane group lq holds output columns 16 (lq & 1) + 8 (lq >> 1) .. + 7 of a PAIR of output tiles
    return f"item(m {i // 256}, n {j // 128}) wave {wm + 4 * wn} (wm {wm} wn {wn}) out tile {ot} (acc a={2 * ot},{2 * ot + 1}) [b={b}] col-in-tile {cc} lrow={lrow}"


def run(name, fn, M, Cc, epi, seed0):
    st = torch.cuda.current_stream().cuda_stream
    N = 8 * Cc if epi else 3 * Cc
    H = N // 2 if epi else N
    parts = 2 * (Cc // 160)
    events = launches = 0
    for rnd in range(ROUNDS):
        torch.manual_seed(seed0 + rnd)
        x = (torch.randn(M, Cc, device="cuda") * 1.5 + 0.4).to(torch.bfloat16)
        w = (torch.randn(N, Cc, device="cuda") / math.sqrt(Cc)).to(torch.bfloat16)
        c1, c2 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        xf = x.float().view(M, parts, Cc // parts)
        rs = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=2).permute(1, 0, 2).contiguous()
        outs = [torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(7)]
        for out in outs:
            rc = fn(st, x.data_ptr(), Cc, w.data_ptr(), c1.data_ptr(), c2.data_ptr(), rs.data_ptr(), parts, 1e-5, out.data_ptr(), H, M, N, Cc, epi)
            assert rc == 0, rc
        torch.cuda.synchronize()
        launches += 7
        iv = [o.view(torch.int16) for o in outs]
        same = [torch.equal(iv[k], iv[0]) for k in range(7)]
        if all(same):
            continue
        # majority = reference
        ref_k = 0 if sum(same) >= 4 else next(k for k in range(7) if sum(torch.equal(iv[k], iv[m]) for m in range(7)) >= 4)
        for k in range(7):
            if torch.equal(iv[k], iv[ref_k]):
                continue
            bad = (iv[k] != iv[ref_k]).nonzero()
            events += 1
            rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
            print(f"[{name}] epi {epi} round {rnd} launch {k}: {bad.shape[0]} elements differ; rows {rows[0]}..{rows[-1]} ({len(rows)}), cols {cols[:8]} ({len(cols)})")
            for (i, j) in bad[:: max(1, bad.shape[0] // 4)][:4].tolist():
                good, wrong = outs[ref_k][i, j], outs[k][i, j]
                line = f"     [{i},{j}] {decode(i, j, epi)}: good {good.item():+.5f} (0x{bits(good):04x}) wrong {wrong.item():+.5f} (0x{bits(wrong):04x})"
                if epi == 0:
                    s = xf[i].sum().double(); q = (xf[i] * xf[i]).sum().double()
                    mean = s / Cc; rstd = 1 / math.sqrt(q / Cc - mean * mean + 1e-5)
                    acc = (x[i].double() @ w[j].double()).item()
                    accl = (x[i, -32:].double() @ w[j, -32:].double()).item()       # the last k-step alone
                    c1j, c2j = c1[j].item(), c2[j].item()
                    cand = {"c2": c2j, "no_mean": rstd * acc + c2j, "acc": acc, "t=acc-mean*c1": acc - mean * c1j, "no_c2": rstd * (acc - mean * c1j),
                            "last_kstep_only": rstd * (accl - mean * c1j) + c2j, "rstd*t": rstd * (acc - mean * c1j), "c1": c1j, "mean": float(mean), "rstd": rstd}
                    wv = wrong.float().item()
                    near = [n for n, v in cand.items() if abs(float(v) - wv) <= abs(wv) * 2 ** -7 + 1e-6]
                    line += f"  matches {near}  (c2 {c2j:+.5f}, acc {acc:+.4f}, mean {float(mean):+.4f}, rstd {rstd:.4f}, c1 {c1j:+.4f})"
                    r0 = i - i % 16
                    line += f"\n          column {j}, rows {r0}..{r0 + 15}: wrong==c2 in {[int(abs(outs[k][r, j].float().item() - c2j) <= abs(c2j) * 2 ** -7) for r in range(r0, r0 + 16)]}"
                print(line)
    print(f"[{name}] M {M} C {Cc} epi {epi}: {events} deviating launches of {launches}", flush=True)
    return events


def main():
    names = ["product"] + [n for n in os.environ.get("LIBS", "").split(",") if n]
    for name in names:
        fn = load(name)
        run(name, fn, 65536, 320, 0, 1000)
        run(name, fn, 16384, 640, 1, 2000)


if __name__ == "__main__":
    main()
