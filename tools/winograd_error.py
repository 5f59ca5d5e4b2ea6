"""How much would Winograd F(2x2, 3x3) cost the 3x3 convs in accuracy at bf16 operands?  (lead a.)  CPU study, no
product code: the direct form as the halo kernel computes it (bf16 inputs / weights, fp32 accumulate, bf16 output) against
F(2x2, 3x3) with the TRANSFORMED inputs and weights rounded to bf16 (what an MFMA version would feed the matrix cores; the
transforms themselves and the accumulation in fp32), both measured against the fp64 convolution of the same bf16-rounded data.
usage: python tools/winograd_error.py [C] [H]"""
import sys
import torch

torch.manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 320
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
r16 = lambda t: t.to(torch.bfloat16).to(torch.float32)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

for name, xs in (("N(0,1) activations (post GroupNorm+SiLU-like: silu(N(0,1)))", lambda x: torch.nn.functional.silu(x)),
                 ("with 3 outlier channels x 20", None)):
    x = torch.randn(1, C, H, H)
    x = torch.nn.functional.silu(x)
    if xs is None:
        x[:, :3] *= 20.0
    w = torch.randn(C, C, 3, 3) / (9 * C) ** 0.5
    x, w = r16(x), r16(w)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    direct = r16(torch.nn.functional.conv2d(x, w, padding=1))                       # fp32 accumulate, bf16 store
    # Winograd: tiles of 4x4 input (stride 2) -> 2x2 outputs
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                       # [1, C, H/2, H/2, 4, 4]
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, tiles, Bt)                          # B^T d B   (fp32)
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                                    # G g G^T   (fp32)
    for tag, Vq, Uq in (("transformed operands rounded to bf16", r16(V), r16(U)), ("only the transformed INPUT rounded", r16(V), U),
                        ("only the transformed WEIGHTS rounded", V, r16(U)), ("... kept in fp32 (transform error only)", V, U)):
        Mm = torch.einsum("ocil,bcyxil->boyxil", Uq, Vq)                             # 16 independent GEMMs over c, fp32
        Y = torch.einsum("ij,boyxjk,lk->boyxil", At, Mm, At)                         # A^T m A -> [1, O, H/2, H/2, 2, 2]
        out = r16(Y.permute(0, 1, 2, 4, 3, 5).reshape(1, C, H, H))
        print(f"  Winograd F(2x2,3x3), {tag}: rel-L2 {rel(out, ref):.3e}")
    print(f"{name}: C = {C}, {H}x{H}: direct bf16 conv rel-L2 {rel(direct, ref):.3e} (bf16 output rounding alone ~2.3e-3)")
