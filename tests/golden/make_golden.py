"""Generates tests/golden/*.json from the CPU oracle (run from the repo root:
``python tests/golden/make_golden.py``).  The reference holds no fixtures for this path and
cannot be imported offline, so these vectors pin the ORACLE against itself over
time (regression) and against the closed-form known-answer values of SURVEY.md App. A.7; they
do not pin the oracle against the reference ("parity unpinned")."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.schedulers import DDIMOracle, DPMSolverOracle, LCMOracle  # noqa: E402
from oracle.unet import UNetConfig, unet_forward  # noqa: E402
from sonicdiffusionbayeslab_amd.weights import UNetConfig as PC, make_synthetic_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sched_vectors():
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(1, 4, 4, 4, generator=g)
    out = {}
    for name, mk, n in (("ddim50", lambda: DDIMOracle(), 50), ("ddim3", lambda: DDIMOracle(), 3),
                        ("dpmpp20", lambda: DPMSolverOracle(algorithm_type="dpmsolver++", solver_order=2, final_sigmas_type="zero"), 20),
                        ("dpmpp5_o3", lambda: DPMSolverOracle(algorithm_type="dpmsolver++", solver_order=3, final_sigmas_type="zero"), 5),
                        ("dpm10", lambda: DPMSolverOracle(algorithm_type="dpmsolver", solver_order=2, final_sigmas_type="sigma_min"), 10),
                        ("lcm4", lambda: LCMOracle(), 4)):
        s = mk()
        s.set_timesteps(n)
        x = x0.clone()
        gg = torch.Generator().manual_seed(11)
        traj = []
        for i, t in enumerate(s.timesteps):
            eps = torch.randn(x.shape, generator=gg)
            kw = {}
            if name.startswith("lcm") and i < n - 1:
                kw["noise"] = torch.randn(x.shape, generator=gg)
            x, pred = s.step(eps, t, x, **kw)
            traj.append([float(x.double().sum()), float(x.double().abs().sum()), float(pred.double().sum())])
        out[name] = {"timesteps": [int(t) for t in s.timesteps], "traj": traj,
                     "sigmas": [float(v) for v in getattr(s, "sigmas", [])][:4]}
    return out


def unet_vector():
    cfg = PC(sample_size=8, block_out_channels=(64, 128, 128, 128), num_heads=2, cross_attention_dim=64, context_len=5)
    sd = make_synthetic_state_dict(cfg, seed=5)
    import dataclasses
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 8, 8, generator=g)
    ctx = torch.randn(2, 5, 64, generator=g)
    with torch.no_grad():
        y = unet_forward(sd, UNetConfig(**dataclasses.asdict(cfg)), x, 501, ctx)
    return {"sum": float(y.double().sum()), "abs_sum": float(y.double().abs().sum()),
            "first": [float(v) for v in y.flatten()[:8]], "numel": y.numel()}


if __name__ == "__main__":
    json.dump(sched_vectors(), open(os.path.join(OUT, "scheduler_vectors.json"), "w"), indent=1)
    json.dump(unet_vector(), open(os.path.join(OUT, "tiny_unet_vector.json"), "w"), indent=1)
    print("golden vectors written to", OUT)
