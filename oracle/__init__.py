"""CPU oracle for the Stable Diffusion sampling hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (Kotstantinovskiy/SonicDiffusionBayesLab) holds no
tests, fixtures or golden vectors for this path, its arithmetic lives in third-party
packages that are absent offline (diffusers==0.32.1, DeepCache==0.1.1), and no model
weights exist on disk.  This package is therefore a literal fp32 PyTorch-CPU restatement
of the published algorithms, anchored on the reference's own call sites
(src/models.py:32-335, src/schedulers.py:14-187, src/experiments/deep_cache.py:20-58)
and on closed-form known-answer values, not on outputs of the
reference itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker.  The product package (sonicdiffusionbayeslab_amd)
never imports it and fails loudly when its HIP library is missing.
"""
