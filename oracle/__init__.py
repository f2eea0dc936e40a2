"""CPU oracle for the NeuMA hot path (differentiable MLS-MPM substep, neural constitutive
nets, Particle-GS binding + rasterizer).

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
`neuma_amd/` never imports, links or executes anything in here; the product path raises if the
HIP library is missing instead of falling back to this code.

What it is: a restatement in PyTorch (fp64 or fp32, CPU) of the algorithm the reference runs
through Warp kernels / the diff-gaussian-rasterization CUDA extension, each function citing the
reference file:line it follows.  Gradients come from torch autograd on the restated forward, which
is what Warp's generated adjoints compute for the same forward code.

Parity pinning status
---------------------
* material nets (meta.py / loralib.py): PINNED — checked against golden vectors produced by
  importing the reference's own `InvariantFullMetaElasticity/Plasticity` classes with the three
  shipped checkpoints (tests/golden/gen_material_golden.py, tests/golden/material_*.npz).
* SH basis, camera matrices, l1/l2 loss: PINNED — reference modules are pure torch/numpy and were
  imported to generate tests/golden/camera_sh_golden.npz.
* MPM kernels (mpm.py:321-498), batch_svd sign rule (svd.py:61-96), deform_cov_by_F: the
  reference can only run them through warp-lang 0.6.1, which is not installed and not
  installable here, and the reference holds no tests or golden vectors => "parity unpinned" for
  the generated adjoints, wp.svd3's internal ordering and the atomics order.  The restatement is
  checked by construction invariants instead (mass / momentum conservation, affine-field
  reproduction, finite differences).
* rasterizer: the reference calls the un-vendored `diff_gaussian_rasterization` CUDA extension
  (graphdeco-inria/gaussian-splatting @ b17ded92b56ba02b6b7eaba2e66a2b0510f27764, README.md:56-61).
  No source, tests or vectors for it exist under /root/reference => "parity unpinned"; the oracle
  restates the published algorithm (3DGS paper + constants listed in SURVEY.md App. D).
"""
