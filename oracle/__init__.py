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
* MPM kernels (mpm.py:321-498), the statics / state initialisers, the in-place roll-out with span enabling and
  forward_extra: PINNED - the reference's own `@wp.kernel` bodies are EXECUTED (unmodified modules imported from
  /root/reference under a scalar numpy stand-in for the ~20 `wp.*` names, tests/golden/warp_scalar.py) in fp64 and fp32;
  tests/golden/gen_mpm_golden.py writes one-substep vectors (both boundary conditions, wall / floor contact, clamp, a
  disabled span - incl. the rows a fresh model.state() holds for disabled particles), central-difference gradient KATs
  and a 12-step roll-out (mpm_step_*, mpm_grad_*, mpm_rollout, mpm_init .npz); tests/test_oracle_pinned.py holds this
  package to them (1e-11 in fp64).
* batch_svd det / sign rule (svd.py:61-96) and deform_cov_by_F (simulation_utils.py:25-48): PINNED the same way
  (svd_rule.npz, cov_deform.npz).  NOT pinnable: what lives inside warp-lang itself - wp.svd3's Jacobi order and its adjoint
  (adj_svd3); the convention adopted for those is stated in DESIGN.md section 2.
* rasterizer: the reference calls the un-vendored `diff_gaussian_rasterization` CUDA extension
  (graphdeco-inria/gaussian-splatting @ b17ded92b56ba02b6b7eaba2e66a2b0510f27764, README.md:56-61).
  No source, tests or vectors for it exist under /root/reference => "parity unpinned"; the oracle
  restates the published algorithm (3DGS paper + constants listed in SURVEY.md App. D).
"""
