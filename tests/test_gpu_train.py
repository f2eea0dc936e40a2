"""GPU: end-to-end LoRA fine-tuning through sim + render (the finetune.py:234-488 loop on the HIP operators)."""
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu


def test_finetune_recovers_towards_ground_truth_and_checkpoints(tmp_path):
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import finetune_constitutive, simulate_video
    scene = synth.make_scene("tiny", override=dict(S=10, V=2))
    frames = 3
    # a pre-stretched body, so the constitutive response (not gravity) drives the first 30 substeps
    F0 = torch.diag(torch.tensor([1.12, 0.92, 1.0])).to(dev())
    # ground-truth video from a "true" material = the same nets with different LoRA factors
    true = SceneRuntime(scene, dev(), fused=True)
    true.F0 = F0.repeat(true.N, 1, 1).contiguous()
    for net in (true.elasticity, true.plasticity):
        for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
            lin.lora_B.data.mul_(10.0)
    gt = simulate_video(true, frames)
    assert len(gt) == frames and gt[0][0].shape == (3, scene.cfg["H"], scene.cfg["W"])
    rt = SceneRuntime(scene, dev(), fused=True)
    rt.F0 = true.F0.clone()
    cfg = dict(num_epochs=12, num_frames=frames, decay_steps=2, elasticity_lr=0.02, plasticity_lr=0.002,
               elasticity_scheduler=dict(type="cos", max_steps=12, learning_rate_alpha=0.1),
               plasticity_scheduler=dict(type="cos", max_steps=12, learning_rate_alpha=0.1), num_lora_ckpts=2)
    logs = []
    losses = finetune_constitutive(rt, gt, cfg, tune_root=tmp_path, log=logs.append)
    assert len(losses) == 12 and all(l == l and l >= 0 for l in losses)
    assert max(losses) < 10 * losses[0]
    files = sorted(p.name for p in tmp_path.glob("*_lora.pt"))
    assert files == ["0010_lora.pt", "0012_lora.pt"]         # epoch 1, 10, 12 saved; newest two kept (finetune.py:470-480)
    ck = torch.load(tmp_path / "0012_lora.pt", map_location="cpu")
    assert set(ck.keys()) == {"elasticity", "plasticity", "loss"}
    assert sorted(ck["elasticity"].keys()) == sorted(f"{p}.fc.lora_{ab}" for p in ("layers.0", "layers.1", "final_layer") for ab in "AB")
    # resume: a fresh runtime picks the newest LoRA checkpoint up (finetune.py:299-309)
    rt2 = SceneRuntime(scene, dev(), fused=True)
    rt2.F0 = true.F0.clone()
    l2 = finetune_constitutive(rt2, gt, dict(cfg, num_epochs=1, resume=True), tune_root=tmp_path)
    assert abs(l2[0] - losses[-1]) < 0.5 * losses[-1] + 1e-12     # continues from the saved adaptor, not from scratch
    # the per-operator (drop-in) path computes the same loss as the fused path for the same weights
    rt2.fused = False
    l3 = finetune_constitutive(rt2, gt, dict(cfg, num_epochs=1, elasticity_lr=0.0, plasticity_lr=0.0), tune_root=None)
    rt2.fused = True
    l4 = finetune_constitutive(rt2, gt, dict(cfg, num_epochs=1, elasticity_lr=0.0, plasticity_lr=0.0), tune_root=None)
    assert abs(l3[0] - l4[0]) < 1e-4 * max(1e-6, abs(l4[0])) + 1e-9


def test_bptt_gradient_is_a_descent_direction_with_first_order_accuracy():
    """dL/dtheta from the full chain (material nets -> MPM -> binding -> rasterizer -> loss, 30 substeps, 3 frames).

    * reference setting (covariances pushed forward by F, a path the reference deliberately leaves non-differentiable,
      tune/utils.py:353-373): the gradient is a descent direction;
    * rest covariances (every dependence on theta goes through differentiable operators): a small step of -eps*g lowers
      the loss by eps*|g|^2 to first order (the loss is only piecewise smooth - alpha cut-offs, tile culling - so the
      check uses a small step and a loose band)."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss
    scene = synth.make_scene("tiny", override=dict(S=10, V=2))
    F0 = torch.diag(torch.tensor([1.12, 0.92, 1.0])).to(dev())
    true = SceneRuntime(scene, dev(), fused=True)
    true.F0 = F0.repeat(true.N, 1, 1).contiguous()
    for net in (true.elasticity, true.plasticity):
        for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
            lin.lora_B.data.mul_(10.0)
    c = dict(DEFAULT_CFG, num_frames=3, decay_steps=2)
    for deform_cov, frac in ((True, 0.2), (False, 0.004)):
        gt = simulate_video(true, 3, deform_cov=deform_cov)
        rt = SceneRuntime(scene, dev(), fused=True)
        rt.F0 = true.F0.clone()
        params = rt.parameters()
        L0 = video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=deform_cov)
        grads = torch.autograd.grad(L0, params)
        g2 = sum(float((g.double() ** 2).sum()) for g in grads)
        assert float(L0) > 0 and g2 > 0 and all(torch.isfinite(g).all() for g in grads)
        eps = frac * float(L0) / g2
        with torch.no_grad():
            for p, g in zip(params, grads):
                p.sub_(eps * g)
            L1 = video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=deform_cov)
        pred = eps * g2
        assert float(L1) < float(L0)
        if not deform_cov:
            ratio = (float(L0) - float(L1)) / pred
            assert 0.6 < ratio < 1.6, (float(L0), float(L1), pred, ratio)
