"""GPU: end-to-end LoRA fine-tuning through sim + render (the finetune.py:234-488 loop on the HIP operators)."""
import pytest
import torch

from gpu_util import dev, measured, far_image

pytestmark = pytest.mark.gpu

# equivalence of two schedules against far targets: >= 10x the worst of the measured runs
# measured on MI355X over 5 x 3 runs (gpurun_out r06b): stream loss <= 9.4e-8, gradients <= 3.3e-7; epoch loss <= 2.4e-7, gradients <= 3.6e-7 (two epochs 1.1e-6)
BOUND_STREAM_LOSS, BOUND_STREAM_GRAD = 4e-6, 5e-6
BOUND_EPOCH_LOSS, BOUND_EPOCH_GRAD = 4e-6, 1e-5


def test_finetune_recovers_towards_ground_truth_and_checkpoints(tmp_path):
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import finetune_constitutive, simulate_video
    scene = synth.make_scene("tiny", override=dict(S=10, V=2))
    frames = 3
    # a pre-stretched body, so the constitutive response (not gravity) drives the first 30 substeps
    torch.manual_seed(0)
    F0 = torch.diag(torch.tensor([1.3, 0.75, 1.0])).to(dev())
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
    # (the two paths sum the scatters in different orders; a last-bit difference in a position can flip one of the
    # rasterizer's hard cut-offs - alpha < 1/255, T < 1e-4 - for a pixel, so at this loss level (1.5e-6) the observed
    # relative difference is bimodal: ~1e-5 or ~7e-4)
    assert abs(l3[0] - l4[0]) < 5e-3 * abs(l4[0]) + 1e-9


def test_bptt_gradient_is_a_descent_direction_with_first_order_accuracy():
    """dL/dtheta from the full chain (material nets -> MPM -> binding -> rasterizer -> loss, 30 substeps, 3 frames).

    * reference setting (covariances pushed forward by F, a path the reference deliberately leaves non-differentiable,
      tune/utils.py:353-373): the gradient is a descent direction;
    * rest covariances (every dependence on theta goes through differentiable operators): the central difference of the
      loss along g equals |g|^2.  The image is only piecewise smooth (alpha >= 1/255 cut-off, tile culling: a footprint
      edge crossing a pixel moves it by up to 1/255), so the ground truth is made different enough (L ~ 1e-5) for the
      smooth part to dominate; what remains is a reproducible ~+13 % of edge terms the analytic gradient - like the
      reference's - does not carry (tools/exp_descent.py: 1.07-1.17 over step sizes 0.2 %-5 %; sim-only and render-only
      checks in tools/exp_gradcheck.py give 1.000 and 1.00 +- 0.03)."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss
    scene = synth.make_scene("tiny", override=dict(S=10, V=2))
    torch.manual_seed(0)
    F0 = torch.diag(torch.tensor([1.3, 0.75, 1.0])).to(dev())
    true = SceneRuntime(scene, dev(), fused=True)
    true.F0 = F0.repeat(true.N, 1, 1).contiguous()
    for net in (true.elasticity, true.plasticity):
        for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
            lin.lora_B.data.mul_(40.0)
    c = dict(DEFAULT_CFG, num_frames=3, decay_steps=2)
    for deform_cov in (True, False):
        torch.manual_seed(0)
        gt = simulate_video(true, 3, deform_cov=deform_cov)
        rt = SceneRuntime(scene, dev(), fused=True)
        rt.F0 = true.F0.clone()
        params = rt.parameters()
        L0 = video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=deform_cov)
        grads = torch.autograd.grad(L0, params)
        g2 = sum(float((g.double() ** 2).sum()) for g in grads)
        assert float(L0) > 0 and g2 > 0 and all(torch.isfinite(g).all() for g in grads)

        def loss_at(step):
            with torch.no_grad():
                for p, g in zip(params, grads):
                    p.add_(step * g)
                L = float(video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=deform_cov))
                for p, g in zip(params, grads):
                    p.sub_(step * g)
            return L

        if deform_cov:
            assert loss_at(-0.2 * float(L0) / g2) < float(L0)            # descent direction
        else:
            eps = 0.01 * float(L0) / g2                                   # central difference: curvature cancels
            ratio = (loss_at(eps) - loss_at(-eps)) / (2 * eps * g2)
            assert float(L0) > 1e-6 and 0.85 < ratio < 1.35, (float(L0), g2, ratio)


def test_stage_a_recovers_the_initial_velocity(tmp_path):
    """optimize_init_velocity (finetune.py:63-231): a global initial velocity is fitted through sim + render BPTT; it
    moves towards the velocity the ground-truth video was made with, init.pt is exported in the reference's layout and
    picked up again on the next call."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import optimize_init_velocity, simulate_video
    scene = synth.make_scene("tiny", override=dict(S=10, V=2))
    true = SceneRuntime(scene, dev(), fused=True)
    v_true = torch.tensor([0.6, -0.5, -0.4], device=dev())
    true.v0 = v_true.unsqueeze(0).expand(true.N, -1).contiguous()
    gt = simulate_video(true, 3)
    rt = SceneRuntime(scene, dev(), fused=True)
    v_fit, losses = optimize_init_velocity(rt, gt, dict(num_epochs=60, num_frames=3, lr=0.1, lambda_reg=1e-9,
                                                        scheduler=dict(type="cos", max_steps=60, learning_rate_alpha=0.1)), tmp_path,
                                           log=print)
    assert len(losses) == 60 and losses[-1] < 0.5 * losses[0]
    assert float((v_fit - v_true).norm()) < 0.6 * float(v_true.norm())
    assert rt.v0.shape == (rt.N, 3) and float((rt.v0 - v_fit).abs().max()) == 0.0
    d = torch.load(tmp_path / "init.pt")
    assert set(d) == {"init_x", "init_v"} and d["init_v"].shape == (rt.N, 3)
    rt2 = SceneRuntime(scene, dev(), fused=True)
    v2, l2 = optimize_init_velocity(rt2, gt, dict(num_epochs=40), tmp_path)
    assert l2 == [] and float((rt2.v0 - rt.v0).abs().max()) == 0.0


def test_render_on_a_second_stream_changes_the_schedule_not_the_result():
    """video_loss(overlap_render=True): binding + render + loss of frame f run on a second HIP stream while frame f+1
    simulates; loss and gradients equal the single-stream epoch."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss
    torch.manual_seed(0)
    scene = synth.make_scene("tiny", override=dict(S=6, V=2))
    true = SceneRuntime(scene, dev(), fused=True)
    true.F0 = torch.diag(torch.tensor([1.25, 0.8, 1.0])).to(dev()).repeat(true.N, 1, 1).contiguous()
    for net in (true.elasticity, true.plasticity):
        for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
            lin.lora_B.data.mul_(30.0)
    gt = simulate_video(true, 4)
    torch.manual_seed(0)
    rt = SceneRuntime(scene, dev(), fused=True)
    rt.F0 = true.F0.clone()
    c = dict(DEFAULT_CFG, num_frames=4, decay_steps=2, exclude_steps=(3,))
    # same kernels, different interleaving: equal up to the order of fp32 atomics.  The loss is taken against FAR targets
    # (gpu_util.far_image): a rasterizer cut-off flipping for a pixel between the two runs (the next test) is then ~1e-8 of the loss
    # instead of 1e-5 .. 1e-3, and the bound is demanded of every run - round 5 retried up to three times here, which an
    # intermittent stream-ordering race could have hidden behind
    gt = [[far_image(img) for img in views] for views in gt]
    for attempt in range(3):            # three independent runs, each held to the bound
        res = {}
        for overlap in (False, True):
            for p in rt.parameters():
                p.grad = None
            L = video_loss(rt, gt, c, 0.7, [0, 1], overlap_render=overlap)
            L.backward()
            torch.cuda.synchronize()
            res[overlap] = (float(L), torch.cat([p.grad.reshape(-1) for p in rt.parameters()]).clone())
        assert res[False][0] > 1e-3
        dl = abs(res[True][0] - res[False][0]) / res[False][0]
        dg = float((res[True][1] - res[False][1]).norm()) / float(res[False][1].norm())
        assert measured(dl, "rel loss, render on a second stream (far targets)") < BOUND_STREAM_LOSS
        assert measured(dg, "rel 2-norm of all gradients (far targets)") < BOUND_STREAM_GRAD


def test_loss_differences_between_equivalent_paths_are_rasterizer_cut_off_events():
    """Root cause of the loose bounds above.  The p2g scatter sums with fp32 atomics, so two runs (or the fused and the
    per-operator path) give positions that differ in the last bits; the rasterizer has two hard cut-offs per
    (pixel, Gaussian) - alpha < 1/255 is skipped, T < 1e-4 ends the pixel (forward.cu semantics) - and a last-bit
    difference flips one of them for a handful of pixels.  Everything else agrees to rounding.  Shown here on the
    images themselves: all pixels agree to 2e-6 except a few, and each of those differs by at most one cut-off's
    worth of colour; with loss ~ 1e-6 those few pixels are the whole 1e-5 ... 1e-3 relative difference.
    Observed over four runs: 0, 2, 1, 2 pixels of 147 456 differ by 1.87e-4, every other pixel by <= 1.3e-6."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import simulate_video
    scene = synth.make_scene("tiny", override=dict(S=10, V=2))
    rt = SceneRuntime(scene, dev(), fused=True)
    rt.F0 = torch.diag(torch.tensor([1.3, 0.75, 1.0])).to(dev()).repeat(rt.N, 1, 1).contiguous()
    vids = []
    for fused in (True, True, False):
        rt.fused = fused
        vids.append(simulate_video(rt, 3))
    npix = 0
    worst_round, flips, worst_flip = 0.0, 0, 0.0
    for other in (vids[1], vids[2]):                    # run-to-run, and fused vs per-operator
        for fa, fb in zip(vids[0], other):
            for ia, ib in zip(fa, fb):
                d = (ia - ib).abs().amax(0)             # per pixel, max over channels
                big = d > 2e-6
                npix += d.numel()
                flips += int(big.sum())
                if bool(big.any()):
                    worst_flip = max(worst_flip, float(d[big].max()))
                if bool((~big).any()):
                    worst_round = max(worst_round, float(d[~big].max()))
    print(f"cut-off events: {flips} of {npix} pixels, worst {worst_flip:.3e}; rounding-level worst {worst_round:.3e}")
    assert worst_round <= 2e-6
    # one Gaussian at the alpha threshold contributes at most (1/255) * T * colour, colour <~ 1.5 with the SH offset
    assert worst_flip <= 1.5 / 255 + 1e-6, worst_flip
    assert flips <= max(8, npix // 2000), (flips, npix)


@pytest.mark.parametrize("overlap", [False, True])
def test_native_epoch_matches_the_composition_of_autograd_nodes(overlap):
    """SceneRuntime.epoch (harness._epoch_forward / _epoch_backward: the whole BPTT epoch of finetune.py:331-414 as two plain
    calls - consecutive roll-outs on one checkpoint buffer, per frame binding against the detached previous frame + renders +
    decayed loss, one reverse sweep with the LoRA gradients summed over the frames) against train.video_loss + loss.backward()
    (one autograd node per frame and operator): same loss, same twelve LoRA gradients - with a decay schedule, an excluded
    frame, and the render of frame f overlapped with the simulation of frame f + 1."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss, epoch_weights, native_epoch_ok
    scene = synth.make_scene("tiny", override=dict(S=4, V=2))
    torch.manual_seed(0)
    F0 = torch.diag(torch.tensor([1.3, 0.75, 1.0])).to(dev())
    true = SceneRuntime(scene, dev(), fused=True)
    true.F0 = F0.repeat(true.N, 1, 1).contiguous()
    for net in (true.elasticity, true.plasticity):
        for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
            lin.lora_B.data.mul_(40.0)
    frames = 5
    gt = [[far_image(img) for img in vs] for vs in simulate_video(true, frames)]      # far targets: no cut-off event can show (gpu_util.far_image)
    c = dict(DEFAULT_CFG, num_frames=frames, decay_steps=2, exclude_steps=(3,))
    decay = 0.7
    rt = SceneRuntime(scene, dev(), fused=True)
    rt.F0 = true.F0.clone()
    assert native_epoch_ok(rt)
    views = [0, 1]
    for p in rt.parameters():
        p.grad = None
    ref_loss = video_loss(rt, gt, c, decay, views)
    ref_loss.backward()
    ref = [p.grad.clone() for p in rt.parameters()]
    for p in rt.parameters():
        p.grad = None
    w, steps = epoch_weights(c, decay)
    assert w[2] is None and abs(w[4] - decay ** 2) < 1e-12 and w[0] == 1.0
    loss = rt.epoch(gt, w, views=views, frame_steps=steps, overlap=overlap)
    got = [p.grad.clone() for p in rt.parameters()]
    assert measured(abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)), "rel loss (far targets)") <= BOUND_EPOCH_LOSS, (float(loss), float(ref_loss))
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all()
        assert measured(float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30), "rel_max LoRA grad (far targets)") <= BOUND_EPOCH_GRAD
    # a second epoch accumulates into .grad like loss.backward() does
    rt.epoch(gt, w, views=views, frame_steps=steps, overlap=overlap)
    for a, b in zip([p.grad for p in rt.parameters()], ref):
        assert measured(float((a - 2 * b).abs().max()) / (float(b.abs().max()) + 1e-30), "rel_max LoRA grad, two epochs (far targets)") <= 2 * BOUND_EPOCH_GRAD
    assert rt.last_epoch_note["frames_with_activation_cache"] + rt.last_epoch_note["frames_recomputing"] == frames
