"""GPU parity: fused roll-out (nm_rollout_*) vs the per-operator drop-in path vs the fp64 oracle chain, and the
frame-level harness (sim + binding + render + loss, forward and backward)."""
import numpy as np
import pytest
import torch

from oracle import material as omat
from oracle import mpm as om
from oracle import raster as orr
from gpu_util import dev, rel_max, abs_max, measured, far_ground_truth

pytestmark = pytest.mark.gpu

# two frames of one runtime against far targets (atomics order only): >= 10x the worst of the measured runs
# measured on MI355X over 5 runs x 3 modes x 2 scenes (gpurun_out r06b): loss <= 3.2e-7, gradients <= 3.5e-7
BOUND_FRAME_LOSS, BOUND_FRAME_GRAD = 4e-6, 5e-6


def _runtime(name="tiny", fused=True, **over):
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    scene = synth.make_scene(name, override=over or None)
    return SceneRuntime(scene, dev(), fused=fused)


def _oracle_rollout(rt, S, x, v, C, F):
    """fp64 chain stress=E(F); sim; F=P(F) with the runtime's merged weights (finetune.py:362-364)."""
    c = rt.model.constant
    const = om.MPMConstant(int(c.num_grids), float(c.dt), int(c.bound), tuple(float(g) for g in c.gravity), float(c.eps), rt.model.bc)
    We = [w.detach().cpu().double().requires_grad_(True) for w in rt.elasticity.effective_weights()]
    Wp = [w.detach().cpu().double().requires_grad_(True) for w in rt.plasticity.effective_weights()]
    st = rt.statics
    vol, rho, clip, en = st.vol.cpu().double(), st.rho.cpu().double(), st.clip_bound.cpu().double(), st.enabled.cpu()
    for _ in range(S):
        stress = omat.elasticity(F, We)
        x, v, C, F = om.step(const, vol, rho, clip, en, x, v, C, F, stress)
        F = omat.plasticity(F, Wp, rt.plasticity.alpha)
    return (x, v, C, F), We, Wp


def test_fused_rollout_equals_per_operator_path_and_oracle():
    S = 3
    rt = _runtime("tiny", fused=True, S=S)
    params = rt.parameters()
    torch.manual_seed(0)
    gws = [torch.randn(rt.N, 3), torch.randn(rt.N, 3), torch.randn(rt.N, 3, 3), torch.randn(rt.N, 3, 3)]
    # start away from F = I (where the reference's SVD adjoint is clamped noise)
    g = torch.Generator().manual_seed(4)
    F0 = (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev())
    res = {}
    for fused in (True, False):
        rt.fused = fused
        ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
        outs = rt.rollout(*ins)
        loss = sum((o * w.to(dev())).sum() for o, w in zip(outs, gws))
        grads = torch.autograd.grad(loss, ins + params)
        res[fused] = ([o.detach() for o in outs], grads)
    for a, b in zip(res[True][0], res[False][0]):
        assert abs_max(a, b) / max(1.0, float(b.abs().max())) < 5e-6      # measured 1.1e-06
    for a, b in zip(res[True][1], res[False][1]):
        assert rel_max(a, b) < 1e-5      # measured 3.3e-06
    oins = [t.detach().cpu().double().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
    oouts, We, Wp = _oracle_rollout(rt, S, *oins)
    for nme, a, b, tol in zip("xvCF", res[True][0], oouts, [3e-7, 3e-6, 1.5e-5, 1.5e-6]):      # measured 8.6e-8 | 7.5e-7 | 4.6e-6 | 3.7e-7
        assert abs_max(a, b) / max(1.0, float(b.abs().max())) < tol, nme
    ol = sum((o * w.double()).sum() for o, w in zip(oouts, gws))
    og = torch.autograd.grad(ol, oins + We + Wp)
    for nme, a, b in zip("xvCF", res[True][1][:4], og[:4]):
        assert rel_max(a, b) < 7e-5, nme      # measured 1.9e-05
    # LoRA factor gradients from the effective-weight gradients (chain rule through W + s B A)
    k = 0
    for net, dW in ((rt.elasticity, og[4:7]), (rt.plasticity, og[7:10])):
        for lin, d in zip((net.layers[0].fc, net.layers[1].fc, net.final_layer.fc), dW):
            gA_ref = lin.scaling * lin.lora_B.detach().cpu().double().T @ d
            gB_ref = lin.scaling * d @ lin.lora_A.detach().cpu().double().T
            assert rel_max(res[True][1][4 + k], gA_ref) < 1.5e-5, (k, "A")      # measured 3.7e-06
            assert rel_max(res[True][1][5 + k], gB_ref) < 1e-5, (k, "B")      # measured 3.3e-06
            k += 2


def test_nonfinite_substep_gradients_are_zeroed_like_the_per_operator_path():
    """interface.py:65-74 sanitises the gradients EVERY substep returns.  An Inf / NaN arriving in dL/dx, dL/dv of the last
    state poisons the grid adjoint of the last substep; the per-operator path zeroes what comes out of that substep and the
    earlier substeps continue with finite values - the fused node must do the same inside (not only at its boundary)."""
    S = 3
    rt = _runtime("tiny", fused=True, S=S)
    params = rt.parameters()
    torch.manual_seed(1)
    gws = [torch.randn(rt.N, 3), torch.randn(rt.N, 3), torch.randn(rt.N, 3, 3), torch.randn(rt.N, 3, 3)]
    gws[0][rt.N // 3, 1] = float("inf")
    gws[1][rt.N // 2, 0] = float("nan")
    g = torch.Generator().manual_seed(4)
    F0 = (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev())
    res = {}
    for fused in (True, False):
        rt.fused = fused
        for p in params:
            p.grad = None
        ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
        outs = rt.rollout(*ins)
        torch.autograd.backward(outs, [w.to(dev()) for w in gws], inputs=ins + params)
        res[fused] = [t.grad.clone() for t in ins + params]
    poisoned = 0
    for a, b in zip(res[True], res[False]):
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        assert rel_max(a, b) < 1.5e-5      # measured 3.4e-06
    # the poison really did reach the neighbours of the two particles through the grid adjoint: against a run with the two
    # entries replaced by zeros, the gradients of more than those two particles differ
    clean = [w.clone() for w in gws]
    clean[0][rt.N // 3, 1] = 0.0
    clean[1][rt.N // 2, 0] = 0.0
    rt.fused = True
    ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
    torch.autograd.backward(rt.rollout(*ins), [w.to(dev()) for w in clean], inputs=ins)
    diff = (res[True][1] - ins[1].grad).abs().amax(1)
    poisoned = int((diff > 1e-3 * float(ins[1].grad.abs().max())).sum())
    assert poisoned > 2


def test_frame_forward_backward_finite_and_consistent():
    rt = _runtime("tiny", fused=True)
    rt.make_ground_truth()
    rt.set_start_state("deformed")      # (at F = I the LoRA gradients of this scene are rounding residue: two runs of the SAME
    r1 = rt.frame()                     #  path differ by 3e-3 there, 4e-7 from a deformed state - tools/exp_grad_noise.py)
    g1 = [p.grad.clone() for p in rt.parameters()]
    assert torch.isfinite(r1.loss) and float(r1.loss) > 0
    assert all(torch.isfinite(g).all() for g in g1) and any(float(g.abs().max()) > 0 for g in g1)
    for p in rt.parameters():
        p.grad = None
    rt.fused = False
    r2 = rt.frame()
    g2 = [p.grad.clone() for p in rt.parameters()]
    assert abs(float(r1.loss) - float(r2.loss)) < 1e-5 * max(1.0, abs(float(r2.loss)))
    for a, b in zip(g1, g2):
        assert rel_max(a, b) < 5e-6      # measured 1.3e-06
    # striped (multi-rank style) loss pieces add up to the full loss
    from neuma_amd.harness import stripe_plan
    rt.fused = True
    total = 0.0
    for rank in range(3):
        rt.world, rt.rank = 3, rank
        total += float(rt.frame(backward=False).loss)
    rt.world, rt.rank = 1, 0
    assert abs(total - float(r1.loss)) < 1e-5 * max(1.0, abs(float(r1.loss)))


def test_the_runtimes_own_kernel_order_changes_nothing_but_the_order(monkeypatch):
    """SceneRuntime keeps its Gaussians in the Hilbert order of their rest positions (NEUMA_GAUSSIAN_ORDER=spatial, the default:
    the binding's gathers stay local) and records the permutation; with `given` it keeps the caller's order.  Same scene either
    way: images, loss, final state and the LoRA gradients agree to the atomics-order level, and the per-kernel arrays of the
    spatial runtime are the caller's arrays permuted by `gaussian_perm`."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    scene = synth.make_scene("tiny", override=dict(S=3, V=2))
    res = {}
    for mode in ("given", "spatial"):
        monkeypatch.setenv("NEUMA_GAUSSIAN_ORDER", mode)
        rt = SceneRuntime(scene, dev(), fused=True)
        assert (rt.gaussian_perm is None) == (mode == "given")
        rt.set_start_state("deformed")
        rt.make_ground_truth()
        with torch.no_grad():
            for p in rt.parameters():
                if p.shape[0] in (64, 9):       # lora_B: a material that is visibly off the ground truth's, so the gradients are a signal
                    p.mul_(-4.0)
        r = rt.frame()
        res[mode] = (rt, r, [p.grad.clone() for p in rt.parameters()], [g.clone() for g in rt.gt])
    (rt0, r0, g0, gt0), (rt1, r1, g1, gt1) = res["given"], res["spatial"]
    perm = rt1.gaussian_perm.to(dev())
    assert sorted(perm.tolist()) == list(range(rt0.K)) and perm.tolist() != list(range(rt0.K))
    assert torch.equal(rt1.gaussians.get_xyz, rt0.gaussians.get_xyz[perm]) and torch.equal(rt1._opacity, rt0._opacity[perm])
    for a, b in zip(gt1, gt0):
        # the ground-truth renders: the same picture (measured 3.0e-07) - apart from single pixels for which a Gaussian sits on the
        # rasterizer's alpha >= 1/255 cut-off: the two runs' particle states differ by the atomics order (1e-7), and such a pixel
        # then gains or loses one contribution of at most 1/255 of a colour (seen: 1 pixel of 12 288 at 7.1e-05, one run in six)
        d = (a - b).abs().amax(0)
        assert int((d > 1e-6).sum()) <= 3 and abs_max(a, b) < 4e-3
        assert measured(d.flatten().kthvalue(d.numel() - 3).values, "abs") < 1e-6
    assert abs(float(r1.loss) - float(r0.loss)) < 2e-5 * max(1e-12, abs(float(r0.loss)))
    assert rel_max(r1.x, r0.x) < 3e-7 and rel_max(r1.F, r0.F) < 7e-7
    for a, b in zip(g1, g0):
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0 and rel_max(a, b) < 3e-6      # measured 1.0e-06


@pytest.mark.parametrize("scene,over", [("tiny", None), ("tiny", dict(S=4, V=2))])
def test_one_node_frame_equals_the_composition_of_nodes(scene, over, monkeypatch):
    """SceneRuntime.frame() on one GPU runs the whole frame as ONE autograd node over the LoRA factors (harness._Frame: merge of
    both nets, nm_rollout_*, binding, renders, loss and their adjoints); NEUMA_LEAN_FRAME=0 keeps the composition LoRA merge ->
    roll-out node -> frame-tail node.  Same library calls: loss, state and the twelve LoRA gradients agree (atomics order only);
    a non-trivial de-normalisation (center / size, nclaw/utils.py:110-118) goes through both as well."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime, _Frame
    for unit in (True, False):
        rt = SceneRuntime(synth.make_scene(scene, override=over), dev(), fused=True)
        if not unit:
            rt.center = torch.tensor([0.1, -0.2, 0.05], device=dev())
            rt.size = torch.tensor([1.5, 0.8, 1.1], device=dev())
        rt.set_start_state("deformed")      # (at F = I the gradients are run-to-run noise of a cancelling sum in either path)
        rt.make_ground_truth()
        far_ground_truth(rt)
        res = {}
        # "direct": the two halves called back to back, gradients added to .grad by the runtime (what frame() does);
        # "graph": the same as an autograd node (harness._Frame) through loss.backward(); "nodes": the composition
        # ("direct": each view's rasterizer adjoint rides behind its own forward pass and loss, no join between the sweeps;
        # "joined": NEUMA_EAGER_RENDER_BWD=0, all adjoints after the join, as the autograd forms have it)
        modes = {"direct": ("1", "0", "1"), "joined": ("1", "0", "0"), "graph": ("1", "1", "1"), "nodes": ("0", "0", "1")}

        def run(mode):
            lean, graph, eager = modes[mode]
            monkeypatch.setenv("NEUMA_LEAN_FRAME", lean)
            monkeypatch.setenv("NEUMA_LEAN_GRAPH", graph)
            monkeypatch.setenv("NEUMA_EAGER_RENDER_BWD", eager)
            assert rt._lean_ok() == (lean == "1")
            for _ in range(2):          # (second frame: cached capacities, pooled buffers, hinted plans)
                for p in rt.parameters():
                    p.grad = None
                r = rt.frame()
            grads = [p.grad.clone() for p in rt.parameters()]
            r = rt.frame()              # a third frame without clearing: the gradients accumulate like loss.backward() does
            for p, g in zip(rt.parameters(), grads):
                assert rel_max(p.grad, 2 * g) < 2e-5
            return r, grads

        for mode in modes:
            res[mode] = run(mode)
        for mode in ("direct", "joined", "graph"):
            # two frames of this scene differ by the order of the scatters' float atomics only: the loss is taken against far
            # targets (gpu_util.far_ground_truth), so that a rasterizer cut-off flipping for a pixel between two frames is below
            # that noise and the bound holds for every pair (round 5 retried here)
            (r0, g0), (r1, g1) = res["nodes"], res[mode]
            assert torch.isfinite(r1.loss) and float(r0.loss) > 1e-3
            assert measured(abs(float(r1.loss) - float(r0.loss)) / abs(float(r0.loss)), "rel loss, " + mode) < BOUND_FRAME_LOSS
            assert rel_max(r1.x, r0.x) < 3e-7 and rel_max(r1.F, r0.F) < 7e-7 and r1.F.shape == r0.F.shape      # measured 9.7e-08
            assert len(g1) == 12 and any(float(g.abs().max()) > 0 for g in g1)
            assert all(a.shape == b.shape and bool(torch.isfinite(a).all()) for a, b in zip(g1, g0))
            worst = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) for a, b in zip(g1, g0))
            assert measured(worst, "rel max of the LoRA gradients, " + mode) < BOUND_FRAME_GRAD


def test_frame_image_matches_oracle_render():
    """End-to-end image parity on the tiny scene: HIP sim+binding+raster vs oracle raster on the HIP state."""
    rt = _runtime("tiny", fused=True)
    from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
    with torch.no_grad():
        x, v, C, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
        means3D = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
        dg = compute_bindings_F(F, rt.bindings)
        img = rt.render_view(means3D, dg, 0)
    cam = rt.cameras[0]
    import math
    s = orr.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), rt.background.cpu().double(),
                     1.0, cam.world_view_transform.cpu().double(), cam.full_proj_transform.cpu().double(),
                     rt.gaussians.active_sh_degree, cam.camera_center.cpu().double())
    cov = orr.deform_cov_by_F(rt._cov.cpu().double(), dg.cpu().double())
    oimg, _ = orr.render(s, means3D.cpu().double(), cov, rt._opacity.cpu().double(), shs=rt._shs.cpu().double())
    assert abs_max(img, oimg) < 3e-6      # measured 7.9e-07


def test_grid_cache_restores_the_same_grid_as_the_recompute():
    """Reverse sweep with the grid cache (nm_mpm_forward_ex / nm_mpm_backward_ex) == recompute of mpm.py:312-315:
    same gradients with an ample capacity, with the auto-sized one, and with a capacity every substep overflows
    (record marked invalid -> transparent fallback)."""
    S = 4
    rt = _runtime("tiny", fused=True, S=S)
    params = rt.parameters()
    torch.manual_seed(1)
    gws = [torch.randn(rt.N, 3), torch.randn(rt.N, 3), torch.randn(rt.N, 3, 3), torch.randn(rt.N, 3, 3)]
    g = torch.Generator().manual_seed(5)
    F0 = (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev())
    blocks, _ = (rt.rollout(rt.x0, rt.v0, rt.C0, F0), rt.model.grid_stats())[1]
    assert blocks > 1
    res = {}
    import neuma_amd.rollout as ro
    # "verified": the record headers have reached the host before the backward pass starts, so the reverse sweep runs
    # without fall-back launches and, from its second substep on, restores the grid inside the plasticity kernel
    # (GridPrologue mode 2); "unverified": header read-back switched off, every substep keeps the guarded launches
    for tag, cap in (("off", 0), ("ample", 4 * blocks), ("verified", 4 * blocks), ("unverified", 4 * blocks), ("overflow", 1),
                     ("auto", None)):
        rt.sim_fused._cache_blocks = cap
        if cap is None:      # auto: the first call sizes the cache, the second one uses it
            rt.rollout(rt.x0, rt.v0, rt.C0, F0)
            assert rt.sim_fused.grid_cache_blocks() == int(1.5 * blocks) + 64
        ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
        was = ro._CACHE_STATUS
        ro._CACHE_STATUS = tag != "unverified"
        try:
            outs = rt.rollout(*ins)
        finally:
            ro._CACHE_STATUS = was
        loss = sum((o * w.to(dev())).sum() for o, w in zip(outs, gws))
        if tag == "verified":
            torch.cuda.synchronize()
            assert outs[0].grad_fn is not None
        res[tag] = torch.autograd.grad(loss, ins + params)
    for tag in ("ample", "verified", "unverified", "overflow", "auto"):
        for a, b in zip(res[tag], res["off"]):
            assert torch.isfinite(a).all()
            assert rel_max(a, b) < 2e-5, tag      # measured 6.7e-06


def test_grid_cache_per_step_api_matches_plain_backward():
    """nm_mpm_forward_ex + nm_mpm_backward_ex against nm_mpm_forward + nm_mpm_backward on one substep."""
    import ctypes as C
    from neuma_amd import _lib as L
    rt = _runtime("tiny", fused=False, S=1)
    lib = L.lib()
    n = rt.N
    d = dev()
    torch.manual_seed(2)
    x, v = rt.x0.clone(), rt.v0.clone() + 0.1 * torch.randn(n, 3, device=d)
    Cm = 0.1 * torch.randn(n, 3, 3, device=d)
    F = torch.eye(3, device=d).repeat(n, 1, 1) + 0.02 * torch.randn(n, 3, 3, device=d)
    stress = torch.randn(n, 3, 3, device=d)
    st = rt.statics.c_struct()

    def parts(*ts):
        return L.nm_particles(*[L.ptr(t) if t is not None else None for t in ts])

    outs = {}
    for tag in ("plain", "cached"):
        nx, nv, nC, nF = (torch.empty_like(t) for t in (x, v, Cm, F))
        cur, nxt = parts(x, v, Cm, F, stress), parts(nx, nv, nC, nF, None)
        torch.manual_seed(3)
        gn = [torch.randn(t.shape, device=d) for t in (x, v, Cm, F)]
        gc = [torch.empty_like(t) for t in (x, v, Cm, F, stress)]
        gnp, gcp = parts(*gn, None), parts(*gc)
        h, s = rt.model.handle(), L.stream_ptr(d)
        if tag == "plain":
            L.check(lib.nm_mpm_forward(h, n, C.byref(st), C.byref(cur), C.byref(nxt), s), "fwd")
            L.check(lib.nm_mpm_backward(h, n, C.byref(st), C.byref(cur), C.byref(nxt), C.byref(gnp), C.byref(gcp), s), "bwd")
        else:
            cap = 4096
            rec = torch.empty(int(lib.nm_mpm_gridcache_bytes(cap)), dtype=torch.uint8, device=d)
            L.check(lib.nm_mpm_forward_ex(h, n, C.byref(st), C.byref(cur), C.byref(nxt), L.ptr(rec), cap, s), "fwd_ex")
            hdr = rec[:4].view(torch.int32)
            assert int(hdr[0]) == rt.model.grid_stats()[0] > 0
            # an unrelated step in between must not matter: the record carries everything the adjoint needs
            L.check(lib.nm_mpm_forward(h, n, C.byref(st), C.byref(parts(x + 0.01, v, Cm, F, stress)), C.byref(parts(*[torch.empty_like(t) for t in (x, v, Cm, F)], None)), s), "fwd")
            L.check(lib.nm_mpm_backward_ex(h, n, C.byref(st), C.byref(cur), C.byref(nxt), C.byref(gnp), C.byref(gcp),
                                           L.ptr(rec), cap, s), "bwd_ex")
        torch.cuda.synchronize()
        outs[tag] = [nx, nv, nC, nF] + gc
    for a, b in zip(outs["cached"], outs["plain"]):
        assert rel_max(a, b) < 7e-6      # measured 2.1e-06


def test_rollout_reorders_shuffled_particles_transparently():
    """NeuMA's data preparation shuffles the particles (tune/utils.py:270-272).  The fused roll-out detects the bad order,
    runs on a Hilbert-sorted copy and hands results and gradients back in the caller's order: same numbers as the sorted
    scene, permuted."""
    rt = _runtime("tiny", fused=True, S=3)
    N = rt.N
    torch.manual_seed(3)
    shuf = torch.randperm(N, device=dev())
    g = torch.Generator().manual_seed(6)
    F0 = (torch.eye(3) + 0.05 * torch.randn(N, 3, 3, generator=g)).to(dev())
    gws = [torch.randn(N, 3, generator=g).to(dev()), torch.randn(N, 3, 3, generator=g).to(dev())]
    params = rt.parameters()

    def run(perm):
        ins = [t[perm].clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
        outs = rt.rollout(*ins)
        loss = (outs[0] * gws[0][perm]).sum() + (outs[3] * gws[1][perm]).sum()
        grads = torch.autograd.grad(loss, ins + params)
        return [o.detach() for o in outs], grads

    ident = torch.arange(N, device=dev())
    assert rt.sim_fused.order_quality(rt.x0) > 0.8
    o_ref, g_ref = run(ident)
    assert rt.sim_fused._perm is False                                   # Hilbert-ordered scene: left alone
    from neuma_amd.rollout import MPMFusedDiffSim
    rt.sim_fused = MPMFusedDiffSim(rt.model, rt.elasticity, rt.plasticity, rt.S)
    assert rt.sim_fused.order_quality(rt.x0[shuf]) < 0.5
    o_sh, g_sh = run(shuf)
    assert torch.is_tensor(rt.sim_fused._perm) and rt.sim_fused._perm.numel() == N
    for a, b in zip(o_sh, o_ref):
        assert abs_max(a, b[shuf]) / max(1.0, float(b.abs().max())) < 3e-6      # measured 8.4e-07
    for a, b in zip(g_sh[:4], g_ref[:4]):
        assert rel_max(a, b[shuf]) < 1.5e-6      # measured 5.0e-07
    for a, b in zip(g_sh[4:], g_ref[4:]):
        assert rel_max(a, b) < 7e-6      # measured 1.9e-06


def test_long_rollout_keeps_the_active_block_list_consistent():
    """600 substeps: the ball falls, hits the floor and deforms, so grid blocks enter and leave the active list all the time
    (k_clear carry-over, three-list rotation, new-block stamping).  At every checkpoint the engine's bookkeeping must agree
    with a dense export of the grid: every node with mass lies in a listed block, the counts match, nothing is NaN, and the
    total grid mass equals the particle mass."""
    rt = _runtime("tiny", fused=True, S=50)
    G = int(rt.model.constant.num_grids)
    x, v, C, F = rt.x0, rt.v0, rt.C0, rt.F0
    pm = float(rt.statics.vol[0] * rt.statics.rho[0]) * rt.N
    hit_floor = False
    with torch.no_grad():
        for chunk in range(12):
            x, v, C, F = rt.rollout(x, v, C, F)
            assert all(torch.isfinite(t).all() for t in (x, v, C, F))
            mv, m, vg = rt.model.grid_export()
            blocks, nodes = rt.model.grid_stats()
            assert nodes == int((m > 0).sum())
            bm = m.reshape(G // 4, 4, G // 4, 4, G // 4, 4).amax(dim=(1, 3, 5)) > 0
            assert int(bm.sum()) <= blocks <= 3 * int(bm.sum()) + 64      # listed blocks cover the massive ones, no runaway growth
            assert abs(float(m.double().sum()) - pm) < 1e-4 * pm
            hit_floor = hit_floor or float(x[:, 1].min()) < 2.5 / G
    assert hit_floor, "the scenario is meant to include wall contact"


def test_svd_and_activation_caches_do_not_change_the_reverse_sweep():
    """nm_rollout_cfg.svd_cache / act_cache: the reverse sweep loads what the forward kernels kept (SVD factors; second-layer
    activations, derivatives, outputs) instead of recomputing it.  Same arithmetic: outputs unchanged, gradients equal to the
    full recompute up to the scatter's atomics order and the last-bit difference between the forward pass's trial F and the one rebuilt from the checkpoints.
    Two nodes alive at once (two leases of the pooled cache buffers)."""
    import neuma_amd.rollout as R
    S = 4
    rt = _runtime("tiny", fused=True, S=S)
    params = rt.parameters()
    gen = torch.Generator().manual_seed(11)
    F0 = (torch.eye(3) + 0.04 * torch.randn(rt.N, 3, 3, generator=gen)).to(dev())
    gw = [torch.randn(rt.N, 3, generator=gen).to(dev()), torch.randn(rt.N, 3, 3, generator=gen).to(dev())]
    saved = (R._SVD_CACHE, R._ACT_CACHE)
    res = {}
    try:
        for mode, (svd, act) in {"recompute": (False, "0"), "svd": (True, "0"), "both": (True, "1")}.items():
            R._SVD_CACHE, R._ACT_CACHE = svd, act
            for p in params:
                p.grad = None
            ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0)]
            out = rt.rollout(ins[0], ins[1], rt.C0, F0)
            out2 = rt.rollout(ins[0], ins[1], rt.C0, F0)            # a second live node: its own buffers
            ((out2[0] * gw[0]).sum() + (out2[3] * gw[1]).sum() + (out[0] * gw[0]).sum() + (out[3] * gw[1]).sum()).backward()
            res[mode] = ([o.detach().clone() for o in out], [t.grad.clone() for t in ins + params])
            if mode == "both":
                assert R.live_bytes() == 0 and sum(len(v) for v in R._POOL.values()) >= 2      # both leases back in the pool
    finally:
        R._SVD_CACHE, R._ACT_CACHE = saved
    for mode in ("svd", "both"):
        for a, b in zip(res[mode][0], res["recompute"][0]):
            assert rel_max(a, b) < 1e-5, mode           # the forward pass itself does not change (fp32 atomics order: not bitwise)
        for a, b in zip(res[mode][1], res["recompute"][1]):
            assert torch.isfinite(a).all() and rel_max(a, b) < 7e-6, mode      # measured 2.1e-06


def test_forward_pair_launch_equals_one_launch_per_net():
    """nm_rollout_set_forward_pair: plasticity(t) + elasticity(t+1) in one launch (F_{t+1} handed over in registers, the grid
    clear of substep t+1 riding along with the velocities left in place) against one launch per net (finetune.py:362-364).
    Same arithmetic per particle; only the scatter's atomics order differs between two runs.  The scene drops and gains grid
    blocks on the way (falling ball), and the checkpoints the reverse sweep reads (F, stress, SVD / activation caches of both
    nets) are written by the other kernel in each mode, so the gradients cover them."""
    from neuma_amd import _lib
    S = 24
    rt = _runtime("tiny", fused=True, S=S)
    params = rt.parameters()
    gen = torch.Generator().manual_seed(5)
    F0 = (torch.eye(3) + 0.04 * torch.randn(rt.N, 3, 3, generator=gen)).to(dev())
    gw = [torch.randn(rt.N, 3, generator=gen).to(dev()), torch.randn(rt.N, 3, 3, generator=gen).to(dev())]
    res = {}
    try:
        for on in (1, 0):
            assert _lib.lib().nm_rollout_set_forward_pair(1 if on else 0) == 0
            for p in params:
                p.grad = None
            ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0)]
            out = rt.rollout(ins[0], ins[1], rt.C0, F0)
            ((out[0] * gw[0]).sum() + (out[3] * gw[1]).sum()).backward()
            mv, m, vg = rt.model.grid_export()
            res[on] = ([o.detach().clone() for o in out] + [m, vg], [t.grad.clone() for t in ins + params])
    finally:
        _lib.lib().nm_rollout_set_forward_pair(1)
    for on in (1,):
        for a, b in zip(res[on][0], res[0][0]):
            assert torch.isfinite(a).all() and rel_max(a, b) < 2e-5, on
        for a, b in zip(res[on][1], res[0][1]):
            assert torch.isfinite(a).all() and rel_max(a, b) < 1e-5, on      # measured 2.7e-06


def test_disabled_particles_get_the_reference_rows_in_the_fused_and_the_per_operator_path():
    """A disabled particle's row of the next state is what the reference's out-of-place sims return for it: the fresh
    model.state() untouched by g2p (zeros, F = I; mpm.py:84-93, 443-444, interface.py:101-123), which the plasticity net then
    maps like any other row (finetune.py:364).  The fused node writes exactly that into its checkpoints; all rows compared."""
    S = 3
    rt = _runtime("tiny", fused=True, S=S)
    lo, hi = rt.N // 3, rt.N // 3 + rt.N // 8
    rt.statics.enabled[lo:hi] = 0
    params = rt.parameters()
    g = torch.Generator().manual_seed(8)
    F0 = (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev())
    gws = [torch.randn(rt.N, 3, generator=g), torch.randn(rt.N, 3, generator=g), torch.randn(rt.N, 3, 3, generator=g),
           torch.randn(rt.N, 3, 3, generator=g)]
    res = {}
    for fused in (True, False):
        rt.fused = fused
        for p in params:
            p.grad = None
        ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
        outs = rt.rollout(*ins)
        torch.autograd.backward(outs, [w.to(dev()) for w in gws], inputs=ins + params)
        res[fused] = ([o.detach().clone() for o in outs], [t.grad.clone() for t in ins + params])
    with torch.no_grad():
        FI = rt.plasticity(torch.eye(3, device=dev()).repeat(hi - lo, 1, 1))
    for fused in (True, False):
        x, v, C, F = res[fused][0]
        assert float(x[lo:hi].abs().max()) == 0 and float(v[lo:hi].abs().max()) == 0 and float(C[lo:hi].abs().max()) == 0, fused
        assert abs_max(F[lo:hi], FI) < 2e-7, fused      # measured 0.0e+00
        for gi in res[fused][1][:4]:
            assert float(gi[lo:hi].abs().max()) == 0, fused          # nothing flows back into a disabled particle's inputs
    for nme, a, b, tol in zip("xvCF", res[True][0], res[False][0], [2e-7, 7e-7, 5e-6, 7e-7]):      # measured 6e-8 | 1.8e-7 | 1.2e-6 | 2.1e-7
        assert abs_max(a, b) / max(1.0, float(b.abs().max())) < tol, nme
    for a, b in zip(res[True][1], res[False][1]):
        assert rel_max(a, b) < 2e-5      # measured 6.7e-06
    rt.statics.enabled.fill_(1)


def test_skipping_the_zero_plasticity_adjoint_of_the_last_substep_changes_nothing(monkeypatch):
    """nm_rollout_cfg.last_gF_zero (round 5): the frame driver promises dL/dF of the last record is zero - the loss sees positions
    only, as the reference's (tune/utils.py:353-373) - and the reverse sweep leaves out the last substep's plasticity adjoint, whose
    output and weight gradients are zeros (and the forward sweep that step's SVD / activation records).  Same loss, same twelve
    LoRA gradients as with the launch - up to the order of the scatters' float atomics (two frames of ONE runtime differ by as
    much: measured 1.8e-6 .. 2.4e-6 over eight runs).  Far targets, so that no rasterizer cut-off event can show: three pairs of
    frames, each held to the bound."""
    from neuma_amd import synth, harness
    from neuma_amd.harness import SceneRuntime
    rt = SceneRuntime(synth.make_scene("tiny", override=dict(S=4, V=2)), dev(), fused=True)
    rt.set_start_state("deformed")
    rt.make_ground_truth()
    with torch.no_grad():
        for p in rt.parameters():
            if p.shape[0] in (64, 9):
                p.mul_(-4.0)
    far_ground_truth(rt)        # a cut-off event between the two frames is below the atomics' noise against far targets: every run is held to the bound
    for attempt in range(3):    # three independent pairs
        res = {}
        for flag in (1, 0):
            monkeypatch.setattr(harness, "_LAST_GF_ZERO", flag)
            for p in rt.parameters():
                p.grad = None
            r = rt.frame()
            res[flag] = (float(r.loss), [p.grad.clone() for p in rt.parameters()])
        dl = abs(res[1][0] - res[0][0]) / abs(res[0][0])
        dg = max(float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(res[1][1], res[0][1]))
        assert all(float(b.abs().max()) > 0 for b in res[0][1]) and res[0][0] > 1e-3
        assert measured(dl, "rel loss, last plasticity adjoint skipped (far targets)") < BOUND_FRAME_LOSS
        assert measured(dg, "rel max of the LoRA gradients (far targets)") < BOUND_FRAME_GRAD


def test_reverse_sweep_refuses_caches_whose_forward_skipped_the_last_records(monkeypatch):
    """nm_rollout_cfg.last_gF_zero on the forward side only: the last plasticity step's SVD / activation records were not written,
    the pooled cache buffers still hold an earlier frame's - a reverse sweep that asks for them gets NM_ERR_INVALID instead of
    stale records.  The library keeps the skipped buffers in a host-side set; a forward sweep that writes the
    records takes them out again."""
    from neuma_amd import synth, harness, _lib
    from neuma_amd.harness import SceneRuntime
    rt = SceneRuntime(synth.make_scene("tiny", override=dict(S=4, V=2)), dev(), fused=True)
    rt.set_start_state("deformed")
    rt.make_ground_truth()
    rt.frame()                                  # (both sides with the flag: fine)
    real = harness._frame_backward

    def other_word(rt_, fs):
        monkeypatch.setattr(harness, "_LAST_GF_ZERO", 0)
        return real(rt_, fs)

    monkeypatch.setattr(harness, "_LAST_GF_ZERO", 1)
    monkeypatch.setattr(harness, "_frame_backward", other_word)
    with pytest.raises(_lib.NeumaHipError, match="last_gF_zero"):
        rt.frame()
    monkeypatch.setattr(harness, "_frame_backward", real)
    for flag in (0, 1):                         # neither side / both sides: the same buffers serve again
        monkeypatch.setattr(harness, "_LAST_GF_ZERO", flag)
        for p in rt.parameters():
            p.grad = None
        r = rt.frame()
        assert torch.isfinite(r.loss) and all(torch.isfinite(p.grad).all() for p in rt.parameters())
