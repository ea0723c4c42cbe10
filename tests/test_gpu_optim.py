"""nerf_shared_amd.optim.Adam (-m gpu): the optimizer step of the reference's training loop (utils.py:163-172,
main.py:104) as one launch, against torch.optim.Adam on the same gradients."""
import copy
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("NERF_AMD_QUIET", "1")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a ROCm device"
    return torch.device("cuda:0")


SHAPES = [(256, 63), (256,), (256, 256), (1, 256), (1,), (128, 283), (3, 128), (3,), (5000,), (2049,)]


def make_params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(dev)) for s in SHAPES]


def set_grads(params, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    for i, p in enumerate(params):
        v = (torch.randn(*p.shape, generator=g) * (10.0 ** ((i % 5) - 3))).to(p.device)
        p.grad = None if i in skip else v


def max_rel(a, b):
    """largest difference relative to the tensor's scale (moments of mixed-sign gradients cancel towards zero)"""
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_matches_torch_adam(dev, wd):
    """15 steps on ten tensors (sizes 1 ... 65536, gradient scales 1e-3 ... 10): parameters and both moments follow
    torch.optim.Adam (single-tensor implementation, the reference's default) to fp32 round-off."""
    from nerf_shared_amd import optim
    ours, ref = make_params(dev, 0), make_params(dev, 0)
    o1 = optim.Adam(ours, lr=5e-4, betas=(0.9, 0.999), weight_decay=wd)
    o2 = torch.optim.Adam(ref, lr=5e-4, betas=(0.9, 0.999), weight_decay=wd, foreach=False, fused=False)
    for step in range(15):
        set_grads(ours, 100 + step)
        set_grads(ref, 100 + step)
        if step == 7:                      # the reference's exponential decay writes a float into param_groups (main.py:109-112)
            for o in (o1, o2):
                o.param_groups[0]["lr"] = 5e-4 * 0.1 ** (step / 250000)
        o1.step()
        o2.step()
    worst = 0.0
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (a.shape, float((a - b).abs().max()))
        worst = max(worst, max_rel(o1.state[a]["exp_avg"], o2.state[b]["exp_avg"]),
                    max_rel(o1.state[a]["exp_avg_sq"], o2.state[b]["exp_avg_sq"]))
    assert worst < 2e-6, worst
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert sd1["param_groups"][0]["lr"] == sd2["param_groups"][0]["lr"]
    for k in sd2["state"]:
        assert float(sd1["state"][k]["step"]) == float(sd2["state"][k]["step"]) == 15.0


def test_adam_state_moves_between_implementations(dev):
    """state_dict of one loads into the other (the reference's checkpoints carry `optimizer_state_dict`,
    utils.py:444-456) and training continues identically; parameters without a gradient are left alone and keep
    their own step count."""
    from nerf_shared_amd import optim
    ours, ref = make_params(dev, 1), make_params(dev, 1)
    o1 = optim.Adam(ours, lr=1e-3)
    o2 = torch.optim.Adam(ref, lr=1e-3, foreach=False, fused=False)
    for step in range(4):
        skip = (2, 5) if step == 2 else ()
        set_grads(ours, step, skip)
        set_grads(ref, step, skip)
        o1.step()
        o2.step()
    sd1 = o1.state_dict()
    assert float(sd1["state"][2]["step"]) == 3.0 and float(sd1["state"][0]["step"]) == 4.0
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    # ours -> torch, torch -> ours
    a_params, b_params = [torch.nn.Parameter(p.detach().clone()) for p in ours], [torch.nn.Parameter(p.detach().clone()) for p in ref]
    oa = torch.optim.Adam(a_params, lr=1e-3, foreach=False, fused=False)
    oa.load_state_dict(copy.deepcopy(sd1))
    ob = optim.Adam(b_params, lr=1e-3)
    ob.load_state_dict(copy.deepcopy(o2.state_dict()))
    for step in range(3):
        set_grads(a_params, 50 + step)
        set_grads(b_params, 50 + step)
        oa.step()
        ob.step()
    for a, b in zip(a_params, b_params):
        assert torch.allclose(a, b, rtol=3e-6, atol=2e-7), float((a - b).abs().max())
    assert float(ob.state_dict()["state"][2]["step"]) == 6.0 and float(ob.state_dict()["state"][0]["step"]) == 7.0


def test_adam_with_a_parameter_that_never_gets_a_gradient(dev):
    """The reference's NeRF(use_viewdirs=False) carries an unused views_linears.0 (nerf.py:83): its .grad stays None on
    every step.  The optimizer leaves it alone (no state, like torch) and still steps the others together."""
    from nerf_shared_amd import optim
    ours, ref = make_params(dev, 4), make_params(dev, 4)
    o1 = optim.Adam(ours, lr=1e-3)
    o2 = torch.optim.Adam(ref, lr=1e-3, foreach=False, fused=False)
    for step in range(5):
        set_grads(ours, step, skip=(3,))
        set_grads(ref, step, skip=(3,))
        o1.step()
        o2.step()
    assert 0 in o1._together and o1._together[0]["step"] == 5            # the cached one-launch path was taken
    for i, (a, b) in enumerate(zip(ours, ref)):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), i
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert sorted(sd1["state"]) == sorted(sd2["state"]) and 3 not in sd1["state"]
    assert all(float(v["step"]) == 5.0 for v in sd1["state"].values())


def test_adam_refuses_what_it_does_not_cover(dev):
    from nerf_shared_amd import _lib, optim
    with pytest.raises(_lib.NerfAmdError):
        optim.Adam(make_params(dev, 2), amsgrad=True)
    cpu = [torch.nn.Parameter(torch.zeros(4))]
    o = optim.Adam(cpu)
    cpu[0].grad = torch.ones(4)
    with pytest.raises(_lib.NerfAmdError):
        o.step()


def test_get_optimizer_returns_the_library_adam_and_models_repack(dev):
    """utils.get_optimizer on GPU models: the library's Adam; a step marks the fields' packed weights stale so the next
    forward runs on the new parameters."""
    from types import SimpleNamespace

    from nerf_shared_amd import nerf, optim, utils
    arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
    c, f = nerf.NeRF(**arch).to(dev), nerf.NeRF(**arch).to(dev)
    opt = utils.get_optimizer(c, f, SimpleNamespace(lrate=5e-4))
    assert isinstance(opt, optim.Adam) and isinstance(opt, torch.optim.Adam)
    pts, vd = torch.rand(8, 4, 3, device=dev), torch.nn.functional.normalize(torch.randn(8, 3, device=dev), dim=-1)
    out0 = c(pts, vd)
    out0.square().mean().backward()
    with torch.no_grad():
        before = c(pts, vd)
    opt.step()
    with torch.no_grad():
        after = c(pts, vd)
    assert torch.equal(before, out0.detach()) and not torch.equal(before, after)


# ---- the other small pieces of the training loop that run as one launch each ------------------------------------------
@pytest.mark.parametrize("n", [1, 3072, 16384, 16385, 1_920_000])
def test_img2mse_matches_the_reference_expression(dev, n):
    """utils.img2mse (utils.py:24) through the library: value within fp32 summation error of torch.mean((x - y) ** 2)
    (fp64 referee), gradients equal to autograd's on the expression to round-off; both launch shapes (one block / partials)."""
    from nerf_shared_amd import utils
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, generator=g).to(dev).requires_grad_(True)
    y = torch.rand(n, generator=g).to(dev).requires_grad_(True)
    ours = utils.img2mse(x, y)
    (3.0 * ours).backward()
    gx, gy = x.grad.clone(), y.grad.clone()
    x.grad = y.grad = None
    ref = torch.mean((x - y) ** 2)
    (3.0 * ref).backward()
    exact = float(((x.detach().double() - y.detach().double()) ** 2).mean())
    assert abs(float(ours) - exact) <= 2e-6 * exact + 1e-12, (float(ours), exact)
    assert abs(float(ours) - exact) <= abs(float(ref) - exact) + 1e-6 * exact         # no worse than torch's own reduction
    assert torch.allclose(gx, x.grad, rtol=1e-6, atol=1e-12) and torch.allclose(gy, y.grad, rtol=1e-6, atol=1e-12)
    # shapes as the loop uses them, and the fall-through for what the kernel does not take
    a, b = torch.rand(64, 3, device=dev), torch.rand(64, 3, device=dev)
    assert float(utils.img2mse(a, b)) == pytest.approx(float(torch.mean((a - b) ** 2)), rel=1e-6)
    assert float(utils.img2mse(a, b[:1])) == pytest.approx(float(torch.mean((a - b[:1]) ** 2)), rel=1e-6)   # broadcasting


@pytest.mark.parametrize("use_viewdirs,ndc", [(True, False), (False, False), (True, True)])
def test_render_rays_argument_batch_assembly(dev, use_viewdirs, ndc):
    """Renderer.render(rays=...) assembles its [N, 8|11] batch in one launch when no gradient flows into the rays; the
    reference's expression (render_utils.py:205-222: normalise, reshape, ones_like * near/far, two cats) is what runs when
    one does.  Same batch bit for bit, unit view directions included (the kernel follows torch.norm's operation order on
    this build, tools/micro/norm_probe.py)."""
    from nerf_shared_amd import render_utils
    captured = []

    class Probe(render_utils.Renderer):
        def render_batch(self, coarse_model, fine_model, rays_flat, chunk=1024 * 32, retraw=False):
            captured.append(rays_flat.detach().clone())
            n = rays_flat.shape[0]
            z = torch.zeros(n, device=rays_flat.device)
            return {"rgb_map": torch.zeros(n, 3, device=rays_flat.device), "disp_map": z, "acc_map": z}

    r = Probe(use_viewdirs=use_viewdirs, ndc=ndc, near=2.0 if not ndc else 0.0, far=6.0 if not ndc else 1.0)
    g = torch.Generator().manual_seed(3)
    H, W = 13, 17
    o = (torch.randn(H, W, 3, generator=g) * 2).to(dev)
    d = torch.randn(H, W, 3, generator=g).to(dev)
    d[..., 2] = -d[..., 2].abs() - 0.5
    K = [[20.0, 0, W / 2], [0, 20.0, H / 2], [0, 0, 1]]
    r.render(H, W, K, None, None, rays=(o, d))
    d_tracked = d.clone().requires_grad_(True)
    r.render(H, W, K, None, None, rays=(o, d_tracked))
    fused, ref = captured
    assert fused.shape == ref.shape == (H * W, 11 if use_viewdirs else 8)
    assert torch.equal(fused, ref)


@pytest.mark.parametrize("precision", ["bf16", "fp32_split", "fp32_small_architecture"])
def test_captured_train_step_equals_the_eager_loop(dev, precision):
    """utils.CapturedTrainStep: the body of main.py:77-104 captured in a HIP graph and replayed -- with a new batch and the
    loop's learning-rate decay (main.py:108-112) between replays -- follows the same loop run eagerly: the same losses, the
    same parameters afterwards, the same optimizer step count (deterministic renderer: perturb 0, so the two see the same
    draws), and gradients are left in .grad after every call."""
    from nerf_shared_amd import nerf, optim, render_utils, synth, utils
    arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
    if precision == "fp32_small_architecture":             # netdepth 4, netwidth 128: the exact-fp32 training path (train_f32.hip)
        arch, precision = dict(arch, D=4, W=128, skips=[2]), "fp32"
    cfg = dict(perturb=0.0, N_importance=32, N_samples=32, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, near=2.0, far=6.0)
    K = synth.lego_intrinsics(400, 400)
    rng = np.random.default_rng(5)
    N, steps = 256, 9
    batches = []
    for _ in range(steps + 3):
        idx = rng.choice(160000, size=N, replace=False)
        ro, rd = synth.rays_np(400, 400, K, synth.LEGO_C2W, idx)
        batches.append((torch.from_numpy(np.stack([ro, rd], 0)).to(dev), torch.from_numpy(rng.uniform(0, 1, size=(N, 3)).astype(np.float32)).to(dev)))

    def fresh():
        ms = []
        for seed in (0, 10):
            m = nerf.NeRF(**arch)
            m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**arch, "skips": tuple(arch["skips"])}))
            m.precision = precision
            ms.append(m.to(dev))
        return ms, optim.Adam(list(ms[0].parameters()) + list(ms[1].parameters()), lr=5e-4, betas=(0.9, 0.999))

    lr_at = lambda i: 5e-4 * (0.1 ** (i / 20.0))           # noqa: E731  (a fast decay, so that a frozen lr would show)
    r = render_utils.Renderer(**cfg)
    # eager loop: 3 warm-up steps on batches 0-2 (what the capture's constructor does on its zero buffers is replaced below)
    (mc, mf), opt = fresh()
    eager = []
    for i in range(steps):
        for g in opt.param_groups:
            g["lr"] = lr_at(i)
        opt.zero_grad()
        rays, tgt = batches[i]
        rgb, _, _, ex = r.render_from_rays(400, 400, K, 32768, rays, mc, mf, retraw=True)
        loss = utils.img2mse(rgb, tgt) + utils.img2mse(ex["rgb0"], tgt)
        loss.backward()
        opt.step()
        eager.append(float(loss))
    # captured loop: constructing the step leaves the training state untouched
    (gc, gf), gopt = fresh()
    start = [p.detach().clone() for p in list(gc.parameters()) + list(gf.parameters())]
    step = utils.CapturedTrainStep(r, 400, 400, K, 32768, gc, gf, gopt, N)
    assert all(torch.equal(p, s0) for p, s0 in zip(list(gc.parameters()) + list(gf.parameters()), start))
    assert gopt._together[0]["step"] == 0 and all(not st["exp_avg"].any() for st in gopt.state.values())
    got = []
    for i in range(steps):
        for g in gopt.param_groups:
            g["lr"] = lr_at(i)
        loss = step(*batches[i])
        got.append(float(loss))
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in gc.parameters())
    print("eager   ", ["%.6f" % v for v in eager])
    print("captured", ["%.6f" % v for v in got])
    np.testing.assert_allclose(got, eager, rtol=2e-5)
    for a, b in zip(list(gc.parameters()) + list(gf.parameters()), list(mc.parameters()) + list(mf.parameters())):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < (2e-3 if precision == "bf16" else 2e-5)
    assert gopt._together[0]["step"] == steps == int(gopt.state_dict()["state"][0]["step"])
    assert np.isfinite(float(step.psnr))
    # inference after the captured steps sees the updated weights (the pack is marked stale by every replay)
    with torch.no_grad():
        a = r.render_from_rays(400, 400, K, 32768, batches[0][0], gc, gf, retraw=False)[0]
        b = r.render_from_rays(400, 400, K, 32768, batches[0][0], mc, mf, retraw=False)[0]
    assert float((a - b).abs().max()) < (5e-2 if precision == "bf16" else 1e-3)
