"""GPU tests (-m gpu) of the exact-fp32 training path (csrc/train_f32.hip, NERF_AMD_PREC_FP32 training): forward with
saves, dX chain and weight / bias gradients for architectures OUTSIDE the fused 8 x 256 family -- what the reference's
netdepth / netwidth / skips flags produce (config_parser.py:18-25, nerf.py:62-94) -- against torch.autograd on the fp32
oracle.  fp32 MFMA chains on both sides of every product: gradients agree to ~1e-6."""
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("NERF_AMD_QUIET", "1")
pytestmark = pytest.mark.gpu

from nerf_shared_amd import synth  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402
from test_gpu_backward import BASE, _batch, rel_err  # noqa: E402

ARCHS = {
    "d4_w128_skip2": dict(D=4, W=128, output_ch=5, skips=[2], use_viewdirs=True, multires=10, multires_views=4),
    "d2_w64_noskip_novd": dict(D=2, W=64, output_ch=5, skips=[], use_viewdirs=False, multires=6, multires_views=4),
    "d6_w96_skips13": dict(D=6, W=96, output_ch=4, skips=[1, 3], use_viewdirs=True, multires=8, multires_views=2),
    "d3_w320_skip0": dict(D=3, W=320, output_ch=3, skips=[0], use_viewdirs=False, multires=4, multires_views=4),
    "d8_w256_identity_embed": dict(D=8, W=256, output_ch=4, skips=[4], use_viewdirs=True, multires=10, multires_views=4, i_embed=-1),
    # wide models: 32 points per workgroup (the rows of 64 would not fit LDS), several 32-row tiles per wave
    "d2_w600_wide": dict(D=2, W=600, output_ch=4, skips=[], use_viewdirs=True, multires=10, multires_views=4),
    "d2_w1024_widest": dict(D=2, W=1024, output_ch=4, skips=[0], use_viewdirs=True, multires=10, multires_views=4),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def build(dev, seed, sharpen, arch, precision="fp32"):
    from nerf_shared_amd import nerf
    kw = {k: v for k, v in arch.items()}
    sd = synth.torch_state_dict(seed, sharpen, **{**kw, "skips": tuple(kw["skips"])})
    m = nerf.NeRF(**kw)
    m.load_state_dict(sd)
    m = m.to(dev)
    m.precision = precision
    cpu = {k: v.clone().requires_grad_(True) for k, v in O.state_dict_to_torch(sd).items()}
    return m, cpu


def check(m, cpu, gate=2e-5):
    worst = 0.0
    for name, p in m.named_parameters():
        if cpu[name].grad is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        e = rel_err(p.grad.cpu(), cpu[name].grad)
        worst = max(worst, e)
        if e > 0.1 * gate:
            print("   %-28s rel-L2 %.2e" % (name, e))
        assert e < gate, (name, e)
    return worst


@pytest.mark.parametrize("name", sorted(ARCHS))
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_field_gradients_of_any_architecture(dev, name, precision):
    """NeRF.forward(inputs, viewdirs) -> a random linear functional -> backward, for models the fused kernels do not cover,
    whatever precision they render in: parameter gradients against fp32 autograd on the oracle, and the training forward's
    raw against the oracle's."""
    arch = ARCHS[name]
    rng = np.random.default_rng(3)
    R, S = 37, 11                                  # 407 points: a ragged last workgroup
    pts = torch.from_numpy(rng.uniform(-2, 2, size=(R, S, 3)).astype(np.float32))
    vd = None
    if arch["use_viewdirs"]:
        vd = torch.from_numpy(rng.normal(size=(R, 3)).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
    out_ch = 4 if arch["use_viewdirs"] else arch["output_ch"]
    coef = torch.from_numpy(rng.normal(size=(R, S, out_ch)).astype(np.float32))
    m, cpu = build(dev, 2, 1.5, arch, precision)
    out = m(pts.to(dev), None if vd is None else vd.to(dev))
    ref = O.nerf_forward(cpu, O.Arch(**arch), pts, vd)
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-5
    (out * coef.to(dev)).sum().backward()
    (ref * coef).sum().backward()
    print(name, precision, "worst parameter-gradient rel-L2 vs fp32 autograd: %.2e" % check(m, cpu))


def test_training_step_and_adam_on_a_small_architecture(dev):
    """The reference's loop (main.py:67-112) for netdepth=4, netwidth=128: render -> mse(rgb) + mse(rgb0) -> backward ->
    Adam, 6 steps against the same loop on the oracle (the fine pass of the gradient comparison on the run's depths)."""
    from nerf_shared_amd import optim, render_utils, utils
    from test_gpu_split_backward import oracle_two_pass
    arch = ARCHS["d4_w128_skip2"]
    batch, target = _batch(96, 5)
    cfg = dict(BASE, N_samples=24, N_importance=24)
    r = render_utils.Renderer(**cfg)
    mc, cc = build(dev, 1, 2.0, arch)
    mf, cf = build(dev, 11, 2.0, arch)
    out = r.render_rays(batch.to(dev), mc, mf, retweights=True)
    t = target.to(dev)
    (((out["rgb_map"] - t) ** 2).mean() + ((out["rgb0"] - t) ** 2).mean()).backward()
    rgb, rgb0 = oracle_two_pass(cfg, batch, (cc, O.Arch(**arch)), (cf, O.Arch(**arch)), out["z_vals"].detach().cpu())
    (((rgb - target) ** 2).mean() + ((rgb0 - target) ** 2).mean()).backward()
    # (a ReLU within rounding of zero gives the two fp32 evaluations different masks for that unit: 1e-4-class differences
    # with this few points, as in test_gpu_split_backward.py; the gate is that file's)
    print("coarse %.2e  fine %.2e" % (check(mc, cc, 1e-3), check(mf, cf, 1e-3)))
    # a few optimizer steps: the loss goes down and follows the oracle's
    mc, cc = build(dev, 1, 2.0, arch)
    mf, cf = build(dev, 11, 2.0, arch)
    opt = optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
    opt2 = torch.optim.Adam(list(cc.values()) + list(cf.values()), lr=5e-4)
    ocfg = O.RenderCfg(**cfg)
    b = batch.to(dev)
    losses, ref = [], []
    for _ in range(6):
        opt.zero_grad()
        rgb, _, _, extras = r.render(400, 400, None, mc, mf, chunk=64, rays=(b[:, 0:3], b[:, 3:6]), retraw=True)
        loss = utils.img2mse(rgb, t) + utils.img2mse(extras["rgb0"], t)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        opt2.zero_grad()
        o = O.render_rays(ocfg, batch, (cc, O.Arch(**arch)), (cf, O.Arch(**arch)))
        l2 = ((o["rgb_map"] - target) ** 2).mean() + ((o["rgb0"] - target) ** 2).mean()
        l2.backward()
        opt2.step()
        ref.append(float(l2))
    print("gpu   ", ["%.6f" % v for v in losses])
    print("oracle", ["%.6f" % v for v in ref])
    assert losses[-1] < losses[0]
    np.testing.assert_allclose(losses, ref, rtol=2e-3)


@pytest.mark.parametrize("name", ["d4_w128_skip2", "d6_w96_skips13", "d8_w256_identity_embed", "d2_w64_noskip_novd"])
def test_point_and_ray_gradients_of_any_architecture(dev, name):
    """Pose estimation on a model outside the fused family (demo_est_rel_pose.py:87-98): dL/d(points, view directions) of the
    field, and dL/d(rays_o, rays_d) through Renderer.render_rays with frozen networks, against fp32 autograd on the oracle
    (fine pass on the run's own depths)."""
    from nerf_shared_amd import render_utils
    from test_gpu_split_backward import oracle_two_pass
    arch = ARCHS[name]
    vdirs = arch["use_viewdirs"]
    rng = np.random.default_rng(21)
    pts = torch.from_numpy(rng.uniform(-2, 2, size=(40, 9, 3)).astype(np.float32))
    vd = torch.from_numpy(rng.normal(size=(40, 3)).astype(np.float32)) if vdirs else None
    out_ch = 4 if vdirs else arch["output_ch"]
    coef = torch.from_numpy(rng.normal(size=(40, 9, out_ch)).astype(np.float32))
    m, cpu = build(dev, 1, 2.0, arch)
    m.requires_grad_(False)
    frozen = {k: v.detach() for k, v in cpu.items()}
    p_gpu = pts.to(dev).requires_grad_(True)
    v_gpu = vd.to(dev).requires_grad_(True) if vdirs else None
    (m(p_gpu, v_gpu) * coef.to(dev)).sum().backward()
    p_cpu = pts.clone().requires_grad_(True)
    v_cpu = vd.clone().requires_grad_(True) if vdirs else None
    (O.nerf_forward(frozen, O.Arch(**arch), p_cpu, v_cpu) * coef).sum().backward()
    e = rel_err(p_gpu.grad.cpu(), p_cpu.grad)
    print(name, "dL/dpts rel-L2 %.2e" % e, ("dL/dviewdirs %.2e" % rel_err(v_gpu.grad.cpu(), v_cpu.grad)) if vdirs else "")
    assert e < 2e-5 and (not vdirs or rel_err(v_gpu.grad.cpu(), v_cpu.grad) < 2e-5)
    # rays through the renderer, both passes, parameters AND rays at once for the fine model
    batch, target = _batch(48, 7)
    if not vdirs:
        batch = batch[:, :8].contiguous()
    cfg = dict(BASE, N_samples=16, N_importance=24, use_viewdirs=vdirs)
    r = render_utils.Renderer(**cfg)
    mf, cf = build(dev, 11, 2.0, arch)

    def assemble(o, d):
        cols = [o, d, 2.0 * torch.ones_like(d[:, :1]), 6.0 * torch.ones_like(d[:, :1])]
        if vdirs:
            cols.append(d / torch.norm(d, dim=-1, keepdim=True))
        return torch.cat(cols, -1)

    ro = batch[:, 0:3].clone().to(dev).requires_grad_(True)
    rd = (batch[:, 3:6] * 1.3).clone().to(dev).requires_grad_(True)
    out = r.render_rays(assemble(ro, rd), m, mf, retweights=True)
    t = target.to(dev)
    (((out["rgb_map"] - t) ** 2).mean() + ((out["rgb0"] - t) ** 2).mean()).backward()
    o = batch[:, 0:3].clone().requires_grad_(True)
    d = (batch[:, 3:6] * 1.3).clone().requires_grad_(True)
    rgb, rgb0 = oracle_two_pass(cfg, assemble(o, d), (frozen, O.Arch(**arch)), (cf, O.Arch(**arch)), out["z_vals"].detach().cpu())
    (((rgb - target) ** 2).mean() + ((rgb0 - target) ** 2).mean()).backward()
    eo, ed = rel_err(ro.grad.cpu(), o.grad), rel_err(rd.grad.cpu(), d.grad)
    print(name, "dL/drays_o %.2e  dL/drays_d %.2e  fine parameters %.2e" % (eo, ed, check(mf, cf, 1e-3)))
    assert eo < 1e-4 and ed < 1e-4
    assert all(p.grad is None for p in m.parameters())


@pytest.mark.parametrize("i", range(24))
def test_random_architectures_train_on_the_exact_path(dev, i):
    """The random networks of tests/test_gpu_fuzz.py::test_random_architectures_on_the_exact_kernel (depth 1..9, widths 2..777,
    skips anywhere, identity embedding) under loss.backward(): parameter gradients -- and, every second case, the gradients with
    respect to the points and view directions -- of a random linear functional of NeRF.forward against fp32 autograd on the
    oracle (csrc/train_f32.hip: any architecture the constructor accepts)."""
    import test_gpu_fuzz as F
    import test_gpu_parity as P
    arch = F.draw_arch(i)
    rng = F._rng(17000, i)
    shape = [(1, 1), (3, 5), (41, 7), (200, 13)][i % 4]
    pts = torch.from_numpy(rng.uniform(-2, 2, size=shape + (3,)).astype(np.float32))
    vd = None
    if arch["use_viewdirs"]:
        vd = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(shape[0], 3)).astype(np.float32)), dim=-1)
    sharpen = 2.0 if arch["W"] >= 31 else 1.0
    sd, oarch = P.cpu_model(i, sharpen, **arch)
    cpu = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with_inputs = i % 2 == 1
    p_cpu = pts.clone().requires_grad_(with_inputs)
    v_cpu = vd.clone().requires_grad_(with_inputs) if vd is not None else None
    ref = P.O.nerf_forward(cpu, oarch, p_cpu, v_cpu)
    coef = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * coef).sum().backward()
    m = P.gpu_model(dev, i, sharpen, "fp32", **arch).requires_grad_(True)
    p_gpu = pts.to(dev).requires_grad_(with_inputs)
    v_gpu = vd.to(dev).requires_grad_(with_inputs) if vd is not None else None
    out = m(p_gpu, v_gpu)
    P.close(out, ref, atol=1e-4 * max(1.0, float(ref.abs().max())), rtol=1e-4)
    (out * coef.to(dev)).sum().backward()

    def same(tag, got, want):
        if want is None:
            assert got is None, (arch, tag)
            return
        assert got is not None and bool(torch.isfinite(got).all()), (arch, tag)
        n = float(want.norm())
        d = float((got.detach().cpu() - want).norm())
        # (a ReLU within rounding of zero flips between the two fp32 evaluations -- about one unit per case is expected for the
        # deep, wide, sharpened draws -- and ONE flipped unit moves a first-layer gradient by ~1 / sqrt(points x width): 1e-3 for
        # 2600 points x 777 units (measured: case 23), a visible fraction for a handful of points and a width of 2 or 3.
        # tests/test_gpu_train_f32.py holds the tight gates (1e-6) on draws without such a unit.)
        assert d <= (5e-3 if arch["W"] >= 31 else 5e-2) * n + 1e-7, (arch, tag, d, n)

    for name, p in m.named_parameters():
        same(name, p.grad, cpu[name].grad)
    if with_inputs:
        same("pts", p_gpu.grad, p_cpu.grad)
        if vd is not None:
            same("viewdirs", v_gpu.grad, v_cpu.grad)
