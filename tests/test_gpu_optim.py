"""nerf_shared_amd.optim.Adam (-m gpu): the optimizer step of the reference's training loop (utils.py:163-172,
main.py:104) as one launch, against torch.optim.Adam on the same gradients."""
import copy
import os

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
