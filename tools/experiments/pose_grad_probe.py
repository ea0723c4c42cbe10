#!/usr/bin/env python3
"""Which rays carry the difference between the split-precision ray gradients and fp32 autograd on the oracle (fine pass on the
run's own depths) in the pose-optimisation test's setting?  (GPU box; measurement tooling: imports the oracle)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import nerf, render_utils, synth, utils  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402
from tools import train_demo  # noqa: E402

VD = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
BASE = dict(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False, near=2.0, far=6.0)


def main():
    dev = torch.device("cuda:0")
    run = train_demo.run(steps=300, res=48, views=8, verbose=False)
    H = W = 14
    K = synth.lego_intrinsics(H, W)
    cfg = dict(BASE, N_samples=32, N_importance=48)
    r = render_utils.Renderer(**cfg)
    pairs = []
    for trained in run[1][:2]:
        sd = {k: v.detach().cpu().clone() for k, v in trained.state_dict().items()}
        m = nerf.NeRF(**VD)
        m.load_state_dict(sd)
        m = m.to(dev)
        m.precision = "fp32_split"
        m.requires_grad_(False)
        pairs.append((m, {k: v.detach() for k, v in O.state_dict_to_torch(sd).items()}))
    (mc, co), (mf, fo) = pairs
    arch = O.Arch(**VD)
    c2w = torch.from_numpy(synth.LEGO_C2W.astype(np.float32))
    def pose_of(w, dt, base):
        z = torch.zeros((), dtype=w.dtype, device=w.device)
        Wx = torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]), torch.stack([-w[1], w[0], z])])
        return torch.cat([torch.matrix_exp(Wx) @ base[:3, :3], (base[:3, 3] + dt)[:, None]], 1)

    w0, dt0 = torch.tensor([0.02, -0.03, 0.015]), torch.tensor([0.06, -0.05, 0.04])
    if len(sys.argv) > 1:                          # one Adam step of lr 2e-3 from there, as the test's iterate 1
        sg = [float(v) for v in sys.argv[1:7]]
        w0 = w0 - 2e-3 * torch.tensor(sg[:3])
        dt0 = dt0 - 2e-3 * torch.tensor(sg[3:])
    c2w_p = pose_of(w0, dt0, c2w)
    ocfg = O.RenderCfg(**cfg)

    def assemble(ro, rd):
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        vdir = rd / torch.norm(rd, dim=-1, keepdim=True)
        return torch.cat([ro, rd, 2.0 * torch.ones_like(rd[:, :1]), 6.0 * torch.ones_like(rd[:, :1]), vdir], -1)

    with torch.no_grad():
        target = O.render_rays(ocfg, assemble(*O.get_rays(H, W, K, c2w)), (co, arch), (fo, arch))["rgb_map"]
    ro0, rd0 = O.get_rays(H, W, K, c2w_p)
    ro0, rd0 = ro0.reshape(-1, 3).contiguous(), rd0.reshape(-1, 3).contiguous()
    # GPU
    ro, rd = ro0.clone().to(dev).requires_grad_(True), rd0.clone().to(dev).requires_grad_(True)
    out = r.render_rays(assemble(ro, rd), mc, mf, retweights=True)
    utils.img2mse(out["rgb_map"], target.to(dev)).backward()
    z = out["z_vals"].detach().cpu()
    w_gpu = out["weights"].detach().cpu()
    res = {}
    for dtype in (torch.float32, torch.float64):
        o, d = ro0.clone().to(dtype).requires_grad_(True), rd0.clone().to(dtype).requires_grad_(True)
        b = assemble(o, d)
        cast = lambda sd: {k: v.to(dtype) for k, v in sd.items()}     # noqa: E731
        pts = b[:, None, 0:3] + b[:, None, 3:6] * z.to(dtype)[:, :, None]
        raw = O.nerf_forward(cast(fo), arch, pts, b[:, 8:11])
        rgb = O.raw2outputs(raw, z.to(dtype), b[:, 3:6], True, None)[0]
        ((rgb - target.to(dtype)) ** 2).mean().backward()
        res[dtype] = (o.grad.double(), d.grad.double(), rgb.detach().double())
    go, gd = ro.grad.cpu().double(), rd.grad.cpu().double()
    o32, d32, rgb32 = res[torch.float32]
    o64, d64, rgb64 = res[torch.float64]
    for name, g, a, b in (("rays_o", go, o32, o64), ("rays_d", gd, d32, d64)):
        print(name, "rel-L2 gpu vs fp32 %.2e   fp64 vs fp32 %.2e   gpu vs fp64 %.2e" % (
            float((g - a).norm() / a.norm()), float((b - a).norm() / a.norm()), float((g - b).norm() / b.norm())))
        print("   SUM over rays (what a pose gradient sees): gpu", g.sum(0).numpy(), "fp32", a.sum(0).numpy(), "fp64", b.sum(0).numpy())
        per = (g - a).norm(dim=1)
        per64 = (b - a).norm(dim=1)
        idx = torch.argsort(per, descending=True)[:6]
        for i in idx.tolist():
            print("   ray %3d  |g| %.3e  |gpu - fp32| %.3e  |fp64 - fp32| %.3e   rgb err %.1e  acc %.4f" % (
                i, float(a[i].norm()), float(per[i]), float(per64[i]), float((out["rgb_map"][i].detach().cpu().double() - rgb32[i]).abs().max()), float(w_gpu[i].sum())))


    # ---- the chain down to the six numbers: get_rays backward and the pose's own derivative
    print("-- down to the pose")
    outs = {}
    for tag, device, dtype in (("gpu", dev, torch.float32), ("cpu fp32", torch.device("cpu"), torch.float32), ("cpu fp64", torch.device("cpu"), torch.float64)):
        w = w0.clone().to(device=device, dtype=dtype).requires_grad_(True)
        dt = dt0.clone().to(device=device, dtype=dtype).requires_grad_(True)
        pose = pose_of(w, dt, c2w.to(device=device, dtype=dtype))
        pose.retain_grad()
        if tag == "gpu":
            o_, d_ = utils.get_rays(H, W, K, pose)
            o2 = r.render_rays(assemble(o_, d_), mc, mf, retweights=True)
            zz = o2["z_vals"].detach().cpu()
            print("  (rays from the GPU's get_rays: depths differ from the CPU-ray run's by at most %.2e in %d of %d places; rays by %.1e)" % (
                float((zz - z).abs().max()), int((zz != z).sum()), z.numel(), float((o_.reshape(-1, 3).cpu() - ro0).abs().max() + (d_.reshape(-1, 3).cpu() - rd0).abs().max())))
            z = zz
            utils.img2mse(o2["rgb_map"], target.to(dev)).backward()
        else:
            o_, d_ = O.get_rays(H, W, K, pose)
            b = assemble(o_, d_)
            cast = lambda sd: {k: v.to(dtype) for k, v in sd.items()}     # noqa: E731
            pts = b[:, None, 0:3] + b[:, None, 3:6] * z.to(dtype)[:, :, None]
            raw = O.nerf_forward(cast(fo), arch, pts, b[:, 8:11])
            rgb = O.raw2outputs(raw, z.to(dtype), b[:, 3:6], True, None)[0]
            ((rgb - target.to(dtype)) ** 2).mean().backward()
        outs[tag] = (pose.grad.detach().cpu().double(), torch.cat([w.grad, dt.grad]).detach().cpu().double())
        print("  %-8s dL/dc2w" % tag, outs[tag][0].numpy().round(7).tolist())
        print("  %-8s dL/d(w, dt)" % tag, outs[tag][1].numpy().tolist())
    for a in ("gpu", "cpu fp64"):
        print("  %s vs cpu fp32: dL/dc2w rel %.2e   dL/d(w,dt) rel %.2e" % (a, float((outs[a][0] - outs["cpu fp32"][0]).norm() / outs["cpu fp32"][0].norm()),
                                                                       float((outs[a][1] - outs["cpu fp32"][1]).norm() / outs["cpu fp32"][1].norm())))


if __name__ == "__main__":
    main()
