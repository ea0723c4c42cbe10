#!/usr/bin/env python3
"""Can the reference's training step (main.py:67-112 body) be captured in a HIP graph as it stands?  Probe: forward + loss +
backward of the 1024-ray step under torch.cuda.graph, replayed; time per replay against the eager step.

    python tools/experiments/graph_probe.py [--precision bf16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import nerf, render_utils, synth, utils  # noqa: E402

ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    models = []
    for seed in (0, 10):
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**ARCH, "skips": (4,)}))
        m.precision = args.precision
        models.append(m.to(dev))
    r = render_utils.Renderer(perturb=1.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True,
                              raw_noise_std=0.0, near=2.0, far=6.0)
    rng = np.random.default_rng(0)
    K = synth.lego_intrinsics(400, 400)
    idx = rng.choice(160000, size=args.rays, replace=False)
    ro, rd = synth.rays_np(400, 400, K, synth.LEGO_C2W, idx)
    rays = (torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev))
    target = torch.rand(args.rays, 3, device=dev)
    params = list(models[0].parameters()) + list(models[1].parameters())

    def fwd_bwd():
        for p in params:
            p.grad = None
        rgb, disp, acc, extras = r.render(400, 400, K, models[0], models[1], chunk=32768, rays=rays, retraw=True)
        loss = utils.img2mse(rgb, target) + utils.img2mse(extras["rgb0"], target)
        loss.backward()
        return loss

    out = {"precision": args.precision}
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd_bwd()
    torch.cuda.synchronize()
    out["eager_ms"] = (time.perf_counter() - t0) / args.steps * 1e3
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = fwd_bwd()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        out["captured_loss"] = float(loss)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        host = (time.perf_counter() - t0) / args.steps * 1e3
        torch.cuda.synchronize()
        out["replay_ms"] = (time.perf_counter() - t0) / args.steps * 1e3
        out["replay_host_ms"] = host
        out["grad_norm"] = float(torch.stack([p.grad.norm() for p in params]).norm())
    except Exception as e:      # noqa: BLE001
        out["capture_error"] = repr(e)[:600]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
