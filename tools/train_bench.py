#!/usr/bin/env python3
"""Time one training step of the reference's loop (main.py:67-112 without the data loader):
N_rand rays -> render (64+128, two 8x256 view-branch models) -> mse(rgb)+mse(rgb0) -> backward -> Adam.

    python tools/train_bench.py [--rays 1024] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import nerf, render_utils, synth, utils  # noqa: E402

ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)      # N_rand of configs/lego.txt:15
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--adam", choices=["foreach", "fused", "amd"], default="amd",
                    help="amd: nerf_shared_amd.optim.Adam (what utils.get_optimizer returns); foreach / fused: torch.optim.Adam")
    ap.add_argument("--tuning", type=int, default=0, help="nerf_amd_set_tuning(0, value): 50 = round-1 weight-gradient kernel")
    ap.add_argument("--precision", choices=["bf16", "fp32_split", "fp32"], default="bf16",
                    help="arithmetic of the field and of its backward pass (fp32 trains on the split-precision kernels)")
    ap.add_argument("--netdepth", type=int, default=8, help="anything but 8 x 256 with skip 4 trains on the exact-fp32 path (csrc/train_f32.hip)")
    ap.add_argument("--netwidth", type=int, default=256)
    ap.add_argument("--skip", type=int, default=4)
    ap.add_argument("--multires", type=int, default=10)
    ap.add_argument("--multires-views", type=int, default=4, help="15 / 6: configs/stonehenge.txt:18-19")
    ap.add_argument("--graph", action="store_true", help="utils.CapturedTrainStep: the step captured in a HIP graph and replayed")
    ap.add_argument("--cprofile", action="store_true", help="print the host-side profile of the timed steps (cProfile)")
    args = ap.parse_args()
    from nerf_shared_amd import _lib
    _lib.check(_lib.lib.nerf_amd_set_tuning(0, args.tuning), "set_tuning")
    dev = torch.device("cuda:0")
    ARCH.update(multires=args.multires, multires_views=args.multires_views, D=args.netdepth, W=args.netwidth, skips=[args.skip])
    models = []
    for seed in (0, 10):
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**ARCH, "skips": tuple(ARCH["skips"])}))
        m.precision = args.precision
        models.append(m.to(dev))
    r = render_utils.Renderer(perturb=1.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True,
                              raw_noise_std=0.0, near=2.0, far=6.0)
    params = list(models[0].parameters()) + list(models[1].parameters())
    if args.adam == "amd":
        from nerf_shared_amd import optim
        opt = optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
    else:
        opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999), fused=args.adam == "fused")
    rng = np.random.default_rng(0)
    K = synth.lego_intrinsics(400, 400)
    idx = rng.choice(160000, size=args.rays, replace=False)
    ro, rd = synth.rays_np(400, 400, K, synth.LEGO_C2W, idx)
    rays = (torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev))
    target = torch.rand(args.rays, 3, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        rgb, disp, acc, extras = r.render(400, 400, K, models[0], models[1], chunk=32768, rays=rays, retraw=True)
        loss = utils.img2mse(rgb, target) + utils.img2mse(extras["rgb0"], target)       # main.py:93-98
        loss.backward()
        opt.step()
        return loss

    if args.graph:
        if args.adam != "amd":
            raise SystemExit("--graph needs --adam amd")
        captured = utils.CapturedTrainStep(r, 400, 400, K, 32768, models[0], models[1], opt, args.rays)
        rays_t = torch.stack(list(rays), 0)

        def step():                                  # noqa: F811  (one replay + the loop's LR decay, main.py:108-112)
            loss = captured(rays_t, target)
            for g in opt.param_groups:
                g["lr"] = 5e-4 * (0.1 ** (captured.optimizer._together[0]["step"] / 250000.0))
            return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.cprofile:
        import cProfile
        import pstats
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(args.steps):
            step()
        prof.disable()
        torch.cuda.synchronize()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
        pstats.Stats(prof).sort_stats("tottime").print_stats(30)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host = (time.perf_counter() - t0) / args.steps        # time to ENQUEUE a step: close to ms_per_step = host-bound
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    pts = args.rays * 256
    print(json.dumps({"rays_per_step": args.rays, "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt,
                      "host_enqueue_ms_per_step": host * 1e3,
                      "rays_per_s": args.rays / dt, "loss": float(loss), "adam": args.adam, "precision": args.precision, "graph": bool(args.graph),
                      "model_tflops": pts * 1186816 * 3 / dt / 1e12}))


if __name__ == "__main__":
    main()
