#!/usr/bin/env python3
"""When did each workgroup of the weight-gradient launch start and end?  (GPU box, scratch build)

    make -C nerf_shared_amd/csrc OUT=../../scratch_libs/libstamps.so OBJDIR=build_stamps EXTRA=-DNERF_AMD_X_DW_STAMPS
    NERF_AMD_LIB=$PWD/scratch_libs/libstamps.so python tools/micro/dw_stamps.py [--tuning 63]

Runs a few 1024-ray training steps (tools/train_bench.py's) and prints, for the LAST dw_multi_kernel launch (the coarse
network's 65536 points), per product: workgroups, their XCDs, first start / last end, spread of the ends (us).
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import _lib, nerf, render_utils, synth, utils  # noqa: E402

ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tuning", type=int, default=0)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--coarse-only", type=int, default=0, metavar="N_SAMPLES", help="N_importance 0: the stamped launch is this many samples per ray")
    ap.add_argument("--rgb0", action="store_true", help="the loss includes the coarse image (main.py:93-98): the stamped launch is the coarse network's, inside the full step")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--multires", type=int, default=10)
    ap.add_argument("--multires-views", type=int, default=4)
    args = ap.parse_args()
    _lib.check(_lib.lib.nerf_amd_set_tuning(0, args.tuning), "set_tuning")
    dev = torch.device("cuda:0")
    ARCH.update(multires=args.multires, multires_views=args.multires_views)
    models = []
    for seed in (0, 10):
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**ARCH, "skips": (4,)}))
        m.precision = args.precision
        models.append(m.to(dev))
    r = render_utils.Renderer(perturb=1.0, N_importance=0 if args.coarse_only else 128, N_samples=args.coarse_only or 64,
                              use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, near=2.0, far=6.0)
    rng = np.random.default_rng(0)
    K = synth.lego_intrinsics(400, 400)
    idx = rng.choice(160000, size=args.rays, replace=False)
    ro, rd = synth.rays_np(400, 400, K, synth.LEGO_C2W, idx)
    rays = (torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev))
    target = torch.rand(args.rays, 3, device=dev)
    for _ in range(4):
        for m in models:
            m.zero_grad(set_to_none=True)
        rgb, disp, acc, extras = r.render(400, 400, K, models[0], None if args.coarse_only else models[1], chunk=32768, rays=rays, retraw=True)
        loss = utils.img2mse(rgb, target)
        if args.rgb0:
            loss = loss + utils.img2mse(extras["rgb0"], target)
        loss.backward()
        torch.cuda.synchronize()
    fn = _lib.lib.nerf_amd_x_dw_stamps
    buf = (ctypes.c_ulonglong * 2048)()
    assert fn(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4).astype(np.int64)
    a = a[a[:, 0] != 0]
    t0 = a[:, 0].min()
    start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    job, shape = (a[:, 2] >> 8) & 0xffffff, a[:, 2] & 255
    xcc, hwid = (a[:, 2] >> 32) & 15, (a[:, 2] >> 40) & 0xffff
    first_chunk = np.where(a[:, 3] > 0, (a[:, 3] & 0xffffffff) / 100.0, np.nan)
    half = np.where(a[:, 3] > 0, (a[:, 3] >> 32) / 100.0, np.nan)
    n_chunks = args.rays * (args.coarse_only or (64 if args.rgb0 else 128 + 64)) // 32
    print(f"{len(a)} workgroups; starts within {start.max():.1f} us; ends {end.min():.1f} .. {end.max():.1f} us")
    for j in np.unique(job):
        s = job == j
        print(f"  product {j:2d} shape {shape[s][0]}  wgs {s.sum():3d}  start {start[s].min():6.1f}..{start[s].max():6.1f}  "
              f"end {end[s].min():6.1f}..{end[s].max():6.1f}  mean {end[s].mean():6.1f}  us/chunk {end[s].mean() / (n_chunks / s.sum()):.3f}  "
              f"first chunk at {np.nanmin(first_chunk[s]):5.1f}..{np.nanmax(first_chunk[s]):5.1f}  half way {np.nanmin(half[s]):6.1f}..{np.nanmax(half[s]):6.1f} (from own start)")


    # the same ends by XCD (bf16 kernel only): is a product's slow workgroup always on the same die?
    if xcc.any():
        wide = shape == 0
        print("  256 x 256 products, mean end by XCD:", ["%d: %.1f" % (x, end[wide & (xcc == x)].mean()) for x in range(8) if (wide & (xcc == x)).any()])
        print("  all products, workgroups per XCD:", np.bincount(xcc, minlength=8).tolist())
        # relative to the product's own mean
        rel = np.zeros(len(a))
        for j in np.unique(job):
            sj = job == j
            rel[sj] = end[sj] - end[sj].mean()
        print("  end minus the product's mean, by XCD:", ["%d: %+.1f" % (x, rel[xcc == x].mean()) for x in range(8)])
        se = (hwid >> 13) & 7      # gfx9 HW_ID: SE_ID bits 15:13, CU_ID 11:8, SH 12
        print("  ... by shader engine (HW_ID 15:13):", ["%d: %+.1f" % (x, rel[se == x].mean()) for x in np.unique(se)])
        order = np.argsort(rel)
        print("  slowest 8: block / xcd / hw_id / end-mean", [(int(i), int(xcc[i]), hex(int(hwid[i])), round(float(rel[i]), 1)) for i in order[-8:]])
        print("  fastest 8:", [(int(i), int(xcc[i]), hex(int(hwid[i])), round(float(rel[i]), 1)) for i in order[:8]])


if __name__ == "__main__":
    main()
