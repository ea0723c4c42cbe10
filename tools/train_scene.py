#!/usr/bin/env python3
"""Train a coarse+fine NeRF pair on a scene on disk and render its test views -- the reference's main.py loop
(main.py:17-147) on the MI355X path, without its config parser:

    python tools/train_scene.py --datadir <scene> --dataset-type blender|llff --iters 2000 --out logs/scene

Scene -> utils.load_datasets -> device-resident ray bank (utils.batch_training_data) -> N_rand rays per step ->
Renderer.render_from_rays -> mse(rgb) + mse(rgb0) -> fused Adam with the reference's exponential lr decay ->
checkpoints in the reference's .tar layout -> test poses through Renderer.render_from_batch_poses (PNG) + PSNR.
`--write-demo-scene DIR` first writes the analytic sphere scene of tools/train_demo.py to DIR in the Blender format.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import image_io, utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--datadir", required=True)
    ap.add_argument("--dataset-type", default="blender", choices=["blender", "llff"])
    ap.add_argument("--out", default="logs/scene")
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--N-rand", type=int, default=1024)
    ap.add_argument("--half-res", action="store_true")
    ap.add_argument("--testskip", type=int, default=8)
    ap.add_argument("--factor", type=int, default=8)
    ap.add_argument("--llffhold", type=int, default=8)
    ap.add_argument("--no-white-bkgd", action="store_true")
    ap.add_argument("--multires", type=int, default=10)
    ap.add_argument("--multires-views", type=int, default=4)
    ap.add_argument("--N-importance", type=int, default=128)
    ap.add_argument("--lrate", type=float, default=5e-4)
    ap.add_argument("--lrate-decay", type=int, default=250)
    ap.add_argument("--chunk", type=int, default=32768)
    ap.add_argument("--max-test-views", type=int, default=8)
    ap.add_argument("--write-demo-scene", action="store_true", help="write the analytic sphere scene to --datadir first")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ckpt", default=None, help="reference-format .tar to start from (with --iters 0: render only)")
    ap.add_argument("--render-path", type=int, default=0, help="also render this many poses of the scene's render path")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(a.seed)
    np.random.seed(a.seed)

    if a.write_demo_scene:
        from nerf_shared_amd import synth
        from tools import train_demo
        H = W = 64
        K = synth.lego_intrinsics(H, W)
        poses = np.stack([np.concatenate([synth.pose_spherical(th, -30.0 + 20.0 * np.sin(i)), [[0, 0, 0, 1]]], 0)
                          for i, th in enumerate(np.linspace(-180, 180, 13)[:-1])], 0).astype(np.float32)
        images = torch.stack([train_demo.sphere_image(H, W, K, p[:3, :4], dev) for p in poses], 0)
        os.makedirs(a.datadir, exist_ok=True)
        train_demo.write_blender_scene(a.datadir, H, W, poses, images, list(range(1, 12)), 0)

    args = SimpleNamespace(dataset_type=a.dataset_type, datadir=a.datadir, half_res=a.half_res, testskip=a.testskip,
                           white_bkgd=not a.no_white_bkgd, render_test=False, factor=a.factor, spherify=False,
                           llffhold=a.llffhold, no_ndc=False,
                           N_rand=a.N_rand, no_batching=False, lrate=a.lrate, lrate_decay=a.lrate_decay, netdepth=8, netwidth=256,
                           netdepth_fine=8, netwidth_fine=256, N_importance=a.N_importance, N_samples=64, use_viewdirs=True,
                           multires=a.multires, multires_views=a.multires_views, i_embed=0, perturb=1.0, raw_noise_std=0.0,
                           lindisp=False, basedir=os.path.dirname(os.path.abspath(a.out)) or ".",
                           expname=os.path.basename(os.path.abspath(a.out)), ft_path=None, no_reload=True)
    images, poses, render_poses, hwf, (i_train, i_val, i_test), K, bds = utils.load_datasets(args)
    H, W, _ = hwf
    ndc = a.dataset_type == "llff"
    if ndc:
        args.white_bkgd = False
    coarse, fine = utils.create_nerf_models(args, dev)
    renderer = utils.get_renderer(args, bds)
    optimizer = utils.get_optimizer(coarse, fine, args)
    start = 0
    if a.ckpt:          # resume / render-only: the package's own reader (handles fine_model None, reference .tar keys)
        args.ft_path, args.no_reload = a.ckpt, False
        start = int(utils.load_checkpoint(coarse, fine, optimizer if a.iters > 0 else None, args,
                                          b_load_ckpnt_as_trainable=True))
        print("loaded", a.ckpt, "at step", start)
    images_t, poses_t, rays_rgb, use_batching, N_rand, i_batch = utils.batch_training_data(args, poses, hwf, K, images, i_train)
    os.makedirs(a.out, exist_ok=True)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(start, start + a.iters):
        batch_rays, target, rays_rgb, i_batch = utils.sample_random_ray_batch(args, images_t, poses_t, rays_rgb, N_rand,
                                                                              use_batching, i_batch, i_train, hwf, K, 0, i)
        optimizer.zero_grad(set_to_none=True)
        rgb, disp, acc, extras = renderer.render_from_rays(H, W, K, a.chunk, batch_rays, coarse, fine, retraw=True)
        loss = utils.img2mse(rgb, target)
        if 'rgb0' in extras:
            loss = loss + utils.img2mse(extras['rgb0'], target)
        loss.backward()
        optimizer.step()
        lr = a.lrate * (0.1 ** (i / (a.lrate_decay * 1000)))          # main.py:108-112
        for g in optimizer.param_groups:
            g['lr'] = lr
        if (i + 1) % 500 == 0:
            print("iter %d  loss %.5f  psnr %.2f" % (i + 1, float(loss.detach()), float(utils.mse2psnr(utils.img2mse(rgb.detach(), target)))))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    ckpt = utils.save_checkpoints(args, coarse, fine, optimizer, start + a.iters, start + a.iters) if a.iters > 0 else a.ckpt

    test_ids = [int(i) for i in np.atleast_1d(i_test)][:a.max_test_views]
    test_poses = [poses_t[i, :3, :4] for i in test_ids]
    renderer.perturb = 0.0
    frames = renderer.render_from_batch_poses(H, W, K, a.chunk, test_poses, coarse, fine, False, os.path.join(a.out, "testset"))
    gt = images_t[test_ids][..., :3].cpu().numpy()
    psnr = [float(-10.0 * np.log10(np.mean((f.astype(np.float64) / 255.0 - g) ** 2))) for f, g in zip(frames, gt)]
    for k, f in enumerate(frames):
        image_io.write_png(os.path.join(a.out, "testset", "gt_%03d.png" % k), utils.to8b(gt[k]))
    n_path = 0
    if a.render_path > 0:
        rp = torch.as_tensor(np.asarray(render_poses), dtype=torch.float32)[:a.render_path, :3, :4]
        n_path = len(renderer.render_from_batch_poses(H, W, K, a.chunk, list(rp), coarse, fine, False, os.path.join(a.out, "path"),
                                                      b_combine_as_video=True))
    print(json.dumps({"iters": a.iters, "train_s": train_s, "it_per_s": a.iters / max(train_s, 1e-9), "checkpoint": ckpt, "path_views": n_path,
                      "test_views": len(frames), "test_psnr_mean": float(np.mean(psnr)), "test_psnr": psnr}))


if __name__ == "__main__":
    main()
