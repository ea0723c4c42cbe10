#!/usr/bin/env python3
"""End-to-end training demo on a synthetic scene (no datasets offline): a shaded sphere on a
white background, rendered analytically from poses on a camera circle, is learned by a fresh
coarse+fine NeRF pair with the reference's loop (main.py:67-112: random ray batches -> render ->
mse(rgb) + mse(rgb0) -> Adam with exponential lr decay).  Prints PSNR on a held-out view.

    python tools/train_demo.py [--steps 600] [--res 64] [--views 12]
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

from nerf_shared_amd import render_utils, synth, utils  # noqa: E402


def sphere_image(H, W, K, c2w, dev, radius=1.0, ss=1):
    """Analytic target: Lambert-shaded unit sphere at the origin, white background.  ss > 1: the mean of ss x ss
    sub-pixel samples (an anti-aliased silhouette, as a rendered dataset frame has)."""
    if ss > 1:
        Ks = np.array(K, np.float64).copy()
        Ks[:2, :] *= ss                                      # an ss-times finer pixel grid over the same field of view
        Ks[0, 2] += 0.5 * (ss - 1)
        Ks[1, 2] += 0.5 * (ss - 1)                           # sub-pixel centres symmetric around the coarse pixel centre
        fine = sphere_image(H * ss, W * ss, Ks, c2w, dev, radius, 1)
        return fine.reshape(H, ss, W, ss, 3).mean((1, 3))
    ro, rd = utils.get_rays(H, W, K, torch.from_numpy(c2w))
    rd_n = rd / rd.norm(dim=-1, keepdim=True)
    b = (ro * rd_n).sum(-1)
    disc = b * b - ((ro * ro).sum(-1) - radius * radius)
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp_min(0))
    n = (ro + rd_n * t[..., None]) / radius
    light = torch.tensor([0.5, 0.4, 0.77], device=dev)
    shade = (n * light).sum(-1).clamp(0.1, 1.0)
    base = 0.5 + 0.5 * n                                    # normal-coloured albedo
    img = torch.where(hit[..., None], base * shade[..., None], torch.ones_like(base))
    return img


def write_blender_scene(root, H, W, poses, images, i_train, i_test):
    """The analytic frames as a NeRF-synthetic scene on disk: transforms_{train,val,test}.json (with the
    reference's near / far keys, load_blender.py:57) + RGBA PNGs (alpha = 0 on the white background)."""
    from nerf_shared_amd import image_io
    angle_x = 2.0 * np.arctan(0.5 * W / synth.lego_intrinsics(H, W)[0, 0])
    for split, ids in (("train", i_train), ("val", [i_test]), ("test", [i_test])):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        meta = {"camera_angle_x": float(angle_x), "near": 2.0, "far": 6.0, "frames": []}
        for k, i in enumerate(ids):
            rgb = images[i].cpu().numpy()
            alpha = (np.abs(rgb - 1.0).max(-1, keepdims=True) > 0).astype(np.float32)       # background is exactly white
            image_io.write_png(os.path.join(root, split, "r_%d.png" % k), utils.to8b(np.concatenate([rgb, alpha], -1)))
            meta["frames"].append({"file_path": "./%s/r_%d" % (split, k), "transform_matrix": poses[i].tolist()})
        with open(os.path.join(root, "transforms_%s.json" % split), "w") as f:
            json.dump(meta, f)


def run(steps=600, res=64, views=12, n_rand=1024, seed=0, verbose=True, scene_dir=None, multires=10, multires_views=4,
        lrate=5e-4, ss=1):
    """scene_dir: write the scene to disk in the Blender format first and train from what
    utils.load_datasets reads back (8-bit frames) instead of from the in-memory float images."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    dev = torch.device("cuda:0")
    H = W = res
    K = synth.lego_intrinsics(H, W)
    poses = np.stack([np.concatenate([synth.pose_spherical(th, -30.0 + 20.0 * np.sin(i)), [[0, 0, 0, 1]]], 0)
                      for i, th in enumerate(np.linspace(-180, 180, views + 1)[:-1])], 0).astype(np.float32)
    images = torch.stack([sphere_image(H, W, K, p[:3, :4], dev, ss=ss) for p in poses], 0)
    i_train, i_test = list(range(1, views)), 0
    near, far = 2.0, 6.0
    if scene_dir is not None:
        write_blender_scene(scene_dir, H, W, poses, images, i_train, i_test)
        largs = SimpleNamespace(dataset_type="blender", datadir=scene_dir, half_res=False, testskip=1, white_bkgd=True,
                                render_test=False)
        imgs_np, poses, _, hwf, (i_train, _, i_test_arr), K, bds = utils.load_datasets(largs)
        assert hwf[:2] == [H, W] and abs(hwf[2] - synth.lego_intrinsics(H, W)[0, 0]) < 1e-3 * hwf[2]
        images = torch.from_numpy(np.ascontiguousarray(imgs_np)).float().to(dev)
        i_train, i_test = list(i_train), int(i_test_arr[0])
        near, far = bds["near"], bds["far"]
    args = SimpleNamespace(N_rand=n_rand, no_batching=False, lrate=lrate, lrate_decay=250, netdepth=8, netwidth=256,
                           netdepth_fine=8, netwidth_fine=256, N_importance=128, use_viewdirs=True, multires=multires,
                           multires_views=multires_views, i_embed=0)
    coarse, fine = utils.create_nerf_models(args, dev)
    renderer = render_utils.Renderer(perturb=1.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True,
                                     raw_noise_std=0.0, near=near, far=far)
    opt = utils.get_optimizer(coarse, fine, args)
    images, poses_t, rays_rgb, use_batching, N_rand, i_batch = utils.batch_training_data(args, poses, (H, W, K[0][0]), K, images, i_train)

    def test_psnr():
        with torch.no_grad():
            rgb = renderer.render(H, W, K, coarse, fine, chunk=32768, c2w=poses_t[i_test, :3, :4], retraw=False)[0]
            return float(utils.mse2psnr(utils.img2mse(rgb, images[i_test])))

    psnr0 = test_psnr()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        batch_rays, target, rays_rgb, i_batch = utils.sample_random_ray_batch(args, images, poses_t, rays_rgb, N_rand, use_batching,
                                                                              i_batch, i_train, (H, W, K[0][0]), K, 0, i)
        opt.zero_grad(set_to_none=True)
        rgb, disp, acc, extras = renderer.render_from_rays(H, W, K, 32768, batch_rays, coarse, fine, retraw=True)
        loss = utils.img2mse(rgb, target) + utils.img2mse(extras['rgb0'], target)
        loss.backward()
        opt.step()
        new_lr = args.lrate * (0.1 ** (i / (args.lrate_decay * 1000)))          # main.py:108-112
        for g in opt.param_groups:
            g['lr'] = new_lr
        if verbose and (i + 1) % 100 == 0:
            print("step %d loss %.5f" % (i + 1, float(loss.detach())))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"steps": steps, "rays_per_step": n_rand, "train_s": dt, "steps_per_s": steps / dt,
           "psnr_before": psnr0, "psnr_after": test_psnr(), "final_loss": float(loss.detach())}
    return out, (coarse, fine, opt, args, renderer, (H, W, K), poses_t, images, i_test)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--scene-dir", default=None, help="write the scene in the Blender format here and train from disk")
    ap.add_argument("--multires", type=int, default=10)
    ap.add_argument("--multires-views", type=int, default=4)
    ap.add_argument("--n-rand", type=int, default=1024)
    ap.add_argument("--lrate", type=float, default=5e-4)
    ap.add_argument("--ss", type=int, default=1, help="ss x ss sub-pixel samples per ground-truth pixel")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.res, a.views, n_rand=a.n_rand, scene_dir=a.scene_dir, multires=a.multires,
                         multires_views=a.multires_views, lrate=a.lrate, ss=a.ss, verbose=not a.quiet)[0]))
