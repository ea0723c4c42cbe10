"""Image output stage: Renderer.render_from_batch_poses over a ring of poses.

Compares, for the same poses / weights / image size:
  render_only      the GPU renders, nothing leaves the device
  reference_style  what render_utils.py:302-315 does per pose: float image -> host, numpy to8b,
                   PNG written before the next pose starts (same PNG encoder as below)
  pipelined        render_from_batch_poses: to8b on the GPU, uint8 D2H on a copy stream into pinned
                   memory, PNG encoding on worker threads while the next pose renders
Prints one JSON line."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
from nerf_shared_amd import image_io, nerf, render_utils, synth, utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--poses", type=int, default=12)
    ap.add_argument("--chunk", type=int, default=32768)
    ap.add_argument("--workers", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H = W = args.size
    K = synth.lego_intrinsics(H, W)
    models = []
    for seed in (0, 10):
        m = nerf.NeRF(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True).to(dev)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, D=8, W=256, output_ch=5, skips=(4,), use_viewdirs=True))
        models.append(m.requires_grad_(False))
    r = render_utils.Renderer(perturb=0., N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True, near=2., far=6.)
    poses = [torch.from_numpy(synth.pose_spherical(a)) for a in np.linspace(-180, 180, args.poses + 1)[:-1]]

    def render(c2w):
        return r.render_from_pose(H, W, K, chunk=args.chunk, c2w=c2w, coarse_model=models[0], fine_model=models[1])[0]

    with torch.no_grad():
        render(poses[0]); torch.cuda.synchronize()
        t = time.perf_counter()
        for c2w in poses:
            render(c2w)
        torch.cuda.synchronize()
        t_render = time.perf_counter() - t
        with tempfile.TemporaryDirectory() as d:
            t = time.perf_counter()
            for i, c2w in enumerate(poses):
                rgb8 = utils.to8b(render(c2w).cpu().numpy())
                image_io.write_png(os.path.join(d, "%03d.png" % i), rgb8)
            t_ref = time.perf_counter() - t
            ref_frames = [image_io.read_image(os.path.join(d, "%03d.png" % i)) for i in range(len(poses))]
        with tempfile.TemporaryDirectory() as d:
            t = time.perf_counter()
            frames = r.render_from_batch_poses(H, W, K, args.chunk, poses, models[0], models[1], False, d, io_workers=args.workers)
            t_pipe = time.perf_counter() - t
            same = all(np.array_equal(a, b) for a, b in zip(frames, ref_frames)) and \
                all(np.array_equal(image_io.read_image(os.path.join(d, "%03d.png" % i)), ref_frames[i]) for i in range(len(poses)))
    n = len(poses)
    print(json.dumps({"image": "%dx%d" % (H, W), "poses": n, "io_workers": args.workers,
                      "render_only_img_per_s": n / t_render, "reference_style_img_per_s": n / t_ref,
                      "pipelined_img_per_s": n / t_pipe, "pipelined_vs_render_only": t_render / t_pipe,
                      "frames_identical_to_reference_style": bool(same)}))


if __name__ == "__main__":
    main()
