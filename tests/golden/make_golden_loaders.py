#!/usr/bin/env python3
"""Golden vectors for the dataset side of the path, from the REFERENCE.

Run in the build container only.  Imports /root/reference/nerf_shared/load_llff.py and
load_blender.py with inert placeholders for imageio / cv2 (absent here; only the file-reading lines
use them) and records what their *pose arithmetic* returns on synthetic camera sets:

  G9  load_blender.pose_spherical; load_llff.normalize / viewmatrix / ptstocam / poses_avg /
      recenter_poses / render_path_spiral / spherify_poses; and load_llff.load_llff_data end to end
      with its file-reading helper `_load_data` replaced by one that returns the synthetic
      (poses, bds, imgs) arrays stored in the fixture -- everything after the read is the reference's.

Only arrays and arguments are written (tests/golden/g9_loaders.npz).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

for name in ("imageio", "cv2"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, "/root/reference")
from nerf_shared import load_blender, load_llff  # noqa: E402


def synthetic_llff(n=9, seed=3):
    """A forward-facing rig in LLFF's poses_bounds layout: poses [3, 5, n], bds [2, n], imgs [h, w, 3, n]."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((3, 5, n))
    for i in range(n):
        a = rng.normal(scale=0.15, size=3)
        rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        poses[:, :3, i] = rx @ ry @ rz
        poses[:, 3, i] = rng.normal(scale=0.6, size=3) + np.array([0.3, -0.2, 0.1])
        poses[:, 4, i] = [12, 16, 11.5]
    bds = np.stack([rng.uniform(1.1, 1.6, n), rng.uniform(7.0, 12.0, n)], 0)
    imgs = rng.uniform(0, 1, size=(12, 16, 3, n))
    return poses, bds, imgs


def main():
    out = {}
    ang = np.array([[-180.0, 0.0, 4.0], [-171.0, 0.0, 4.0], [37.5, -30.0, 4.0], [90.0, 12.0, 2.5]])
    out["ps_args"] = ang
    out["ps_out"] = np.stack([load_blender.pose_spherical(*a).numpy() for a in ang])

    rng = np.random.default_rng(0)
    v = rng.normal(size=(3, 3))
    out["vm_in"] = v
    out["normalize_out"] = load_llff.normalize(v[0])
    out["viewmatrix_out"] = load_llff.viewmatrix(v[0], v[1], v[2])

    poses, bds, imgs = synthetic_llff()
    out["raw_poses"], out["raw_bds"], out["raw_imgs"] = poses, bds, imgs
    # the layout load_llff_data works in: [n, 3, 5], rotation columns reordered
    p = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    p = np.moveaxis(p, -1, 0).astype(np.float32)
    out["p_in"] = p
    out["poses_avg_out"] = load_llff.poses_avg(p)
    out["recenter_out"] = load_llff.recenter_poses(p)
    pts = rng.normal(size=(5, 3))
    out["ptstocam_pts"] = pts
    out["ptstocam_out"] = load_llff.ptstocam(pts, p[0])
    c2w = load_llff.poses_avg(p)
    out["spiral_out"] = np.array(load_llff.render_path_spiral(c2w, load_llff.normalize(p[:, :3, 1].sum(0)),
                                                              np.array([0.3, 0.2, 0.1]), 3.5, 0.2, zrate=.5, rots=2, N=7))
    sp, sr, sb = load_llff.spherify_poses(p.copy(), np.moveaxis(bds, -1, 0).astype(np.float32).copy())
    out["spherify_poses"], out["spherify_render"], out["spherify_bds"] = sp, sr, sb

    real_load = load_llff._load_data
    cases = [dict(recenter=True, bd_factor=.75, spherify=False, path_zflat=False),
             dict(recenter=True, bd_factor=.75, spherify=True, path_zflat=False),
             # path_zflat=True cannot be captured: the reference halves N_views into a float and numpy >= 1.18
             # refuses it in linspace (load_llff.py:296, :158)
             dict(recenter=True, bd_factor=None, spherify=False, path_zflat=False),
             dict(recenter=False, bd_factor=.75, spherify=False, path_zflat=False)]
    try:
        for k, kw in enumerate(cases):
            load_llff._load_data = lambda basedir, factor=None, **_: (poses.copy(), bds.copy(), imgs.copy())
            images, ps, bd, rp, i_test = load_llff.load_llff_data("unused", factor=8, **kw)
            out["llff%d_args" % k] = np.array([kw["recenter"], -1.0 if kw["bd_factor"] is None else kw["bd_factor"],
                                               kw["spherify"], kw["path_zflat"]], np.float64)
            out["llff%d_images" % k], out["llff%d_poses" % k], out["llff%d_bds" % k] = images, ps, bd
            out["llff%d_render_poses" % k], out["llff%d_i_test" % k] = rp, np.int64(i_test)
    finally:
        load_llff._load_data = real_load
    np.savez_compressed(os.path.join(HERE, "g9_loaders.npz"), **out)
    print("wrote g9_loaders.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
