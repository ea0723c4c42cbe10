"""NeRF-synthetic ("Blender") scenes: the on-disk format in front of the render path.

Mirrors /root/reference/nerf_shared/load_blender.py (`load_blender_data` :44-98, `pose_spherical`
:36-41) without imageio / cv2: frames are read with image_io.read_image (PNG natively), and the
half-resolution option averages 2x2 blocks, which is what ``cv2.resize(..., INTER_AREA)`` does for an
exact factor of two (cv2 is not installed here, so that equivalence is by OpenCV's documentation, not
by a run).  `near` / `far` come from the JSON like the reference (:57, keys its authors added); stock
nerf-synthetic files do not have them and get the conventional 2.0 / 6.0.
"""
import json
import os

import numpy as np
import torch

from . import image_io

STOCK_NEAR, STOCK_FAR = 2.0, 6.0


def _rot_x(phi):
    c, s = np.cos(phi), np.sin(phi)
    return torch.Tensor([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]]).float()


def _rot_y(th):
    c, s = np.cos(th), np.sin(th)
    return torch.Tensor([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]]).float()


# the reference's camera rig: a fixed offset applied after the two rotations (load_blender.py:40);
# `radius` is accepted and ignored exactly like there
_RIG_OFFSET = torch.Tensor(np.array([[1, 0, 0, 3], [0, 1, 0, 0.3], [0, 0, 1, -1], [0, 0, 0, 1]]))


def pose_spherical(theta, phi, radius):
    """4x4 fp32 camera-to-world of the reference's render path (angles in degrees)."""
    return _RIG_OFFSET @ (_rot_y(theta / 180. * np.pi) @ _rot_x(phi / 180. * np.pi))


def halve_area(img):
    """[H, W, C] float32 -> [H//2, W//2, C]: mean of each 2x2 block, summed in row-major order in fp32."""
    H2, W2 = img.shape[0] // 2, img.shape[1] // 2
    a = img[:2 * H2, :2 * W2].astype(np.float32, copy=False)
    s = a[0::2, 0::2] + a[0::2, 1::2]
    s = s + a[1::2, 0::2]
    s = s + a[1::2, 1::2]
    return s * np.float32(0.25)


def load_blender_data(basedir, half_res=False, testskip=1):
    """-> imgs [N,H,W,4] in 0..1, poses [N,4,4] fp32, render_poses [40,4,4], [H, W, focal],
    [i_train, i_val, i_test], near, far   (load_blender.py:44-98)."""
    per_split, counts = [], [0]
    meta = None
    for split in ('train', 'val', 'test'):
        with open(os.path.join(basedir, 'transforms_{}.json'.format(split)), 'r') as fp:
            meta = json.load(fp)
        near, far = meta.get('near', STOCK_NEAR), meta.get('far', STOCK_FAR)
        step = 1 if (split == 'train' or testskip == 0) else testskip
        frames = meta['frames'][::step]
        imgs = [image_io.read_image(os.path.join(basedir, f['file_path'] + '.png')) for f in frames]
        imgs = (np.array(imgs) / 255.).astype(np.float32)          # all four channels stay (RGBA)
        poses = np.array([f['transform_matrix'] for f in frames]).astype(np.float32)
        per_split.append((imgs, poses))
        counts.append(counts[-1] + imgs.shape[0])
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate([p[0] for p in per_split], 0)
    poses = np.concatenate([p[1] for p in per_split], 0)

    H, W = imgs[0].shape[:2]
    focal = .5 * W / np.tan(.5 * float(meta['camera_angle_x']))
    render_poses = torch.stack([pose_spherical(angle, 0, 4.0) for angle in np.linspace(-180, 180, 40 + 1)[:-1]], 0)

    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        small = np.zeros((imgs.shape[0], H, W, 4))                   # float64 container, like the reference's
        for i, img in enumerate(imgs):
            small[i] = halve_area(img)
        imgs = small
    return imgs, poses, render_poses, [H, W, focal], i_split, near, far
