"""LLFF forward-facing scenes (`poses_bounds.npy` + `images[_N]/`): the other on-disk format in
front of the render path.  Mirrors /root/reference/nerf_shared/load_llff.py: `_load_data` (:59-120),
the pose arithmetic (`normalize`, `viewmatrix`, `ptstocam`, `poses_avg`, `render_path_spiral`,
`recenter_poses`, `spherify_poses`, :126-238) and `load_llff_data` (:243-316).

Differences, all on the file side: frames are read with image_io.read_image instead of imageio (JPEGs
need Pillow); a missing down-sized folder is produced with Pillow's LANCZOS resampling rather than by
shelling out to ImageMagick's `mogrify` (:34-56), so freshly minified pixels are close to but not
bit-identical with the reference's -- scenes that ship their `images_N` folders are unaffected; and
`path_zflat=True` works (the reference halves N_views into a float, which numpy >= 1.18 refuses).
"""
import os

import numpy as np

from . import image_io

_IMG_EXT = ('JPG', 'jpg', 'png', 'jpeg', 'PNG')


def _minify(basedir, factors=(), resolutions=()):
    """Create images_<factor>/ or images_<W>x<H>/ next to images/ when they do not exist yet."""
    todo = [r for r in list(factors) + list(resolutions)
            if not os.path.exists(os.path.join(basedir, 'images_{}'.format(r) if isinstance(r, int)
                                               else 'images_{}x{}'.format(r[1], r[0])))]
    if not todo:
        return
    try:
        from PIL import Image
    except ImportError:
        raise ImportError("down-sizing LLFF frames needs Pillow (or ship the images_N folder with the scene)")
    src = os.path.join(basedir, 'images')
    names = [f for f in sorted(os.listdir(src)) if f.endswith(_IMG_EXT)]
    for r in todo:
        dst = os.path.join(basedir, 'images_{}'.format(r) if isinstance(r, int) else 'images_{}x{}'.format(r[1], r[0]))
        print('Minifying', r, basedir)
        os.makedirs(dst)
        for f in names:
            with Image.open(os.path.join(src, f)) as im:
                size = (int(round(im.width / r)), int(round(im.height / r))) if isinstance(r, int) else (r[1], r[0])
                im.convert('RGB').resize(size, Image.LANCZOS).save(os.path.join(dst, os.path.splitext(f)[0] + '.png'))


def _frame_files(folder):
    return [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith(('JPG', 'jpg', 'png'))]


def _load_data(basedir, factor=None, width=None, height=None, load_imgs=True):
    """poses [3,5,N] (with H, W, focal/factor in column 4), bds [2,N], imgs [H,W,3,N] in 0..1.
    Exactly one of factor / height / width selects the down-sized frame folder (none: images/)."""
    table = np.load(os.path.join(basedir, 'poses_bounds.npy'))            # [N, 17]: 3x5 pose, near, far
    poses = table[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = table[:, -2:].transpose([1, 0])

    full_h, full_w = image_io.read_image(_frame_files(os.path.join(basedir, 'images'))[0]).shape[:2]
    suffix = ''
    if factor is not None:
        _minify(basedir, factors=[factor])
        suffix = '_{}'.format(factor)
    elif height is not None or width is not None:
        if height is not None:
            factor = full_h / float(height)
            width = int(full_w / factor)
        else:
            factor = full_w / float(width)
            height = int(full_h / factor)
        _minify(basedir, resolutions=[[height, width]])
        suffix = '_{}x{}'.format(width, height)
    else:
        factor = 1

    folder = os.path.join(basedir, 'images' + suffix)
    if not os.path.exists(folder):
        print(folder, 'does not exist, returning')
        return
    files = _frame_files(folder)
    if len(files) != poses.shape[-1]:
        print('Mismatch between imgs {} and poses {} !!!!'.format(len(files), poses.shape[-1]))
        return
    # the hwf column describes the frames actually loaded: their size, and the focal length scaled with them
    h, w = image_io.read_image(files[0]).shape[:2]
    poses[:2, 4, :] = np.array([h, w]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bds
    imgs = np.stack([image_io.read_image(f)[..., :3] / 255. for f in files], -1)
    print('Loaded image data', imgs.shape, poses[:, -1, 0])
    return poses, bds, imgs


# ---------------------------------------------------------------- pose arithmetic (numpy, host)
def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """[3,4] camera frame looking along z with `up` roughly up, at pos (columns x, y, z, origin)."""
    zc = normalize(z)
    xc = normalize(np.cross(up, zc))
    yc = normalize(np.cross(zc, xc))
    return np.stack([xc, yc, zc, pos], 1)


def ptstocam(pts, c2w):
    return np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0]


def poses_avg(poses):
    """Average camera [3,5] of poses [N,3,5]: mean origin, summed z and y axes, hwf of the first."""
    origin = poses[:, :3, 3].mean(0)
    z = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(z, up, origin), poses[0, :3, -1:]], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """N poses [3,5] on a spiral around c2w, all looking at the point `focal` in front of it."""
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    target = np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.]))
    out = []
    for theta in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        eye = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        out.append(np.concatenate([viewmatrix(normalize(eye - target), up, eye), hwf], 1))
    return out


def _to44(p34):
    """[N,3,4] -> [N,4,4] with a (0,0,0,1) row."""
    row = np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p34.shape[0], 1, 1])
    return np.concatenate([p34, row], 1)


def recenter_poses(poses):
    """Express poses [N,3,5] in the frame of their average camera (hwf column untouched)."""
    out = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    avg = np.concatenate([poses_avg(poses)[:3, :4], bottom], -2)
    stacked = np.concatenate([poses[:, :3, :4], np.tile(np.reshape(bottom, [1, 1, 4]), [poses.shape[0], 1, 1])], -2)
    out[:, :3, :4] = (np.linalg.inv(avg) @ stacked)[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """Inward-facing captures: recentre on the point closest to all optical axes, scale the rig to
    the unit sphere (bds is scaled IN PLACE like the reference) and return a 120-pose circle."""
    dirs, origins = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - dirs * np.transpose(dirs, [0, 2, 1])
    b = -A @ origins
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))

    up = normalize((poses[:, :3, 3] - center).mean(0))
    x = normalize(np.cross([.1, .2, .3], up))
    y = normalize(np.cross(up, x))
    frame = np.stack([x, y, up, center], 1)
    reset = np.linalg.inv(_to44(frame[None])) @ _to44(poses[:, :3, :4])

    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    rcircle = np.sqrt(rad ** 2 - zh ** 2)
    ring = []
    for th in np.linspace(0., 2. * np.pi, 120):
        eye = np.array([rcircle * np.cos(th), rcircle * np.sin(th), zh])
        zc = normalize(eye)
        xc = normalize(np.cross(zc, np.array([0, 0, -1.])))
        yc = normalize(np.cross(zc, xc))
        ring.append(np.stack([xc, yc, zc, eye], 1))
    ring = np.stack(ring, 0)
    hwf = poses[0, :3, -1:]
    ring = np.concatenate([ring, np.broadcast_to(hwf, ring[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, ring, bds


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """-> images [N,H,W,3] fp32, poses [N,3,5] fp32, bds [N,2] fp32, render_poses [M,3,5] fp32, i_test."""
    poses, bds, imgs = _load_data(basedir, factor=factor)
    print('Loaded', basedir, bds.min(), bds.max())

    # LLFF stores rotations as (down, right, back): reorder to (right, up, back); frame axis first
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)

    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)

    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        c2w = poses_avg(poses)
        print('recentered', c2w.shape)
        print(c2w[:3, :4])
        up = normalize(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)          # focus depth of the spiral
        zdelta = close_depth * .2
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            c2w[:3, 3] = c2w[:3, 3] + (-close_depth * .1) * c2w[:3, 2]
            rads[2] = 0.
            n_rots, n_views = 1, n_views // 2
        render_poses = render_path_spiral(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)
    render_poses = np.array(render_poses).astype(np.float32)

    c2w = poses_avg(poses)
    print('Data:')
    print(poses.shape, images.shape, bds.shape)
    i_test = np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1))
    print('HOLDOUT view is', i_test)
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test
