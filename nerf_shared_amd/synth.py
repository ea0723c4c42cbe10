"""Deterministic synthetic inputs for the render hot path: weights, cameras, rays.

There are no datasets or checkpoints offline, so every test, fixture and bench
in this repo draws its weights and poses from here.  Nothing is stored but the
seed: ``numpy.random.default_rng`` (PCG64) is stable across numpy versions.

Shapes and key names follow the reference ``NeRF.__init__``
(/root/reference/nerf_shared/nerf.py:62-94): PyTorch ``[out, in]`` row-major
fp32 ``nn.Linear`` parameters under ``pts_linears.{i}``, ``views_linears.0``,
``feature_linear``, ``alpha_linear``, ``rgb_linear`` (viewdirs) or
``output_linear`` (no viewdirs; the unused ``views_linears.0 [W/2, W]`` is
still part of the state_dict, nerf.py:83).

The Lego camera constants mirror the geometry the Blender loader derives
(/root/reference/nerf_shared/load_blender.py:81-82 focal formula;
/root/reference/nerf_shared/utils.py:296-301 K matrix); the pose itself is a
fixed synthetic constant, not a dataset value.
"""
import math

import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618

# Fixed synthetic camera-to-world pose on the Lego camera sphere (radius ~4.03).
LEGO_C2W = np.array([
    [-0.9999021887779236, 0.004192245192825794, -0.013345719315111637, -0.05379832163453102],
    [-0.013988681137561798, -0.2996590733528137, 0.95394366979599, 3.845470428466797],
    [-4.656612873077393e-10, 0.9540371894836426, 0.29968830943107605, 1.2080823183059692],
], dtype=np.float32)


def embed_dim(multires, i_embed=0):
    """Output width of get_embedder(multires, i_embed) (nerf.py:43-58)."""
    return 3 if i_embed == -1 else 3 + 6 * multires


def layer_shapes(D=8, W=256, output_ch=4, skips=(4,), use_viewdirs=False,
                 multires=10, multires_views=4, i_embed=0):
    """Ordered {state_dict key: (out, in)} for a reference NeRF (nerf.py:62-94)."""
    input_ch = embed_dim(multires, i_embed)
    input_ch_views = embed_dim(multires_views, i_embed) if use_viewdirs else 0
    shapes = {"pts_linears.0": (W, input_ch)}
    for i in range(D - 1):
        shapes["pts_linears.%d" % (i + 1)] = (W, W + input_ch if i in skips else W)
    shapes["views_linears.0"] = (W // 2, input_ch_views + W)
    if use_viewdirs:
        shapes["feature_linear"] = (W, W)
        shapes["alpha_linear"] = (1, W)
        shapes["rgb_linear"] = (3, W // 2)
    else:
        shapes["output_linear"] = (output_ch, W)
    return shapes


def make_state_dict(seed=0, sharpen=1.0, **arch):
    """numpy state_dict with nn.Linear-scale uniform init U(-1/sqrt(fan_in), +).

    ``sharpen`` multiplies the 2-D parameters (weights, not biases); 3.0 gives
    the "sharpened" stress set of SURVEY.md section 7 (sigma range about -6..13).
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for name, (n_out, n_in) in layer_shapes(**arch).items():
        bound = 1.0 / math.sqrt(n_in)
        w = rng.uniform(-bound, bound, size=(n_out, n_in)).astype(np.float32)
        b = rng.uniform(-bound, bound, size=(n_out,)).astype(np.float32)
        sd[name + ".weight"] = (w * np.float32(sharpen)).astype(np.float32)
        sd[name + ".bias"] = b
    return sd


def torch_state_dict(seed=0, sharpen=1.0, **arch):
    import torch
    return {k: torch.from_numpy(v.copy())
            for k, v in make_state_dict(seed, sharpen, **arch).items()}


def lego_intrinsics(H, W):
    """K for the Lego geometry (load_blender.py:81-82, utils.py:296-301)."""
    focal = 0.5 * W / math.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float64)


def pose_spherical(theta_deg, phi_deg=-30.0, radius=4.031128874):
    """Synthetic c2w on a camera circle around the origin: translate along z,
    rotate about x by phi, about y by theta, then swap axes to the Blender
    convention (the classic nerf-synthetic construction; the reference keeps
    it commented out at load_blender.py:29-34).  Only used to make inputs."""
    t = np.eye(4, dtype=np.float64)
    t[2, 3] = radius
    phi = phi_deg / 180.0 * math.pi
    rp = np.eye(4, dtype=np.float64)
    rp[1, 1], rp[1, 2], rp[2, 1], rp[2, 2] = math.cos(phi), -math.sin(phi), math.sin(phi), math.cos(phi)
    th = theta_deg / 180.0 * math.pi
    rt = np.eye(4, dtype=np.float64)
    rt[0, 0], rt[0, 2], rt[2, 0], rt[2, 2] = math.cos(th), -math.sin(th), math.sin(th), math.cos(th)
    c2w = rt @ rp @ t
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    return (flip @ c2w).astype(np.float32)[:3, :4]


def circle_poses(n):
    """n poses, theta = linspace(-180, 180, n)."""
    return [pose_spherical(float(th)) for th in np.linspace(-180.0, 180.0, n)]


def rays_np(H, W, K, c2w, pixel_index=None):
    """Pinhole rays (no +0.5 pixel offset), numpy twin of get_rays
    (utils.py:45-52).  Returns flat [N,3] origins and directions for the given
    flat pixel indices (row-major) or the whole image."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    c2w = np.asarray(c2w, dtype=np.float32)
    rays_d = np.sum(dirs[..., None, :] * c2w[:3, :3], -1).astype(np.float32).reshape(-1, 3)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).astype(np.float32)
    if pixel_index is not None:
        rays_o, rays_d = rays_o[pixel_index], rays_d[pixel_index]
    return np.ascontiguousarray(rays_o), np.ascontiguousarray(rays_d)


def ray_batch_np(rays_o, rays_d, near, far, use_viewdirs=True):
    """Assemble the [N, 8|11] ray batch exactly as Renderer.render does
    (render_utils.py:205-226): viewdirs = normalised rays_d."""
    n = rays_o.shape[0]
    cols = [rays_o, rays_d, np.full((n, 1), near, np.float32), np.full((n, 1), far, np.float32)]
    if use_viewdirs:
        cols.append(rays_d / np.linalg.norm(rays_d, axis=-1, keepdims=True))
    return np.concatenate(cols, -1).astype(np.float32)
