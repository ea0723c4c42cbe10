"""CPU oracle for the nerf_shared render hot path -- TEST INFRASTRUCTURE ONLY.

This is a from-scratch restatement, on torch-CPU fp32 tensors, of the
algorithm the reference runs in
    Renderer.render -> render_batch -> render_rays -> {NeRF.forward/MLP,
    raw2outputs, sample_pdf}
(/root/reference/nerf_shared/render_utils.py, nerf.py, utils.py).  The
reference is pure Python/PyTorch, so the restatement is torch too (it keeps
the same ATen arithmetic, which makes CPU agreement bit-exact), but it is
written as stateless functions over plain state_dicts with every random draw
injected by the caller.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the
reported CPU baseline.  The product (``nerf_shared_amd``) never imports it and
fails loudly when its HIP library is missing.

Parity pin: every function here is checked against golden vectors produced by
the reference itself (tests/golden/make_golden.py imports /root/reference in
the build container; fixtures are committed under tests/golden/*.npz) by
tests/test_oracle_golden.py.
"""

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# random draws of the reference's ``pytest=True`` path
# --------------------------------------------------------------------------
def pytest_uniform(shape):
    """The seeded draw the reference substitutes when ``pytest=True``:
    ``np.random.seed(0); np.random.rand(*shape)`` cast to fp32
    (render_utils.py:124-127, :267-270; utils.py:89-97)."""
    np.random.seed(0)
    return torch.Tensor(np.random.rand(*list(shape)))


# --------------------------------------------------------------------------
# a1: positional encoding          nerf.py:16-41, :43-58
# --------------------------------------------------------------------------
def embed(x, multires, i_embed=0):
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)],
    frequency-major, (sin, cos) inside a frequency, xyz innermost."""
    if i_embed == -1:
        return x
    bands = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires)
    parts = [x]
    for f in bands:
        xf = x * f
        parts.append(torch.sin(xf))
        parts.append(torch.cos(xf))
    return torch.cat(parts, -1)


# --------------------------------------------------------------------------
# a2-a5: the radiance field         nerf.py:62-143
# --------------------------------------------------------------------------
class Arch:
    """Architecture description of one reference NeRF (nerf.py:62-94)."""

    def __init__(self, D=8, W=256, output_ch=4, skips=(4,), use_viewdirs=False,
                 multires=10, multires_views=4, i_embed=0):
        self.D, self.W, self.output_ch = D, W, output_ch
        self.skips = tuple(skips)
        self.use_viewdirs = use_viewdirs
        self.multires, self.multires_views, self.i_embed = multires, multires_views, i_embed
        self.input_ch = 3 if i_embed == -1 else 3 + 6 * multires
        self.input_ch_views = (3 if i_embed == -1 else 3 + 6 * multires_views) if use_viewdirs else 0

    def kwargs(self):
        return dict(D=self.D, W=self.W, output_ch=self.output_ch, skips=list(self.skips),
                    use_viewdirs=self.use_viewdirs, multires=self.multires,
                    multires_views=self.multires_views, i_embed=self.i_embed)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def mlp(sd, arch, x):
    """NeRF.MLP (nerf.py:110-134) on already-embedded rows [P, input_ch(+views)]."""
    x_pts = x[:, :arch.input_ch]
    x_dirs = x[:, arch.input_ch:arch.input_ch + arch.input_ch_views]
    h = x_pts
    for i in range(arch.D):
        h = torch.relu(_lin(sd, "pts_linears.%d" % i, h))
        if i in arch.skips:
            h = torch.cat([x_pts, h], -1)
    if not arch.use_viewdirs:
        return _lin(sd, "output_linear", h)
    sigma = _lin(sd, "alpha_linear", h)
    feat = _lin(sd, "feature_linear", h)
    hv = torch.relu(_lin(sd, "views_linears.0", torch.cat([feat, x_dirs], -1)))
    return torch.cat([_lin(sd, "rgb_linear", hv), sigma], -1)


def nerf_forward(sd, arch, pts, viewdirs, netchunk=1024 * 64):
    """NeRF.forward (nerf.py:96-108): pts [..., S, 3], viewdirs [R, 3] or None
    -> [..., S, 4|output_ch]."""
    flat = pts.reshape(-1, pts.shape[-1])
    e = embed(flat, arch.multires, arch.i_embed)
    if viewdirs is not None:
        d = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
        e = torch.cat([e, embed(d, arch.multires_views, arch.i_embed)], -1)
    outs = [mlp(sd, arch, e[i:i + netchunk]) for i in range(0, e.shape[0], netchunk)]
    out = torch.cat(outs, 0)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


def get_density(sd, arch, points, netchunk=1024 * 64):
    """NeRF.get_density (nerf.py:136-143): raw sigma with an all-ones view dir."""
    ones = torch.ones_like(points[..., 0, :])
    return nerf_forward(sd, arch, points, ones, netchunk)[..., -1]


# --------------------------------------------------------------------------
# a10: alpha compositing            render_utils.py:241-290
# --------------------------------------------------------------------------
def raw2outputs(raw, z_vals, rays_d, white_bkgd, noise=None):
    """Returns rgb_map [R,3], disp_map [R], acc_map [R], weights [R,S], depth_map [R].
    ``noise`` is the already-scaled additive sigma noise (or None)."""
    gaps = z_vals[..., 1:] - z_vals[..., :-1]
    gaps = torch.cat([gaps, torch.full_like(gaps[..., :1], 1e10)], -1)
    gaps = gaps * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sigma = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * gaps)
    one = torch.ones((alpha.shape[0], 1))
    trans = torch.cumprod(torch.cat([one, 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / torch.sum(weights, -1))
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# --------------------------------------------------------------------------
# a11: hierarchical sampling        utils.py:74-117
# --------------------------------------------------------------------------
def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """Inverse-CDF samples.  ``u`` [R, n_samples] overrides the draw (the
    reference draws torch.rand, or the pytest_uniform() tensor)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0.0, 1.0, steps=n_samples).expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def pytest_u_for_sample_pdf(n_rays, n_samples, det):
    """u of the reference's pytest path (utils.py:89-97)."""
    np.random.seed(0)
    if det:
        u = np.broadcast_to(np.linspace(0.0, 1.0, n_samples), [n_rays, n_samples])
    else:
        u = np.random.rand(n_rays, n_samples)
    return torch.Tensor(u)


# --------------------------------------------------------------------------
# a12-a13: ray math                 utils.py:33-71
# --------------------------------------------------------------------------
def get_rays(H, W, K, c2w):
    xs = torch.linspace(0, W - 1, W)
    ys = torch.linspace(0, H - 1, H)
    i = xs[None, :].expand(H, W)
    j = ys[:, None].expand(H, W)
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    sx = -1.0 / (W / (2.0 * focal))
    sy = -1.0 / (H / (2.0 * focal))
    o0 = sx * rays_o[..., 0] / rays_o[..., 2]
    o1 = sy * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------
# a6-a9: the renderer               render_utils.py:14-238
# --------------------------------------------------------------------------
class RenderCfg:
    """Renderer constructor arguments (render_utils.py:14-31)."""

    def __init__(self, perturb=True, N_importance=128, N_samples=64, use_viewdirs=True,
                 white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False,
                 near=0.0, far=1.0):
        self.perturb, self.N_importance, self.N_samples = perturb, N_importance, N_samples
        self.use_viewdirs, self.white_bkgd, self.raw_noise_std = use_viewdirs, white_bkgd, raw_noise_std
        self.ndc, self.lindisp, self.near, self.far = ndc, lindisp, near, far

    def kwargs(self):
        return dict(self.__dict__)


def coarse_z_vals(cfg, near, far, n_rays, t_rand=None):
    """Stratified depths (render_utils.py:105-129).  near/far are [R,1]."""
    t = torch.linspace(0.0, 1.0, steps=cfg.N_samples)
    if not cfg.lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand([n_rays, cfg.N_samples])
    if cfg.perturb > 0.0:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        if t_rand is None:
            t_rand = torch.rand(z.shape)
        z = lower + (upper - lower) * t_rand
    return z


def render_rays(cfg, ray_batch, coarse, fine, retraw=False, retweights=False,
                pytest=False, t_rand=None, noise0=None, noise1=None, u=None):
    """Renderer.render_rays (render_utils.py:67-174).

    ``coarse`` / ``fine`` are (state_dict, Arch) pairs (``fine`` may be None).
    With ``pytest=True`` every draw is the reference's seeded numpy draw;
    otherwise draws come from the injected tensors, or torch's generator.
    """
    n_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    bounds = ray_batch[..., 6:8].reshape(-1, 1, 2)
    near, far = bounds[..., 0], bounds[..., 1]

    if pytest and cfg.perturb > 0.0:
        t_rand = pytest_uniform([n_rays, cfg.N_samples])
    z = coarse_z_vals(cfg, near, far, n_rays, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]

    def noise_for(shape, injected):
        if not cfg.raw_noise_std > 0.0:
            return None
        if pytest:
            return pytest_uniform(shape) * cfg.raw_noise_std
        if injected is not None:
            return injected
        return torch.randn(shape) * cfg.raw_noise_std

    raw = nerf_forward(coarse[0], coarse[1], pts, viewdirs)
    rgb, disp, acc, weights, _ = raw2outputs(raw, z, rays_d, cfg.white_bkgd,
                                             noise_for(raw[..., 3].shape, noise0))
    out = {}
    if cfg.N_importance > 0:
        rgb0, disp0, acc0 = rgb, disp, acc
        z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
        det = (cfg.perturb == 0.0)
        if pytest:
            u = pytest_u_for_sample_pdf(n_rays, cfg.N_importance, det)
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], cfg.N_importance, det=det, u=u).detach()
        z, _ = torch.sort(torch.cat([z, z_samples], -1), -1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        net = coarse if fine is None else fine
        raw = nerf_forward(net[0], net[1], pts, viewdirs)
        rgb, disp, acc, weights, _ = raw2outputs(raw, z, rays_d, cfg.white_bkgd,
                                                 noise_for(raw[..., 3].shape, noise1))
    out.update(rgb_map=rgb, disp_map=disp, acc_map=acc)
    if retraw:
        out["raw"] = raw
    if retweights:
        out["weights"] = weights
        out["z_vals"] = z
    if cfg.N_importance > 0:
        out.update(rgb0=rgb0, disp0=disp0, acc0=acc0,
                   z_std=torch.std(z_samples, dim=-1, unbiased=False))
    return out


def render(cfg, H, W, K, coarse, fine, chunk=1024 * 32, rays=None, retraw=True,
           c2w=None, c2w_staticcam=None, **draws):
    """Renderer.render + render_batch (render_utils.py:51-65, :176-238).
    Returns [rgb, disp, acc, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    viewdirs = None
    if cfg.use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = viewdirs.reshape(-1, 3).float()
    shape = rays_d.shape
    if cfg.ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1.0, rays_o, rays_d)
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    near = cfg.near * torch.ones_like(rays_d[..., :1])
    far = cfg.far * torch.ones_like(rays_d[..., :1])
    batch = torch.cat([rays_o, rays_d, near, far], -1)
    if cfg.use_viewdirs:
        batch = torch.cat([batch, viewdirs], -1)
    parts = {}
    for i in range(0, batch.shape[0], chunk):
        r = render_rays(cfg, batch[i:i + chunk], coarse, fine, retraw, **draws)
        for k, v in r.items():
            parts.setdefault(k, []).append(v)
    full = {k: torch.cat(v, 0) for k, v in parts.items()}
    for k in full:
        full[k] = full[k].reshape(list(shape[:-1]) + list(full[k].shape[1:]))
    head = ["rgb_map", "disp_map", "acc_map"]
    return [full[k] for k in head] + [{k: v for k, v in full.items() if k not in head}]


# --------------------------------------------------------------------------
# a14: metrics                      utils.py:24-30
# --------------------------------------------------------------------------
def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10.0 * torch.log(x) / torch.log(torch.Tensor([10.0]))


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def state_dict_to_torch(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).float()
            for k, v in sd.items()}
