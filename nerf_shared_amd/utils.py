"""Drop-in for the hot-path half of nerf_shared/utils.py: ray generation, the NDC
warp, hierarchical sampling, metrics and the two factories that build models
and renderers from an argparse namespace.

Reference: /root/reference/nerf_shared/utils.py:24-161 for the hot path.  From the
callers either side of it this module also carries the optimizer
factory and checkpoint format (utils.py:163-214, :444-456) and a device-resident version
of the training ray batching (utils.py:360-442).  Dataset loaders are not reproduced
(no datasets offline).
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib

# ---------------------------------------------------------------- metrics (utils.py:24-30)
class _Img2MseFn(torch.autograd.Function):
    """mean((x - y)^2) of two fp32 device tensors of one shape: one launch forward, one backward (the torch expression
    is three forward and five backward launches; the loss of main.py:93-98 evaluates it twice per step)."""

    @staticmethod
    def forward(ctx, x, y):
        dev, n = x.device, x.numel()
        out = torch.empty((), device=dev, dtype=torch.float32)
        partials = torch.empty(256, device=dev, dtype=torch.float32) if n > 16384 else None
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_img2mse(x.data_ptr(), y.data_ptr(), n, out.data_ptr(), _lib.ptr(partials),
                                            _lib.stream_of(dev)), "nerf_amd_img2mse")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        dev = x.device
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        g = g.contiguous().float()
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_img2mse_backward(x.data_ptr(), y.data_ptr(), x.numel(), g.data_ptr(), _lib.ptr(gx),
                                                     _lib.ptr(gy), _lib.stream_of(dev)), "nerf_amd_img2mse_backward")
        return gx, gy


def img2mse(x, y):
    """torch.mean((x - y) ** 2) (utils.py:24).  Two fp32 tensors of one shape on a ROCm device go through the library
    (one launch each way); anything else -- host tensors, broadcasting, other dtypes -- is the reference's expression."""
    if (isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.is_cuda and y.is_cuda and x.device == y.device
            and x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape and x.numel() > 0
            and x.is_contiguous() and y.is_contiguous()):
        return _Img2MseFn.apply(x, y)
    return torch.mean((x - y) ** 2)


mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.Tensor([10.]).to(x.device))  # noqa: E731


def to8b(x):
    """uint8(255 * clip(x, 0, 1)) (utils.py:30).  numpy in -> numpy out exactly like the reference;
    a device tensor is quantised on the GPU (uint8 device tensor out, one quarter of the bytes to
    copy or gather afterwards)."""
    if isinstance(x, torch.Tensor):
        _lib.require_device(x, "x")
        src = x.detach().contiguous().float()
        if src.data_ptr() % 16:
            src = src.clone()                      # a view into the middle of a storage: the kernel loads 16 bytes at a time
        out = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
        with torch.cuda.device(src.device):
            _lib.check(lib.nerf_amd_to8b(src.data_ptr(), src.numel(), out.data_ptr(), _lib.stream_of(src.device)),
                       "nerf_amd_to8b")
        return out
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _default_device():
    if not torch.cuda.is_available():
        raise _lib.NerfAmdError("no ROCm device visible; nerf_shared_amd has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _host_pose(c2w):
    """First three rows of a camera-to-world matrix as 12 host floats (3x4 row-major)."""
    if isinstance(c2w, torch.Tensor):
        c2w = c2w.detach().cpu().numpy()
    c2w = np.asarray(c2w, dtype=np.float32)
    if c2w.ndim != 2 or c2w.shape[0] < 3 or c2w.shape[1] != 4:
        raise ValueError("c2w must be [>=3, 4], got %s" % (c2w.shape,))
    return np.ascontiguousarray(c2w[:3, :4])


def make_ray_batch(H, W, K, c2w, near, far, use_viewdirs, ndc, c2w_staticcam=None, device=None,
                   pix0=0, n=None):
    """get_rays + viewdirs + (ndc_rays) + near/far assembled as the [n, 8|11] batch
    that Renderer.render builds (render_utils.py:200-226), in one kernel, for the
    flat pixel range [pix0, pix0+n) (default: the whole image)."""
    device = device or _default_device()
    n = H * W - pix0 if n is None else n
    K4 = (ctypes.c_double * 4)(float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
    pose = _host_pose(c2w)
    pose_s = _host_pose(c2w_staticcam) if c2w_staticcam is not None else None
    ch = 11 if use_viewdirs else 8
    out = torch.empty(n, ch, device=device, dtype=torch.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    with torch.cuda.device(device):
        _lib.check(lib.nerf_amd_make_rays(int(H), int(W), K4, pose.ctypes.data_as(fp),
                                          pose_s.ctypes.data_as(fp) if pose_s is not None else None,
                                          int(pix0), int(n), float(near), float(far), int(bool(use_viewdirs)),
                                          int(bool(ndc)), out.data_ptr(), _lib.stream_of(device)),
                   "nerf_amd_make_rays")
    return out


class _GetRaysFn(torch.autograd.Function):
    """get_rays with a HIP backward with respect to the pose (nerf_amd_get_rays_backward)."""

    @staticmethod
    def forward(ctx, c2w, H, W, K4):
        b = make_ray_batch(H, W, [[K4[0], 0, K4[2]], [0, K4[1], K4[3]]], c2w, 0.0, 1.0, False, False, device=c2w.device)
        ctx.meta = (H, W, K4, tuple(c2w.shape))
        return b[:, 0:3].reshape(H, W, 3), b[:, 3:6].reshape(H, W, 3)

    @staticmethod
    def backward(ctx, g_o, g_d):
        H, W, K4, shape = ctx.meta
        g_o = None if g_o is None else g_o.reshape(-1, 3).contiguous().float()
        g_d = None if g_d is None else g_d.reshape(-1, 3).contiguous().float()
        dev = (g_o if g_o is not None else g_d).device
        out = torch.empty(12, device=dev, dtype=torch.float32)
        k4 = (ctypes.c_double * 4)(*K4)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_get_rays_backward(int(H), int(W), k4, 0, H * W, _lib.ptr(g_o), _lib.ptr(g_d),
                                                      out.data_ptr(), _lib.stream_of(dev)), "nerf_amd_get_rays_backward")
        g = torch.zeros(shape, device=dev, dtype=torch.float32)
        g[:3, :4] = out.reshape(3, 4)
        return g, None, None, None


def get_rays(H, W, K, c2w):
    """rays_o, rays_d [H, W, 3] (utils.py:33-42); pixel centres at integer coordinates, camera looks
    down -z.  Differentiable with respect to a device-resident ``c2w`` that requires grad."""
    if isinstance(c2w, torch.Tensor) and c2w.is_cuda and c2w.requires_grad and torch.is_grad_enabled():
        return _GetRaysFn.apply(c2w.float(), int(H), int(W), (float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])))
    dev = c2w.device if isinstance(c2w, torch.Tensor) and c2w.is_cuda else _default_device()
    b = make_ray_batch(H, W, K, c2w, 0.0, 1.0, False, False, device=dev)
    return b[:, 0:3].reshape(H, W, 3), b[:, 3:6].reshape(H, W, 3)


def get_rays_np(H, W, K, c2w):
    """numpy twin used by the reference's training-data batching (utils.py:45-52)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def _ndc_forward(H, W, focal, near, o, d):
    oo, od = torch.empty_like(o), torch.empty_like(d)
    with torch.cuda.device(d.device):
        _lib.check(lib.nerf_amd_ndc_rays(int(H), int(W), float(focal), float(near), o.data_ptr(), d.data_ptr(),
                                         o.shape[0], oo.data_ptr(), od.data_ptr(), _lib.stream_of(d.device)),
                   "nerf_amd_ndc_rays")
    return oo, od


class _NdcRaysFn(torch.autograd.Function):
    """ndc_rays with gradients with respect to the rays (nerf_amd_ndc_rays_backward)."""

    @staticmethod
    def forward(ctx, o, d, H, W, focal, near):
        ctx.save_for_backward(o, d)
        ctx.args = (int(H), int(W), float(focal), float(near))
        return _ndc_forward(H, W, focal, near, o, d)

    @staticmethod
    def backward(ctx, g_oo, g_od):
        o, d = ctx.saved_tensors
        H, W, focal, near = ctx.args
        g_oo = None if g_oo is None else g_oo.contiguous().float()
        g_od = None if g_od is None else g_od.contiguous().float()
        g_o, g_d = torch.empty_like(o), torch.empty_like(d)
        with torch.cuda.device(d.device):
            _lib.check(lib.nerf_amd_ndc_rays_backward(H, W, focal, near, o.data_ptr(), d.data_ptr(), _lib.ptr(g_oo), _lib.ptr(g_od),
                                                      o.shape[0], g_o.data_ptr(), g_d.data_ptr(), _lib.stream_of(d.device)),
                       "nerf_amd_ndc_rays_backward")
        return g_o, g_d, None, None, None, None


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Forward-facing NDC warp of explicit rays (utils.py:54-71).  Differentiable with respect to the rays."""
    _lib.require_device(rays_d, "rays_d")
    shape = rays_d.shape
    if torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad):
        o = rays_o.expand(shape).reshape(-1, 3).contiguous().float()
        d = rays_d.reshape(-1, 3).contiguous().float()
        oo, od = _NdcRaysFn.apply(o, d, H, W, focal, near)
    else:
        o = rays_o.detach().expand(shape).reshape(-1, 3).contiguous().float()
        d = rays_d.detach().reshape(-1, 3).contiguous().float()
        oo, od = _ndc_forward(H, W, focal, near, o, d)
    return oo.reshape(shape), od.reshape(shape)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Inverse-CDF sampling along each ray (utils.py:74-117).
    bins [R, M], weights [R, M-1] -> samples [R, N_samples]."""
    _lib.require_device(bins, "bins")
    dev = bins.device
    lead = list(bins.shape[:-1])
    b = bins.detach().reshape(-1, bins.shape[-1]).contiguous().float()
    w = weights.detach().reshape(-1, weights.shape[-1]).contiguous().float()
    if w.shape[-1] != b.shape[-1] - 1 or w.shape[0] != b.shape[0]:
        raise _lib.NerfAmdError("weights must be [..., len(bins)-1]")
    R = b.shape[0]
    u = t_lin = None
    if pytest:
        np.random.seed(0)
        if det:
            un = np.broadcast_to(np.linspace(0., 1., N_samples), [R, N_samples])
        else:
            un = np.random.rand(R, N_samples)
        u = torch.Tensor(np.ascontiguousarray(un)).to(dev)
    elif det:
        t_lin = torch.linspace(0., 1., steps=N_samples, device=dev)
    else:
        u = torch.rand([R, N_samples], device=dev)
    out = torch.empty(R, N_samples, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.nerf_amd_sample_pdf(b.data_ptr(), w.data_ptr(), _lib.ptr(u), _lib.ptr(t_lin), R,
                                           b.shape[-1], int(N_samples), out.data_ptr(), _lib.stream_of(dev)),
                   "nerf_amd_sample_pdf")
    return out.reshape(lead + [N_samples])


# ---------------------------------------------------------------- factories (utils.py:119-161)
_FIELD_SKIPS = [4]          # the reference hard-codes the skip layer (utils.py:122)


def create_nerf_models(args, device=None):
    """(coarse_model, fine_model) for an args namespace, what utils.py:119-139 builds: one field, or two when
    N_importance > 0 (then both emit 5 channels); the fine field takes its depth / width from
    netdepth_fine / netwidth_fine.  fine_model is None without importance sampling."""
    from . import nerf
    device = device or _default_device()
    two_pass = args.N_importance > 0
    shared = dict(output_ch=5 if two_pass else 4, skips=list(_FIELD_SKIPS), use_viewdirs=args.use_viewdirs,
                  multires=args.multires, multires_views=args.multires_views, i_embed=args.i_embed)
    sizes = [(args.netdepth, args.netwidth)] + ([(args.netdepth_fine, args.netwidth_fine)] if two_pass else [])
    fields = [nerf.NeRF(D=depth, W=width, **shared).to(device) for depth, width in sizes]
    return fields[0], (fields[1] if two_pass else None)


def get_renderer(args, bds_dict):
    """Renderer for an args namespace and the scene's {'near', 'far'} (utils.py:141-161).  NDC rays are used
    for forward-facing LLFF scenes only, and not when args.no_ndc asks for world-space sampling."""
    from . import render_utils
    ndc = args.dataset_type == 'llff' and not args.no_ndc
    if not ndc:
        print('Not ndc!')
    return render_utils.Renderer(perturb=args.perturb, N_importance=args.N_importance, N_samples=args.N_samples,
                                 use_viewdirs=args.use_viewdirs, white_bkgd=args.white_bkgd,
                                 raw_noise_std=args.raw_noise_std, ndc=ndc, lindisp=args.lindisp, **bds_dict)


# ---------------------------------------------------------------- optimizer + checkpoints (utils.py:163-214, 444-456)
def get_optimizer(coarse_model, fine_model, args):
    """Adam over both models' parameters, lr = args.lrate (utils.py:163-172).  With the parameters on the GPU this is
    `nerf_shared_amd.optim.Adam`: a torch.optim.Adam (same state, same checkpoints) whose step is one kernel launch for
    all 48 tensors -- the 1024-ray training step is host-bound otherwise.  CPU parameters get the plain torch optimizer."""
    params = list(coarse_model.parameters())
    if fine_model is not None:
        params += list(fine_model.parameters())
    if len(params) > 0 and all(p.is_cuda for p in params):
        from . import optim
        return optim.Adam(params=params, lr=args.lrate, betas=(0.9, 0.999))
    return torch.optim.Adam(params=params, lr=args.lrate, betas=(0.9, 0.999))


class CapturedTrainStep:
    """One iteration of the reference's training loop (main.py:77-104: render_from_rays -> img2mse(rgb) [+ img2mse(rgb0)]
    -> loss.backward() -> optimizer.step()) captured once in a HIP graph and replayed.

    The eager step costs about a hundred kernel launches; at the reference's batch size (N_rand = 1024) Python and the
    runtime spend as long enqueueing them as the GPU spends executing them.  A replay is one call.  What keeps the loop's
    semantics between replays:
      * rays / target are copied into the capture's input buffers on every call (any N_rand-ray batch of the same shape);
      * the learning rate is read from device memory: change optimizer.param_groups[i]['lr'] as main.py:108-112 does and the
        next call picks it up; the optimizer's step count advances on the device and is mirrored on the host;
      * gradients are left in `.grad` after every call (the captured step starts from zeroed gradients, like
        optimizer.zero_grad() at the top of the loop body);
      * random draws (perturb, raw_noise_std) advance with every replay (torch's graph-safe generator offsets).
    Everything the step touches must stay alive and in place (models, optimizer state).  Needs nerf_shared_amd.optim.Adam
    (utils.get_optimizer) and models the training kernels cover.  Constructing it leaves parameters, moments and step
    counts as they were (the warm-up steps the capture needs are undone).

        step = utils.CapturedTrainStep(renderer, H, W, K, chunk, coarse, fine, optimizer, N_rand)
        for i in range(n_iters):
            rays, target = sample_random_ray_batch(...)          # [2, N_rand, 3], [N_rand, 3]
            loss = step(rays, target)                            # device scalar; step.psnr likewise
            for g in optimizer.param_groups: g['lr'] = new_lrate
    """

    def __init__(self, renderer, H, W, K, chunk, coarse_model, fine_model, optimizer, n_rays, warmup=3):
        from . import optim
        if not isinstance(optimizer, optim.Adam):
            raise _lib.NerfAmdError("CapturedTrainStep needs nerf_shared_amd.optim.Adam (utils.get_optimizer): torch's optimizers "
                                    "pass step-dependent scalars as kernel arguments, which a graph would freeze")
        dev = next(coarse_model.parameters()).device
        self.renderer, self.models, self.optimizer = renderer, (coarse_model, fine_model), optimizer
        self.args = (H, W, K, chunk)
        self.rays = torch.zeros(2, n_rays, 3, device=dev)
        self.rays[1, :, 2] = -1.0                     # a valid placeholder batch for the warm-up (zero directions have no unit vector)
        self.target = torch.full((n_rays, 3), 0.5, device=dev)
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.loss = self.psnr = None
        self._lr = None
        # Constructing the object must not train: the warm-up steps (which the capture needs -- lazy allocations, the
        # optimizer's per-group fast path) run on the placeholder batch, and parameters, moments and step counts are put
        # back afterwards.
        with torch.no_grad():
            snap_p = [p.detach().clone() for p in self.params]
            snap_s = {p: {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in optimizer.state[p].items()}
                      for p in self.params if len(optimizer.state.get(p, {})) > 0}
        optimizer._sync_steps()
        steps_before = {gi: c["step"] for gi, c in optimizer._together.items()}
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for i in range(max(1, warmup)):          # (the first step also establishes the optimizer's per-group fast path)
                self._body()
                if i == 0:
                    optimizer.enable_device_scalars()
        torch.cuda.current_stream(dev).wait_stream(side)
        for m in self.models:
            if m is not None:
                m.weights_changed()                   # the re-pack of the parameters belongs inside the captured step
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        with torch.no_grad():
            for p, s0 in zip(self.params, snap_p):
                p.copy_(s0)
            for p in self.params:
                st = optimizer.state[p]
                if p in snap_s:
                    st["exp_avg"].copy_(snap_s[p]["exp_avg"])
                    st["exp_avg_sq"].copy_(snap_s[p]["exp_avg_sq"])
                elif len(st) > 0:
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
                if p.grad is not None:
                    p.grad.zero_()                    # (the capture's gradient tensors stay attached: replays fill them)
        for gi, c in optimizer._together.items():
            c["step"] = steps_before.get(gi, int(snap_s[c["params"][0]]["step"]) if c["params"][0] in snap_s else 0)
            optimizer._device_scalars[gi][0].fill_(c["step"])

    def _body(self):
        H, W, K, chunk = self.args
        coarse, fine = self.models
        for p in self.params:
            p.grad = None
        rgb, disp, acc, extras = self.renderer.render_from_rays(H, W, K, chunk, self.rays, coarse, fine, retraw=True)
        loss = img2mse(rgb, self.target)
        self.psnr = -10. * torch.log(loss.detach()) / math.log(10.)       # mse2psnr without its host-side constant tensor
        if 'rgb0' in extras:
            loss = loss + img2mse(extras['rgb0'], self.target)
        loss.backward()
        self.optimizer.step()
        self.loss = loss.detach()

    def __call__(self, rays, target):
        self.rays.copy_(rays if isinstance(rays, torch.Tensor) else torch.stack(list(rays), 0), non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        lr = tuple(float(g["lr"]) for g in self.optimizer.param_groups)
        if lr != self._lr:
            self.optimizer.sync_lr()
            self._lr = lr
        self.graph.replay()
        self.optimizer.note_replayed_step()
        return self.loss


def save_checkpoints(args, coarse_model, fine_model, optimizer, global_step, i):
    """<basedir>/<expname>/<i:06d>.tar with the reference's four keys (utils.py:444-456), so
    checkpoints move freely between this package and the reference."""
    path = os.path.join(args.basedir, args.expname, '{:06d}.tar'.format(i))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({
        'global_step': global_step,
        'coarse_model_state_dict': coarse_model.state_dict(),
        'fine_model_state_dict': fine_model.state_dict() if fine_model is not None else None,
        'optimizer_state_dict': optimizer.state_dict(),
    }, path)
    print('Saved checkpoints at', path)
    return path


def load_checkpoint(coarse_model, fine_model, optimizer, args, b_load_ckpnt_as_trainable=False, checkpoint_index=None):
    """Reload the newest (or the indexed) *.tar of the experiment, or args.ft_path (utils.py:174-214).
    Returns the stored global_step (0 when nothing was loaded)."""
    if getattr(args, 'ft_path', None) is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        folder = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if 'tar' in f] if os.path.isdir(folder) else []
    print('Found ckpts', ckpts)
    if not ckpts or getattr(args, 'no_reload', False):
        return 0
    path = ckpts[checkpoint_index] if checkpoint_index is not None else ckpts[-1]
    print('Reloading from', path)
    device = next(coarse_model.parameters()).device
    ckpt = torch.load(path, map_location=device)
    if optimizer is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    coarse_model.load_state_dict(ckpt['coarse_model_state_dict'], strict=False)
    coarse_model.requires_grad_(b_load_ckpnt_as_trainable)
    if fine_model is not None:
        fine_model.load_state_dict(ckpt['fine_model_state_dict'])
        fine_model.requires_grad_(b_load_ckpnt_as_trainable)
    return ckpt['global_step']


# ---------------------------------------------------------------- datasets (utils.py:216-313)
def _scene_llff(args):
    """LLFF scene -> images, poses [N,3,4], render_poses, hwf, splits, (near, far)  (utils.py:220-252)."""
    from . import load_llff
    images, poses, bds, render_poses, i_test = load_llff.load_llff_data(
        args.datadir, args.factor, recenter=True, bd_factor=.75, spherify=args.spherify)
    hwf, poses = poses[0, :3, -1], poses[:, :3, :4]
    print('Loaded llff', images.shape, render_poses.shape, hwf, args.datadir)
    held_out = list(i_test) if isinstance(i_test, (list, tuple, np.ndarray)) else [i_test]
    if args.llffhold > 0:
        print('Auto LLFF holdout,', args.llffhold)
        held_out = np.arange(images.shape[0])[::args.llffhold]
    i_train = np.array([i for i in range(int(images.shape[0])) if i not in held_out])
    # world-space depth range from the scene bounds, or the unit NDC range
    near_far = (np.ndarray.min(bds) * .9, np.ndarray.max(bds) * 1.) if args.no_ndc else (0., 1.)
    print('NEAR FAR', *near_far)
    return images, poses, render_poses, hwf, (i_train, held_out, held_out), near_far


def _scene_blender(args):
    """NeRF-synthetic scene; RGBA is composited over white or the alpha dropped  (utils.py:254-266)."""
    from . import load_blender
    images, poses, render_poses, hwf, i_split, near, far = load_blender.load_blender_data(
        args.datadir, args.half_res, args.testskip)
    print('Loaded blender', images.shape, render_poses.shape, hwf, args.datadir)
    rgb, alpha = images[..., :3], images[..., -1:]
    images = rgb * alpha + (1. - alpha) if args.white_bkgd else rgb
    return images, poses, render_poses, hwf, tuple(i_split), (near, far)


_SCENE_READERS = {'llff': _scene_llff, 'blender': _scene_blender}


def load_datasets(args):
    """Scene on disk -> (images, poses, render_poses, [H, W, focal], (i_train, i_val, i_test), K,
    {'near', 'far'}) for dataset_type 'blender' and 'llff' (utils.py:216-313).  The LINEMOD and
    deepvoxels loaders of the reference are not part of this build."""
    reader = _SCENE_READERS.get(args.dataset_type)
    if reader is None:
        raise NotImplementedError("dataset_type %r: this build reads %s scenes" % (args.dataset_type, sorted(_SCENE_READERS)))
    images, poses, render_poses, hwf, (i_train, i_val, i_test), (near, far) = reader(args)
    H, W, focal = int(hwf[0]), int(hwf[1]), hwf[2]
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    if getattr(args, 'render_test', False):
        render_poses = np.array(poses[i_test])
    return images, poses, render_poses, [H, W, focal], (i_train, i_val, i_test), K, {'near': near, 'far': far}


# ---------------------------------------------------------------- training ray batches (utils.py:360-442)
def batch_training_data(args, poses, hwf, K, images, i_train):
    """Device-resident version of utils.py:360-392: with ray batching (not args.no_batching) the
    rays of every training image are generated on the GPU (make_rays kernel, no host meshgrid),
    joined with their pixels into rays_rgb [N_train*H*W, 3, 3] = (origin, direction, colour) and
    shuffled there.  Returns the reference's tuple."""
    H, W = int(hwf[0]), int(hwf[1])
    device = _default_device()
    images = torch.as_tensor(np.asarray(images), dtype=torch.float32, device=device) if not isinstance(images, torch.Tensor) \
        else images.to(device).float()
    poses = torch.as_tensor(np.asarray(poses), dtype=torch.float32, device=device) if not isinstance(poses, torch.Tensor) \
        else poses.to(device).float()
    use_batching = not args.no_batching
    if not use_batching:
        return images, poses, torch.empty(0, device=device), use_batching, args.N_rand, None
    blocks = []
    for i in i_train:
        b = make_ray_batch(H, W, K, poses[i, :3, :4], 0.0, 1.0, False, False, device=device)     # [H*W, 8]
        blocks.append(torch.stack([b[:, 0:3], b[:, 3:6], images[i].reshape(-1, 3)[:, :3]], 1))  # [H*W, 3, 3]
    rays_rgb = torch.cat(blocks, 0)
    rays_rgb = rays_rgb[torch.randperm(rays_rgb.shape[0], device=device)]
    return images, poses, rays_rgb, use_batching, args.N_rand, 0


def sample_random_ray_batch(args, images, poses, rays_rgb, N_rand, use_batching, i_batch, i_train, hwf, K, start, i):
    """One training batch (utils.py:394-442): the next N_rand rows of the shuffled ray bank
    (reshuffled on the device after an epoch), or N_rand random pixels of one random training
    image, centre-cropped during the first args.precrop_iters iterations.
    Returns batch_rays [2, N_rand, 3], target_s [N_rand, 3], rays_rgb, i_batch."""
    H, W = int(hwf[0]), int(hwf[1])
    if use_batching:
        batch = rays_rgb[i_batch:i_batch + N_rand].transpose(0, 1)
        batch_rays, target_s = batch[:2], batch[2]
        i_batch += N_rand
        if i_batch >= rays_rgb.shape[0]:
            print("Shuffle data after an epoch!")
            rays_rgb = rays_rgb[torch.randperm(rays_rgb.shape[0], device=rays_rgb.device)]
            i_batch = 0
        return batch_rays, target_s, rays_rgb, i_batch
    img_i = int(np.random.choice(i_train))
    target = images[img_i]
    rays_o, rays_d = get_rays(H, W, K, poses[img_i, :3, :4])
    if i < getattr(args, 'precrop_iters', 0):
        dH, dW = int(H // 2 * args.precrop_frac), int(W // 2 * args.precrop_frac)
        ys = torch.arange(H // 2 - dH, H // 2 + dH, device=target.device)
        xs = torch.arange(W // 2 - dW, W // 2 + dW, device=target.device)
        if i == start:
            print(f"[Config] Center cropping of size {2*dH} x {2*dW} is enabled until iter {args.precrop_iters}")
    else:
        ys, xs = torch.arange(H, device=target.device), torch.arange(W, device=target.device)
    n = ys.numel() * xs.numel()
    if N_rand > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")    # np.random.choice's error
    sel = torch.randperm(n, device=target.device)[:N_rand]            # without replacement, as np.random.choice(replace=False)
    yy, xx = ys[sel // xs.numel()], xs[sel % xs.numel()]
    batch_rays = torch.stack([rays_o[yy, xx], rays_d[yy, xx]], 0)
    return batch_rays, target[yy, xx][:, :3], rays_rgb, i_batch
