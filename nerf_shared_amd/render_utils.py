"""Drop-in for nerf_shared/render_utils.py: the volumetric Renderer, running on
the MI355X kernels behind include/nerf_amd.h.

Same constructor, methods, argument names, return structures and key order as
/root/reference/nerf_shared/render_utils.py:13-319.  Differences, all at the
edges of the hot path:
  * temporaries are allocated on ``ray_batch.device`` (the reference uses the
    process-wide default device, main.py:150-152);
  * random draws use torch's generator for that device in the reference's order
    and shapes -- per render_rays call, i.e. per chunk of render_batch: t_rand,
    coarse noise, u, fine noise (render_utils.py:121,264, utils.py:86) -- so a
    seeded multi-chunk render equals seeded per-chunk render_rays calls in every
    mode of render_batch; ``pytest=True`` reproduces the reference's seeded numpy
    draws bit for bit;
  * the NaN/Inf scan of every output (render_utils.py:170-172) only runs when
    DEBUG is set -- it never changes outputs and costs a device sync per key;
  * autograd reaches the models' parameters and the rays (standard model, bf16 mode).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, utils
from . import nerf as nerf_mod
from ._lib import lib
from .nerf import NeRF

DEBUG = False

_KEYS_MAIN = ('rgb_map', 'disp_map', 'acc_map')


def _workspace(device, nbytes):
    """Scratch for one render_rays call (raw / z / weights of the two passes), allocated per call on the
    stream that is current: torch's caching allocator recycles the block and keeps its reuse ordered
    on that stream, so renders enqueued on different streams never share scratch."""
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


_side_streams = {}
_draw_streams = {}


def _draw_stream(device):
    """The stream the whole-batch path makes its random draws on (see Renderer._launch_batch)."""
    s = _draw_streams.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _draw_streams[device] = s
    return s


def _chunk_streams(device, n=2):
    """The two side streams of the opt-in chunk-overlap mode of render_batch."""
    s = _side_streams.get(device)
    if s is None:
        s = [torch.cuda.Stream(device=device) for _ in range(n)]
        _side_streams[device] = s
    return s


_linspace_cache = {}


def _linspace01(n, device):
    key = (n, device)
    t = _linspace_cache.get(key)
    if t is None:
        t = torch.linspace(0., 1., steps=n, device=device)
        _linspace_cache[key] = t
    return t


def _pytest_uniform(shape, device):
    """np.random.seed(0); np.random.rand(*shape) -> fp32 (render_utils.py:124-127)."""
    np.random.seed(0)
    return torch.Tensor(np.random.rand(*list(shape))).to(device)


class _Raw2OutputsFn(torch.autograd.Function):
    """raw2outputs with a HIP backward (nerf_amd_raw2outputs_backward) with respect to ``raw`` and to
    ``rays_d`` (dists scale with |rays_d|, render_utils.py:259).  z_vals are constants."""

    @staticmethod
    def forward(ctx, raw, z, d, noise, white):
        dev = raw.device
        R, S, ch = raw.shape
        f = dict(device=dev, dtype=torch.float32)
        rgb_map, disp_map, acc_map = torch.empty(R, 3, **f), torch.empty(R, **f), torch.empty(R, **f)
        weights, depth_map = torch.empty(R, S, **f), torch.empty(R, **f)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_raw2outputs(raw.data_ptr(), ch, z.data_ptr(), d.data_ptr(), d.stride(0), _lib.ptr(noise),
                                                R, S, int(white), rgb_map.data_ptr(), disp_map.data_ptr(),
                                                acc_map.data_ptr(), weights.data_ptr(), depth_map.data_ptr(),
                                                _lib.stream_of(dev)), "nerf_amd_raw2outputs")
        ctx.save_for_backward(raw, z, d, noise if noise is not None else torch.empty(0, device=dev))
        ctx.white = white
        ctx.set_materialize_grads(False)
        return rgb_map, disp_map, acc_map, weights, depth_map

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw, z, d, noise = ctx.saved_tensors
        noise = noise if noise.numel() else None
        R, S, ch = raw.shape
        g = [None if t is None else t.contiguous().float() for t in (g_rgb, g_disp, g_acc, g_depth, g_w)]
        g_raw = torch.empty_like(raw)
        g_d = torch.empty(R, 3, device=raw.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(raw.device):
            _lib.check(lib.nerf_amd_raw2outputs_backward(raw.data_ptr(), ch, z.data_ptr(), d.data_ptr(), d.stride(0),
                                                         _lib.ptr(noise), R, S, int(ctx.white), _lib.ptr(g[0]),
                                                         _lib.ptr(g[1]), _lib.ptr(g[2]), _lib.ptr(g[3]), _lib.ptr(g[4]),
                                                         g_raw.data_ptr(), _lib.ptr(g_d), _lib.stream_of(raw.device)),
                       "nerf_amd_raw2outputs_backward")
        return g_raw, None, g_d, None, None


class Renderer(torch.nn.Module):
    def __init__(self, perturb=True, N_importance=128, N_samples=64, use_viewdirs=True,
                 white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False,
                 near=0.0, far=1.0):
        """
        Stores values as class data.
        """
        super(Renderer, self).__init__()
        self.perturb = perturb
        self.N_importance = N_importance
        self.N_samples = N_samples
        self.use_viewdirs = use_viewdirs
        self.white_bkgd = white_bkgd
        self.raw_noise_std = raw_noise_std
        self.ndc = ndc
        self.lindisp = lindisp
        self.near = near
        self.far = far
        if os.environ.get("NERF_AMD_QUIET") != "1":
            print(self.__dict__)     # the reference prints its configuration (render_utils.py:32)

    # ------------------------------------------------------------------ wrappers
    def render_from_pose(self, H, W, K, chunk, c2w, coarse_model, fine_model, retraw=True):
        rgb, disp, acc, extras = self.render(
            H, W, K, coarse_model, fine_model, chunk=chunk, c2w=c2w, retraw=retraw)
        return rgb, disp, acc, extras

    def render_from_rays(self, H, W, K, chunk, rays, coarse_model, fine_model, retraw=True):
        rgb, disp, acc, extras = self.render(H, W, K, coarse_model, fine_model,
                                             chunk=chunk, rays=rays, retraw=retraw)
        return rgb, disp, acc, extras

    def render_path(self):
        pass

    # ------------------------------------------------------------------ core
    def _cfg(self, precision):
        cfg = _lib.RenderCfg()
        cfg.N_samples, cfg.N_importance = int(self.N_samples), int(self.N_importance)
        cfg.perturb = int(self.perturb > 0.)
        cfg.lindisp, cfg.white_bkgd = int(bool(self.lindisp)), int(bool(self.white_bkgd))
        cfg.use_noise = int(self.raw_noise_std > 0.)
        cfg.precision = precision
        return cfg

    @staticmethod
    def _check_model(m, name):
        if not isinstance(m, NeRF):
            raise TypeError("%s must be a nerf_shared_amd.nerf.NeRF or a reference-style NeRF module (got %s)" % (name, type(m).__name__))

    def _handles(self, dev, coarse_model, fine_model):
        """Library handles (re-packed if the weights changed), render cfg and output width."""
        self._check_model(coarse_model, "coarse_model")
        if fine_model is not None:
            self._check_model(fine_model, "fine_model")
        use_fine = fine_model is not None and int(self.N_importance) > 0
        coarse_model._ensure_handle(dev)
        if use_fine:
            fine_model._ensure_handle(dev)
        prec = coarse_model._precision_code()
        if use_fine and fine_model._precision_code() != prec:
            prec = _lib.PREC_FP32
        hc = coarse_model._model_handle(dev, _lib.COPY_OF[prec])      # only the packed copy this precision reads
        hf = fine_model._model_handle(dev, _lib.COPY_OF[prec]) if use_fine else None
        return hc, hf, self._cfg(prec), lib.nerf_amd_model_out_ch(hc)

    def _chunk_io(self, rays, cfg, out_ch, outs, pytest, ws=None, z_pre=None, t_rand_out=None):
        """nerf_amd_render_io of one render_rays call on contiguous fp32 rays [R, 8|11] writing into the tensors
        of ``outs`` (rows [0, R)): random draws in the reference's order and shapes (t_rand, noise0, u, noise1).
        ws: workspace to use (default: a fresh one).  z_pre: this chunk's rows of a coarse-depth buffer that one
        launch fills for all chunks afterwards; the jitter draws it needs go into t_rand_out (same rows).
        Returns (io, tensors to keep alive)."""
        R, dev = rays.shape[0], rays.device
        Nc, Ni = int(self.N_samples), int(self.N_importance)
        Nf = Nc + Ni
        t_rand = noise0 = noise1 = u = None
        if self.perturb > 0.:
            if t_rand_out is not None:
                t_rand_out.uniform_()        # the same draw as torch.rand([R, Nc]): rand() is empty().uniform_(0, 1)
            else:
                t_rand = _pytest_uniform([R, Nc], dev) if pytest else torch.rand([R, Nc], device=dev)
        if self.raw_noise_std > 0.:
            noise0 = (_pytest_uniform([R, Nc], dev) if pytest else torch.randn([R, Nc], device=dev)) * self.raw_noise_std
        t_lin = None
        if Ni > 0:
            det = (self.perturb == 0.)
            if pytest:
                np.random.seed(0)
                if det:
                    u = np.broadcast_to(np.linspace(0., 1., Ni), [R, Ni])
                else:
                    u = np.random.rand(R, Ni)
                u = torch.Tensor(np.ascontiguousarray(u)).to(dev)
            elif det:
                t_lin = _linspace01(Ni, dev)
            else:
                u = torch.rand([R, Ni], device=dev)
            if self.raw_noise_std > 0.:
                noise1 = (_pytest_uniform([R, Nf], dev) if pytest else torch.randn([R, Nf], device=dev)) * self.raw_noise_std

        io = _lib.RenderIO()
        io.rays, io.ray_ch = rays.data_ptr(), rays.shape[1]
        io.t_vals = _linspace01(Nc, dev).data_ptr()
        io.t_rand, io.noise0, io.noise1 = _lib.ptr(t_rand), _lib.ptr(noise0), _lib.ptr(noise1)
        io.u, io.t_lin_imp = _lib.ptr(u), _lib.ptr(t_lin)
        io.z_coarse = _lib.ptr(z_pre)
        for k in ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'raw', 'weights', 'z_vals'):
            setattr(io, k, _lib.ptr(outs.get(k)))
        nbytes = lib.nerf_amd_render_rays_workspace(cfg, R, out_ch)
        if ws is None or ws.numel() < nbytes:
            ws = _workspace(dev, nbytes)
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
        return io, (rays, t_rand, noise0, noise1, u, z_pre, ws)

    def _launch(self, rays, coarse_model, fine_model, outs, retraw, retweights, pytest):
        """Enqueue render_rays for contiguous fp32 rays [R, 8|11]; writes into the
        tensors of ``outs`` (rows [0, R))."""
        dev = rays.device
        hc, hf, cfg, out_ch = self._handles(dev, coarse_model, fine_model)
        io, keep = self._chunk_io(rays, cfg, out_ch, outs, pytest)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_render_rays(cfg, hc, hf, io, rays.shape[0], _lib.stream_of(dev)), "nerf_amd_render_rays")
        # keep the draws alive until the stream has consumed them (caching allocator is stream-ordered)
        return keep

    def _launch_batch(self, rays, starts, chunk, coarse_model, fine_model, full):
        """render_batch as ONE library call over whole-batch buffers (nerf_amd_render_batch).  The random draws are made
        per chunk, in the reference's order (t_rand, noise0, u, noise1 of chunk 0, then of chunk 1, ...), into the rows of
        one buffer each -- `slice.uniform_()` / `slice.normal_()` consume the generator exactly like the reference's
        torch.rand / torch.randn of the same shape -- so the result equals per-chunk render_rays calls bit for bit, while
        the library is free to launch in groups that suit the GPU and to run the per-ray kernels beside the field kernels."""
        dev = rays.device
        hc, hf, cfg, out_ch = self._handles(dev, coarse_model, fine_model)
        N, Nc, Ni = rays.shape[0], int(self.N_samples), int(self.N_importance)
        f = dict(device=dev, dtype=torch.float32)
        z_all = torch.empty(N, Nc, **f)
        noisy = self.raw_noise_std > 0.
        t_all = n0_all = n1_all = u_all = None
        if self.perturb > 0. or noisy:
            # The draws depend on nothing but the generator (whose state lives on the host), so they go on a stream of
            # their own: the dozens of small RNG launches of this call then run beside whatever the device is still busy
            # with (the field kernels of the previous frame, when frames are rendered back to back) instead of in front
            # of this frame's first field kernel.  Same values as on the caller's stream.
            cur, ds = torch.cuda.current_stream(dev), _draw_stream(dev)
            with torch.cuda.stream(ds):
                t_all = torch.empty(N, Nc, **f) if self.perturb > 0. else None
                n0_all = torch.empty(N, Nc, **f) if noisy else None
                n1_all = torch.empty(N, Nc + Ni, **f) if (noisy and Ni > 0) else None
                u_all = torch.empty(N, Ni, **f) if (Ni > 0 and self.perturb > 0.) else None
                for i in starts:                               # the reference's order inside every render_rays call
                    if t_all is not None:
                        t_all[i:i + chunk].uniform_()
                    if n0_all is not None:
                        n0_all[i:i + chunk].normal_()
                    if u_all is not None:
                        u_all[i:i + chunk].uniform_()
                    if n1_all is not None:
                        n1_all[i:i + chunk].normal_()
                if noisy:
                    n0_all.mul_(self.raw_noise_std)
                    if n1_all is not None:
                        n1_all.mul_(self.raw_noise_std)
                drawn = torch.cuda.Event()
                drawn.record(ds)
            cur.wait_event(drawn)
            for t in (t_all, n0_all, n1_all, u_all):           # allocated on the draw stream, consumed on the caller's
                if t is not None:
                    t.record_stream(cur)
        io = _lib.RenderIO()
        io.rays, io.ray_ch = rays.data_ptr(), rays.shape[1]
        io.t_vals = _linspace01(Nc, dev).data_ptr()
        io.t_rand, io.noise0, io.noise1, io.u = _lib.ptr(t_all), _lib.ptr(n0_all), _lib.ptr(n1_all), _lib.ptr(u_all)
        io.t_lin_imp = _linspace01(Ni, dev).data_ptr() if (Ni > 0 and u_all is None) else None
        io.z_coarse = z_all.data_ptr()
        for k in ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'raw', 'weights', 'z_vals'):
            setattr(io, k, _lib.ptr(full.get(k)))
        ws = _workspace(dev, lib.nerf_amd_render_batch_workspace(cfg, N, out_ch))
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_coarse_z(rays.data_ptr(), rays.shape[1], _linspace01(Nc, dev).data_ptr(),
                                             _lib.ptr(t_all), N, Nc, int(bool(self.lindisp)), int(self.perturb > 0.),
                                             z_all.data_ptr(), _lib.stream_of(dev)), "nerf_amd_coarse_z")
            _lib.check(lib.nerf_amd_render_batch(cfg, hc, hf, io, N, _lib.stream_of(dev)), "nerf_amd_render_batch")
        # the library's side stream reads these too; its work is ordered before the caller's stream continues, so
        # releasing them to the caching allocator (stream-ordered on the caller's stream) is safe
        return (rays, z_all, t_all, n0_all, n1_all, u_all, ws)

    def _launch_chunks(self, rays, starts, chunk, coarse_model, fine_model, full):
        """render_batch's chunk loop as one library call (nerf_amd_render_chunks): per-chunk draws (in the
        reference's per-chunk order) and output rows exactly as _launch makes them, two workspaces used
        alternately, and the coarse depths of all chunks from one launch in front."""
        dev = rays.device
        hc, hf, cfg, out_ch = self._handles(dev, coarse_model, fine_model)
        n, N, Nc = len(starts), rays.shape[0], int(self.N_samples)
        ios = (_lib.RenderIO * n)()
        counts = (ctypes.c_int64 * n)()
        keep = []
        z_all = torch.empty(N, Nc, device=dev, dtype=torch.float32)
        t_all = torch.empty(N, Nc, device=dev, dtype=torch.float32) if self.perturb > 0. else None
        nbytes = lib.nerf_amd_render_rays_workspace(cfg, min(chunk, N), out_ch)
        wss = [_workspace(dev, nbytes), _workspace(dev, nbytes)]
        for j, i in enumerate(starts):
            part = {k: v[i:i + chunk] for k, v in full.items()}
            r = rays[i:i + chunk]
            ios[j], alive = self._chunk_io(r, cfg, out_ch, part, False, wss[j % 2], z_all[i:i + chunk],
                                           None if t_all is None else t_all[i:i + chunk])
            counts[j] = r.shape[0]
            keep.append(alive)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_coarse_z(rays.data_ptr(), rays.shape[1], _linspace01(Nc, dev).data_ptr(),
                                             _lib.ptr(t_all), N, Nc, int(bool(self.lindisp)), int(self.perturb > 0.),
                                             z_all.data_ptr(), _lib.stream_of(dev)), "nerf_amd_coarse_z")
            _lib.check(lib.nerf_amd_render_chunks(cfg, hc, hf, ios, counts, n, _lib.stream_of(dev)), "nerf_amd_render_chunks")
        return keep

    def _draws(self, R, dev, pytest):
        """Random draws of one render_rays call in the reference's order and shapes."""
        Nc, Ni = int(self.N_samples), int(self.N_importance)
        t_rand = noise0 = noise1 = u = t_lin = None
        if self.perturb > 0.:
            t_rand = _pytest_uniform([R, Nc], dev) if pytest else torch.rand([R, Nc], device=dev)
        if self.raw_noise_std > 0.:
            noise0 = (_pytest_uniform([R, Nc], dev) if pytest else torch.randn([R, Nc], device=dev)) * self.raw_noise_std
        if Ni > 0:
            det = (self.perturb == 0.)
            if pytest:
                np.random.seed(0)
                un = np.broadcast_to(np.linspace(0., 1., Ni), [R, Ni]) if det else np.random.rand(R, Ni)
                u = torch.Tensor(np.ascontiguousarray(un)).to(dev)
            elif det:
                t_lin = _linspace01(Ni, dev)
            else:
                u = torch.rand([R, Ni], device=dev)
            if self.raw_noise_std > 0.:
                noise1 = (_pytest_uniform([R, Nc + Ni], dev) if pytest else torch.randn([R, Nc + Ni], device=dev)) * self.raw_noise_std
        return t_rand, noise0, noise1, u, t_lin

    def _render_rays_train(self, rays, coarse_model, fine_model, retraw, retweights, pytest):
        """render_rays with autograd into the models' parameters (what main.py:77-104 needs):
        the same kernels stage by stage -- z_vals, training forward of the field (activations
        saved), compositing -- with the field and raw2outputs as autograd Functions whose
        backward passes are HIP kernels.  z_samples are detached, as in the reference
        (render_utils.py:145); `rays` may require grad (pose estimation)."""
        R, dev = rays.shape[0], rays.device
        Nc, Ni = int(self.N_samples), int(self.N_importance)
        t_rand, noise0, noise1, u, t_lin = self._draws(R, dev, pytest)
        f = dict(device=dev, dtype=torch.float32)
        stream = _lib.stream_of(dev)
        z = torch.empty(R, Nc, **f)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_coarse_z(rays.data_ptr(), rays.shape[1], _linspace01(Nc, dev).data_ptr(), _lib.ptr(t_rand),
                                             R, Nc, int(bool(self.lindisp)), int(self.perturb > 0.), z.data_ptr(), stream),
                       "nerf_amd_coarse_z")
        rays_d = rays[:, 3:6]            # a view: composite gradients with respect to |d| flow back into `rays`

        def field(net, zz):
            # a model that asks for no gradient (frozen, and the rays carry none) keeps its own forward-only kernel and
            # precision; only models that train go through the training kernels
            if net._grad_request(dev, rays)[0] == "train":
                return net.forward_rays(rays, zz)
            with torch.no_grad():
                pts = rays[:, None, 0:3] + rays[:, None, 3:6] * zz[:, :, None]      # render_utils.py:131, two rounded ops
                return net.forward(pts, rays[:, 8:11] if rays.shape[1] > 8 else None)

        raw = field(coarse_model, z)
        rgb, disp, acc, weights, _ = _Raw2OutputsFn.apply(raw, z, rays_d, noise0, bool(self.white_bkgd))
        ret = {}
        if Ni > 0:
            rgb0, disp0, acc0 = rgb, disp, acc
            z_f, z_std = torch.empty(R, Nc + Ni, **f), torch.empty(R, **f)
            w_c = weights.detach()
            with torch.cuda.device(dev):
                _lib.check(lib.nerf_amd_resample(z.data_ptr(), w_c.data_ptr(), _lib.ptr(u), _lib.ptr(t_lin), R, Nc, Ni,
                                                 z_f.data_ptr(), z_std.data_ptr(), stream), "nerf_amd_resample")
            z = z_f
            net = coarse_model if fine_model is None else fine_model
            raw = field(net, z)
            rgb, disp, acc, weights, _ = _Raw2OutputsFn.apply(raw, z, rays_d, noise1, bool(self.white_bkgd))
        ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc)
        if retraw:
            ret['raw'] = raw
        if retweights:
            ret['weights'] = weights if z.shape[1] > 1 else weights[:, :0]     # one sample: see raw2outputs
            ret['z_vals'] = z
        if Ni > 0:
            ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0, z_std=z_std)
        return ret

    def _alloc_outputs(self, R, dev, out_ch, retraw, retweights):
        Ni = int(self.N_importance)
        S_last = int(self.N_samples) + Ni
        f = dict(device=dev, dtype=torch.float32)
        outs = {'rgb_map': torch.empty(R, 3, **f), 'disp_map': torch.empty(R, **f), 'acc_map': torch.empty(R, **f)}
        if retraw:
            outs['raw'] = torch.empty(R, S_last, out_ch, **f)
        if retweights:
            outs['weights'] = torch.empty(R, S_last, **f)
            outs['z_vals'] = torch.empty(R, S_last, **f)
        if Ni > 0:
            outs['rgb0'] = torch.empty(R, 3, **f)
            outs['disp0'] = torch.empty(R, **f)
            outs['acc0'] = torch.empty(R, **f)
            outs['z_std'] = torch.empty(R, **f)
        return outs

    def _grad_mode(self, coarse_model, fine_model, rays):
        """The models' gradient requests for one call, combined: "train" only if every model that asks for gradients is
        covered by the training kernels; one uncovered model defers the whole call (nerf.NeRF._grad_request)."""
        reqs = [coarse_model._grad_request(rays.device, rays)]
        if fine_model is not None and self.N_importance > 0:
            reqs.append(fine_model._grad_request(rays.device, rays))
        for m, d in reqs:
            if m == "defer":
                return m, d
        return ("train", None) if any(m == "train" for m, _ in reqs) else ("none", None)

    def render_rays(self, ray_batch, coarse_model, fine_model, retraw=False, retweights=False,
                    verbose=False, pytest=False):
        """Volumetric rendering of one ray batch (render_utils.py:67-174).

        ray_batch [N, 8|11] = origin, direction, near, far, (unit view dir).
        Returns the reference's dict: rgb_map, disp_map, acc_map, [raw],
        [weights, z_vals], and with N_importance > 0: rgb0, disp0, acc0, z_std.
        """
        _lib.require_device(ray_batch, "ray_batch")
        coarse_model, fine_model = nerf_mod.adopt(coarse_model), nerf_mod.adopt(fine_model)     # reference-class models: a twin sharing their parameters
        rays = ray_batch.detach().contiguous().float()
        if rays.dim() != 2 or rays.shape[1] not in (8, 11):
            raise _lib.NerfAmdError("ray_batch must be [N, 8] or [N, 11], got %s" % (tuple(ray_batch.shape),))
        self._check_model(coarse_model, "coarse_model")
        if fine_model is not None:
            self._check_model(fine_model, "fine_model")
        mode, deferred = self._grad_mode(coarse_model, fine_model, ray_batch) if rays.shape[0] > 0 else ("none", None)
        if mode == "train":
            live = ray_batch.contiguous().float() if (torch.is_grad_enabled() and ray_batch.requires_grad) else rays
            return self._render_rays_train(live, coarse_model, fine_model, retraw, retweights, pytest)
        out_ch = 4 if coarse_model.use_viewdirs else coarse_model.output_ch
        outs = self._alloc_outputs(rays.shape[0], rays.device, out_ch, retraw, retweights)
        self._launch(rays, coarse_model, fine_model, outs, retraw, retweights, pytest)
        if retweights and outs['weights'].shape[1] == 1:
            outs['weights'] = outs['weights'][:, :0]                           # one sample: see raw2outputs
        ret = nerf_mod.attach_deferred_grad(outs, deferred)
        if DEBUG:
            for k in ret:
                if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                    print(f"! [Numerical Error] {k} contains nan or inf.")
        return ret

    pipeline_batch = True         # render_batch: one library call over whole-batch buffers (nerf_amd_render_batch): launch
                                  # groups of 32768 rays, per-ray kernels on a side stream beside the next field kernel
    fuse_chunk_launches = True    # (pipeline_batch off) one library call per chunk list (three dependent launches per chunk)
    overlap_chunks = False        # (pipeline_batch off) opt-in: two-stream chunk pipeline, two field kernels at a time

    def render_batch(self, coarse_model, fine_model, rays_flat, chunk=1024 * 32, retraw=False):
        """Render rays in chunks (render_utils.py:51-65).  Outputs of every chunk land
        directly in one preallocated tensor per key (the reference's per-key torch.cat,
        without the copy).  By default (``Renderer.pipeline_batch``) the whole batch goes to the
        library in one call: `chunk` then only decides how the random draws are grouped (as in
        the reference), not how the GPU is launched -- every kernel is per-ray independent, so the
        result is the same bit for bit.  The older modes remain for A/B runs:
        with ``Renderer.overlap_chunks = True`` consecutive chunks
        alternate between two HIP streams (each with its own workspace) so the small
        per-ray kernels and launch/drain gaps of one chunk hide under the field kernel
        of the other; results do not depend on it (random draws come from the device's one
        generator in chunk order in every mode).  Off by default: the gain is ~3 % and
        concurrent kernels blur per-kernel timings."""
        _lib.require_device(rays_flat, "rays_flat")
        coarse_model, fine_model = nerf_mod.adopt(coarse_model), nerf_mod.adopt(fine_model)
        rays = rays_flat.detach().contiguous().float()
        self._check_model(coarse_model, "coarse_model")
        N, dev = rays.shape[0], rays.device
        out_ch = 4 if coarse_model.use_viewdirs else coarse_model.output_ch
        starts = list(range(0, N, chunk))
        mode, deferred = self._grad_mode(coarse_model, fine_model, rays_flat) if N > 0 else ("none", None)
        if mode == "defer":            # forward-only kernels now, a backward that explains itself later (nerf._grad_request)
            with torch.no_grad():
                plain = self.render_batch(coarse_model, fine_model, rays_flat, chunk, retraw)
            return nerf_mod.attach_deferred_grad(plain, deferred)
        if mode == "train":
            live = rays_flat if (torch.is_grad_enabled() and rays_flat.requires_grad) else rays
            parts = {}                     # training: autograd graph per chunk, concatenated like the reference
            for i in starts:
                r = self.render_rays(live[i:i + chunk], coarse_model, fine_model, retraw)
                for k, v in r.items():
                    parts.setdefault(k, []).append(v)
            # (one chunk: the reference's torch.cat of a single tensor is a copy of it)
            return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in parts.items()}
        full = self._alloc_outputs(N, dev, out_ch, retraw, False)
        if self.pipeline_batch and N > 0 and (len(starts) > 1 or N > 49152):
            self._launch_batch(rays, starts, chunk, coarse_model, fine_model, full)
            return full
        if not (self.overlap_chunks and len(starts) > 1):
            if len(starts) > 1 and self.fuse_chunk_launches:
                self._launch_chunks(rays, starts, chunk, coarse_model, fine_model, full)
                return full
            for i in starts:
                part = {k: v[i:i + chunk] for k, v in full.items()}
                self._launch(rays[i:i + chunk], coarse_model, fine_model, part, retraw, False, False)
            return full
        # pack parameters and build the cached linspaces on the caller's stream first, then fan out
        _linspace01(int(self.N_samples), dev)
        if self.N_importance > 0:
            _linspace01(int(self.N_importance), dev)
        self._handles(dev, coarse_model, fine_model)
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        side = _chunk_streams(dev)
        for s in side:
            s.wait_event(ready)
        for n, i in enumerate(starts):
            with torch.cuda.stream(side[n % len(side)]):      # draws and scratch are made on the chunk's own stream
                part = {k: v[i:i + chunk] for k, v in full.items()}
                self._launch(rays[i:i + chunk], coarse_model, fine_model, part, retraw, False, False)
        for s in side:
            cur.wait_stream(s)
        for v in full.values():
            v.record_stream(side[0]); v.record_stream(side[1])
        return full

    def render(self, H, W, K, coarse_model, fine_model, chunk=1024 * 32, rays=None, retraw=True,
               c2w=None, c2w_staticcam=None):
        """Render a full image from ``c2w`` or an explicit ray set (render_utils.py:176-238).
        Returns [rgb_map, disp_map, acc_map, extras]."""
        coarse_model, fine_model = nerf_mod.adopt(coarse_model), nerf_mod.adopt(fine_model)
        if c2w is not None:
            dev = next(coarse_model.parameters()).device
            if not dev.type == 'cuda':
                raise _lib.NerfAmdError("models are on %s; nerf_shared_amd runs on ROCm devices only" % dev)
            batch = utils.make_ray_batch(H, W, K, c2w, self.near, self.far, self.use_viewdirs, self.ndc,
                                         c2w_staticcam=c2w_staticcam, device=dev)
            sh = (H, W, 3)
        else:
            rays_o, rays_d = rays
            _lib.require_device(rays_d, "rays")
            sh = rays_d.shape
            tracked = torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)
            if (not tracked and c2w_staticcam is None and rays_o.dtype == torch.float32 and rays_d.dtype == torch.float32
                    and rays_o.is_cuda and rays_o.shape == rays_d.shape and rays_d.numel() > 0):
                # no gradient flows into the rays: the ten small launches below as one (nerf_amd_assemble_rays)
                src = rays_d.detach().reshape(-1, 3).contiguous() if self.use_viewdirs else None
                o, d = rays_o.detach(), rays_d.detach()
                if self.ndc:
                    o, d = utils.ndc_rays(H, W, K[0][0], 1., o, d)
                o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
                batch = torch.empty(o.shape[0], 11 if self.use_viewdirs else 8, device=o.device, dtype=torch.float32)
                with torch.cuda.device(o.device):
                    _lib.check(lib.nerf_amd_assemble_rays(o.data_ptr(), d.data_ptr(), _lib.ptr(src), o.shape[0], float(self.near),
                                                          float(self.far), batch.data_ptr(), _lib.stream_of(o.device)),
                               "nerf_amd_assemble_rays")
            else:
                viewdirs = None
                if self.use_viewdirs:
                    viewdirs = rays_d
                    if c2w_staticcam is not None:
                        rays_o, rays_d = utils.get_rays(H, W, K, c2w_staticcam)
                    viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
                    viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
                sh = rays_d.shape
                if self.ndc:
                    rays_o, rays_d = utils.ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
                rays_o = torch.reshape(rays_o, [-1, 3]).float()
                rays_d = torch.reshape(rays_d, [-1, 3]).float()
                near, far = self.near * torch.ones_like(rays_d[..., :1]), self.far * torch.ones_like(rays_d[..., :1])
                batch = torch.cat([rays_o, rays_d, near, far], -1)
                if self.use_viewdirs:
                    batch = torch.cat([batch, viewdirs], -1)

        all_ret = self.render_batch(coarse_model, fine_model, batch, chunk, retraw)
        for k in all_ret:
            k_sh = list(sh[:-1]) + list(all_ret[k].shape[1:])
            all_ret[k] = torch.reshape(all_ret[k], k_sh)

        ret_list = [all_ret[k] for k in _KEYS_MAIN]
        ret_dict = {k: all_ret[k] for k in all_ret if k not in _KEYS_MAIN}
        return ret_list + [ret_dict]

    def raw2outputs(self, raw, z_vals, rays_d, pytest=False):
        """raw [R,S,>=4], z_vals [R,S], rays_d [R,3] -> rgb_map, disp_map, acc_map,
        weights, depth_map (render_utils.py:241-290).  Differentiable with respect to
        ``raw`` (HIP backward kernel); z_vals and rays_d are constants."""
        _lib.require_device(raw, "raw")
        dev = raw.device
        raw_c = raw.detach().contiguous().float()
        if raw_c.dim() != 3 or raw_c.shape[-1] < 4:
            raise _lib.NerfAmdError("raw must be [num_rays, num_samples, >=4]")
        R, S, ch = raw_c.shape
        z = z_vals.detach().contiguous().float()
        d = rays_d.detach().contiguous().float()
        noise = None
        if self.raw_noise_std > 0.:
            noise = (_pytest_uniform([R, S], dev) if pytest else torch.randn([R, S], device=dev)) * self.raw_noise_std
        if torch.is_grad_enabled() and (raw.requires_grad or rays_d.requires_grad):
            out = _Raw2OutputsFn.apply(raw.contiguous().float(), z, rays_d.contiguous().float(), noise, bool(self.white_bkgd))
        else:
            with torch.no_grad():
                out = _Raw2OutputsFn.apply(raw_c, z, d, noise, bool(self.white_bkgd))
        if S == 1:
            # With one sample the reference's `dists` is empty (it expands the 1e10 tail to dists[..., :1].shape = [R, 0],
            # render_utils.py:256-258), and so are alpha and the weights: the kernels return the sums over nothing
            # (background colour, acc 0, disp NaN, zero gradients) and the weights come back as [R, 0].
            out = (out[0], out[1], out[2], out[3][:, :0], out[4])
        return out

    def render_from_batch_poses(self, H, W, K, chunk, batch_c2w, coarse_model, fine_model,
                                retraw, save_directory, b_combine_as_video=False, tb_writer=None, io_workers=4):
        """Render a set of poses and save them as 000.png, 001.png, ... (render_utils.py:293-319).

        The reference copies every float image to the host, quantises it with numpy and writes the
        PNG before it renders the next pose.  Here the image is quantised on the GPU (utils.to8b ->
        nerf_amd_to8b), one quarter of the bytes crosses PCIe into pinned memory on a copy stream,
        and PNG encoding + file writes run on `io_workers` threads while the next pose renders
        (io_workers=0: write synchronously).  Returns the uint8 frames [H, W, 3] (the reference
        returns nothing).  Video output needs imageio's ffmpeg plugin and is skipped without it."""
        from . import image_io
        os.makedirs(save_directory, exist_ok=True)
        keep_float = b_combine_as_video or tb_writer is not None
        rgbs, frames = [], []
        writer = image_io.AsyncImageWriter(io_workers) if io_workers > 0 else None
        copy_stream = None
        try:
            with torch.no_grad():
                for i, c2w in enumerate(batch_c2w):
                    rgb, _, _, _ = self.render_from_pose(H, W, K, chunk=chunk, c2w=c2w,
                                                         coarse_model=coarse_model, fine_model=fine_model)
                    rgb8 = utils.to8b(rgb)
                    filename = os.path.join(save_directory, '{:03d}.png'.format(i))
                    if keep_float:
                        rgbs.append(rgb.cpu().numpy())
                    if writer is None:
                        frames.append(rgb8.cpu().numpy())
                        image_io.write_png(filename, frames[-1])
                        continue
                    dev = rgb8.device
                    if copy_stream is None:
                        copy_stream = torch.cuda.Stream(dev)
                    host = torch.empty(rgb8.shape, dtype=torch.uint8, pin_memory=True)
                    copy_stream.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(copy_stream):
                        host.copy_(rgb8, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record(copy_stream)
                    rgb8.record_stream(copy_stream)
                    frames.append(host.numpy())
                    writer.submit(filename, frames[-1], before=done.synchronize)
        finally:
            if writer is not None:
                writer.close()
        if b_combine_as_video:
            try:
                import imageio
                imageio.mimwrite(os.path.join(save_directory, 'video.mp4'), utils.to8b(np.stack(rgbs)), fps=30, quality=8)
            except ImportError:
                try:                         # no imageio/ffmpeg here: an animated GIF from the uint8 frames instead
                    from PIL import Image
                    imgs = [Image.fromarray(f) for f in frames]
                    imgs[0].save(os.path.join(save_directory, 'video.gif'), save_all=True, append_images=imgs[1:],
                                 duration=33, loop=0)
                    print("render_from_batch_poses: imageio is not installed, wrote video.gif instead of video.mp4")
                except ImportError:
                    print("render_from_batch_poses: neither imageio nor Pillow is installed, video skipped (frames are on disk)")
        if tb_writer is not None:
            tb_writer.add_images('Test/Images', torch.tensor(utils.to8b(np.stack(rgbs))), dataformats="NHWC")
        return frames
