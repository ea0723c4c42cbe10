"""Drop-in for nerf_shared/nerf.py: Embedder, get_embedder and the NeRF field,
evaluated by the HIP kernels in csrc/ (fused bf16 MFMA, or exact fp32 MFMA).

Signatures, attribute names and state_dict keys follow the reference
(/root/reference/nerf_shared/nerf.py:11-143) so a reference checkpoint
(`coarse_model_state_dict` / `fine_model_state_dict`, utils.py:450-455) loads
with ``load_state_dict`` unchanged.

Inference (torch.no_grad(), or nothing requires grad) runs the forward-only kernels.  With
gradients enabled, the standard model (D=8, W=256, skips=[4]; viewdirs with multires 10/4 or 15/6, or no
view branch with multires 10 / 15) runs the training kernels -- in bf16, or in split precision (fp32-class
gradients) when the model's precision is 'fp32_split' or 'fp32' -- and loss.backward() reaches its parameters,
the points / view directions and the rays.  Any other architecture renders on the
forward-only kernels and raises NerfAmdError from backward(): outputs never silently come back without
autograd history.
"""
import ctypes
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import lib

# "fp32_split": fp32-class results on the 16-bit matrix pipe (fp16 operand pairs, three MFMAs per product; csrc/mlp_split.hip)
_PRECISIONS = {"fp32": _lib.PREC_FP32, "bf16": _lib.PREC_BF16, "fp32_split": _lib.PREC_FP32_SPLIT}
_default_precision = "bf16"


def set_default_precision(name):
    """'bf16' (fused kernel, bf16 operands / fp32 accumulate; falls back to fp32 for
    architectures it does not cover), 'fp32_split' (fp32-class results from fp16 operand pairs on the 16-bit matrix pipe,
    forward AND backward; needs every encoded input and hidden activation below 65504 in magnitude -- beyond that the fp16
    hi part overflows to inf, which then spreads through the output: use 'fp32' for such models) or 'fp32' (exact-fp32
    MFMA parity mode; trains on the split-precision kernels, NeRF._train_precision)."""
    global _default_precision
    if name not in _PRECISIONS:
        raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
    _default_precision = name


def get_default_precision():
    return _default_precision


# ------------------------------------------------------------------ embedding
class Embedder:
    """Positional encoding (nerf.py:11-41).  Same kwargs as the reference; the
    HIP kernel covers the configuration get_embedder builds (3 inputs,
    include_input, log sampling, [sin, cos])."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        kw = self.kwargs
        d = kw['input_dims']
        if d != 3 or not kw['include_input'] or not kw['log_sampling'] \
                or kw['max_freq_log2'] != kw['num_freqs'] - 1 \
                or list(kw['periodic_fns']) != [torch.sin, torch.cos]:
            raise NotImplementedError("the HIP embedder implements get_embedder's configuration "
                                      "(3-d input, include_input, log sampling, [sin, cos])")
        self.num_freqs = int(kw['num_freqs'])
        self.out_dim = d + 2 * d * self.num_freqs

    def embed(self, inputs):
        _lib.require_device(inputs, "inputs")
        x = inputs.detach().reshape(-1, 3).contiguous().float()
        out = torch.empty(x.shape[0], self.out_dim, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(lib.nerf_amd_embed(x.data_ptr(), x.shape[0], self.num_freqs, out.data_ptr(),
                                          _lib.stream_of(x.device)), "nerf_amd_embed")
        return out.reshape(list(inputs.shape[:-1]) + [self.out_dim])


def get_embedder(multires, i=0):
    """(embed_fn, out_dim), nerf.py:43-58."""
    if i == -1:
        return nn.Identity(), 3
    embed_kwargs = {
        'include_input': True,
        'input_dims': 3,
        'max_freq_log2': multires - 1,
        'num_freqs': multires,
        'log_sampling': True,
        'periodic_fns': [torch.sin, torch.cos],
    }
    embedder_obj = Embedder(**embed_kwargs)
    embed = lambda x, eo=embedder_obj: eo.embed(x)   # noqa: E731  (same closure shape as the reference)
    return embed, embedder_obj.out_dim


# ------------------------------------------------------------------ the field
def adopt(model):
    """The NeRF of this package that evaluates `model`: `model` itself when it is one, else -- for a model built by the
    reference's own class (nerf_shared.nerf.NeRF, nerf.py:61-94, or anything with its attributes and nn.Linear layout) -- a
    twin that SHARES the model's Parameter objects (nothing is copied: optimizer steps, load_state_dict and .to() on the
    reference model are seen by the twin), made once and kept on the model.  This is what lets a caller keep
    `from nerf_shared import nerf` for its models and swap only the renderer."""
    if model is None or isinstance(model, NeRF):
        return model
    twin = getattr(model, "__dict__", {}).get("_nerf_amd_twin")
    if twin is not None:
        return twin
    try:
        D, W, skips, vd = int(model.D), int(model.W), list(model.skips), bool(model.use_viewdirs)
        ic, icv = int(model.input_ch), int(model.input_ch_views)
        pts, views = model.pts_linears, model.views_linears
    except AttributeError as e:
        raise TypeError("expected a nerf_shared_amd.nerf.NeRF or a model with the reference NeRF's attributes (D, W, skips, "
                        "use_viewdirs, input_ch, input_ch_views, pts_linears, views_linears, ...); got %s: %s"
                        % (type(model).__name__, e))
    i_embed = -1 if ic == 3 else 0
    L, Lv = (ic - 3) // 6, ((icv - 3) // 6 if vd else 4)
    if (i_embed == 0 and 3 + 6 * L != ic) or (vd and i_embed == 0 and 3 + 6 * Lv != icv) or (vd and i_embed == -1 and icv != 3):
        raise TypeError("cannot tell the positional encodings of %s from input_ch=%d / input_ch_views=%d"
                        % (type(model).__name__, ic, icv))
    out_ch = 4 if vd else int(model.output_linear.out_features)
    twin = NeRF(D=D, W=W, output_ch=out_ch, skips=skips, use_viewdirs=vd, multires=max(L, 0), multires_views=max(Lv, 0),
                i_embed=i_embed)
    pairs = list(zip(twin.pts_linears, pts)) + [(twin.views_linears[0], views[0])]
    if vd:
        pairs += [(twin.feature_linear, model.feature_linear), (twin.alpha_linear, model.alpha_linear),
                  (twin.rgb_linear, model.rgb_linear)]
    else:
        pairs += [(twin.output_linear, model.output_linear)]
    for mine, theirs in pairs:
        if tuple(mine.weight.shape) != tuple(theirs.weight.shape) or tuple(mine.bias.shape) != tuple(theirs.bias.shape):
            raise TypeError("layer shapes of %s do not follow the reference NeRF (%s vs %s)"
                            % (type(model).__name__, tuple(theirs.weight.shape), tuple(mine.weight.shape)))
        mine.weight, mine.bias = theirs.weight, theirs.bias          # the same Parameter objects
    twin.precision = getattr(model, "precision", None)
    model.__dict__["_nerf_amd_twin"] = twin                           # (not a registered submodule: state_dict keys stay the reference's)
    return twin


def _destroy_handle(handle):
    if handle:
        lib.nerf_amd_model_destroy(handle)


class _DeferredGradFn(torch.autograd.Function):
    """Identity that ties a forward-only result to a tensor that requires grad, with a backward that raises: what a
    model outside the training kernels returns when gradients were requested (NeRF._grad_request).  The reference would
    return a differentiable result here; this path renders it the same way and fails loudly -- with the reason -- the
    moment someone calls backward() through it, instead of refusing the forward (plain inference written without
    torch.no_grad() keeps working) or silently returning outputs with no history (main.py:103 would train nothing)."""

    @staticmethod
    def forward(ctx, x, anchor, msg):
        ctx.msg = msg
        return x.clone()          # not a view of the input: callers may write into the result in place (rgb.clamp_(), ...)

    @staticmethod
    def backward(ctx, g):
        raise _lib.NerfAmdError(ctx.msg)


def attach_deferred_grad(out, deferred):
    """out: tensor or dict of tensors; deferred: (anchor tensor, message) from NeRF._grad_request, or None."""
    if deferred is None:
        return out
    anchor, msg = deferred
    if isinstance(out, dict):
        return {k: _DeferredGradFn.apply(v, anchor, msg) for k, v in out.items()}
    return _DeferredGradFn.apply(out, anchor, msg)


class _FieldTrainFn(torch.autograd.Function):
    """The fused field with HIP backward kernels, in bf16 or in split precision (`prec`:
    _lib.PREC_BF16 / PREC_FP32_SPLIT): forward saves
    every layer's activations in a workspace tensor; backward runs the dX-chain kernel and the
    weight-gradient kernels (nerf_amd_field_backward) and returns gradients for the parameters and,
    when asked, for the points / view directions (explicit-points mode) or the ray batch (rays mode;
    z_vals are constants: the coarse depths depend on near/far only and the fine ones are detached)."""

    @staticmethod
    def forward(ctx, model, prec, pts, viewdirs, rays, z_vals, n_rays, n_samples, *params):
        dev = params[0].device
        handle = model._model_handle(dev, _lib.TRAIN_COPIES[prec])      # the forward's copy and the backward's, one launch
        P = n_rays * n_samples
        nbytes = lib.nerf_amd_train_workspace(handle, P, prec)
        if nbytes < 0:
            raise _lib.NerfAmdError("this architecture has no training kernels")
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        raw = torch.empty(P, 4 if model.use_viewdirs else model.output_ch, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_field_forward_train(handle, _lib.ptr(pts), _lib.ptr(viewdirs), _lib.ptr(rays),
                                                        rays.shape[1] if rays is not None else 0, _lib.ptr(z_vals),
                                                        n_rays, n_samples, raw.data_ptr(), ws.data_ptr(), ws.numel(), prec,
                                                        _lib.stream_of(dev)), "nerf_amd_field_forward_train")
        ctx.model, ctx.ws, ctx.R, ctx.S, ctx.prec = model, ws, n_rays, n_samples, prec
        ctx.inputs = (pts, viewdirs, rays, z_vals)
        ctx.pack_key = model._packed_key          # the weights this forward ran on (backward must see the same pack)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        model, ws, R, S = ctx.model, ctx.ws, ctx.R, ctx.S
        pts, viewdirs, rays, z_vals = ctx.inputs
        if ws is None:
            raise _lib.NerfAmdError("backward through this field evaluation ran already and released its saved "
                                    "activations; a second backward (retain_graph=True) is not supported")
        if model._packed_key is None or model._packed_key != ctx.pack_key:
            raise _lib.NerfAmdError("the model's parameters changed between this forward and its backward "
                                    "(optimizer step, load_state_dict or weights_changed()): the gradients would be "
                                    "taken against the wrong weights; run backward before updating the parameters")
        dev = ws.device
        g = g_raw.contiguous().float()
        # one zeroed buffer for every gradient (head and ray gradients accumulate with atomics), views per tensor
        sizes, shapes = model._grad_layout()
        n = len(shapes)
        flat = torch.zeros(sizes[-1], device=dev, dtype=torch.float32)
        base = flat.data_ptr()
        wp = (ctypes.c_void_p * n)(*[base + 4 * o for o in sizes[0:n]])
        bp = (ctypes.c_void_p * n)(*[base + 4 * o for o in sizes[n:2 * n]])
        need_pts, need_vd, need_rays = ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        g_pts = torch.empty(R * S, 3, device=dev, dtype=torch.float32) if (pts is not None and need_pts) else None
        g_rays6 = torch.zeros(R, 6, device=dev, dtype=torch.float32) if (rays is not None and need_rays) else None
        g_vd = torch.zeros(R, 3, device=dev, dtype=torch.float32) if (model.use_viewdirs and ((pts is not None and need_vd) or g_rays6 is not None)) else None
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_field_backward(model._handle, g.data_ptr(), _lib.ptr(pts), _lib.ptr(viewdirs), _lib.ptr(rays),
                                                   rays.shape[1] if rays is not None else 0, _lib.ptr(z_vals), R, S,
                                                   ws.data_ptr(), ws.numel(), wp, bp, n, _lib.ptr(g_pts), _lib.ptr(g_rays6),
                                                   _lib.ptr(g_vd), ctx.prec, _lib.stream_of(dev)), "nerf_amd_field_backward")
        ctx.ws = None
        g_rays = None
        if g_rays6 is not None:           # [R, 11] = d/d(o, d, near, far, viewdir); [R, 8] without view branch
            g_rays = torch.cat([g_rays6, torch.zeros(R, 2, device=dev)] + ([g_vd] if g_vd is not None else []), -1)
        grads = []                        # parameter order of _train_params(): weight, bias per linear
        parts = flat.split([sh[0] * sh[1] for sh in shapes] + [sh[0] for sh in shapes])
        for i, shape in enumerate(shapes):
            grads.append(parts[i].view(shape))
            grads.append(parts[n + i])
        return (None, None, g_pts, g_vd if (pts is not None and need_vd and g_vd is not None) else None, g_rays, None, None, None) + tuple(grads)


# The packed device copy of a model's weights is refreshed when a parameter's (data_ptr, _version)
# changes.  Optimizers that update through fused multi-tensor kernels (torch.optim.Adam(fused=True))
# do not bump `_version`, so every optimizer step also marks the models it updated stale.  Writes through
# `param.data` are invisible to both: call `model.weights_changed()` after them.
_LIVE_MODELS = weakref.WeakSet()


def _after_optimizer_step(optimizer, args, kwargs):
    stepped = {id(p) for g in optimizer.param_groups for p in g["params"]}
    for m in list(_LIVE_MODELS):
        if any(id(p) in stepped for p in m.parameters()):
            m._packed_key = None


from torch.optim import optimizer as _torch_optimizer  # noqa: E402
_torch_optimizer.register_optimizer_step_post_hook(_after_optimizer_step)


class NeRF(nn.Module):
    """The reference's field model (nerf.py:61-143), same constructor, same
    parameters; ``forward`` runs on the MI355X kernels."""

    def __init__(self, D=8, W=256, output_ch=4, skips=[4], use_viewdirs=False, multires=10,
                 multires_views=4, i_embed=0):
        super(NeRF, self).__init__()
        self.D = D
        self.W = W
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.output_ch = output_ch
        self.multires, self.multires_views, self.i_embed = multires, multires_views, i_embed

        self.embed_fn, self.input_ch = get_embedder(multires, i_embed)
        self.input_ch_views = 0
        self.embeddirs_fn = None
        if use_viewdirs:
            self.embeddirs_fn, self.input_ch_views = get_embedder(multires_views, i_embed)

        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] +
            [nn.Linear(W + self.input_ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

        self.precision = None          # None: follow set_default_precision()
        self._handle = None
        self._handle_device = None
        self._packed_key = None
        self._finalizer = None
        _LIVE_MODELS.add(self)

    def __getstate__(self):
        # copies / pickles never share the library handle: the copy creates its own on first use
        state = self.__dict__.copy()
        state.update(_handle=None, _handle_device=None, _packed_key=None, _finalizer=None)
        for k in ('_mods_cache', '_grad_layout_cache', '_trainable_kernels'):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        _LIVE_MODELS.add(self)

    def weights_changed(self):
        """Force a re-pack of the parameters on the next call (needed only after writing through
        `param.data` or other paths that bypass autograd's version counters)."""
        self._packed_key = None

    # -- device-side packed copy of the parameters --------------------------------
    def _linears(self):
        """nn.Linear modules in the C ABI's tensor order (include/nerf_amd.h)."""
        mods = self.__dict__.get('_mods_cache')
        if mods is None:
            mods = list(self.pts_linears)
            if self.use_viewdirs:
                mods += [self.feature_linear, self.alpha_linear, self.views_linears[0], self.rgb_linear]
            else:
                mods += [self.output_linear]
            self.__dict__['_mods_cache'] = mods
        return mods

    def _grad_layout(self):
        """Offsets (in floats) of every weight, then every bias, in one flat gradient buffer (+ the total), and the
        weight shapes -- constants of the architecture."""
        lay = self.__dict__.get('_grad_layout_cache')
        if lay is None:
            mods = self._linears()
            shapes = [tuple(m.weight.shape) for m in mods]
            offs, o = [], 0
            for sh in shapes:
                offs.append(o); o += sh[0] * sh[1]
            for sh in shapes:
                offs.append(o); o += sh[0]
            lay = (offs + [o], shapes)
            self.__dict__['_grad_layout_cache'] = lay
        return lay

    def _ensure_handle(self, device):
        if self._handle is None or self._handle_device != device:
            if self._finalizer is not None:
                self._finalizer()
            arch = _lib.make_arch(self.D, self.W, self.output_ch, self.skips, self.use_viewdirs,
                                  self.multires, self.multires_views, self.i_embed)
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.nerf_amd_model_create(ctypes.byref(arch), device.index or 0, ctypes.byref(h)),
                           "nerf_amd_model_create")
            self._handle, self._handle_device, self._packed_key = h, device, None
            self._finalizer = weakref.finalize(self, _destroy_handle, h)
            self.__dict__['_trainable_kernels'] = bool(lib.nerf_amd_model_supports_training(h, _lib.PREC_BF16))     # the fused 8x256 family
            self.__dict__['_trainable_f32'] = bool(lib.nerf_amd_model_supports_training(h, _lib.PREC_FP32))       # any architecture, exact fp32
        return self._handle

    def _model_handle(self, device, copies=None):
        """Create the library handle on first use and re-pack when any parameter changed -- the packed copies the caller is
        about to use (_lib.COPY_*; None: the one this model's own precision renders with).  A copy that was not asked for
        since the parameters last changed is packed when someone first asks for it."""
        self._ensure_handle(device)
        if copies is None:
            copies = _lib.COPY_OF[self._precision_code()]
        # (data_ptr, _version) of every parameter: unchanged = same storage, same contents, hence same device and dtype
        params = [m._parameters[k] for k in ('weight', 'bias') for m in self._linears()]
        key = tuple([t.data_ptr() for t in params] + [t._version for t in params])
        same = key == self._packed_key
        have = self.__dict__.get('_packed_copies', 0) if same else 0
        missing = copies & ~have
        if missing:
            for t in params:
                if t.device != device or t.dtype != torch.float32:
                    raise _lib.NerfAmdError("NeRF parameters must be fp32 on %s (found %s on %s); call model.to(device)"
                                            % (device, t.dtype, t.device))
            n = len(params) // 2
            live = [t.detach().contiguous() for t in params]       # parameters are contiguous unless someone made them not
            wp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in live[:n]])
            bp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in live[n:]])
            with torch.cuda.device(device):
                _lib.check(lib.nerf_amd_model_update_copies(self._handle, wp, bp, n, missing, 1 if same else 0, _lib.stream_of(device)),
                           "nerf_amd_model_update_copies")
            self._packed_key = key
            self.__dict__['_packed_copies'] = have | missing
        return self._handle

    def _train_params(self):
        out = []
        for m in self._linears():
            p = m._parameters
            out += [p['weight'], p['bias']]
        return out

    def _grad_request(self, device, *inputs):
        """("none", None): no gradient is requested (grad mode off, or nothing requires grad) -> forward-only kernels.
        ("train", None): gradients are requested and the training kernels cover this model -> autograd Functions
        with HIP backward kernels, in the arithmetic of _train_precision().
        ("defer", (anchor, message)): gradients are requested but the training kernels do not cover this model or precision:
        the call runs on the forward-only kernels and its results are tied to `anchor` by a backward that raises `message`
        (attach_deferred_grad) -- outputs never silently lose their autograd history (the reference's loss.backward(),
        main.py:103, would train nothing), and inference written without torch.no_grad() still works as in the reference."""
        if not torch.is_grad_enabled():
            return "none", None
        anchor = next((p for p in self._train_params() if p.requires_grad), None)
        if anchor is None:
            anchor = next((t for t in inputs if t is not None and t.requires_grad), None)
        if anchor is None:
            return "none", None
        prec = self.precision or _default_precision
        self._ensure_handle(device)
        if self.__dict__['_trainable_kernels']:
            return "train", None
        if self.__dict__['_trainable_f32']:
            return "train", None           # any other architecture: the exact-fp32 training path (csrc/train_f32.hip)
        msg = ("backward() reached a result of the forward-only kernels: this model is %s(D=%d, W=%d, skips=%s, use_viewdirs=%s, "
               "multires=%d, multires_views=%d, output_ch=%d) in precision '%s', and neither the fused training kernels (NeRF(D=8, "
               "W=256, skips=[4]), multires 10/4, 15/6, or 10 / 15 without view branch) nor the exact-fp32 training path (its "
               "feature rows must fit the CU's LDS) cover it.  Gradients were requested when it was evaluated (grad mode on, a "
               "parameter or input requiring grad), so the result was given this backward instead of none; there is no PyTorch fallback"
               % (type(self).__name__, self.D, self.W, list(self.skips), self.use_viewdirs, self.multires,
                  self.multires_views, self.output_ch, prec))
        return "defer", (anchor, msg)

    def _train_precision(self):
        """Arithmetic of the training kernels for this model's precision: 'bf16' trains in bf16; 'fp32_split' AND 'fp32'
        train on the split-precision kernels (fp16 operand pairs, three MFMAs per product -- forward, dX chain and weight
        gradients; gradients agree with fp32 autograd to ~1e-6): an 'fp32' model of the fused family has the split-precision
        training forward (1e-5 from its exact inference forward on |raw| <= 20).  Every other architecture trains on the
        exact-fp32 path (csrc/train_f32.hip: fp32 MFMA rate)."""
        name = self.precision or _default_precision
        if name not in _PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
        if not self.__dict__.get('_trainable_kernels', True):
            return _lib.PREC_FP32          # not the fused family: the exact-fp32 training path, whatever the inference precision
        return _lib.PREC_BF16 if name == "bf16" else _lib.PREC_FP32_SPLIT

    def _wants_grad(self, device, *inputs):
        """True when the training kernels will run for this call (see _grad_request)."""
        return self._grad_request(device, *inputs)[0] == "train"

    def forward_rays(self, rays, z_vals):
        """raw [R, S, 4] of rays [R, 11] at depths z_vals [R, S] (pts = o + d z formed in the kernel);
        differentiable with respect to the parameters and the ray batch."""
        R, S = z_vals.shape
        raw = _FieldTrainFn.apply(self, self._train_precision(), None, None, rays, z_vals, R, S, *self._train_params())
        return raw.reshape(R, S, 4 if self.use_viewdirs else self.output_ch)

    def _precision_code(self):
        name = self.precision or _default_precision
        if name not in _PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
        code = _PRECISIONS[name]
        if code == _lib.PREC_BF16 and self._handle is not None and not lib.nerf_amd_model_supports_bf16(self._handle):
            code = _lib.PREC_FP32      # still a HIP MFMA kernel, at the fp32 rate
        if code == _lib.PREC_FP32_SPLIT and self._handle is not None and not lib.nerf_amd_model_supports_split(self._handle):
            code = _lib.PREC_FP32      # the exact kernel covers every architecture
        return code

    def supports_bf16(self, device=None):
        dev = device or next(self.parameters()).device
        return bool(lib.nerf_amd_model_supports_bf16(self._ensure_handle(torch.device(dev))))

    # -- reference API ------------------------------------------------------------
    def forward(self, inputs, viewdirs, netchunk=1024 * 64):
        """inputs [..., S, 3], viewdirs [R, 3] or None -> [..., S, 4 | output_ch]
        (nerf.py:96-108).  ``netchunk`` is accepted for compatibility; the kernel
        tiles points itself, so chunking cannot change the result."""
        _lib.require_device(inputs, "inputs")
        dev = inputs.device
        mode, deferred = self._grad_request(dev, inputs, viewdirs)
        want = mode == "train"
        pts = (inputs if want else inputs.detach()).reshape(-1, 3).contiguous().float()
        n_samples = inputs.shape[-2] if inputs.dim() >= 2 else 1
        vd = None
        if viewdirs is not None:
            if not self.use_viewdirs:
                raise _lib.NerfAmdError("viewdirs given to a NeRF built with use_viewdirs=False")
            if inputs.dim() != 3 or viewdirs.shape[0] != inputs.shape[0]:
                raise _lib.NerfAmdError("with viewdirs, inputs must be [R, S, 3] and viewdirs [R, 3]")
            vd = (viewdirs if want else viewdirs.detach()).reshape(-1, 3).contiguous().float()
        elif self.use_viewdirs:
            raise _lib.NerfAmdError("this NeRF was built with use_viewdirs=True: viewdirs is required")
        out_ch = 4 if self.use_viewdirs else self.output_ch
        n_rays = pts.shape[0] // n_samples
        if want:
            raw = _FieldTrainFn.apply(self, self._train_precision(), pts, vd, None, None, n_rays, n_samples, *self._train_params())
            return raw.reshape(list(inputs.shape[:-1]) + [out_ch])
        handle = self._model_handle(dev)
        out = torch.empty(pts.shape[0], out_ch, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_nerf_forward(handle, pts.data_ptr(), _lib.ptr(vd), n_rays, n_samples,
                                                 out.data_ptr(), self._precision_code(), _lib.stream_of(dev)),
                       "nerf_amd_nerf_forward")
        return attach_deferred_grad(out.reshape(list(inputs.shape[:-1]) + [out_ch]), deferred)

    def MLP(self, x):
        """Already-embedded rows [P, input_ch + input_ch_views] -> [P, 4|output_ch]
        (nerf.py:110-134).  Runs on the exact-fp32 kernel (the fused bf16 kernel
        generates the encoding itself and is reached through forward())."""
        _lib.require_device(x, "x")
        dev = x.device
        width = self.input_ch + self.input_ch_views
        xe = x.detach().reshape(-1, x.shape[-1]).contiguous().float()
        if xe.shape[-1] != width:
            raise _lib.NerfAmdError("MLP expects %d embedded columns, got %d" % (width, xe.shape[-1]))
        handle = self._model_handle(dev, _lib.COPY_FP32)
        out_ch = 4 if self.use_viewdirs else self.output_ch
        out = torch.empty(xe.shape[0], out_ch, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_mlp_embedded(handle, xe.data_ptr(), xe.shape[0], out.data_ptr(),
                                                 _lib.stream_of(dev)), "nerf_amd_mlp_embedded")
        return out.reshape(list(x.shape[:-1]) + [out_ch])

    def get_density(self, points, chunk=1024 * 64):
        """Raw sigma with an all-ones view direction (nerf.py:136-143)."""
        view_dir = torch.ones_like(points[..., 0, :]) if self.use_viewdirs else None
        output = self.forward(points, view_dir, chunk)
        return output[..., -1]

    def load_weights_from_keras(self, weights):
        """The weight list of a Keras NeRF (kernel [in, out], bias per layer, in the order pts_linears, feature,
        views, rgb, alpha) into this model (nerf.py:146-173): kernels transposed to nn.Linear's [out, in].  Like the
        reference it covers the view-branch model only and replaces `.data`; the packed device copy is refreshed on
        the next call."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        targets = [self.pts_linears[i] for i in range(self.D)] + [self.feature_linear, self.views_linears[0],
                                                                   self.rgb_linear, self.alpha_linear]
        for j, lin in enumerate(targets):
            w = torch.from_numpy(np.transpose(weights[2 * j]))
            b = torch.from_numpy(np.transpose(weights[2 * j + 1]))
            lin.weight.data = w.to(lin.weight.device)
            lin.bias.data = b.to(lin.bias.device)
        self.__dict__.pop('_mods_cache', None)
        self.weights_changed()
