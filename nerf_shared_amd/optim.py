"""Optimizer of the reference's training loop on the MI355X library.

The reference builds `torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))` over the
parameters of both fields (utils.py:163-172) and calls `optimizer.step()` once per batch (main.py:104).
`Adam` here IS a torch.optim.Adam -- same constructor, same `param_groups`, same per-parameter state
(`step`, `exp_avg`, `exp_avg_sq`), so `state_dict()` / `load_state_dict()` and the reference's `.tar`
checkpoints (utils.py:174-214, 444-456) go both ways -- whose `step()` updates every parameter tensor of a
group with ONE kernel launch (nerf_amd_adam_step) instead of torch's multi-tensor passes.  At the reference's
batch size the training step is host-bound (48 small tensors), which is where the time goes.

Anything the kernel does not cover (amsgrad, maximize, capturable, differentiable, tensor lr, sparse / non-fp32 /
non-contiguous / CPU tensors) raises: there is no second implementation behind this class -- construct a plain
torch.optim.Adam for those.
"""
import ctypes

import torch

from . import _lib
from ._lib import lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise _lib.NerfAmdError("nerf_shared_amd.optim.Adam has no amsgrad variant; use torch.optim.Adam")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                         foreach=False, fused=False)
        # Per group, while every parameter of the group steps together (the training loop's case): the common step
        # count and the pointer tables of exp_avg / exp_avg_sq.  The per-parameter `step` tensors of the state (CPU
        # scalars, as in torch.optim.Adam) are brought up to date by state_dict() / copies / pickles -- 48 scalar tensor
        # increments per step would cost more host time than the launch; code that reads optimizer.state[p]["step"]
        # between steps calls optimizer.state_dict() first (or _sync_steps()).  Replacing a state tensor by hand
        # (state[p]["exp_avg"] = ...) needs load_state_dict() or add_param_group() to drop the cached tables; a parameter
        # that moved is noticed by its data_ptr.
        self._together = {}
        self._device_scalars = None        # enable_device_scalars(): {group index: (step int64[1], lr float64[1], scratch float32[2])}

    # -- captured steps ------------------------------------------------------------
    def enable_device_scalars(self):
        """Keep the step count and the learning rate of every group in device memory and let the update read them there
        (nerf_amd_adam_step_device), so that step() can be captured in a HIP graph and replayed: a replay re-runs the
        kernels with the arguments they were captured with, and the bias corrections change every step.  Call it after at
        least one ordinary step (the per-group fast path must be established); from then on step() advances the device
        count, and whoever replays a captured step calls note_replayed_step() per replay and sync_lr() after changing
        param_group['lr'] (utils.CapturedTrainStep does both)."""
        if not self._together or len(self._together) != len(self.param_groups):
            raise _lib.NerfAmdError("enable_device_scalars(): take one ordinary optimizer step first (every group's parameters "
                                    "must have gradients and step together)")
        self._device_scalars = {}
        for gi, c in self._together.items():
            dev = c["params"][0].device
            self._device_scalars[gi] = (torch.full((1,), int(c["step"]), dtype=torch.int64, device=dev),
                                        torch.full((1,), float(self.param_groups[gi]["lr"]), dtype=torch.float64, device=dev),
                                        torch.zeros(2, dtype=torch.float32, device=dev))

    def sync_lr(self):
        """param_group['lr'] (a Python float, main.py:109-112) -> the device copy a captured step reads."""
        for gi, (_, lr_dev, _) in (self._device_scalars or {}).items():
            lr_dev.fill_(float(self.param_groups[gi]["lr"]))

    def note_replayed_step(self, n=1):
        """A captured step() was replayed n times: the host-side counts (state_dict(), checkpoints) follow the device's."""
        for c in self._together.values():
            c["step"] += n

    # -- torch.optim.Optimizer surface -------------------------------------------
    def __getstate__(self):
        self._sync_steps()                 # copies and pickles carry the state tensors only (torch's own __getstate__)
        return super().__getstate__()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._together = {}
        self._device_scalars = None

    def state_dict(self):
        self._sync_steps()
        sd = super().state_dict()
        for g in sd["param_groups"]:       # a checkpoint loaded into the reference's torch.optim.Adam must not pin its slow
            g["foreach"] = g["fused"] = None   # single-tensor loop: leave the implementation choice to the loading side
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._together = {}
        self._device_scalars = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._together = {}
        self._device_scalars = None

    def _sync_steps(self, gi=None):
        for k, c in list(self._together.items()):
            if gi is not None and k != gi:
                continue
            for p in c["params"]:
                self.state[p]["step"].fill_(float(c["step"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
                raise _lib.NerfAmdError("nerf_shared_amd.optim.Adam covers plain Adam only (amsgrad / maximize / capturable / "
                                        "differentiable are not provided); use torch.optim.Adam")
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                raise _lib.NerfAmdError("nerf_shared_amd.optim.Adam takes a Python float lr (the reference sets "
                                        "param_group['lr'] to a float, main.py:109-112)")
            params = group["params"]
            grads = [p.grad for p in params]
            mask = tuple(g is not None for g in grads)
            c = self._together.get(gi)
            if c is not None and c["mask"] == mask and c["ptrs"] != tuple(p.data_ptr() for p in c["params"]):
                # a parameter moved (model.to(), .half() and back, a replaced tensor): the cached pointer tables are stale
                self._sync_steps(gi)
                del self._together[gi]
                c = None
            if c is not None and c["mask"] == mask:
                # the same parameters as last time got gradients (a model's unused tensors never do -- the reference's
                # NeRF(use_viewdirs=False) keeps an idle views_linears.0): one counter, cached tables, one launch
                c["step"] += 1
                self._launch(c["params"], [g for g in grads if g is not None], c["tables"], c["step"], group,
                             None if self._device_scalars is None else self._device_scalars[gi])
                continue
            # general path: parameters step individually (first step, or the set with gradients changed)
            if self._device_scalars is not None:
                raise _lib.NerfAmdError("the set of parameters with gradients changed after enable_device_scalars(): a captured "
                                        "step covers one fixed set; build a new optimizer / capture")
            if c is not None:
                self._sync_steps(gi)
                del self._together[gi]
            by_step = {}
            for p, g in zip(params, grads):
                if g is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    self._check(p)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)            # torch.optim.Adam's own layout
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append((p, g))
            for step, pg in by_step.items():
                ps, gs = [p for p, _ in pg], [g for _, g in pg]
                tables = self._tables(ps)
                self._launch(ps, gs, tables, step, group)
                if len(by_step) == 1:
                    self._together[gi] = {"mask": mask, "step": step, "tables": tables, "params": ps,
                                          "ptrs": tuple(p.data_ptr() for p in ps)}
        return loss

    @staticmethod
    def _check(p):
        if not p.is_cuda or p.dtype != torch.float32 or p.is_sparse or not p.is_contiguous():
            raise _lib.NerfAmdError("nerf_shared_amd.optim.Adam updates contiguous fp32 parameters on a ROCm device "
                                    "(got %s on %s); use torch.optim.Adam for others" % (p.dtype, p.device))

    def _tables(self, ps):
        n = len(ps)
        arr = ctypes.c_void_p * n
        for p in ps:
            self._check(p)
            st = self.state[p]
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                    raise _lib.NerfAmdError("optimizer state %s must be contiguous fp32 on %s" % (k, p.device))
        return (arr(*[self.state[p]["exp_avg"].data_ptr() for p in ps]),
                arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in ps]),
                (ctypes.c_int64 * n)(*[p.numel() for p in ps]),
                [self.state[p]["exp_avg"] for p in ps] + [self.state[p]["exp_avg_sq"] for p in ps])   # keep-alive

    def _launch(self, ps, grads, tables, step, group, device_scalars=None):
        n = len(ps)
        arr = ctypes.c_void_p * n
        dev = ps[0].device
        gp, keep = [], []
        for p, g in zip(ps, grads):
            if g.dtype != torch.float32 or g.device != dev or g.is_sparse or p.device != dev:
                raise _lib.NerfAmdError("nerf_shared_amd.optim.Adam needs dense fp32 gradients, all on one device")
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)          # alive until the launch is enqueued
            gp.append(g.data_ptr())
        beta1, beta2 = group["betas"]
        if device_scalars is not None:
            step_dev, lr_dev, scratch = device_scalars
            with torch.cuda.device(dev):
                _lib.check(lib.nerf_amd_adam_step_device(n, arr(*[p.data_ptr() for p in ps]), arr(*gp), tables[0], tables[1], tables[2],
                                                         step_dev.data_ptr(), lr_dev.data_ptr(), float(beta1), float(beta2),
                                                         float(group["eps"]), float(group["weight_decay"]), scratch.data_ptr(),
                                                         _lib.stream_of(dev)), "nerf_amd_adam_step_device")
            return
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_amd_adam_step(n, arr(*[p.data_ptr() for p in ps]), arr(*gp), tables[0], tables[1], tables[2],
                                              step, float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                              float(group["weight_decay"]), _lib.stream_of(dev)), "nerf_amd_adam_step")
