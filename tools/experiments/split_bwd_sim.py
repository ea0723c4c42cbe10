"""CPU simulation of the split-precision (fp16 hi/lo pair) backward pass, run BEFORE the kernels were written: which operands
of the dX chain and of the weight-gradient products need pairs for gradients within 1e-3 (relative L2) of fp32 autograd
on the oracle, and where fp32 autograd itself sits against fp64 (the floor any implementation inherits).

    python tools/experiments/split_bwd_sim.py            # tables for the three gradient tests' configurations

Emulation: x -> hi = fp16(x), lo = fp16((x - hi) 2^11) (csrc/mlp_split.hip); a product A.B of pairs is
A_hi B_hi + 2^-11 (A_hi B_lo + A_lo B_hi), matmuls in fp64 then rounded to fp32 (the MFMA's fp32 accumulation order is
not modelled).  Gradients entering the chain are multiplied by a power-of-two loss scale first (fp16 range).
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from nerf_shared_amd import synth  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

VD = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def split(x):
    hi = x.to(torch.float16).to(x.dtype)
    lo = ((x - hi) * 2048.0).to(torch.float16).to(x.dtype)
    return hi, lo


def mm_pairs(a, b, mode):
    """a [m,k] @ b [k,n] with operands as fp16 pairs.  mode: 'full' 3 terms, 'a_pair' (a pair x b hi), 'b_pair', 'hi'."""
    a64, b64 = a.double(), b.double()
    if mode == "full_u":       # unscaled lo parts (fp16 denormals where the residual is below 2^-14), ONE accumulator
        ah, bh = a64.to(torch.float16).double(), b64.to(torch.float16).double()
        al, bl = (a64 - ah).to(torch.float16).double(), (b64 - bh).to(torch.float16).double()
        return (ah @ bh + ah @ bl + al @ bh).float()
    if mode == "full_sym":     # both lo parts scaled by 2^11 (as the chain holds them); the cross terms use hi * 2^-11 rounded
        ah, al = split(a64)    # to fp16 (denormal for |hi| < 2^-3) made in the product kernel, so ONE accumulator serves
        bh, bl = split(b64)
        ahs, bhs = (ah / 2048.0).to(torch.float16).double(), (bh / 2048.0).to(torch.float16).double()
        return (ah @ bh + ahs @ bl + al @ bhs).float()
    ah, al = split(a64)
    bh, bl = split(b64)
    y = ah @ bh
    if mode in ("full", "b_pair"):
        y = y + (ah @ bl) / 2048.0
    if mode in ("full", "a_pair"):
        y = y + (al @ bh) / 2048.0
    return y.float()


class SplitLinear(torch.autograd.Function):
    """y = x W^T + b with the chosen operand precisions in forward, dX and dW."""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        ctx.save_for_backward(x, w)
        ctx.cfg = cfg
        return mm_pairs(x, w.t(), cfg["fwd"]) + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        s = cfg["scale"]
        gs = g * s
        gx = mm_pairs(gs, w, cfg["dx"]) / s
        gw = mm_pairs(gs.t(), x, cfg["dw"]) / s
        gb = gs.sum(0) / s if cfg["dw"] != "hi" else split(gs.double())[0].sum(0).float() / s
        return gx, gw, gb, None


def split_field(cfg):
    def field(sd, pts, vd):
        lin = lambda n, x: SplitLinear.apply(x, sd[n + ".weight"], sd[n + ".bias"], cfg)   # noqa: E731
        e = O.embed(pts.reshape(-1, 3), 10)
        h = e
        for i in range(8):
            h = torch.relu(lin("pts_linears.%d" % i, h))
            if i == 4:
                h = torch.cat([e, h], -1)
        d = O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)
        sigma = lin("alpha_linear", h)
        feat = lin("feature_linear", h)
        hv = torch.relu(lin("views_linears.0", torch.cat([feat, d], -1)))
        return torch.cat([lin("rgb_linear", hv), sigma], -1).reshape(list(pts.shape[:-1]) + [4])
    return field


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))


def sds(seed, sharpen, dtype=torch.float32):
    sd = synth.torch_state_dict(seed, sharpen, **{**VD, "skips": (4,)})
    return {k: v.to(dtype).clone().requires_grad_(True) for k, v in O.state_dict_to_torch(sd).items()}


BASE = dict(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False,
            lindisp=False, near=2.0, far=6.0)


def batch(n, seed):
    rng = np.random.default_rng(seed)
    K = synth.lego_intrinsics(400, 400)
    idx = np.sort(rng.choice(160000, size=n, replace=False))
    ro, rd = synth.rays_np(400, 400, K, synth.LEGO_C2W, idx)
    target = torch.from_numpy(rng.uniform(0, 1, size=(n, 3)).astype(np.float32))
    return torch.from_numpy(synth.ray_batch_np(ro, rd, 2.0, 6.0, True)), target


def training_grads(field, dtype=torch.float32, z_from=None):
    b, target = batch(96, 3)
    cfg = dict(BASE, N_samples=32, N_importance=48)
    c, f = sds(1, 2.0, dtype), sds(11, 2.0, dtype)
    orig = O.nerf_forward
    if field is not None:
        O.nerf_forward = lambda sd, arch, pts, vd, netchunk=0: field(sd, pts, vd)
    try:
        o = O.render_rays(O.RenderCfg(**cfg), b.to(dtype), (c, O.Arch(**VD)), (f, O.Arch(**VD)), retweights=True)
        loss = ((o["rgb_map"] - target.to(dtype)) ** 2).mean() + ((o["rgb0"] - target.to(dtype)) ** 2).mean()
        loss.backward()
    finally:
        O.nerf_forward = orig
    return float(loss), c, f


def ray_grads(field, dtype=torch.float32):
    b, target = batch(80, 7)
    cfg = dict(BASE, N_samples=32, N_importance=48)
    c, f = sds(1, 2.0, dtype), sds(11, 2.0, dtype)
    orig = O.nerf_forward
    if field is not None:
        O.nerf_forward = lambda sd, arch, pts, vd, netchunk=0: field(sd, pts, vd)
    try:
        o = b[:, 0:3].to(dtype).clone().requires_grad_(True)
        d = (b[:, 3:6] * 1.3).to(dtype).clone().requires_grad_(True)
        out = O.render(O.RenderCfg(**cfg), 400, 400, None, ({k: v.detach() for k, v in c.items()}, O.Arch(**VD)),
                       ({k: v.detach() for k, v in f.items()}, O.Arch(**VD)), chunk=64, rays=(o, d), retraw=False)
        (((out[0] - target.to(dtype)) ** 2).mean() + ((out[3]["rgb0"] - target.to(dtype)) ** 2).mean()).backward()
    finally:
        O.nerf_forward = orig
    return o.grad, d.grad


def field_grads(field, seed, sharpen, dtype=torch.float32):
    rng = np.random.default_rng(11)
    R, S = 70, 13
    pts = torch.from_numpy(rng.uniform(-2, 2, size=(R, S, 3)).astype(np.float32)).to(dtype)
    vd = torch.from_numpy(rng.normal(size=(R, 3)).astype(np.float32))
    vd = (vd / vd.norm(dim=-1, keepdim=True)).to(dtype)
    coef = torch.from_numpy(rng.normal(size=(R, S, 4)).astype(np.float32)).to(dtype)
    c = sds(seed, sharpen, dtype)
    out = O.nerf_forward(c, O.Arch(**VD), pts, vd) if field is None else field(c, pts, vd)
    (out * coef).sum().backward()
    return c


def table(title, ref, others):
    print("==", title)
    names = list(ref.keys()) if isinstance(ref, dict) else None
    for tag, g in others:
        if names:
            worst = max(((rel(g[n].grad, ref[n].grad), n) for n in names if ref[n].grad is not None))
            wc = min((cos(g[n].grad, ref[n].grad) for n in names if ref[n].grad is not None))
            print("  %-34s worst rel-L2 %.2e (%s)   min cos %.7f" % (tag, worst[0], worst[1], wc))


if __name__ == "__main__":
    torch.set_num_threads(8)
    variants = {
        "pairs everywhere, scale 2^12": dict(fwd="full", dx="full", dw="full", scale=4096.0),
        "dW hi x hi only": dict(fwd="full", dx="full", dw="hi", scale=4096.0),
        "dW G pair x X hi": dict(fwd="full", dx="full", dw="a_pair", scale=4096.0),
        "dX W pair x g hi": dict(fwd="full", dx="b_pair", dw="full", scale=4096.0),
        "dX hi only": dict(fwd="full", dx="hi", dw="full", scale=4096.0),
        "pairs, no loss scale": dict(fwd="full", dx="full", dw="full", scale=1.0),
        "dW sym, scale 2^0": dict(fwd="full", dx="full", dw="full_sym", scale=1.0),
        "dW sym, scale 2^4": dict(fwd="full", dx="full", dw="full_sym", scale=16.0),
        "dW sym, scale 2^8": dict(fwd="full", dx="full", dw="full_sym", scale=256.0),
        "dW sym, scale 2^12": dict(fwd="full", dx="full", dw="full_sym", scale=4096.0),
        "dW unscaled lo, scale 2^12": dict(fwd="full", dx="full", dw="full_u", scale=4096.0),
        "dW unscaled lo, scale 2^4": dict(fwd="full", dx="full", dw="full_u", scale=16.0),
        "dW unscaled lo, scale 2^20": dict(fwd="full", dx="full", dw="full_u", scale=2.0 ** 20),
    }
    # ---- field gradients, random linear loss
    for seed, sharpen in ((0, 1.0), (1, 2.0)):
        ref = field_grads(None, seed, sharpen)
        r64 = field_grads(None, seed, sharpen, torch.float64)
        rows = [("fp64 autograd", r64)] + [(k, field_grads(split_field(v), seed, sharpen)) for k, v in variants.items()]
        table("field gradients seed %d x%.0f vs fp32 autograd" % (seed, sharpen), ref, rows)
    # ---- training step
    l32, c32, f32 = training_grads(None)
    l64, c64, f64 = training_grads(None, torch.float64)
    print("loss fp32 %.8f fp64 %.8f" % (l32, l64))
    rows_c, rows_f = [("fp64 autograd", c64)], [("fp64 autograd", f64)]
    for k, v in variants.items():
        l, c, f = training_grads(split_field(v))
        print("loss %-30s %.8f" % (k, l))
        rows_c.append((k, c)); rows_f.append((k, f))
    table("training step, coarse vs fp32 autograd", c32, rows_c)
    table("training step, fine vs fp32 autograd", f32, rows_f)
    # ---- ray gradients
    o32, d32 = ray_grads(None)
    print("== ray gradients vs fp32 autograd   (no fp64 row: the oracle's render() casts the rays to fp32)")
    for k, v in variants.items():
        o, d = ray_grads(split_field(v))
        print("  %-34s rays_o %.2e cos %.7f   rays_d %.2e cos %.7f" % (k, rel(o, o32), cos(o, o32), rel(d, d32), cos(d, d32)))
