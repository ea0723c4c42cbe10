#!/usr/bin/env python3
"""Per-ray attribution of the differences between a render of this build and the reference's (oracle's) render of the same
rays, for the precision modes that claim the reference's arithmetic ("fp32", "fp32_split") -- what tools/bf16_census.py
does for the bf16 mode.  TEST / MEASUREMENT INFRASTRUCTURE: it imports the oracle, so only tests/ and bench.py's check use it.

Two places of the reference turn a last-bit difference into a visible one, and both are properties of ITS arithmetic:

 (a) sample_pdf (utils.py:105-113): a fine sample is  bins_lo + (u - cdf_lo) / (cdf_hi - cdf_lo) * (bins_hi - bins_lo).
     The cdf is an fp32 running sum (error ~1e-7); a sample drawn in a bin of mass m moves by (1e-7 / m) of the bin width,
     and `denom < 1e-5 -> 1` switches the formula altogether.  The field is evaluated at 2^9 x (positional encoding), so a
     sample that moves by 1e-5 changes its raw output by ~1e-2 of its range.  A ray is "ill-conditioned in (a)" when one of
     its fine samples was drawn in a bin of mass below `mass_thresh`.
 (b) raw2outputs (render_utils.py:257): the last interval is 1e10, so the last sample's alpha is a step function of the
     sign of its sigma.  A ray is "ill-conditioned in (b)" when |sigma_last| of the pass is below `sigma_thresh` (the stage
     tolerance of raw) while the transmittance that reaches the last sample is not negligible.

census() classifies every ray: within `tol` of the reference; or moved and carrying (a) or (b); or UNEXPLAINED -- a real
defect that moved rays without either cause shows up there.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nerf_oracle as O  # noqa: E402


def oracle_stages(cfg, batch, coarse, fine, pytest=False, t_rand=None, noise0=None, noise1=None, u=None):
    """The oracle's render_rays (render_utils.py:67-174) stage by stage, keeping what the census needs: per-sample bin masses of
    the fine samples, the last sigma and the transmittance in front of it for both passes, and the reference maps."""
    n = batch.shape[0]
    rays_o, rays_d = batch[:, 0:3], batch[:, 3:6]
    viewdirs = batch[:, -3:] if batch.shape[-1] > 8 else None
    near, far = batch[:, 6:7], batch[:, 7:8]
    if pytest and cfg.perturb > 0.0:
        t_rand = O.pytest_uniform([n, cfg.N_samples])
    z = O.coarse_z_vals(cfg, near, far, n, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]

    def noise_for(shape, injected):
        if not cfg.raw_noise_std > 0.0:
            return None
        if pytest:
            return O.pytest_uniform(shape) * cfg.raw_noise_std
        return injected

    raw0 = O.nerf_forward(coarse[0], coarse[1], pts, viewdirs)
    nz0 = noise_for(raw0[..., 3].shape, noise0)
    rgb0, disp0, acc0, w0, _ = O.raw2outputs(raw0, z, rays_d, cfg.white_bkgd, nz0)
    out = {"rgb0": rgb0, "acc0": acc0, "disp0": disp0, "sigma_last0": raw0[:, -1, 3] + (nz0[:, -1] if nz0 is not None else 0.0),
           "t_last0": 1.0 - w0[:, :-1].sum(-1)}
    if cfg.N_importance <= 0:
        out.update(rgb_map=rgb0, acc_map=acc0, disp_map=disp0)
        return out
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    det = cfg.perturb == 0.0
    if pytest:
        u = O.pytest_u_for_sample_pdf(n, cfg.N_importance, det)
    w = w0[..., 1:-1] + 1e-5                                   # sample_pdf's own first lines (utils.py:76-79)
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    uu = u if u is not None else torch.linspace(0.0, 1.0, steps=cfg.N_importance).expand(n, cfg.N_importance)
    idx = torch.searchsorted(cdf, uu.contiguous(), right=True)
    lo, hi = torch.clamp(idx - 1, min=0), torch.clamp(idx, max=cdf.shape[-1] - 1)
    mass = torch.gather(cdf, -1, hi) - torch.gather(cdf, -1, lo)                     # denom before the < 1e-5 switch
    # (u = 1 lands past the last cdf entry: lo == hi, the "bin" has zero width and the sample is bins[-1] whatever t is)
    out["bin_mass"] = torch.where(lo == hi, torch.full_like(mass, float("inf")), mass)
    out["bin_width"] = torch.gather(z_mid, -1, hi) - torch.gather(z_mid, -1, lo)
    z_samples = O.sample_pdf(z_mid, w0[..., 1:-1], cfg.N_importance, det=det, u=u)
    zf, _ = torch.sort(torch.cat([z, z_samples], -1), -1)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * zf[..., :, None]
    net = coarse if fine is None else fine
    raw1 = O.nerf_forward(net[0], net[1], pts, viewdirs)
    nz1 = noise_for(raw1[..., 3].shape, noise1)
    rgb, disp, acc, w1, _ = O.raw2outputs(raw1, zf, rays_d, cfg.white_bkgd, nz1)
    out.update(rgb_map=rgb, acc_map=acc, disp_map=disp, z_vals=zf, raw=raw1, weights=w1,
               sigma_last=raw1[:, -1, 3] + (nz1[:, -1] if nz1 is not None else 0.0), t_last=1.0 - w1[:, :-1].sum(-1))
    return out


def attribute(cfg, batch, coarse, fine, got, ref=None, pytest=False, tol=2e-4, rtol=2e-4, big=1e-3, sigma_thresh=2e-4,
              t_thresh=1e-3, ref_maps=None):
    """Per-ray attribution of a two-pass render of the build against the reference.

    cfg: O.RenderCfg; batch [R, 8|11] (CPU); coarse / fine: the oracle's (state_dict, Arch) pairs; got: the build's
    render_rays(batch, ..., retraw=True, retweights=True) outputs (CPU tensors: rgb_map, acc_map, raw, z_vals, weights);
    ref: oracle_stages(...) of the same call (computed here when None); ref_maps: optional {"rgb_map", "acc_map", "z_vals"} of the
    REFERENCE itself (a golden fixture) to measure the movement against instead of the oracle's maps.

    Three statements per ray, together a complete account (no fraction of rays is waved through):
      staged      the build's raw equals the oracle's field evaluated ON THE BUILD'S OWN DEPTHS within the stage tolerance,
                  and its maps equal the oracle's compositing of the build's own raw: `staged_*` maxima over ALL rays;
      displaced   what moved against the reference (max |rgb, acc difference| > tol) must have a cause in the reference's own
                  conditioning: (a) its depths -- sample_pdf (utils.py:105-113) turns last-bit differences of the coarse
                  weights into moved samples, and the field is evaluated at 2^9 x the depth.  Either a fine sample differs
                  from the reference's by more than 4 ulp, or THE ORACLE ITSELF, given the build's depths, moves at least
                  half as far from the reference as the build did (`d_depth`: a sharpened field turns 3 ulp of depth into
                  3e-4 of colour; found by the round-4 sweep with new seeds) -- or (b) a last sample with |sigma| below the
                  stage tolerance and transmittance left (dists[-1] = 1e10 makes its alpha a step function,
                  render_utils.py:257);
      unexplained rays that moved with neither: must be 0.
    """
    if ref is None:
        with torch.no_grad():
            ref = oracle_stages(cfg, batch, coarse, fine, pytest=pytest)
    maps = ref_maps if ref_maps is not None else ref

    def f64(x):
        return x.detach().cpu().double().numpy() if torch.is_tensor(x) else np.asarray(x, np.float64)

    r_rgb, g_rgb = f64(maps["rgb_map"]), f64(got["rgb_map"])
    d = np.abs(g_rgb - r_rgb).max(-1) / (1.0 + rtol / tol * np.abs(r_rgb).max(-1))
    d = np.maximum(d, np.abs(f64(got["acc_map"]) - f64(maps["acc_map"])) / (1.0 + rtol / tol * np.abs(f64(maps["acc_map"]))))
    z_ref = maps["z_vals"] if "z_vals" in maps else ref["z_vals"]
    z_ref32 = (z_ref.detach().cpu().numpy() if torch.is_tensor(z_ref) else np.asarray(z_ref)).astype(np.float32)
    dz = np.abs(f64(got["z_vals"]) - z_ref32.astype(np.float64))
    displaced = (dz > 4.0 * np.spacing(np.abs(z_ref32)).astype(np.float64)).any(-1)
    cond_b = (np.abs(f64(ref["sigma_last"])) < sigma_thresh) & (f64(ref["t_last"]) > t_thresh)
    moved, moved_big = d > tol, d > big
    # ---- staged: the oracle's fine pass on the build's own depths, the oracle's compositing of the build's own raw -- and the
    # oracle's compositing of ITS OWN raw on those depths: what the depths alone do to the maps
    with torch.no_grad():
        z = got["z_vals"].detach().cpu()
        pts = batch[:, None, 0:3] + batch[:, None, 3:6] * z[..., None]
        net = coarse if fine is None else fine
        raw_o = O.nerf_forward(net[0], net[1], pts, batch[:, -3:] if batch.shape[-1] > 8 else None)
        noise1 = (O.pytest_uniform(list(z.shape)) * cfg.raw_noise_std) if (pytest and cfg.raw_noise_std > 0.0) else None
        rgb_o, disp_o, acc_o, w_o, _ = O.raw2outputs(got["raw"].detach().cpu(), z, batch[:, 3:6], cfg.white_bkgd, noise1)
        rgb_f, _, acc_f, _, _ = O.raw2outputs(raw_o, z, batch[:, 3:6], cfg.white_bkgd, noise1)
    d_depth = np.abs(f64(rgb_f) - r_rgb).max(-1) / (1.0 + rtol / tol * np.abs(r_rgb).max(-1))
    d_depth = np.maximum(d_depth, np.abs(f64(acc_f) - f64(maps["acc_map"])) / (1.0 + rtol / tol * np.abs(f64(maps["acc_map"]))))
    by_depth = displaced | (d_depth >= 0.5 * d)
    out = {"rays": int(d.shape[0]), "max_abs": float(d.max()), "median_abs": float(np.median(d)),
           "frac_gt_tol": float(moved.mean()), "frac_gt_1e-3": float(moved_big.mean()),
           "frac_displaced": float(displaced.mean()), "frac_last_flip_prone": float(cond_b.mean()),
           "unexplained": int((moved & ~by_depth & ~cond_b).sum()),
           "max_abs_undisplaced": float(d[~by_depth & ~cond_b].max()) if (~by_depth & ~cond_b).any() else 0.0,
           "z_displacement_max": float(dz.max())}
    raw_g = got["raw"].detach().cpu()
    out["_staged_disp"], out["_staged_acc"], out["_staged_weights"] = disp_o, acc_o, w_o
    out["staged_raw_max"] = float(((raw_g - raw_o).abs() / (1.0 + raw_o.abs())).max())
    out["staged_rgb_max"] = float((got["rgb_map"].detach().cpu() - rgb_o).abs().max())
    out["staged_acc_max"] = float((got["acc_map"].detach().cpu() - acc_o).abs().max())
    out["_d"], out["_displaced"] = d, displaced
    # the rays the caller will ask about first: how far they moved, and by how many ulp their worst sample is displaced
    bad = np.nonzero(moved & ~by_depth & ~cond_b)[0]
    ulp = (dz / np.spacing(np.abs(z_ref32)).astype(np.float64)).max(-1)
    out["unexplained_rays"] = [(int(i), float(d[i]), float(ulp[i])) for i in bad[:8]]
    return out


def census(ref, got, tol=2e-4, big=1e-3, mass_thresh=1e-3, sigma_thresh=2e-4, t_thresh=1e-3):
    """ref: oracle_stages(...) output; got: the build's render_rays output (rgb_map, acc_map [, rgb0, acc0]) on the same rays.
    Returns the counts for the final maps (and the coarse maps, where only (b) applies)."""
    def to_np(x):
        return x.detach().cpu().double().numpy() if torch.is_tensor(x) else np.asarray(x, np.float64)

    def one(rgb_k, acc_k, sig_k, t_k, with_a):
        r, g = to_np(ref[rgb_k]), to_np(got[rgb_k])
        d = np.abs(g - r).max(-1)
        d = np.maximum(d, np.abs(to_np(got[acc_k]) - to_np(ref[acc_k])))
        n = d.shape[0]
        cond_b = (np.abs(to_np(ref[sig_k])) < sigma_thresh) & (to_np(ref[t_k]) > t_thresh)
        cond_a = (to_np(ref["bin_mass"]).min(-1) < mass_thresh) if (with_a and "bin_mass" in ref) else np.zeros(n, bool)
        moved, moved_big = d > tol, d > big
        explained = cond_a | cond_b
        return {"rays": int(n), "max_abs": float(d.max()), "median_abs": float(np.median(d)),
                "frac_gt_tol": float(moved.mean()), "frac_gt_1e-3": float(moved_big.mean()),
                "frac_ill_conditioned": float(explained.mean()), "frac_cond_a": float(cond_a.mean()), "frac_cond_b": float(cond_b.mean()),
                "unexplained_gt_tol": int((moved & ~explained).sum()), "unexplained": int((moved_big & ~explained).sum()),
                "max_abs_well_conditioned": float(d[~explained].max()) if (~explained).any() else 0.0,
                "_d": d, "_explained": explained}

    out = {"final": one("rgb_map", "acc_map", "sigma_last" if "sigma_last" in ref else "sigma_last0",
                        "t_last" if "t_last" in ref else "t_last0", True)}
    if "rgb0" in got and "sigma_last" in ref:
        out["coarse"] = one("rgb0", "acc0", "sigma_last0", "t_last0", False)
    return out


def strip(c):
    """census() output without the per-ray arrays (for JSON)."""
    return {k: {kk: vv for kk, vv in v.items() if not kk.startswith("_")} for k, v in c.items()}
