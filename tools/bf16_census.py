#!/usr/bin/env python3
"""Whole-frame census of the bf16 field mode against the exact-fp32 mode (both on the GPU, same weights, same rays).

The reference sets the last interval of every ray to 1e10 (render_utils.py:257), so the last sample's alpha is a step
function of the sign of its raw sigma: 1 - exp(-relu(sigma) * 1e10) is 0 or 1.  Where the field leaves sigma ~ 0 at the far
plane (random-init weights do, everywhere) any rounding difference flips a whole ray's remaining transmittance onto or
off the last sample's colour.  This tool counts what a frame's bf16-vs-fp32 PSNR is made of:

  psnr_whole_frame          PSNR(bf16 frame, fp32 frame) over every pixel
  frac_rays_off_by_0p1      fraction of rays with max_c |rgb_bf16 - rgb_fp32| > 0.1  ("flagged")
  flagged_last_sign_flips   flagged rays whose LAST fine sample has sigma of opposite sign in the two modes
  flagged_explained_by_last flagged rays that are no longer flagged once the bf16 ray is re-composited with the fp32
                            mode's last-sample sigma (everything else bf16) -- the flip alone explains them
  psnr_without_flagged      PSNR over the unflagged rays
  psnr_last_neutralised     PSNR of the frame with every flagged ray re-composited that way

    python tools/bf16_census.py [--workload c3|c4] [--out gpurun_out/census.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402


def _psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2) + 1e-30))


def census(renderer, H, W, K, c2w, coarse, fine, chunk=32768, thresh=0.1, max_flagged=65536, frames=None):
    """coarse / fine: GPU NeRF modules (their .precision is switched and restored).  frames: optional
    {"bf16": rgb [N,3], "fp32": rgb [N,3]} already rendered through Renderer.render."""
    from nerf_shared_amd import utils
    dev = next(coarse.parameters()).device
    N = H * W
    was = (coarse.precision, None if fine is None else fine.precision)

    def set_prec(p):
        coarse.precision = p
        if fine is not None:
            fine.precision = p

    try:
        with torch.no_grad():
            if frames is None:
                frames = {}
                for prec in ("fp32", "bf16"):
                    set_prec(prec)
                    frames[prec] = renderer.render(H, W, K, coarse, fine, chunk=chunk, c2w=c2w, retraw=False)[0].reshape(N, 3)
            rb, rf = frames["bf16"].reshape(N, 3), frames["fp32"].reshape(N, 3)
            d = (rb - rf).abs().max(-1).values
            flagged = torch.nonzero(d > thresh).flatten()
            out = {"rays": int(N), "psnr_whole_frame": _psnr(rb, rf), "flagged": int(flagged.numel()),
                   "frac_rays_off_by_0p1": float(flagged.numel()) / N,
                   "median_abs_err": float((rb - rf).abs().median()), "max_abs_err": float(d.max())}
            keep = torch.ones(N, dtype=torch.bool, device=dev)
            keep[flagged] = False
            out["psnr_without_flagged"] = _psnr(rb[keep], rf[keep])
            if flagged.numel() == 0:
                out.update(flagged_last_sign_flips=0, flagged_explained_by_last=0, psnr_last_neutralised=out["psnr_whole_frame"])
                return out
            sel = flagged[:max_flagged]
            batch = utils.make_ray_batch(H, W, K, c2w, renderer.near, renderer.far, renderer.use_viewdirs, renderer.ndc,
                                         device=dev)[sel].contiguous()
            part = {}
            for prec in ("fp32", "bf16"):
                set_prec(prec)
                part[prec] = renderer.render_rays(batch, coarse, fine, retraw=True, retweights=True)
            # the frame's own values come back (chunk invariance), so the census talks about the frame
            assert torch.equal(part["bf16"]["rgb_map"], rb[sel]) and torch.equal(part["fp32"]["rgb_map"], rf[sel])
            sig_b, sig_f = part["bf16"]["raw"][:, -1, 3], part["fp32"]["raw"][:, -1, 3]
            flips = (sig_b > 0) != (sig_f > 0)
            raw_fix = part["bf16"]["raw"].clone()
            raw_fix[:, -1, 3] = sig_f
            rgb_fix = renderer.raw2outputs(raw_fix, part["bf16"]["z_vals"], batch[:, 3:6])[0]
            still = (rgb_fix - rf[sel]).abs().max(-1).values > thresh
            out.update(flagged_examined=int(sel.numel()), flagged_last_sign_flips=int(flips.sum()),
                       flagged_explained_by_last=int((~still).sum()),
                       last_sigma_abs_median_fp32=float(sig_f.abs().median()))
            rb2 = rb.clone()
            rb2[sel] = rgb_fix
            out["psnr_last_neutralised"] = _psnr(rb2, rf)
            # the same census on the coarse pass would need rgb0 frames; the fine pass is what the image shows
            return out
    finally:
        coarse.precision = was[0]
        if fine is not None:
            fine.precision = was[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=["c3", "c4"])
    ap.add_argument("--sharpen", type=float, default=3.0)
    ap.add_argument("--seeds", default=None, help="coarse,fine weight seeds (default: the frame tests' 1,19 / 1,12)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from nerf_shared_amd import nerf, render_utils, synth
    dev = torch.device("cuda:0")
    arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
    base = dict(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0,
                ndc=False, lindisp=False, near=2.0, far=6.0)
    if a.workload == "c3":
        H = W = 800
        K, c2w, cfg, seeds = synth.lego_intrinsics(H, W), synth.LEGO_C2W, base, (1, 19)
    else:
        H, W, focal = 378, 504, 408.0
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
        cfg, seeds = dict(base, ndc=True, near=0.0, far=1.0, N_importance=64, white_bkgd=False), (1, 12)
    if a.seeds:
        seeds = tuple(int(s) for s in a.seeds.split(","))
    models = []
    for seed in seeds:
        m = nerf.NeRF(**arch)
        m.load_state_dict(synth.torch_state_dict(seed, a.sharpen, **{**arch, "skips": (4,)}))
        models.append(m.to(dev).requires_grad_(False))
    r = render_utils.Renderer(**cfg)
    out = census(r, H, W, K, torch.from_numpy(np.asarray(c2w, np.float32)), models[0], models[1])
    out.update(workload=a.workload, sharpen=a.sharpen, seeds=list(seeds))
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
