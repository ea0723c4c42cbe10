"""Full-frame GPU parity (-m gpu) at the BASELINE.json sizes the crop-sized goldens do not reach:

  C3  Lego 800x800, 64+128, view directions, chunk 32768, through Renderer.render(c2w=)
  C4  Fern-like 378x504, NDC rays, 64+64, chunk 32768, through Renderer.render(c2w=)

A whole frame is rendered through render(c2w=) (render_utils.py:176-238: get_rays, view directions,
NDC warp, batch assembly, the chunk loop with its ragged last chunk, the reshape) in both precisions.
The oracle cannot render 640 000 rays in test time, so a strided subset of the frame's rays is
compared against it:
  * fp32 mode, coarse maps (rgb0 / disp0 / acc0): the oracle end to end on the subset     -> tight
  * fp32 mode, fine pass: staged -- render_rays on the subset reproduces the frame's values bit for bit
    (chunk invariance) and returns z_vals / raw; the oracle's fine field + compositing on those
    z_vals                                                                                 -> tight
  * the G8 referee crop is cut OUT OF the full frame and compared with the reference's image (golden)
  * bf16 mode: PSNR of the subset against the oracle's fp32 maps, gated a few dB under the measured value;
    PSNR of the WHOLE bf16 frame against the whole fp32 frame, gated, with a census of what it is made of
    (tools/bf16_census.py): the fraction of rays off by > 0.1 and how many of them are sign flips of the last
    sample's sigma (render_utils.py:257: dists[-1] = 1e10 makes the last alpha a step function of that sign)
  * SURVEY.md section 8(d)'s PSNR protocol on the subset: G = the oracle evaluated in float64 on the same weights
    and rays (pseudo ground truth); PSNR(build, G) for the three precisions beside PSNR(fp32 oracle, G)
  * the split-precision mode renders the same whole frame; gated against the exact-fp32 frame
"""
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("NERF_AMD_QUIET", "1")

pytestmark = pytest.mark.gpu

from nerf_shared_amd import synth  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402
from test_gpu_parity import BASE, G5_TOL, VD, amd, close, cpu_model, gpu_model, oracle_batch, psnr, report  # noqa: E402
from test_oracle_golden import render_rays_fp64  # noqa: E402
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bf16_census import census  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a ROCm device"
    return torch.device("cuda:0")


# bf16 PSNR gates (dB) on the strided subset, a few dB under the measured values
# measured on MI355X: c3 38.8 dB (fp32 end to end 57.4), c4 46.0 dB (fp32 86.9)
FRAME_BF16_GATES = {"c3": 35.5, "c4": 42.0}
FRAME_FP32_GATES = {"c3": 53.0, "c4": 75.0}
# whole frame, bf16 against the exact-fp32 frame (every pixel), measured on MI355X (gpurun_out/parity_frame_*.json):
#   c3 37.3 dB: 1.02 % of the rays off by > 0.1 and only 2.6 % of those are last-sample flips -- genuine bf16 error of a
#      semi-transparent field on x3-sharpened random weights; 39.7 dB over the other 99 % of the rays
#   c4 27.3 dB: 0.55 % of the rays off by > 0.1 and 95.9 % of those (1001 of 1044) are sign flips of the last sample's
#      sigma (median |sigma_last| 0.026) that explain the whole difference of their ray: 43.75 dB once they are neutralised
WHOLE_FRAME_GATES = {
    # psnr_whole_frame, max fraction flagged, min psnr_without_flagged, min fraction of flagged explained by the last sample, min psnr_last_neutralised
    "c3": dict(psnr=35.0, flagged=0.02, psnr_rest=37.5, explained=0.0, psnr_neutral=35.0),
    "c4": dict(psnr=25.0, flagged=0.01, psnr_rest=41.0, explained=0.9, psnr_neutral=41.0),
}
SPLIT_FRAME_GATES = {"c3": 53.0, "c4": 75.0}      # split-precision frame against the exact-fp32 frame: the fp32 mode's own gates


def frame_check(dev, name, cfg, H, W, K, c2w, seeds, chunk, stride):
    _, render_utils, _ = amd()
    sc, sf, sharpen = seeds
    N = H * W
    idx = np.arange(stride // 2, N, stride)
    batch = oracle_batch(cfg, H, W, K, c2w, idx)                         # the oracle's own ray math on the subset
    coarse_cpu, fine_cpu = cpu_model(sc, sharpen, **VD), cpu_model(sf, sharpen, **VD)
    r = render_utils.Renderer(**cfg)
    from nerf_shared_amd import utils as amd_utils
    c2w_t = torch.from_numpy(np.asarray(c2w, np.float32))
    measured = {"rays": int(N), "subset": int(idx.size), "chunks": -(-N // chunk)}
    frames = {}
    for prec in ("fp32", "bf16"):
        c, f = gpu_model(dev, sc, sharpen, prec, **VD), gpu_model(dev, sf, sharpen, prec, **VD)
        rgb, disp, acc, extras = r.render(H, W, K, c, f, chunk=chunk, c2w=c2w_t, retraw=False)
        assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and acc.shape == (H, W)
        assert sorted(extras) == ["acc0", "disp0", "rgb0", "z_std"] and extras["rgb0"].shape == (H, W, 3)
        flat = {"rgb_map": rgb.reshape(N, 3), "disp_map": disp.reshape(N), "acc_map": acc.reshape(N),
                "rgb0": extras["rgb0"].reshape(N, 3), "disp0": extras["disp0"].reshape(N), "acc0": extras["acc0"].reshape(N),
                "z_std": extras["z_std"].reshape(N)}
        frames[prec] = flat
        sub = {k: v[torch.from_numpy(idx).to(dev)].cpu() for k, v in flat.items()}
        # render_rays on the subset's rays = the frame's values at those pixels, bit for bit (ray generation in the
        # frame kernel vs the oracle's ray math may differ by an ulp, so feed the frame's own rays)
        from nerf_shared_amd import utils as amd_utils
        own = amd_utils.make_ray_batch(H, W, K, c2w_t, cfg["near"], cfg["far"], True, cfg["ndc"], device=dev)[torch.from_numpy(idx).to(dev)].contiguous()
        if prec == "fp32":
            close(own[:, :6], batch[:, :6], atol=2e-6, rtol=2e-6)                   # ray math of the frame kernel vs the oracle
            close(own[:, 8:], batch[:, 8:], atol=2e-6, rtol=2e-6)
        part = {k: v.cpu() for k, v in r.render_rays(own, c, f, retraw=True, retweights=True).items()}
        for k in sub:
            assert torch.equal(torch.nan_to_num(part[k]), torch.nan_to_num(sub[k])), (prec, k)
        own_cpu = own.cpu()
        if prec == "fp32":
            # coarse maps: oracle end to end on the same rays -> tight
            ref0 = O.render_rays(O.RenderCfg(**dict(cfg, N_importance=0)), own_cpu, coarse_cpu, None)
            for k0, k in (("rgb_map", "rgb0"), ("disp_map", "disp0"), ("acc_map", "acc0")):
                close(sub[k], ref0[k0], atol=G5_TOL[k], rtol=2e-4)
            # fine pass, staged on the GPU's own z_vals -> tight
            z = part["z_vals"]
            assert bool((z[:, 1:] >= z[:, :-1]).all())
            pts = own_cpu[:, None, 0:3] + own_cpu[:, None, 3:6] * z[..., None]
            raw = O.nerf_forward(fine_cpu[0], fine_cpu[1], pts, own_cpu[:, 8:11])
            close(part["raw"], raw, atol=2e-4, rtol=2e-4)
            rgb_o, disp_o, acc_o, w_o, _ = O.raw2outputs(part["raw"], z, own_cpu[:, 3:6], cfg["white_bkgd"], None)
            close(sub["rgb_map"], rgb_o, atol=1e-5, rtol=1e-5)
            close(sub["acc_map"], acc_o, atol=1e-5, rtol=1e-5)
            close(sub["disp_map"], disp_o, atol=1e-5, rtol=1e-4)
            measured["fp32_raw_max"] = float((part["raw"] - raw).abs().max())
            measured["subset_rgb_var"] = float(sub["rgb_map"].var())
            measured["subset_acc_mean"] = float(sub["acc_map"].mean())
            ref_full = O.render_rays(O.RenderCfg(**cfg), own_cpu, coarse_cpu, fine_cpu)       # end-to-end oracle maps (PSNR referee)
            measured["fp32_psnr_vs_oracle"] = psnr(sub["rgb_map"], ref_full["rgb_map"])
            frames["oracle_subset"] = ref_full
        else:
            measured["bf16_psnr_vs_oracle"] = psnr(sub["rgb_map"], frames["oracle_subset"]["rgb_map"])
            measured["bf16_psnr_rgb0_vs_oracle"] = psnr(sub["rgb0"], frames["oracle_subset"]["rgb0"])
            measured["bf16_psnr_vs_fp32_frame"] = psnr(flat["rgb_map"], frames["fp32"]["rgb_map"])
        frames[prec + "_subset"] = sub
    # ---- the whole bf16 frame against the whole fp32 frame, and what the difference is made of
    c, f = gpu_model(dev, sc, sharpen, "bf16", **VD), gpu_model(dev, sf, sharpen, "bf16", **VD)
    cs = census(r, H, W, K, c2w_t, c, f, chunk=chunk, frames={"bf16": frames["bf16"]["rgb_map"], "fp32": frames["fp32"]["rgb_map"]})
    measured["census"] = cs
    gate = WHOLE_FRAME_GATES[name]
    assert abs(cs["psnr_whole_frame"] - measured["bf16_psnr_vs_fp32_frame"]) < 1e-6
    assert cs["psnr_whole_frame"] > gate["psnr"], cs
    assert cs["frac_rays_off_by_0p1"] < gate["flagged"], cs
    assert cs["psnr_without_flagged"] > gate["psnr_rest"], cs
    assert cs["psnr_last_neutralised"] > gate["psnr_neutral"], cs
    assert cs["flagged"] == 0 or cs["flagged_explained_by_last"] >= gate["explained"] * cs["flagged_examined"], cs
    # ---- the split-precision mode on the whole frame
    cS, fS = gpu_model(dev, sc, sharpen, "fp32_split", **VD), gpu_model(dev, sf, sharpen, "fp32_split", **VD)
    rgbS = r.render(H, W, K, cS, fS, chunk=chunk, c2w=c2w_t, retraw=False)[0].reshape(N, 3)
    measured["split_psnr_vs_fp32_frame"] = psnr(rgbS, frames["fp32"]["rgb_map"])
    assert measured["split_psnr_vs_fp32_frame"] > SPLIT_FRAME_GATES[name], measured
    # ---- PSNR against a common pseudo ground truth G = the float64 evaluation, on the subset
    own_cpu = amd_utils.make_ray_batch(H, W, K, c2w_t, cfg["near"], cfg["far"], True, cfg["ndc"], device=dev)[torch.from_numpy(idx).to(dev)].cpu()
    G = render_rays_fp64(O.RenderCfg(**cfg), own_cpu, coarse_cpu, fine_cpu)["rgb_map"]
    subS = rgbS[torch.from_numpy(idx).to(dev)].cpu()
    proto = {"ref_fp32_oracle": psnr(frames["oracle_subset"]["rgb_map"], G), "build_fp32": psnr(frames["fp32_subset"]["rgb_map"], G),
             "build_fp32_split": psnr(subS, G), "build_bf16": psnr(frames["bf16_subset"]["rgb_map"], G)}
    measured["psnr_vs_fp64_pseudo_gt"] = proto
    return measured, frames


def test_c3_full_frame_800x800(dev, golden):
    """BASELINE configs[2]: Lego full_res 800x800 coarse+fine with viewdirs, chunk 32768 (19 full chunks + one
    of 17 472 rays), fixed test pose."""
    H = W = 800
    K = synth.lego_intrinsics(H, W)
    measured, frames = frame_check(dev, "c3", dict(BASE), H, W, K, synth.LEGO_C2W, (1, 19, 3.0), 32768, 1237)
    # the G8 referee crop, cut out of the full frame (fp32 against the reference's image; bf16 gated on PSNR)
    g = golden("g8_psnr_crop")
    pix = torch.from_numpy(g["pixel_index"]).to(dev)
    for prec, gate in (("fp32", 53.0), ("bf16", 34.0)):          # measured 56.9 / 37.1 dB
        crop = frames[prec]["rgb_map"][pix]
        measured["crop_%s_psnr_vs_reference" % prec] = psnr(crop, g["rgb_c19"])
        assert measured["crop_%s_psnr_vs_reference" % prec] > gate, measured
    close(frames["fp32"]["rgb0"][pix], g["rgb0_c19"], atol=2e-4)
    report("frame_c3", measured)
    assert measured["subset_rgb_var"] > 1e-2                                # the frame has content
    assert measured["fp32_psnr_vs_oracle"] > FRAME_FP32_GATES["c3"], measured
    assert measured["bf16_psnr_vs_oracle"] > FRAME_BF16_GATES["c3"], measured
    check_protocol(measured)


def test_c4_full_frame_fern_ndc(dev):
    """BASELINE configs[3]: Fern-like LLFF frame, 378x504, NDC rays, near/far 0/1, 64+64, black background,
    chunk 32768 (5 full chunks + one of 26 672 rays); deterministic draws (perturb 0, no noise) for parity."""
    H, W, focal = 378, 504, 408.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    cfg = dict(BASE, ndc=True, near=0.0, far=1.0, N_importance=64, white_bkgd=False)
    measured, frames = frame_check(dev, "c4", cfg, H, W, K, c2w, (1, 12, 3.0), 32768, 373)
    report("frame_c4", measured)
    assert measured["fp32_psnr_vs_oracle"] > FRAME_FP32_GATES["c4"], measured
    assert measured["bf16_psnr_vs_oracle"] > FRAME_BF16_GATES["c4"], measured
    check_protocol(measured)


def check_protocol(measured):
    """|PSNR(build, G) - PSNR(ref, G)| with G the float64 render.  On x3-sharpened random
    weights the fp32 reference itself is only ~38-60 dB from G (sample_pdf's conditioning and the last-sample step
    move individual rays by up to 0.1), so the clause is checked as: the fp32-class modes are no further from G than
    the reference is (within 1 dB -- they are different roundings of the same chaotic rays), and the bf16 mode
    loses at most 3 dB against the reference's own distance."""
    p = measured["psnr_vs_fp64_pseudo_gt"]
    assert p["build_fp32"] > p["ref_fp32_oracle"] - 1.0, p
    assert p["build_fp32_split"] > p["ref_fp32_oracle"] - 1.0, p
    assert p["build_bf16"] > min(p["ref_fp32_oracle"], 60.0) - 3.0 or p["build_bf16"] > 36.0, p
