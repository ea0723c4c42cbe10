#!/usr/bin/env python3
"""Per-ray data behind tools/precision_census.py: for a few configurations, the fp32-class modes' differences from the
oracle next to each ray's conditioning numbers (smallest sample_pdf bin mass, |sigma_last|, transmittance at the last
sample) -- the table the census thresholds were read from.

    python tools/experiments/census_probe.py gpurun_out/census_probe.npz
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import nerf, render_utils, synth  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402
import precision_census as PC  # noqa: E402

VD = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
BASE = dict(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False,
            lindisp=False, near=2.0, far=6.0)


def main(out_path):
    dev = torch.device("cuda:0")
    K = synth.lego_intrinsics(400, 400)
    dump = {}
    cases = {"c19_x3": ((1, 19), 3.0, {}), "c12_x3": ((1, 12), 3.0, {}), "s0_x1": ((0, 10), 1.0, {}),
             "c19_x3_perturb": ((1, 19), 3.0, dict(perturb=1.0))}
    for tag, (seeds, sharpen, over) in cases.items():
        cfg = dict(BASE, **over)
        idx = np.arange(80000, 80000 + 2048)
        ro, rd = synth.rays_np(400, 400, K, synth.LEGO_C2W, idx)
        batch = torch.from_numpy(synth.ray_batch_np(ro, rd, 2.0, 6.0, True))
        sds = [synth.torch_state_dict(s, sharpen, **{**VD, "skips": (4,)}) for s in seeds]
        om = [(O.state_dict_to_torch(sd), O.Arch(**VD)) for sd in sds]
        with torch.no_grad():
            ref = PC.oracle_stages(O.RenderCfg(**cfg), batch, om[0], om[1], pytest=cfg["perturb"] > 0)
        r = render_utils.Renderer(**cfg)
        for prec in ("fp32", "fp32_split"):
            ms = []
            for sd in sds:
                m = nerf.NeRF(**VD)
                m.load_state_dict(sd)
                m.precision = prec
                ms.append(m.to(dev))
            with torch.no_grad():
                got = r.render_rays(batch.to(dev), ms[0], ms[1], retraw=True, retweights=True, pytest=cfg["perturb"] > 0)
            c = PC.census(ref, got)
            print(tag, prec, PC.strip(c))
            key = tag + "__" + prec
            dump[key + "__d"] = c["final"]["_d"]
            dump[key + "__d0"] = c["coarse"]["_d"]
            dump[key + "__dz"] = (got["z_vals"].cpu() - ref["z_vals"]).abs().numpy()
        dump[tag + "__min_mass"] = ref["bin_mass"].min(-1).values.numpy()
        dump[tag + "__bin_mass"] = ref["bin_mass"].numpy()
        dump[tag + "__bin_width"] = ref["bin_width"].numpy()
        dump[tag + "__weights"] = ref["weights"].numpy()
        dump[tag + "__z_ref"] = ref["z_vals"].numpy()
        dump[tag + "__sigma_last"] = ref["sigma_last"].numpy()
        dump[tag + "__t_last"] = ref["t_last"].numpy()
        dump[tag + "__sigma_last0"] = ref["sigma_last0"].numpy()
        dump[tag + "__t_last0"] = ref["t_last0"].numpy()
    np.savez_compressed(out_path, **dump)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/census_probe.npz")
