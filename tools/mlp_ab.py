#!/usr/bin/env python3
"""A/B timing of fused-MLP kernel variants, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24).  Variants are (tuning key, value) settings of
nerf_amd_set_tuning; outputs must be bit-identical across variants.

    python tools/mlp_ab.py [--rounds 15] [--rays 4096] [--samples 192] --variants 0:0 0:41 0:40

Values of key 0 in a normal build: 0 = default (pipelined kernel, pinned read-ahead, split DMA), 41 = the simple per-tile
kernel with the same pipeline shape, 40 = round 1's pipeline shape, 100+ = the 32x32x16 kernel.  A build with
-DNERF_AMD_EXPERIMENTS adds the shapes listed in launch_mlp_bf16_s16 (mlp_bf16_s16.hip).
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import torch  # noqa: E402

from nerf_shared_amd import _lib, nerf, synth  # noqa: E402

FLOP = 1186816
ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--variants", nargs="+", default=["0:8", "0:4"])
    ap.add_argument("--sharpen", type=float, nargs="+", default=[1.0, 3.0])
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-check", action="store_true", help="ablation variants produce wrong results on purpose")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    variants = [tuple(int(x) for x in v.split(":")) for v in args.variants]
    g = torch.Generator(device="cpu").manual_seed(0)
    pts = (torch.rand(args.rays, args.samples, 3, generator=g) * 6 - 3).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(args.rays, 3, generator=g), dim=-1).to(dev)
    P = args.rays * args.samples
    report = {}
    for sharpen in args.sharpen:
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(1, sharpen, **{**ARCH, "skips": (4,)}))
        m = m.to(dev).requires_grad_(False)          # inference kernels (with gradients on, NeRF.forward saves activations)
        m.precision = args.precision
        outs, times = {}, {v: [] for v in variants}
        for v in variants:                        # warm-up + reference outputs
            _lib.check(_lib.lib.nerf_amd_set_tuning(*v), "set_tuning")
            outs[v] = m(pts, vd).clone()
        torch.cuda.synchronize()
        for v in ([] if args.no_check else variants[1:]):
            assert torch.equal(outs[v], outs[variants[0]]), "variant %s changes the result" % (v,)
        for _ in range(args.rounds):
            for v in variants:
                _lib.check(_lib.lib.nerf_amd_set_tuning(*v), "set_tuning")
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(4):
                    m(pts, vd)
                b.record()
                b.synchronize()
                times[v].append(a.elapsed_time(b) / 4)
        for v in variants:
            med, mn = statistics.median(times[v]), min(times[v])
            report["sharpen%g_%d:%d" % (sharpen, v[0], v[1])] = {
                "median_ms": med, "min_ms": mn, "tflops_median": P * FLOP / med / 1e9, "tflops_best": P * FLOP / mn / 1e9}
    _lib.lib.nerf_amd_set_tuning(0, 0)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
