#!/usr/bin/env python3
"""Same-process A/B of whole-view renders (the bench's workload: 400x400, 64+128, 8x256 with view branch) under two
values of nerf_amd_set_tuning(0, .): interleaved rounds, bit-identical outputs required.

    python tools/micro/view_ab.py 0 44 42     # 44: the fine-pass field kernels on a stream of their own; 42: static tile deal
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import torch  # noqa: E402

from nerf_shared_amd import _lib, nerf, render_utils, synth  # noqa: E402

ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0, 44]
    dev = torch.device("cuda:0")
    ms = []
    for seed in (1, 19):
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 3.0, **{**ARCH, "skips": (4,)}))
        ms.append(m.to(dev).requires_grad_(False))
    r = render_utils.Renderer(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, near=2.0, far=6.0)
    K = synth.lego_intrinsics(400, 400)
    c2w = torch.from_numpy(synth.LEGO_C2W)
    times = {v: [] for v in variants}
    outs = {}
    with torch.no_grad():
        for rnd in range(12):
            for v in variants:
                _lib.check(_lib.lib.nerf_amd_set_tuning(0, v), "tuning")
                for _ in range(2):
                    r.render(400, 400, K, ms[0], ms[1], chunk=32768, c2w=c2w, retraw=False)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    out = r.render(400, 400, K, ms[0], ms[1], chunk=32768, c2w=c2w, retraw=False)
                b.record()
                torch.cuda.synchronize()
                times[v].append(a.elapsed_time(b) / 5)
                outs[v] = [t.clone() for t in out[:3]]
    _lib.lib.nerf_amd_set_tuning(0, 0)
    for v in variants:
        print("tuning %3d: median %.3f ms per view, best %.3f" % (v, statistics.median(times[v]), min(times[v])))
    ref = outs[variants[0]]
    for v in variants[1:]:
        print("tuning %d bit-identical to %d:" % (v, variants[0]), all(torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) for x, y in zip(ref, outs[v])))


if __name__ == "__main__":
    main()
