#!/usr/bin/env python3
"""Do the eight dies finish the exact-fp32 field kernel together?  (-DNERF_AMD_STAMPS build, GPU box)

    make -C nerf_shared_amd/csrc OUT=../../scratch_libs/libfstamps.so OBJDIR=build_fstamps EXTRA=-DNERF_AMD_STAMPS
    NERF_AMD_LIB=$PWD/scratch_libs/libfstamps.so python tools/micro/f32_xcd_ends.py

One workgroup per 64 points, dispatched round robin over the XCDs: every die gets an eighth of the workgroups, and the
launch lasts as long as the slowest die needs for its eighth.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerf_shared_amd import _lib, nerf, synth  # noqa: E402

ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)


def main():
    dev = torch.device("cuda:0")
    fn = _lib.lib.nerf_amd_x_f32_xcd
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    m = nerf.NeRF(**ARCH)
    m.load_state_dict(synth.torch_state_dict(1, 3.0, **{**ARCH, "skips": (4,)}))
    m = m.to(dev).requires_grad_(False)
    m.precision = "fp32"
    g = torch.Generator(device="cpu").manual_seed(0)
    R, S = 8192, 192
    pts = (torch.rand(R, S, 3, generator=g) * 6 - 3).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    for _ in range(3):
        m(pts, vd)
    torch.cuda.synchronize()
    for rep in range(3):
        assert fn(None, 1) == 0
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        m(pts, vd)
        b_.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        assert fn(buf, 0) == 0
        v = np.frombuffer(buf, dtype=np.uint64).reshape(8, 2).astype(np.float64) / 100.0
        t0 = v[:, 0].min()
        ms = a.elapsed_time(b_)
        print("launch %.2f ms; per XCD start / end (us after the first start):" % ms,
              ["%d: %.0f / %.0f" % (x, v[x, 0] - t0, v[x, 1] - t0) for x in range(8)])
        ends = v[:, 1] - t0
        print("   the dies' ends span %.0f us = %.1f %% of the launch; mean idle of a die at the end %.1f %%" % (
            ends.max() - ends.min(), 100 * (ends.max() - ends.min()) / (ms * 1e3), 100 * (ends.max() - ends).mean() / (ms * 1e3)))


if __name__ == "__main__":
    main()
