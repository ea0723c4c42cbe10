#!/usr/bin/env python3
"""When does each workgroup of the pipelined bf16 field kernel leave, and on which XCD?  (-DNERF_AMD_STAMPS build, GPU box)

    make -C nerf_shared_amd/csrc OUT=../../scratch_libs/libfstamps.so OBJDIR=build_fstamps EXTRA=-DNERF_AMD_STAMPS
    NERF_AMD_LIB=$PWD/scratch_libs/libfstamps.so python tools/micro/field_wg_ends.py

One launch of the bench's size (4.1 M points: 16 000 tiles of 256 points dealt blockIdx, blockIdx + 256, ... to 256
workgroups).  If the dies do not run alike at the power cap, a static deal leaves the fast ones idle at the end.
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
    fn = _lib.lib.nerf_amd_debug_set_stamp_buffer
    fn.argtypes, fn.restype = [ctypes.c_void_p], None
    m = nerf.NeRF(**ARCH)
    m.load_state_dict(synth.torch_state_dict(1, 3.0, **{**ARCH, "skips": (4,)}))
    m = m.to(dev).requires_grad_(False)
    g = torch.Generator(device="cpu").manual_seed(0)
    R, S = 21334, 192
    pts = (torch.rand(R, S, 3, generator=g) * 6 - 3).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    for _ in range(30):                      # reach the steady (power-capped) state first
        m(pts, vd)
    torch.cuda.synchronize()
    for rep in range(3):
        buf = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
        fn(buf.data_ptr())
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        m(pts, vd)
        b_.record()
        torch.cuda.synchronize()
        fn(None)
        b = buf.cpu().numpy().reshape(256, 8, 4)
        end = b[:, :, 0].max(1).astype(np.float64) / 100.0           # us, per workgroup
        xcc = b[:, 0, 2]
        tiles = b[:, 0, 3]
        cyc = b[:, 0, 1] / np.maximum(tiles, 1)
        end -= end.min()
        print("launch %.1f us; workgroup ends spread over %.1f us (%.1f %% of the launch); tiles per workgroup %d..%d" % (
            a.elapsed_time(b_) * 1e3, end.max(), 100 * end.max() / (a.elapsed_time(b_) * 1e3), tiles.min(), tiles.max()))
        for t in np.unique(tiles):
            s = tiles == t
            print("   %d tiles: %3d workgroups, ends %.1f .. %.1f (mean %.1f)" % (t, s.sum(), end[s].min(), end[s].max(), end[s].mean()))
        print("   by XCD: mean end", ["%d: %.1f" % (x, end[xcc == x].mean()) for x in range(8)])
        print("   by XCD: shader cycles per tile", ["%d: %.0f" % (x, cyc[xcc == x].mean()) for x in range(8)])
        print("   idle at the end, summed over workgroups: %.2f %% of the launch's workgroup-time" % (100 * (end.max() - end).sum() / (256 * a.elapsed_time(b_) * 1e3)))


if __name__ == "__main__":
    main()
