#!/usr/bin/env python3
"""Diagnostic (needs a library built with EXTRA=-DNERF_AMD_STAMPS): where a wave's cycles go inside one
256-point tile of the fused bf16 field kernel -- tile start (coordinate loads, encoding, first weight block),
the layers, the end-of-tile drain.  Prints per-segment shader cycles per tile, median over waves."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
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
    R, S = 4096, 192
    pts = (torch.rand(R, S, 3, generator=g) * 6 - 3).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    out = {}
    for label, variant in (("pipelined", 0), ("per_tile", 41)):
        _lib.check(_lib.lib.nerf_amd_set_tuning(0, variant), "set_tuning")
        for _ in range(20):
            m(pts, vd)
        torch.cuda.synchronize()
        buf = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
        fn(buf.data_ptr())
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            m(pts, vd)
        b_.record()
        torch.cuda.synchronize()
        fn(None)
        b = buf.cpu().reshape(256, 8, 4).double()
        per_tile = b[..., :3] / b[..., 3:4]
        rec = {"ms_per_launch": a.elapsed_time(b_) / 10}
        for name, half in (("waves0-3", slice(0, 4)), ("waves4-7", slice(4, 8))):
            x = per_tile[:, half].reshape(-1, 3)
            med = x.median(0).values
            rec[name] = {"start": float(med[0]), "layers": float(med[1]), "drain": float(med[2]), "tile": float(med.sum())}
        rec["tiles_per_wave"] = float(b[..., 3].mean())
        # effective shader clock while the kernel runs: cycles a wave spent / wall time of the launches
        rec["approx_clock_ghz"] = float(b[..., :3].sum(-1).median() / 10) / (rec["ms_per_launch"] * 1e6)
        out[label] = rec
    _lib.lib.nerf_amd_set_tuning(0, 0)
    out["mfma_floor_cycles_per_tile"] = 2344 * 16 * 2
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
