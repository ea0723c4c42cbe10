"""Same-process A/B of the exact-fp32 field kernel's points per workgroup: 96 (three column halves, nerf_amd_set_tuning(0, 61))
against the default 64, interleaved rounds, NeRF.forward on 4096 x 64 points;
the outputs must be bit-identical (the per-point arithmetic does not depend on the workgroup shape)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NERF_AMD_QUIET", "1")
import torch  # noqa: E402

from nerf_shared_amd import _lib, nerf, synth  # noqa: E402

ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
m = nerf.NeRF(**ARCH)
m.load_state_dict(synth.torch_state_dict(1, 3.0, **{**ARCH, "skips": (4,)}))
m = m.cuda()
m.precision = "fp32"
pts = torch.rand(4096, 64, 3, device="cuda") * 2 - 1
vd = torch.nn.functional.normalize(torch.randn(4096, 3, device="cuda"), dim=-1)
outs = {}
with torch.no_grad():
    for r in range(4):
        for tag, v in (("96 points", 61), ("64 points", 0)):
            _lib.check(_lib.lib.nerf_amd_set_tuning(0, v), "tuning")
            for _ in range(2):
                m(pts, vd)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                out = m(pts, vd)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 5 * 1e3
            print("%s: %.3f ms = %.1f TFLOP/s" % (tag, ms, 4096 * 64 * 1186816 / ms / 1e9))
            outs[tag] = out
_lib.lib.nerf_amd_set_tuning(0, 0)
print("bit-identical:", bool(torch.equal(outs["96 points"], outs["64 points"])))
