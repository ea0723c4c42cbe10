"""A/B of the exact-fp32 field kernel between two builds of the library in one session (interleaved rounds):
    python tools/micro/f32_ab.py scratch_libs/libold_f32.so
prints ms per launch of NeRF.forward on 4096 x 64 points for the in-tree build and the given one."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("NERF_AMD_QUIET", "1")
from nerf_shared_amd import nerf, synth
ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
m = nerf.NeRF(**ARCH); m.load_state_dict(synth.torch_state_dict(0, 1.0, **{**ARCH, "skips": (4,)})); m = m.cuda(); m.precision = "fp32"
pts = torch.rand(4096, 64, 3, device="cuda") * 2 - 1
vd = torch.nn.functional.normalize(torch.randn(4096, 3, device="cuda"), dim=-1)
with torch.no_grad():
    for _ in range(2): m(pts, vd)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): out = m(pts, vd)
    torch.cuda.synchronize()
print("%.3f ms  checksum %.6f" % ((time.perf_counter() - t) / 5 * 1e3, float(out.double().sum())))
'''
for r in range(3):
    for lib in [None] + sys.argv[1:]:
        env = dict(os.environ)
        if lib:
            env["NERF_AMD_LIB"] = os.path.abspath(lib)
        else:
            env.pop("NERF_AMD_LIB", None)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib or "in-tree", out.stdout.strip() or out.stderr[-300:])
