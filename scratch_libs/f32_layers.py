import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("NERF_AMD_QUIET", "1")
from nerf_shared_amd import _lib, nerf, synth
ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
m = nerf.NeRF(**ARCH); m.load_state_dict(synth.torch_state_dict(0, 1.0, **{**ARCH, "skips": (4,)})); m = m.cuda(); m.precision = "fp32"
pts = torch.rand(32768, 64, 3, device="cuda") * 2 - 1
vd = torch.nn.functional.normalize(torch.randn(32768, 3, device="cuda"), dim=-1)
buf = (ctypes.c_ulonglong * 64)()
with torch.no_grad():
    m(pts, vd); torch.cuda.synchronize()
    _lib.lib.nerf_amd_debug_f32_stamps(buf)
    m(pts, vd); torch.cuda.synchronize()
    _lib.lib.nerf_amd_debug_f32_stamps(buf)
n = buf[15]
names = ["pts0 (K=63)", "pts1", "pts2", "pts3", "pts4", "pts5 (K=319)", "pts6", "pts7", "alpha (4x4x1)", "feature", "views (4 tiles, K=283)", "rgb (4x4x1)"]
tot = sum(buf[i] for i in range(12))
print("  zero-fill + encode       wave0 %8.0f" % (buf[13] / n))
for i, nm in enumerate(names):
    print("  %-24s wave0 %8.0f %5.1f %%   wave7 %8.0f" % (nm, buf[i] / n, 100.0 * buf[i] / tot, buf[16 + i] / n))
print("sum of layers per tile", tot / n)
