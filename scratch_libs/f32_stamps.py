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
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); m(pts, vd); t1.record(); torch.cuda.synchronize()
    _lib.lib.nerf_amd_debug_f32_stamps(buf)
n = buf[7]
names = ["zero-fill", "encode", "layer compute (own)", "wait at barrier 1", "write-back", "wait at barrier 2"]
tot = sum(buf[i] for i in range(6))
print("launch %.2f ms, %d sampled workgroups, %.0f memtime ticks per tile (100 MHz ticks?)" % (t0.elapsed_time(t1), n, tot / n))
for i, nm in enumerate(names):
    print("  %-22s wave0 %10.0f  %5.1f %%   wave7 %10.0f" % (nm, buf[i] / n, 100.0 * buf[i] / tot, buf[8 + i] / n))
