#!/usr/bin/env python3
"""Where a tile of the exact-fp32 kernel spends its cycles ("fp32 (exact) mode").

    python tools/experiments/f32_stamps.py build     # anywhere (hipcc cross-compiles): scratch_libs/libf32stamps.so
    python tools/experiments/f32_stamps.py run       # on the GPU box: per-layer cycles of wave 0 and wave 7

`build` instruments a COPY of csrc/mlp_fp32.hip by text substitution (s_memtime at the top of every layer of the layer loop,
sums of the sampled workgroups added to a device array, an extern "C" getter), compiles it and links it with the in-tree
objects of the other translation units (run `make -C nerf_shared_amd/csrc` first).  Nothing of this is in the shipping library.
"""
import ctypes
import glob
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "nerf_shared_amd", "csrc")
OUT = os.path.join(REPO, "scratch_libs")


def sub(s, old, new):
    assert old in s, "csrc/mlp_fp32.hip changed: %r not found" % old[:60]
    return s.replace(old, new, 1)


def build():
    os.makedirs(OUT, exist_ok=True)
    s = open(os.path.join(CSRC, "mlp_fp32.hip")).read()
    s = sub(s, "namespace na {\n", "namespace na {\n__device__ unsigned long long g_f32_stamps[64];\n")
    s = sub(s, "    for (int i = tid; i < rows * PTS; i += 512) act[i] = 0.0f;\n    __syncthreads();\n",
            "    unsigned long long t_prev = __builtin_amdgcn_s_memtime();\n"
            "    unsigned long long acc_t[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};\n"
            "    for (int i = tid; i < rows * PTS; i += 512) act[i] = 0.0f;\n    __syncthreads();\n")
    s = sub(s, "    for (int li = 0; li < a.n_layers; ++li) {\n        const LayerF32 L = a.layers[li];",
            "    { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t[13] += t - t_prev; t_prev = t; }\n"
            "    for (int li = 0; li < a.n_layers; ++li) {\n"
            "        { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (li > 0 && li <= 12) acc_t[li - 1] += t - t_prev; t_prev = t; }\n"
            "        const LayerF32 L = a.layers[li];")
    s = sub(s, "        __syncthreads();\n    }\n}\n\nint launch_mlp_f32",
            "        __syncthreads();\n    }\n"
            "    { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t[a.n_layers - 1] += t - t_prev; }\n"
            "    if (tid == 0 && blockIdx.x % 97 == 0) {\n"
            "        for (int i = 0; i < 14; ++i) atomicAdd(&g_f32_stamps[i], acc_t[i]);\n"
            "        atomicAdd(&g_f32_stamps[15], 1ull);\n    }\n"
            "    if (tid == 448 && blockIdx.x % 97 == 0)\n"
            "        for (int i = 0; i < 14; ++i) atomicAdd(&g_f32_stamps[16 + i], acc_t[i]);\n"
            "}\n\n"
            "extern \"C\" void nerf_amd_debug_f32_stamps(unsigned long long *out) {\n"
            "    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32_stamps), sizeof(unsigned long long) * 64);\n"
            "    unsigned long long z[64] = {0};\n"
            "    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f32_stamps), z, sizeof(z));\n}\n\nint launch_mlp_f32")
    src = os.path.join(CSRC, "_f32_stamps_tmp.hip")         # beside the headers it includes
    open(src, "w").write(s)
    try:
        obj = os.path.join(OUT, "mlp_fp32_stamps.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                        "-ffp-contract=off", "-c", src, "-o", obj], check=True)
    finally:
        os.remove(src)
    objs = [o for o in glob.glob(os.path.join(CSRC, "build", "*.o")) if not o.endswith("mlp_fp32.o")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libf32stamps.so")] + objs + [obj],
                   check=True)
    print("built", os.path.join(OUT, "libf32stamps.so"))


def run():
    os.environ["NERF_AMD_LIB"] = os.path.join(OUT, "libf32stamps.so")
    os.environ.setdefault("NERF_AMD_QUIET", "1")
    sys.path.insert(0, REPO)
    import torch
    from nerf_shared_amd import _lib, nerf, synth
    arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
    m = nerf.NeRF(**arch)
    m.load_state_dict(synth.torch_state_dict(0, 1.0, **{**arch, "skips": (4,)}))
    m = m.cuda()
    m.precision = "fp32"
    pts = torch.rand(32768, 64, 3, device="cuda") * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(32768, 3, device="cuda"), dim=-1)
    buf = (ctypes.c_ulonglong * 64)()
    with torch.no_grad():
        for _ in range(2):
            m(pts, vd)
            torch.cuda.synchronize()
            _lib.lib.nerf_amd_debug_f32_stamps(buf)
    n = buf[15]
    names = ["pts_linears.0 (K=63)", "pts_linears.1", ".2", ".3", ".4", ".5 (K=319)", ".6", ".7", "alpha_linear", "feature_linear",
             "views_linears.0 (4 tiles)", "rgb_linear"]
    tot = sum(buf[i] for i in range(12)) + buf[13]
    print("%d sampled workgroups, %.0f cycles per 64-point tile" % (n, tot / n))
    print("  %-28s wave0 %8.0f %5.1f %%" % ("zero-fill + encode", buf[13] / n, 100.0 * buf[13] / tot))
    for i, nm in enumerate(names):
        print("  %-28s wave0 %8.0f %5.1f %%   wave7 %8.0f" % (nm, buf[i] / n, 100.0 * buf[i] / tot, buf[16 + i] / n))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "build"]()
