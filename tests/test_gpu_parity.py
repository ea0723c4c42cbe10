"""GPU parity tests (-m gpu): the HIP path, called through the C ABI by the
Python drop-in, against (a) the golden vectors the reference produced and
(b) the CPU oracle on the same seeded inputs; plus size-independent
properties at the benchmark's full batch size.

Tolerances (fp32 inputs everywhere):
  * per-ray stages (z_vals, raw2outputs, sample_pdf, sort): 1e-5 abs on
    quantities of order 1 -- same arithmetic, libm-level differences only;
  * field MLP, fp32 mode (exact-fp32 MFMA, different summation order than the
    CPU GEMM): 1e-4 abs + 1e-4 rel on raw outputs of magnitude <= ~20;
  * field MLP, bf16 mode (bf16 operands, fp32 accumulate): relative L2 error
    <= 3e-2 on raw outputs, image PSNR >= 30 dB against the fp32 reference on
    the sharpened weights (the SURVEY's CPU bf16-autocast of the reference
    measured 39.8 dB there and 61 dB on default-scale weights).
"""
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("NERF_AMD_QUIET", "1")

pytestmark = pytest.mark.gpu

from nerf_shared_amd import synth  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VD = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
NOVD = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=False, multires=10, multires_views=4)
BASE = dict(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True,
            raw_noise_std=0.0, ndc=False, lindisp=False, near=2.0, far=6.0)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a ROCm device"
    return torch.device("cuda:0")


def amd():
    from nerf_shared_amd import nerf, render_utils, utils
    return nerf, render_utils, utils


def gpu_model(dev, seed, sharpen, precision, **arch):
    nerf, _, _ = amd()
    m = nerf.NeRF(**arch)
    m.load_state_dict(synth.torch_state_dict(seed, sharpen, **{**arch, "skips": tuple(arch["skips"])}))
    m.precision = precision
    return m.to(dev).requires_grad_(False)       # parity tests exercise the forward-only kernels


def cpu_model(seed, sharpen, **arch):
    return O.state_dict_to_torch(synth.make_state_dict(seed, sharpen, **{**arch, "skips": tuple(arch["skips"])})), O.Arch(**arch)


def close(a, b, atol, rtol=0.0):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(np.nan_to_num(a), np.nan_to_num(b), atol=atol, rtol=rtol)


def close_frac(a, b, atol, rtol=0.0, frac=0.97, mask=None):
    """At least `frac` of the (masked) elements agree within atol + rtol*|b|; NaNs must coincide.
    Used where the reference itself is ill-conditioned: sample_pdf divides a ~1e-7 cdf
    rounding difference by bin masses down to 1e-5 (utils.py:110-112), which moves a few
    percent of the fine samples by up to ~1e-3 and everything downstream of them."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    ok = np.abs(np.nan_to_num(a) - np.nan_to_num(b)) <= atol + rtol * np.abs(np.nan_to_num(b))
    if mask is not None:
        ok = ok[mask]
    assert ok.size == 0 or ok.mean() >= frac, "only %.4f of elements within tolerance (need %.2f)" % (ok.mean(), frac)


def close_disp(a, b, acc, n_samples, atol, rtol, raw_tol=0.0, far=None):
    """disp = 1 / max(1e-10, depth / acc) (render_utils.py:284) is a ratio of two sums of the same weights, and every
    weight carries the absolute error of 1 - exp(-x) near x = 0 (an ulp of 1.0, whatever libm or the device computes): the
    ratio's relative error is about 2 * n_samples * eps / acc.  For an opaque ray that is nothing; for a nearly empty one
    (acc 1e-3) it is percents -- in the reference too.  Tolerance = the stage's own + that conditioning term.
    raw_tol: when the two sides composite DIFFERENT raw values (end-to-end comparisons: the field outputs agree to
    raw_tol), the weights move by about raw_tol in absolute terms and the ratio by raw_tol / acc.  far: the largest depth --
    the depth sum's absolute error is that of the weights times up to `far`, so a ray whose mass sits at tiny depths (NDC:
    disp in the thousands) carries far * disp / acc of relative error on top."""
    a, b, acc = (t.detach().cpu().double().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, np.float64) for t in (a, b, acc))
    assert (np.isnan(a) == np.isnan(b)).all()
    ok = ~np.isnan(b)
    cond = (4.0 * n_samples * 1.2e-7 + raw_tol) / np.maximum(np.abs(acc[ok]), 1e-30)
    if far is not None:
        cond = cond * (1.0 + far * np.abs(b[ok]))
    tol = atol + np.abs(b[ok]) * (rtol + cond)
    bad = np.abs(a[ok] - b[ok]) > tol
    assert not bad.any(), ("disp", int(bad.sum()), float(np.abs(a[ok] - b[ok])[bad].max()), float(acc[ok][bad].min()))


def pdf_denominators(bins, weights, u):
    """cdf[above] - cdf[below] of every sample (the conditioning of utils.py:110-113)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    idx = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo, hi = torch.clamp(idx - 1, min=0), torch.clamp(idx, max=cdf.shape[-1] - 1)
    return (torch.gather(cdf, -1, hi) - torch.gather(cdf, -1, lo)).numpy()


def rel_l2(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def psnr(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float(-10.0 * torch.log10(torch.mean((a - b) ** 2) + 1e-30))


def report(name, values):
    """Measured parity figures go to gpurun_out/parity_<name>.json (merged back by gpurun) so the gates
    in this file can be checked against what the kernels achieve."""
    print("parity report %s: %s" % (name, values))
    if os.path.isdir("gpurun_out"):
        import json
        with open(os.path.join("gpurun_out", "parity_%s.json" % name), "w") as f:
            json.dump(values, f, indent=1, sort_keys=True)


def test_native_library_loaded(dev):
    from nerf_shared_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libnerf_amd.so" in f.read()



# Gates of the per-ray attribution (tools/precision_census.py attribute): what may move against the reference end to end
# is what the reference's own conditioning moves -- rays with a displaced fine sample or a flip-prone last sample; the
# fraction of rays beyond 1e-3 is reported and capped at twice what the x3-sharpened content fields measure (3-4 %).
ATTR_FRAC_BIG = 0.08


def attributed_close(cfg, batch_cpu, coarse_cpu, fine_cpu, got, ref_maps=None, pytest_flag=False, label="", n_samples=None):
    """End-to-end comparison of a two-pass render with the reference WITHOUT a fraction criterion: every ray's raw / maps are
    tight against the oracle evaluated on the ray's own depths, and every ray that moved against the reference (ref_maps: a
    golden fixture's rgb_map / acc_map [/ z_vals]; default: the oracle's own end-to-end maps) carries a displaced fine
    sample or a flip-prone last sample -- unexplained rays = 0.  got: render_rays(..., retraw=True, retweights=True) on CPU."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(REPO, "tools"))
    import precision_census as PC
    ocfg = O.RenderCfg(**cfg)
    with torch.no_grad():
        ref = PC.oracle_stages(ocfg, batch_cpu, coarse_cpu, fine_cpu, pytest=pytest_flag)
    maps = None
    if ref_maps is not None:
        maps = {"rgb_map": ref_maps["rgb_map"], "acc_map": ref_maps["acc_map"], "z_vals": ref_maps.get("z_vals", ref["z_vals"])}
    att = PC.attribute(ocfg, batch_cpu, coarse_cpu, fine_cpu, got, ref=ref, pytest=pytest_flag, ref_maps=maps)
    summary = {k: v for k, v in att.items() if not k.startswith("_")}
    assert att["unexplained"] == 0, (label, att["unexplained_rays"], summary)      # (ray, how far it moved, its worst sample's displacement in ulp)
    assert att["staged_raw_max"] < 2e-4 and att["staged_rgb_max"] < 1e-5 and att["staged_acc_max"] < 1e-5, (label, summary)
    assert att["frac_gt_1e-3"] <= ATTR_FRAC_BIG, (label, summary)
    close(got["weights"], att["_staged_weights"], atol=1e-6, rtol=1e-4)
    close_disp(got["disp_map"], att["_staged_disp"], att["_staged_acc"], n_samples or got["z_vals"].shape[-1], atol=1e-5, rtol=1e-4)
    return summary

# ------------------------------------------------------------------ G1
def test_embedder_golden(dev, golden):
    nerf, _, _ = amd()
    g = golden("g1_embedder")
    x = torch.from_numpy(g["x"]).to(dev)
    for L in (10, 4, 15, 6):
        fn, dim = nerf.get_embedder(L, 0)
        assert dim == 3 + 6 * L
        close(fn(x), g["L%d" % L], atol=2e-6)
    fn, dim = nerf.get_embedder(10, -1)
    assert dim == 3
    close(fn(x), g["identity"], atol=0)
    # leading dims are preserved
    fn, _ = nerf.get_embedder(4, 0)
    assert fn(x.reshape(8, 8, 3)).shape == (8, 8, 27)


# ------------------------------------------------------------------ G2
@pytest.mark.parametrize("tag,seed,sharpen", [("s0", 0, 1.0), ("s1", 1, 3.0)])
def test_nerf_forward_fp32_golden(dev, golden, tag, seed, sharpen):
    g = golden("g2_nerf")
    pts, vd = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    m = gpu_model(dev, seed, sharpen, "fp32", **VD)
    close(m(pts, vd), g["vd_" + tag], atol=1e-4, rtol=1e-4)
    close(m.get_density(pts), g["density_" + tag], atol=1e-4, rtol=1e-4)
    m2 = gpu_model(dev, seed, sharpen, "fp32", **NOVD)
    out = m2(pts, None)
    assert out.shape == (32, 8, 5)
    close(out, g["novd_" + tag], atol=1e-4, rtol=1e-4)


def test_nerf_forward_fp32_other_archs(dev, golden):
    g = golden("g2_nerf")
    pts, vd = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    small = dict(D=4, W=128, output_ch=4, skips=[1], use_viewdirs=True, multires=6, multires_views=2)
    close(gpu_model(dev, 5, 3.0, "fp32", **small)(pts, vd), g["small_s1"], atol=1e-4, rtol=1e-4)
    wide = dict(VD, multires=15, multires_views=6)
    close(gpu_model(dev, 6, 3.0, "fp32", **wide)(pts, vd), g["wide_s1"], atol=2e-4, rtol=2e-4)
    # bf16 requested on an architecture the fused kernel does not cover: still HIP, fp32 rate
    close(gpu_model(dev, 5, 3.0, "bf16", **small)(pts, vd), g["small_s1"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("arch", [dict(D=3, W=100, output_ch=4, skips=[1], use_viewdirs=True, multires=5, multires_views=3),
                                  dict(D=5, W=37, output_ch=7, skips=[2], use_viewdirs=False, multires=4, multires_views=4),
                                  dict(D=2, W=250, output_ch=4, skips=[], use_viewdirs=True, multires=10, multires_views=4)],
                         ids=["W100", "W37_novd", "W250"])
def test_nerf_forward_any_width(dev, arch):
    """The reference constructor takes any width (nerf.py:62-94: W and W // 2 are plain nn.Linear sizes): widths that are
    not multiples of the fp32 kernel's 32-row tiles / 8-column groups against the oracle, view branch with an odd W // 2
    included.  (bf16 / split precision requested on such a model run on the exact kernel.)"""
    rng = np.random.default_rng(4)
    pts = torch.from_numpy(rng.uniform(-2, 2, size=(41, 7, 3)).astype(np.float32))
    vd = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(41, 3)).astype(np.float32)), dim=-1) if arch["use_viewdirs"] else None
    ref = O.nerf_forward(*cpu_model(9, 2.0, **arch), pts, vd)
    for prec in ("fp32", "bf16", "fp32_split"):
        out = gpu_model(dev, 9, 2.0, prec, **arch)(pts.to(dev), vd.to(dev) if vd is not None else None)
        close(out, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("tag,seed,sharpen", [("s0", 0, 1.0), ("s1", 1, 3.0)])
def test_nerf_forward_bf16_golden(dev, golden, tag, seed, sharpen):
    g = golden("g2_nerf")
    pts, vd = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    m = gpu_model(dev, seed, sharpen, "bf16", **VD)
    assert m.supports_bf16()
    assert rel_l2(m(pts, vd), g["vd_" + tag]) < 3e-2
    m2 = gpu_model(dev, seed, sharpen, "bf16", **NOVD)
    assert rel_l2(m2(pts, None), g["novd_" + tag]) < 3e-2
    wide = dict(VD, multires=15, multires_views=6)
    if tag == "s1":
        assert rel_l2(gpu_model(dev, 6, 3.0, "bf16", **wide)(pts, vd), g["wide_s1"]) < 3e-2


def test_mlp_on_embedded_rows(dev, golden):
    """NeRF.MLP(x) (nerf.py:110-134) on rows embedded by the HIP embedder == forward()."""
    nerf, _, _ = amd()
    g = golden("g2_nerf")
    pts, vd = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    m = gpu_model(dev, 1, 3.0, "fp32", **VD)
    e = m.embed_fn(pts.reshape(-1, 3))
    ed = m.embeddirs_fn(vd[:, None].expand(pts.shape).reshape(-1, 3))
    out = m.MLP(torch.cat([e, ed], -1))
    assert out.shape == (256, 4)
    close(out.reshape(32, 8, 4), g["vd_s1"], atol=1e-4, rtol=1e-4)
    m2 = gpu_model(dev, 1, 3.0, "fp32", **NOVD)
    close(m2.MLP(m2.embed_fn(pts.reshape(-1, 3))).reshape(32, 8, 5), g["novd_s1"], atol=1e-4, rtol=1e-4)


def test_nerf_forward_ragged_and_large(dev, golden):
    """Point counts that are not multiples of the 256-point workgroup tile, and the
    reference's > netchunk case (70400 points, strided subset pinned by the golden)."""
    g = golden("g2_nerf")
    rng2 = np.random.default_rng(203)
    big = torch.from_numpy(rng2.uniform(-3, 3, size=(1100, 64, 3)).astype(np.float32)).to(dev)
    bvd = rng2.normal(size=(1100, 3)).astype(np.float32)
    bvd /= np.linalg.norm(bvd, axis=-1, keepdims=True)
    bvd = torch.from_numpy(bvd).to(dev)
    stride = int(g["big_stride"])
    m32 = gpu_model(dev, 1, 3.0, "fp32", **VD)
    full32 = m32(big, bvd)
    close(full32.reshape(-1, 4)[::stride], g["big_subset"], atol=1e-4, rtol=1e-4)
    mbf = gpu_model(dev, 1, 3.0, "bf16", **VD)
    fullbf = mbf(big, bvd)
    assert rel_l2(fullbf.reshape(-1, 4)[::stride], g["big_subset"]) < 3e-2
    # ragged: 37 rays x 5 samples = 185 points; results must equal the same rays inside the big batch
    sub, subvd = big[:37, :5].contiguous(), bvd[:37].contiguous()
    close(m32(sub, subvd), full32[:37, :5], atol=0)
    close(mbf(sub, subvd), fullbf[:37, :5], atol=0)
    # a single point
    one = mbf(big[:1, :1].contiguous(), bvd[:1].contiguous())
    close(one, fullbf[:1, :1], atol=0)


def test_pipelined_kernel_equals_per_tile_kernel_on_many_shapes(dev):
    """The default inference kernel carries its weight stream and the next tile's encoding across tile boundaries
    (continuous ring: run-time choice of one sync's vmcnt count, next tile's first blocks fetched early, slot table
    rotated per tile).  Its outputs must equal the simple per-tile kernel's (nerf_amd_set_tuning(0, 41)) bit for bit for
    every shape of launch: fewer tiles than CUs, exactly one / two / three tiles per workgroup with and without a ragged
    last tile, one point, both input modes (explicit points, rays + depths), both encodings, with and without view branch."""
    nerf, _, _ = amd()
    from nerf_shared_amd import _lib
    rng = np.random.default_rng(77)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    tile = 256
    sizes = [1, 31, 255, 256, 257, 1000, tile * n_cu - 1, tile * n_cu, tile * n_cu + 1, tile * n_cu + 300,
             2 * tile * n_cu - 7, 2 * tile * n_cu, 2 * tile * n_cu + 513, 3 * tile * n_cu + 1, 5 * tile * n_cu - 255]
    archs = [(VD, 1), (dict(VD, multires=15, multires_views=6), 6), (NOVD, 2), (dict(NOVD, multires=15), 2)]
    try:
        for arch, seed in archs:
            m = gpu_model(dev, seed, 3.0, "bf16", **arch)
            vd_on = arch["use_viewdirs"]
            for P in sizes:
                S = int(rng.choice([1, 3, 64, 192, 128]))
                R = -(-P // S)
                pts = torch.from_numpy(rng.uniform(-3, 3, size=(R, S, 3)).astype(np.float32)).to(dev)
                vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1) if vd_on else None
                outs = []
                for variant in (0, 41):
                    _lib.check(_lib.lib.nerf_amd_set_tuning(0, variant), "set_tuning")
                    outs.append(m(pts, vd).clone())
                torch.cuda.synchronize()
                assert torch.equal(outs[0], outs[1]), (arch["multires"], vd_on, "points", R, S)
        # rays + depths mode (what render_rays launches), view-branch model
        _, render_utils, utils = amd()
        K = synth.lego_intrinsics(400, 400)
        c, f = gpu_model(dev, 1, 3.0, "bf16", **VD), gpu_model(dev, 19, 3.0, "bf16", **VD)
        r = render_utils.Renderer(**BASE)
        for n in (1, 5, 341, 1024, 1365, 4096, 4097):
            batch = utils.make_ray_batch(400, 400, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, pix0=70000, n=n)
            outs = []
            for variant in (0, 41):
                _lib.check(_lib.lib.nerf_amd_set_tuning(0, variant), "set_tuning")
                outs.append({k: v.clone() for k, v in r.render_rays(batch, c, f, retraw=True).items()})
            torch.cuda.synchronize()
            for k in outs[0]:
                assert torch.equal(torch.nan_to_num(outs[0][k]), torch.nan_to_num(outs[1][k])), ("rays", n, k)
    finally:
        _lib.lib.nerf_amd_set_tuning(0, 0)


def test_weight_update_repacks(dev, golden):
    g = golden("g2_nerf")
    pts, vd = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    m = gpu_model(dev, 0, 1.0, "fp32", **VD)
    close(m(pts, vd), g["vd_s0"], atol=1e-4, rtol=1e-4)
    m.load_state_dict({k: v.to(dev) for k, v in synth.torch_state_dict(1, 3.0, **{**VD, "skips": (4,)}).items()})
    close(m(pts, vd), g["vd_s1"], atol=1e-4, rtol=1e-4)


def test_packed_copies_follow_the_weights_and_the_precision(dev, golden):
    """Only the packed copies a call needs are refreshed (nerf_amd_model_update_copies): switching the precision of a model
    adds the missing copy for the SAME weights; changing the weights makes every copy stale, whichever precision asks next;
    a training step in one precision followed by a render in another sees the stepped weights; and the C ABI refuses a
    copy that is stale or was never made."""
    import ctypes
    nerf, render_utils, utils = amd()
    from nerf_shared_amd import _lib
    g = golden("g2_nerf")
    pts, vd = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    m = gpu_model(dev, 0, 1.0, "bf16", **VD)
    fresh = {p: gpu_model(dev, 0, 1.0, p, **VD)(pts, vd) for p in ("bf16", "fp32_split", "fp32")}
    for p in ("bf16", "fp32_split", "fp32", "bf16"):       # one model, a copy added per precision, no weight change
        m.precision = p
        assert torch.equal(m(pts, vd), fresh[p]), p
    assert m._packed_copies == _lib.COPY_BF16 | _lib.COPY_SPLIT | _lib.COPY_FP32
    m.load_state_dict({k: v.to(dev) for k, v in synth.torch_state_dict(1, 3.0, **{**VD, "skips": (4,)}).items()})
    m.precision = "fp32"                                    # the weights changed: the next call packs ITS copy only ...
    close(m(pts, vd), g["vd_s1"], atol=1e-4, rtol=1e-4)
    assert m._packed_copies == _lib.COPY_FP32
    m.precision = "fp32_split"                              # ... and another precision gets the new weights too
    close(m(pts, vd), g["vd_s1"], atol=1e-4, rtol=1e-4)
    # a bf16 training step, then the same model rendered in split precision: the stepped weights, not the packed-before ones
    t = gpu_model(dev, 0, 1.0, "bf16", **VD).requires_grad_(True)
    opt = torch.optim.SGD(t.parameters(), lr=1e-2)
    t(pts, vd).square().mean().backward()
    assert t._packed_copies == _lib.COPY_BF16 | _lib.COPY_BWD
    opt.step()
    t.precision = "fp32_split"
    with torch.no_grad():
        after = t(pts, vd)
    ref = gpu_model(dev, 0, 1.0, "fp32_split", **VD)
    ref.load_state_dict(t.state_dict())
    assert torch.equal(after, ref(pts, vd))
    # the C ABI: a copy that was not packed is refused, loudly
    h = m._handle
    params = [mod._parameters[k] for k in ("weight", "bias") for mod in m._linears()]
    n = len(params) // 2
    wp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in params[:n]])
    bp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in params[n:]])
    lib = _lib.lib
    assert lib.nerf_amd_model_update_copies(h, wp, bp, n, _lib.COPY_BF16, 0, _lib.stream_of(dev)) == 0
    out = torch.empty(pts.shape[0] * pts.shape[1], 4, device=dev)
    args = (h, pts.data_ptr(), vd.data_ptr(), pts.shape[0], pts.shape[1], out.data_ptr())
    assert lib.nerf_amd_nerf_forward(*args, _lib.PREC_BF16, _lib.stream_of(dev)) == 0
    rc = lib.nerf_amd_nerf_forward(*args, _lib.PREC_FP32, _lib.stream_of(dev))
    assert rc != 0 and b"stale or was never made" in lib.nerf_amd_last_error()
    assert lib.nerf_amd_model_update_copies(h, wp, bp, n, 64, 0, _lib.stream_of(dev)) != 0      # unknown bit
    m.weights_changed()                                     # hand the handle back to the shim in a known state


# ------------------------------------------------------------------ G3
def test_raw2outputs_golden(dev, golden):
    _, render_utils, _ = amd()
    g = golden("g3_raw2outputs")
    raw, z, rd = (torch.from_numpy(g[k]).to(dev) for k in ("raw", "z_vals", "rays_d"))
    names = ("rgb", "disp", "acc", "weights", "depth")
    for white in (True, False):
        r = render_utils.Renderer(perturb=0.0, white_bkgd=white, raw_noise_std=0.0)
        for n, v in zip(names, r.raw2outputs(raw, z, rd)):
            close(v, g["%s_white%d" % (n, white)], atol=2e-6, rtol=2e-5)
    r = render_utils.Renderer(perturb=0.0, white_bkgd=True, raw_noise_std=1.0)
    for n, v in zip(names, r.raw2outputs(raw, z, rd, pytest=True)):
        close(v, g["%s_noise" % n], atol=2e-6, rtol=2e-5)
    r = render_utils.Renderer(perturb=0.0, white_bkgd=True, raw_noise_std=0.0)
    res = r.raw2outputs(*(torch.from_numpy(g[k]).to(dev) for k in ("raw_192", "z_vals_192", "rays_d_192")))
    for n, v in zip(names, res):
        close(v, g["%s_192" % n], atol=2e-6, rtol=2e-5)
    # 5-channel raw (no-viewdirs models emit output_ch=5; the 5th channel is ignored)
    raw5 = torch.cat([raw, torch.randn_like(raw[..., :1])], -1)
    for n, v in zip(names, r.raw2outputs(raw5, z, rd)):
        close(v, g["%s_white1" % n], atol=2e-6, rtol=2e-5)


# ------------------------------------------------------------------ G4
def test_sample_pdf_golden(dev, golden):
    _, _, utils = amd()
    g = golden("g4_sample_pdf")
    bins, w = torch.from_numpy(g["bins"]).to(dev), torch.from_numpy(g["weights"]).to(dev)
    bc, wc = torch.from_numpy(g["bins"]), torch.from_numpy(g["weights"])
    for N in (64, 128):
        for key, det, pt in (("det_N%d", True, False), ("detpytest_N%d", True, True), ("rand_N%d", False, True)):
            got = utils.sample_pdf(bins, w, N, det=det, pytest=pt)
            u = O.pytest_u_for_sample_pdf(128, N, det) if pt else torch.linspace(0., 1., N).expand(128, N)
            well = pdf_denominators(bc, wc, u) > 1e-3        # error <= 1e-7 / 1e-3 of a bin width
            assert well.mean() > 0.9
            close_frac(got, g[key % N], atol=1e-5, frac=1.0, mask=well)
            close(got, g[key % N], atol=2e-3)                 # ill-conditioned rest: still inside the bin
    known = utils.sample_pdf(torch.from_numpy(g["known_bins"]).to(dev), torch.from_numpy(g["known_weights"]).to(dev), 8, det=True)
    close(known, g["known_det8"], atol=1e-6)
    # random draws: range and monotone dependence on u are the testable properties
    s = utils.sample_pdf(bins, w, 128, det=False)
    assert s.shape == (128, 128)
    assert bool((s >= bins[:, :1] - 1e-6).all()) and bool((s <= bins[:, -1:] + 1e-6).all())


# ------------------------------------------------------------------ G5
G5_CASES = {
    "det_s0": (dict(), VD, (0, 10, 1.0), False),
    "det_s1": (dict(), VD, (1, 11, 3.0), False),
    "perturb_s1": (dict(perturb=1.0), VD, (1, 11, 3.0), True),
    "lindisp_s1": (dict(lindisp=True), VD, (1, 11, 3.0), False),
    "coarseonly_s1": (dict(N_importance=0), VD, (1, None, 3.0), False),
    "nofine_s1": (dict(), VD, (1, None, 3.0), False),
    "black_noise_s1": (dict(white_bkgd=False, raw_noise_std=1.0, perturb=1.0), VD, (1, 11, 3.0), True),
    "novd_s1": (dict(use_viewdirs=False), NOVD, (2, 12, 3.0), False),
    "fern_s1": (dict(N_importance=64, ndc=True, near=0.0, far=1.0, white_bkgd=False,
                     raw_noise_std=1.0, perturb=1.0), VD, (1, 11, 3.0), True),
    # fine fields with content (seed 11's fine pass is empty space)
    "det_c19": (dict(), VD, (1, 19, 3.0), False),
    "perturb_c12": (dict(perturb=1.0), VD, (1, 12, 3.0), True),
    "fern_c12": (dict(N_importance=64, ndc=True, near=0.0, far=1.0, white_bkgd=False,
                      raw_noise_std=1.0, perturb=1.0), VD, (1, 12, 3.0), True),
}
G5_KEYS = ("rgb_map", "disp_map", "acc_map", "raw", "weights", "z_vals", "rgb0", "disp0", "acc0", "z_std")
G5_TOL = {"rgb_map": 2e-4, "acc_map": 2e-4, "rgb0": 2e-4, "acc0": 2e-4, "weights": 2e-4,
          "raw": 2e-4, "z_vals": 2e-5, "z_std": 2e-5, "disp_map": 2e-4, "disp0": 2e-4}


@pytest.mark.parametrize("tag", sorted(G5_CASES))
def test_render_rays_fp32_golden(dev, golden, tag):
    render_rays_golden_check(dev, golden, tag, "fp32")


def render_rays_golden_check(dev, golden, tag, precision):
    """The fp32 gates of render_rays against the reference's own outputs (golden G5), for a precision mode that
    claims fp32-class results ("fp32": the exact kernel; "fp32_split": tests/test_gpu_split.py)."""
    _, render_utils, _ = amd()
    g = golden("g5_render_rays")
    over, arch, (sc, sf, sharpen), pytest_flag = G5_CASES[tag]
    r = render_utils.Renderer(**dict(BASE, **over))
    coarse = gpu_model(dev, sc, sharpen, precision, **arch)
    fine = gpu_model(dev, sf, sharpen, precision, **arch) if sf is not None else None
    ret = r.render_rays(torch.from_numpy(g[tag + "__batch"]).to(dev), coarse, fine,
                        retraw=True, retweights=True, pytest=pytest_flag)
    expected = [k for k in G5_KEYS if tag + "__" + k in g]
    assert list(ret.keys()) == [k for k in ("rgb_map", "disp_map", "acc_map", "raw", "weights", "z_vals",
                                            "rgb0", "disp0", "acc0", "z_std") if k in expected]
    resampled = "rgb0" in expected
    matched = None
    if resampled:
        dz = np.abs(ret["z_vals"].cpu().numpy() - g[tag + "__z_vals"])
        assert (dz < 2e-5).mean() > 0.9, (dz < 2e-5).mean()
        assert dz.max() < 5e-3
        matched = dz < 1e-6           # (about 2-4 ulp of z) samples whose position agrees: compare those per sample
        assert matched.mean() > 0.5, matched.mean()
    for k in expected:
        ref = g[tag + "__" + k]
        if not resampled or k in ("rgb0", "disp0", "acc0"):
            close(ret[k], ref, atol=G5_TOL[k], rtol=2e-4)             # no resampling upstream: tight
        elif k == "weights":
            # samples whose depth agrees with the reference's to a few ulp: their weights agree too
            got_w = ret[k].cpu().numpy()[matched]
            bad = np.abs(got_w - ref[matched]) > 2e-3 + 2e-4 * np.abs(ref[matched])
            assert bad.mean() < 0.03, (tag, float(bad.mean()))
    if resampled:
        # everything downstream of the resampling, ray by ray (no fraction waved through): tools/precision_census.py
        batch_cpu = torch.from_numpy(g[tag + "__batch"])
        coarse_cpu = cpu_model(sc, sharpen, **arch)
        fine_cpu = cpu_model(sf, sharpen, **arch) if sf is not None else None
        ref_maps = {k: torch.from_numpy(g[tag + "__" + k]) for k in ("rgb_map", "acc_map", "z_vals")}
        att = attributed_close(dict(BASE, **over), batch_cpu, coarse_cpu, fine_cpu, {k: v.cpu() for k, v in ret.items()},
                               ref_maps=ref_maps, pytest_flag=pytest_flag, label=tag + "/" + precision)
        report("attribution_%s_%s" % (precision, tag), att)
        close(ret["rgb_map"], g[tag + "__rgb_map"], atol=0.15)        # and nothing moves by more than a last-sample flip can
    # without retraw / retweights those keys are absent
    ret2 = r.render_rays(torch.from_numpy(g[tag + "__batch"]).to(dev), coarse, fine, pytest=pytest_flag)
    assert "raw" not in ret2 and "weights" not in ret2 and "z_vals" not in ret2
    close(ret2["rgb_map"], ret["rgb_map"], atol=0)


# (median |err| gate, fraction of rays off by > 0.1, PSNR gate in dB) per case, set a little under what the
# kernel measures (gpurun_out/parity_bf16_golden_*.json; DESIGN.md section 2 lists the measured values)
BF16_GATES = {
    # measured (rgb0 / rgb_map):    median |err|          rays off by > 0.1    PSNR
    "det_s0": (5e-4, 0.0, 75.0),    # 7.5e-5 / 0          0 / 0                85.7 / 300 (fine pass: empty space)
    "det_s1": (5e-3, 0.0, 50.0),    # 1.5e-3 / 0          0 / 0                56.4 / 300 (fine pass: empty space)
    "novd_s1": (2e-2, 0.0, 38.0),   # 1.4e-3 / 6.2e-3     0 / 0                57.1 / 42.3
    "fern_s1": (3e-3, 0.03, 25.0),  # 6.4e-4 / 0          1.0 % / 0            28.1 / 73.4 (sigma noise + sign flips of the last sample)
    # fine passes with content:
    "det_c19": (2e-2, 0.0, 36.0),       # 1.5e-3 / 5.6e-3   0 / 0               56.4 / 39.8 (opaque)
    "perturb_c12": (2.5e-2, 0.10, 27.0),  # 8.7e-4 / 8.9e-3   0 / 6.3 %           56.1 / 30.1 (semi-transparent, random u)
    "fern_c12": (2e-2, 0.03, 25.0),     # 6.4e-4 / 5.6e-3   1.0 % / 0           28.1 / 41.0
}


@pytest.mark.parametrize("tag", sorted(BF16_GATES))
def test_render_rays_bf16_golden(dev, golden, tag):
    """bf16 mode: depth samples are a discontinuous function of the coarse weights
    only through searchsorted, so compare maps, not per-sample raw."""
    _, render_utils, _ = amd()
    g = golden("g5_render_rays")
    over, arch, (sc, sf, sharpen), pytest_flag = G5_CASES[tag]
    r = render_utils.Renderer(**dict(BASE, **over))
    coarse = gpu_model(dev, sc, sharpen, "bf16", **arch)
    fine = gpu_model(dev, sf, sharpen, "bf16", **arch)
    ret = r.render_rays(torch.from_numpy(g[tag + "__batch"]).to(dev), coarse, fine, pytest=pytest_flag)
    med_gate, frac_gate, psnr_gate = BF16_GATES[tag]
    measured = {}
    for k in ("rgb_map", "rgb0"):
        a, b = ret[k].cpu().numpy(), g[tag + "__" + k]
        # the last sample's alpha is a step function of the sign of sigma (dists[-1] = 1e10,
        # render_utils.py:257): with sigma ~ 0 a bf16 rounding flips whole rays, so bound the
        # median and the fraction of rays that moved, and the PSNR of the map, instead of the max
        err = np.abs(a - b).max(-1)
        measured[k] = dict(median=float(np.median(err)), frac_gt_0p1=float((err > 0.1).mean()), psnr=psnr(a, b))
    report("bf16_golden_" + tag, measured)
    for k, m in measured.items():
        assert m["median"] < med_gate, (k, m)
        assert m["frac_gt_0p1"] <= frac_gate, (k, m)
        assert m["psnr"] > psnr_gate, (k, m)


# ------------------------------------------------------------------ G6
def test_rays_golden(dev, golden):
    _, _, utils = amd()
    g = golden("g6_rays")
    c2w = torch.from_numpy(synth.LEGO_C2W)
    ro, rd = utils.get_rays(4, 6, g["small_K"], c2w)
    assert ro.shape == (4, 6, 3) and ro.is_cuda
    close(ro, g["small_rays_o"], atol=0)
    close(rd, g["small_rays_d"], atol=2e-7, rtol=2e-7)
    c2w4 = torch.eye(4)
    c2w4[:3, :4] = c2w
    ro, rd = utils.get_rays(4, 6, g["small_K"], c2w4.to(dev))
    close(rd, g["small4_rays_d"], atol=2e-7, rtol=2e-7)
    K = synth.lego_intrinsics(400, 400)
    ro, rd = utils.get_rays(400, 400, K, c2w)
    close(rd.reshape(-1, 3)[torch.from_numpy(g["lego_corners"]).to(dev)], g["lego_rays_d"], atol=2e-7, rtol=2e-7)
    H, W, focal = g["ndc_HWf"]
    o2, d2 = utils.ndc_rays(int(H), int(W), float(focal), 1.0, torch.from_numpy(g["ndc_in_o"]).to(dev),
                            torch.from_numpy(g["ndc_in_d"]).to(dev))
    close(o2, g["ndc_out_o"], atol=1e-6, rtol=1e-6)
    close(d2, g["ndc_out_d"], atol=1e-6, rtol=1e-6)
    rn_o, rn_d = utils.get_rays_np(4, 6, g["small_K"], synth.LEGO_C2W)
    close(rn_d, g["small_rays_d_np"], atol=1e-12)


# ------------------------------------------------------------------ G7
def render_attribution(dev, rend, cfg, H, W, K, c2w, coarse, fine, seeds, ref_rgb, ref_acc, label, rays=None, c2w_staticcam=None, arch=None):
    """The end-to-end maps of Renderer.render against the reference's, ray by ray (attributed_close): the render's own ray
    batch goes through render_rays once more for its depths and raw (bit-identical maps: chunk invariance)."""
    _, _, utils = amd()
    (sc, sf, sharpen) = seeds
    arch = arch or VD
    if rays is None:
        batch = utils.make_ray_batch(H, W, K, c2w, cfg["near"], cfg["far"], cfg["use_viewdirs"], cfg["ndc"], c2w_staticcam=c2w_staticcam,
                                     device=dev)
    else:
        ro, rd = rays[0].reshape(-1, 3).to(dev), rays[1].reshape(-1, 3).to(dev)     # Renderer.render's own sequence (render_utils.py:205-226)
        vdir = rd / torch.norm(rd, dim=-1, keepdim=True)
        if cfg["ndc"]:
            ro, rd = utils.ndc_rays(H, W, K[0][0], 1., ro, rd)
        parts = [ro, rd, cfg["near"] * torch.ones_like(rd[:, :1]), cfg["far"] * torch.ones_like(rd[:, :1])]
        if cfg["use_viewdirs"]:
            parts.append(vdir)
        batch = torch.cat(parts, -1).contiguous().float()
    got = {k: v.cpu() for k, v in rend.render_rays(batch, coarse, fine, retraw=True, retweights=True).items()}
    ref_maps = {"rgb_map": torch.as_tensor(ref_rgb).reshape(-1, 3), "acc_map": torch.as_tensor(ref_acc).reshape(-1)}
    att = attributed_close(cfg, batch.cpu(), cpu_model(sc, sharpen, **arch), cpu_model(sf, sharpen, **arch), got, ref_maps=ref_maps, label=label)
    return got, att


def test_render_golden(dev, golden):
    _, render_utils, _ = amd()
    g = golden("g7_render")
    r = render_utils.Renderer(**BASE)
    coarse, fine = gpu_model(dev, 1, 3.0, "fp32", **VD), gpu_model(dev, 11, 3.0, "fp32", **VD)
    seeds = (1, 11, 3.0)
    c2w = torch.from_numpy(g["c2w"])
    rgb, disp, acc, extras = r.render(16, 16, g["K"], coarse, fine, chunk=100, c2w=c2w, retraw=True)
    assert rgb.shape == (16, 16, 3) and disp.shape == (16, 16)
    got, att = render_attribution(dev, r, BASE, 16, 16, g["K"], c2w, coarse, fine, seeds, g["pose_rgb"], g["pose_acc"], "g7 pose")
    report("attribution_g7_pose", att)
    assert torch.equal(got["rgb_map"], rgb.reshape(-1, 3).cpu()) and torch.equal(got["acc_map"], acc.reshape(-1).cpu())   # chunk invariance
    assert torch.equal(torch.nan_to_num(got["disp_map"]), torch.nan_to_num(disp.reshape(-1).cpu()))
    close(rgb, g["pose_rgb"], atol=5e-2)
    assert sorted(extras) == sorted(k[len("pose_extra_"):] for k in g if k.startswith("pose_extra_"))
    assert extras["raw"].shape == (16, 16, 192, 4)
    for k in ("rgb0", "disp0", "acc0"):
        close(extras[k], g["pose_extra_" + k], atol=2e-4, rtol=2e-4)
    rays = torch.from_numpy(g["rays_in"]).to(dev)
    rgb, disp, acc, extras = r.render(16, 16, g["K"], coarse, fine, chunk=32768, rays=rays, retraw=False)
    got, att = render_attribution(dev, r, BASE, 16, 16, g["K"], None, coarse, fine, seeds, g["rays_rgb"], acc.cpu(), "g7 rays", rays=rays)
    assert torch.equal(got["rgb_map"], rgb.reshape(-1, 3).cpu())
    assert "raw" not in extras
    for k in ("rgb0", "disp0", "acc0"):
        close(extras[k], g["rays_extra_" + k], atol=2e-4, rtol=2e-4)
    cfg_n = dict(BASE, ndc=True, near=0.0, far=1.0, N_importance=64, white_bkgd=False)
    rn = render_utils.Renderer(**cfg_n)
    ndc_c2w = torch.from_numpy(g["ndc_c2w"])
    rgb, disp, acc, extras = rn.render(12, 16, g["ndc_K"], coarse, fine, chunk=77, c2w=ndc_c2w, retraw=False)
    got, att = render_attribution(dev, rn, cfg_n, 12, 16, g["ndc_K"], ndc_c2w, coarse, fine, seeds, g["ndc_rgb"], g["ndc_acc"], "g7 ndc")
    report("attribution_g7_ndc", att)
    assert torch.equal(got["rgb_map"], rgb.reshape(-1, 3).cpu())
    close(extras["rgb0"], g["ndc_extra_rgb0"], atol=2e-4)
    # wrappers: the same values as render()
    out = r.render_from_pose(16, 16, g["K"], 100, c2w, coarse, fine, retraw=False)
    close(out[0], g["pose_rgb"], atol=5e-2)
    first = r.render(16, 16, g["K"], coarse, fine, chunk=100, c2w=c2w, retraw=False)[0]
    assert torch.equal(out[0], first)
    out = r.render_from_rays(16, 16, g["K"], 32768, rays, coarse, fine, retraw=False)
    assert torch.equal(out[0], r.render(16, 16, g["K"], coarse, fine, chunk=32768, rays=rays, retraw=False)[0])


# ------------------------------------------------------------------ G8 + PSNR
# PSNR gates (dB) of the 64x64 referee crop, a few dB under the measured values
# measured on MI355X: fp32 c19 60.7 (acc 70.9), c12 54.6 (acc 52.2) -- the exact-fp32 kernel end to end, limited by the
# conditioning of sample_pdf (see close_frac; the staged tests are the tight check); bf16 c19 37.1 (acc 44.1),
# c12 31.7 (acc 28.5).  SURVEY.md measured 39.8 dB for a CPU bf16 autocast of the reference on x3 weights.
PSNR_GATES = {("c19", "fp32"): 57.0, ("c12", "fp32"): 51.0, ("c19", "bf16"): 34.0, ("c12", "bf16"): 28.5}
G8_LEGS = (("s1", (1, 11, 3.0)), ("c19", (1, 19, 3.0)), ("c12", (1, 12, 3.0)))


def test_psnr_crop(dev, golden):
    """PSNR referee: the 64x64 crop of the 800x800 pose against the reference's fp32 image, on weight sets whose
    fine field puts CONTENT into the crop (an all-white crop scores 300 dB whatever the kernel does):
    c19 opaque, c12 semi-transparent.  s1 (near-empty) only checks the NaN pattern of disp."""
    _, render_utils, _ = amd()
    g = golden("g8_psnr_crop")
    H = W = 800
    K = synth.lego_intrinsics(H, W)
    ro, rd = synth.rays_np(H, W, K, synth.LEGO_C2W, g["pixel_index"])
    rays = torch.from_numpy(np.stack([ro, rd], 0)).to(dev)
    r = render_utils.Renderer(**BASE)
    out = {}
    for tag, (sc, sf, sh) in G8_LEGS:
        if tag != "s1":
            assert g["rgb_" + tag].var() > 1e-2, "referee crop %s has no content" % tag
        for prec in ("fp32", "bf16"):
            c, f = gpu_model(dev, sc, sh, prec, **VD), gpu_model(dev, sf, sh, prec, **VD)
            rgb, disp, acc, extras = r.render(H, W, K, c, f, chunk=4096, rays=rays, retraw=False)
            out["%s_%s" % (tag, prec)] = psnr(rgb, g["rgb_" + tag])
            out["%s_%s_acc" % (tag, prec)] = psnr(acc, g["acc_" + tag])
            if prec == "fp32":
                # disp is NaN exactly where acc == 0 (render_utils.py:284); a ray whose only weight is ~1e-38 may fall
                # either side, so the pattern has to agree on all but a handful of the 4096 rays
                nan_diff = np.isnan(disp.cpu().numpy()) != np.isnan(g["disp_" + tag])
                out["%s_nan_mismatch" % tag] = int(nan_diff.sum())
                assert nan_diff.mean() < 2e-3, (tag, nan_diff.sum())
                close(extras["rgb0"], g["rgb0_" + tag], atol=2e-4)
    report("psnr_crop", out)
    for (tag, prec), gate in PSNR_GATES.items():
        assert out["%s_%s" % (tag, prec)] > gate, (tag, prec, out)
        assert out["%s_%s_acc" % (tag, prec)] > gate - 3.5, (tag, prec, out)


# ------------------------------------------------------------------ oracle on fresh seeded inputs
def oracle_batch(cfg, H, W, K, c2w, idx):
    """The [n, 8|11] batch Renderer.render assembles (render_utils.py:200-226), built with the oracle's ray
    math: viewdirs before the NDC warp, near/far columns, flat pixel subset `idx`."""
    c2w = torch.as_tensor(np.asarray(c2w, np.float32))
    ro, rd = O.get_rays(H, W, K, c2w)
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    if cfg["ndc"]:
        ro, rd = O.ndc_rays(H, W, K[0][0], 1.0, ro, rd)
    ro, rd, vd = (t.reshape(-1, 3).float()[idx] for t in (ro, rd, vd))
    cols = [ro, rd, torch.full_like(rd[:, :1], cfg["near"]), torch.full_like(rd[:, :1], cfg["far"])]
    if cfg["use_viewdirs"]:
        cols.append(vd)
    return torch.cat(cols, -1).contiguous()


def staged_check(dev, cfg, arch, batch, seeds, use_pytest, label, precision="fp32"):
    """Every stage of render_rays against the oracle ON IDENTICAL INPUTS, so the ill-conditioning of
    sample_pdf cannot hide (or fake) a discrepancy:
      coarse pass : oracle end to end (same seeded draws)                      -> tight
      resampling  : oracle sample_pdf + sort on the GPU's own coarse weights, compared where the
                    bin mass makes the samples well-conditioned
      fine pass   : oracle field + compositing (same noise draws) on the GPU's own z_vals -> tight
    Returns the measured maxima for the parity report."""
    _, render_utils, _ = amd()
    R, Nc, Ni = batch.shape[0], cfg["N_samples"], cfg["N_importance"]
    (sc, sf, sharpen) = seeds
    coarse_cpu, fine_cpu = cpu_model(sc, sharpen, **arch), cpu_model(sf, sharpen, **arch)
    coarse_gpu, fine_gpu = gpu_model(dev, sc, sharpen, precision, **arch), gpu_model(dev, sf, sharpen, precision, **arch)
    r = render_utils.Renderer(**cfg)
    out = {k: v.cpu() for k, v in r.render_rays(batch.to(dev), coarse_gpu, fine_gpu, retraw=True, retweights=True,
                                                pytest=use_pytest).items()}
    # coarse pass (the GPU coarse-only run sees the same seeded draws: the pytest path reseeds per call)
    cfg0 = dict(cfg, N_importance=0)
    r0 = render_utils.Renderer(**cfg0)
    out0 = {k: v.cpu() for k, v in r0.render_rays(batch.to(dev), coarse_gpu, None, retraw=True, retweights=True,
                                                  pytest=use_pytest).items()}
    ref0 = O.render_rays(O.RenderCfg(**cfg0), batch, coarse_cpu, None, retraw=True, retweights=True, pytest=use_pytest)
    for k in ref0:
        if k == "disp_map":
            close_disp(out0[k], ref0[k], ref0["acc_map"], Nc, atol=G5_TOL[k], rtol=2e-4, raw_tol=G5_TOL["raw"], far=cfg["far"])
        else:
            close(out0[k], ref0[k], atol=G5_TOL[k], rtol=2e-4)
    for k0, k in (("rgb_map", "rgb0"), ("disp_map", "disp0"), ("acc_map", "acc0")):
        close(out[k], out0[k0], atol=0)                      # same kernels, same inputs: bit identical
    # resampling on the GPU's own coarse weights
    z_c, w_c = out0["z_vals"], out0["weights"]
    z_mid = 0.5 * (z_c[..., 1:] + z_c[..., :-1])
    det = cfg["perturb"] == 0.0
    u = O.pytest_u_for_sample_pdf(R, Ni, det) if use_pytest else torch.linspace(0., 1., Ni).expand(R, Ni)
    z_samples = O.sample_pdf(z_mid, w_c[..., 1:-1], Ni, det=det, u=u.contiguous())
    well = pdf_denominators(z_mid, w_c[..., 1:-1], u) > 1e-3
    z_ref, order = torch.sort(torch.cat([z_c, z_samples], -1), -1)
    well_sorted = np.take_along_axis(np.concatenate([np.ones((R, Nc), bool), well], -1), order.numpy(), -1)
    dz = np.abs(out["z_vals"].numpy() - z_ref.numpy())
    span = float(cfg["far"] - cfg["near"]) if not cfg["ndc"] else 1.0
    n_off = int((dz[well_sorted] >= 5e-6 * span).sum())      # (a few may sit at the end of an empty bin: see the bound below)
    assert n_off <= max(6, int(3e-3 * well_sorted.sum())), (label, n_off, int(well_sorted.sum()))   # (a displaced sample shifts the rank of every depth it passes)
    # The loose bound for everything else.  Where the bin mass is tiny the reference's own formula is discontinuous:
    # `denom < 1e-5 -> 1` (utils.py:110) flips on the last bit of a cumsum difference -- an empty bin of an opaque ray has
    # pdf = 1e-5 / (sum w + 62e-5), right at the threshold -- and moves the sample from the bin's edge to anywhere inside
    # it; in the sorted row every depth it passes shifts one place.  One bin width bounds both.
    bin_w = (z_mid[:, 1:] - z_mid[:, :-1]).max(-1)[0].numpy()[:, None] if Nc > 2 else np.full((R, 1), span)
    assert (dz <= bin_w * 1.001 + 1e-6 * span).all(), (label, float((dz - bin_w).max()))
    # z_std (render_utils.py:168) is the spread of the fine samples alone.  Against the oracle's samples it can only be
    # compared on rays without an ill-conditioned sample (the u = 1 sample of an opaque ray is one: its bin is empty and it
    # sits at either end of it); on EVERY ray it must be the spread of the samples this run drew, which are the merged row
    # minus the coarse depths.
    all_well = well.all(-1)
    if all_well.any():
        close_frac(out["z_std"][all_well], torch.std(z_samples, dim=-1, unbiased=False)[all_well], atol=5e-6 * span, rtol=1e-4,
                   frac=0.9)
    zg, zc = out["z_vals"].numpy(), z_c.numpy()
    drawn = np.ones(zg.shape, bool)
    for r_ in range(R):
        drawn[r_, np.searchsorted(zg[r_], zc[r_], side="left")] = False
    assert (drawn.sum(-1) == Ni).all(), label
    own = zg[drawn].reshape(R, Ni).astype(np.float64)
    close(out["z_std"], own.std(-1), atol=5e-6 * span, rtol=1e-4)
    # fine pass on the GPU's own z_vals
    z = out["z_vals"]
    pts = batch[:, None, 0:3] + batch[:, None, 3:6] * z[..., None]
    raw = O.nerf_forward(fine_cpu[0], fine_cpu[1], pts, batch[:, 8:11] if cfg["use_viewdirs"] else None)
    close(out["raw"], raw, atol=2e-4, rtol=2e-4)
    noise1 = None
    if cfg["raw_noise_std"] > 0.:
        noise1 = O.pytest_uniform([R, Nc + Ni]) * cfg["raw_noise_std"]
    rgb, disp, acc, weights, _ = O.raw2outputs(out["raw"], z, batch[:, 3:6], cfg["white_bkgd"], noise1)
    close(out["rgb_map"], rgb, atol=1e-5, rtol=1e-5)
    close(out["acc_map"], acc, atol=1e-5, rtol=1e-5)
    close_disp(out["disp_map"], disp, acc, Nc + Ni, atol=1e-5, rtol=1e-4)
    close(out["weights"], weights, atol=1e-6, rtol=1e-4)
    return dict(raw_max=float((out["raw"] - raw).abs().max()), rgb_max=float((out["rgb_map"] - rgb).abs().max()),
                z_well_max=float(dz[well_sorted].max()), z_max=float(dz.max()), rgb_var=float(out["rgb_map"].var()))


STAGED_CASES = {
    # name: (cfg overrides, arch, (coarse seed, fine seed, sharpen), pytest draws, ndc camera)
    "det": (dict(N_samples=48, N_importance=80), VD, (21, 22, 3.0), False, False),
    "perturb": (dict(perturb=1.0), VD, (1, 19, 3.0), True, False),
    "lindisp": (dict(lindisp=True, N_samples=40, N_importance=72), VD, (1, 12, 3.0), False, False),
    "noise_black_perturb": (dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False), VD, (1, 19, 3.0), True, False),
    "novd": (dict(use_viewdirs=False), NOVD, (2, 12, 3.0), False, False),
    "ndc_fern": (dict(ndc=True, near=0.0, far=1.0, N_importance=64, white_bkgd=False, perturb=1.0, raw_noise_std=1.0),
                 VD, (1, 12, 3.0), True, True),
}


@pytest.mark.parametrize("name", sorted(STAGED_CASES))
def test_render_rays_vs_oracle_staged(dev, name):
    """Staged oracle comparison (see staged_check) for the deterministic configuration and for every branch
    of render_rays the end-to-end goldens only cover with a fraction criterion: stratified jitter,
    lindisp, sigma noise + black background, no view directions, NDC rays (render_utils.py:105-156).
    Ragged sizes on purpose: 333 rays."""
    report("staged_" + name, run_staged_case(dev, name, "fp32"))


def run_staged_case(dev, name, precision):
    over, arch, seeds, use_pytest, ndc_cam = STAGED_CASES[name]
    cfg = dict(BASE, **over)
    rng = np.random.default_rng(99)
    if ndc_cam:
        H, W, focal = 378, 504, 408.0
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        c2w = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    else:
        H = W = 400
        K = synth.lego_intrinsics(H, W)
        c2w = synth.pose_spherical(37.0)
    idx = np.sort(rng.choice(H * W, size=333, replace=False))
    batch = oracle_batch(cfg, H, W, K, c2w, idx)
    return staged_check(dev, cfg, arch, batch, seeds, use_pytest, name, precision)


def test_empty_and_errors(dev):
    nerf, render_utils, utils = amd()
    from nerf_shared_amd._lib import NerfAmdError
    m = gpu_model(dev, 0, 1.0, "bf16", **VD)
    r = render_utils.Renderer(**BASE)
    out = r.render_rays(torch.empty(0, 11, device=dev), m, m)
    assert out["rgb_map"].shape == (0, 3) and out["z_std"].shape == (0,)
    with pytest.raises(NerfAmdError):
        r.render_rays(torch.zeros(4, 11), m, m)                      # CPU tensor: no CPU path
    with pytest.raises(NerfAmdError):
        r.render_rays(torch.zeros(4, 7, device=dev), m, m)           # bad width
    with pytest.raises(NerfAmdError):
        r.render_rays(torch.zeros(4, 8, device=dev), m, m)           # viewdirs model, no viewdirs in batch
    with pytest.raises(TypeError):
        r.render_rays(torch.zeros(4, 11, device=dev), torch.nn.Linear(3, 4), None)
    with pytest.raises(NerfAmdError):
        m(torch.zeros(2, 3, 3, device=dev), None)                    # use_viewdirs model needs viewdirs


def test_sample_counts_beyond_the_lds_are_refused_with_the_real_reason(dev):
    """The per-ray kernels keep a ray's samples in LDS.  Counts that do not fit the CU's 160 KiB are refused before
    anything is launched, with the byte counts in the message (not a HIP launch failure with a made-up limit); counts
    that need more than the 64-KiB default opt in and run."""
    nerf, render_utils, utils = amd()
    from nerf_shared_amd._lib import NerfAmdError
    K = synth.lego_intrinsics(40, 40)
    m = gpu_model(dev, 1, 3.0, "bf16", **VD)
    batch = utils.make_ray_batch(40, 40, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, n=8)
    big = render_utils.Renderer(**dict(BASE, N_samples=2049, N_importance=2047))
    with pytest.raises(NerfAmdError, match="bytes of LDS"):
        big.render_rays(batch, m, m)
    mid = render_utils.Renderer(**dict(BASE, N_samples=1024, N_importance=1024))      # 4 rays x 20 KB: beyond the 64-KiB default
    out = mid.render_rays(batch, m, m, retweights=True)
    torch.cuda.synchronize()
    z = out["z_vals"]
    assert z.shape == (8, 2048) and bool((z[:, 1:] >= z[:, :-1]).all()) and bool(torch.isfinite(out["rgb_map"]).all())
    r = render_utils.Renderer(**BASE)
    with pytest.raises(NerfAmdError, match="bytes of LDS"):
        utils.sample_pdf(torch.rand(4, 6000, device=dev).sort(-1)[0], torch.rand(4, 5999, device=dev), 16, det=True)
    raw = torch.randn(4, 4000, 4, device=dev, requires_grad=True)
    zz = torch.rand(4, 4000, device=dev).sort(-1)[0]
    rgb = r.raw2outputs(raw, zz, torch.randn(4, 3, device=dev))[0]
    with pytest.raises(NerfAmdError, match="bytes of LDS"):
        rgb.sum().backward()


def test_gradient_requests_outside_the_fused_training_kernels(dev):
    """main.py:103 calls loss.backward() on whatever render() returned.  For a model the FUSED training kernels do not cover
    (other depths and widths, an output_linear model with a multires they are not instantiated for -- in any precision) a
    call made with gradients requested runs on the exact-fp32 training path (csrc/train_f32.hip; round 4 -- it used to
    raise): same values as plain inference, finite gradients for every parameter, and for the rays when THEY ask
    (test_gpu_train_f32.py holds the comparisons with autograd).  Nothing comes back silently without history."""
    nerf, render_utils, utils = amd()
    K = synth.lego_intrinsics(40, 40)
    small = dict(D=4, W=128, output_ch=4, skips=[1], use_viewdirs=True, multires=6, multires_views=2)
    cases = [("D=4 W=128, fp32", small, "fp32", True), ("no view branch, multires 6, split precision", dict(NOVD, multires=6), "fp32_split", False),
             ("no view branch, multires 6", dict(NOVD, multires=6), "bf16", False), ("D=4 W=128", small, "bf16", True)]
    for label, arch, prec, vd in cases:
        rr = render_utils.Renderer(**dict(BASE, N_samples=16, N_importance=16, use_viewdirs=vd))
        torch.manual_seed(5)
        m = nerf.NeRF(**arch).to(dev)
        m.precision = prec
        with torch.no_grad():                                          # lift the density: an empty volume has zero gradients
            (m.alpha_linear.bias if vd else m.output_linear.bias[3:4]).add_(1.0)
        batch = utils.make_ray_batch(40, 40, K, synth.LEGO_C2W, 2.0, 6.0, vd, False, device=dev, n=64)
        assert all(p.requires_grad for p in m.parameters())
        with torch.no_grad():                                          # the reference values: plain inference
            want = rr.render_rays(batch, m, m)
        assert want["rgb_map"].shape == (64, 3) and not want["rgb_map"].requires_grad
        pts = torch.rand(4, 3, 3, device=dev)
        calls = [lambda: rr.render_rays(batch, m, m)["rgb_map"],                      # grad mode on, parameters require grad
                 lambda: rr.render_batch(m, m, batch, chunk=32)["rgb_map"],
                 lambda: rr.render(40, 40, K, m, m, chunk=48, c2w=torch.from_numpy(synth.LEGO_C2W), retraw=False)[0],
                 lambda: m(pts, torch.ones(4, 3, device=dev) if vd else None)]
        for i, call in enumerate(calls):
            m.zero_grad(set_to_none=True)
            out = call()
            assert out.requires_grad and out.grad_fn is not None, (label, i)          # history, not a bare tensor
            if i < 2:
                close(out.detach(), want["rgb_map"], atol=2e-6)                       # exact fp32 on both sides (these models render on the exact kernel)
            out.sum().backward()
            used = [p for n, p in m.named_parameters() if not (not vd and n.startswith("views_linears"))]
            assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in used), (label, i)
            assert float(m.pts_linears[0].weight.grad.abs().sum()) > 0, (label, i)
        m.requires_grad_(False)
        m.zero_grad(set_to_none=True)
        out = rr.render_rays(batch, m, m)                              # nothing requires grad: no history, no trap
        assert not out["rgb_map"].requires_grad
        rays = batch.clone().requires_grad_(True)                      # ... unless the rays do (pose estimation): this path has those too
        out = rr.render_rays(rays, m, m)["rgb_map"]
        assert out.requires_grad
        close(out.detach(), want["rgb_map"], atol=2e-6)
        out.sum().backward()
        assert rays.grad is not None and bool(torch.isfinite(rays.grad).all()) and float(rays.grad[:, :6].abs().sum()) > 0
        assert all(p.grad is None for p in m.parameters())
    r = render_utils.Renderer(**dict(BASE, N_samples=16, N_importance=16))
    assert r is not None
    # the covered models keep their history in every precision (bf16 kernels, or the split-precision ones for 'fp32_split'
    # and 'fp32'): with the view branch, and the output_linear model that NeRF() / config_parser.py:50 build by default
    for arch, vd, prec in ((VD, True, "bf16"), (NOVD, False, "bf16"), (dict(D=8, W=256, skips=[4]), False, "bf16"),
                           (VD, True, "fp32_split"), (NOVD, False, "fp32_split"), (VD, True, "fp32")):
        torch.manual_seed(5)                  # nn.Linear's own init: fix it, and lift the density bias so that the volume is
        m = nerf.NeRF(**arch).to(dev)         # not empty (sigma <= 0 everywhere means zero gradients, legitimately)
        m.precision = prec
        with torch.no_grad():
            (m.alpha_linear.bias if vd else m.output_linear.bias[3:4]).add_(1.0)
        rr = render_utils.Renderer(**dict(BASE, N_samples=16, N_importance=16, use_viewdirs=vd))
        out = rr.render_rays(utils.make_ray_batch(40, 40, K, synth.LEGO_C2W, 2.0, 6.0, vd, False, device=dev, n=64), m, m)
        assert out["rgb_map"].requires_grad
        out["rgb_map"].sum().backward()
        assert m.pts_linears[0].weight.grad is not None and bool(torch.isfinite(m.pts_linears[0].weight.grad).all())
        head = m.rgb_linear if vd else m.output_linear
        assert head.weight.grad is not None and float(head.weight.grad.abs().sum()) > 0


# ------------------------------------------------------------------ properties at the benchmark batch size
def test_full_batch_properties(dev):
    """C2 size: 4096 rays x (64 + 128) samples, bf16.  Properties that hold for any input."""
    _, render_utils, utils = amd()
    K = synth.lego_intrinsics(400, 400)
    batch = utils.make_ray_batch(400, 400, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, pix0=80000, n=4096)
    c, f = gpu_model(dev, 1, 3.0, "bf16", **VD), gpu_model(dev, 11, 3.0, "bf16", **VD)
    r = render_utils.Renderer(**BASE)
    out = r.render_rays(batch, c, f, retraw=True, retweights=True)
    z, w = out["z_vals"], out["weights"]
    assert z.shape == (4096, 192) and bool((z[:, 1:] >= z[:, :-1]).all())          # sorted
    assert bool((z >= 2.0 - 1e-5).all()) and bool((z <= 6.0 + 1e-5).all())
    assert bool((w >= 0).all()) and bool((w.sum(-1) <= 1 + 1e-4).all())
    close(out["acc_map"], w.sum(-1), atol=1e-5)
    assert bool(((out["rgb_map"] >= -1e-5) & (out["rgb_map"] <= 1 + 1e-4)).all())
    # the coarse grid is a subset of the merged samples
    t = torch.linspace(0., 1., 64, device=dev)
    zc = 2.0 * (1 - t) + 6.0 * t
    pos = torch.searchsorted(z.contiguous(), zc.expand(4096, 64).contiguous())
    assert bool((torch.gather(z, 1, pos.clamp(max=191)) == zc).all())
    # chunk invariance (deterministic config): different chunking, bit-identical maps
    full = r.render_batch(c, f, batch, chunk=4096)
    parts = r.render_batch(c, f, batch, chunk=1000)
    for k in full:
        close(full[k], parts[k], atol=0)
    close(full["rgb_map"], out["rgb_map"], atol=0)
    # ray-permutation equivariance, bit exact
    perm = torch.randperm(4096, device=dev)
    outp = r.render_rays(batch[perm].contiguous(), c, f)
    close(outp["rgb_map"], out["rgb_map"][perm], atol=0)
    close(outp["disp_map"], out["disp_map"][perm], atol=0)
    # bf16 vs fp32 on the same rays: image-level agreement
    for m in (c, f):
        m.precision = "fp32"
    out32 = r.render_rays(batch, c, f)
    assert psnr(out["rgb_map"], out32["rgb_map"]) > 30


def test_perturbed_run_is_seed_reproducible(dev):
    _, render_utils, utils = amd()
    K = synth.lego_intrinsics(400, 400)
    batch = utils.make_ray_batch(400, 400, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, pix0=1000, n=512)
    c, f = gpu_model(dev, 1, 3.0, "bf16", **VD), gpu_model(dev, 11, 3.0, "bf16", **VD)
    r = render_utils.Renderer(**dict(BASE, perturb=1.0, raw_noise_std=1.0))
    torch.manual_seed(5)
    a = r.render_rays(batch, c, f, retweights=True)
    torch.manual_seed(5)
    b = r.render_rays(batch, c, f, retweights=True)
    close(a["rgb_map"], b["rgb_map"], atol=0)
    close(a["z_vals"], b["z_vals"], atol=0)
    assert bool((a["z_vals"][:, 1:] >= a["z_vals"][:, :-1]).all())
    cdet = render_utils.Renderer(**BASE).render_rays(batch, c, f, retweights=True)
    assert not torch.equal(cdet["z_vals"], a["z_vals"])


def test_seeded_multichunk_render_equals_per_chunk_render_rays(dev):
    """The reference draws t_rand, noise0, u, noise1 inside every render_rays call, i.e. per chunk of
    render_batch (render_utils.py:56-57,121,264; utils.py:86).  A seeded multi-chunk render must therefore
    equal seeded per-chunk render_rays calls -- in the one-library-call mode, the per-chunk mode and the
    two-stream overlap mode alike."""
    _, render_utils, utils = amd()
    K = synth.lego_intrinsics(400, 400)
    batch = utils.make_ray_batch(400, 400, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, pix0=70000, n=1000)
    c, f = gpu_model(dev, 1, 3.0, "bf16", **VD), gpu_model(dev, 19, 3.0, "bf16", **VD)
    r = render_utils.Renderer(**dict(BASE, perturb=1.0, raw_noise_std=1.0))
    chunk = 300                                                       # 4 chunks, the last one ragged
    torch.manual_seed(11)
    want = [r.render_rays(batch[i:i + chunk], c, f) for i in range(0, 1000, chunk)]
    want = {k: torch.cat([w[k] for w in want], 0) for k in want[0]}
    R = render_utils.Renderer
    modes = {"pipeline_batch": (True, True, False), "fused": (False, True, False), "per_chunk": (False, False, False),
             "overlap": (False, True, True)}
    try:
        for name, (pipe, fuse, overlap) in modes.items():
            R.pipeline_batch, R.fuse_chunk_launches, R.overlap_chunks = pipe, fuse, overlap
            torch.manual_seed(11)
            got = r.render_batch(c, f, batch, chunk=chunk)
            torch.cuda.synchronize()
            for k in want:
                assert torch.equal(torch.nan_to_num(got[k]), torch.nan_to_num(want[k])), (name, k)
    finally:
        R.pipeline_batch, R.fuse_chunk_launches, R.overlap_chunks = True, True, False
    torch.manual_seed(12)                                             # and another seed gives another image
    other = r.render_batch(c, f, batch, chunk=chunk)
    assert not torch.equal(other["rgb_map"], want["rgb_map"])


def test_pipelined_batch_over_several_launch_groups(dev):
    """nerf_amd_render_batch: 100 000 rays = three full 32768-ray launch groups and a ragged fourth, per-ray kernels on the
    library's side stream.  Every output equals the chunk-at-a-time path bit for bit -- deterministic, with random draws
    (perturb + noise, chunk 4096: 25 chunks of draws inside 4 launch groups), coarse-only, and with raw / no fine model."""
    _, render_utils, utils = amd()
    R = render_utils.Renderer
    K = synth.lego_intrinsics(400, 400)
    batch = utils.make_ray_batch(400, 400, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, pix0=30000, n=100000)
    c, f = gpu_model(dev, 1, 3.0, "bf16", **VD), gpu_model(dev, 19, 3.0, "bf16", **VD)
    cases = ((dict(), 32768, f, False, None), (dict(perturb=1.0, raw_noise_std=1.0), 4096, f, False, 5),
             (dict(N_importance=0), 40000, None, False, None), (dict(N_samples=32, N_importance=32), 50000, None, True, None))
    for over, chunk, fine, retraw, seed in cases:
        r = R(**dict(BASE, **over))
        outs = []
        for pipe in (True, False):
            R.pipeline_batch = pipe
            try:
                if seed is not None:
                    torch.manual_seed(seed)
                outs.append(r.render_batch(c, fine, batch, chunk=chunk, retraw=retraw))
                torch.cuda.synchronize()
            finally:
                R.pipeline_batch = True
        assert sorted(outs[0]) == sorted(outs[1])
        for k in outs[0]:
            assert torch.equal(torch.nan_to_num(outs[0][k]), torch.nan_to_num(outs[1][k])), (over, k)
    # back-to-back calls reuse the side stream and its events: results stay put
    r = R(**BASE)
    a = r.render_batch(c, f, batch, chunk=32768)
    b = r.render_batch(c, f, batch, chunk=32768)
    torch.cuda.synchronize()
    assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["z_std"], b["z_std"])


def test_renders_on_two_streams_do_not_share_scratch(dev):
    """Two renders enqueued on different streams without a sync in between: each call owns its workspace
    (allocated on its stream), so neither overwrites the other's raw / z / weights."""
    _, render_utils, utils = amd()
    K = synth.lego_intrinsics(400, 400)
    c, f = gpu_model(dev, 1, 3.0, "bf16", **VD), gpu_model(dev, 19, 3.0, "bf16", **VD)
    r = render_utils.Renderer(**BASE)
    b1 = utils.make_ray_batch(400, 400, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, pix0=60000, n=3000)
    b2 = utils.make_ray_batch(400, 400, K, synth.pose_spherical(50.0), 2.0, 6.0, True, False, device=dev, pix0=90000, n=3000)
    want1, want2 = r.render_batch(c, f, b1, chunk=1024), r.render_batch(c, f, b2, chunk=1024)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for _ in range(3):
        with torch.cuda.stream(s1):
            got1 = r.render_batch(c, f, b1, chunk=1024)
        with torch.cuda.stream(s2):
            got2 = r.render_batch(c, f, b2, chunk=1024)
        torch.cuda.synchronize()
        for k in want1:
            assert torch.equal(torch.nan_to_num(got1[k]), torch.nan_to_num(want1[k])), k
            assert torch.equal(torch.nan_to_num(got2[k]), torch.nan_to_num(want2[k])), k


# ------------------------------------------------------------------ multi-rank rehearsal on one GPU
_SHARD_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
os.environ["NERF_AMD_QUIET"] = "1"
from nerf_shared_amd import dist as nd, nerf, render_utils, synth
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
models = []
for seed in (1, 11):
    m = nerf.NeRF(**arch); m.load_state_dict(synth.torch_state_dict(seed, 3.0, **{**arch, "skips": (4,)})); models.append(m.to(dev))
r = render_utils.Renderer(perturb=0.0, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=True,
                          raw_noise_std=0.0, ndc=False, lindisp=False, near=2.0, far=6.0)
H, W = 37, 41                                   # 1517 pixels: ragged shards
K = synth.lego_intrinsics(H, W)
with torch.no_grad():
    got = nd.render_image_sharded(r, H, W, K, synth.LEGO_C2W, models[0], models[1], chunk=500)
    ok = True
    if rank == 0:
        rgb, disp, acc, _ = r.render(H, W, K, models[0], models[1], chunk=4096, c2w=torch.from_numpy(synth.LEGO_C2W), retraw=False)
        ok = torch.equal(got[0], rgb) and torch.equal(got[2], acc) and torch.equal(torch.nan_to_num(got[1]), torch.nan_to_num(disp))
    else:
        ok = got is None
    # uint8 gather: quantised on every rank, 3 bytes per pixel on the wire
    got8 = nd.render_image_sharded(r, H, W, K, synth.LEGO_C2W, models[0], models[1], chunk=500, as_uint8=True)
    if rank == 0:
        from nerf_shared_amd import utils
        ok = ok and got8.dtype == torch.uint8 and torch.equal(got8, utils.to8b(rgb))
    else:
        ok = ok and got8 is None
    # the C5 loop: every pose rendered by all ranks together, gathers overlapped with the next frame
    poses3 = [torch.from_numpy(synth.pose_spherical(a)) for a in (10.0, 130.0, 250.0)]
    frames = nd.render_poses_gathered(r, H, W, K, 500, poses3, models[0], models[1])
    if rank == 0:
        ok = ok and len(frames) == 3
        for (g_rgb, g_disp, g_acc), c2w in zip(frames, poses3):
            w_rgb, w_disp, w_acc, _ = r.render(H, W, K, models[0], models[1], chunk=4096, c2w=c2w, retraw=False)
            ok = ok and torch.equal(g_rgb, w_rgb) and torch.equal(g_acc, w_acc) and torch.equal(torch.nan_to_num(g_disp), torch.nan_to_num(w_disp))
    else:
        ok = ok and frames is None
    # whole frames dealt round-robin, every rank writes its own PNGs
    import numpy as np
    from nerf_shared_amd import image_io, utils
    poses = [torch.from_numpy(synth.pose_spherical(a)) for a in (0.0, 40.0, 80.0)]
    mine = nd.render_poses_sharded(r, H, W, K, 4096, poses, models[0], models[1], sys.argv[4])
    ok = ok and mine == list(range(rank, 3, world))
    if rank == 0:
        ok = ok and sorted(os.listdir(sys.argv[4])) == ["000.png", "001.png", "002.png"]
        for i, c2w in enumerate(poses):
            want = utils.to8b(r.render(H, W, K, models[0], models[1], chunk=4096, c2w=c2w, retraw=False)[0]).cpu().numpy()
            ok = ok and np.array_equal(image_io.read_image(os.path.join(sys.argv[4], "%%03d.png" %% i)), want)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_sharded_render_two_ranks_matches_single(dev, tmp_path):
    """Ray-range sharding + gather (nerf_shared_amd/dist.py) with 2 ranks sharing this GPU
    (gloo stands in for RCCL): the gathered image is bit-identical to a single-rank render."""
    import subprocess
    import sys
    script = tmp_path / "shard_worker.py"
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script.write_text(_SHARD_WORKER % {"repo": repo})
    port = str(29600 + os.getpid() % 2000)
    frames = tmp_path / "frames"
    procs = [subprocess.Popen([sys.executable, str(script), str(rk), "2", port, str(frames)]) for rk in range(2)]
    try:
        codes = [p.wait(timeout=170) for p in procs]
    finally:
        for p in procs:                       # never leave a worker holding the GPU behind a failed test
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0, 0]


def test_staticcam_overlap_and_batch_poses(dev, tmp_path):
    """c2w_staticcam (render_utils.py:208-210), the opt-in two-stream chunk pipeline, and
    render_from_batch_poses (render_utils.py:293-319)."""
    _, render_utils, _ = amd()
    H = W = 24
    K = synth.lego_intrinsics(H, W)
    c2w, c2w_s = torch.from_numpy(synth.LEGO_C2W), torch.from_numpy(synth.pose_spherical(20.0))
    cfg = dict(BASE, N_samples=32, N_importance=32)
    coarse_gpu, fine_gpu = gpu_model(dev, 1, 3.0, "fp32", **VD), gpu_model(dev, 11, 3.0, "fp32", **VD)
    r = render_utils.Renderer(**cfg)
    rgb, disp, acc, extras = r.render(H, W, K, coarse_gpu, fine_gpu, chunk=200, c2w=c2w, c2w_staticcam=c2w_s, retraw=False)
    ref = O.render(O.RenderCfg(**cfg), H, W, K, cpu_model(1, 3.0, **VD), cpu_model(11, 3.0, **VD), chunk=200,
                   c2w=c2w, c2w_staticcam=c2w_s, retraw=False)
    close(extras["rgb0"], ref[3]["rgb0"], atol=2e-4)          # coarse pass: tight
    got, att = render_attribution(dev, r, cfg, H, W, K, c2w, coarse_gpu, fine_gpu, (1, 11, 3.0), ref[0], ref[2], "staticcam",
                                  c2w_staticcam=c2w_s)
    assert torch.equal(got["rgb_map"], rgb.reshape(-1, 3).cpu())
    # the static-camera image differs from the plain one (view directions come from c2w, rays from c2w_s)
    plain = r.render(H, W, K, coarse_gpu, fine_gpu, chunk=200, c2w=c2w_s, retraw=False)[0]
    assert not torch.equal(plain, rgb)
    # two-stream chunk overlap: bit-identical results
    try:
        render_utils.Renderer.overlap_chunks, render_utils.Renderer.pipeline_batch = True, False
        rgb2 = r.render(H, W, K, coarse_gpu, fine_gpu, chunk=100, c2w=c2w, c2w_staticcam=c2w_s, retraw=False)[0]
    finally:
        render_utils.Renderer.overlap_chunks, render_utils.Renderer.pipeline_batch = False, True
    rgb1 = r.render(H, W, K, coarse_gpu, fine_gpu, chunk=100, c2w=c2w, c2w_staticcam=c2w_s, retraw=False)[0]
    close(rgb2, rgb1, atol=0)
    # batch of poses -> frames on disk
    # render_batch as one library call (nerf_amd_render_chunks: final compositing of chunk k-1 inside chunk k's
    # compositing/resampling launch; 6 chunks here, the last one ragged) against one render_rays call per
    # chunk: bit-identical, every output, deterministic and with random draws
    assert render_utils.Renderer.fuse_chunk_launches and render_utils.Renderer.pipeline_batch
    for rcfg, seed in ((cfg, None), (dict(cfg, perturb=1.0, raw_noise_std=1.0), 3), (dict(cfg, N_importance=0), None)):
        rr = render_utils.Renderer(**rcfg)
        outs = []
        for pipe, fused in ((True, True), (False, True), (False, False)):      # whole-batch call / chunk list / one call per chunk
            render_utils.Renderer.pipeline_batch, render_utils.Renderer.fuse_chunk_launches = pipe, fused
            try:
                if seed is not None:
                    torch.manual_seed(seed)
                outs.append(rr.render(H, W, K, coarse_gpu, fine_gpu, chunk=100, c2w=c2w, retraw=True))
            finally:
                render_utils.Renderer.pipeline_batch, render_utils.Renderer.fuse_chunk_launches = True, True
        for other in outs[1:]:
            for a, b in zip(outs[0][:3], other[:3]):
                assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
            assert sorted(outs[0][3]) == sorted(other[3])
            for k in outs[0][3]:
                assert torch.equal(torch.nan_to_num(outs[0][3][k]), torch.nan_to_num(other[3][k])), k
    # batch of poses -> quantised on the GPU, PNG frames on disk (asynchronous and synchronous writers)
    from nerf_shared_amd import image_io, utils as amd_utils
    want8 = amd_utils.to8b(plain.cpu().numpy())                 # the reference's numpy to8b
    for workers in (4, 0):
        d = tmp_path / ("frames%d" % workers)
        out = r.render_from_batch_poses(H, W, K, 4096, [c2w, c2w_s], coarse_gpu, fine_gpu, False, str(d), io_workers=workers)
        assert len(out) == 2 and out[0].shape == (H, W, 3) and out[0].dtype == np.uint8
        assert sorted(os.listdir(d)) == ["000.png", "001.png"]
        assert np.array_equal(out[1], want8)
        assert np.array_equal(image_io.read_image(str(d / "001.png")), want8)
        assert np.array_equal(image_io.decode_png(open(d / "001.png", "rb").read()), want8)
    # b_combine_as_video without imageio: an animated GIF of the same frames
    d = tmp_path / "video"
    r.render_from_batch_poses(H, W, K, 4096, [c2w, c2w_s], coarse_gpu, fine_gpu, False, str(d), b_combine_as_video=True)
    try:
        import imageio  # noqa: F401
        assert os.path.exists(d / "video.mp4")
    except ImportError:
        pytest.importorskip("PIL")
        from PIL import Image
        with Image.open(d / "video.gif") as g:
            assert g.n_frames == 2 and g.size == (W, H)


def test_to8b_matches_numpy_bit_for_bit(dev):
    """utils.to8b on the device (nerf_amd_to8b) against the reference's numpy expression
    (utils.py:30) on edge values, every multiple of 1/255 +- 1 ulp, and random data of odd length."""
    from nerf_shared_amd import utils as amd_utils
    k = np.arange(256, dtype=np.float32) / np.float32(255)
    edge = np.concatenate([k, np.nextafter(k, np.float32(2)), np.nextafter(k, np.float32(-1)),
                           np.array([-1e30, -1.0, -0.0, 0.0, 1e-45, 0.5, 1.0, 1.0000001, 7.0, 1e30, np.inf, -np.inf], np.float32)])
    rng = np.random.default_rng(5)
    rnd = rng.uniform(-0.2, 1.2, size=(401, 399, 3)).astype(np.float32)        # odd element count: tail path
    for x in (edge, rnd, rnd.reshape(-1)[:7], rnd[:0]):
        got = amd_utils.to8b(torch.from_numpy(x).to(dev))
        assert got.dtype == torch.uint8 and got.shape == x.shape
        assert np.array_equal(got.cpu().numpy(), (255 * np.clip(x, 0, 1)).astype(np.uint8))
    assert amd_utils.to8b(torch.tensor([float("nan")], device=dev)).item() == 0


def test_resample_stage_against_sample_pdf_and_torch_sort(dev):
    """nerf_amd_resample = z_mid -> sample_pdf(weights[1:-1]) -> z_std -> sort(cat) (render_utils.py:140-148,
    :168).  The kernel merges two runs instead of sorting the concatenation: check it against
    utils.sample_pdf + torch.sort bit for bit when (a) both runs arrive sorted (deterministic u), (b) the
    samples arrive unsorted (random u), (c) the coarse depths themselves are unsorted and full of ties."""
    from nerf_shared_amd import _lib, utils as amd_utils
    lib = _lib.lib
    rng = np.random.default_rng(9)
    R, Nc, Ni = 333, 64, 128
    for case in ("sorted", "random_u", "unsorted_coarse_with_ties", "odd_sizes"):
        nc, ni = (Nc, Ni) if case != "odd_sizes" else (37, 50)
        z = np.sort(rng.uniform(2, 6, size=(R, nc)).astype(np.float32), -1)
        if case == "unsorted_coarse_with_ties":
            z = np.round(z * 4) / 4                                     # many equal depths
            z = np.take_along_axis(z, rng.permuted(np.tile(np.arange(nc), (R, 1)), axis=1), -1).astype(np.float32)
        w = rng.uniform(0, 1, size=(R, nc)).astype(np.float32)
        w[5] = 0.0                                                      # all-zero row: uniform pdf
        u = None if case == "sorted" else rng.uniform(0, 1, size=(R, ni)).astype(np.float32)
        zt, wt = torch.from_numpy(z).to(dev), torch.from_numpy(w).to(dev)
        ut = None if u is None else torch.from_numpy(u).to(dev)
        t_lin = torch.linspace(0., 1., ni, device=dev)
        z_fine = torch.empty(R, nc + ni, device=dev)
        z_std = torch.empty(R, device=dev)
        _lib.check(lib.nerf_amd_resample(zt.data_ptr(), wt.data_ptr(), _lib.ptr(ut), t_lin.data_ptr(), R, nc, ni,
                                         z_fine.data_ptr(), z_std.data_ptr(), _lib.stream_of(dev)), "nerf_amd_resample")
        # expected: the same library's sample_pdf (itself golden-tested) on the draws, then torch
        z_mid = .5 * (zt[:, 1:] + zt[:, :-1])
        samples = torch.empty(R, ni, device=dev)
        zm, wm = z_mid.contiguous(), wt[:, 1:-1].contiguous()
        _lib.check(lib.nerf_amd_sample_pdf(zm.data_ptr(), wm.data_ptr(), _lib.ptr(ut), t_lin.data_ptr(),
                                           R, nc - 1, ni, samples.data_ptr(), _lib.stream_of(dev)), "nerf_amd_sample_pdf")
        want = torch.sort(torch.cat([zt, samples], -1), -1)[0]
        assert torch.equal(z_fine, want), case
        close(z_std, torch.std(samples, -1, unbiased=False), atol=2e-6)


def test_renderer_takes_reference_class_models(dev):
    """A model built by the reference's own class (here: a stand-in with its attributes and layers, tests/test_host_logic.py)
    renders bit-identically to this package's NeRF holding the same weights, and training through the renderer updates the
    reference model's own parameters."""
    nerf, render_utils, utils = amd()
    from test_host_logic import _ReferenceStyleNeRF
    K = synth.lego_intrinsics(40, 40)
    r = render_utils.Renderer(**dict(BASE, N_samples=32, N_importance=32))
    refs, ours = [], []
    for seed in (1, 11):
        sd = synth.torch_state_dict(seed, 3.0, **{**VD, "skips": (4,)})
        ref = _ReferenceStyleNeRF(use_viewdirs=True, output_ch=5)
        ref.load_state_dict(sd)
        refs.append(ref.to(dev))
        ours.append(gpu_model(dev, seed, 3.0, "bf16", **VD))
    c2w = torch.from_numpy(synth.LEGO_C2W)
    with torch.no_grad():
        a = r.render(40, 40, K, refs[0], refs[1], chunk=500, c2w=c2w, retraw=False)
        b = r.render(40, 40, K, ours[0], ours[1], chunk=500, c2w=c2w, retraw=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    batch = utils.make_ray_batch(40, 40, K, synth.LEGO_C2W, 2.0, 6.0, True, False, device=dev, n=64)
    opt = torch.optim.SGD(list(refs[0].parameters()) + list(refs[1].parameters()), lr=1e-2)
    before = refs[0].pts_linears[3].weight.detach().clone()
    out = r.render_rays(batch, refs[0], refs[1])
    (((out["rgb_map"] - 0.3) ** 2).mean() + ((out["rgb0"] - 0.3) ** 2).mean()).backward()
    assert refs[1].rgb_linear.weight.grad is not None and refs[0].pts_linears[3].weight.grad is not None
    opt.step()
    assert not torch.equal(refs[0].pts_linears[3].weight, before)
    with torch.no_grad():                                   # the packed copy followed the reference model's update
        c = r.render(40, 40, K, refs[0], refs[1], chunk=500, c2w=c2w, retraw=False)
    assert not torch.equal(c[0], a[0])
