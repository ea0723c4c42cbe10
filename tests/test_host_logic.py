"""CPU tests of the host side: the C-ABI library loads and exports every symbol
declared in include/nerf_amd.h, the Python drop-in keeps the reference's
surface (signatures, state_dict keys), refuses CPU tensors loudly, and the
multi-GPU sharding/gather logic is correct (gloo, world_size 2).  No compute
call reaches a GPU here.
"""
import ctypes
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

os.environ.setdefault("NERF_AMD_QUIET", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from nerf_shared_amd import _lib, dist as nd, nerf, render_utils, synth, utils  # noqa: E402


def test_library_exports_every_declared_symbol():
    with open(os.path.join(REPO, "include", "nerf_amd.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = set(re.findall(r"\b(nerf_amd_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.nerf_amd_abi_version() == _lib.ABI_VERSION == 7


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.Arch) == 4 * (8 + 8)
    assert ctypes.sizeof(_lib.RenderCfg) == 32
    assert ctypes.sizeof(_lib.RenderIO) == 8 + 8 + 8 * 18 + 8


def test_program_validation_errors():
    lib = _lib.lib
    h = ctypes.c_void_p()
    bad = _lib.make_arch(8, 1056, 4, [4], True, 10, 4, 0)       # W beyond the fp32 kernel's tile budget
    nf = ctypes.c_int64()
    assert lib.nerf_amd_pack_bf16_host(ctypes.byref(bad), 32, None, None, 0, None, ctypes.byref(nf), None, None) == -1
    assert b"W 2..1024" in lib.nerf_amd_last_error()
    odd = _lib.make_arch(8, 250, 4, [4], True, 10, 4, 0)        # any other width is a valid program (exact-fp32 kernel only)
    assert lib.nerf_amd_pack_bf16_host(ctypes.byref(odd), 32, None, None, 0, None, ctypes.byref(nf), None, None) == -3
    bad = _lib.make_arch(8, 256, 4, [7], True, 10, 4, 0)        # skip on the last layer: the reference fails too
    assert lib.nerf_amd_pack_bf16_host(ctypes.byref(bad), 32, None, None, 0, None, ctypes.byref(nf), None, None) == -1
    with pytest.raises(ValueError):
        _lib.make_arch(8, 256, 4, list(range(9)), True, 10, 4, 0)
    del h


def test_nerf_surface_matches_reference():
    sig = inspect.signature(nerf.NeRF.__init__)
    assert list(sig.parameters)[1:] == ["D", "W", "output_ch", "skips", "use_viewdirs", "multires", "multires_views", "i_embed"]
    assert [p.default for p in list(sig.parameters.values())[1:]] == [8, 256, 4, [4], False, 10, 4, 0]
    assert list(inspect.signature(nerf.NeRF.forward).parameters) == ["self", "inputs", "viewdirs", "netchunk"]
    assert list(inspect.signature(nerf.NeRF.get_density).parameters) == ["self", "points", "chunk"]
    m = nerf.NeRF(use_viewdirs=True, output_ch=5)
    assert (m.D, m.W, m.skips, m.use_viewdirs, m.input_ch, m.input_ch_views) == (8, 256, [4], True, 63, 27)
    keys = set(m.state_dict().keys())
    expect = {"pts_linears.%d.%s" % (i, s) for i in range(8) for s in ("weight", "bias")}
    expect |= {"%s.%s" % (n, s) for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear")
               for s in ("weight", "bias")}
    assert keys == expect
    assert sum(p.numel() for p in m.parameters()) == 595844
    assert tuple(m.pts_linears[5].weight.shape) == (256, 319)
    assert tuple(m.views_linears[0].weight.shape) == (128, 283)
    m2 = nerf.NeRF(use_viewdirs=False, output_ch=5)
    assert "output_linear.weight" in m2.state_dict() and "views_linears.0.weight" in m2.state_dict()
    assert tuple(m2.output_linear.weight.shape) == (5, 256)
    # synthetic state dicts have exactly these keys and shapes
    sd = synth.torch_state_dict(0, 1.0, D=8, W=256, output_ch=5, skips=(4,), use_viewdirs=True)
    m.load_state_dict(sd)
    fn, dim = nerf.get_embedder(10, 0)
    assert dim == 63 and callable(fn)
    fn, dim = nerf.get_embedder(10, -1)
    assert dim == 3 and isinstance(fn, torch.nn.Identity)


def test_renderer_surface_matches_reference():
    sig = inspect.signature(render_utils.Renderer.__init__)
    assert list(sig.parameters)[1:] == ["perturb", "N_importance", "N_samples", "use_viewdirs", "white_bkgd",
                                        "raw_noise_std", "ndc", "lindisp", "near", "far"]
    assert [p.default for p in list(sig.parameters.values())[1:]] == [True, 128, 64, True, True, 0.0, False, False, 0.0, 1.0]
    R = render_utils.Renderer
    assert list(inspect.signature(R.render).parameters) == ["self", "H", "W", "K", "coarse_model", "fine_model", "chunk",
                                                            "rays", "retraw", "c2w", "c2w_staticcam"]
    assert inspect.signature(R.render).parameters["chunk"].default == 1024 * 32
    assert inspect.signature(R.render).parameters["retraw"].default is True
    assert list(inspect.signature(R.render_rays).parameters) == ["self", "ray_batch", "coarse_model", "fine_model",
                                                                 "retraw", "retweights", "verbose", "pytest"]
    assert list(inspect.signature(R.render_batch).parameters) == ["self", "coarse_model", "fine_model", "rays_flat", "chunk", "retraw"]
    assert list(inspect.signature(R.raw2outputs).parameters) == ["self", "raw", "z_vals", "rays_d", "pytest"]
    assert list(inspect.signature(R.render_from_pose).parameters) == ["self", "H", "W", "K", "chunk", "c2w", "coarse_model", "fine_model", "retraw"]
    assert list(inspect.signature(R.render_from_rays).parameters) == ["self", "H", "W", "K", "chunk", "rays", "coarse_model", "fine_model", "retraw"]
    assert list(inspect.signature(utils.sample_pdf).parameters) == ["bins", "weights", "N_samples", "det", "pytest"]
    assert list(inspect.signature(utils.get_rays).parameters) == ["H", "W", "K", "c2w"]
    assert list(inspect.signature(utils.ndc_rays).parameters) == ["H", "W", "focal", "near", "rays_o", "rays_d"]
    r = R(perturb=0.0, near=2.0, far=6.0)
    assert isinstance(r, torch.nn.Module) and len(list(r.parameters())) == 0


def test_cpu_tensors_are_refused_loudly():
    """The product has no CPU path: it must raise, never fall back."""
    m = nerf.NeRF(use_viewdirs=True)
    r = render_utils.Renderer(perturb=0.0, near=2.0, far=6.0)
    with pytest.raises(_lib.NerfAmdError, match="no CPU path"):
        m(torch.zeros(2, 4, 3), torch.zeros(2, 3))
    with pytest.raises(_lib.NerfAmdError, match="no CPU path"):
        r.render_rays(torch.zeros(4, 11), m, m)
    with pytest.raises(_lib.NerfAmdError, match="no CPU path"):
        r.raw2outputs(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    with pytest.raises(_lib.NerfAmdError, match="no CPU path"):
        utils.sample_pdf(torch.zeros(2, 5), torch.zeros(2, 4), 8, det=True)
    fn, _ = nerf.get_embedder(4, 0)
    with pytest.raises(_lib.NerfAmdError, match="no CPU path"):
        fn(torch.zeros(2, 3))


def test_product_does_not_import_the_oracle():
    """Only tests, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(REPO, "nerf_shared_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(root, fn)) as f:
                    src = f.read()
                assert "nerf_oracle" not in src and "from oracle" not in src and "import oracle" not in src, fn


def test_metrics_and_numpy_twin():
    x = torch.tensor([[0.0, 0.5], [1.0, 0.25]])
    y = torch.zeros(2, 2)
    assert float(utils.img2mse(x, y)) == pytest.approx((0.25 + 1 + 0.0625) / 4)
    assert float(utils.mse2psnr(torch.tensor(0.01))) == pytest.approx(20.0, abs=1e-4)
    np.testing.assert_array_equal(utils.to8b(np.array([-1.0, 0.0, 0.5, 0.999, 1.0, 2.0])), [0, 0, 127, 254, 255, 255])
    K = synth.lego_intrinsics(4, 6)
    ro, rd = utils.get_rays_np(4, 6, K, synth.LEGO_C2W)
    ro2, rd2 = synth.rays_np(4, 6, K, synth.LEGO_C2W)
    np.testing.assert_allclose(rd.reshape(-1, 3), rd2, atol=1e-6)


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 160000, 640000, 160001):
        for world in (1, 2, 3, 4, 8):
            spans = [nd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = nd.shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
os.environ["NERF_AMD_QUIET"] = "1"
from nerf_shared_amd import dist as nd
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
ok = True
for n in (10, 11, 4096):
    full = torch.arange(n * 5, dtype=torch.float32).reshape(n, 5)
    lo, hi = nd.shard_range(n, rank, world)
    got = nd.gather_rows(full[lo:hi].clone(), n, 0)
    if rank == 0:
        ok = ok and got is not None and torch.equal(got, full)
    else:
        ok = ok and got is None
    # quantised colours (the image output stage gathers uint8 rows): same collective, 1-byte elements
    full8 = (torch.arange(n * 3) %% 251).to(torch.uint8).reshape(n, 3)
    got8 = nd.gather_rows(full8[lo:hi].clone(), n, 0)
    ok = ok and ((got8 is not None and got8.dtype == torch.uint8 and torch.equal(got8, full8)) if rank == 0 else got8 is None)
    # round-robin frame assignment of render_poses_sharded
    ok = ok and list(range(rank, n, world)) == [i for i in range(n) if i %% world == rank]
# weights replicated from rank 0: every rank ends with rank 0's parameters and a stale pack key
import torch.nn as tnn
class _M(tnn.Module):
    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.a, self.b = tnn.Linear(5, 7), tnn.Linear(7, 3)
        self.stale = False
    def weights_changed(self):
        self.stale = True
mine, ref = _M(100 + rank), _M(100)
nd.broadcast_parameters([mine, None], src=0)
ok = ok and all(torch.equal(p, q) for p, q in zip(mine.parameters(), ref.parameters())) and mine.stale
# the overlapped per-frame gather: frames come back in submission order with the right rows, whatever the
# depth of the in-flight window, for even and ragged shards; submit() does not complete the gather itself
for n, depth in ((12, 1), (12, 2), (11, 2), (4097, 3)):
    lo, hi = nd.shard_range(n, rank, world)
    g = nd.OverlappedGather(n, 0, None, depth)
    frames = [torch.arange(n * 5, dtype=torch.float32).reshape(n, 5) * (k + 1) + k for k in range(7)]
    got = []
    for k, full in enumerate(frames):
        g.submit(full[lo:hi].clone())
        ok = ok and len(g._pending) <= depth and len(g._pending) >= 1
        if k == 3:
            got += g.collect()                     # a mid-sequence collect drains what is in flight, in order
            ok = ok and len(g._pending) == 0
    got += g.collect()
    ok = ok and len(got) == 7 and g.submitted == 7
    for k, (a, b) in enumerate(zip(got, frames)):
        ok = ok and ((a is not None and torch.equal(a, b)) if rank == 0 else a is None)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_gather_rows_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"repo": REPO})
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port]) for r in range(2)]
    try:
        codes = [p.wait(timeout=180) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0, 0]


SHIPPING_CXXFLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"      # csrc/Makefile with EXTRA empty


def device_isa(stem, tmp_path):
    """Device assembly of csrc/<stem>.hip as the shipping library contains it: the file the library build left beside the
    object (csrc/Makefile, -save-temps) when (a) the default library is the one under test (no NERF_AMD_LIB override),
    (b) the flag line recorded beside it is the shipping one (not an EXTRA=-D... scratch build) and (c) it is newer than
    the source, every header, the Makefile and the experiment includes; else a fresh `hipcc -S` with the shipping flags
    (minutes for the field kernel)."""
    import shutil
    csrc = os.path.join(REPO, "nerf_shared_amd", "csrc")
    src = os.path.join(csrc, stem + ".hip")
    kept = os.path.join(csrc, "build", stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    flags = os.path.join(csrc, "build", stem + ".flags")
    exp = os.path.join(REPO, "tools", "experiments")
    deps = [src, os.path.join(csrc, "Makefile")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + \
           [os.path.join(REPO, "include", "nerf_amd.h")] + [os.path.join(exp, f) for f in os.listdir(exp) if f.endswith(".inc")]
    if (not os.environ.get("NERF_AMD_LIB") and os.path.exists(kept) and os.path.exists(flags)
            and open(flags).read().split() == SHIPPING_CXXFLAGS.split()
            and os.path.getmtime(kept) >= max(os.path.getmtime(d) for d in deps)):
        return kept
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = str(tmp_path / (stem + ".s"))
    subprocess.run([hipcc] + SHIPPING_CXXFLAGS.split() + ["--cuda-device-only", "-S", src, "-o", asm],
                   check=True, capture_output=True, timeout=900)
    return asm


def test_shipping_flags_are_the_makefiles():
    """device_isa's idea of the shipping flags is the Makefile's CXXFLAGS with EXTRA empty."""
    mk = open(os.path.join(REPO, "nerf_shared_amd", "csrc", "Makefile")).read()
    line = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1)
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
    assert line.replace("$(ARCH)", arch).replace("$(EXTRA)", "").split() == SHIPPING_CXXFLAGS.split()


def test_backward_kernel_vmcnt_ledger_matches_its_isa(tmp_path):
    """The dX-chain kernel's ring syncs count compiler-issued stores into their vmcnt waits
    (csrc/pipeline.h LEDGER).  Replay the compiled instruction stream and check that no sync
    publishes a weight block whose DMA could still be in flight."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_vmcnt
    asm = device_isa("mlp_bwd_s16", tmp_path)
    # one instantiation per supported encoding pair; fragments = LayoutB<KE, KD>::F_END, 16 per ring block
    # (the trailing template argument: with / without the encoding products, i.e. ray gradients asked for or not -- the
    # skipped fragments keep their syncs and DMA sites, pipeline.h skip_frags)
    for tag, frags in ((("mlp_bwd_s16_kernelILi10ELi4E", "EEELb1EEEv"), 1184), (("mlp_bwd_s16_kernelILi15ELi6E", "EEELb1EEEv"), 1224),
                       (("mlp_bwd_s16_kernelILi10ELi4E", "EEELb0EEEv"), 1184), (("mlp_bwd_s16_kernelILi15ELi6E", "EEELb0EEEv"), 1224)):
        stats = check_vmcnt.check(asm, tag, verbose=False)
        assert stats["kernels"] == 1 and stats["ok"], tag
        assert stats["syncs"] == -(-frags // 16) and stats["dma_pieces"] == 2 * -(-frags // 16), (tag, stats)
        # the checker must be able to fail: claim more stores than the ISA has at every sync
        assert not check_vmcnt.check(asm, tag, verbose=False, slack=-4)["ok"]
    # the split-precision dX chain (mlp_bwd_split.hip): fragments = LayoutBS<KE, KD>::F_END, the same ring
    asm = device_isa("mlp_bwd_split", tmp_path)
    for tag, frags in ((("mlp_bwd_split_kernelILi10ELi4ELb1E", "EEELb1EEEv"), 2368), (("mlp_bwd_split_kernelILi15ELi6ELb1E", "EEELb1EEEv"), 2448),
                       (("mlp_bwd_split_kernelILi10ELi0ELb0E", "EEELb1EEEv"), 1952), (("mlp_bwd_split_kernelILi10ELi4ELb1E", "EEELb0EEEv"), 2368),
                       (("mlp_bwd_split_kernelILi15ELi6ELb1E", "EEELb0EEEv"), 2448)):
        stats = check_vmcnt.check(asm, tag, verbose=False)
        assert stats["kernels"] == 1 and stats["ok"], (tag, stats)
        assert stats["syncs"] == -(-frags // 16) and stats["dma_pieces"] == 2 * -(-frags // 16), (tag, stats)
        assert not check_vmcnt.check(asm, tag, verbose=False, slack=-4)["ok"]


def test_field_kernel_keeps_its_weight_read_ahead(tmp_path):
    """The fused kernels read every weight fragment from LDS through a software queue, several MFMAs ahead of its use.
    The machine scheduler once sank half of those reads back to their first use (38 % with zero MFMAs in between),
    which no test of the RESULTS can see; csrc pins them with scheduling groups (mlp_bf16_s16.hip sched_step).  Check the
    compiled ISA of the default inference kernel: nearly every fragment read is issued >= 4 MFMAs before its MFMA."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import isa_readahead
    asm = device_isa("mlp_bf16_s16", tmp_path)
    for tag, n_mfma in (("mlp_bf16_s16p_kernelILi10ELi4ELb1E", 2344),):
        hist = isa_readahead.main(asm, tag)
        total = sum(hist.values())
        assert total == n_mfma // 2, (tag, total)                  # one read per weight fragment, two MFMAs per fragment
        ahead = sum(v for d, v in hist.items() if d >= 4)
        assert ahead >= 0.97 * total, (tag, dict(hist))
        assert hist.get(0, 0) <= 0.01 * total, (tag, dict(hist))
    # The counted vmcnt waits of the ring syncs, replayed on the same ISA for both halves of the workgroup (SPLIT_DMA
    # selects the DMA site per half inside the asm statement; the DMA is buffer_load ... lds): no sync may publish a block
    # whose pieces can still be in flight.  Straight-line kernels only: the per-tile inference kernel (every architecture)
    # and the training forward; the pipelined kernel picks one count at run time (has_next) and is covered by bit-exact
    # comparison with the per-tile kernel instead (tools/mlp_ab.py, the GPU parity tests).
    import check_vmcnt
    for tag, syncs in (("mlp_bf16_s16_kernelILi10ELi4ELb1ENS_3CtxILi8ELi16ELi4ELi8ELi4ELi0ELi1ELi76", 74),
                       ("mlp_bf16_s16_kernelILi15ELi6ELb1ENS_3CtxILi8ELi16ELi4ELi8ELi4ELi0ELi1ELi76", 76),
                       ("mlp_bf16_s16_kernelILi10ELi0ELb0ENS_3CtxILi8ELi16ELi4ELi8ELi4ELi0ELi1ELi76", 61),
                       ("mlp_bf16_s16_kernelILi10ELi4ELb1ENS_3CtxILi8ELi16ELi4ELi8ELi2ELi0ELi1ELi0ENS_8NoLedgerEEELb1", 74)):
        for lag in (0, 2):                    # the two DMA issue phases
            st = check_vmcnt.check(asm, tag, verbose=False, lag=lag)
            # (+ 1: the end-of-tile drain, `s_waitcnt vmcnt(0)`, where it sits right in front of the barrier that publishes the
            # next tile's ticket -- a wait for everything, which the replay accepts like any other sync)
            assert st["kernels"] == 1 and st["ok"] and st["syncs"] - syncs in (0, 1) and st["dma_pieces"] == 2 * syncs, (tag, lag, st)
            assert not check_vmcnt.check(asm, tag, verbose=False, lag=lag, slack=3)["ok"]      # the checker can fail


def test_bench_reports_traffic_only_for_the_build_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a profiles/ PMC summary; it must not survive a kernel change."""
    sys.path.insert(0, REPO)
    import bench
    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    grids = {"131072": {"FETCH_SIZE": {"launches": 4, "mean_per_launch": 100.0}, "WRITE_SIZE": {"launches": 4, "mean_per_launch": 50.0},
                        "MfmaUtil": {"launches": 2, "mean_per_launch": 70.0},
                        "hbm_bytes_per_launch": {"read_corrected_x2": 204800.0, "write": 51200.0, "total": 256000.0}}}
    import json
    (prof / "r99_pmc_summary_x.json").write_text(json.dumps({"_meta": {"csrc_hash": h}, "void na::mlp_bf16_s16p_kernel<10>": grids}))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_hash", lambda: h)
    got = bench.measured_pmc("mlp_bf16_s16p_kernel")
    assert got == {"traffic": 256000.0, "mfma_util": 70.0, "file": "r99_pmc_summary_x.json"}
    monkeypatch.setattr(bench, "csrc_hash", lambda: "0" * 16)           # the kernels changed: the stale figure is dropped
    assert bench.measured_pmc("mlp_bf16_s16p_kernel") is None


def test_bench_launches_its_own_ranks_and_fails_cleanly_without_a_gpu():
    """`python bench.py --gpus 2` started plainly (the form the driver records) must itself start two ranks under
    torch.distributed.run -- from a parent that never touches the GPU -- and, with no ROCm device visible, every
    rank refuses loudly and the launcher's exit status is non-zero with no result line on stdout."""
    import subprocess
    env = dict(os.environ, NERF_AMD_DIST_BACKEND="gloo", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""                                   # no JSON line from a failed run
    assert "rank 0/2: needs a ROCm device" in p.stderr or "rank 1/2: needs a ROCm device" in p.stderr, p.stderr[-2000:]
    assert "the 2-rank run failed" in p.stderr
    # a rank count that contradicts the environment is refused as well (a stale WORLD_SIZE, a wrong --nproc-per-node)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"],
                       env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_bench_counts_the_cores_it_is_granted(monkeypatch):
    """cpu_baseline uses every core granted to the process: the affinity mask capped by a cgroup quota."""
    sys.path.insert(0, REPO)
    import bench
    cores, info = bench.granted_cores()
    assert 1 <= cores <= info["affinity"] == len(os.sched_getaffinity(0))
    assert info["cgroup_quota"] is None or cores <= info["cgroup_quota"] + 1


def test_png_codec_roundtrip_and_async_writer(tmp_path):
    """image_io: the zlib PNG writer/reader that replaces imageio for the render output and the
    dataset frames (render_utils.py:312-315, load_blender.py:69)."""
    from nerf_shared_amd import image_io
    rng = np.random.default_rng(0)
    for shape in ((5, 7), (9, 4, 1), (16, 11, 3), (8, 8, 4)):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        back = image_io.decode_png(image_io.encode_png(img))
        assert np.array_equal(back, img.reshape(back.shape))
    with pytest.raises(TypeError):
        image_io.encode_png(np.zeros((4, 4, 3), np.float32))
    # a smooth image makes Pillow's encoder choose the Sub/Up/Average/Paeth filters: decode them all
    try:
        from PIL import Image
    except ImportError:
        Image = None
    yy, xx = np.mgrid[0:64, 0:48]
    smooth = np.stack([(xx * 5) % 256, (yy * 3 + xx) % 256, (xx * yy) % 256, 255 - xx], -1).astype(np.uint8)
    if Image is not None:
        path = str(tmp_path / "pil.png")
        Image.fromarray(smooth).save(path, optimize=True)
        data = open(path, "rb").read()
        assert np.array_equal(image_io.decode_png(data), smooth)
        assert np.array_equal(image_io.read_image(path), smooth)
        own = str(tmp_path / "own.png")
        image_io.write_png(own, smooth[..., :3])
        assert np.array_equal(np.asarray(Image.open(own)), smooth[..., :3])
    # asynchronous writer: many frames, results identical to the synchronous path, errors surface
    frames = [rng.integers(0, 256, size=(40, 30, 3), dtype=np.uint8) for _ in range(12)]
    with image_io.AsyncImageWriter(workers=3) as w:
        for i, f in enumerate(frames):
            w.submit(str(tmp_path / ("%03d.png" % i)), f)
    for i, f in enumerate(frames):
        assert np.array_equal(image_io.read_image(str(tmp_path / ("%03d.png" % i))), f)
    w = image_io.AsyncImageWriter(workers=1)
    w.submit(str(tmp_path / "missing_dir" / "x.png"), frames[0])
    with pytest.raises(OSError):
        w.close()


def test_packed_weights_go_stale_after_optimizer_steps_and_copies_get_their_own_handle():
    """The device-side packed weights are keyed on (data_ptr, _version); fused optimizers do not bump
    _version, so any optimizer step must invalidate the key.  Copies must not share a library handle."""
    import copy
    m = nerf.NeRF(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
    other = nerf.NeRF(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
    m._packed_key, other._packed_key = "packed", "packed"
    lin = torch.nn.Linear(2, 2)                       # an optimizer over unrelated parameters leaves the models alone
    unrelated = torch.optim.SGD(lin.parameters(), lr=0.1)
    lin.weight.grad, lin.bias.grad = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
    unrelated.step()
    assert m._packed_key == "packed" and other._packed_key == "packed"
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    for p_ in m.parameters():
        p_.grad = torch.zeros_like(p_)
    opt.step()
    assert m._packed_key is None and other._packed_key == "packed"
    m._packed_key = "packed"
    m.weights_changed()
    assert m._packed_key is None
    m._handle, m._packed_key = "library handle", "packed"
    c = copy.deepcopy(m)
    assert c._handle is None and c._packed_key is None and m._handle == "library handle"
    c._packed_key = "packed"
    copt = torch.optim.SGD(c.parameters(), lr=0.0)
    for p_ in c.parameters():
        p_.grad = torch.zeros_like(p_)
    copt.step()
    assert c._packed_key is None                      # the copy is tracked too
    m._handle = None
    assert torch.equal(c.pts_linears[3].weight, m.pts_linears[3].weight)


def test_library_adam_is_a_torch_adam_and_survives_copies():
    """nerf_shared_amd.optim.Adam keeps torch.optim.Adam's surface (constructor, param_groups, state_dict keys, pickling);
    what its kernel does not cover is refused at construction or at step time, never silently handled elsewhere."""
    import copy
    import pickle

    from nerf_shared_amd import _lib, optim
    p = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.ones(2, 3))]
    o = optim.Adam(p, lr=5e-4, betas=(0.9, 0.999))
    assert isinstance(o, torch.optim.Adam)
    ref = torch.optim.Adam([torch.nn.Parameter(t.detach().clone()) for t in p], lr=5e-4, betas=(0.9, 0.999))
    assert {k: v for k, v in o.param_groups[0].items() if k not in ("params", "foreach", "fused")} == \
           {k: v for k, v in ref.param_groups[0].items() if k not in ("params", "foreach", "fused")}
    assert sorted(o.state_dict()) == sorted(ref.state_dict())
    for clone in (copy.deepcopy(o), pickle.loads(pickle.dumps(o))):
        assert isinstance(clone, optim.Adam) and clone._together == {} and clone.param_groups[0]["lr"] == 5e-4
    with pytest.raises(_lib.NerfAmdError):
        optim.Adam(p, amsgrad=True)
    p[0].grad = torch.ones(4)
    with pytest.raises(_lib.NerfAmdError):            # host tensors: no CPU path behind the library optimizer
        o.step()


def test_deferred_gradient_trap_is_plain_autograd():
    """nerf.attach_deferred_grad (what a model outside the training kernels returns when gradients were requested): the
    values pass through untouched, the result has history, and backward raises NerfAmdError with the stored reason -- on
    any device, tensors or dicts of tensors."""
    from nerf_shared_amd import nerf
    x = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    anchor = torch.ones(1, requires_grad=True)
    out = nerf.attach_deferred_grad({"rgb_map": x, "acc_map": x[:, 0]}, (anchor, "no backward kernels for this model"))
    assert sorted(out) == ["acc_map", "rgb_map"] and torch.equal(out["rgb_map"], x)
    assert out["rgb_map"].requires_grad and out["acc_map"].grad_fn is not None
    with pytest.raises(_lib.NerfAmdError, match="no backward kernels for this model"):
        (out["rgb_map"].sum() + out["acc_map"].sum()).backward()
    assert anchor.grad is None
    assert nerf.attach_deferred_grad(x, None) is x                     # nothing requested: nothing attached


def test_load_weights_from_keras_maps_the_list_like_the_reference():
    """nerf.py:146-173: kernels [in, out] transposed, list order pts_linears (2 per layer), feature, views, rgb, alpha;
    only for the view-branch model (the reference asserts)."""
    import numpy as np
    import torch
    from nerf_shared_amd import nerf, synth
    arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True)
    sd = synth.make_state_dict(3, 1.0, **{**arch, "skips": (4,)})
    order = ["pts_linears.%d" % i for i in range(8)] + ["feature_linear", "views_linears.0", "rgb_linear", "alpha_linear"]
    keras = []
    for name in order:
        keras += [np.ascontiguousarray(sd[name + ".weight"].T), sd[name + ".bias"].copy()]
    m = nerf.NeRF(**arch)
    m.load_weights_from_keras(keras)
    got = m.state_dict()
    for name in order:
        assert torch.equal(got[name + ".weight"], torch.from_numpy(sd[name + ".weight"])), name
        assert torch.equal(got[name + ".bias"], torch.from_numpy(sd[name + ".bias"])), name
    assert m._packed_key is None
    with pytest.raises(AssertionError):
        nerf.NeRF(D=8, W=256, use_viewdirs=False).load_weights_from_keras(keras)


class _ReferenceStyleNeRF(torch.nn.Module):
    """What nerf_shared.nerf.NeRF.__init__ builds (nerf.py:62-94): the attributes and nn.Linear layout, nothing of its code."""

    def __init__(self, D=8, W=256, output_ch=4, skips=(4,), use_viewdirs=False, multires=10, multires_views=4):
        super().__init__()
        self.D, self.W, self.skips, self.use_viewdirs = D, W, list(skips), use_viewdirs
        self.embed_fn, self.input_ch = None, 3 + 6 * multires
        self.input_ch_views, self.embeddirs_fn = (3 + 6 * multires_views if use_viewdirs else 0), None
        lin = torch.nn.Linear
        self.pts_linears = torch.nn.ModuleList([lin(self.input_ch, W)] + [lin(W, W) if i not in self.skips else lin(W + self.input_ch, W)
                                                                         for i in range(D - 1)])
        self.views_linears = torch.nn.ModuleList([lin(self.input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear, self.alpha_linear, self.rgb_linear = lin(W, W), lin(W, 1), lin(W // 2, 3)
        else:
            self.output_linear = lin(W, output_ch)


def test_reference_class_models_are_adopted_by_sharing_their_parameters():
    """Renderer takes models built by the reference's own NeRF class (main.py keeps `from nerf_shared import nerf`): nerf.adopt
    makes a twin whose parameters ARE the model's Parameter objects; the model's state_dict keys stay the reference's."""
    for kw in (dict(use_viewdirs=True, output_ch=5), dict(use_viewdirs=False, output_ch=5), dict(use_viewdirs=True, multires=15, multires_views=6)):
        ref = _ReferenceStyleNeRF(**kw)
        keys = list(ref.state_dict().keys())
        twin = nerf.adopt(ref)
        assert isinstance(twin, nerf.NeRF) and nerf.adopt(ref) is twin and nerf.adopt(twin) is twin
        assert list(ref.state_dict().keys()) == keys and list(twin.state_dict().keys()) == keys
        assert all(a is b for a, b in zip(twin.parameters(), ref.parameters()))
        assert (twin.D, twin.W, twin.use_viewdirs, twin.input_ch, twin.input_ch_views) == (ref.D, ref.W, ref.use_viewdirs, ref.input_ch, ref.input_ch_views)
        assert twin.output_ch == (4 if kw["use_viewdirs"] else 5) or kw["use_viewdirs"]
    with pytest.raises(TypeError, match="reference NeRF's attributes"):
        nerf.adopt(torch.nn.Linear(3, 3))
    bad = _ReferenceStyleNeRF(use_viewdirs=True)
    bad.rgb_linear = torch.nn.Linear(128, 4)
    with pytest.raises(TypeError, match="layer shapes"):
        nerf.adopt(bad)
    assert nerf.adopt(None) is None
