"""bench.py as the driver starts it (-m gpu): `python bench.py --gpus N`, a plain process that launches its own ranks.

  * N = 1 with NERF_AMD_FORCE_COLLECTIVE=1: one rank under torch.distributed.run on the `nccl` backend (RCCL), the C5
    workload, so the side-stream gather of dist.OverlappedGather (async collective under a stream context, stream-level
    wait, events) executes on real hardware every round -- with a world of one, which is all a 1-GPU box offers.
  * N = 2 with NERF_AMD_DIST_BACKEND=gloo: two ranks time-sharing this GPU (gloo stands in for RCCL, which refuses two
    ranks on one device): the launcher, the rendezvous, the pixel-range shards, the gather and the max-over-ranks timing.
  * N = 2 on RCCL with one visible GPU: refused by every rank before any collective, launcher exits non-zero.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *argv, timeout=420):
    env = dict(os.environ, NERF_AMD_QUIET="1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(argv), env=env, capture_output=True,
                       text=True, timeout=timeout)
    return p


def result_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_self_launched_rccl_world_of_one_runs_the_c5_gather():
    assert torch.cuda.is_available()
    p = run_bench({"NERF_AMD_FORCE_COLLECTIVE": "1"}, "--gpus", "1", "--workload", "c5", "--steps", "3", "--warmup", "1",
                  "--no-subrecords", "--no-cpu-baseline")
    out = result_line(p)
    assert out["n_gpus"] == 1 and out["dist"]["backend"] == "nccl" and out["dist"]["rccl_world"] == 1
    assert len(out["dist"]["rank_devices"]) == 1 and "cuda:0" in out["dist"]["rank_devices"][0]
    assert out["config"]["workload"].startswith("lego_fullres_800x800") and out["config"]["rays_per_step"] == 640000
    assert out["value"] > 1e6 and out["roofline"]["launches"] > 0
    assert out["dist"]["gathers"] == 3 and out["dist"]["gather_path"] == "collective on a side stream"
    # the run explains itself: every rank's render time per frame, the part of the gathers its renders did not hide, what
    # rank 0 received, and which rank finished last
    pr = out["dist"]["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["frames"] == 3 and out["dist"]["slowest_rank"] == 0
    assert 50.0 < pr[0]["render_ms_per_frame"] < 400.0 and 0.0 <= pr[0]["exposed_gather_ms_per_frame"] < 20.0
    assert pr[0]["recv_bytes_per_frame"] == 0 and pr[0]["own_ms_per_frame"] >= pr[0]["render_ms_per_frame"] * 0.9


def test_self_launched_two_ranks_share_the_gpu_over_gloo():
    p = run_bench({"NERF_AMD_DIST_BACKEND": "gloo"}, "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    out = result_line(p)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["dist"]["backend"] == "gloo" and out["dist"]["rccl_world"] is None and len(out["dist"]["rank_devices"]) == 2
    assert out["config"]["rays_per_step"] == 640000 and out["value"] > 1e6
    assert out["frames_round_robin"]["steps"] == 4 and out["frames_round_robin"]["value"] > 1e6
    pr = out["dist"]["per_rank"]
    assert [d["rank"] for d in pr] == [0, 1] and all(d["frames"] == 4 for d in pr) and out["dist"]["slowest_rank"] in (0, 1)
    assert pr[0]["recv_bytes_per_frame"] == 320000 * 20 and pr[1]["recv_bytes_per_frame"] == 0


def test_more_rccl_ranks_than_gpus_is_refused_before_any_collective():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    p = run_bench({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "2 ranks over RCCL need 2 GPUs, 1 visible" in p.stderr


def test_default_line_carries_the_roofline_the_cpu_baseline_and_a_meaningful_psnr_check():
    """`python bench.py` as the driver runs it at N = 1 (short: 3 steps, no sub-records): one JSON line with the contract's
    keys, `roofline` and `cpu_baseline`; and `cpu_baseline.check` -- the metric's "PSNR vs ref" -- on weights whose field has
    content (an all-white view would score infinity whatever the kernels do): fp32-class modes above 50 dB against the CPU
    image, bf16 above 33 dB."""
    p = run_bench({}, "--steps", "3", "--warmup", "1", "--no-subrecords")
    line = result_line(p)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["dtype"] == "bf16" and line["value"] > 1e6
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and 0.3 < rf["frac"] < 1.0 and rf["launches"] == 30
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 50 and cb["cores"] >= 1 and "sample" in cb
    chk = cb["check"]
    assert chk["rays"] == 4096 and chk["ref_rgb_variance"] > 1e-2, chk
    m = chk["modes"]
    assert m["fp32"]["psnr_db"] > 50 and m["fp32_split"]["psnr_db"] > 50 and m["bf16"]["psnr_db"] > 33, m


_RCCL_WORLD1 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
os.environ["NERF_AMD_QUIET"] = "1"
import numpy as np
from nerf_shared_amd import dist as nd, nerf, render_utils, synth
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
arch = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
ms = []
for seed in (1, 19):
    m = nerf.NeRF(**arch); m.load_state_dict(synth.torch_state_dict(seed, 3.0, **{**arch, "skips": (4,)})); ms.append(m.to(dev).requires_grad_(False))
r = render_utils.Renderer(perturb=0.0, N_importance=32, N_samples=32, use_viewdirs=True, white_bkgd=True, near=2.0, far=6.0)
K = synth.lego_intrinsics(40, 40)
c2w = torch.from_numpy(synth.LEGO_C2W)
with torch.no_grad():
    img8 = nd.render_image_sharded(r, 40, 40, K, c2w, ms[0], ms[1], chunk=700, as_uint8=True)
    rgb, disp, acc = nd.render_image_sharded(r, 40, 40, K, c2w, ms[0], ms[1], chunk=700)
    want = r.render(40, 40, K, ms[0], ms[1], chunk=700, c2w=c2w, retraw=False)[0]
assert img8.dtype == torch.uint8 and img8.shape == (40, 40, 3)
assert torch.equal(rgb, want)
from nerf_shared_amd import utils
assert torch.equal(img8, utils.to8b(want))
# broadcast_parameters on the RCCL backend (a world of one returns early; force the collective by hand as the function does)
flat = torch.cat([p.detach().reshape(-1) for p in ms[0].parameters()])
before = flat.clone()
dist.broadcast(flat, src=0)
torch.cuda.synchronize()
assert torch.equal(flat, before)
nd.broadcast_parameters(ms, src=0)
# the uint8 gather path with the collective forced in a one-rank group
os.environ["NERF_AMD_FORCE_COLLECTIVE"] = "1"
g = nd.OverlappedGather(1600, 0, None, 2)
rows8 = utils.to8b(want.reshape(-1, 3))
g.submit(rows8)
back = g.collect()[0]
torch.cuda.synchronize()
assert back.dtype == torch.uint8 and torch.equal(back, rows8)
print("RCCL_WORLD1_OK", dist.get_backend(), dist.get_world_size())
dist.destroy_process_group()
"""


def test_rccl_world_of_one_sharded_image_uint8_and_parameter_broadcast(tmp_path):
    """render_image_sharded (float and as_uint8=True), broadcast_parameters and a uint8 OverlappedGather on the RCCL backend
    with a world of one -- the multi-GPU entry points the C5 bench line does not pass through."""
    script = tmp_path / "rccl1.py"
    script.write_text(_RCCL_WORLD1 % {"repo": REPO})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK nccl 1" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
