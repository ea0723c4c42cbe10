#!/usr/bin/env python3
"""bench.py -- rays/sec of the render hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4|c5] [--precision bf16|fp32]

One "step" = one call of Renderer.render(H, W, K, coarse, fine, chunk, c2w=pose, retraw=False)
(reference render_utils.py:176-238: rays from the pose, batch assembly, the chunk loop, both field
passes, compositing, resampling), end to end, everything resident in HBM: the models' packed
weights before the timed region, the maps after it.  Random-init 8x256 viewdirs fields (seeds 0/10),
synthetic Lego camera, perturb=1 (stratified jitter + random sample_pdf draws,
the reference's own default and what SURVEY.md section 8d prescribes for throughput runs).

N = 1 (the default, what BENCH_rNN records): BASELINE.json configs[1], the configuration the metric
is quoted on -- Lego half-res 400x400 = 160 000 rays, 64 coarse + 128 fine samples, 4096-ray chunks.
The line also carries sub-records measured in the same process after the main timed region:
  perturb0  the same workload with perturb=0 (the deterministic parity configuration)
  fp32      the same workload on the exact-fp32 kernel (rays/s, TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak)
  c5_n1     the multi-GPU workload (below) on this one GPU: the N=1 point of the scaling curve

N > 1 (torch.distributed.run, one rank per GPU, RCCL): BASELINE.json configs[4] = C5, STRONG scaling.
A step is one 800x800 frame (640 000 rays, 64+128, chunk 32768) of a fixed list of 200 poses on the
camera circle (synth.circle_poses(200)); rank r renders flat pixel range r of every frame and the finished
[rays, 5] rows are gathered to rank 0 -- the gather of frame k runs on a side stream under the render of
frame k+1 (nerf_shared_amd.dist.OverlappedGather); all gathers are complete when the clock stops.
Sub-record frames_round_robin: the same K frames dealt whole to the ranks (rank r renders frames
r, r+N, ...; no data-path collective, the frames stay where they were rendered).

The JSON line also carries
  roofline     -- the field-MLP kernel: algorithmic FLOP (1 186 816 per point) / device
                  time from hipEvents recorded around every launch on its stream (nerf_amd_profile_*),
                  against the 2.5 PFLOP/s dense bf16 MFMA peak.  `traffic` (HBM bytes per launch from
                  rocprofv3 --pmc passes) is reported only from a profiles/ summary made from THIS build
                  of the kernels (csrc hash match), else null.
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference, "port") on this box's host cores:
                  all cores granted to the process and 1 thread, on bounded samples (N = 1 only).  Its `check`
                  entry is the metric's "PSNR vs ref": the rays of the 4096-ray batch the CPU leg timed, rendered by the
                  bf16 / fp32_split / fp32 modes and compared with the CPU image (PSNR, max |d rgb|, max |d acc|)
                  on weights whose field has content; for the fp32-class modes also the per-ray attribution of
                  tools/precision_census.py (frac_gt_1e-3, unexplained, staged maxima).
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
os.environ.setdefault("NERF_AMD_QUIET", "1")

FLOP_PER_POINT = 1186816            # 2 x 593 408 MAC, viewdirs 8x256 MLP
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3            # fp32 MFMA peak
ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
N_POSES_C5 = 200

# BASELINE.json configs; c2 is the one the metric is quoted on (the default, and what the driver runs at N=1),
# c5 the multi-GPU one (the default at N>1).
WORKLOADS = {
    "c1": dict(name="lego_halfres_400x400_64c_coarse_only_chunk32768", H=400, W=400, chunk=32768, Nc=64, Ni=0),
    "c2": dict(name="lego_halfres_400x400_64c+128f_viewdirs_8x256_chunk4096", H=400, W=400, chunk=4096, Nc=64, Ni=128),
    "c3": dict(name="lego_fullres_800x800_64c+128f_viewdirs_8x256_chunk32768", H=800, W=800, chunk=32768, Nc=64, Ni=128),
    "c4": dict(name="fern_llff_378x504_ndc_64c+64f_viewdirs_8x256_chunk32768", H=378, W=504, chunk=32768, Nc=64, Ni=64,
               ndc=True, near=0.0, far=1.0, white_bkgd=False, focal=408.0, raw_noise_std=1.0),
    "c5": dict(name="lego_fullres_800x800_64c+128f_200_pose_testset_chunk32768", H=800, W=800, chunk=32768, Nc=64, Ni=128,
               poses=N_POSES_C5),
}


def renderer_cfg(w, perturb):
    return dict(perturb=perturb, N_importance=w["Ni"], N_samples=w["Nc"], use_viewdirs=True,
                white_bkgd=w.get("white_bkgd", True), raw_noise_std=w.get("raw_noise_std", 0.0) if perturb else 0.0,
                ndc=w.get("ndc", False), lindisp=False, near=w.get("near", 2.0), far=w.get("far", 6.0))


def camera(w, synth):
    """(K, [poses]) of a workload: the Lego test pose, the Fern-like forward-facing pose, or the 200-pose circle."""
    import numpy as np
    K = synth.lego_intrinsics(w["H"], w["W"])
    poses = [synth.LEGO_C2W]
    if w.get("focal"):      # forward-facing LLFF-style camera
        K[0][0] = K[1][1] = w["focal"]
        poses = [np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)]
    if w.get("poses"):
        poses = synth.circle_poses(w["poses"])
    return K, poses


CHECK_SEEDS, CHECK_SHARPEN = (1, 19), 3.0      # weights of cpu_baseline.check: a field with content (tests' referee set c19)

FIELD_KERNEL_SOURCES = ("mlp_bf16_s16.hip", "mlp_bf16.hip", "mlp_fp32.hip", "mlp_split.hip", "pipeline.h", "program.h", "program.cpp",
                        "pack.hip", "kernels.h", "launch_util.h", "Makefile")


def csrc_hash():
    """Identity of the field-kernel sources a profile was made from (profiles/*pmc_summary*.json carry it): the
    translation units and headers the inference field kernels and their weight packing are compiled from."""
    h = hashlib.sha256()
    root = os.path.join(REPO, "nerf_shared_amd", "csrc")
    for name in FIELD_KERNEL_SOURCES:
        path = os.path.join(root, name)
        if os.path.exists(path):
            with open(path, "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def measured_pmc(kernel_tag):
    """Counters of the dominant kernel from committed rocprofv3 --pmc passes (tools/pmc_summary.py), only if that
    summary was made from the kernels this process runs (same csrc hash) -- PMC counters cannot be read inside the
    timed run.  Returns {"traffic": HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md),
    "mfma_util": MfmaUtil in percent, "file": ...} averaged over the kernel's launches, or None."""
    want = csrc_hash()
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("_meta", {}).get("csrc_hash") != want:
            continue
        for name, grids in d.items():
            if name == "_meta" or kernel_tag not in name:
                continue
            tot = n = 0
            util = un = 0
            for g in grids.values():
                if "hbm_bytes_per_launch" in g:
                    k = g["FETCH_SIZE"]["launches"]
                    tot += g["hbm_bytes_per_launch"]["total"] * k
                    n += k
                if "MfmaUtil" in g:
                    k = g["MfmaUtil"]["launches"]
                    util += g["MfmaUtil"]["mean_per_launch"] * k
                    un += k
            if n:
                return {"traffic": tot / n, "mfma_util": (util / un) if un else None, "file": os.path.basename(path)}
    return None


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def granted_cores():
    """Host cores this process may actually use: the scheduler affinity, capped by a cgroup CPU quota if there is one.
    Returns (cores, {"affinity": .., "cgroup_quota": .., "visible": ..})."""
    import math
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            with open(path) as f:
                quota = parse(f.read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    cores = aff if not quota else max(1, min(aff, int(math.ceil(quota))))
    return cores, {"affinity": aff, "cgroup_quota": quota, "visible": os.cpu_count()}


def cpu_baseline(torch, synth, w, full_c1=True):
    """The oracle on host cores (rank 0, N=1), bounded to about a minute.  C2: one 4096-ray 64+128 batch on ALL cores
    granted to the process (1 warm-up + best of 2; when more than 16 are granted the same batch is also timed on 16
    threads and the better rate is the value, with the thread count that produced it in `cores`), and a 512-ray
    batch on 1 thread.  `c1` = BASELINE.json configs[0], the reference's own CPU-runnable case: the full 400x400
    coarse-only view, chunk 32768, wall clock (BASELINE.md "CPU-baseline plan")."""
    from oracle import nerf_oracle as O
    import numpy as np
    granted, info = granted_cores()
    K, poses = camera(w, synth)
    H, W = w["H"], w["W"]
    models = []
    for seed in (0, 10):
        sd = synth.make_state_dict(seed, 1.0, **{**ARCH, "skips": (4,)})
        models.append((O.state_dict_to_torch(sd), O.Arch(**ARCH)))
    cfg = O.RenderCfg(**renderer_cfg(w, 0.0))
    fine = models[1] if w["Ni"] > 0 else None
    out = {"unit": "rays/s", "kind": "port", "cpu_model": cpu_model_name(), "host_cores_visible": os.cpu_count(),
           "cores_granted": granted, "cores_detail": info}

    def full_view_c1(threads):
        wc = WORKLOADS["c1"]
        Kc, pc = camera(wc, synth)
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        O.render(O.RenderCfg(**renderer_cfg(wc, 0.0)), wc["H"], wc["W"], Kc, models[0], None, chunk=wc["chunk"],
                 c2w=torch.from_numpy(pc[0]), retraw=False)
        dt = time.perf_counter() - t0
        return {"value": wc["H"] * wc["W"] / dt, "unit": "rays/s", "cores": threads, "workload": wc["name"],
                "sample": "one full %dx%d coarse-only view (64 samples), chunk %d, torch-CPU oracle fp32, wall clock %.2f s"
                          % (wc["H"], wc["W"], wc["chunk"], dt)}

    with torch.no_grad():
        if w["Ni"] == 0:          # C1 as the workload: one whole coarse-only view
            out.update(full_view_c1(granted))
            return out

        def timed(n_rays, n_threads, reps):
            torch.set_num_threads(n_threads)
            ro, rd = synth.rays_np(H, W, K, poses[0], np.arange(H * W // 2, H * W // 2 + n_rays))
            batch = torch.from_numpy(synth.ray_batch_np(ro, rd, cfg.near, cfg.far, True))
            best = float("inf")
            for i in range(reps + 1):
                t0 = time.perf_counter()
                O.render_rays(cfg, batch, models[0], fine)
                if i > 0:
                    best = min(best, time.perf_counter() - t0)
            return n_rays / best, best
        tries = {}
        for threads in sorted({granted, min(granted, 16)}, reverse=True):
            v, best = timed(4096, threads, 2)
            tries[threads] = {"value": v, "seconds": best}
        best_threads = max(tries, key=lambda t: tries[t]["value"])
        out.update(value=tries[best_threads]["value"], cores=best_threads, tried={str(k): v for k, v in tries.items()},
                   sample="one 4096-ray batch, %d+%d samples, torch-CPU oracle fp32, best of 2 after warm-up (%.2f s)"
                          % (w["Nc"], w["Ni"], tries[best_threads]["seconds"]))
        v1, best1 = timed(512, 1, 1)
        out["single_thread"] = {"value": v1, "cores": 1,
                                "sample": "one 512-ray batch, same workload, 1 torch thread, after warm-up (%.2f s)" % best1}
        if full_c1:
            out["c1"] = full_view_c1(best_threads)
        torch.set_num_threads(best_threads)
        # The referee image for main's `check`: the same 4096 rays on weights whose field has CONTENT (the timed workload's
        # default-scale random weights render an empty, all-white view -- every mode would agree with it to the last bit):
        # the x3-sharpened sets of the PSNR referee in tests/ (coarse seed 1, fine seed 19: an opaque object).
        ro, rd = synth.rays_np(H, W, K, poses[0], np.arange(H * W // 2, H * W // 2 + 4096))
        batch = torch.from_numpy(synth.ray_batch_np(ro, rd, cfg.near, cfg.far, True))
        ref_models = [(O.state_dict_to_torch(synth.make_state_dict(sd, CHECK_SHARPEN, **{**ARCH, "skips": (4,)})), O.Arch(**ARCH))
                      for sd in CHECK_SEEDS]
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import precision_census as PC                      # measurement infrastructure, like the oracle it drives
        res = PC.oracle_stages(cfg, batch, ref_models[0], ref_models[1])
        out["_check"] = (batch, res, ref_models, cfg)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp32_split"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true")
    ap.add_argument("--perturb", type=float, default=1.0)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly (no RANK in the environment): this process becomes a launcher that never
    touches the GPU.  It starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on this same file as a
    fresh child (one rank per GPU, RCCL), relays rank 0's single JSON line, and returns the child's exit status
    (non-zero if any rank failed).  Nothing is re-executed in a process that has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:               # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in proc.stdout:                 # rank 0 prints exactly one JSON line; anything else goes to stderr untouched
        if line.startswith("{") and '"metric"' in line:
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed %d result lines\n" % lines)
        rc = 1
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank run failed (torch.distributed.run exit status %d)\n" % (args.gpus, rc))
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # Under torch.distributed.run RANK/WORLD_SIZE are set and this process is one rank.  Started plainly with
    # --gpus N > 1 (how the driver records its command), or with NERF_AMD_FORCE_COLLECTIVE=1 at N = 1 (the RCCL
    # world-1 smoke of the C5 path), this process only launches the ranks -- before anything touches the GPU.
    if "RANK" not in os.environ and (args.gpus > 1 or os.environ.get("NERF_AMD_FORCE_COLLECTIVE") == "1"):
        sys.exit(spawn_ranks(args))
    from nerf_shared_amd import render_utils as _ru
    if os.environ.get("NERF_AMD_OVERLAP_CHUNKS") == "1":      # A/B knob: two-stream chunk pipeline of render_batch
        _ru.Renderer.overlap_chunks = True
    if os.environ.get("NERF_AMD_FUSE_CHUNKS") == "0":         # A/B knob: one render_rays call per chunk
        _ru.Renderer.fuse_chunk_launches = False

    import torch
    import torch.distributed as dist
    from nerf_shared_amd import _lib, dist as nd, nerf, render_utils, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py rank %d: --gpus %d but WORLD_SIZE=%d (start it plainly, or under torch.distributed.run "
                         "with --nproc-per-node %d)" % (rank, args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py rank %d/%d: needs a ROCm device; there is no CPU path" % (rank, world))
    key = args.workload or ("c5" if world > 1 else "c2")
    w = WORKLOADS[key]
    # NERF_AMD_DIST_BACKEND=gloo rehearses the N>1 path with several ranks on one GPU (no RCCL)
    backend = os.environ.get("NERF_AMD_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit("bench.py rank %d/%d: %d ranks over RCCL need %d GPUs, %d visible (NERF_AMD_DIST_BACKEND=gloo "
                         "rehearses with the ranks sharing a GPU)" % (rank, world, world, world, n_dev))
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    if world > 1 or "RANK" in os.environ:        # under torch.distributed.run also with one rank (RCCL smoke of the C5 path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    # what the process group actually is: the world size the backend initialised and the device of every rank
    dist_info = {"backend": None, "rccl_world": None, "rank_devices": [str(dev)]}
    if dist.is_initialized():
        devs = [None] * world
        dist.all_gather_object(devs, "%s %s" % (torch.cuda.get_device_properties(dev).name, dev))
        dist_info.update(backend=dist.get_backend(), rank_devices=devs,
                         rccl_world=dist.get_world_size() if dist.get_backend() == "nccl" else None)

    models = []
    for seed in (0, 10):
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**ARCH, "skips": (4,)}))
        m.precision = args.precision
        models.append(m.to(dev).requires_grad_(False))
    fine_model = models[1]
    torch.manual_seed(1234 + rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def collect_profile():
        launches, ms, pts = (ctypes.c_int64 * 3)(), (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
        _lib.lib.nerf_amd_profile_collect(launches, ms, pts)      # classes: 0 exact fp32, 1 bf16, 2 split precision
        return [(int(launches[c]), float(ms[c]), float(pts[c])) for c in range(3)]

    def run(wk, perturb, steps, warmup, precision=None, mode="pixel_ranges", profile=False):
        """Time `steps` steps of workload wk.  Returns (seconds [max over ranks], rays per step, profile)."""
        for m in models:
            m.precision = precision or args.precision
        renderer = render_utils.Renderer(**renderer_cfg(wk, perturb))
        K, poses = camera(wk, synth)
        H, W, chunk = wk["H"], wk["W"], wk["chunk"]
        poses_t = [torch.from_numpy(p) for p in poses]
        fm = fine_model if wk["Ni"] > 0 else None

        def frames(first, count):
            if world == 1 and not (wk.get("poses") and dist.is_initialized()):
                for k in range(first, first + count):
                    renderer.render(H, W, K, models[0], fm, chunk=chunk, c2w=poses_t[k % len(poses_t)], retraw=False)
            elif mode == "pixel_ranges" or world == 1:
                nd.render_poses_gathered(renderer, H, W, K, chunk, [poses_t[k % len(poses_t)] for k in range(first, first + count)],
                                         models[0], fm, on_frame=lambda i, rgb, disp, acc: None)
            else:       # whole frames round-robin: rank r renders frames r, r + world, ...
                for k in range(first + rank, first + count, world):
                    renderer.render(H, W, K, models[0], fm, chunk=chunk, c2w=poses_t[k % len(poses_t)], retraw=False)

        with torch.no_grad():
            if world > 1:      # communicator set-up must not land in the timed region, whatever --warmup is
                nd.gather_rows(torch.zeros(nd.shard_sizes(H * W, world)[rank], 5, device=dev), H * W, 0)
            frames(0, warmup)
            fence()
            if profile:
                collect_profile()
                _lib.lib.nerf_amd_profile_enable(1)
                nd.gather_stats(reset=True)
                nd.enable_diagnostics(dist.is_initialized() and mode == "pixel_ranges")
            t0 = time.perf_counter()
            frames(warmup, steps)
            torch.cuda.synchronize()
            own_dt = time.perf_counter() - t0          # this rank's own work is done (before the barrier makes them equal)
            fence()
            dt = time.perf_counter() - t0
            prof = None
            if profile:
                _lib.lib.nerf_amd_profile_enable(0)
                prof = collect_profile()
                if dist.is_initialized() and mode == "pixel_ranges":
                    # every rank's own account of the timed region (render, the part of the gathers its renders did not
                    # hide, what rank 0 received): one run explains its own efficiency
                    mine = dict(nd.diagnostics(), rank=rank, own_ms_per_frame=own_dt / max(1, steps) * 1e3)
                    per_rank = [None] * world
                    dist.all_gather_object(per_rank, mine)
                    run.per_rank = per_rank
                nd.enable_diagnostics(False)
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), H * W, prof

    dt, rays_per_step, prof = run(w, args.perturb, args.steps, args.warmup, profile=True)
    dist_info.update(nd.gather_stats())          # per-frame gathers issued inside the timed region, and by which path
    if getattr(run, "per_rank", None):
        pr = run.per_rank
        dist_info["per_rank"] = pr
        dist_info["slowest_rank"] = max(pr, key=lambda d: d.get("own_ms_per_frame", 0.0))["rank"]
    cls = {"fp32": 0, "bf16": 1, "fp32_split": 2}[args.precision]
    launches, kern_ms, kern_pts = prof[cls]

    sub = {}
    if not args.no_subrecords:
        if world == 1 and key == "c2":
            d0, n0, _ = run(w, 0.0, max(2, args.steps // 4), 1)
            sub["perturb0"] = {"value": n0 * max(2, args.steps // 4) / d0, "unit": "rays/s",
                               "ms_per_step": d0 / max(2, args.steps // 4) * 1e3, "steps": max(2, args.steps // 4)}
            if args.precision == "bf16":
                # the two modes at the reference's precision: the exact fp32 MFMA kernel and the split-precision kernel
                # (fp16 operand pairs, 3 MFMAs per product: its roofline is a third of the dense 16-bit peak)
                for name, kern, ci, peak, nsteps in (("fp32", "mlp_f32_kernel", 0, PEAK_FP32_TFLOPS, 2),
                                                      ("fp32_split", "mlp_split_kernel", 2, PEAK_BF16_TFLOPS / 3.0, 4)):
                    dd, nn, pp = run(w, args.perturb, nsteps, 1, precision=name, profile=True)
                    ll, mss, ptss = pp[ci]
                    ach = ptss * FLOP_PER_POINT / (mss / 1e3) / 1e12 if mss else 0.0
                    pm = measured_pmc(kern)
                    sub[name] = {"value": nn * nsteps / dd, "unit": "rays/s", "ms_per_step": dd / nsteps * 1e3, "steps": nsteps,
                                 "dtype": "f32" if name == "fp32" else "f16x2 (fp32-class: hi/lo fp16 operand pairs, fp32 accumulate)",
                                 "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                              "peak_note": "fp32 MFMA peak" if name == "fp32" else
                                                           "dense fp16 MFMA peak / 3 (three MFMAs per algorithmic product)",
                                              "mfma_util": pm["mfma_util"] if pm else None,
                                              "traffic": pm["traffic"] if pm else None,
                                              "kernel": kern, "launches": ll, "avg_launch_ms": mss / ll if ll else None}}
            n5 = max(3, args.steps // 4)
            d5, r5, _ = run(WORKLOADS["c5"], args.perturb, n5, 1)
            sub["c5_n1"] = {"value": r5 * n5 / d5, "unit": "rays/s", "ms_per_step": d5 / n5 * 1e3, "steps": n5,
                            "workload": WORKLOADS["c5"]["name"],
                            "note": "the N>1 workload on one GPU: the N=1 point of the strong-scaling curve"}
            n1 = max(3, args.steps // 4)
            d1, r1, _ = run(WORKLOADS["c1"], args.perturb, n1, 1)
            sub["c1"] = {"value": r1 * n1 / d1, "unit": "rays/s", "ms_per_step": d1 / n1 * 1e3, "steps": n1,
                         "workload": WORKLOADS["c1"]["name"],
                         "note": "BASELINE configs[0] (the reference's CPU-runnable case) on the GPU; its CPU figure is cpu_baseline.c1"}
        if world > 1:
            k_rr = -(-args.steps // world) * world          # whole rounds, so that every rank renders the same number of frames
            drr, rrr, _ = run(w, args.perturb, k_rr, min(args.warmup, 1), mode="frames_round_robin")
            sub["frames_round_robin"] = {"value": rrr * k_rr / drr, "unit": "rays/s", "ms_per_step": drr / k_rr * 1e3,
                                         "steps": k_rr,
                                         "note": "whole frames dealt round-robin (steps rounded up to whole rounds), no data-path collective, "
                                                 "frames stay on their rank"}

    if rank == 0:
        value = rays_per_step * args.steps / dt
        kern_s = kern_ms / 1e3
        achieved = (kern_pts * FLOP_PER_POINT / kern_s / 1e12) if kern_s > 0 else 0.0
        peak = {"bf16": PEAK_BF16_TFLOPS, "fp32": PEAK_FP32_TFLOPS, "fp32_split": PEAK_BF16_TFLOPS / 3.0}[args.precision]
        kernel = {"bf16": "mlp_bf16_s16p_kernel", "fp32": "mlp_f32_kernel", "fp32_split": "mlp_split_kernel"}[args.precision]
        pmc = measured_pmc(kernel) if key == "c2" else None
        # launches are per 32768-ray group (nerf_amd_render_batch), whatever the API chunk: coarse and fine launch alternate
        chunk_rays = min(32768, rays_per_step)
        launches_per_group = 2 if w["Ni"] else 1
        avg_launch_s = (kern_s / launches) if launches else None
        # 44 B in + 44 B out per ray, shared by the field launches of a ray group
        algorithmic_bytes = chunk_rays * 88 / launches_per_group
        # what this design moves per launch on top of that: raw [P,4] out (16 B/point) + z in (4 B/point) + the rays,
        # because the field kernel and the per-ray kernels are separate launches
        design_bytes = chunk_rays * 44 + (chunk_rays * (2 * w["Nc"] + w["Ni"]) // launches_per_group) * 20
        out = {
            "metric": "rays_per_sec", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            # N > 1 is BASELINE configs[4] (a fixed 200-pose 800x800 test set sharded over the ranks): strong scaling.
            # N = 1 is configs[1], the configuration the metric is quoted on; the N = 1 point of the configs[4] curve is
            # the `c5_n1` sub-record of this line (same rays/s per GPU to within the run-to-run spread).
            "scaling": "strong", "scaling_curve_workload": WORKLOADS["c5"]["name"],
            "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32", "fp32_split": "f16x2"}[args.precision], "data": "synthetic",
            "config": {"workload": w["name"], "entry": "Renderer.render(c2w=pose, retraw=False)",
                       "rays_per_step": rays_per_step, "chunk": w["chunk"], "perturb": args.perturb,
                       "N_samples": w["Nc"], "N_importance": w["Ni"], "weights": "random-init seeds 0/10",
                       "parallelism": ("flat pixel-range shards x%d of every frame, gather of [rays,5] rows to rank 0 "
                                       "overlapped with the next frame (strong scaling over a fixed %d-pose list)" % (world, N_POSES_C5))
                       if world > 1 else "single GPU"},
            "dist": dist_info,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": pmc["traffic"] if pmc else None,
                         "traffic_unit": "bytes per launch (HBM, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": ("profiles/" + pmc["file"]) if pmc else None,
                         "mfma_util": pmc["mfma_util"] if pmc else None,
                         "hbm_gbps": (pmc["traffic"] / avg_launch_s / 1e9) if (pmc and avg_launch_s) else None,
                         "csrc_hash": csrc_hash(),
                         "algorithmic_bytes_per_launch": algorithmic_bytes,
                         "design_bytes_per_launch": design_bytes,
                         "traffic_over_algorithmic": (pmc["traffic"] / algorithmic_bytes) if pmc else None,
                         "kernel": kernel, "launches": launches,
                         "avg_launch_ms": (kern_ms / launches) if launches else None,
                         "flop_per_point": FLOP_PER_POINT, "points": kern_pts, "rank": 0},
        }
        out.update(sub)
        if world == 1 and not args.no_cpu_baseline and key in ("c1", "c2"):
            cpu = cpu_baseline(torch, synth, w, full_c1=not args.no_subrecords)
            chk = cpu.pop("_check", None)
            if chk is not None:
                # BASELINE.json's metric also says "PSNR vs ref": the 4096-ray batch the CPU leg just rendered (perturb 0, the
                # bench's own weights), rendered by every GPU mode and compared with the CPU image -- the oracle as checker
                import math
                batch, ref, ref_models, ocfg = chk
                import precision_census as PC
                rr = render_utils.Renderer(**renderer_cfg(w, 0.0))
                cm = []
                for seed in CHECK_SEEDS:
                    m = nerf.NeRF(**ARCH)
                    m.load_state_dict(synth.torch_state_dict(seed, CHECK_SHARPEN, **{**ARCH, "skips": (4,)}))
                    cm.append(m.to(dev).requires_grad_(False))
                check = {"rays": int(batch.shape[0]),
                         "against": "the CPU oracle's fp32 image of the same rays (perturb 0)",
                         "weights": "x%g-sharpened random weights, seeds %d / %d: a field with content (the timed workload's "
                                    "default-scale weights render an empty view)" % (CHECK_SHARPEN, CHECK_SEEDS[0], CHECK_SEEDS[1]),
                         "ref_rgb_variance": float(ref["rgb_map"].var()), "ref_acc_mean": float(ref["acc_map"].mean()),
                         "modes": {}}
                for prec in ("bf16", "fp32_split", "fp32"):
                    for m in cm:
                        m.precision = prec
                    with torch.no_grad():
                        o = {k: v.cpu() for k, v in rr.render_rays(batch.to(dev), cm[0], cm[1], retraw=True, retweights=True).items()}
                    d = o["rgb_map"].double() - ref["rgb_map"].double()
                    mse = float((d ** 2).mean())
                    check["modes"][prec] = {"psnr_db": (-10.0 * math.log10(mse)) if mse > 0 else None,
                                            "max_abs_rgb": float(d.abs().max()),
                                            "max_abs_acc": float((o["acc_map"].double() - ref["acc_map"].double()).abs().max())}
                    if prec != "bf16":
                        # the fp32-class modes, ray by ray (tools/precision_census.py attribute): what moved against the CPU
                        # image is what the reference's own conditioning moves -- `unexplained` counts rays beyond 2e-4
                        # with neither a displaced fine sample nor a flip-prone last sample; `staged_*` are the maxima over
                        # ALL rays of the differences from the oracle evaluated on the mode's own depths / raw
                        att = PC.attribute(ocfg, batch, ref_models[0], ref_models[1], o, ref=ref)
                        check["modes"][prec].update({k: att[k] for k in ("frac_gt_tol", "frac_gt_1e-3", "frac_displaced", "unexplained",
                                                                         "max_abs_undisplaced", "staged_raw_max", "staged_rgb_max", "staged_acc_max")})
                cpu["check"] = check
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
