#!/usr/bin/env python3
"""bench.py -- rays/sec of the render hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of Renderer.render_batch over a resident [rays, 11] batch:
Lego half-res geometry (400x400 = 160 000 rays per GPU), 64 coarse + 128 fine
samples, 8x256 viewdirs MLP x 2 (coarse + fine, random-init seeds 0/10),
4096-ray chunks, perturb=0, white background (BASELINE.json configs[1]).
Rays are generated into HBM before the timed region; outputs stay in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling.
The view is 400 x (400 N) pixels; rank r renders flat pixel range r of it and the
finished [rays, 5] rows are gathered to rank 0 inside the timed step (the one real
exchange of the path).

The JSON line also carries
  roofline     -- the fused bf16 field kernel (fine-pass launches dominate):
                  algorithmic FLOP (1 186 816 per point) / device time
                  from hipEvents recorded around every launch on its stream
                  (nerf_amd_profile_*), against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference, "port")
                  timed on this box's host cores on a bounded sample of the same
                  workload (one 4096-ray batch, 64+128), rank 0, N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
os.environ.setdefault("NERF_AMD_QUIET", "1")

FLOP_PER_POINT = 1186816            # 2 x 593 408 MAC, viewdirs 8x256 MLP
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA peak, MI355X_MICROARCH.md
H, W_PER_GPU, CHUNK = 400, 400, 4096
N_SAMPLES, N_IMPORTANCE = 64, 128
ARCH = dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4)
RCFG = dict(perturb=0.0, N_importance=N_IMPORTANCE, N_samples=N_SAMPLES, use_viewdirs=True, white_bkgd=True,
            raw_noise_std=0.0, ndc=False, lindisp=False, near=2.0, far=6.0)
NDC_FOCAL = None

# BASELINE.json configs; c2 is the one the metric is quoted on (the default, and what the driver runs).
WORKLOADS = {
    "c1": dict(name="lego_halfres_400x400_64c_coarse_only_chunk32768", H=400, W=400, chunk=32768, Nc=64, Ni=0),
    "c2": dict(name="lego_halfres_400x400_64c+128f_viewdirs_8x256_chunk4096", H=400, W=400, chunk=4096, Nc=64, Ni=128),
    "c3": dict(name="lego_fullres_800x800_64c+128f_viewdirs_8x256_chunk32768", H=800, W=800, chunk=32768, Nc=64, Ni=128),
    "c4": dict(name="fern_llff_378x504_ndc_64c+64f_viewdirs_8x256_chunk32768", H=378, W=504, chunk=32768, Nc=64, Ni=64,
               ndc=True, near=0.0, far=1.0, white_bkgd=False, focal=408.0),
}


def select_workload(key):
    global H, W_PER_GPU, CHUNK, N_SAMPLES, N_IMPORTANCE, RCFG, NDC_FOCAL
    w = WORKLOADS[key]
    H, W_PER_GPU, CHUNK, N_SAMPLES, N_IMPORTANCE = w["H"], w["W"], w["chunk"], w["Nc"], w["Ni"]
    RCFG = dict(RCFG, N_samples=w["Nc"], N_importance=w["Ni"], ndc=w.get("ndc", False), near=w.get("near", 2.0),
                far=w.get("far", 6.0), white_bkgd=w.get("white_bkgd", True))
    NDC_FOCAL = w.get("focal")
    return w["name"]


def cpu_baseline(torch, synth):
    """Oracle on host cores: one 4096-ray 64+128 batch, 1 warm-up + best of 2."""
    from oracle import nerf_oracle as O
    import numpy as np
    # the GPU box reports every host core but grants a 16-core share per GPU
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    K = synth.lego_intrinsics(H, W_PER_GPU)
    ro, rd = synth.rays_np(H, W_PER_GPU, K, synth.LEGO_C2W, np.arange(80000, 80000 + CHUNK))
    batch = torch.from_numpy(synth.ray_batch_np(ro, rd, 2.0, 6.0, True))
    models = []
    for seed in (0, 10):
        sd = synth.make_state_dict(seed, 1.0, **{**ARCH, "skips": (4,)})
        models.append((O.state_dict_to_torch(sd), O.Arch(**ARCH)))
    cfg = O.RenderCfg(**RCFG)
    best = float("inf")
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            O.render_rays(cfg, batch, models[0], models[1])
            dt = time.perf_counter() - t0
            if i > 0:
                best = min(best, dt)
    return {"value": CHUNK / best, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": "one %d-ray batch, 64+128 samples, torch-CPU oracle fp32, best of 2 after warm-up (%.2f s)" % (CHUNK, best)}


def measured_traffic(kernel_tag, chunk_points):
    """HBM bytes per launch of the field kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_summary_*.json, made by tools/pmc_summary.py: FETCH_SIZE x2 + WRITE_SIZE per
    MI355X_MICROARCH.md), averaged over the coarse and fine launch shapes like `achieved`.
    PMC counters cannot be read from inside the timed run; None if no summary is present."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary_*.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for name, grids in d.items():
            if kernel_tag not in name:
                continue
            vals = []
            for pts in chunk_points:
                g = grids.get(str((pts // 256) * 512))
                if g and "hbm_bytes_per_launch" in g:
                    vals.append(g["hbm_bytes_per_launch"]["total"])
            if len(vals) == len(chunk_points):
                best = (sum(vals) / len(vals), os.path.basename(path))
                continue
            # one workgroup per CU walks the tiles: coarse and fine launches share the grid (256 CUs x 512
            # threads) and the summary's mean per launch already averages the two shapes
            g = grids.get(str(256 * 512))
            if g and "hbm_bytes_per_launch" in g and all(pts >= 256 * 256 for pts in chunk_points):
                best = (g["hbm_bytes_per_launch"]["total"], os.path.basename(path))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    args = ap.parse_args()
    workload_name = select_workload(args.workload)
    if os.environ.get("NERF_AMD_OVERLAP_CHUNKS") == "1":      # A/B knob: two-stream chunk pipeline of render_batch
        from nerf_shared_amd import render_utils as _ru
        _ru.Renderer.overlap_chunks = True
    if os.environ.get("NERF_AMD_FUSE_CHUNKS") == "0":         # A/B knob: one render_rays call per chunk
        from nerf_shared_amd import render_utils as _ru
        _ru.Renderer.fuse_chunk_launches = False

    import torch
    import torch.distributed as dist
    from nerf_shared_amd import _lib, dist as nd, nerf, render_utils, synth, utils

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device; there is no CPU path")
    # NERF_AMD_DIST_BACKEND=gloo rehearses the N>1 path with several ranks on one GPU (no RCCL)
    backend = os.environ.get("NERF_AMD_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    models = []
    for seed in (0, 10):
        m = nerf.NeRF(**ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**ARCH, "skips": (4,)}))
        m.precision = args.precision
        models.append(m.to(dev))
    renderer = render_utils.Renderer(**RCFG)

    Wimg = W_PER_GPU * world
    K = synth.lego_intrinsics(H, W_PER_GPU)
    pose = synth.LEGO_C2W
    if NDC_FOCAL:      # forward-facing LLFF-style camera
        K[0][0] = K[1][1] = NDC_FOCAL
        K[1][2] = 0.5 * H
        import numpy as np
        pose = np.array([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 0.1]], np.float32)
    K[0][2] = 0.5 * Wimg
    n_total = H * Wimg
    lo, hi = nd.shard_range(n_total, rank, world)
    rays = utils.make_ray_batch(H, Wimg, K, pose, RCFG["near"], RCFG["far"], True, RCFG["ndc"], device=dev,
                                pix0=lo, n=hi - lo)
    torch.cuda.synchronize()

    def step():
        ret = renderer.render_batch(models[0], models[1], rays, CHUNK, False)
        if world > 1:
            return nd.gather_rows(nd.pack_maps(ret), n_total, 0)
        return ret

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        if world > 1:      # communicator set-up must not land in the timed region, whatever --warmup is
            nd.gather_rows(torch.zeros(hi - lo, 5, device=dev), n_total, 0)
        for _ in range(args.warmup):
            step()
        fence()
        _lib.lib.nerf_amd_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        _lib.lib.nerf_amd_profile_enable(0)

    launches = (ctypes.c_int64 * 2)()
    ms = (ctypes.c_double * 2)()
    pts = (ctypes.c_double * 2)()
    _lib.lib.nerf_amd_profile_collect(launches, ms, pts)
    cls = 1 if args.precision == "bf16" else 0

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        rays_per_step = n_total
        value = rays_per_step * args.steps / dt
        kern_s = ms[cls] / 1e3
        achieved = (pts[cls] * FLOP_PER_POINT / kern_s / 1e12) if kern_s > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3
        traffic = (measured_traffic("mlp_bf16_s16_kernel", [CHUNK * N_SAMPLES, CHUNK * (N_SAMPLES + N_IMPORTANCE)])
                   if cls == 1 and args.workload == "c2" else None)
        out = {
            "metric": "rays_per_sec", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": workload_name,
                       "rays_per_step": rays_per_step, "rays_per_gpu_per_step": H * W_PER_GPU, "chunk": CHUNK,
                       "N_samples": N_SAMPLES, "N_importance": N_IMPORTANCE, "weights": "random-init seeds 0/10",
                       "parallelism": "ray-range shards x%d, one gather of [rays,5] to rank 0 per step" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic[0] if traffic else None,
                         "traffic_unit": "bytes per launch (HBM, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": ("profiles/" + traffic[1]) if traffic else None,
                         "traffic_note": "measured traffic = algorithmic bytes + one L2 fill of the 1.2 MB bf16 weight stream per XCD "
                                         "(8 x 1.2 MB; FETCH_SIZE counts L2 misses, including those the MALL serves); "
                                         "the kernel is MFMA-bound, 21.6 MB per 0.43 ms launch is 50 GB/s",
                         "algorithmic_bytes_per_launch": CHUNK * 44 + (CHUNK * (2 * N_SAMPLES + N_IMPORTANCE) // 2) * 20,
                         "kernel": "mlp_bf16_s16_kernel" if cls == 1 else "mlp_f32_kernel",
                         "launches": int(launches[cls]),
                         "avg_launch_ms": (ms[cls] / launches[cls]) if launches[cls] else None,
                         "flop_per_point": FLOP_PER_POINT, "points": pts[cls], "rank": 0},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            out["cpu_baseline"] = cpu_baseline(torch, synth)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
