"""What one rank of an N-GPU C5 run does, timed on ONE GPU: the per-rank share of the strong-scaling curve.

bench.py --gpus N shards every 800x800 frame by flat pixel range (dist.render_poses_gathered).  No multi-GPU box is
available to this repository's sessions, so this tool renders the share of rank r of G -- the same calls
(utils.make_ray_batch(pix0, n) -> Renderer.render_batch -> dist.pack_maps), no collective -- for G = 1, 2, 4, 8 and
reports the frame time of the slowest sampled rank, the aggregate rate G such ranks would reach if the gather is
hidden (it is issued on a side stream under the next frame), and the efficiency against G = 1.  It measures the
fixed per-frame cost of a rank (launch-group tail, small kernels, Python), which is what bounds the curve; it does
not measure RCCL.

  python tools/shard_share_bench.py [--frames 12] [--out gpurun_out/shard_share.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    import bench
    from nerf_shared_amd import dist as nd, nerf, render_utils, synth, utils

    dev = torch.device("cuda:0")
    wk = bench.WORKLOADS["c5"]
    H, W, chunk = wk["H"], wk["W"], wk["chunk"]
    models = []
    for seed in (0, 10):
        m = nerf.NeRF(**bench.ARCH)
        m.load_state_dict(synth.torch_state_dict(seed, 1.0, **{**bench.ARCH, "skips": (4,)}))
        models.append(m.to(dev).requires_grad_(False))
    torch.manual_seed(1234)
    renderer = render_utils.Renderer(**bench.renderer_cfg(wk, 1.0))
    K, poses = bench.camera(wk, synth)
    poses_t = [torch.from_numpy(p) for p in poses]

    def share(rank, world, first, count):
        lo, hi = nd.shard_range(H * W, rank, world)
        for k in range(first, first + count):
            batch = utils.make_ray_batch(H, W, K, poses_t[k % len(poses_t)], renderer.near, renderer.far,
                                         renderer.use_viewdirs, renderer.ndc, device=dev, pix0=lo, n=hi - lo)
            ret = renderer.render_batch(models[0], models[1], batch, chunk, False)
            nd.pack_maps(ret)
        return hi - lo

    rows = []
    with torch.no_grad():
        share(0, 1, 0, 2)
        torch.cuda.synchronize()
        for world in (1, 2, 4, 8):
            worst = None
            for rank in sorted({0, world // 2, world - 1}):
                share(rank, world, 0, 2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = share(rank, world, 2, args.frames)
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                rec = {"rank": rank, "rays": n, "ms_per_frame": dt / args.frames * 1e3,
                       "host_enqueue_ms_per_frame": t_host / args.frames * 1e3}
                if worst is None or rec["ms_per_frame"] > worst["ms_per_frame"]:
                    worst = rec
            rows.append({"world": world, "slowest_sampled_rank": worst,
                         "aggregate_rays_per_s_if_gather_hidden": H * W / (worst["ms_per_frame"] / 1e3)})
    base = rows[0]["aggregate_rays_per_s_if_gather_hidden"]
    for r in rows:
        r["efficiency_vs_world_1"] = r["aggregate_rays_per_s_if_gather_hidden"] / (base * r["world"])
        print("world %d: %.2f ms per frame share (%d rays, host %.2f ms) -> %.2f M rays/s aggregate, efficiency %.3f"
              % (r["world"], r["slowest_sampled_rank"]["ms_per_frame"], r["slowest_sampled_rank"]["rays"],
                 r["slowest_sampled_rank"]["host_enqueue_ms_per_frame"],
                 r["aggregate_rays_per_s_if_gather_hidden"] / 1e6, r["efficiency_vs_world_1"]))
    out = {"workload": wk["name"], "frames": args.frames, "device": torch.cuda.get_device_name(0), "rows": rows,
           "note": "one GPU rendering the pixel range of rank r of G; no collective (tools/shard_share_bench.py)"}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
