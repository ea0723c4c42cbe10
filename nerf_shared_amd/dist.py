"""Ray-range sharding of a render across the GPUs of one node.

Rays are independent (no cross-ray term anywhere in render_rays), so the path
shards with no data-path collective: rank r renders the contiguous flat pixel
range [N*r/G, N*(r+1)/G) of every image, generating its own rays from (K, c2w).
The only exchange is one gather of the finished [N/G, 5] (rgb, disp, acc) rows to
rank 0 per image (RCCL over xGMI when the backend is "nccl"; gloo on CPU tests).
The reference has no distributed code; this is the build's
own multi-GPU row (section 8e).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) share of n items for `rank` of `world` (sizes differ by at most 1)."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_rows(local, n_total, dst=0, group=None):
    """Gather row-shards (split by shard_range) of a [n_local, C] tensor to `dst`.
    Returns the [n_total, C] tensor on dst, None elsewhere.  One collective."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    sizes = shard_sizes(n_total, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    C = local.shape[1]
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (several ranks sharing one GPU, no RCCL): stage through the host
        out = gather_rows(local.cpu(), n_total, dst, group)
        return out.to(local.device) if out is not None else None
    if max(sizes) == min(sizes):
        out = torch.empty(n_total, C, device=local.device, dtype=local.dtype) if rank == dst else None
        dist.gather(local.contiguous(), list(out.split(sizes[0])) if rank == dst else None, dst=dst, group=group)
        return out
    # ragged shards: pad to the largest, gather, trim
    m = max(sizes)
    pad = torch.zeros(m, C, device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def pack_maps(ret):
    """[n, 5] = rgb(3) | disp | acc rows of a render_batch result: the gather payload."""
    return torch.cat([ret['rgb_map'], ret['disp_map'][:, None], ret['acc_map'][:, None]], -1)


def render_image_sharded(renderer, H, W, K, c2w, coarse_model, fine_model, chunk=1024 * 32,
                         gather=True, group=None):
    """Render this rank's pixel range of an H x W view; with gather=True rank 0
    receives the whole (rgb [H,W,3], disp [H,W], acc [H,W]) and other ranks None."""
    from . import utils
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(H * W, rank, world)
    dev = next(coarse_model.parameters()).device
    batch = utils.make_ray_batch(H, W, K, c2w, renderer.near, renderer.far, renderer.use_viewdirs,
                                 renderer.ndc, device=dev, pix0=lo, n=hi - lo)
    local = pack_maps(renderer.render_batch(coarse_model, fine_model, batch, chunk, False))
    if world == 1:
        full = local
    elif gather:
        full = gather_rows(local, H * W, 0, group)
    else:
        return local
    if full is None:
        return None
    return full[:, 0:3].reshape(H, W, 3), full[:, 3].reshape(H, W), full[:, 4].reshape(H, W)
