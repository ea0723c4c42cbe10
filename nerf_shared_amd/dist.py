"""Ray-range sharding of a render across the GPUs of one node.

Rays are independent (no cross-ray term anywhere in render_rays), so the path
shards with no data-path collective: rank r renders the contiguous flat pixel
range [N*r/G, N*(r+1)/G) of every image, generating its own rays from (K, c2w).
The only exchange is one gather of the finished [N/G, 5] (rgb, disp, acc) rows to
rank 0 per image (RCCL over xGMI when the backend is "nccl"; gloo on CPU tests).
The reference has no distributed code; this is the build's
own multi-GPU row (section 8e).
"""
import os

import torch
import torch.distributed as dist


# what OverlappedGather did in this process: how many per-frame gathers it issued, and by which path
# ("collective on a side stream" = RCCL, "async collective" = gloo on CPU tensors, "host-staged" = the gloo rehearsal
# with GPU tensors, "local" = a world of one without a forced collective).  bench.py reports it.
_STATS = {"gathers": 0, "gather_path": None}


def gather_stats(reset=False):
    out = dict(_STATS)
    if reset:
        _STATS.update(gathers=0, gather_path=None)
    return out


# Per-rank diagnostics of render_poses_gathered (bench.py --gpus N prints them for every rank, so that ONE multi-GPU run
# explains its own efficiency): timing events around every frame's render on the caller's stream and around every wait of
# the caller's stream for a gather (the part of the gather the next frames' renders did not hide).
_DIAG = {"on": False, "frames": [], "waits": [], "recv_bytes": 0}


def enable_diagnostics(on=True):
    _DIAG.update(on=bool(on), frames=[], waits=[], recv_bytes=0)


def diagnostics():
    """After the device is idle (the caller synchronised): {frames, render_ms_per_frame, exposed_gather_ms_per_frame,
    recv_bytes_per_frame} of this rank since enable_diagnostics().  The waits sit inside later frames' render intervals;
    render_ms_per_frame has them taken out."""
    n = len(_DIAG["frames"])
    if n == 0:
        return {"frames": 0}
    wait = sum(a.elapsed_time(b) for a, b in _DIAG["waits"])
    total = sum(a.elapsed_time(b) for a, b in _DIAG["frames"])
    return {"frames": n, "render_ms_per_frame": (total - wait) / n, "exposed_gather_ms_per_frame": wait / n,
            "recv_bytes_per_frame": _DIAG["recv_bytes"] // n}


def shard_range(n, rank, world):
    """Contiguous [lo, hi) share of n items for `rank` of `world` (sizes differ by at most 1)."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def _start_gather(local, n_total, dst, group, async_op):
    """Issue the one collective of gather_rows.  Returns (work or None, finish) where finish() -> the
    [n_total, C] tensor on dst (None elsewhere) once the collective is complete."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:                                   # a one-rank group (forced collective): the gather list is one tensor
        out = torch.empty_like(local)
        work = dist.gather(local.contiguous(), [out], dst=dst, group=group, async_op=async_op)
        return work, (lambda: out)
    sizes = shard_sizes(n_total, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    C = local.shape[1]
    if max(sizes) == min(sizes):
        out = torch.empty(n_total, C, device=local.device, dtype=local.dtype) if rank == dst else None
        work = dist.gather(local.contiguous(), list(out.split(sizes[0])) if rank == dst else None, dst=dst, group=group,
                           async_op=async_op)
        return work, (lambda: out)
    # ragged shards: pad to the largest, gather, trim
    m = max(sizes)
    pad = torch.zeros(m, C, device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    work = dist.gather(pad, bufs, dst=dst, group=group, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        if pad.is_cuda:             # allocated under the gather's stream, read by the cat on the caller's current stream
            cur = torch.cuda.current_stream(pad.device)
            for b in bufs:
                b.record_stream(cur)
        return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)
    return work, finish


def gather_rows(local, n_total, dst=0, group=None):
    """Gather row-shards (split by shard_range) of a [n_local, C] tensor to `dst`.
    Returns the [n_total, C] tensor on dst, None elsewhere.  One collective."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (several ranks sharing one GPU, no RCCL): stage through the host
        out = gather_rows(local.cpu(), n_total, dst, group)
        return out.to(local.device) if out is not None else None
    _, finish = _start_gather(local, n_total, dst, group, False)
    return finish()


class OverlappedGather:
    """The per-frame gather of SURVEY.md section 8e, taken off the render's critical path: ``submit(rows)``
    issues the gather of frame k's finished rows on a side stream (RCCL) / as an asynchronous operation
    (gloo on CPU tensors) and returns at once, so the caller enqueues the render of frame k+1 while the rows
    of frame k cross xGMI; ``collect()`` hands the gathered frames back in submission order (None on ranks
    other than `dst`).  At most `depth` gathers are kept in flight (the oldest is completed first).

    Stream contract on the GPU: the side stream waits for an event recorded on the caller's stream at
    submit (the rows are complete), the rows are kept alive and marked as used by the side stream, and
    collect() makes the caller's current stream wait for the gathers it returns."""

    def __init__(self, n_total, dst=0, group=None, depth=2):
        self.n_total, self.dst, self.group, self.depth = n_total, dst, group, max(1, int(depth))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # NERF_AMD_FORCE_COLLECTIVE=1: issue the collective even in a one-rank group (a 1-GPU box can then exercise the
        # RCCL side-stream path: async gather under a stream context, stream-level wait, events)
        self.single = self.world == 1 and not (dist.is_initialized() and os.environ.get("NERF_AMD_FORCE_COLLECTIVE") == "1")
        self._pending = []        # (finish, work, done_event, keepalive) in submission order
        self._ready = []
        self._side = None
        self.submitted = 0

    def _complete_oldest(self):
        finish, work, done, _keep = self._pending.pop(0)
        if done is not None:                       # GPU: the caller's stream waits for the side stream's gather
            if _DIAG["on"]:
                pre, post = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                pre.record()
                torch.cuda.current_stream().wait_event(done)
                post.record()
                _DIAG["waits"].append((pre, post))
            else:
                torch.cuda.current_stream().wait_event(done)
        elif work is not None:
            work.wait()
        res = finish()
        if res is not None and res.is_cuda:        # allocated under the side stream, consumed on the caller's
            res.record_stream(torch.cuda.current_stream(res.device))
        self._ready.append(res)

    def ready_count(self):
        return len(self._ready)

    def poll_ready(self):
        """Frames whose gather has already been completed by an earlier submit (oldest first), without waiting for
        the ones still in flight."""
        out, self._ready = self._ready, []
        return out

    def _count(self, path):
        _STATS["gathers"] += 1
        _STATS["gather_path"] = path

    def submit(self, rows):
        self.submitted += 1
        if self.single:
            self._count("local")
            self._ready.append(rows)
            return
        while len(self._pending) >= self.depth:
            self._complete_oldest()
        if rows.is_cuda and dist.get_backend(self.group) == "gloo":     # rehearsal only: synchronous, through the host
            self._count("host-staged")
            self._ready.append(gather_rows(rows, self.n_total, self.dst, self.group))
            return
        if rows.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(rows.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(rows.device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                work, finish = _start_gather(rows, self.n_total, self.dst, self.group, True)
                work.wait()                        # stream-level: the side stream waits for the collective, the host does not
                done = torch.cuda.Event()
                done.record(self._side)
            rows.record_stream(self._side)
            self._count("collective on a side stream")
            self._pending.append((finish, work, done, rows))
        else:
            self._count("async collective")
            work, finish = _start_gather(rows, self.n_total, self.dst, self.group, True)
            self._pending.append((finish, work, None, rows))

    def collect(self):
        """All frames submitted so far, oldest first (completes what is still in flight)."""
        while self._pending:
            self._complete_oldest()
        out, self._ready = self._ready, []
        return out


def broadcast_parameters(models, src=0, group=None):
    """Replicate the parameters of `models` (NeRF modules; None entries are skipped) from rank `src` to every rank:
    one broadcast of one flat buffer per model (2.4 MB for the 8x256 field), then the packed device copies are marked
    stale so the next render re-packs them.  The ranks of a sharded render must hold identical weights (SURVEY.md
    section 8e: weights replicated once at load); loading the same checkpoint on every rank makes this unnecessary."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in models:
        if m is None:
            continue
        params = [p for p in m.parameters()]
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.cpu()
            dist.broadcast(host, src=src, group=group)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src, group=group)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n
        if hasattr(m, "weights_changed"):
            m.weights_changed()


def pack_maps(ret):
    """[n, 5] = rgb(3) | disp | acc rows of a render_batch result: the gather payload."""
    return torch.cat([ret['rgb_map'], ret['disp_map'][:, None], ret['acc_map'][:, None]], -1)


def render_image_sharded(renderer, H, W, K, c2w, coarse_model, fine_model, chunk=1024 * 32,
                         gather=True, group=None, as_uint8=False):
    """Render this rank's pixel range of an H x W view; with gather=True rank 0
    receives the whole (rgb [H,W,3], disp [H,W], acc [H,W]) and other ranks None.
    as_uint8=True quantises the colours on each rank first (utils.to8b on the device) and gathers
    only the [n, 3] uint8 rows -- 3 instead of 20 bytes per pixel on the wire; rank 0 gets the
    uint8 image [H, W, 3] ready for the PNG writer."""
    from . import utils
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(H * W, rank, world)
    dev = next(coarse_model.parameters()).device
    batch = utils.make_ray_batch(H, W, K, c2w, renderer.near, renderer.far, renderer.use_viewdirs,
                                 renderer.ndc, device=dev, pix0=lo, n=hi - lo)
    ret = renderer.render_batch(coarse_model, fine_model, batch, chunk, False)
    if as_uint8:
        local8 = utils.to8b(ret['rgb_map'])
        if world > 1:
            if not gather:
                return local8
            local8 = gather_rows(local8, H * W, 0, group)
        return None if local8 is None else local8.reshape(H, W, 3)
    local = pack_maps(ret)
    if world == 1:
        full = local
    elif gather:
        full = gather_rows(local, H * W, 0, group)
    else:
        return local
    if full is None:
        return None
    return full[:, 0:3].reshape(H, W, 3), full[:, 3].reshape(H, W), full[:, 4].reshape(H, W)


def render_poses_gathered(renderer, H, W, K, chunk, batch_c2w, coarse_model, fine_model, group=None, depth=2,
                          on_frame=None):
    """The C5 workload (BASELINE.json configs[4]; the pose loop of render_utils.py:293-319 over the ranks of one
    node): every pose is rendered by all ranks together -- rank r takes flat pixel range r of the frame and
    generates its own rays from (K, c2w) -- and the finished [n, 5] rows (rgb, disp, acc) are gathered to
    rank 0 with the gather of frame k overlapped with the render of frame k+1 (OverlappedGather).
    Returns the frames as [(rgb [H,W,3], disp [H,W], acc [H,W]), ...] on rank 0 (or hands each to
    on_frame(i, rgb, disp, acc) and returns the count), None / count elsewhere."""
    from . import utils
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(H * W, rank, world)
    dev = next(coarse_model.parameters()).device
    gatherer = OverlappedGather(H * W, 0, group, depth)
    frames, n_done = [], 0

    def drain(rows_list):
        nonlocal n_done
        for full in rows_list:
            if full is not None:
                img = (full[:, 0:3].reshape(H, W, 3), full[:, 3].reshape(H, W), full[:, 4].reshape(H, W))
                if on_frame is not None:
                    on_frame(n_done, *img)
                else:
                    frames.append(img)
            n_done += 1

    for c2w in batch_c2w:
        if _DIAG["on"]:
            t_a = torch.cuda.Event(enable_timing=True)
            t_a.record()
        batch = utils.make_ray_batch(H, W, K, c2w, renderer.near, renderer.far, renderer.use_viewdirs, renderer.ndc,
                                     device=dev, pix0=lo, n=hi - lo)
        ret = renderer.render_batch(coarse_model, fine_model, batch, chunk, False)
        gatherer.submit(pack_maps(ret))
        if _DIAG["on"]:
            t_b = torch.cuda.Event(enable_timing=True)
            t_b.record()
            _DIAG["frames"].append((t_a, t_b))
            if rank == 0 and world > 1:
                _DIAG["recv_bytes"] += (H * W - (hi - lo)) * 5 * 4
        if gatherer.ready_count() >= 4:            # hand finished frames on without waiting for the ones in flight
            drain(gatherer.poll_ready())
    drain(gatherer.collect())
    if on_frame is not None:
        return n_done
    return frames if rank == 0 else None


def render_poses_sharded(renderer, H, W, K, chunk, batch_c2w, coarse_model, fine_model, save_directory,
                         group=None, io_workers=4):
    """Renderer.render_from_batch_poses (render_utils.py:293-319) over the ranks of one node: whole
    frames are dealt round-robin (rank r renders poses r, r+G, ...), each rank quantises and writes
    its own 'NNN.png' files, so the only exchange is the barrier at the end.  Returns the indices of
    the frames this rank wrote."""
    import os
    from . import image_io, utils
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    os.makedirs(save_directory, exist_ok=True)
    mine = list(range(rank, len(batch_c2w), world))
    copy_stream = None
    with image_io.AsyncImageWriter(max(1, io_workers)) as writer, torch.no_grad():
        for i in mine:
            rgb = renderer.render_from_pose(H, W, K, chunk=chunk, c2w=batch_c2w[i], coarse_model=coarse_model,
                                            fine_model=fine_model)[0]
            rgb8 = utils.to8b(rgb)
            if copy_stream is None:
                copy_stream = torch.cuda.Stream(rgb8.device)
            host = torch.empty(rgb8.shape, dtype=torch.uint8, pin_memory=True)
            copy_stream.wait_stream(torch.cuda.current_stream(rgb8.device))
            with torch.cuda.stream(copy_stream):
                host.copy_(rgb8, non_blocking=True)
                done = torch.cuda.Event()
                done.record(copy_stream)
            rgb8.record_stream(copy_stream)
            writer.submit(os.path.join(save_directory, '{:03d}.png'.format(i)), host.numpy(), before=done.synchronize)
    if world > 1:
        dist.barrier(group)
    return mine
