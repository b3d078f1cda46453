"""Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-process; rays are independent, so
  * rendering shards the image's rows over the ranks -- no exchange during the pass, ONE gather of
    (rgb, disp, acc) = 20 B/ray to rank 0 at the end;
  * training is data-parallel: each rank takes its slice of the ray batch and ONE all-reduce(sum) of a
    flat f32 gradient buffer (both nets: 1 191 688 floats = 4.77 MB, latency-bound on xGMI) precedes
    optimizer.step().
"""
import torch
import torch.distributed as dist


def row_partition(H, world):
    """Contiguous row blocks, remainder spread over the first ranks: [(row0, nrows)] * world."""
    base, rem = divmod(H, world)
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((r0, n))
        r0 += n
    return out


def slice_partition(n, world):
    return row_partition(n, world)


def warm_up(device, group=None):
    """One tiny all-reduce and one tiny gather: RCCL connects ring and point-to-point channels on first use, which
    belongs to start-up, not to the first timed step (bench.py calls this once after init_process_group)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dev = device if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.ones(16, device=dev)
    dist.all_reduce(t, group=group)
    gathered = [torch.empty_like(t) for _ in range(dist.get_world_size(group))] if dist.get_rank(group) == 0 else None
    dist.gather(t, gathered, dst=0, group=group)
    if dev != "cpu":
        torch.cuda.synchronize()


def render_sharded(render_rows, H, W, rank=None, world=None, group=None):
    """render_rows(row0, nrows) -> [nrows*W, C] tensor on this rank's device (C = 5: rgb, disp, acc).
    Returns the [H, W, C] image on rank 0 (None elsewhere)."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    parts = row_partition(H, world)
    row0, nrows = parts[rank]
    mine = render_rows(row0, nrows)
    if world == 1:
        return mine.reshape(H, W, -1)
    C = mine.shape[-1]
    maxrows = max(n for _, n in parts)
    buf = mine.new_zeros((maxrows * W, C))
    buf[: nrows * W] = mine
    dev = buf.device
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        buf = buf.cpu()             # gloo gathers host tensors only (CPU tests, rehearsals); RCCL gathers in place
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([g[: n * W] for g, (_, n) in zip(gathered, parts)], 0).reshape(H, W, C).to(dev)


def render_rows(H, W, K, c2w, row0, nrows, chunk=1024 * 32, ndc=True, near=0., far=1., use_viewdirs=False,
                c2w_staticcam=None, **kwargs):
    """Rows row0 .. row0+nrows-1 of render(H, W, K, c2w=...) as one [nrows*W, 5] tensor (rgb, disp, acc): a rank's
    share of an image is a pure (row range, pose, intrinsics) descriptor -- no ray data is moved between ranks."""
    from . import ops
    from .render import batchify_rays
    ray_matrix = ops.gen_rays(H, W, K, c2w, near, far, use_viewdirs, ndc, c2w_staticcam, row0=row0, nrows=nrows)
    ret = batchify_rays(ray_matrix, chunk, **kwargs)
    return torch.cat([ret['rgb_map'], ret['disp_map'][:, None], ret['acc_map'][:, None]], -1)


class FlatGradAllReduce:
    """Averages the gradients of `params` over the ranks with ONE all-reduce of a flat buffer."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self):
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        # flatten with ONE kernel (cat), reduce, scale, scatter back with one fused multi-tensor copy: a handful of
        # launches instead of two per parameter -- at 512 rays per rank the step is short enough for that to show
        missing = [p for p in self.params if p.grad is None]
        for p in missing:
            p.grad = torch.zeros_like(p)
        grads = [p.grad for p in self.params]
        self.flat = torch.cat([g.reshape(-1) for g in grads])
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":   # CPU tests / rehearsals: staged on the host
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / world)
        pieces = [t.view_as(g) for t, g in zip(self.flat.split([g.numel() for g in grads]), grads)]
        torch._foreach_copy_(grads, pieces)
