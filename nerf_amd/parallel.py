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


def render_sharded(render_rows, H, W, rank=None, world=None, group=None, timings=None):
    """render_rows(row0, nrows) -> [nrows*W, C] tensor on this rank's device (C = 5: rgb, disp, acc).
    Returns the [H, W, C] image on rank 0 (None elsewhere).  timings: optional dict; a (start, end) pair of HIP events
    around the gather is appended to timings["gather"] (read them after a synchronise)."""
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
    ev = None
    if timings is not None and buf.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.gather(buf, gathered, dst=0, group=group)
    if ev is not None:
        ev[1].record()
        timings.setdefault("gather", []).append(ev)
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


def render_rows_dnerf(H, W, K, c2w, time_vals, row0, nrows, chunk=1024 * 32, near=0., far=1., use_viewdirs=False,
                      c2w_staticcam=None, **kwargs):
    """Rows row0 .. row0+nrows-1 of render_dnerf(H, W, K, time_vals, c2w=...) (vanilla_dynamic/render.py:368-433) as one
    [nrows*W, 5] tensor (rgb, disp, acc).  time_vals: a scalar, or a per-pixel tensor of H*W times of which this rank
    takes its rows -- like the rays, a rank's share of the times is a row range, nothing moves between ranks."""
    from . import ops
    from .vanilla_dynamic.render import batchify_rays_dnerf
    ray_matrix = ops.gen_rays(H, W, K, c2w, near, far, use_viewdirs, False, c2w_staticcam, row0=row0, nrows=nrows)
    n = ray_matrix.shape[0]
    if isinstance(time_vals, torch.Tensor) and time_vals.numel() > 1:
        tv = time_vals.reshape(-1, 1).float().to(ray_matrix.device)
        if tv.shape[0] != H * W:            # wrong count: the reference broadcasts the mean (render.py:414-416)
            tv = torch.ones(n, 1, device=ray_matrix.device) * tv.mean()
        else:
            tv = tv[row0 * W:(row0 + nrows) * W]
    else:
        tv = torch.ones(n, 1, device=ray_matrix.device) * float(time_vals)
    ret = batchify_rays_dnerf(ray_matrix, tv, chunk, **kwargs)
    return torch.cat([ret['rgb_map'], ret['disp_map'][:, None], ret['acc_map'][:, None]], -1)


class BucketedGradAllReduce:
    """Data-parallel gradient averaging overlapped with the backward pass: `buckets` are lists of
    parameters in the order their gradients become ready -- for train.py:232-242's loss that is the FINE net (its branch
    of the graph was built last, so autograd runs it first) and then the coarse net.  When the last gradient of a bucket
    has been accumulated, the bucket is flattened and its all-reduce is started on a side stream while the backward of
    the other net is still running on the main stream; `finish()` (call it after loss.backward()) waits, scales by
    1/world and scatters the averages back into the .grad tensors.  Two messages of 2.4 MB instead of one of 4.8 MB:
    both are latency-bound on xGMI, the first one now costs nothing.

    Semantics are those of one flat all-reduce (train.py's single-process gradient, averaged over ranks): a bucket whose
    hook never fired on this rank (a parameter without gradient) is reduced at finish() with zeros in its place, and the
    parameter then HAS a gradient (the mean over the ranks, zero if no rank produced one; Adam moves it by its momentum
    where a single process would have skipped a grad=None parameter -- FlatGradAllReduce does the same).

    Collectives are issued in BUCKET-INDEX order on every rank, whatever order the hooks fire in: a bucket that becomes
    ready before a lower-numbered one is queued until that one has been launched (finish() launches what is left, in
    order), so ranks whose graphs differ cannot pair all-reduces of different buckets.  One backward() per finish(): a
    gradient arriving for a bucket whose all-reduce is already in flight raises."""

    def __init__(self, buckets, group=None, min_world=2):
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.group = group
        self.min_world = min_world   # 1: run the collectives even in a one-rank group (single-GPU test of the stream logic)
        self._pending = [set(id(p) for p in b) for b in self.buckets]
        self._inflight = {}          # bucket index -> (flat, work, grads)
        self._ready = set()          # buckets whose gradients are complete but that wait for a lower index to launch
        self._next = 0               # lowest bucket index not launched yet
        self.launch_order = []       # bucket indices in the order their all-reduces were started (tests read this)
        self._side = None
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_hook(self, bi):
        def hook(p):
            if bi in self._inflight:
                raise RuntimeError("BucketedGradAllReduce: a gradient arrived for a bucket whose all-reduce is already in "
                                   "flight (a second backward() before finish()?): call finish() after every backward()")
            pend = self._pending[bi]
            pend.discard(id(p))
            if not pend:
                self._ready.add(bi)
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Start every ready bucket whose predecessors have all been started: index order on every rank."""
        while self._next < len(self.buckets) and self._next in self._ready:
            self._ready.discard(self._next)
            self._launch(self._next)
            self._next += 1

    def _overlapped(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "nccl"

    def _launch(self, bi):
        if not dist.is_initialized() or dist.get_world_size(self.group) < self.min_world:
            return
        params = self.buckets[bi]
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in params]
        self.launch_order.append(bi)
        if self._overlapped(grads[0]):
            if self._side is None:
                self._side = torch.cuda.Stream()
            ready = torch.cuda.Event()
            ready.record()                                   # this bucket's gradients are complete here
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                flat = torch.cat([g.reshape(-1) for g in grads])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._inflight[bi] = (flat, work, grads)
        else:                                                # gloo (CPU tests, rehearsals): same order, staged on the host
            flat = torch.cat([g.reshape(-1) for g in grads])
            host = flat.cpu() if flat.is_cuda else flat
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            if host is not flat:
                flat.copy_(host)
            self._inflight[bi] = (flat, None, grads)

    def finish(self):
        """After loss.backward(): every bucket reduced, averaged and written back.  Returns the launch order."""
        if not dist.is_initialized() or dist.get_world_size(self.group) < self.min_world:
            self._reset()
            return []
        world = dist.get_world_size(self.group)
        for bi in range(self._next, len(self.buckets)):      # queued buckets and those whose last hook never fired here
            self._launch(bi)
        self._next = len(self.buckets)
        for bi, (flat, work, grads) in self._inflight.items():
            if work is not None:
                work.wait()                                  # the current stream waits for the collective
                torch.cuda.current_stream().wait_stream(self._side)
                flat.record_stream(torch.cuda.current_stream())
            flat.mul_(1.0 / world)
            pieces = [t.view_as(g) for t, g in zip(flat.split([g.numel() for g in grads]), grads)]
            torch._foreach_copy_(grads, pieces)
        order = list(self.launch_order)
        self._reset()
        return order

    __call__ = finish

    def _reset(self):
        self._pending = [set(id(p) for p in b) for b in self.buckets]
        self._inflight = {}
        self._ready = set()
        self._next = 0
        self.launch_order = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


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
