"""Volumetric rendering -- counterpart of the reference's render.py, same call signatures:
batchify :14-28, run_network :31-56, batchify_rays :59-79, raw2outputs :82-131, render_rays :134-244,
render :247-315, render_path :318-371.

What runs where (all on one HIP stream, no host synchronisation inside render_rays):
  coarse depths / points      nrf_sample_coarse, nrf_points       (bit-exact with the reference)
  network query               nrf_query_fwd (PE + MLP fused)      via NeRF.query_points
  alpha compositing           nrf_composite_fwd / _bwd            via _CompositeFunction
  hierarchical resampling     nrf_sample_fine                     (bit-exact on identical inputs)
`chunk` / `netchunk` keep their meaning as upper bounds on work per call; results do not depend on them.
"""
import os
import time

import numpy as np
import torch

from . import ops
from .network import NeRF
from .positional_encoding import _EmbedFn
from .rays import get_rays, ndc_rays
from .utils import DEBUG, to8b


def batchify(fn, chunk):
    """fn applied to slices of `chunk` rows, results concatenated (chunk None: fn itself)."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def _fusable(fn, embed_fn, embeddirs_fn, viewdirs):
    if not isinstance(fn, NeRF) or not isinstance(embed_fn, _EmbedFn) or not embed_fn.fusable:
        return False
    if fn.kernel_shape_reason() is not None:      # e.g. --netwidth 96: the generic route below, fn composes GPU torch ops
        return False
    if embed_fn.out_dim != fn.input_ch or embed_fn.multires > 10:
        return False
    if fn.use_viewdirs:
        if viewdirs is None or not isinstance(embeddirs_fn, _EmbedFn) or not embeddirs_fn.fusable:
            return False
        if embeddirs_fn.out_dim != fn.input_ch_views or embeddirs_fn.multires > 4:
            return False
    elif fn.input_ch_views != 0:
        return False
    return True


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """inputs [..., 3] points, viewdirs [N_rays, 3] or None -> [..., out_ch]."""
    if _fusable(fn, embed_fn, embeddirs_fn, viewdirs):
        flat = inputs.reshape(-1, inputs.shape[-1])
        n_rays = viewdirs.shape[0] if viewdirs is not None else 1
        pts_per_ray = flat.shape[0] // n_rays if viewdirs is not None else 1
        out = fn.query_points(flat, viewdirs, pts_per_ray, embed_fn.multires,
                              embeddirs_fn.multires if fn.use_viewdirs else 0)
        return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])
    # generic composition (arbitrary fn / embedders): embed, concatenate, evaluate in netchunk slices
    flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(flat)
    if viewdirs is not None:
        dirs = viewdirs[:, None].expand(inputs.shape)
        embedded = torch.cat([embedded, embeddirs_fn(torch.reshape(dirs, [-1, dirs.shape[-1]]))], -1)
    out = batchify(fn, netchunk)(embedded)
    return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """render_rays over `chunk`-ray slices, per-key concatenation."""
    pieces = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k, v in ret.items():
            pieces.setdefault(k, []).append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in pieces.items()}


class _CompositeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise, white_bkgd):
        rgb, disp, acc, weights, depth = ops.composite_fwd(raw, z_vals, rays_d, noise, white_bkgd)
        ctx.save_for_backward(raw, z_vals, rays_d, noise if noise is not None else raw.new_empty(0), disp, acc, depth)
        ctx.white_bkgd = white_bkgd
        ctx.set_materialize_grads(False)   # unused outputs (disp, acc, weights, depth) arrive as None
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_weights, g_depth):
        raw, z_vals, rays_d, noise, disp, acc, depth = ctx.saved_tensors
        noise = noise if noise.numel() else None
        if g_rgb is None:
            g_rgb = torch.zeros((raw.shape[0], 3), device=raw.device)
        if g_disp is not None:
            # disp = 1 / max(1e-10, depth/acc)  (render.py:125): fold its gradient into depth and acc
            q = depth / acc
            live = (q > 1e-10).to(raw.dtype)
            g_q = -g_disp * disp * disp * live
            g_depth = (g_depth if g_depth is not None else 0) + g_q / acc
            g_acc = (g_acc if g_acc is not None else 0) - g_q * q / acc
        g_raw = ops.composite_bwd(raw, z_vals, rays_d, g_rgb.contiguous(), g_acc, g_depth, g_weights, noise,
                                  ctx.white_bkgd)
        return g_raw, None, None, None, None


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """[N,S,4] raw predictions -> (rgb_map [N,3], disp_map [N], acc_map [N], weights [N,S], depth_map [N])."""
    if raw.shape[-1] > 4:
        # model.py:30 builds output_ch = 5 nets when N_importance > 0 and the reference reads raw[..., :3] and
        # raw[..., 3] only (render.py:100-103); autograd returns zeros for the channels left out
        raw = raw[..., :4]
    noise = None
    if raw_noise_std > 0.:
        if pytest:  # the reference's fixed-random route draws UNIFORM numbers here (render.py:114-117)
            np.random.seed(0)
            noise = torch.tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std, dtype=torch.float32,
                                 device=raw.device)
        else:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
    return _CompositeFunction.apply(raw, z_vals, rays_d, noise, bool(white_bkgd))


def render_rays(ray_batch,
                network_fn,
                network_query_fn,
                N_samples,
                retraw=False,
                lindisp=False,
                perturb=0.,
                N_importance=0,
                network_fine=None,
                white_bkgd=False,
                raw_noise_std=0.,
                verbose=False,
                pytest=False):
    """ray_batch [N, 8|11] = o(3) d(3) near far [viewdirs(3)] -> dict (keys of render.py:231-238)."""
    if not ray_batch.is_cuda:
        raise RuntimeError("nerf_amd.render_rays: ray_batch is on the CPU; nerf_amd has no CPU path")
    ray_batch = ray_batch.float().contiguous()
    N_rays = ray_batch.shape[0]
    rays_d = ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None

    t_rand = None
    if perturb > 0.:
        if pytest:
            np.random.seed(0)
            t_rand = torch.tensor(np.random.rand(N_rays, N_samples), dtype=torch.float32, device=ray_batch.device)
        else:
            t_rand = torch.rand((N_rays, N_samples), device=ray_batch.device)
    z_vals = ops.sample_coarse(ray_batch, N_samples, lindisp, t_rand)
    pts = ops.points(ray_batch, z_vals)

    raw = network_query_fn(pts, viewdirs, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                 pytest=pytest)
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        u = None
        if pytest:
            np.random.seed(0)
            u = np.linspace(0., 1., N_importance) if perturb == 0. else np.random.rand(N_rays, N_importance)
            u = torch.tensor(np.ascontiguousarray(np.broadcast_to(u, (N_rays, N_importance))), dtype=torch.float32,
                             device=ray_batch.device)
        elif perturb != 0.:
            u = torch.rand((N_rays, N_importance), device=ray_batch.device)
        # mid-points, inverse-CDF samples on weights[...,1:-1], sort(cat) and std in one kernel; no gradient
        # flows through the sample positions (render.py:221)
        _, z_vals, z_std = ops.sample_fine(z_vals, weights.detach(), N_importance, u, want_samples=False)
        pts = ops.points(ray_batch, z_vals)
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                     pytest=pytest)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'] = rgb_map_0, disp_map_0, acc_map_0
        ret['z_std'] = z_std
    if DEBUG:  # the reference evaluates this scan (and its host syncs) even with DEBUG off (render.py:240-242)
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def _ray_matrix_torch(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam):
    """render.py:276-304 with torch ops: explicit ray batches (training) and host-side poses."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays

    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:  # camera from one pose, view directions from another
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = torch.reshape(viewdirs, [-1, 3]).float()

    sh = rays_d.shape
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)

    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    near_t = near * torch.ones_like(rays_d[..., :1])
    far_t = far * torch.ones_like(rays_d[..., :1])
    cols = [rays_o, rays_d, near_t, far_t] + ([viewdirs] if use_viewdirs else [])
    return torch.cat(cols, -1), sh


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True,
           near=0., far=1.,
           use_viewdirs=False, c2w_staticcam=None,
           **kwargs):
    """Full image (c2w given) or an explicit ray batch rays=[2, N, 3] -> [rgb, disp, acc, extras]."""
    if c2w is not None and isinstance(c2w, torch.Tensor) and c2w.is_cuda:
        # whole-image case: the ray matrix comes from one kernel (get_rays + viewdirs + NDC + bounds fused)
        ray_matrix = ops.gen_rays(H, W, K, c2w, near, far, use_viewdirs, ndc, c2w_staticcam)
        sh = (H, W, 3)
    else:
        ray_matrix, sh = _ray_matrix_torch(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)

    all_ret = batchify_rays(ray_matrix, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))

    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]


class RenderGraph:
    """render() of one whole frame captured ONCE as a HIP graph and replayed per pose: the frames of a path share every
    shape, so the ~250 launches of a frame (20 chunks x sampling, points, two queries, two composites, concatenation)
    become one graph launch and the GPU no longer waits on the host between the short kernels.  Results are bit-identical
    to render().  Deterministic renders only (perturb = 0, raw_noise_std = 0: a captured random draw would repeat);
    static NeRF nets.  The returned tensors are the graph's own buffers: valid until the next call, clone to keep."""

    def __init__(self, H, W, K, chunk=1024 * 32, device="cuda", **render_kwargs):
        if render_kwargs.get("perturb", 0.) or render_kwargs.get("raw_noise_std", 0.):
            raise ValueError("nerf_amd.RenderGraph: perturb and raw_noise_std must be 0 (captured noise would repeat)")
        if render_kwargs.get("rays") is not None:
            raise ValueError("nerf_amd.RenderGraph renders whole frames from a pose (c2w), not explicit ray batches")
        K = np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K, dtype=np.float64)   # host scalars only
        self._nets = [n for n in (render_kwargs.get("network_fn"), render_kwargs.get("network_fine")) if n is not None]
        self.c2w = torch.eye(4, device=device, dtype=torch.float32)[:3, :4].contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():       # warm-up off the capture: packs weights, sizes the pools
            render(H, W, K, chunk=chunk, c2w=self.c2w, **render_kwargs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.rgb, self.disp, self.acc, self.extras = render(H, W, K, chunk=chunk, c2w=self.c2w, **render_kwargs)

    def __call__(self, c2w):
        for net in self._nets:                               # parameters may have changed since the capture: the packed
            for pk in getattr(net, "_packed_cache", {}).values():   # copies the graph reads are refreshed in place
                pk.refresh()
        self.c2w.copy_(torch.as_tensor(c2w)[:3, :4], non_blocking=True)
        self.graph.replay()
        return self.rgb, self.disp, self.acc, self.extras


class _FrameDrain:
    """Streaming of finished frames to the host: frame i's device->host copy, 8-bit quantisation and
    PNG write happen after frame i+1's kernels have been enqueued, so the GPU never idles on host work.  The copy
    goes through pinned memory on a side stream; the only host wait is on the event of the frame being drained."""

    def __init__(self, savedir, depth=1):
        self.savedir, self.depth = savedir, depth
        self.pending, self.rgbs, self.disps = [], [], []
        self.copy_stream = torch.cuda.Stream()

    def push(self, i, rgb, disp):
        ready = torch.cuda.Event()
        ready.record()                                   # on the render stream: frame i is complete here
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            h_rgb = torch.empty(rgb.shape, dtype=rgb.dtype, pin_memory=True)
            h_disp = torch.empty(disp.shape, dtype=disp.dtype, pin_memory=True)
            h_rgb.copy_(rgb, non_blocking=True)
            h_disp.copy_(disp, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        self.pending.append((i, rgb, disp, h_rgb, h_disp, done))   # rgb/disp kept alive until the copy is done
        while len(self.pending) > self.depth:
            self._drain_one()

    def _drain_one(self):
        i, _, _, h_rgb, h_disp, done = self.pending.pop(0)
        done.synchronize()
        self.rgbs.append(h_rgb.numpy())
        self.disps.append(h_disp.numpy())
        if self.savedir is not None:
            _write_png(os.path.join(self.savedir, '{:03d}.png'.format(i)), to8b(self.rgbs[-1]))

    def finish(self):
        while self.pending:
            self._drain_one()
        return np.stack(self.rgbs, 0), np.stack(self.disps, 0)


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0, graph=False):
    """One render() per pose -> (rgbs [N,H,W,3], disps [N,H,W]) as numpy; PNGs into savedir if given.  Same results
    as the reference's loop (render.py:347-366); host-side work of frame i overlaps the rendering of frame i+1.
    graph=True (not in the reference's signature) captures the frame once as a HIP graph and replays it per pose."""
    H, W, focal = hwf
    if render_factor != 0:
        # NB: like the reference (render.py:337-341) K itself is not rescaled here
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor

    drain = _FrameDrain(savedir)
    t0 = time.time()
    frame = RenderGraph(H, W, K, chunk=chunk, **render_kwargs) if graph else None
    for i, c2w in enumerate(render_poses):
        if frame is not None:
            rgb, disp, _, _ = frame(c2w)
            rgb, disp = rgb.clone(), disp.clone()        # the graph's buffers are overwritten by the next frame
        else:
            rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        drain.push(i, rgb, disp)
        if i == 0:
            print(rgb.shape, disp.shape, f"first frame enqueued after {time.time() - t0:.3f}s")
    return drain.finish()


def _write_png(path, rgb8):
    try:
        import imageio
        imageio.imwrite(path, rgb8)
    except ImportError:  # image writing is outside the rendering path; fall back to a raw dump
        np.save(os.path.splitext(path)[0] + ".npy", rgb8)
