"""Dynamic-scene volumetric rendering -- counterpart of vanilla_dynamic/render.py, same signatures:
batchify :23-46, run_network_dnerf :49-124, raw2outputs :127-174, sample_pdf :177-215,
render_rays_dnerf :218-326, batchify_rays_dnerf :329-350, get_rays :353-365, render_dnerf :368-433,
render_path_dnerf :436-493.

Kernels: the static path's sampling / compositing kernels (same arithmetic in the reference, copy for copy)
plus nrf_dnerf_query_fwd, which fuses gamma(x), gamma(t), the time net, x + dx, the re-embedding, gamma(d) and
the canonical net into one launch per query.
"""
import os
import time

import numpy as np
import torch

from .. import ops
from ..positional_encoding import _EmbedFn
from ..rays import get_rays  # noqa: F401  (same arithmetic as render.py:353-365)
from ..render import _FrameDrain, _write_png, raw2outputs  # noqa: F401
from ..sampling import sample_pdf  # noqa: F401
from ..utils import DEBUG, to8b
from .network import DNeRF_Deformation, DNeRF_Straightforward


def batchify(fn, chunk, is_dynamic=True):
    if chunk is None:
        return fn

    def ret(inputs):
        results = [fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)]
        if is_dynamic:
            return torch.cat([r[0] for r in results], 0), torch.cat([r[1] for r in results], 0)
        return torch.cat(results, 0)
    return ret


def _fusable(fn, embed_fn, embeddirs_fn, embed_time_fn, viewdirs):
    if not isinstance(fn, DNeRF_Deformation) or not fn.use_viewdirs or viewdirs is None \
            or (fn.use_layernorm and fn._training_call()) or fn.kernel_shape_reason() is not None:
        return False
    for e, ch, lmax in ((embed_fn, fn.input_ch, 10), (embeddirs_fn, fn.input_ch_views, 4),
                        (embed_time_fn, fn.input_ch_time, 10)):
        if not isinstance(e, _EmbedFn) or not e.fusable or e.out_dim != ch or e.multires > lmax:
            return False
    # the network re-embeds x + dx with its own embed_fn (network.py:332-333); fuse only when it is the same map
    own = fn.embed_fn
    return isinstance(own, _EmbedFn) and own.fusable and own.multires == embed_fn.multires


def run_network_dnerf(inputs, viewdirs, times, fn, embed_fn, embeddirs_fn, embed_time_fn, netchunk=1024 * 64,
                      network_type='deformation'):
    """inputs [B, N, 3]; viewdirs [B, 3] or None; times [B, N, 1] or [B, 1] -> (outputs [B, N, C], dx [B, N, 3])."""
    B, N, _ = inputs.shape
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    per_point = times.dim() == 3 and times.shape[1] == N

    if _fusable(fn, embed_fn, embeddirs_fn, embed_time_fn, viewdirs):
        if per_point and times.stride(1) == 0:      # the per-ray times of render_rays_dnerf, expanded
            t_arg, per_point = times[:, 0, 0], False
        elif per_point:
            t_arg = times.reshape(-1)
        else:
            t_arg = times.reshape(B)
        raw, dx = fn.query_points(inputs_flat, viewdirs, t_arg, N, per_point, netchunk, embed_fn.multires,
                                  embeddirs_fn.multires, embed_time_fn.multires)
        return raw.reshape(B, N, raw.shape[-1]), dx.reshape(B, N, 3)

    # DNeRF_Straightforward.  At the reference's HEAD this call raises TypeError (the loop below passes pts_raw= / t_raw=,
    # which that forward(x, ts) does not take: vanilla_dynamic/render.py:112 vs network.py:151) although eight shipped configs
    # name the network; here it is evaluated as its forward defines -- the one deliberate deviation from the reference's
    # behaviour on this path.  With the standard embedders: one fused launch on raw points (training: the fused
    # forward that saves activations + the HIP backward kernels).
    straight = isinstance(fn, DNeRF_Straightforward)
    if straight and viewdirs is not None and inputs.is_cuda and not inputs.requires_grad and not times.requires_grad \
            and all(isinstance(e, _EmbedFn) and e.fusable and e.out_dim == ch and e.multires <= lmax
                    for e, ch, lmax in ((embed_fn, fn.input_ch, 10), (embeddirs_fn, fn.input_ch_views, 4),
                                        (embed_time_fn, fn.input_ch_time, 10))) \
            and fn._kernel_takes():
        if per_point and times.stride(1) == 0:
            t_arg, pp = times[:, 0, 0], False
        elif per_point:
            t_arg, pp = times.reshape(-1), True
        else:
            t_arg, pp = times.reshape(B), False
        raw = fn.query_points(inputs_flat, viewdirs, t_arg, N, pp, embed_fn.multires, embeddirs_fn.multires,
                              embed_time_fn.multires)
        return raw.reshape(B, N, raw.shape[-1]), torch.zeros_like(inputs)

    # generic composition: arbitrary networks / embedders, netchunk slices as in the reference
    embedded = embed_fn(inputs_flat)
    if per_point:
        t_flat = torch.reshape(times, [-1, 1])
    else:
        t_exp = times[:, None].expand([B, N, 1]) if times.dim() == 2 else times.unsqueeze(1).expand([B, N, 1])
        t_flat = torch.reshape(t_exp, [-1, 1])
    embedded_time = embed_time_fn(t_flat)
    if viewdirs is not None:
        dirs = viewdirs[:, None].expand(inputs.shape)
        embedded = torch.cat([embedded, embeddirs_fn(torch.reshape(dirs, [-1, dirs.shape[-1]]))], -1)
    outs, dxs = [], []
    for i in range(0, embedded.shape[0], netchunk):
        if straight:
            out, dx = fn(embedded[i:i + netchunk], [embedded_time[i:i + netchunk]] * 2)
        else:
            out, dx = fn(embedded[i:i + netchunk], [embedded_time[i:i + netchunk]] * 2,
                         pts_raw=inputs_flat[i:i + netchunk], t_raw=t_flat[i:i + netchunk])
        outs.append(out)
        dxs.append(dx)
    outputs = torch.cat(outs, 0)
    return (torch.reshape(outputs, list(inputs.shape[:-1]) + [outputs.shape[-1]]),
            torch.reshape(torch.cat(dxs, 0), list(inputs.shape[:-1]) + [3]))


def render_rays_dnerf(ray_batch, time_vals,
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
                      pytest=False,
                      network_type='deformation',
                      ndc=False,
                      use_viewdirs=False,
                      **kwargs):
    """ray_batch [N, 8|11], time_vals [N, 1] -> dict (keys of render.py:312-320, incl. dx / dx0)."""
    if not ray_batch.is_cuda:
        raise RuntimeError("nerf_amd.render_rays_dnerf: ray_batch is on the CPU; nerf_amd has no CPU path")
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
    times = time_vals.unsqueeze(1).expand(-1, pts.shape[1], -1)

    raw, dx = network_query_fn(pts, viewdirs, times, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                 pytest=pytest)
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0, dx_0 = rgb_map, disp_map, acc_map, dx
        u = None
        if pytest:
            np.random.seed(0)
            u = np.linspace(0., 1., N_importance) if perturb == 0. else np.random.rand(N_rays, N_importance)
            u = torch.tensor(np.ascontiguousarray(np.broadcast_to(u, (N_rays, N_importance))), dtype=torch.float32,
                             device=ray_batch.device)
        elif perturb != 0.:
            u = torch.rand((N_rays, N_importance), device=ray_batch.device)
        _, z_vals, z_std = ops.sample_fine(z_vals, weights.detach(), N_importance, u, want_samples=False)
        pts = ops.points(ray_batch, z_vals)
        times = time_vals.unsqueeze(1).expand(-1, pts.shape[1], -1)
        run_fn = network_fn if network_fine is None else network_fine
        raw, dx = network_query_fn(pts, viewdirs, times, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                     pytest=pytest)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'dx': dx}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'], ret['dx0'] = rgb_map_0, disp_map_0, acc_map_0, dx_0
        ret['z_std'] = z_std
    if DEBUG:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def batchify_rays_dnerf(rays_flat, time_vals, chunk=1024 * 32, **kwargs):
    pieces = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays_dnerf(rays_flat[i:i + chunk], time_vals[i:i + chunk], **kwargs)
        for k, v in ret.items():
            pieces.setdefault(k, []).append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in pieces.items()}


def render_dnerf(H, W, K, time_vals, chunk=1024 * 32, rays=None, c2w=None,
                 near=0., far=1., use_viewdirs=False, c2w_staticcam=None, **kwargs):
    """time_vals: scalar or per-ray tensor -> [rgb, disp, acc, extras]."""
    if c2w is not None and isinstance(c2w, torch.Tensor) and c2w.is_cuda:
        ray_matrix = ops.gen_rays(H, W, K, c2w, near, far, use_viewdirs, False, c2w_staticcam)   # one kernel
        sh = (H, W, 3)
    else:
        if c2w is not None:
            rays_o, rays_d = get_rays(H, W, K, c2w)
        else:
            rays_o, rays_d = rays
        if use_viewdirs:
            viewdirs = rays_d
            if c2w_staticcam is not None:
                rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
            viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
            viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
        sh = rays_d.shape
        rays_o = torch.reshape(rays_o, [-1, 3]).float()
        rays_d = torch.reshape(rays_d, [-1, 3]).float()
        near_t, far_t = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
        ray_matrix = torch.cat([rays_o, rays_d, near_t, far_t] + ([viewdirs] if use_viewdirs else []), -1)

    n_rays = ray_matrix.shape[0]
    if isinstance(time_vals, torch.Tensor):
        time_vals_flat = time_vals.reshape(-1, 1).float().to(ray_matrix.device)
        if time_vals_flat.shape[0] != n_rays:   # wrong count: the reference broadcasts the mean (render.py:414-416)
            time_vals_flat = torch.ones(n_rays, 1, device=ray_matrix.device) * time_vals_flat.mean()
    else:
        time_vals_flat = torch.ones(n_rays, 1, device=ray_matrix.device) * float(time_vals)

    all_ret = batchify_rays_dnerf(ray_matrix, time_vals_flat, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))

    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]


def render_path_dnerf(render_poses, render_times, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None,
                      render_factor=0):
    """One render_dnerf() per (pose, time) -> (rgbs [N,H,W,3], disps [N,H,W]) numpy.  Unlike the static
    render_path, K IS rescaled by render_factor here (render.py:457-465)."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
        K = K.copy() if isinstance(K, np.ndarray) else K.clone()
        K[0, 0] = focal
        K[1, 1] = focal
        K[0, 2] = W / 2.
        K[1, 2] = H / 2.

    drain = _FrameDrain(savedir)          # host-side work of frame i overlaps the rendering of frame i+1
    t0 = time.time()
    for i, (c2w, time_val) in enumerate(zip(render_poses, render_times)):
        if isinstance(time_val, torch.Tensor):
            time_val = time_val.item()
        rgb, disp, acc, _ = render_dnerf(H, W, K, time_val, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        drain.push(i, rgb, disp)
        if i == 0:
            print(f"First frame enqueued after {time.time() - t0:.2f}s")
            print(f"RGB shape: {rgb.shape}, Disp shape: {disp.shape}")
    return drain.finish()
