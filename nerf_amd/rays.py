"""Ray generation -- counterpart of the reference's rays.py (get_rays :5-27, get_rays_np :30-49,
ndc_rays :52-83).  Plain torch/numpy elementwise work (<1 % of the path, stays torch)."""
import numpy as np
import torch


def get_rays(H, W, K, c2w):
    dev = c2w.device
    xs = torch.linspace(0, W - 1, W, device=dev)
    ys = torch.linspace(0, H - 1, H, device=dev)
    j, i = torch.meshgrid(ys, xs, indexing="ij")          # i: column index, j: row index, both [H, W]
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rot = c2w[:3, :3]
    rays_d = torch.sum(dirs[..., None, :] * rot, -1)       # rotate camera-frame directions into the world
    rays_o = c2w[:3, -1].expand(rays_d.shape)              # all rays start at the camera centre
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """NDC warp for forward-facing scenes.  The ORDER of the f32 operations follows rays.py:68-78
    exactly ((scale * o_x) / o_z, not scale * (o_x / o_z)): downstream sample positions are compared
    bit-for-bit with the reference, so the rays feeding them must be bit-identical too."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]          # shift origins onto the near plane
    rays_o = rays_o + t[..., None] * rays_d
    sx, sy = -1. / (W / (2. * focal)), -1. / (H / (2. * focal))
    ox, oy, oz = rays_o[..., 0], rays_o[..., 1], rays_o[..., 2]
    dx, dy, dz = rays_d[..., 0], rays_d[..., 1], rays_d[..., 2]
    o = torch.stack([sx * ox / oz, sy * oy / oz, 1. + 2. * near / oz], -1)
    d = torch.stack([sx * (dx / dz - ox / oz), sy * (dy / dz - oy / oz), -2. * near / oz], -1)
    return o, d
