"""Deterministic synthetic inputs shared by the golden-vector generator, the tests and bench.py.

Everything here is numpy-only and seed-driven so that the same weights / poses / rays can be
rebuilt on the GPU box (where /root/reference does not exist) and fed both to the oracle and to
the HIP path.  Nothing in this file comes from the reference except two *formulas*, cited below.
"""
import math
from collections import OrderedDict

import numpy as np


def _linear(rng, out_f, in_f):
    # same family as nn.Linear's default init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias
    bound = 1.0 / math.sqrt(in_f)
    w = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    return w, b


def make_nerf_state(seed, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,),
                    use_viewdirs=True, peaky=False, prefix=""):
    """state_dict (numpy) with the key names / shapes of reference network.py:29-44."""
    rng = np.random.RandomState(seed)
    sd = OrderedDict()
    in_f = input_ch
    for i in range(D):
        w, b = _linear(rng, W, in_f)
        sd[f"{prefix}pts_linears.{i}.weight"] = w
        sd[f"{prefix}pts_linears.{i}.bias"] = b
        in_f = W + input_ch if i in skips else W
    w, b = _linear(rng, W // 2, input_ch_views + W)
    sd[f"{prefix}views_linears.0.weight"] = w
    sd[f"{prefix}views_linears.0.bias"] = b
    if use_viewdirs:
        w, b = _linear(rng, W, W)
        sd[f"{prefix}feature_linear.weight"] = w
        sd[f"{prefix}feature_linear.bias"] = b
        w, b = _linear(rng, 1, W)
        sd[f"{prefix}alpha_linear.weight"] = w
        sd[f"{prefix}alpha_linear.bias"] = b
        w, b = _linear(rng, 3, W // 2)
        sd[f"{prefix}rgb_linear.weight"] = w
        sd[f"{prefix}rgb_linear.bias"] = b
        if peaky:
            # a raw random-init net gives near-uniform weights along a ray; scale the heads so that
            # sample_pdf sees peaked pdfs
            sd[f"{prefix}alpha_linear.weight"] = (sd[f"{prefix}alpha_linear.weight"] * 30).astype(np.float32)
            sd[f"{prefix}rgb_linear.weight"] = (sd[f"{prefix}rgb_linear.weight"] * 5).astype(np.float32)
    else:
        w, b = _linear(rng, output_ch, W)
        if peaky:
            w = (w * 20).astype(np.float32)
        sd[f"{prefix}output_linear.weight"] = w
        sd[f"{prefix}output_linear.bias"] = b
    return sd


def make_dnerf_state(seed, D=8, W=256, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                     input_ch_time=21, skips=(4,), peaky=False, dx_scale=1.0):
    """state_dict of vanilla_dynamic DNeRF_Deformation (network.py:235-251): `_occ.*`, `_time.*`, `_time_out.*`."""
    sd = OrderedDict()
    sd.update(make_nerf_state(seed, D, W, input_ch, input_ch_views, 4, skips, True, peaky, prefix="_occ."))
    rng = np.random.RandomState(seed + 7919)
    in_f = input_ch + input_ch_time
    for i in range(D_deform):
        w, b = _linear(rng, W_deform, in_f)
        sd[f"_time.{i}.weight"] = w
        sd[f"_time.{i}.bias"] = b
        in_f = W_deform + input_ch if i in skips else W_deform
    w, b = _linear(rng, 3, W_deform)
    sd["_time_out.weight"] = (w * dx_scale).astype(np.float32)
    sd["_time_out.bias"] = (b * dx_scale).astype(np.float32)
    return sd


def add_layernorm_state(sd, seed, prefix, names):
    """LayerNorm parameters (vanilla_dynamic/network.py:46-48, :275-277) away from their (1, 0) init, so that a
    kernel that dropped the affine part would be caught.  `names` = [(module name, width), ...]."""
    rng = np.random.RandomState(seed)
    for name, width in names:
        sd[f"{prefix}{name}.weight"] = rng.uniform(0.5, 1.5, size=(width,)).astype(np.float32)
        sd[f"{prefix}{name}.bias"] = rng.uniform(-0.2, 0.2, size=(width,)).astype(np.float32)
    return sd


def make_dnerf_layernorm_state(seed, D=8, W=256, D_deform=8, W_deform=256, skips=(4,), dx_scale=1.0):
    """state_dict of DNeRF_Deformation(use_layernorm=True): make_dnerf_state + `_occ.pts_layernorms.*`,
    `_occ.views_layernorm.*`, `_time_layernorms.*`."""
    sd = make_dnerf_state(seed, D=D, W=W, D_deform=D_deform, W_deform=W_deform, skips=skips, dx_scale=dx_scale)
    add_layernorm_state(sd, seed + 101, "_occ.", [(f"pts_layernorms.{i}", W) for i in range(D)] + [("views_layernorm", W // 2)])
    add_layernorm_state(sd, seed + 202, "", [(f"_time_layernorms.{i}", W_deform) for i in range(D_deform)])
    return sd


def make_straightforward_state(seed, D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, skips=(4,)):
    """state_dict of DNeRF_Straightforward (vanilla_dynamic/network.py:135-149): the NeRF layout with
    gamma(x) ++ gamma(t) as the trunk input (and as the skip)."""
    return make_nerf_state(seed, D, W, input_ch + input_ch_time, input_ch_views, 4, skips, True)


def pose_spherical(theta_deg, phi_deg, radius):
    """Camera-to-world 4x4 on a sphere looking at the origin (formula of load_blender.py:10-34)."""
    th, ph = theta_deg / 180.0 * np.pi, phi_deg / 180.0 * np.pi
    trans = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float32)
    rphi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                     [0, 0, 0, 1]], np.float32)
    rth = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                    [0, 0, 0, 1]], np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    return (flip @ (rth @ (rphi @ trans))).astype(np.float32)


CAMERA_ANGLE_X = 0.6911112070083618  # public nerf_synthetic lego value (any fixed value would do)


def intrinsics(H, W, camera_angle_x=CAMERA_ANGLE_X):
    """K = [[f,0,W/2],[0,f,H/2],[0,0,1]] with f = .5 W / tan(.5 angle) (train.py:74-79, load_blender.py:72-73)."""
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], np.float64)
    return K, float(focal)


def pinhole_rays_np(H, W, K, c2w, pix=None):
    """float32 pinhole rays, restating rays.py:30-49; `pix` = optional [n,2] (row, col) subset."""
    if pix is None:
        i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    else:
        j = pix[:, 0].astype(np.float32)
        i = pix[:, 1].astype(np.float32)
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., None, :] * c2w[:3, :3], -1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).astype(np.float32)
    return rays_o, rays_d


def train_batch(n_rays=4096, H=400, W=400, seed=1, n_poses=100):
    """C3 workload: rays from `n_poses` spherical poses at random pixels + random targets."""
    rng = np.random.RandomState(seed)
    K, _ = intrinsics(H, W)
    thetas = np.linspace(-180, 180, n_poses, endpoint=False)
    per = n_rays // n_poses + 1
    ro, rd = [], []
    for th in thetas:
        pix = np.stack([rng.randint(0, H, per), rng.randint(0, W, per)], -1)
        o, d = pinhole_rays_np(H, W, K, pose_spherical(th, -30.0, 4.0), pix)
        ro.append(o)
        rd.append(d)
    ro = np.concatenate(ro)[:n_rays]
    rd = np.concatenate(rd)[:n_rays]
    target = rng.rand(n_rays, 3).astype(np.float32)
    return np.ascontiguousarray(ro), np.ascontiguousarray(rd), target


def blender_rays(n, seed, near=2.0, far=6.0, with_viewdirs=True):
    """[n, 8|11] ray matrix in the layout render_rays unpacks (render.py:181-184)."""
    ro, rd, _ = train_batch(n, seed=seed)
    cols = [ro, rd, np.full((n, 1), near, np.float32), np.full((n, 1), far, np.float32)]
    if with_viewdirs:
        vd = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
        cols.append(vd.astype(np.float32))
    return np.ascontiguousarray(np.concatenate(cols, -1).astype(np.float32))


def ndc_like_rays(n, seed):
    """Rays with NDC-style bounds 0..1 and origins near the z=-1 plane (LLFF after ndc_rays)."""
    rng = np.random.RandomState(seed)
    ro = np.concatenate([rng.uniform(-1, 1, (n, 2)), -np.ones((n, 1))], -1).astype(np.float32)
    rd = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), 2 * np.ones((n, 1))], -1).astype(np.float32)
    vd = rng.normal(size=(n, 3)).astype(np.float32)
    vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    return np.ascontiguousarray(np.concatenate(
        [ro, rd, np.zeros((n, 1), np.float32), np.ones((n, 1), np.float32), vd.astype(np.float32)], -1))
