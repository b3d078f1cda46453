"""ORACLE -- test infrastructure, not product code.

numpy/ctypes front end of the CPU restatement in oracle/orc_sampling.c and oracle/orc_mlp.c, plus
the orchestration of render_rays / render (render.py:134-315) and the MLP backward (numpy) used to
check the HIP path's gradients.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module.  Parity with the reference is pinned by
tests/golden/{static,dnerf,static2,dnerf2}.npz (see tests/test_oracle_golden.py, tests/test_oracle_golden2.py).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_D = 16
f32p = C.POINTER(C.c_float)


class OrcNet(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("input_ch", C.c_int32), ("input_ch_views", C.c_int32),
                ("output_ch", C.c_int32), ("use_viewdirs", C.c_int32), ("skips_mask", C.c_int32),
                ("pad_", C.c_int32),
                ("pts_w", f32p * MAX_D), ("pts_b", f32p * MAX_D),
                ("views_w", f32p), ("views_b", f32p), ("feature_w", f32p), ("feature_b", f32p),
                ("alpha_w", f32p), ("alpha_b", f32p), ("rgb_w", f32p), ("rgb_b", f32p),
                ("out_w", f32p), ("out_b", f32p)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liborc.so missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.orc_aten_row_sum.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(f32p) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ simple units
def linspace(start, end, n):
    out = np.empty(n, np.float32)
    lib().orc_linspace(C.c_float(start), C.c_float(end), n, _p(out))
    return out


def aten_row_sum(x):
    x = _f(x)
    return np.float32(lib().orc_aten_row_sum(_p(x), x.size))


def embed(x, num_freqs, max_freq_log2=None, log_sampling=True, include_input=True):
    """positional_encoding.py:5-36; x [..., dims]."""
    x = _f(x)
    dims = x.shape[-1]
    if max_freq_log2 is None:
        max_freq_log2 = num_freqs - 1
    od = dims * ((1 if include_input else 0) + 2 * num_freqs)
    flat = x.reshape(-1, dims)
    out = np.empty((flat.shape[0], od), np.float32)
    lib().orc_embed(_p(flat), C.c_int64(flat.shape[0]), dims, num_freqs, C.c_float(max_freq_log2),
                    int(log_sampling), int(include_input), _p(out))
    return out.reshape(x.shape[:-1] + (od,))


def sample_coarse(rays, S, lindisp=False, want_pts=True):
    rays = _f(rays)
    N, ncol = rays.shape
    z = np.empty((N, S), np.float32)
    pts = np.empty((N, S, 3), np.float32) if want_pts else None
    lib().orc_sample_coarse(_p(rays), C.c_int64(N), ncol, S, int(lindisp), _p(z), _p(pts))
    return z, pts


def points(rays, z):
    rays, z = _f(rays), _f(z)
    N, S = z.shape
    pts = np.empty((N, S, 3), np.float32)
    lib().orc_points(_p(rays), C.c_int64(N), rays.shape[1], S, _p(z), _p(pts))
    return pts


def sample_pdf(bins, weights, n_samples, u=None):
    bins, weights = _f(bins), _f(weights)
    N, nb = bins.shape
    assert weights.shape == (N, nb - 1)
    u = _f(u) if u is not None else None
    out = np.empty((N, n_samples), np.float32)
    lib().orc_sample_pdf(_p(bins), _p(weights), C.c_int64(N), nb, n_samples, _p(u), _p(out))
    return out


def sample_fine(z_vals, weights, n_importance, u=None):
    """render.py:219-224, :238 -> (z_samples, z_merged, z_std)."""
    z_vals, weights = _f(z_vals), _f(weights)
    N, S = z_vals.shape
    u = _f(u) if u is not None else None
    zs = np.empty((N, n_importance), np.float32)
    zm = np.empty((N, S + n_importance), np.float32)
    sd = np.empty((N,), np.float32)
    lib().orc_sample_fine(_p(z_vals), _p(weights), C.c_int64(N), S, n_importance, _p(u), _p(zs), _p(zm), _p(sd))
    return zs, zm, sd


def raw2outputs(raw, z_vals, rays_d, white_bkgd=False, noise=None):
    raw, z_vals, rays_d = _f(raw), _f(z_vals), _f(rays_d)
    N, S = z_vals.shape
    rgb = np.empty((N, 3), np.float32)
    disp, acc, depth = (np.empty((N,), np.float32) for _ in range(3))
    w = np.empty((N, S), np.float32)
    noise = _f(noise) if noise is not None else None
    lib().orc_raw2outputs(_p(raw), _p(z_vals), _p(rays_d), rays_d.shape[1], C.c_int64(N), S, _p(noise),
                          int(white_bkgd), _p(rgb), _p(disp), _p(acc), _p(w), _p(depth))
    return rgb, disp, acc, w, depth


def raw2outputs_bwd(raw, z_vals, rays_d, g_rgb, white_bkgd=False):
    raw, z_vals, rays_d, g_rgb = _f(raw), _f(z_vals), _f(rays_d), _f(g_rgb)
    N, S = z_vals.shape
    g_raw = np.empty((N, S, 4), np.float32)
    lib().orc_raw2outputs_bwd(_p(raw), _p(z_vals), _p(rays_d), rays_d.shape[1], C.c_int64(N), S, int(white_bkgd),
                              _p(g_rgb), _p(g_raw))
    return g_raw


def get_rays(H, W, K, c2w, row0=0, nrows=None):
    nrows = H - row0 if nrows is None else nrows
    c2w = _f(np.asarray(c2w)[:3, :4])
    ro = np.empty((nrows, W, 3), np.float32)
    rd = np.empty((nrows, W, 3), np.float32)
    lib().orc_get_rays(H, W, C.c_float(K[0][0]), C.c_float(K[1][1]), C.c_float(K[0][2]), C.c_float(K[1][2]),
                       _p(c2w), row0, nrows, _p(ro), _p(rd))
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    ro, rd = _f(rays_o).copy(), _f(rays_d).copy()
    lib().orc_ndc_rays(H, W, C.c_float(focal), C.c_float(near), C.c_int64(ro.size // 3), _p(ro), _p(rd))
    return ro, rd


# ------------------------------------------------------------------ networks
class Net:
    """Holds a numpy state_dict (reference key names) and the ctypes view of it."""

    def __init__(self, sd, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,),
                 use_viewdirs=True, prefix=""):
        self.sd = {k: _f(v) for k, v in sd.items()}
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.output_ch, self.skips, self.use_viewdirs, self.prefix = output_ch, tuple(skips), use_viewdirs, prefix
        n = OrcNet()
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch = D, W, input_ch, input_ch_views, output_ch
        n.use_viewdirs = int(use_viewdirs)
        n.skips_mask = sum(1 << s for s in skips)
        g = lambda k: _p(self.sd[prefix + k])  # noqa: E731
        for i in range(D):
            n.pts_w[i] = g(f"pts_linears.{i}.weight")
            n.pts_b[i] = g(f"pts_linears.{i}.bias")
        if use_viewdirs:
            n.views_w, n.views_b = g("views_linears.0.weight"), g("views_linears.0.bias")
            n.feature_w, n.feature_b = g("feature_linear.weight"), g("feature_linear.bias")
            n.alpha_w, n.alpha_b = g("alpha_linear.weight"), g("alpha_linear.bias")
            n.rgb_w, n.rgb_b = g("rgb_linear.weight"), g("rgb_linear.bias")
        else:
            n.out_w, n.out_b = g("output_linear.weight"), g("output_linear.bias")
        self.c = n

    def forward(self, x):
        """NeRF.forward (network.py:46-69) on embedded inputs x [P, input_ch+input_ch_views]."""
        x = _f(x)
        P = x.shape[0]
        oc = 4 if self.use_viewdirs else self.output_ch
        out = np.empty((P, oc), np.float32)
        lib().orc_nerf_forward(C.byref(self.c), _p(x), C.c_int64(P), _p(out))
        return out

    # numpy forward with saved activations + backward: the gradient oracle (autograd of network.py:46-69)
    def forward_backward(self, x, g_out, masks=None):
        """masks: optional {"trunk": [D x bool[P,W]], "hv": bool[P,W/2]} ReLU gates to use instead of the
        oracle's own (a pre-activation within rounding of 0 can gate differently in f32 and f64; a gradient
        check should not depend on that coin flip)."""
        sd, pre = self.sd, self.prefix
        x = _f(x).astype(np.float64)
        xp, xv = x[:, :self.input_ch], x[:, self.input_ch:]
        acts, h = [], xp
        for i in range(self.D):
            Wt, b = sd[f"{pre}pts_linears.{i}.weight"].astype(np.float64), sd[f"{pre}pts_linears.{i}.bias"]
            y = np.maximum(h @ Wt.T + b, 0)
            acts.append((h, y))
            h = np.concatenate([xp, y], -1) if i in self.skips else y
        grads = {}
        g_out = np.asarray(g_out, np.float64)
        assert self.use_viewdirs
        Wa, Wf = sd[pre + "alpha_linear.weight"].astype(np.float64), sd[pre + "feature_linear.weight"].astype(np.float64)
        Wv, Wr = sd[pre + "views_linears.0.weight"].astype(np.float64), sd[pre + "rgb_linear.weight"].astype(np.float64)
        feat = h @ Wf.T + sd[pre + "feature_linear.bias"]
        hv_in = np.concatenate([feat, xv], -1)
        hv = np.maximum(hv_in @ Wv.T + sd[pre + "views_linears.0.bias"], 0)
        out = np.concatenate([hv @ Wr.T + sd[pre + "rgb_linear.bias"], h @ Wa.T + sd[pre + "alpha_linear.bias"]], -1)
        g_rgb, g_alpha = g_out[:, :3], g_out[:, 3:4]
        grads[pre + "rgb_linear.weight"], grads[pre + "rgb_linear.bias"] = g_rgb.T @ hv, g_rgb.sum(0)
        g_hv = (g_rgb @ Wr) * (masks["hv"] if masks is not None else (hv > 0))
        grads[pre + "views_linears.0.weight"], grads[pre + "views_linears.0.bias"] = g_hv.T @ hv_in, g_hv.sum(0)
        g_feat = (g_hv @ Wv)[:, :self.W]
        grads[pre + "feature_linear.weight"], grads[pre + "feature_linear.bias"] = g_feat.T @ h, g_feat.sum(0)
        grads[pre + "alpha_linear.weight"], grads[pre + "alpha_linear.bias"] = g_alpha.T @ h, g_alpha.sum(0)
        g_h = g_feat @ Wf + g_alpha @ Wa
        for i in reversed(range(self.D)):
            hin, y = acts[i]
            if i in self.skips:
                g_h = g_h[:, self.input_ch:]
            g_y = g_h * (masks["trunk"][i] if masks is not None else (y > 0))
            grads[f"{pre}pts_linears.{i}.weight"], grads[f"{pre}pts_linears.{i}.bias"] = g_y.T @ hin, g_y.sum(0)
            g_h = g_y @ sd[f"{pre}pts_linears.{i}.weight"].astype(np.float64)
        return out.astype(np.float32), grads


class TimeNet:
    """DNeRF_Deformation._time/_time_out (vanilla_dynamic/network.py:254-305)."""

    def __init__(self, sd, D=8, W=256, input_ch=63, input_ch_time=21, skips=(4,)):
        self.sd = {k: _f(v) for k, v in sd.items()}
        n = OrcNet()
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch, n.use_viewdirs = D, W, input_ch, input_ch_time, 3, 0
        n.skips_mask = sum(1 << s for s in skips)
        for i in range(D):
            n.pts_w[i] = _p(self.sd[f"_time.{i}.weight"])
            n.pts_b[i] = _p(self.sd[f"_time.{i}.bias"])
        n.out_w, n.out_b = _p(self.sd["_time_out.weight"]), _p(self.sd["_time_out.bias"])
        self.c = n

    def forward(self, xe, te):
        xe, te = _f(xe), _f(te)
        P = xe.shape[0]
        dx = np.empty((P, 3), np.float32)
        lib().orc_time_forward(C.byref(self.c), _p(xe), _p(te), C.c_int64(P), _p(dx))
        return dx


def run_network(pts, viewdirs, net, multires=10, multires_views=4):
    """render.py:31-56 (netchunk does not affect results)."""
    N, S, _ = pts.shape
    e = embed(pts.reshape(-1, 3), multires)
    if viewdirs is not None:
        d = np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3)
        e = np.concatenate([e, embed(d, multires_views)], -1)
    return net.forward(e).reshape(N, S, -1)


def run_network_dnerf(pts, viewdirs, times, occ, timenet, netchunk=65536, zero_canonical=True,
                      multires=10, multires_views=4, multires_time=10):
    """vanilla_dynamic/render.py:49-124 + network.py:307-341 (deformation), including the per-netchunk
    `t_raw[0,0] == 0` quirk that skips the time net for a whole chunk."""
    N, S, _ = pts.shape
    flat = _f(pts.reshape(-1, 3))
    t_flat = _f(np.broadcast_to(times[:, None, :], (N, S, 1)).reshape(-1, 1))
    ed = embed(np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3), multires_views)
    outs, dxs = [], []
    for i in range(0, flat.shape[0], netchunk):
        p, t = flat[i:i + netchunk], t_flat[i:i + netchunk]
        if float(t[0, 0]) == 0.0 and zero_canonical:
            dx = np.zeros_like(p)
            xe = embed(p, multires)
        else:
            dx = timenet.forward(embed(p, multires), embed(t, multires_time))
            xe = embed(p + dx, multires)
        outs.append(occ.forward(np.concatenate([xe, ed[i:i + netchunk]], -1)))
        dxs.append(dx)
    return np.concatenate(outs).reshape(N, S, 4), np.concatenate(dxs).reshape(N, S, 3)


def render_rays(rays, net, net_fine, N_samples=64, N_importance=0, lindisp=False, white_bkgd=False,
                retraw=False, query=None):
    """render.py:134-244 with perturb=0, raw_noise_std=0."""
    rays = _f(rays)
    viewdirs = rays[:, -3:] if rays.shape[-1] > 8 else None
    rays_d = rays[:, 3:6]
    query = query or (lambda p, v, n: run_network(p, v, n))
    z, pts = sample_coarse(rays, N_samples, lindisp)
    raw = query(pts, viewdirs, net)
    rgb, disp, acc, w, depth = raw2outputs(raw, z, rays_d, white_bkgd)
    ret = {}
    if N_importance > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc, z_vals0=z, weights0=w)
        zs, z, zstd = sample_fine(z, w, N_importance)
        pts = points(rays, z)
        raw = query(pts, viewdirs, net if net_fine is None else net_fine)
        rgb, disp, acc, w, depth = raw2outputs(raw, z, rays_d, white_bkgd)
        ret.update(z_std=zstd, z_samples=zs)
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, z_vals=z, weights=w, depth_map=depth)
    if retraw:
        ret["raw"] = raw
    return ret


def build_rays(H, W, K, c2w=None, rays=None, ndc=False, near=0., far=1., use_viewdirs=True, c2w_staticcam=None):
    """render.py:276-304 -> [N, 8|11] ray matrix."""
    if c2w is not None:
        ro, rd = get_rays(H, W, K, c2w)
    else:
        ro, rd = rays
    ro, rd = _f(ro).reshape(-1, 3), _f(rd).reshape(-1, 3)
    cols = []
    if use_viewdirs:
        vd = rd
        if c2w_staticcam is not None:
            ro, rd = get_rays(H, W, K, c2w_staticcam)
            ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        nrm = np.sqrt((vd * vd).sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
        vd = (vd / nrm).astype(np.float32)
    if ndc:
        ro, rd = ndc_rays(H, W, K[0][0], 1.0, ro, rd)
    n = ro.shape[0]
    cols = [ro, rd, np.full((n, 1), near, np.float32), np.full((n, 1), far, np.float32)]
    if use_viewdirs:
        cols.append(vd)
    return np.ascontiguousarray(np.concatenate(cols, -1), dtype=np.float32)


# ------------------------------------------------------------------ vanilla_dynamic variants (numpy, f64 internally)
def _layernorm(h, w, b, eps=1e-5):
    """nn.LayerNorm over the last axis (biased variance), as vanilla_dynamic/network.py:46-48 applies it."""
    mu = h.mean(-1, keepdims=True)
    var = ((h - mu) ** 2).mean(-1, keepdims=True)
    return (h - mu) / np.sqrt(var + eps) * w + b


def _trunk(sd, pre, names, x_in, x_skip, D, skips, ln_names=None):
    """D x (Linear [+ LayerNorm] + ReLU) with cat[x_skip, h] after the layers in `skips` (network.py:73-79)."""
    h = x_in
    for i in range(D):
        h = h @ sd[f"{pre}{names}.{i}.weight"].astype(np.float64).T + sd[f"{pre}{names}.{i}.bias"]
        if ln_names is not None:
            h = _layernorm(h, sd[f"{pre}{ln_names}.{i}.weight"], sd[f"{pre}{ln_names}.{i}.bias"])
        h = np.maximum(h, 0)
        if i in skips:
            h = np.concatenate([x_skip, h], -1)
    return h


def _views_head(sd, pre, h, xv, layernorm):
    """alpha / feature / view layer / rgb (network.py:81-93)."""
    alpha = h @ sd[pre + "alpha_linear.weight"].astype(np.float64).T + sd[pre + "alpha_linear.bias"]
    feat = h @ sd[pre + "feature_linear.weight"].astype(np.float64).T + sd[pre + "feature_linear.bias"]
    hv = np.concatenate([feat, xv], -1) @ sd[pre + "views_linears.0.weight"].astype(np.float64).T \
        + sd[pre + "views_linears.0.bias"]
    if layernorm:
        hv = _layernorm(hv, sd[pre + "views_layernorm.weight"], sd[pre + "views_layernorm.bias"])
    hv = np.maximum(hv, 0)
    rgb = hv @ sd[pre + "rgb_linear.weight"].astype(np.float64).T + sd[pre + "rgb_linear.bias"]
    return np.concatenate([rgb, alpha], -1)


def nerf_forward_np(sd, x, D=8, input_ch=63, skips=(4,), prefix="", layernorm=False):
    """vanilla_dynamic NeRF.forward (network.py:60-97), optional LayerNorm, on embedded rows x [P, input_ch + views]."""
    x = np.asarray(x, np.float64)
    xp, xv = x[:, :input_ch], x[:, input_ch:]
    h = _trunk(sd, prefix, "pts_linears", xp, xp, D, skips, "pts_layernorms" if layernorm else None)
    return _views_head(sd, prefix, h, xv, layernorm).astype(np.float32)


def deformation_forward_np(sd, pts, dirs, times, D=8, D_deform=8, skips=(4,), layernorm=False, multires=10,
                           multires_views=4, multires_time=10):
    """DNeRF_Deformation.forward for t != 0 (network.py:307-341) on raw points / directions / times [P,1]
    -> (out [P,4], dx [P,3])."""
    xe = embed(pts, multires).astype(np.float64)
    te = embed(times, multires_time).astype(np.float64)
    h = _trunk(sd, "", "_time", np.concatenate([xe, te], -1), xe, D_deform, skips,
               "_time_layernorms" if layernorm else None)
    dx = (h @ sd["_time_out.weight"].astype(np.float64).T + sd["_time_out.bias"]).astype(np.float32)
    x2 = embed(_f(pts) + dx, multires)
    xin = np.concatenate([x2, embed(dirs, multires_views)], -1)
    return nerf_forward_np(sd, xin, D, xe.shape[1], skips, "_occ.", layernorm), dx


def straightforward_forward_np(sd, pts, dirs, times, D=8, skips=(4,), multires=10, multires_views=4, multires_time=10):
    """DNeRF_Straightforward.forward (network.py:151-196): trunk input and skip are cat[gamma(x), gamma(t)]."""
    xt = np.concatenate([embed(pts, multires), embed(times, multires_time)], -1).astype(np.float64)
    h = _trunk(sd, "", "pts_linears", xt, xt, D, skips)
    return _views_head(sd, "", h, embed(dirs, multires_views).astype(np.float64), False).astype(np.float32)
