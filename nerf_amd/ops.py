"""Tensor-level wrappers over the C ABI: torch is used for device memory and streams only.

Every function takes contiguous float32 CUDA(HIP) tensors, allocates its outputs with torch.empty and
enqueues one or more HIP kernels on torch's current stream.  No CPU fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import NrfEmbed, NrfNet, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: nerf_amd kernels need a GPU tensor (got device {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---------------------------------------------------------------- rays
def _pose_rows(c2w):
    """[>=3, >=4] pose tensor -> (tensor kept alive, row stride) with unit column stride."""
    c2w = _f32(c2w, "c2w") if c2w.stride(-1) != 1 or c2w.dtype != torch.float32 or not c2w.is_cuda else c2w
    return c2w, int(c2w.stride(0))


def gen_rays(H, W, K, c2w, near, far, use_viewdirs, ndc, c2w_staticcam=None, row0=0, nrows=None, ndc_near=1.0):
    """The [N, 8|11] ray matrix render() feeds to render_rays, for image rows row0 .. row0+nrows-1 (one kernel)."""
    nrows = H - row0 if nrows is None else nrows
    c2w, stride = _pose_rows(c2w)
    cs, cs_stride = (None, 4)
    if c2w_staticcam is not None and use_viewdirs:
        cs, cs_stride = _pose_rows(c2w_staticcam)
    rays = torch.empty((nrows * W, 11 if use_viewdirs else 8), device=c2w.device, dtype=torch.float32)
    check(_lib.lib().nrf_gen_rays(int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]),
                                  _ptr(c2w), stride, _ptr(cs), cs_stride, int(row0), int(nrows), float(near), float(far),
                                  int(bool(use_viewdirs)), int(bool(ndc)), float(K[0][0]), float(ndc_near), _ptr(rays),
                                  _stream()), "nrf_gen_rays")
    return rays


def gather_ray_batch(rows, images, poses, H, W, K, img_index=None, f64=True):
    """Rows of train.py:147-153's UNSHUFFLED rays_rgb table -> (batch_rays [2, n, 3], target_s [n, 3]), computed from
    (images [M, H, W, 3], poses [M, >=3, >=4]) on the fly.  rows: int64 device tensor; img_index: int32 device tensor
    mapping table slots to images (i_train)."""
    images = _f32(images, "images")
    poses = _f32(poses, "poses")
    if rows.dtype != torch.int64 or not rows.is_cuda:
        raise RuntimeError("rows: expected an int64 GPU tensor")
    rows = rows.contiguous()
    if images.dim() != 4 or images.shape[1] != H or images.shape[2] != W or images.shape[3] != 3:
        raise ValueError(f"images: expected [M, {H}, {W}, 3], got {tuple(images.shape)}")
    if poses.dim() != 3 or poses.shape[1] < 3 or poses.shape[2] < 4 or poses.shape[0] != images.shape[0]:
        raise ValueError(f"poses: expected [{images.shape[0]}, >=3, >=4], got {tuple(poses.shape)}")
    n_slots = images.shape[0]
    if img_index is not None:
        if img_index.dtype != torch.int32 or not img_index.is_cuda:
            raise RuntimeError("img_index: expected an int32 GPU tensor")
        img_index = img_index.contiguous()
        n_slots = img_index.numel()
    n = rows.numel()
    rays = torch.empty((2, n, 3), device=images.device, dtype=torch.float32)
    rgb = torch.empty((n, 3), device=images.device, dtype=torch.float32)
    check(_lib.lib().nrf_gather_ray_batch(_ptr(rows), n, _ptr(img_index), int(n_slots), _ptr(poses), int(poses.stride(1)),
                                          int(poses.stride(0)), _ptr(images), int(images.shape[0]), int(H), int(W),
                                          float(K[0][0]),
                                          float(K[1][1]), float(K[0][2]), float(K[1][2]), int(bool(f64)), _ptr(rays),
                                          _ptr(rgb), _stream()), "nrf_gather_ray_batch")
    return rays, rgb


# ---------------------------------------------------------------- sampling
def sample_coarse(rays, n_samples, lindisp=False, t_rand=None):
    """render.py:186-208 -> z_vals [N, S]."""
    rays = _f32(rays, "rays")
    N, ncol = rays.shape
    z = torch.empty((N, n_samples), device=rays.device, dtype=torch.float32)
    if t_rand is not None:
        t_rand = _f32(t_rand, "t_rand")
        assert t_rand.shape == (N, n_samples)
    check(_lib.lib().nrf_sample_coarse(_ptr(rays), N, ncol, n_samples, int(bool(lindisp)), _ptr(t_rand), _ptr(z),
                                       _stream()), "nrf_sample_coarse")
    return z


def points(rays, z_vals):
    """render.py:210 / :224 -> pts [N, S, 3]."""
    rays, z_vals = _f32(rays, "rays"), _f32(z_vals, "z_vals")
    N, S = z_vals.shape
    pts = torch.empty((N, S, 3), device=rays.device, dtype=torch.float32)
    check(_lib.lib().nrf_points(_ptr(rays), N, rays.shape[1], _ptr(z_vals), S, _ptr(pts), _stream()), "nrf_points")
    return pts


def sample_fine(z_vals, weights, n_importance, u=None, want_samples=True):
    """render.py:219-224, :238 -> (z_samples, z_merged, z_std)."""
    z_vals, weights = _f32(z_vals, "z_vals"), _f32(weights, "weights")
    N, S = z_vals.shape
    assert weights.shape == (N, S)
    dev = z_vals.device
    zs = torch.empty((N, n_importance), device=dev, dtype=torch.float32) if want_samples else None
    zm = torch.empty((N, S + n_importance), device=dev, dtype=torch.float32)
    sd = torch.empty((N,), device=dev, dtype=torch.float32)
    if u is not None:
        u = _f32(u, "u")
        assert u.shape == (N, n_importance)
    check(_lib.lib().nrf_sample_fine(_ptr(z_vals), _ptr(weights), N, S, n_importance, _ptr(u), _ptr(zs), _ptr(zm),
                                     _ptr(sd), _stream()), "nrf_sample_fine")
    return zs, zm, sd


def sample_pdf(bins, weights, n_samples, u=None):
    """sampling.py:5-60."""
    bins, weights = _f32(bins, "bins"), _f32(weights, "weights")
    lead = bins.shape[:-1]
    nb = bins.shape[-1]
    b2, w2 = bins.reshape(-1, nb), weights.reshape(-1, nb - 1)
    out = torch.empty((b2.shape[0], n_samples), device=bins.device, dtype=torch.float32)
    if u is not None:
        u = _f32(u, "u").reshape(-1, n_samples)
    check(_lib.lib().nrf_sample_pdf(_ptr(b2), _ptr(w2), b2.shape[0], nb, n_samples, _ptr(u), _ptr(out), _stream()),
          "nrf_sample_pdf")
    return out.reshape(*lead, n_samples)


# ---------------------------------------------------------------- positional encoding
def pe_fwd(x, num_freqs, max_freq_log2, log_sampling=True, include_input=True):
    x = _f32(x, "x")
    dims = x.shape[-1]
    flat = x.reshape(-1, dims)
    od = dims * ((1 if include_input else 0) + 2 * num_freqs)
    out = torch.empty((flat.shape[0], od), device=x.device, dtype=torch.float32)
    check(_lib.lib().nrf_pe_fwd(_ptr(flat), flat.shape[0], dims, num_freqs, float(max_freq_log2), int(log_sampling),
                                int(include_input), _ptr(out), _stream()), "nrf_pe_fwd")
    return out.reshape(*x.shape[:-1], od)


def pe_bwd(x, g_out, num_freqs, max_freq_log2, log_sampling=True, include_input=True):
    """d(loss)/dx from d(loss)/d(gamma(x))."""
    x, g_out = _f32(x, "x"), _f32(g_out, "g_out")
    dims = x.shape[-1]
    flat = x.reshape(-1, dims)
    od = dims * ((1 if include_input else 0) + 2 * num_freqs)
    g2 = g_out.reshape(-1, od)
    assert g2.shape[0] == flat.shape[0]
    g_x = torch.empty_like(flat)
    check(_lib.lib().nrf_pe_bwd(_ptr(flat), _ptr(g2), flat.shape[0], dims, num_freqs, float(max_freq_log2),
                                int(log_sampling), int(include_input), _ptr(g_x), _stream()), "nrf_pe_bwd")
    return g_x.reshape(x.shape)


# ---------------------------------------------------------------- compositing
def _dirs(rays_d):
    """rays_d may be a strided view into the [N, ncol] ray matrix; pass (base pointer, row stride)."""
    if rays_d.dtype == torch.float32 and rays_d.is_cuda and rays_d.dim() == 2 and rays_d.shape[1] == 3 \
            and rays_d.stride(1) == 1 and rays_d.stride(0) >= 3:
        return rays_d, rays_d.stride(0)
    d = _f32(rays_d, "rays_d")
    return d, 3


def composite_fwd(raw, z_vals, rays_d, noise=None, white_bkgd=False):
    raw, z_vals = _f32(raw, "raw"), _f32(z_vals, "z_vals")
    N, S = z_vals.shape
    if raw.shape != (N, S, 4):
        raise ValueError(f"composite_fwd: raw must be [N,S,4] = {(N, S, 4)}, got {tuple(raw.shape)} "
                         "(raw2outputs slices wider network outputs, render.py:100-103)")
    d, ds = _dirs(rays_d)
    assert d.shape[0] == N
    dev = raw.device
    rgb = torch.empty((N, 3), device=dev, dtype=torch.float32)
    disp, acc, depth = (torch.empty((N,), device=dev, dtype=torch.float32) for _ in range(3))
    w = torch.empty((N, S), device=dev, dtype=torch.float32)
    if noise is not None:
        noise = _f32(noise, "noise")
    check(_lib.lib().nrf_composite_fwd(_ptr(raw), _ptr(z_vals), _ptr(d), ds, N, S, _ptr(noise), int(bool(white_bkgd)),
                                       _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(w), _ptr(depth), _stream()),
          "nrf_composite_fwd")
    return rgb, disp, acc, w, depth


def composite_bwd(raw, z_vals, rays_d, g_rgb, g_acc=None, g_depth=None, g_weights=None, noise=None,
                  white_bkgd=False):
    raw, z_vals, g_rgb = _f32(raw, "raw"), _f32(z_vals, "z_vals"), _f32(g_rgb, "g_rgb")
    N, S = z_vals.shape
    d, ds = _dirs(rays_d)
    g_raw = torch.empty((N, S, 4), device=raw.device, dtype=torch.float32)
    opt = [None if t is None else _f32(t, "grad") for t in (g_acc, g_depth, g_weights, noise)]
    check(_lib.lib().nrf_composite_bwd(_ptr(raw), _ptr(z_vals), _ptr(d), ds, N, S, _ptr(opt[3]),
                                       int(bool(white_bkgd)), _ptr(g_rgb), _ptr(opt[0]), _ptr(opt[1]), _ptr(opt[2]),
                                       _ptr(g_raw), _stream()), "nrf_composite_bwd")
    return g_raw


# ---------------------------------------------------------------- network query
class PackedNet:
    """MFMA-ordered copy of a NeRF module's nn.Linear parameters (network.py:29-44).

    The nn.Linear parameters stay the source of truth (checkpoints, Adam); `refresh()` repacks when
    any parameter's version counter moved (after optimizer.step() or load_state_dict())."""

    def __init__(self, params, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, mode=0,
                 multires=0, multires_views=0, multires_time=0):
        self.params = params  # dict name -> tensor with reference state_dict key names
        self.net = NrfNet()
        n = self.net
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch = D, W, input_ch, input_ch_views, output_ch
        n.use_viewdirs = int(bool(use_viewdirs))
        n.skips_mask = sum(1 << int(s) for s in skips)
        n.use_layernorm = int(any("layernorm" in k for k in params))
        self.emb = NrfEmbed(mode, multires, multires_views, multires_time)
        self.n_floats = _lib.lib().nrf_mlp_packed_floats(C.byref(n), C.byref(self.emb))
        if self.n_floats <= 0:
            raise RuntimeError("nrf_mlp_packed_floats: " + _lib.lib().nrf_last_error_string().decode())
        self.packed = None
        self._versions = None
        self.out_ch = 4 if use_viewdirs else output_ch

    def _fill_pointers(self):
        n, g = self.net, lambda k: C.c_void_p(self._p[k].data_ptr())  # noqa: E731
        self._p = {k: _f32(v.detach(), k) for k, v in self.params.items()}
        for i in range(n.D):
            n.pts_w[i] = g(f"pts_linears.{i}.weight")
            n.pts_b[i] = g(f"pts_linears.{i}.bias")
        if n.use_viewdirs:
            n.views_w, n.views_b = g("views_linears.0.weight"), g("views_linears.0.bias")
            n.feature_w, n.feature_b = g("feature_linear.weight"), g("feature_linear.bias")
            n.alpha_w, n.alpha_b = g("alpha_linear.weight"), g("alpha_linear.bias")
            n.rgb_w, n.rgb_b = g("rgb_linear.weight"), g("rgb_linear.bias")
        else:
            n.out_w, n.out_b = g("output_linear.weight"), g("output_linear.bias")
        if n.use_layernorm:       # vanilla_dynamic/network.py:46-48
            for i in range(n.D):
                n.ln_w[i], n.ln_b[i] = g(f"pts_layernorms.{i}.weight"), g(f"pts_layernorms.{i}.bias")
            if n.use_viewdirs:
                n.views_ln_w, n.views_ln_b = g("views_layernorm.weight"), g("views_layernorm.bias")

    def refresh(self):
        vers = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if self.packed is not None and vers == self._versions:
            return
        self._fill_pointers()
        dev = next(iter(self._p.values())).device
        if self.packed is None or self.packed.device != dev:
            self.packed = torch.empty((self.n_floats,), device=dev, dtype=torch.float32)
        check(_lib.lib().nrf_mlp_pack_weights(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _stream()),
              "nrf_mlp_pack_weights")
        self._versions = vers

    def query(self, pts, viewdirs=None, pts_per_ray=1):
        """mode 0: pts [P,3] (+ viewdirs [P/pts_per_ray,3]); mode 1: pts = embedded rows [P, in+views]."""
        self.refresh()
        pts = _f32(pts, "pts")
        P = pts.shape[0]
        raw = torch.empty((P, self.out_ch), device=pts.device, dtype=torch.float32)
        if viewdirs is not None:
            viewdirs = _f32(viewdirs, "viewdirs")
        check(_lib.lib().nrf_query_fwd(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                       _ptr(viewdirs), P, int(pts_per_ray), _ptr(raw), _stream()), "nrf_query_fwd")
        return raw

    def query_train(self, pts, viewdirs=None, pts_per_ray=1):
        """Forward that also saves inputs/activations for query_bwd -> (raw, saved workspace)."""
        self.refresh()
        pts = _f32(pts, "pts")
        P = pts.shape[0]
        raw = torch.empty((P, self.out_ch), device=pts.device, dtype=torch.float32)
        n = _lib.lib().nrf_train_acts_floats(C.byref(self.net), P)
        acts = torch.empty((max(n, 1),), device=pts.device, dtype=torch.float32)
        if viewdirs is not None:
            viewdirs = _f32(viewdirs, "viewdirs")
        check(_lib.lib().nrf_query_fwd_train(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                             _ptr(viewdirs), P, int(pts_per_ray), _ptr(raw), _ptr(acts), _stream()),
              "nrf_query_fwd_train")
        return raw, acts

    def query_bwd(self, g_raw, n_points, acts, pts=None, x3=False, a16=False):
        """-> dict name -> gradient tensor (same shapes as the parameters); with `pts` (the query points of the
        forward) -> (that dict, d(loss)/d(pts) [P,3]).  x3: the weight-gradient GEMMs on the bf16x3 arithmetic.
        a16: `acts` holds f16 activations (PackedNetBf16x3.query_train(a16=True)); needs x3=True, parameters only.
        a16 = "w16": the gradient blocks are saved as (scaled) f16 too and the weight-gradient GEMMs run on the f16 MFMA."""
        g_raw = _f32(g_raw, "g_raw")
        dev = g_raw.device
        lib = _lib.lib()
        gbuf = torch.empty((max(lib.nrf_train_grads_floats(C.byref(self.net), n_points), 1),), device=dev)
        slab = torch.empty((max(lib.nrf_train_slab_floats(C.byref(self.net), n_points), 1),), device=dev)
        grads = {k: torch.empty_like(v, dtype=torch.float32).contiguous() for k, v in self.params.items()}
        gn = NrfNet()
        g = lambda k: C.c_void_p(grads[k].data_ptr())  # noqa: E731
        n = self.net
        for i in range(n.D):
            gn.pts_w[i], gn.pts_b[i] = g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias")
        if n.use_viewdirs:
            gn.views_w, gn.views_b = g("views_linears.0.weight"), g("views_linears.0.bias")
            gn.feature_w, gn.feature_b = g("feature_linear.weight"), g("feature_linear.bias")
            gn.alpha_w, gn.alpha_b = g("alpha_linear.weight"), g("alpha_linear.bias")
            gn.rgb_w, gn.rgb_b = g("rgb_linear.weight"), g("rgb_linear.bias")
        else:
            gn.out_w, gn.out_b = g("output_linear.weight"), g("output_linear.bias")
        if n.use_layernorm:      # the affine parameters' gradients come out of the same pass (ln_param_grad_kernel)
            if x3:
                raise RuntimeError("nerf_amd: LayerNorm nets train on the f32 kernels (train_precision 'fp32')")
            for i in range(n.D):
                gn.ln_w[i], gn.ln_b[i] = g(f"pts_layernorms.{i}.weight"), g(f"pts_layernorms.{i}.bias")
            if n.use_viewdirs:
                gn.views_ln_w, gn.views_ln_b = g("views_layernorm.weight"), g("views_layernorm.bias")
        if x3 and self._bwd_fn() is not lib.nrf_query_bwd:      # deformation net: weight-gradient GEMMs on the bf16 pipe
            check(lib.nrf_dnerf_time_bwd_bf16x3(C.byref(self.net), C.byref(self.emb), _ptr(self.packed),
                                                _ptr(self.bwd_x3_stream() if x3 != "wgrad" else None), _ptr(g_raw),
                                                _ptr(acts), _ptr(gbuf), _ptr(slab), n_points, C.byref(gn), _stream()),
                  "nrf_dnerf_time_bwd_bf16x3")
            return grads
        if a16:
            if not x3 or x3 == "wgrad" or pts is not None or self.bwd_x3_stream() is None:
                raise RuntimeError("nerf_amd: the f16 activation workspace belongs to the bf16x3 training kernels of static "
                                   "nets (W = 128, 256; parameter gradients only)")
            fn = "nrf_query_bwd_bf16x3_w16" if a16 == "w16" else "nrf_query_bwd_bf16x3_a16"
            check(getattr(lib, fn)(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(self.bwd_x3_stream()),
                                   _ptr(g_raw), _ptr(acts), _ptr(gbuf), _ptr(slab), n_points, C.byref(gn), _stream()), fn)
            return grads
        if x3:
            g_pts = None
            if pts is not None:
                pts = _f32(pts, "pts")
                g_pts = torch.empty((n_points, 3), device=dev, dtype=torch.float32)
            check(lib.nrf_query_bwd_bf16x3(C.byref(self.net), C.byref(self.emb), _ptr(self.packed),
                                           _ptr(self.bwd_x3_stream() if x3 != "wgrad" else None), _ptr(g_raw), _ptr(acts),
                                           _ptr(gbuf), _ptr(slab), n_points, C.byref(gn), _ptr(pts), _ptr(g_pts),
                                           _stream()), "nrf_query_bwd_bf16x3")
            return grads if pts is None else (grads, g_pts)
        if pts is not None:
            pts = _f32(pts, "pts")
            g_pts = torch.empty((n_points, 3), device=dev, dtype=torch.float32)
            check(lib.nrf_query_bwd_input(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(g_raw),
                                          _ptr(acts), _ptr(gbuf), _ptr(slab), n_points, C.byref(gn), _ptr(pts),
                                          _ptr(g_pts), _stream()), "nrf_query_bwd_input")
            return grads, g_pts
        check(self._bwd_fn()(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(g_raw), _ptr(acts),
                             _ptr(gbuf), _ptr(slab), n_points, C.byref(gn), _stream()), "nrf_query_bwd")
        return grads

    def _bwd_fn(self):
        return _lib.lib().nrf_query_bwd

    def bwd_x3_stream(self):
        """W^T stream of the bf16x3 activation-gradient chain, repacked with the f32 copy; None where the shape has no
        bf16x3 kernel (the library then runs that chain in f32)."""
        if self.emb.mode != 0 or self.net.W not in (128, 256) or self.net.D < 2:
            return None
        # the pack reads the LIVE parameters, the f32 copy and the saved activations belong to the last refresh(): a
        # parameter changed in between must not be packed under the old label
        live = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if live != self._versions:
            raise RuntimeError("nerf_amd: a parameter of this network was modified after its weights were packed for the "
                               "forward; the backward's W^T stream would not match the saved activations")
        if getattr(self, "_bwd_x3_versions", None) != self._versions or getattr(self, "_bwd_x3", None) is None:
            lib = _lib.lib()
            nb = lib.nrf_mlp_bf16x3_bwd_packed_bytes(C.byref(self.net), C.byref(self.emb))
            if nb <= 0:
                return None
            if getattr(self, "_bwd_x3", None) is None or self._bwd_x3.numel() != nb:
                self._bwd_x3 = torch.empty((nb,), device=self.packed.device, dtype=torch.uint8)
            check(lib.nrf_mlp_bf16x3_bwd_pack_weights(C.byref(self.net), C.byref(self.emb), _ptr(self._bwd_x3), _stream()),
                  "nrf_mlp_bf16x3_bwd_pack_weights")
            self._bwd_x3_versions = self._versions
        return self._bwd_x3


class PackedTimeNet(PackedNet):
    """The deformation ("time") net of DNeRF_Deformation (vanilla_dynamic/network.py:254-305) behind the
    nrf_dnerf_time_* entry points.  `params` uses PackedNet's key names: pts_linears.{i}.* = the `_time` layers,
    output_linear.* = `_time_out`; input_ch_views carries the TIME channel count."""

    def __init__(self, params, D, W, input_ch, input_ch_time, skips, multires, multires_time):
        self.params = params
        self.net = NrfNet()
        n = self.net
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch, n.use_viewdirs = D, W, input_ch, input_ch_time, 3, 0
        n.skips_mask = sum(1 << int(s) for s in skips)
        n.use_layernorm = int(any("layernorm" in k for k in params))     # `_time_layernorms` as pts_layernorms.{i}.*
        self.emb = NrfEmbed(0, multires, 0, multires_time)
        self.n_floats = _lib.lib().nrf_dnerf_time_packed_floats(C.byref(n), C.byref(self.emb))
        if self.n_floats <= 0:
            raise RuntimeError("nrf_dnerf_time_packed_floats: " + _lib.lib().nrf_last_error_string().decode())
        self.packed = None
        self._versions = None
        self.out_ch = 3

    def refresh(self):
        vers = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if self.packed is not None and vers == self._versions:
            return
        self._fill_pointers()
        dev = next(iter(self._p.values())).device
        if self.packed is None or self.packed.device != dev:
            self.packed = torch.empty((self.n_floats,), device=dev, dtype=torch.float32)
        check(_lib.lib().nrf_dnerf_pack_time_weights(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _stream()),
              "nrf_dnerf_pack_time_weights")
        self._versions = vers

    def query(self, *a, **k):
        raise NotImplementedError("the time net alone is only evaluated for training; rendering uses nrf_dnerf_query_fwd")

    def query_train(self, pts, times, pts_per_ray=1, times_per_point=False, acts=None):
        """-> (dx [P,3], saved workspace).  `acts`: a workspace of an earlier call to reuse when it is large enough."""
        self.refresh()
        pts, times = _f32(pts, "pts"), _f32(times, "times")
        P = pts.shape[0]
        dx = torch.empty((P, 3), device=pts.device, dtype=torch.float32)
        n = _lib.lib().nrf_train_acts_floats(C.byref(self.net), P)
        if acts is None or acts.numel() < max(n, 1) or acts.device != pts.device:
            acts = torch.empty((max(n, 1),), device=pts.device, dtype=torch.float32)
        check(_lib.lib().nrf_dnerf_time_fwd_train(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                                  _ptr(times), P, int(pts_per_ray), int(bool(times_per_point)),
                                                  _ptr(dx), _ptr(acts), _stream()), "nrf_dnerf_time_fwd_train")
        return dx, acts

    def _bwd_fn(self):
        return _lib.lib().nrf_dnerf_time_bwd


class PackedStraightNet(PackedNet):
    """DNeRF_Straightforward (vanilla_dynamic/network.py:100-196) behind nrf_dnerf_straight_*: NeRF-shaped parameters whose
    trunk input and skip are cat[gamma(x), gamma(t)].  mode 1 = the embedded rows forward(x, ts) receives."""

    def __init__(self, params, D, W, input_ch, input_ch_views, input_ch_time, skips, mode=1, multires=0, multires_views=0,
                 multires_time=0):
        self.params = params
        self.net = NrfNet()
        n = self.net
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch, n.use_viewdirs = D, W, input_ch, input_ch_views, 4, 1
        n.reserved = input_ch_time
        n.skips_mask = sum(1 << int(s) for s in skips)
        self.emb = NrfEmbed(mode, multires, multires_views, multires_time)
        self.n_floats = _lib.lib().nrf_dnerf_straight_packed_floats(C.byref(n), C.byref(self.emb))
        if self.n_floats <= 0:
            raise RuntimeError("nrf_dnerf_straight_packed_floats: " + _lib.lib().nrf_last_error_string().decode())
        self.packed = None
        self._versions = None
        self.out_ch = 4

    def refresh(self):
        vers = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if self.packed is not None and vers == self._versions:
            return
        self._fill_pointers()
        dev = next(iter(self._p.values())).device
        if self.packed is None or self.packed.device != dev:
            self.packed = torch.empty((self.n_floats,), device=dev, dtype=torch.float32)
        check(_lib.lib().nrf_dnerf_straight_pack_weights(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _stream()),
              "nrf_dnerf_straight_pack_weights")
        self._versions = vers

    def query(self, pts, viewdirs, times, pts_per_ray=1, times_per_point=True):
        """mode 1: pts = x [P, input_ch + input_ch_views], times = ts [P, input_ch_time] (both embedded), viewdirs None.
        mode 0: pts [P,3], viewdirs [P/pts_per_ray,3], times [P/pts_per_ray] or [P]."""
        self.refresh()
        pts, times = _f32(pts, "pts"), _f32(times, "times")
        P = pts.shape[0]
        raw = torch.empty((P, 4), device=pts.device, dtype=torch.float32)
        if viewdirs is not None:
            viewdirs = _f32(viewdirs, "viewdirs")
        check(_lib.lib().nrf_dnerf_straight_query_fwd(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                                      _ptr(viewdirs), _ptr(times), P, int(pts_per_ray),
                                                      int(bool(times_per_point)), _ptr(raw), _stream()),
              "nrf_dnerf_straight_query_fwd")
        return raw

    def query_train(self, pts, viewdirs, times, pts_per_ray=1, times_per_point=False):
        """mode 0 -> (raw [P,4], saved workspace) for query_bwd."""
        if self.emb.mode != 0:
            raise NotImplementedError("the straightforward training kernels take raw points (embed mode 0)")
        self.refresh()
        pts, times, viewdirs = _f32(pts, "pts"), _f32(times, "times"), _f32(viewdirs, "viewdirs")
        P = pts.shape[0]
        raw = torch.empty((P, 4), device=pts.device, dtype=torch.float32)
        n = _lib.lib().nrf_train_acts_floats(C.byref(self.net), P)
        acts = torch.empty((max(n, 1),), device=pts.device, dtype=torch.float32)
        check(_lib.lib().nrf_dnerf_straight_fwd_train(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                                      _ptr(viewdirs), _ptr(times), P, int(pts_per_ray),
                                                      int(bool(times_per_point)), _ptr(raw), _ptr(acts), _stream()),
              "nrf_dnerf_straight_fwd_train")
        return raw, acts

    def _bwd_fn(self):
        return _lib.lib().nrf_dnerf_straight_bwd


class PackedStraightNetBf16x3(PackedStraightNet):
    """bf16 weight stream of a DNeRF_Straightforward for nrf_dnerf_straight_query_fwd_bf16x3 (kind "bf16x3": hi/residual
    pairs) or nrf_dnerf_straight_query_fwd_bf16 (kind "bf16"); raw points."""

    def __init__(self, params, D, W, input_ch, input_ch_views, input_ch_time, skips, multires, multires_views, multires_time,
                 kind="bf16x3"):
        self.kind = kind
        self.params = params
        self.net = NrfNet()
        n = self.net
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch, n.use_viewdirs = D, W, input_ch, input_ch_views, 4, 1
        n.reserved = input_ch_time
        n.skips_mask = sum(1 << int(s) for s in skips)
        self.emb = NrfEmbed(0, multires, multires_views, multires_time)
        self.n_bytes = getattr(_lib.lib(), f"nrf_dnerf_straight_{kind}_packed_bytes")(C.byref(n), C.byref(self.emb))
        if self.n_bytes <= 0:
            raise RuntimeError(f"nrf_dnerf_straight_{kind}_packed_bytes: " + _lib.lib().nrf_last_error_string().decode())
        self.packed = None
        self._versions = None
        self.out_ch = 4

    def refresh(self):
        vers = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if self.packed is not None and vers == self._versions:
            return
        self._fill_pointers()
        dev = next(iter(self._p.values())).device
        if self.packed is None or self.packed.device != dev:
            self.packed = torch.empty((self.n_bytes,), device=dev, dtype=torch.uint8)
        check(getattr(_lib.lib(), f"nrf_dnerf_straight_{self.kind}_pack_weights")(
            C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _stream()), f"nrf_dnerf_straight_{self.kind}_pack_weights")
        self._versions = vers

    def query(self, pts, viewdirs, times, pts_per_ray=1, times_per_point=True):
        self.refresh()
        pts, times, viewdirs = _f32(pts, "pts"), _f32(times, "times"), _f32(viewdirs, "viewdirs")
        P = pts.shape[0]
        raw = torch.empty((P, 4), device=pts.device, dtype=torch.float32)
        check(getattr(_lib.lib(), f"nrf_dnerf_straight_query_fwd_{self.kind}")(
            C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts), _ptr(viewdirs), _ptr(times), P,
            int(pts_per_ray), int(bool(times_per_point)), _ptr(raw), _stream()), f"nrf_dnerf_straight_query_fwd_{self.kind}")
        return raw

    def query_train(self, *a, **k):
        raise NotImplementedError("training runs on the exact f32 kernels")


class PackedNetBf16(PackedNet):
    """bf16 weight stream of a NeRF module for nrf_query_fwd_bf16 (raw points + in-kernel encoding only)."""
    _pack_fn, _query_fn = "nrf_mlp_bf16_pack_weights", "nrf_query_fwd_bf16"

    def __init__(self, params, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, multires=0,
                 multires_views=0):
        self.params = params
        self.net = NrfNet()
        n = self.net
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch = D, W, input_ch, input_ch_views, output_ch
        n.use_viewdirs = int(bool(use_viewdirs))
        n.skips_mask = sum(1 << int(s) for s in skips)
        n.use_layernorm = int(any("layernorm" in k for k in params))      # refused by the library: f32 kernels only
        self.emb = NrfEmbed(0, multires, multires_views, 0)
        self.n_bytes = _lib.lib().nrf_mlp_bf16_packed_bytes(C.byref(n), C.byref(self.emb))
        if self.n_bytes <= 0:
            raise RuntimeError("nrf_mlp_bf16_packed_bytes: " + _lib.lib().nrf_last_error_string().decode())
        self.packed = None
        self._versions = None
        self.out_ch = 4 if use_viewdirs else output_ch

    def refresh(self):
        vers = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if self.packed is not None and vers == self._versions:
            return
        self._fill_pointers()
        dev = next(iter(self._p.values())).device
        if self.packed is None or self.packed.device != dev:
            self.packed = torch.empty((self.n_bytes,), device=dev, dtype=torch.uint8)
        check(getattr(_lib.lib(), self._pack_fn)(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _stream()),
              self._pack_fn)
        self._versions = vers

    def query(self, pts, viewdirs=None, pts_per_ray=1):
        self.refresh()
        pts = _f32(pts, "pts")
        P = pts.shape[0]
        raw = torch.empty((P, self.out_ch), device=pts.device, dtype=torch.float32)
        if viewdirs is not None:
            viewdirs = _f32(viewdirs, "viewdirs")
        check(getattr(_lib.lib(), self._query_fn)(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                                  _ptr(viewdirs), P, int(pts_per_ray), _ptr(raw), _stream()),
              self._query_fn)
        return raw

    def query_train(self, *a, **k):
        raise NotImplementedError("the bf16 path is inference only; training runs on the f32 kernels")


class PackedNetF16(PackedNetBf16):
    """The 16-bit variant with f16 hidden activations and weights (saturating at +-65504): nrf_query_fwd_f16.  Same rate as
    bf16, 8x finer rounding; inference only."""
    _pack_fn, _query_fn = "nrf_mlp_f16_pack_weights", "nrf_query_fwd_f16"


class PackedNetBf16x3(PackedNetBf16):
    """bf16x3 weight stream (hi + residual fragments) for nrf_query_fwd_bf16x3: f32-grade accuracy on the bf16 matrix
    pipe, about three times the f32 kernel's rate.  Inference only."""

    def __init__(self, params, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, multires=0,
                 multires_views=0):
        self.params = params
        self.net = NrfNet()
        n = self.net
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch = D, W, input_ch, input_ch_views, output_ch
        n.use_viewdirs = int(bool(use_viewdirs))
        n.skips_mask = sum(1 << int(s) for s in skips)
        self.emb = NrfEmbed(0, multires, multires_views, 0)
        self.n_bytes = _lib.lib().nrf_mlp_bf16x3_packed_bytes(C.byref(n), C.byref(self.emb))
        if self.n_bytes <= 0:
            raise RuntimeError("nrf_mlp_bf16x3_packed_bytes: " + _lib.lib().nrf_last_error_string().decode())
        self.packed = None
        self._versions = None
        self.out_ch = 4 if use_viewdirs else output_ch

    def refresh(self):
        vers = tuple((v.data_ptr(), v._version) for v in self.params.values())
        if self.packed is not None and vers == self._versions:
            return
        self._fill_pointers()
        dev = next(iter(self._p.values())).device
        if self.packed is None or self.packed.device != dev:
            self.packed = torch.empty((self.n_bytes,), device=dev, dtype=torch.uint8)
        check(_lib.lib().nrf_mlp_bf16x3_pack_weights(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _stream()),
              "nrf_mlp_bf16x3_pack_weights")
        self._versions = vers

    def query(self, pts, viewdirs=None, pts_per_ray=1):
        self.refresh()
        pts = _f32(pts, "pts")
        P = pts.shape[0]
        raw = torch.empty((P, self.out_ch), device=pts.device, dtype=torch.float32)
        if viewdirs is not None:
            viewdirs = _f32(viewdirs, "viewdirs")
        check(_lib.lib().nrf_query_fwd_bf16x3(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                              _ptr(viewdirs), P, int(pts_per_ray), _ptr(raw), _stream()),
              "nrf_query_fwd_bf16x3")
        return raw

    def query_train(self, pts, viewdirs=None, pts_per_ray=1, a16=False):
        """Training forward on the bf16x3 arithmetic -> (raw, acts); `acts` has the f32 kernels' layout, so the
        backward entry points of the f32 PackedNet of the same module consume it.  a16: the activations are saved as f16
        (half the bytes; PackedNet.query_bwd(..., x3=True, a16=True) reads them)."""
        self.refresh()
        pts = _f32(pts, "pts")
        P = pts.shape[0]
        raw = torch.empty((P, self.out_ch), device=pts.device, dtype=torch.float32)
        n = _lib.lib().nrf_train_acts_floats(C.byref(self.net), P)
        acts = torch.empty((max(n, 1),), device=pts.device, dtype=torch.float32)
        if viewdirs is not None:
            viewdirs = _f32(viewdirs, "viewdirs")
        fn = "nrf_query_fwd_train_bf16x3_a16" if a16 else "nrf_query_fwd_train_bf16x3"
        check(getattr(_lib.lib(), fn)(C.byref(self.net), C.byref(self.emb), _ptr(self.packed), _ptr(pts),
                                      _ptr(viewdirs), P, int(pts_per_ray), _ptr(raw), _ptr(acts), _stream()), fn)
        return raw, acts
