"""D-NeRF networks -- counterpart of vanilla_dynamic/network.py.

  NeRF                   :13-97    canonical net; forward(x, ts=None) -> (out, zeros[P,3]); optional LayerNorm
  DNeRF_Straightforward  :100-196  one net on cat[gamma(x), gamma(t)]
  DNeRF_Deformation      :199-341  time net -> dx -> gamma(x + dx) -> canonical net

State_dict keys are the reference's (`_occ.*`, `_time.{i}.*`, `_time_out.*`, `pts_layernorms.*`, ...).
DNeRF_Deformation runs as ONE fused HIP kernel (nrf_dnerf_query_fwd) whenever the inputs allow it: raw
points / raw time available (run_network_dnerf always passes them), standard embedders, no LayerNorm.
Training goes through `_DeformFunction`: time-net forward with saved activations -> x + dx -> canonical forward-train;
backward = nrf_query_bwd_input (canonical parameter gradients + d/d(x + dx)) -> nrf_dnerf_time_bwd.  Configurations
outside the fused kernels (LayerNorm, non-standard embedders) are composed from GPU torch ops + nrf_pe_fwd/_bwd.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops
from ..network import NeRF as _StaticNeRF


class _DeformFunction(torch.autograd.Function):
    """(raw, dx) = deformation chain with saved activations; backward = parameter gradients of both nets
    (autograd of vanilla_dynamic/network.py:307-341 incl. the path through embed_fn(x + dx))."""

    @staticmethod
    def forward(ctx, module, pts, viewdirs, times, pts_per_ray, times_per_point, netchunk, L, Lv, Lt, *params):
        tnet = module._time_packed(L, Lt)
        occ = module._occ._packed((0, L, Lv))
        P = pts.shape[0]
        dx, acts_t = tnet.query_train(pts, times, pts_per_ray, times_per_point)
        # network.py:315-322: a netchunk whose FIRST point has t == 0 takes the canonical shortcut (dx = 0, no time net)
        t_pt = times if times_per_point else times.repeat_interleave(pts_per_ray)
        first = (torch.arange(P, device=pts.device) // netchunk) * netchunk
        skip = (t_pt[first] == 0.0) if module.zero_canonical else torch.zeros(P, dtype=torch.bool, device=pts.device)
        dx = torch.where(skip[:, None], torch.zeros_like(dx), dx)
        x2 = pts + dx
        # train_precision "bf16x3": the canonical net's forward and every weight-gradient GEMM as three bf16 MFMAs per
        # product, and the deformation net's activation-gradient chain; the deformation net's forward (dx feeds a frequency-512
        # encoding) and the canonical net's activation-gradient chain (which carries d/dx) stay exact f32
        x3 = (module.train_precision == "bf16x3" and module.W in (128, 256) and module.W_deform in (128, 256)
              and not module.use_layernorm)      # the LayerNorm backward is built in f32
        if module.train_precision not in ("fp32", "bf16x3"):
            raise ValueError(f"train_precision must be 'fp32' or 'bf16x3', not {module.train_precision!r}")
        occ.refresh()
        fwd = module._occ._packed((0, L, Lv, "bf16x3")) if x3 else occ
        raw, acts_c = fwd.query_train(x2, viewdirs, pts_per_ray)
        ctx.x3 = x3
        ctx.module, ctx.keys, ctx.n_points = module, (L, Lv, Lt), P
        ctx.versions = (tnet._versions, occ._versions)
        ctx.save_for_backward(acts_t, acts_c, x2, skip)
        ctx.set_materialize_grads(False)
        return raw, dx

    @staticmethod
    def backward(ctx, g_raw, g_dx_out):
        acts_t, acts_c, x2, skip = ctx.saved_tensors
        module, (L, Lv, Lt) = ctx.module, ctx.keys
        occ = module._occ._packed((0, L, Lv))
        tnet = module._time_packed(L, Lt)
        if (tnet._versions, occ._versions) != ctx.versions:
            raise RuntimeError("nerf_amd: a parameter of this network was modified between the forward and this backward; "
                               "the packed weights no longer match the saved activations")
        if g_raw is None:
            g_raw = torch.zeros((ctx.n_points, 4), device=x2.device)
        g_occ, g_x2 = occ.query_bwd(g_raw.contiguous(), ctx.n_points, acts_c, pts=x2, x3="wgrad" if ctx.x3 else False)
        g_dx = g_x2 if g_dx_out is None else g_x2 + g_dx_out
        g_dx = torch.where(skip[:, None], torch.zeros_like(g_dx), g_dx)
        # the deformation net's own chain ends at its parameters: on the bf16x3 pipe as well
        g_time = tnet.query_bwd(g_dx.contiguous(), ctx.n_points, acts_t, x3=bool(ctx.x3))
        out = []
        for name, _ in module.named_parameters():
            if name.startswith("_occ."):
                out.append(g_occ.get(name[5:]))
            elif name.startswith("_time_out."):
                out.append(g_time.get("output_linear." + name[10:]))
            elif name.startswith("_time."):
                out.append(g_time.get("pts_linears." + name[6:]))
            elif name.startswith("_time_layernorms."):
                out.append(g_time.get("pts_layernorms." + name[17:]))
            else:
                out.append(None)
        return (None,) * 10 + tuple(out)


def _layer_norm(h, ln):
    """nn.LayerNorm(W) (vanilla_dynamic/network.py:46-48, :276-277) from elementwise ops and two row reductions.
    The module `ln` only holds the parameters (state_dict keys): torch-ROCm's fused LayerNorm BACKWARD kernel was
    measured 8e-3 off in f32 on gfx950 (gradient norms against the reference's CPU autograd and against the same graph in
    f64; this composition: 2e-5), which would break gradient parity of the use_layernorm configurations."""
    mu = h.mean(-1, keepdim=True)
    d = h - mu
    var = (d * d).mean(-1, keepdim=True)
    return d / torch.sqrt(var + ln.eps) * ln.weight + ln.bias


class NeRF(_StaticNeRF):
    """Canonical net: the static NeRF plus the (out, zeros) return convention and optional LayerNorm."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 use_layernorm=False):
        super().__init__(D=D, W=W, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=output_ch,
                         skips=skips, use_viewdirs=use_viewdirs)
        self.use_layernorm = use_layernorm
        if use_layernorm:
            self.pts_layernorms = nn.ModuleList([nn.LayerNorm(W) for _ in range(D)])
            self.views_layernorm = nn.LayerNorm(W // 2)

    def _pack_params(self):
        params = super()._pack_params()
        if not self.use_viewdirs:
            params = {k: v for k, v in params.items() if not k.startswith("views_layernorm")}
        return params

    def kernel_shape_reason(self, embedded_rows=False):
        if self.use_layernorm and self.W not in (64, 128, 256):
            return f"W = {self.W} with LayerNorm (the fused LayerNorm kernels are built for 64, 128, 256)"
        return super().kernel_shape_reason(embedded_rows)

    def _ln_fused(self, x):
        # the fused f32 kernels: the LayerNorm instantiation of the query kernel, and -- when parameters need gradients --
        # its training twin with the LayerNorm backward inside the dgrad chain; d/d(embedded rows) is the composition's
        return (x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad)
                and self.kernel_shape_reason(embedded_rows=True) is None)

    def forward(self, x, ts=None):
        if self.use_layernorm and not self._ln_fused(x):
            out = self._forward_layernorm(x)
        else:
            out = super().forward(x)
        return out, torch.zeros_like(x[:, :3])

    def query_points(self, pts, viewdirs, pts_per_ray, multires, multires_views):
        if self.use_layernorm and self.train_precision != "fp32" and self._need_grad(pts, input_grad_ok=True):
            raise RuntimeError("nerf_amd: LayerNorm nets train on the f32 kernels only (train_precision = 'fp32')")
        return super().query_points(pts, viewdirs, pts_per_ray, multires, multires_views)

    def _forward_layernorm(self, x):
        # LayerNorm variant (network.py:46-48, :75-76, :88-89) from GPU torch ops: taken when the caller differentiates
        # w.r.t. the embedded rows, and for widths without a fused LayerNorm kernel.
        input_pts, input_views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = input_pts
        for i in range(self.D):
            h = F.relu(_layer_norm(self.pts_linears[i](h), self.pts_layernorms[i]))
            if i in self.skips:
                h = torch.cat([input_pts, h], -1)
        if not self.use_viewdirs:
            return self.output_linear(h)
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), input_views], -1)
        h = F.relu(_layer_norm(self.views_linears[0](h), self.views_layernorm))
        return torch.cat([self.rgb_linear(h), alpha], -1)


class _StraightFunction(torch.autograd.Function):
    """raw = DNeRF_Straightforward on raw points; backward = parameter gradients from the HIP backward kernels
    (nrf_dnerf_straight_fwd_train / nrf_dnerf_straight_bwd)."""

    @staticmethod
    def forward(ctx, module, pts, viewdirs, times, pts_per_ray, times_per_point, key, *params):
        packed = module._raw_packed(key)
        raw, acts = packed.query_train(pts, viewdirs, times, pts_per_ray, times_per_point)
        ctx.module, ctx.key, ctx.n_points, ctx.versions = module, key, pts.shape[0], packed._versions
        ctx.save_for_backward(acts)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        (acts,) = ctx.saved_tensors
        packed = ctx.module._raw_packed(ctx.key)
        if packed._versions != ctx.versions:
            raise RuntimeError("nerf_amd: a parameter of this network was modified between the forward and this backward; "
                               "the packed weights no longer match the saved activations")
        grads = packed.query_bwd(g_raw.contiguous(), ctx.n_points, acts)
        return (None,) * 7 + tuple(grads.get(n) for n, _ in ctx.module.named_parameters())


class DNeRF_Straightforward(nn.Module):
    """cat[gamma(x), gamma(t)] -> NeRF-shaped MLP (network.py:100-196).  NB: at the reference's HEAD this
    network type cannot be rendered (run_network_dnerf passes pts_raw=/t_raw= which this forward does not
    accept), but forward(x, ts) itself is callable: without gradients it runs the fused f32 kernel on the
    embedded rows (nrf_dnerf_straight_query_fwd, embed mode 1); with gradients, or for shapes the kernel does not take, the
    same arithmetic composed from GPU torch ops.  Both pinned to the reference's outputs (tests/golden/dnerf2.npz)."""

    precision_choices = ("fp32", "bf16", "bf16x3")
    train_precision_choices = ("fp32",)

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, input_ch_time=1, output_ch=4, skips=[4],
                 use_viewdirs=True):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views, self.input_ch_time = input_ch, input_ch_views, input_ch_time
        self.skips, self.use_viewdirs = skips, use_viewdirs
        in0 = input_ch + input_ch_time
        self.pts_linears = nn.ModuleList(
            [nn.Linear(in0, W)] + [nn.Linear(W + in0 if i in skips else W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._packed = None
        # "fp32" (the reference's arithmetic), "bf16" (bf16 MFMA, tolerance 1e-2) or "bf16x3" (three bf16 MFMAs per product,
        # f32-grade results): inference
        # through the render stack's fused query only; training stays exact f32.  Not a constructor argument of the reference
        self.precision = "fp32"

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        self._packed_raw = None
        self._packed_x3 = None
        return super()._apply(fn, *args, **kwargs)

    def _kernel_takes(self):
        return (self.use_viewdirs and self.W in (64, 128, 256) and self.D <= 16
                and self.input_ch <= 64 and self.input_ch_views <= 32 and self.input_ch_time <= 24
                and (self.D - 1) not in self.skips and len([s for s in self.skips if s < self.D - 1]) <= 1)

    def _fusable(self, x, t_embed):
        need_grad = torch.is_grad_enabled() and (x.requires_grad or t_embed.requires_grad
                                                 or any(p.requires_grad for p in self.parameters()))
        return x.is_cuda and not need_grad and self._kernel_takes()

    def query_points(self, pts, viewdirs, times, pts_per_ray, times_per_point, multires, multires_views, multires_time):
        """Raw points [P,3], per-ray directions [P/pts_per_ray,3], times [P] or [P/pts_per_ray] -> raw [P,4]: the fused f32
        kernel with in-kernel encodings (nrf_dnerf_straight_query_fwd, embed mode 0).  Inference only."""
        key = (multires, multires_views, multires_time)
        times = times.reshape(-1)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if pts.requires_grad or times.requires_grad or viewdirs.requires_grad:
                raise NotImplementedError("nerf_amd: the fused straightforward query differentiates w.r.t. the parameters "
                                          "only; call forward(x, ts) on embedded rows for input gradients")
            return _StraightFunction.apply(self, pts, viewdirs, times, int(pts_per_ray), bool(times_per_point), key,
                                           *[p for _, p in self.named_parameters()])
        if self.precision in ("bf16", "bf16x3"):
            key = key + (self.precision,)
            if getattr(self, "_packed_x3_key", None) != key or getattr(self, "_packed_x3", None) is None:
                self._packed_x3 = ops.PackedStraightNetBf16x3(dict(self.named_parameters()), self.D, self.W, self.input_ch,
                                                              self.input_ch_views, self.input_ch_time,
                                                              [s for s in self.skips if s < self.D - 1], *key[:3],
                                                              kind=self.precision)
                self._packed_x3_key = key
            return self._packed_x3.query(pts, viewdirs, times, pts_per_ray, times_per_point)
        if self.precision != "fp32":
            raise ValueError(f"nerf_amd: unknown precision {self.precision!r} (fp32, bf16 or bf16x3)")
        return self._raw_packed(key).query(pts, viewdirs, times, pts_per_ray, times_per_point)

    def _raw_packed(self, key):
        if getattr(self, "_packed_raw_key", None) != key or getattr(self, "_packed_raw", None) is None:
            self._packed_raw = ops.PackedStraightNet(dict(self.named_parameters()), self.D, self.W, self.input_ch,
                                                     self.input_ch_views, self.input_ch_time,
                                                     [s for s in self.skips if s < self.D - 1], mode=0, multires=key[0],
                                                     multires_views=key[1], multires_time=key[2])
            self._packed_raw_key = key
        return self._packed_raw

    def forward(self, x, ts):
        t_embed = ts[0] if isinstance(ts, (list, tuple)) else ts
        if self._fusable(x, t_embed):
            if self._packed is None:
                params = {k: v for k, v in self.named_parameters()}
                self._packed = ops.PackedStraightNet(params, self.D, self.W, self.input_ch, self.input_ch_views,
                                                     self.input_ch_time, [s for s in self.skips if s < self.D - 1])
            out = self._packed.query(x.reshape(-1, x.shape[-1]), None, t_embed.reshape(-1, t_embed.shape[-1]))
            return out, torch.zeros_like(x[:, :3])
        input_pts, input_views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        xt = torch.cat([input_pts, t_embed], dim=-1)
        h = xt
        for i in range(self.D):
            h = F.relu(self.pts_linears[i](h))
            if i in self.skips:
                h = torch.cat([xt, h], -1)
        if self.use_viewdirs:
            alpha = self.alpha_linear(h)
            h = torch.cat([self.feature_linear(h), input_views], -1)
            h = F.relu(self.views_linears[0](h))
            out = torch.cat([self.rgb_linear(h), alpha], -1)
        else:
            out = self.output_linear(h)
        return out, torch.zeros_like(input_pts[:, :3])


_TWO_LAUNCH_POINTS = 65536      # points per time-net launch of the two-launch inference chain (bounds its workspace)


class DNeRF_Deformation(nn.Module):
    precision_choices = ("fp32", "bf16", "bf16x3")
    train_precision_choices = ("fp32", "bf16x3")

    def __init__(self, D=8, W=256, D_deform=8, W_deform=256, input_ch=3, input_ch_views=3, input_ch_time=1,
                 output_ch=4, skips=[4], skips_deform=[4], use_viewdirs=True, zero_canonical=True,
                 use_layernorm=False):
        super().__init__()
        self.D, self.W = D, W
        self.D_deform, self.W_deform = D_deform, W_deform
        self.input_ch, self.input_ch_views, self.input_ch_time = input_ch, input_ch_views, input_ch_time
        self.skips, self.use_viewdirs = skips, use_viewdirs
        self.zero_canonical, self.use_layernorm = zero_canonical, use_layernorm
        self.embed_fn = None
        self._occ = NeRF(D=D, W=W, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=output_ch,
                         skips=skips, use_viewdirs=use_viewdirs, use_layernorm=use_layernorm)
        self.skips_deform = skips   # the `skips_deform` argument is ignored, as in the reference (network.py:245)
        widths = [input_ch + input_ch_time] + [W_deform + input_ch if i in self.skips_deform else W_deform
                                               for i in range(D_deform - 1)]
        self._time = nn.ModuleList([nn.Linear(w_in, W_deform) for w_in in widths])
        self._time_out = nn.Linear(W_deform, 3)
        self._time_layernorms = nn.ModuleList([nn.LayerNorm(W_deform) for _ in range(D_deform)]) \
            if use_layernorm else None
        self._packed_time = None
        self._time_versions = None
        self._packed_bf16 = None
        self._bf16_versions = None
        self._time_net = None
        self._time_net_key = None
        # "fp32" (the reference's arithmetic), "bf16" (bf16 MFMA, f32 accumulate) or "bf16x3" (three bf16 MFMAs per
        # product, f32-grade results); the last two are inference only.  An attribute, not a constructor argument of
        # the reference
        self.precision = "fp32"
        self.train_precision = "fp32"     # "bf16x3": see _DeformFunction.forward

    def _apply(self, fn, *args, **kwargs):
        self._packed_time = None
        self._packed_bf16 = None
        self._time_net = None
        return super()._apply(fn, *args, **kwargs)

    def _training_call(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def kernel_shape_reason(self):
        """None when the fused kernels take both nets (nerf_amd.network.NeRF.kernel_shape_reason); otherwise forward()
        composes the chain from GPU torch ops."""
        if self.W_deform not in (64, 128, 256):
            return f"W_deform = {self.W_deform} (the fused kernels are built for 64, 128, 256)"
        if self.D_deform > 16:
            return f"D_deform = {self.D_deform} (the fused kernels take <= 16 layers)"
        return self._occ.kernel_shape_reason()

    # ------------------------------------------------------------------ fused path
    def _time_desc(self, L, Lt):
        n = _lib.NrfNet()
        n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch, n.use_viewdirs = \
            self.D_deform, self.W_deform, self.input_ch, self.input_ch_time, 3, 0
        n.skips_mask = sum(1 << int(s) for s in self.skips_deform)
        return n, _lib.NrfEmbed(0, L, 0, Lt)

    def _refresh_time(self, L, Lt):
        params = [p for l in self._time for p in (l.weight, l.bias)] + [self._time_out.weight, self._time_out.bias]
        vers = (L, Lt) + tuple((p.data_ptr(), p._version) for p in params)
        if self._packed_time is not None and vers == self._time_versions:
            return
        n, e = self._time_desc(L, Lt)
        self._tkeep = [ops._f32(p.detach(), "time-net parameter") for p in params]
        for i in range(self.D_deform):
            n.pts_w[i] = C.c_void_p(self._tkeep[2 * i].data_ptr())
            n.pts_b[i] = C.c_void_p(self._tkeep[2 * i + 1].data_ptr())
        n.out_w, n.out_b = C.c_void_p(self._tkeep[-2].data_ptr()), C.c_void_p(self._tkeep[-1].data_ptr())
        lib = _lib.lib()
        nfl = lib.nrf_dnerf_time_packed_floats(C.byref(n), C.byref(e))
        if nfl <= 0:
            raise RuntimeError("nrf_dnerf_time_packed_floats: " + lib.nrf_last_error_string().decode())
        if self._packed_time is None or self._packed_time.numel() != nfl:
            self._packed_time = torch.empty((nfl,), device=self._tkeep[0].device, dtype=torch.float32)
        _lib.check(lib.nrf_dnerf_pack_time_weights(C.byref(n), C.byref(e), ops._ptr(self._packed_time), ops._stream()),
                   "nrf_dnerf_pack_time_weights")
        self._time_versions = vers

    def _time_packed(self, L, Lt):
        key = (L, Lt)
        if self._time_net is None or self._time_net_key != key:
            params = {}
            for i, l in enumerate(self._time):
                params[f"pts_linears.{i}.weight"], params[f"pts_linears.{i}.bias"] = l.weight, l.bias
            params["output_linear.weight"], params["output_linear.bias"] = self._time_out.weight, self._time_out.bias
            if self._time_layernorms is not None:
                for i, ln in enumerate(self._time_layernorms):
                    params[f"pts_layernorms.{i}.weight"], params[f"pts_layernorms.{i}.bias"] = ln.weight, ln.bias
            self._time_net = ops.PackedTimeNet(params, self.D_deform, self.W_deform, self.input_ch, self.input_ch_time,
                                               list(self.skips_deform), L, Lt)
            self._time_net_key = key
        return self._time_net

    def _time_params(self):
        return [p for l in self._time for p in (l.weight, l.bias)] + [self._time_out.weight, self._time_out.bias]

    def _fill_time_pointers(self, n, keep):
        for i in range(self.D_deform):
            n.pts_w[i] = C.c_void_p(keep[2 * i].data_ptr())
            n.pts_b[i] = C.c_void_p(keep[2 * i + 1].data_ptr())
        n.out_w, n.out_b = C.c_void_p(keep[-2].data_ptr()), C.c_void_p(keep[-1].data_ptr())

    def _query_points_bf16(self, pts, viewdirs, times, pts_per_ray, times_per_point, netchunk, L, Lv, Lt, kind="bf16"):
        """nrf_dnerf_query_fwd_bf16 / _bf16x3: one bf16 (or hi/lo bf16 pair) weight stream for both nets."""
        lib = _lib.lib()
        f_bytes = getattr(lib, f"nrf_dnerf_{kind}_packed_bytes")
        f_pack = getattr(lib, f"nrf_dnerf_{kind}_pack_weights")
        f_query = getattr(lib, f"nrf_dnerf_query_fwd_{kind}")
        occ = self._occ._packed((0, L, Lv))          # its descriptor carries the canonical net's f32 parameter pointers
        occ._fill_pointers()
        tn, e = self._time_desc(L, Lt)
        e.multires_views = Lv
        tparams = self._time_params()
        keep = [ops._f32(p.detach(), "time-net parameter") for p in tparams]
        self._fill_time_pointers(tn, keep)
        vers = (kind, L, Lv, Lt) + tuple((p.data_ptr(), p._version) for p in tparams + list(self._occ.parameters()))
        if self._packed_bf16 is None or vers != self._bf16_versions:
            nb = f_bytes(C.byref(tn), C.byref(occ.net), C.byref(e))
            if nb <= 0:
                raise RuntimeError(f"nrf_dnerf_{kind}_packed_bytes: " + lib.nrf_last_error_string().decode())
            if self._packed_bf16 is None or self._packed_bf16.numel() != nb:
                self._packed_bf16 = torch.empty((nb,), device=keep[0].device, dtype=torch.uint8)
            _lib.check(f_pack(C.byref(tn), C.byref(occ.net), C.byref(e), ops._ptr(self._packed_bf16), ops._stream()),
                       f"nrf_dnerf_{kind}_pack_weights")
            self._bf16_versions = vers
        P = pts.shape[0]
        raw = torch.empty((P, 4), device=pts.device, dtype=torch.float32)
        dx = torch.empty((P, 3), device=pts.device, dtype=torch.float32)
        _lib.check(f_query(
            C.byref(tn), C.byref(occ.net), C.byref(e), ops._ptr(self._packed_bf16), ops._ptr(pts), ops._ptr(viewdirs),
            ops._ptr(times), P, int(pts_per_ray), int(bool(times_per_point)), int(netchunk),
            int(bool(self.zero_canonical)), ops._ptr(raw), ops._ptr(dx), ops._stream()), f"nrf_dnerf_query_fwd_{kind}")
        return raw, dx

    def _query_points_two_launch(self, pts, viewdirs, times, pts_per_ray, times_per_point, netchunk, L, Lv, Lt):
        """The chain as time-net launch -> x + dx -> canonical launch (the kernels of the training path, nothing kept):
        same arithmetic as the fused kernel, for nets of different widths."""
        tnet = self._time_packed(L, Lt)
        # the time net runs on the kernel instantiation that saves activations (~10 KB per point at D = 8, W = 256, which
        # inference never reads): in slices of whole rays of at most _TWO_LAUNCH_POINTS points through ONE reused
        # workspace (0.66 GB), not over the 6.3 M points of a 32768-ray fine chunk at once (60 GB)
        P = pts.shape[0]
        step = max(1, _TWO_LAUNCH_POINTS // pts_per_ray) * pts_per_ray
        if P <= step:
            dx, _ = tnet.query_train(pts, times, pts_per_ray, times_per_point)
        else:
            dx, acts = torch.empty((P, 3), device=pts.device, dtype=torch.float32), None
            for p0 in range(0, P, step):
                p1 = min(P, p0 + step)
                t = times[p0:p1] if times_per_point else times[p0 // pts_per_ray:-(-p1 // pts_per_ray)]
                dx[p0:p1], acts = tnet.query_train(pts[p0:p1], t, pts_per_ray, times_per_point, acts=acts)
            del acts
        if self.zero_canonical:   # network.py:315-322: a netchunk whose FIRST point has t == 0 skips the time net
            P = pts.shape[0]
            t_pt = times if times_per_point else times.repeat_interleave(pts_per_ray)
            first = (torch.arange(P, device=pts.device) // netchunk) * netchunk
            dx = torch.where((t_pt[first] == 0.0)[:, None], torch.zeros_like(dx), dx)
        raw = self._occ._packed((0, L, Lv)).query(pts + dx, viewdirs, pts_per_ray)
        return raw, dx

    def query_points(self, pts, viewdirs, times, pts_per_ray, times_per_point, netchunk, multires, multires_views,
                     multires_time):
        """Fused deformation chain on raw inputs: pts [P,3], viewdirs [P/pts_per_ray,3], times [P] or
        [P/pts_per_ray] -> (raw [P,4], dx [P,3])."""
        if not self.use_viewdirs:
            raise RuntimeError("nerf_amd: the fused deformation kernel needs use_viewdirs=True")
        pts, viewdirs = ops._f32(pts, "pts"), ops._f32(viewdirs, "viewdirs")
        times = ops._f32(times, "times").reshape(-1)
        if self.use_layernorm:
            # network.py:276-300 + :46-48: both nets normalise between Linear and ReLU.  Inference runs the chain as its
            # two halves on the LayerNorm instantiations of the f32 query kernel; training (below) on their training twins
            if self._training_call():
                if pts.requires_grad or times.requires_grad:
                    raise NotImplementedError("nerf_amd: gradients w.r.t. points / times are not built (parameters only)")
                return _DeformFunction.apply(self, pts, viewdirs, times, int(pts_per_ray), bool(times_per_point),
                                             int(netchunk), multires, multires_views, multires_time,
                                             *[p for _, p in self.named_parameters()])
            if self.precision != "fp32":
                raise ValueError(f"nerf_amd: precision {self.precision!r} is not built for use_layernorm nets (fp32 only)")
            return self._query_points_two_launch(pts, viewdirs, times, int(pts_per_ray), bool(times_per_point),
                                                 int(netchunk), multires, multires_views, multires_time)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: exact f32 kernels with saved activations (whatever `precision` says)
            if pts.requires_grad or times.requires_grad:
                raise NotImplementedError("nerf_amd: gradients w.r.t. points / times are not built (parameters only)")
            return _DeformFunction.apply(self, pts, viewdirs, times, int(pts_per_ray), bool(times_per_point),
                                         int(netchunk), multires, multires_views, multires_time,
                                         *[p for _, p in self.named_parameters()])
        if self.W != self.W_deform:
            # create_dnerf fixes the deformation net at 8 x 256 whatever --netwidth says (vanilla_dynamic/model.py:75-77);
            # the fused chain kernels need both nets at one width, so unequal widths run the chain as its two halves
            if self.precision != "fp32":
                raise ValueError(f"nerf_amd: precision {self.precision!r} needs the deformation and canonical nets at "
                                 f"the same width (W_deform = {self.W_deform}, W = {self.W})")
            return self._query_points_two_launch(pts, viewdirs, times, int(pts_per_ray), bool(times_per_point),
                                                 int(netchunk), multires, multires_views, multires_time)
        if self.precision in ("bf16", "bf16x3"):
            return self._query_points_bf16(pts, viewdirs, times, pts_per_ray, times_per_point, netchunk, multires,
                                           multires_views, multires_time, kind=self.precision)
        if self.precision != "fp32":
            raise ValueError(f"nerf_amd: unknown precision {self.precision!r} (fp32, bf16 or bf16x3)")
        self._refresh_time(multires, multires_time)
        occ = self._occ._packed((0, multires, multires_views))
        occ.refresh()
        tn, e = self._time_desc(multires, multires_time)
        e.multires_views = multires_views
        P = pts.shape[0]
        raw = torch.empty((P, 4), device=pts.device, dtype=torch.float32)
        dx = torch.empty((P, 3), device=pts.device, dtype=torch.float32)
        _lib.check(_lib.lib().nrf_dnerf_query_fwd(
            C.byref(tn), C.byref(occ.net), C.byref(e), ops._ptr(self._packed_time), ops._ptr(occ.packed),
            ops._ptr(pts), ops._ptr(viewdirs), ops._ptr(times), P, int(pts_per_ray), int(bool(times_per_point)),
            int(netchunk), int(bool(self.zero_canonical)), ops._ptr(raw), ops._ptr(dx), ops._stream()),
            "nrf_dnerf_query_fwd")
        return raw, dx

    # ------------------------------------------------------------------ reference-shaped forward
    def _fusable_forward(self, x, pts_raw, t_raw):
        from ..positional_encoding import _EmbedFn
        ef = self.embed_fn
        return (pts_raw is not None and t_raw is not None and isinstance(ef, _EmbedFn) and ef.fusable
                and ef.out_dim == self.input_ch and self.use_viewdirs and self.kernel_shape_reason() is None
                and not (self.use_layernorm and (self.W_deform not in (64, 128, 256) or self.W not in (64, 128, 256)))
                and (self.input_ch_views - 3) % 6 == 0 and (self.input_ch_time - 1) % 2 == 0
                and (self.input_ch_views - 3) // 6 <= 4 and (self.input_ch_time - 1) // 2 <= 10
                and x.is_cuda)

    def forward(self, x, ts, pts_raw=None, t_raw=None):
        """x [P, input_ch + input_ch_views] embedded; ts embedded time (or [t, t]); returns (out, dx)."""
        if self._fusable_forward(x, pts_raw, t_raw):
            # the direction embedding starts with the raw direction (include_input): x[:, input_ch : input_ch+3]
            dirs = x[:, self.input_ch:self.input_ch + 3]
            P = x.shape[0]
            return self.query_points(pts_raw, dirs, t_raw, 1, True, max(P, 1), self.embed_fn.multires,
                                     (self.input_ch_views - 3) // 6, (self.input_ch_time - 1) // 2)
        return self._forward_torch(x, ts, pts_raw, t_raw)

    def _forward_torch(self, x, ts, pts_raw, t_raw):
        # generic composition for configurations the fused kernel does not cover (LayerNorm, exotic embedders)
        input_pts, input_views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        t = ts[0] if isinstance(ts, (list, tuple)) else ts
        probe = t_raw if t_raw is not None else t
        cur_time = probe[0, 0].item() if probe.numel() > 0 else 1.0
        if cur_time == 0.0 and self.zero_canonical:
            dx = torch.zeros(input_pts.shape[0], 3, device=input_pts.device)
            canonical_input = torch.cat([input_pts, input_views], dim=-1)
        else:
            h = torch.cat([input_pts, t], dim=-1)
            for i in range(self.D_deform):
                h = self._time[i](h)
                if self._time_layernorms is not None:
                    h = _layer_norm(h, self._time_layernorms[i])
                h = F.relu(h)
                if i in self.skips_deform:
                    h = torch.cat([input_pts, h], -1)
            dx = self._time_out(h)
            base = pts_raw if pts_raw is not None else input_pts[:, :3]
            if self.embed_fn is None:
                raise RuntimeError("embed_fn must be set for deformation network")
            canonical_input = torch.cat([self.embed_fn(base + dx), input_views], dim=-1)
        out, _ = self._occ(canonical_input, t)
        return out, dx
