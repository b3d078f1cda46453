"""NeRF MLP -- counterpart of the reference's network.py.

`NeRF` keeps the reference's constructor, attributes and state_dict keys (real nn.Linear parameters:
`pts_linears.{i}`, `views_linears.0`, `feature_linear`, `alpha_linear`, `rgb_linear` / `output_linear`;
network.py:29-44), so checkpoints and optimizer state round-trip.  The arithmetic runs in the fused
f32-MFMA HIP kernels (nrf_query_fwd / nrf_query_bwd):

  * `forward(x)`            -- x already embedded [P, input_ch + input_ch_views]  (network.py:46-69)
  * `query_points(...)`     -- raw points + per-ray directions, encoding fused in-kernel; used by
                               render.run_network when the embedders are the standard ones.

The packed (MFMA-ordered) weight copy is refreshed lazily whenever a parameter's version changes.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops


class _QueryFunction(torch.autograd.Function):
    """raw = net(points); backward produces parameter gradients with the HIP backward kernels."""

    @staticmethod
    def forward(ctx, module, key, pts, viewdirs, pts_per_ray, need_grad, *params):
        packed = module._packed(key)
        # need_grad is decided by the caller: inside forward grad mode is always off, and
        # ctx.needs_input_grad ignores torch.no_grad()
        if need_grad:
            packed.refresh()                     # the backward reads this f32 copy (W^T blocks of the dgrad chain)
            fwd = packed
            x3_shape = key[0] == 0 and module.W in (128, 256)
            if module.train_precision in ("bf16x3", "bf16x3_a16", "bf16x3_w16") and x3_shape:
                fwd = module._packed(key + ("bf16x3",))      # the forward alone runs on the bf16 matrix pipe
            elif module.train_precision not in ("fp32", "bf16x3", "bf16x3_a16", "bf16x3_w16"):
                raise ValueError("train_precision must be 'fp32', 'bf16x3', 'bf16x3_a16' or 'bf16x3_w16', not "
                                 f"{module.train_precision!r}")
            # "bf16x3_a16": the activations are saved as f16 (half the workspace bytes written and read back); needs the
            # bf16x3 chain kernels (D >= 2) and parameter gradients only, otherwise plain bf16x3
            # "bf16x3_w16": the gradient blocks as (scaled) f16 too, weight-gradient GEMMs as one f16 MFMA per tile pair
            a16 = (fwd is not packed and module.train_precision in ("bf16x3_a16", "bf16x3_w16") and module.D >= 2
                   and not pts.requires_grad)
            if a16 and module.train_precision == "bf16x3_w16":
                a16 = "w16"
            raw, acts = fwd.query_train(pts, viewdirs, pts_per_ray, a16=True) if a16 else fwd.query_train(pts, viewdirs, pts_per_ray)
            ctx.a16 = a16
            ctx.module, ctx.key, ctx.n_points = module, key, pts.shape[0]
            ctx.versions = packed._versions     # the dgrad chain reads the packed copy: it must still be this one
            ctx.x3 = fwd is not packed
            ctx.want_gpts = bool(pts.requires_grad)
            ctx.save_for_backward(acts, *((pts,) if ctx.want_gpts else ()))
            return raw
        return packed.query(pts, viewdirs, pts_per_ray)

    @staticmethod
    def backward(ctx, g_raw):
        acts = ctx.saved_tensors[0]
        packed = ctx.module._packed(ctx.key)
        if packed._versions != ctx.versions:
            # torch autograd raises the same way when a tensor saved for backward was modified in place
            raise RuntimeError("nerf_amd: a parameter of this network was modified (optimizer.step(), load_state_dict) "
                               "between the forward and this backward; the packed weights no longer match the saved "
                               "activations")
        g_pts = None
        if ctx.want_gpts:
            # d(raw)/d(pts) through the in-kernel encoding (what the reference's autograd gives a caller that differentiates
            # w.r.t. the sample positions): nrf_query_bwd_input, the chain the deformation net's training uses
            grads, g_pts = packed.query_bwd(g_raw.contiguous(), ctx.n_points, acts, pts=ctx.saved_tensors[1], x3=ctx.x3)
        else:
            grads = packed.query_bwd(g_raw.contiguous(), ctx.n_points, acts, x3=ctx.x3, a16=ctx.a16)
        order = [name for name, _ in ctx.module.named_parameters()]
        return (None, None, g_pts, None, None, None) + tuple(grads.get(n) for n in order)


class NeRF(nn.Module):
    precision_choices = ("fp32", "bf16", "f16", "bf16x3")                            # inference arithmetic (`precision`)
    train_precision_choices = ("fp32", "bf16x3", "bf16x3_a16", "bf16x3_w16")        # training arithmetic

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.output_ch = output_ch
        self.skips = skips
        self.use_viewdirs = use_viewdirs

        # layer i+1 reads cat[input_pts, h] when i is a skip index (network.py:52-53)
        widths = [input_ch] + [W + input_ch if i in skips else W for i in range(D - 1)]
        self.pts_linears = nn.ModuleList([nn.Linear(w_in, W) for w_in in widths])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._packed_cache = {}
        # "fp32" (exact f32 MFMA, the reference's arithmetic), "bf16" (bf16 MFMA, f32 accumulate), "f16" (the same 16-bit
        # kernel with f16 hidden operands, saturating at +-65504: 8x finer rounding at the same rate) or "bf16x3" (every
        # product as three bf16 MFMAs on hi/residual operands: f32-grade accuracy at ~3x the f32 kernel's rate); the
        # two bf16 modes serve inference through render()'s fused query only.  Not part of the reference's constructor:
        # set as an attribute.
        self.precision = "fp32"
        # arithmetic of queries that need gradients: "fp32" = exact f32 MFMA (the reference's arithmetic); "bf16x3" = the
        # training forward and weight-gradient GEMMs as three bf16 MFMAs per product on hi/residual operands (f32-grade:
        # raw within 1e-5 of its scale, gradients within the same 2e-3 of the reference's autograd); "bf16x3_a16" = the
        # same with the saved activations in f16 (half the bytes of the step's largest workspace: weight gradients carry
        # a 2^-12 rounding of the activations, ~2e-4 of their norm); "bf16x3_w16" = that plus the gradient blocks saved as
        # f16 under a per-pass power-of-two scale, the weight-gradient GEMMs then run as single f16 MFMAs
        self.train_precision = "fp32"

    # ------------------------------------------------------------------ packed weights
    def _packed(self, key):
        """key = (mode, multires, multires_views)."""
        pk = self._packed_cache.get(key)
        if pk is None:
            params = self._pack_params()
            mode, L, Lv = key[:3]
            if len(key) > 3 and key[3] in ("bf16", "f16", "bf16x3"):
                cls = {"bf16": ops.PackedNetBf16, "f16": ops.PackedNetF16, "bf16x3": ops.PackedNetBf16x3}[key[3]]
                pk = cls(params, self.D, self.W, self.input_ch, self.input_ch_views, self.output_ch,
                         list(self.skips), self.use_viewdirs, multires=L, multires_views=Lv)
            else:
                pk = ops.PackedNet(params, self.D, self.W, self.input_ch, self.input_ch_views, self.output_ch,
                                   list(self.skips), self.use_viewdirs, mode=mode, multires=L, multires_views=Lv)
            self._packed_cache[key] = pk
        return pk

    def _pack_params(self):
        params = dict(self.named_parameters())
        if not self.use_viewdirs:
            params = {k: v for k, v in params.items() if not k.startswith("views_linears")}
        return params

    def _apply(self, fn, *args, **kwargs):  # .to()/.cuda() replace parameter tensors: drop stale packed copies
        self._packed_cache = {}
        return super()._apply(fn, *args, **kwargs)

    def _need_grad(self, x=None, input_grad_ok=False):
        if not torch.is_grad_enabled():
            return False
        if x is not None and x.requires_grad:
            if input_grad_ok:
                return True
            # forward(x) on embedded rows differentiates w.r.t. x through _forward_autograd; anything else that reaches
            # here would silently drop d/dx, so refuse
            raise NotImplementedError("nerf_amd: gradients w.r.t. this input are not built for this call "
                                      "(only parameter gradients); detach the input or run under torch.no_grad()")
        return any(p.requires_grad for p in self.parameters())

    def _forward_autograd(self, x):
        """network.py:46-69 composed from torch ops ON THE GPU: taken only when the caller needs d(out)/d(x) of
        already-embedded rows (D-NeRF training with a non-standard embedder re-embeds pts + dx outside the fused
        chain, vanilla_dynamic/network.py:331-339); autograd then covers what the reference's does."""
        F = torch.nn.functional
        input_pts, input_views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = input_pts
        for i in range(self.D):
            h = F.relu(self.pts_linears[i](h))
            if i in self.skips:
                h = torch.cat([input_pts, h], -1)
        if not self.use_viewdirs:
            return self.output_linear(h)
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), input_views], -1)
        h = F.relu(self.views_linears[0](h))
        return torch.cat([self.rgb_linear(h), alpha], -1)

    def _params_in_order(self):
        return [p for _, p in self.named_parameters()]

    # ------------------------------------------------------------------ what the fused kernels take
    def kernel_shape_reason(self, embedded_rows=False):
        """None when the fused kernels are built for this architecture, else the reason (supported ranges).
        The reference has no such limits: architectures outside them run the same arithmetic as a GPU torch composition
        (forward() on embedded rows, which is also what render.run_network's generic route calls), announced once."""
        if not 2 <= self.W <= 256:
            return f"W = {self.W} (the fused kernels take widths up to 256)"
        if self.D > 16:
            return f"D = {self.D} (the fused kernels take D <= 16)"
        if not self.use_viewdirs and self.output_ch > 8:
            return f"output_ch = {self.output_ch} (the fused head writes <= 8 channels)"
        if embedded_rows and (not 1 <= self.input_ch <= 64 or self.input_ch_views > 32):
            return (f"{self.input_ch} + {self.input_ch_views} embedded channels (the fused kernels read <= 64 position and "
                    "<= 32 direction channels)")
        return None

    def _composed(self, x2, reason):
        if not getattr(self, "_warned_composed", False):
            import warnings
            warnings.warn(f"nerf_amd: {reason}: this network runs as a GPU torch composition, not on the fused HIP kernels")
            self._warned_composed = True
        return self._forward_autograd(x2)

    # ------------------------------------------------------------------ queries
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("nerf_amd.NeRF.forward: input is on the CPU; nerf_amd has no CPU path")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        reason = self.kernel_shape_reason(embedded_rows=True)
        if reason is not None:
            return self._composed(x2, reason).reshape(*lead, -1)
        if torch.is_grad_enabled() and x2.requires_grad:
            return self._forward_autograd(x2).reshape(*lead, -1)
        out = _QueryFunction.apply(self, (1, 0, 0), x2, None, 1, self._need_grad(x2), *self._params_in_order())
        return out.reshape(*lead, out.shape[-1])

    def query_points(self, pts, viewdirs, pts_per_ray, multires, multires_views):
        """pts [P,3]; viewdirs [P/pts_per_ray, 3] or None -> raw [P, 4|output_ch] (render.py:31-56 fused)."""
        key = (0, multires, multires_views if self.use_viewdirs else 0)
        if torch.is_grad_enabled() and viewdirs is not None and viewdirs.requires_grad and self.use_viewdirs:
            raise NotImplementedError("nerf_amd: gradients w.r.t. the view directions are not built (parameters and query "
                                      "points only); detach viewdirs")
        need_grad = self._need_grad(pts, input_grad_ok=True)      # d/d(pts): nrf_query_bwd_input
        if self.precision in ("bf16", "f16", "bf16x3") and not need_grad:   # inference only; training stays exact f32
            key = key + (self.precision,)
        elif self.precision not in ("fp32", "bf16", "f16", "bf16x3"):
            raise ValueError(f"precision must be 'fp32', 'bf16', 'f16' or 'bf16x3', not {self.precision!r}")
        return _QueryFunction.apply(self, key, pts, viewdirs if self.use_viewdirs else None, pts_per_ray,
                                    need_grad, *self._params_in_order())

    # ------------------------------------------------------------------ Keras import (network.py:71-99)
    def load_weights_from_keras(self, weights):
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"

        def put(linear, idx):
            linear.weight.data = torch.from_numpy(np.transpose(weights[idx]))
            linear.bias.data = torch.from_numpy(np.transpose(weights[idx + 1]))
        for i in range(self.D):
            put(self.pts_linears[i], 2 * i)
        put(self.feature_linear, 2 * self.D)
        put(self.views_linears[0], 2 * self.D + 2)
        put(self.rgb_linear, 2 * self.D + 4)
        put(self.alpha_linear, 2 * self.D + 6)
        self._packed_cache = {}
