"""Positional encoding -- counterpart of the reference's positional_encoding.py.

Same surface: `Embedder(**kwargs)` with `.embed(x)` / `.out_dim` (positional_encoding.py:5-36; also
built directly by train_2d.py:136 with input_dims=2) and `get_embedder(multires, i=0)` ->
(callable, out_dim) (positional_encoding.py:39-64).

On the render path the encoding is NOT materialised: `run_network` recognises an `_EmbedFn` whose
description the fused query kernel supports and hands raw points to it (nrf_query_fwd).  Calling the
embedder directly runs the standalone HIP kernel nrf_pe_fwd.
"""
import torch
import torch.nn as nn

from . import ops


class _PEFunction(torch.autograd.Function):
    """gamma(x) with a hand-written backward (needed by D-NeRF, where the canonical net embeds pts + dx)."""

    @staticmethod
    def forward(ctx, x, num_freqs, max_freq_log2, log_sampling, include_input):
        ctx.save_for_backward(x)
        ctx.cfg = (num_freqs, max_freq_log2, log_sampling, include_input)
        return ops.pe_fwd(x, num_freqs, max_freq_log2, log_sampling, include_input)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        num_freqs, max_freq_log2, log_sampling, include_input = ctx.cfg
        return ops.pe_bwd(x, g.contiguous(), num_freqs, max_freq_log2, log_sampling, include_input), None, None, None, None


class Embedder:
    """kwargs: include_input, input_dims, max_freq_log2, num_freqs, log_sampling, periodic_fns."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        kw = self.kwargs
        d, n = kw['input_dims'], kw['num_freqs']
        fns = list(kw['periodic_fns'])
        self.out_dim = d * ((1 if kw['include_input'] else 0) + n * len(fns))
        # the HIP kernel implements the (sin, cos) pair the reference always uses
        self._native = len(fns) == 2 and fns[0] is torch.sin and fns[1] is torch.cos
        if kw['log_sampling']:
            self.freq_bands = 2. ** torch.linspace(0., kw['max_freq_log2'], steps=n)
        else:
            self.freq_bands = torch.linspace(2. ** 0., 2. ** kw['max_freq_log2'], steps=n)

    def embed(self, inputs):
        kw = self.kwargs
        if self._native and inputs.is_cuda:
            args = (kw['num_freqs'], float(kw['max_freq_log2']), bool(kw['log_sampling']), bool(kw['include_input']))
            if inputs.requires_grad and torch.is_grad_enabled():
                return _PEFunction.apply(inputs, *args)
            return ops.pe_fwd(inputs, *args)
        if not inputs.is_cuda:
            raise RuntimeError("nerf_amd.Embedder.embed: input is on the CPU; nerf_amd has no CPU path "
                               "(torch.set_default_device('cuda') as run_nerf.py:14-15 does)")
        # exotic periodic_fns: plain torch ops on the GPU, in the reference's channel order
        parts = [inputs] if kw['include_input'] else []
        for f in self.freq_bands.to(inputs.device):
            parts += [fn(inputs * f) for fn in kw['periodic_fns']]
        return torch.cat(parts, -1)


class _EmbedFn:
    """Callable returned by get_embedder; carries what the fused query kernel needs to know."""

    def __init__(self, embedder, multires):
        self.embedder = embedder
        self.multires = multires
        self.out_dim = embedder.out_dim

    def __call__(self, x):
        return self.embedder.embed(x)

    @property
    def fusable(self):
        kw = self.embedder.kwargs
        return (self.embedder._native and kw['include_input'] and kw['log_sampling']
                and kw['max_freq_log2'] == kw['num_freqs'] - 1 and kw['num_freqs'] == self.multires)


def get_embedder(multires, i=0):
    if i == -1:
        return nn.Identity(), 3
    embedder = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                        log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return _EmbedFn(embedder, multires), embedder.out_dim
