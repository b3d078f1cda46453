"""Counterpart of vanilla_dynamic/positional_encoding.py: get_embedder(multires, i, input_dims, include_input)
:41-68 and get_time_embedder(multires_time, include_input) :71-97.  The Embedder itself is shared with the
static stack (same class in the reference, copy for copy)."""
import torch
import torch.nn as nn

from ..positional_encoding import Embedder, _EmbedFn


def get_embedder(multires, i=0, input_dims=3, include_input=True):
    if i == -1:
        return nn.Identity(), input_dims
    emb = Embedder(include_input=include_input, input_dims=input_dims, max_freq_log2=multires - 1,
                   num_freqs=multires, log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return _EmbedFn(emb, multires), emb.out_dim


class _IdentityTime:
    multires = 0
    out_dim = 1
    fusable = False

    def __call__(self, x):
        return x


def get_time_embedder(multires_time, include_input=True):
    if multires_time <= 0:
        return _IdentityTime(), 1
    emb = Embedder(include_input=include_input, input_dims=1, max_freq_log2=multires_time - 1,
                   num_freqs=multires_time, log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return _EmbedFn(emb, multires_time), emb.out_dim
