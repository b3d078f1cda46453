"""Hierarchical sampling -- counterpart of the reference's sampling.py:5-60, on the HIP kernel
nrf_sample_pdf (bit-exact with the reference's CPU path on identical inputs)."""
import numpy as np
import torch

from . import ops


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    u = None
    lead = list(bins.shape[:-1])
    if pytest:
        # the reference's fixed-random route (sampling.py:32-40): numpy f64 values rounded to f32
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0., 1., N_samples), lead + [N_samples])
        else:
            u = np.random.rand(*(lead + [N_samples]))
        u = torch.tensor(np.ascontiguousarray(u), dtype=torch.float32, device=bins.device)
    elif not det:
        u = torch.rand(lead + [N_samples], device=bins.device)
    return ops.sample_pdf(bins, weights, N_samples, u)
