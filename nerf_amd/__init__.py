"""nerf_amd: MI355X-native implementation of the duudual/NeRF per-ray rendering path.

Python modules here mirror the reference's call signatures (positional_encoding, network, sampling,
render, rays, utils, model; vanilla_dynamic/*) and run on hand-written HIP kernels in libnrf.so
through the C ABI of include/nrf.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
