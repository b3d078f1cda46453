"""nrf_query_bwd_static -- the allocation-free variant of nrf_query_bwd (workspaces in a file-scope device array) --
called through the C ABI and checked against nrf_query_bwd on the same inputs, twice in a row, the second call with a
smaller input (the array keeps the first call's contents).  The two entry points run the same kernels; the static one
splits the wgrad reduction into <= 8 partial slabs instead of <= 256, so the sums differ in order only: the bar is 1e-5
of each gradient's scale (the f32 kernels are held to 2e-3 of the reference autograd's gradients elsewhere)."""
import ctypes as C

import numpy as np
import pytest
import torch

import nerf_fixtures as fx

pytestmark = pytest.mark.gpu


def _grad_desc(params):
    from nerf_amd import _lib
    grads = {k: torch.full_like(v, float("nan")) for k, v in params.items()}
    gn = _lib.NrfNet()
    g = lambda k: C.c_void_p(grads[k].data_ptr())  # noqa: E731
    i = 0
    while f"pts_linears.{i}.weight" in grads:
        gn.pts_w[i], gn.pts_b[i] = g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias")
        i += 1
    if "rgb_linear.weight" in grads:
        gn.views_w, gn.views_b = g("views_linears.0.weight"), g("views_linears.0.bias")
        gn.feature_w, gn.feature_b = g("feature_linear.weight"), g("feature_linear.bias")
        gn.alpha_w, gn.alpha_b = g("alpha_linear.weight"), g("alpha_linear.bias")
        gn.rgb_w, gn.rgb_b = g("rgb_linear.weight"), g("rgb_linear.bias")
    else:
        gn.out_w, gn.out_b = g("output_linear.weight"), g("output_linear.bias")
    return grads, gn


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _both(net, n_rays, seed):
    """(gradients by nrf_query_bwd, gradients by nrf_query_bwd_static) for n_rays x 64 points."""
    from nerf_amd import _lib, ops
    lib = _lib.lib()
    rng = np.random.RandomState(seed)
    P = n_rays * 64
    pts = torch.from_numpy(rng.uniform(-2, 2, (P, 3)).astype(np.float32)).cuda()
    dirs = torch.from_numpy(fx.blender_rays(n_rays, seed)[:, -3:].copy()).cuda()
    g_raw = torch.from_numpy(rng.normal(size=(P, 4)).astype(np.float32)).cuda()
    pk = net._packed((0, 10, 4))
    raw, acts = pk.query_train(pts, dirs, 64)
    # existing entry point, caller-owned workspaces
    ref, gn = _grad_desc(pk.params)
    gbuf = torch.empty((lib.nrf_train_grads_floats(C.byref(pk.net), P),), device="cuda")
    slab = torch.empty((lib.nrf_train_slab_floats(C.byref(pk.net), P),), device="cuda")
    _lib.check(lib.nrf_query_bwd(C.byref(pk.net), C.byref(pk.emb), ops._ptr(pk.packed), ops._ptr(g_raw), ops._ptr(acts),
                                 ops._ptr(gbuf), ops._ptr(slab), P, C.byref(gn), _stream()), "nrf_query_bwd")
    # the allocation-free one: inputs and outputs only
    got, gs = _grad_desc(pk.params)
    torch.cuda.synchronize()
    # two counters: torch's caching allocator (sees only torch) and the DRIVER's free bytes (hipMemGetInfo: a hipMalloc
    # inside the library would show here; the scratch array was resolved by the warm-up call of the test)
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    _lib.check(lib.nrf_query_bwd_static(C.byref(pk.net), C.byref(pk.emb), ops._ptr(pk.packed), ops._ptr(g_raw),
                                        ops._ptr(acts), P, C.byref(gs), _stream()), "nrf_query_bwd_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, (free_before, torch.cuda.mem_get_info()[0])
    return ref, got


def _compare(ref, got):
    for k in ref:
        r, g = ref[k].cpu().numpy(), got[k].cpu().numpy()
        assert np.isfinite(r).all() and np.isfinite(g).all(), k
        scale = max(float(np.abs(r).max()), 1e-12)
        assert float(np.abs(g - r).max()) <= 1e-5 * scale, (k, float(np.abs(g - r).max()), scale)


def test_static_backward_matches_twice():
    from nerf_amd import _lib, ops
    from nerf_amd.network import NeRF
    net = NeRF(8, 256, 63, 27, 5, [4], True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(21, peaky=True).items()})
    net = net.cuda()
    _both(net, 1, 0)                 # warm-up: resolves the scratch symbol on this device (and loads the kernels)
    _compare(*_both(net, 47, 1))     # 3008 points: 94 of the 95 tiles this network may pass (8 slabs of 12 tiles)
    _compare(*_both(net, 5, 2))      # then 320 points: 10 tiles, the scratch still holds the first call's data
    # no view directions, W = 128, a ragged tile
    small = NeRF(4, 128, 63, 0, 4, [2], False)
    sd = fx.make_nerf_state(23, D=4, W=128, input_ch=63, input_ch_views=0, output_ch=4, skips=(2,), use_viewdirs=False)
    small.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    small = small.cuda()
    lib = _lib.lib()
    pk = small._packed((0, 10, 0))
    P = 7 * 64 + 13
    pts = torch.from_numpy(np.random.RandomState(3).uniform(-2, 2, (P, 3)).astype(np.float32)).cuda()
    g_raw = torch.from_numpy(np.random.RandomState(4).normal(size=(P, 4)).astype(np.float32)).cuda()
    raw, acts = pk.query_train(pts, None, 1)
    ref = pk.query_bwd(g_raw, P, acts)
    got, gs = _grad_desc(pk.params)
    _lib.check(lib.nrf_query_bwd_static(C.byref(pk.net), C.byref(pk.emb), ops._ptr(pk.packed), ops._ptr(g_raw),
                                        ops._ptr(acts), P, C.byref(gs), _stream()), "nrf_query_bwd_static")
    _compare(ref, got)


def test_static_backward_rejects_what_does_not_fit():
    from nerf_amd import _lib, ops
    from nerf_amd.network import NeRF
    net = NeRF(8, 256, 63, 27, 5, [4], True).cuda()
    pk = net._packed((0, 10, 4))
    pk.refresh()
    lib = _lib.lib()
    dummy = torch.zeros(16, device="cuda")
    _, gs = _grad_desc(pk.params)
    for P in (4096 * 64, 3041):          # far beyond, and one point past the limit the header documents (3040)
        rc = lib.nrf_query_bwd_static(C.byref(pk.net), C.byref(pk.emb), ops._ptr(pk.packed), ops._ptr(dummy),
                                      ops._ptr(dummy), P, C.byref(gs), C.c_void_p(0))
        assert rc == 2      # NRF_ERR_UNSUPPORTED, nothing launched
        msg = lib.nrf_last_error_string().decode()
        assert "static scratch" in msg and "at most 3040 points" in msg, msg
    # a shape the backward kernels do not take (W beyond 256) is refused as such, whatever its size
    bad = _lib.NrfNet()
    C.memmove(C.byref(bad), C.byref(pk.net), C.sizeof(bad))
    bad.W = 300
    rc = lib.nrf_query_bwd_static(C.byref(bad), C.byref(pk.emb), ops._ptr(pk.packed), ops._ptr(dummy), ops._ptr(dummy), 64,
                                  C.byref(gs), C.c_void_p(0))
    assert rc == 2 and "unsupported network shape" in lib.nrf_last_error_string().decode()
