"""CPU-only checks of the C ABI: libnrf.so loads without a GPU, exports every symbol include/nrf.h
declares, its host-side planners answer size queries, and argument validation fails loudly before any
launch.  (No compute calls here: those are the `-m gpu` tests.)"""
import ctypes as C
import os
import re

import pytest

from nerf_amd import _lib
from nerf_amd._lib import NrfEmbed, NrfNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nrf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nrf.h but not exported by libnrf.so"
    # and the Python binding knows each of them
    assert sorted(_lib.exported_symbols()) == syms


def net(D=8, W=256, ic=63, icv=27, oc=4, uv=1, skips=(4,)):
    n = NrfNet()
    n.D, n.W, n.input_ch, n.input_ch_views, n.output_ch, n.use_viewdirs = D, W, ic, icv, oc, uv
    n.skips_mask = sum(1 << s for s in skips)
    return n


def test_version_and_size_queries():
    l = _lib.lib()
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "nrf.h")).read()
    assert l.nrf_version() == _lib.ABI_VERSION == int(re.search(r"#define NRF_VERSION (\d+)", hdr).group(1))
    # the ctypes mirror of nrf_net_t: 8 int32, 2 x 16 + 10 pointers, 2 int32, 2 x 16 + 2 pointers (include/nrf.h)
    assert C.sizeof(_lib.NrfNet) == 8 * 4 + (32 + 10) * 8 + 2 * 4 + (32 + 2) * 8
    n, e = net(), NrfEmbed(0, 10, 4, 0)
    floats = l.nrf_mlp_packed_floats(C.byref(n), C.byref(e))
    # forward blocks (593 920 padded MACs' worth of weights + biases/heads) + transposed hidden blocks for dgrad
    assert floats > 595844 and floats < 3 * 595844
    pts = 4096 * 192
    per_pt = l.nrf_train_acts_floats(C.byref(n), pts) / pts
    assert per_pt == 64 + 32 + 24 + 8 * 256 + 256 + 128 + 8 * 9      # k-space inputs + trunk + feature + view layer
    assert l.nrf_train_grads_floats(C.byref(n), pts) / pts == 8 * 256 + 288 + 128 + 32 + 64
    assert l.nrf_train_slab_floats(C.byref(n), pts) == 256 * 11 * (288 * 352 + 288) + 64   # every GEMM its own slab region (+ the w16 scalar)
    # ragged point counts round up to whole 32-point tiles
    assert l.nrf_train_acts_floats(C.byref(n), 33) == 2 * 32 * per_pt


@pytest.mark.parametrize("kw,msg", [
    (dict(W=1), "W must be"),
    (dict(W=512), "W must be"),
    (dict(D=17), "D must be"),
    (dict(ic=60), "input_ch == 3"),
    (dict(icv=30), "input_ch_views == 3"),
    (dict(skips=(7,)), "skip at the last"),
    (dict(uv=0, oc=9), "output_ch"),
])
def test_unsupported_shapes_are_reported(kw, msg):
    l = _lib.lib()
    n, e = net(**kw), NrfEmbed(0, 10, 4, 0)
    assert l.nrf_mlp_packed_floats(C.byref(n), C.byref(e)) <= 0
    assert msg in l.nrf_last_error_string().decode()


def test_any_width_up_to_256_packs_in_whole_tiles():
    """--netwidth takes any value (config.py:20-27): a width that is not a multiple of 32 is zero-padded to whole 32-channel
    tiles, so its packed weights and workspaces have the next multiple's sizes."""
    l = _lib.lib()
    e = NrfEmbed(0, 10, 4, 0)
    for W, Wp in ((100, 128), (33, 64), (250, 256), (2, 32)):
        a, b = net(W=W), net(W=Wp)
        assert l.nrf_mlp_packed_floats(C.byref(a), C.byref(e)) == l.nrf_mlp_packed_floats(C.byref(b), C.byref(e)) > 0
        for f in (l.nrf_train_acts_floats, l.nrf_train_grads_floats, l.nrf_train_slab_floats):
            assert f(C.byref(a), 1000) == f(C.byref(b), 1000) > 0


def test_null_and_bad_arguments_fail_before_launch():
    l = _lib.lib()
    null = C.c_void_p(0)
    assert l.nrf_sample_coarse(null, 10, 11, 64, 0, null, null, null) == 1
    assert "null" in l.nrf_last_error_string().decode()
    assert l.nrf_composite_fwd(null, null, null, 3, 1, 64, null, 0, null, null, null, null, null, null) == 1
    assert l.nrf_sample_fine(null, null, 1, 64, 128, null, null, null, null, null) == 1
    assert l.nrf_pe_fwd(null, 1, 3, 10, 9.0, 1, 1, null, null) == 1
    n, e = net(), NrfEmbed(0, 10, 4, 0)
    assert l.nrf_query_fwd(C.byref(n), C.byref(e), null, null, null, 32, 1, null, null) == 1
    assert l.nrf_mlp_pack_weights(C.byref(n), C.byref(e), null, null) == 1


def test_python_surface_refuses_cpu_tensors():
    import torch
    from nerf_amd import ops
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render_rays
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.sample_coarse(torch.zeros(4, 8), 8)
    emb, dim = get_embedder(10, 0)
    assert dim == 63 and emb.fusable
    with pytest.raises(RuntimeError, match="no CPU path"):
        emb(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        render_rays(torch.zeros(4, 11), None, None, 64)


def test_state_dict_contract():
    """Checkpoint compatibility: parameter names / shapes of network.py:29-44."""
    import nerf_fixtures as fx
    from nerf_amd.network import NeRF
    n = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    ref = fx.make_nerf_state(0)
    sd = n.state_dict()
    assert list(sd) == list(ref) or sorted(sd) == sorted(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == v.shape, k
    assert sum(v.numel() for v in sd.values()) == 595844
    n2 = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=4, skips=[4], use_viewdirs=False)
    assert "output_linear.weight" in n2.state_dict() and "views_linears.0.weight" in n2.state_dict()
