"""Corners of the reference's call surface that the parity suites do not pass through: the Keras weight import
(network.py:71-99), batchify (render.py:14-28), the straightforward D-NeRF network class, D-NeRF checkpoint helpers
(vanilla_dynamic/model.py:189-240), create_dnerf with network_type='straightforward'."""
import types

import numpy as np
import pytest
import torch

import nerf_fixtures as fx

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def test_load_weights_from_keras_equals_state_dict():
    """The flat Keras list is [kernel, bias] per layer with kernels [in, out]: pts_linears 0..D-1, feature, views,
    rgb, alpha (network.py:76-99).  Loading it must give the network load_state_dict gives."""
    from nerf_amd.network import NeRF
    sd = fx.make_nerf_state(41)
    order = [f"pts_linears.{i}" for i in range(8)] + ["feature_linear", "views_linears.0", "rgb_linear", "alpha_linear"]
    keras = []
    for name in order:
        keras += [np.ascontiguousarray(sd[name + ".weight"].T), sd[name + ".bias"].copy()]
    a = NeRF(8, 256, 63, 27, 4, [4], True)
    a.load_weights_from_keras(keras)
    a = a.cuda()
    b = NeRF(8, 256, 63, 27, 4, [4], True)
    b.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    b = b.cuda()
    for (k, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p, q), k
    pts = dev(np.random.RandomState(1).uniform(-2, 2, (300, 3)))
    dirs = dev(fx.blender_rays(3, 1)[:, -3:])
    with torch.no_grad():
        assert torch.equal(a.query_points(pts, dirs, 100, 10, 4), b.query_points(pts, dirs, 100, 10, 4))
    c = NeRF(8, 256, 63, 0, 4, [4], False)
    with pytest.raises(AssertionError, match="use_viewdirs"):                 # network.py:72
        c.load_weights_from_keras(keras)


def test_batchify_slices_and_concatenates():
    from nerf_amd.render import batchify
    calls = []

    def fn(x):
        calls.append(x.shape[0])
        return x * 2.0
    x = torch.arange(23, dtype=torch.float32, device="cuda")[:, None].repeat(1, 3)
    assert batchify(fn, None) is fn
    out = batchify(fn, 10)(x)
    assert calls == [10, 10, 3] and torch.equal(out, x * 2.0)


def test_straightforward_network_and_dnerf_checkpoint_helpers(tmp_path):
    import nerf_amd.vanilla_dynamic as vd
    from nerf_amd.vanilla_dynamic.model import create_dnerf, load_checkpoint, save_checkpoint
    args = types.SimpleNamespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=8, N_samples=8, netdepth=3,
        netwidth=64, netdepth_fine=3, netwidth_fine=64, netchunk=65536, lrate=5e-4, ft_path=None,
        basedir=str(tmp_path), expname="s", no_reload=False, perturb=0., white_bkgd=True, raw_noise_std=0.,
        dataset_type='blender', no_ndc=False, lindisp=False, multires_time=10, network_type='straightforward',
        zero_canonical=True)
    train, _, _, grad_vars, opt = create_dnerf(args)
    coarse, fine = train['network_fn'], train['network_fine']
    assert isinstance(coarse, vd.DNeRF_Straightforward) and isinstance(fine, vd.DNeRF_Straightforward)   # model.py:93-99
    assert coarse.pts_linears[0].in_features == 63 + 21 and len(grad_vars) == 2 * len(list(coarse.parameters()))
    x = torch.randn(17, 63 + 27, device="cuda")
    te = torch.randn(17, 21, device="cuda")
    out, dx = coarse(x, [te, te])
    assert out.shape == (17, 4) and dx.shape == (17, 3) and float(dx.abs().max()) == 0.0
    # checkpoint helpers: what save_checkpoint writes, load_checkpoint restores (weights, optimizer, step, args)
    out.sum().backward()
    opt.step()
    path = str(tmp_path / "ck.tar")
    save_checkpoint(path, 1234, coarse, fine, opt, args)
    keys = set(torch.load(path, weights_only=True).keys())
    assert keys == {'global_step', 'network_fn_state_dict', 'network_fine_state_dict', 'optimizer_state_dict', 'args'}
    other, _, _, _, opt2 = create_dnerf(args)
    step, saved_args = load_checkpoint(path, other['network_fn'], other['network_fine'], opt2)
    assert step == 1234 and saved_args['network_type'] == 'straightforward'
    for (k, p), (_, q) in zip(coarse.state_dict().items(), other['network_fn'].state_dict().items()):
        assert torch.equal(p, q), k


def test_documented_ctypes_binding():
    """INTEGRATION.md section 3, as written there: a plain ctypes call of one entry point with raw device pointers,
    against the package's own wrapper; and the error path (status code + nrf_last_error_string)."""
    import ctypes as C
    import os
    from nerf_amd import ops
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_amd", "libnrf.so"))
    lib.nrf_composite_fwd.restype = C.c_int
    lib.nrf_last_error_string.restype = C.c_char_p
    rng = np.random.RandomState(2)
    N, S = 37, 50
    raw, rays_d = dev(rng.normal(size=(N, S, 4))), dev(rng.normal(size=(N, 3)))
    z_vals = dev(np.sort(rng.uniform(2, 6, (N, S)), -1))
    rgb = torch.empty(N, 3, device="cuda")
    disp, acc, depth = (torch.empty(N, device="cuda") for _ in range(3))
    weights = torch.empty(N, S, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = lib.nrf_composite_fwd(p(raw), p(z_vals), p(rays_d), 3, C.c_int64(N), S, None, 1,
                               p(rgb), p(disp), p(acc), p(weights), p(depth), st)
    assert rc == 0, lib.nrf_last_error_string().decode()
    want = ops.composite_fwd(raw, z_vals, rays_d, None, True)
    for got, w in zip((rgb, disp, acc, weights, depth), want):
        assert torch.equal(got, w)
    rc = lib.nrf_composite_fwd(p(raw), p(z_vals), p(rays_d), 3, C.c_int64(N), 0, None, 1,
                               p(rgb), p(disp), p(acc), p(weights), p(depth), st)
    assert rc != 0 and b"nrf_composite_fwd" in lib.nrf_last_error_string()
