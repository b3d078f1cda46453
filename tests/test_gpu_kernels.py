"""Parity of each HIP kernel (called through the C ABI via nerf_amd.ops) with the oracle and the golden
vectors.  Bars: sample positions BIT-EXACT; network / compositing values to the stated tolerances
(fp32 MFMA sums are a different-but-exact-f32 summation order from the CPU's)."""
import numpy as np
import pytest
import torch

import nerf_fixtures as fx
from conftest import assert_bit_exact
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from nerf_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("name,lindisp", [("blender", False), ("ndc", False), ("lindisp", True)])
def test_sample_coarse_bit_exact(ops, gold, name, lindisp):
    rays = np.concatenate([gold[f"zc_{name}_o"], gold[f"zc_{name}_d"], gold[f"zc_{name}_near"],
                           gold[f"zc_{name}_far"]], -1)
    z = ops.sample_coarse(dev(rays), 64, lindisp)
    assert_bit_exact(host(z), gold[f"zc_{name}_z"], "z_vals vs reference")
    pts = ops.points(dev(rays), z)
    assert_bit_exact(host(pts), gold[f"zc_{name}_pts"], "pts vs reference")


@pytest.mark.parametrize("S", [1, 2, 7, 64, 65, 192])
def test_sample_coarse_sizes(ops, S):
    rays = fx.blender_rays(301, 17)
    z = ops.sample_coarse(dev(rays), S)
    zo, po = orc.sample_coarse(rays, S) if S > 1 else (np.full((301, 1), 2.0, np.float32), None)
    assert_bit_exact(host(z), zo, f"S={S}")
    if S > 1:
        assert_bit_exact(host(ops.points(dev(rays), z)), po, "pts")


def test_sample_coarse_perturb(ops):
    # stratified jitter with externally supplied randoms (the pytest=True route, render.py:203-206)
    rng = np.random.RandomState(3)
    rays = fx.blender_rays(100, 5)
    t_rand = rng.rand(100, 64).astype(np.float32)
    z = host(ops.sample_coarse(dev(rays), 64, False, dev(t_rand)))
    z0, _ = orc.sample_coarse(rays, 64)
    mids = (.5 * (z0[:, 1:] + z0[:, :-1])).astype(np.float32)
    upper = np.concatenate([mids, z0[:, -1:]], -1)
    lower = np.concatenate([z0[:, :1], mids], -1)
    ref = (lower + ((upper - lower).astype(np.float32) * t_rand).astype(np.float32)).astype(np.float32)
    assert_bit_exact(z, ref, "perturbed z_vals")


# ------------------------------------------------------------------ K5
@pytest.mark.parametrize("name", ["blender_p1", "blender_p6", "ndc_p30"])
@pytest.mark.parametrize("nimp", [128, 64])
def test_sample_fine_bit_exact_vs_reference(ops, gold, name, nimp):
    z, w = gold[f"spdf_{name}_z"], gold[f"spdf_{name}_w"]
    zs, zm, sd = ops.sample_fine(dev(z), dev(w), nimp)
    assert_bit_exact(host(zs), gold[f"spdf_{name}_n{nimp}_zs"], "z_samples vs reference")
    assert_bit_exact(host(zm), gold[f"spdf_{name}_n{nimp}_zm"], "merged z_vals vs reference")
    np.testing.assert_allclose(host(sd), gold[f"spdf_{name}_n{nimp}_std"], rtol=1e-5, atol=1e-7)
    mids = (.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    out = ops.sample_pdf(dev(mids), dev(w[:, 1:-1]), nimp)
    assert_bit_exact(host(out), gold[f"spdf_{name}_n{nimp}_zs"], "sample_pdf vs reference")


@pytest.mark.parametrize("S,nimp", [(3, 1), (8, 5), (64, 128), (65, 100), (128, 256), (200, 300)])
def test_sample_fine_sizes_and_random_u(ops, S, nimp):
    rng = np.random.RandomState(S * 1000 + nimp)
    n = 203   # not a multiple of the 4 rays a block handles
    z = np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), -1)
    w = (rng.rand(n, S) ** 4).astype(np.float32)
    w[0] = 0
    zs, zm, sd = ops.sample_fine(dev(z), dev(w), nimp)
    zs_o, zm_o, sd_o = orc.sample_fine(z, w, nimp)
    assert_bit_exact(host(zs), zs_o, "z_samples")
    assert_bit_exact(host(zm), zm_o, "merged")
    np.testing.assert_allclose(host(sd), sd_o, rtol=1e-5, atol=1e-7)
    # random (unsorted) u: the merged list then needs the full sort path
    u = rng.rand(n, nimp).astype(np.float32)
    zs, zm, _ = ops.sample_fine(dev(z), dev(w), nimp, u=dev(u))
    zs_o, zm_o, _ = orc.sample_fine(z, w, nimp, u=u)
    assert_bit_exact(host(zs), zs_o, "z_samples (random u)")
    assert_bit_exact(host(zm), zm_o, "merged (random u)")


def test_sample_fine_sortedness_property(ops):
    # full-size property check: 800x800 rays worth of rows would be 640k; a 64k slice keeps it quick
    rng = np.random.RandomState(9)
    n = 65536
    z = np.broadcast_to(orc.sample_coarse(fx.blender_rays(1, 1), 64)[0], (n, 64)).copy()
    w = (rng.rand(n, 64) ** 8).astype(np.float32)
    _, zm, _ = ops.sample_fine(dev(z), dev(w), 128)
    zm = host(zm)
    assert (np.diff(zm, axis=-1) >= 0).all()
    assert zm.min() >= 2.0 and zm.max() <= 6.0
    # every coarse depth survives in the merged list
    assert all(np.isin(z[i], zm[i]).all() for i in range(0, n, 4099))


# ------------------------------------------------------------------ A1
def test_pe(ops, gold):
    x = gold["pe_x"]
    np.testing.assert_allclose(host(ops.pe_fwd(dev(x), 10, 9)), gold["pe10"], atol=3e-7, rtol=0)
    np.testing.assert_allclose(host(ops.pe_fwd(dev(x), 4, 3)), gold["pe4"], atol=3e-7, rtol=0)
    np.testing.assert_allclose(host(ops.pe_fwd(dev(gold["pe2d_x"]), 6, 5)), gold["pe2d_L6"], atol=3e-7, rtol=0)
    np.testing.assert_allclose(host(ops.pe_fwd(dev(x[:64]), 5, 3, log_sampling=False, include_input=False)),
                               gold["pe_lin_noinput"], atol=2e-6, rtol=0)


# ------------------------------------------------------------------ K4 / K4b
@pytest.mark.parametrize("name,wb", [("s64_white", True), ("s192", False), ("s7", True)])
def test_composite_vs_reference(ops, gold, name, wb):
    g = lambda k: gold[f"r2o_{name}_{k}"]  # noqa: E731
    rgb, disp, acc, w, depth = (host(t) for t in ops.composite_fwd(dev(g("raw")), dev(g("z")), dev(g("d")),
                                                                  white_bkgd=wb))
    np.testing.assert_allclose(w, g("w"), atol=5e-7, rtol=1e-5)
    np.testing.assert_allclose(rgb, g("rgb"), atol=3e-6, rtol=0)
    np.testing.assert_allclose(acc, g("acc"), atol=3e-6, rtol=0)
    np.testing.assert_allclose(depth, g("depth"), atol=2e-5, rtol=0)
    ok = np.isfinite(g("disp"))
    assert (np.isnan(disp) == np.isnan(g("disp"))).all()
    np.testing.assert_allclose(disp[ok], g("disp")[ok], rtol=2e-5)
    graw = host(ops.composite_bwd(dev(g("raw")), dev(g("z")), dev(g("d")), dev(g("grgb")), white_bkgd=wb))
    ref = g("graw")
    np.testing.assert_allclose(graw, ref, atol=1e-6 + 2e-5 * np.abs(ref).max(), rtol=2e-4)


def test_composite_strided_dirs_and_noise(ops):
    rng = np.random.RandomState(2)
    rays = fx.blender_rays(777, 3)
    S = 100
    raw = (rng.normal(size=(777, S, 4)) * [1, 1, 1, 3]).astype(np.float32)
    z = np.sort(rng.uniform(2, 6, (777, S)).astype(np.float32), -1)
    noise = rng.normal(size=(777, S)).astype(np.float32)
    rays_t = dev(rays)
    out = ops.composite_fwd(dev(raw), dev(z), rays_t[:, 3:6], noise=dev(noise), white_bkgd=True)
    ref = orc.raw2outputs(raw, z, rays[:, 3:6], True, noise=noise)
    for a, b, tol in zip(out, ref, (3e-6, None, 3e-6, 5e-7, 2e-5)):
        if tol is not None:
            np.testing.assert_allclose(host(a), b, atol=tol, rtol=1e-5)


def test_composite_bwd_all_grads_vs_torch(ops):
    # every upstream gradient path (rgb, acc, depth, weights) against autograd of a plain fp32 torch restatement
    rng = np.random.RandomState(4)
    n, S = 129, 192
    raw = dev((rng.normal(size=(n, S, 4)) * [1, 1, 1, 2]).astype(np.float32)).requires_grad_(True)
    z = dev(np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), -1))
    d = dev(rng.normal(size=(n, 3)).astype(np.float32))
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * torch.norm(d[:, None, :], dim=-1)
    alpha = 1. - torch.exp(-torch.relu(raw[..., 3]) * dists)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    rgb = (w[..., None] * torch.sigmoid(raw[..., :3])).sum(-2)
    depth, acc = (w * z).sum(-1), w.sum(-1)
    rgb = rgb + (1. - acc[..., None])
    gr, ga, gd, gw = (dev(rng.normal(size=s).astype(np.float32)) for s in ((n, 3), (n,), (n,), (n, S)))
    ((rgb * gr).sum() + (acc * ga).sum() + (depth * gd).sum() + (w * gw).sum()).backward()
    g = ops.composite_bwd(raw.detach(), z, d, gr, g_acc=ga, g_depth=gd, g_weights=gw, white_bkgd=True)
    ref = host(raw.grad)
    np.testing.assert_allclose(host(g), ref, atol=1e-6 + 2e-5 * np.abs(ref).max(), rtol=5e-4)


# ------------------------------------------------------------------ K2+K3
def _packed(ops, sd, mode=0, **kw):
    cfg = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,), use_viewdirs=True)
    cfg.update(kw)
    params = {k: dev(v) for k, v in sd.items()}
    L = (cfg["input_ch"] - 3) // 6
    Lv = (cfg["input_ch_views"] - 3) // 6 if cfg["use_viewdirs"] else 0
    return ops.PackedNet(params, mode=mode, multires=L, multires_views=Lv, **cfg)


@pytest.mark.parametrize("name,seed,peaky", [("default", 11, False), ("peaky", 12, True)])
def test_query_vs_reference(ops, gold, name, seed, peaky):
    net = _packed(ops, fx.make_nerf_state(seed, peaky=peaky))
    ref = gold[f"mlp_out_{name}"]
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    # fused positional encoding: one direction per point
    raw = host(net.query(dev(gold["mlp_pts"]), dev(gold["mlp_dirs"]), pts_per_ray=1))
    np.testing.assert_allclose(raw, ref, atol=tol, rtol=0)
    # embedded-input mode (NeRF.forward(x), network.py:46-47)
    net1 = _packed(ops, fx.make_nerf_state(seed, peaky=peaky), mode=1)
    raw1 = host(net1.query(dev(gold["mlp_x"])))
    np.testing.assert_allclose(raw1, ref, atol=tol, rtol=0)


def test_query_noview_small(ops, gold):
    kw = dict(D=4, W=128, input_ch=63, input_ch_views=0, output_ch=5, skips=(2,), use_viewdirs=False)
    net = _packed(ops, fx.make_nerf_state(13, **kw), **kw)
    out = host(net.query(dev(gold["mlp_pts"])))
    np.testing.assert_allclose(out, gold["mlp_out_noview"], atol=2e-5, rtol=0)
    net1 = _packed(ops, fx.make_nerf_state(13, **kw), mode=1, **kw)
    np.testing.assert_allclose(host(net1.query(dev(gold["mlp_x"][:, :63]))), gold["mlp_out_noview"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("W,D,skips", [(64, 3, (1,)), (128, 8, (4,)), (256, 2, ()), (256, 10, (2, 6))])
def test_query_shapes_vs_oracle(ops, W, D, skips):
    rng = np.random.RandomState(W + D)
    n_rays, S = 37, 11      # 407 points: ragged last tile, several points per direction
    pts = rng.uniform(-3, 3, (n_rays * S, 3)).astype(np.float32)
    dirs = rng.normal(size=(n_rays, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    sd = fx.make_nerf_state(5, D=D, W=W, skips=skips)
    net = _packed(ops, sd, D=D, W=W, skips=skips)
    raw = host(net.query(dev(pts), dev(dirs), pts_per_ray=S))
    onet = orc.Net(sd, D=D, W=W, skips=skips)
    ref = orc.run_network(pts.reshape(n_rays, S, 3), dirs, onet).reshape(-1, 4)
    np.testing.assert_allclose(raw, ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=0)


def test_query_repack_after_update(ops, gold):
    sd = fx.make_nerf_state(11)
    net = _packed(ops, sd)
    a = host(net.query(dev(gold["mlp_pts"]), dev(gold["mlp_dirs"])))
    with torch.no_grad():
        net.params["rgb_linear.bias"].add_(1.0)       # in-place update bumps the version counter
    b = host(net.query(dev(gold["mlp_pts"]), dev(gold["mlp_dirs"])))
    np.testing.assert_allclose(b[:, :3], a[:, :3] + 1.0, atol=1e-6)
    np.testing.assert_allclose(b[:, 3], a[:, 3], atol=0)


def test_unsupported_shapes_fail_loudly(ops):
    sd = fx.make_nerf_state(1, W=300, D=2, skips=())
    with pytest.raises(RuntimeError, match="W must be"):
        _packed(ops, sd, W=300, D=2, skips=())
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.sample_coarse(torch.zeros(4, 8), 8)


# ------------------------------------------------------------------ K3b (training forward + backward)
def _grad_check(ops, sd, pts, dirs, S, g_out, **kw):
    cfg = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,), use_viewdirs=True)
    cfg.update(kw)
    net = _packed(ops, sd, **kw)
    raw_ref = net.query(dev(pts), dev(dirs) if dirs is not None else None, pts_per_ray=S)
    raw, acts = net.query_train(dev(pts), dev(dirs) if dirs is not None else None, pts_per_ray=S)
    assert torch.equal(raw, raw_ref), "training-mode forward must equal the inference forward bit for bit"
    grads = net.query_bwd(dev(g_out), pts.shape[0], acts)
    onet = orc.Net(sd, **cfg)
    e = orc.embed(pts, 10)
    if dirs is not None:
        e = np.concatenate([e, orc.embed(np.repeat(dirs, S, axis=0), 4)], -1)
    # saved activations: decode the T8 workspace ([4][C][8] per 32-point tile) and check it against the oracle
    D, W, P = cfg["D"], cfg["W"], pts.shape[0]
    a = host(acts).reshape(-1, 32 * (64 + 32 + 24 + D * W + W + W // 2) + 256 * (D + 1))   # + ReLU gate bits

    def section(off, C):
        t = a[:, off:off + 32 * C].reshape(-1, 4, C, 8)                 # tile, m, channel, q7
        return t.transpose(0, 1, 3, 2).reshape(-1, C)[:P]              # [P, C]
    base = 32 * (64 + 32 + 24)
    trunk = [section(base + 32 * W * l, W) for l in range(D)]
    hv = section(base + 32 * W * D + 32 * W, W // 2)
    x64 = e.astype(np.float64)
    h = x64[:, :63]
    for l in range(D):
        y = np.maximum(h @ sd[f"pts_linears.{l}.weight"].astype(np.float64).T + sd[f"pts_linears.{l}.bias"], 0)
        np.testing.assert_allclose(trunk[l], y, atol=2e-5 * max(1.0, np.abs(y).max()), rtol=0,
                                   err_msg=f"saved activation of trunk layer {l}")
        h = np.concatenate([x64[:, :63], y], -1) if l in cfg["skips"] else y
    # ReLU gates exactly as the GPU saw them (a pre-activation within rounding of zero may gate differently
    # in f32 and in the oracle's f64)
    masks = {"trunk": [t > 0 for t in trunk], "hv": hv > 0}
    _, ref = onet.forward_backward(e, g_out, masks=masks)
    for k, g in ref.items():
        got = host(grads[k]).astype(np.float64)
        scale = np.abs(g).max() + 1e-30
        err = np.abs(got - g).max() / scale
        assert err < 2e-4, (k, err, scale)
    return grads


def test_query_bwd_vs_oracle(ops):
    rng = np.random.RandomState(7)
    n_rays, S = 13, 31      # 403 points: ragged last tile
    pts = rng.uniform(-3, 3, (n_rays * S, 3)).astype(np.float32)
    dirs = rng.normal(size=(n_rays, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    g_out = rng.normal(size=(n_rays * S, 4)).astype(np.float32)
    _grad_check(ops, fx.make_nerf_state(31, peaky=True), pts, dirs, S, g_out)


@pytest.mark.parametrize("W,D,skips", [(64, 3, (1,)), (128, 4, (2,))])
def test_query_bwd_shapes(ops, W, D, skips):
    rng = np.random.RandomState(W)
    n_rays, S = 9, 50
    pts = rng.uniform(-3, 3, (n_rays * S, 3)).astype(np.float32)
    dirs = rng.normal(size=(n_rays, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    g_out = rng.normal(size=(n_rays * S, 4)).astype(np.float32)
    _grad_check(ops, fx.make_nerf_state(32, D=D, W=W, skips=skips), pts, dirs, S, g_out, D=D, W=W, skips=skips)


def test_query_bwd_many_chunks(ops):
    # > 256 tiles so that chunks hold several tiles each
    rng = np.random.RandomState(11)
    n_rays, S = 300, 64
    pts = rng.uniform(-3, 3, (n_rays * S, 3)).astype(np.float32)
    dirs = rng.normal(size=(n_rays, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    g_out = (rng.normal(size=(n_rays * S, 4)) * (rng.rand(n_rays * S, 1) < 0.3)).astype(np.float32)
    _grad_check(ops, fx.make_nerf_state(33), pts, dirs, S, g_out)
