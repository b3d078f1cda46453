"""Seeded random architectures / sizes through the fused query (all three precisions) and the full render_rays chain,
against the oracle: depths, widths, skip positions, encoder sizes, with and without view directions, ragged point
counts (the fixed cases elsewhere pin the reference's own configurations; these look for what they do not reach)."""
import numpy as np
import pytest
import torch

import nerf_fixtures as fx
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    D = int(rng.randint(2, 9))
    W = int(rng.choice([64, 128, 256]))
    L, Lv = int(rng.choice([4, 6, 10])), int(rng.choice([2, 4]))
    views = bool(rng.randint(0, 2))
    skip = int(rng.randint(0, D - 1))             # (a skip at the last trunk layer breaks the reference's head fan-in)
    cfg = dict(D=D, W=W, input_ch=3 + 6 * L, input_ch_views=(3 + 6 * Lv) if views else 0, output_ch=int(rng.choice([4, 5])),
               skips=(skip,), use_viewdirs=views)
    n_rays, S = int(rng.randint(1, 40)), int(rng.randint(1, 70))
    return rng, cfg, L, Lv, n_rays, S


@pytest.mark.parametrize("seed", range(16))
def test_random_net_query_all_precisions(seed):
    from nerf_amd.network import NeRF
    rng, cfg, L, Lv, n_rays, S = _case(seed)
    sd = fx.make_nerf_state(500 + seed, **cfg)
    net = NeRF(D=cfg["D"], W=cfg["W"], input_ch=cfg["input_ch"], input_ch_views=cfg["input_ch_views"],
               output_ch=cfg["output_ch"], skips=list(cfg["skips"]), use_viewdirs=cfg["use_viewdirs"])
    if not cfg["use_viewdirs"]:
        sd = {k: v for k, v in sd.items() if k in net.state_dict()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.cuda()
    pts = rng.uniform(-2.5, 2.5, (n_rays, S, 3)).astype(np.float32)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    full = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    onet = orc.Net(full, **cfg)
    want = orc.run_network(pts, d if cfg["use_viewdirs"] else None, onet, L, Lv).reshape(n_rays * S, -1)
    scale = max(np.abs(want).max(), 1e-6)
    q = lambda: net.query_points(dev(pts.reshape(-1, 3)), dev(d) if cfg["use_viewdirs"] else None, S, L, Lv)  # noqa: E731
    with torch.no_grad():
        got = q().cpu().numpy()
    assert got.shape == want.shape, (cfg, got.shape, want.shape)
    assert np.abs(got - want).max() <= 2e-5 * scale, (cfg, n_rays, S, np.abs(got - want).max() / scale)
    if cfg["W"] in (128, 256):
        for prec, bar in (("bf16x3", 5e-5), ("bf16", 3e-2)):
            net.precision = prec
            with torch.no_grad():
                g2 = q().cpu().numpy()
            assert np.abs(g2 - want).max() <= bar * scale, (prec, cfg, n_rays, S, np.abs(g2 - want).max() / scale)


@pytest.mark.parametrize("views", [True, False])
@pytest.mark.parametrize("W", [32, 96, 160, 192, 224, 100, 65, 250, 6])
def test_every_width_up_to_256(W, views):
    """--netwidth takes any value (config.py:20-27); every width up to 256 runs on the fused f32 kernels in whole 32-channel
    tiles -- zero-padded in the packed weights when W is not a multiple of 32 (W = 100, 65, 250, 6), as the view layer's
    W/2 outputs are when W/2 is not (W = 96, 160, 224): the query on
    raw points and on embedded rows against the oracle (2e-5 of scale), every parameter gradient against float64 autograd
    (2e-3 of its scale), bitwise repeatable, and NeRF._composed (the GPU torch composition) is never reached."""
    from nerf_amd.network import NeRF
    rng = np.random.RandomState(W + views)
    D, L, Lv = 5, 10, 4
    cfg = dict(D=D, W=W, input_ch=3 + 6 * L, input_ch_views=(3 + 6 * Lv) if views else 0, output_ch=4, skips=(2,),
               use_viewdirs=views)
    sd = fx.make_nerf_state(800 + W, **cfg)
    net = NeRF(D=D, W=W, input_ch=cfg["input_ch"], input_ch_views=cfg["input_ch_views"], output_ch=4, skips=[2],
               use_viewdirs=views)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in net.state_dict()}, strict=False)
    net = net.cuda()
    assert net.kernel_shape_reason() is None
    net._composed = None                       # reaching the composition would raise
    n_rays, S = 23, 37                         # 851 points: ragged tiles
    pts = rng.uniform(-2.5, 2.5, (n_rays, S, 3)).astype(np.float32)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    full = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = orc.run_network(pts, d if views else None, orc.Net(full, **cfg), L, Lv).reshape(n_rays * S, -1)
    scale = max(np.abs(want).max(), 1e-6)
    P, dirs = dev(pts.reshape(-1, 3)), dev(d)
    with torch.no_grad():
        got = net.query_points(P, dirs if views else None, S, L, Lv)
        # the same through forward() on embedded rows (embed mode 1: what the generic run_network route calls)
        from nerf_amd.positional_encoding import get_embedder
        e, _ = get_embedder(L, 0)
        rows = e(P)
        if views:
            ev, _ = get_embedder(Lv, 0)
            rows = torch.cat([rows, ev(dirs.repeat_interleave(S, 0))], -1)
        got_rows = net(rows)
    assert np.abs(host(got) - want).max() <= 2e-5 * scale, np.abs(host(got) - want).max() / scale
    assert np.abs(host(got_rows) - want).max() <= 2e-5 * scale
    # training: forward with saved activations, dgrad, wgrad, reduction
    out = net.query_points(P, dirs if views else None, S, L, Lv)
    assert np.abs(host(out) - want).max() <= 2e-5 * scale
    g = dev(rng.normal(size=tuple(out.shape)))
    (out * g).sum().backward()
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in net.state_dict(keep_vars=True).items()}
    ref = _torch_forward64(sd64, cfg, P.double(), L, Lv, dirs.double().repeat_interleave(S, 0))
    (ref * g.double()).sum().backward()
    used = [k for k in sd64 if views or not k.startswith("views_linears")]
    params = dict(net.named_parameters())
    for k in used:
        gw, ww = params[k].grad, sd64[k].grad
        assert gw is not None and gw.shape == ww.shape, k
        err = float((gw.double() - ww).abs().max()) / (float(ww.abs().max()) + 1e-30)
        assert err < 2e-3, (k, W, err)
    first = {k: params[k].grad.clone() for k in used}
    net.zero_grad()
    (net.query_points(P, dirs if views else None, S, L, Lv) * g).sum().backward()
    assert all(torch.equal(first[k], params[k].grad) for k in used)
    # gradients w.r.t. the query points (nrf_query_bwd_input: the GX instantiation of the dgrad chain)
    Pg = P.clone().requires_grad_(True)
    (net.query_points(Pg, dirs if views else None, S, L, Lv) * g).sum().backward()
    P64 = P.double().requires_grad_(True)
    sd64b = {k: v.detach().double() for k, v in net.state_dict().items()}
    (_torch_forward64(sd64b, cfg, P64, L, Lv, dirs.double().repeat_interleave(S, 0)) * g.double()).sum().backward()
    err = float((Pg.grad.double() - P64.grad).norm() / P64.grad.norm())
    assert err < 1e-3, err


@pytest.mark.parametrize("seed", range(8))
def test_random_render_rays_vs_oracle(seed):
    """Whole chain (coarse sampling, query, composite, inverse-CDF resampling, fine pass) at random sample counts, with
    well-conditioned nets: positions of the coarse pass bit-exact by construction, rendered values to 1e-4."""
    from nerf_amd.network import NeRF
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render_rays, run_network
    rng = np.random.RandomState(2000 + seed)
    Nc, Nf = int(rng.randint(3, 65)), int(rng.choice([0, 1, 17, 64, 128]))   # resampling needs an interior coarse weight
    n_rays = int(rng.randint(1, 300))
    white = bool(rng.randint(0, 2))
    sds = [fx.make_nerf_state(600 + 2 * seed), fx.make_nerf_state(601 + 2 * seed)]
    nets = []
    for sd in sds:
        net = NeRF(8, 256, 63, 27, 4, [4], True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net.cuda())
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    rays = fx.blender_rays(n_rays, 50 + seed)
    with torch.no_grad():
        ret = render_rays(dev(rays), nets[0], lambda i, v, f: run_network(i, v, f, e10, e4), Nc, N_importance=Nf,
                          network_fine=nets[1], white_bkgd=white)
    ref = orc.render_rays(rays, orc.Net(sds[0]), orc.Net(sds[1]), Nc, Nf, False, white)
    for k in ("rgb_map", "acc_map") + (("rgb0", "acc0") if Nf > 0 else ()):
        err = np.abs(ret[k].cpu().numpy() - ref[k]).max()
        assert err < 1e-4, (k, Nc, Nf, n_rays, white, err)


def _torch_forward64(sd, cfg, x, L, Lv, d=None):
    """network.py:46-69 with plain torch ops in float64 (the encodings of positional_encoding.py:24-33 included)."""
    def embed(v, n):
        out = [v]
        for k in range(n):
            out += [torch.sin(v * 2.0 ** k), torch.cos(v * 2.0 ** k)]
        return torch.cat(out, -1)
    lin = lambda h, name: torch.nn.functional.linear(h, sd[name + ".weight"], sd[name + ".bias"])  # noqa: E731
    xp = embed(x, L)
    h = xp
    for i in range(cfg["D"]):
        h = torch.relu(lin(h, f"pts_linears.{i}"))
        if i in cfg["skips"]:
            h = torch.cat([xp, h], -1)
    if cfg["use_viewdirs"]:
        alpha = lin(h, "alpha_linear")
        h = torch.cat([lin(h, "feature_linear"), embed(d, Lv)], -1)
        h = torch.relu(lin(h, "views_linears.0"))
        return torch.cat([lin(h, "rgb_linear"), alpha], -1)
    return lin(h, "output_linear")


@pytest.mark.parametrize("train_precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("seed", range(10))
def test_random_net_parameter_gradients(seed, train_precision):
    """Training forward + dgrad + wgrad + slab reduction on random architectures and ragged point counts against torch
    autograd of the same network in float64: every parameter's gradient within 2e-3 of its scale -- in the f32 kernels
    and on the bf16x3 arithmetic (W = 64 nets have no bf16x3 kernels: there the f32 ones run either way)."""
    from nerf_amd.network import NeRF
    rng, cfg, L, Lv, n_rays, S = _case(100 + seed)
    n_rays = n_rays * int(rng.choice([1, 7]))          # up to ~19 000 points: several wgrad chunks
    sd = fx.make_nerf_state(700 + seed, **cfg)
    net = NeRF(D=cfg["D"], W=cfg["W"], input_ch=cfg["input_ch"], input_ch_views=cfg["input_ch_views"],
               output_ch=cfg["output_ch"], skips=list(cfg["skips"]), use_viewdirs=cfg["use_viewdirs"])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in net.state_dict()}, strict=False)
    net = net.cuda()
    net.train_precision = train_precision
    pts = dev(rng.uniform(-2.5, 2.5, (n_rays * S, 3)))
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    dirs = dev(d / np.linalg.norm(d, axis=-1, keepdims=True))
    out = net.query_points(pts, dirs if cfg["use_viewdirs"] else None, S, L, Lv)
    g = dev(rng.normal(size=tuple(out.shape)))
    (out * g).sum().backward()
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in net.state_dict(keep_vars=True).items()}
    ref = _torch_forward64(sd64, cfg, pts.double(), L, Lv, dirs.double().repeat_interleave(S, 0))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(),
                               atol=2e-5 * max(float(ref.detach().abs().max()), 1e-6), rtol=0)
    (ref * g.double()).sum().backward()
    used = [k for k in sd64 if cfg["use_viewdirs"] or not k.startswith("views_linears")]
    worst = 0.0
    for k in used:
        got, want = dict(net.named_parameters())[k].grad, sd64[k].grad
        assert got is not None and want is not None, k
        scale = float(want.abs().max()) + 1e-30
        err = float((got.double() - want).abs().max()) / scale
        if train_precision == "fp32":
            assert err < 2e-3, (k, cfg, n_rays * S, err)
        else:
            # bf16x3 carries 16 mantissa bits: a pre-activation within ~1e-5 of zero gates differently than in float64 --
            # about ten of a case's ~1 M gates -- and the GRADIENT is discontinuous there: one flipped gate toggles one
            # point's term in one row of dW, which on these small point counts (600 - 19 000) is up to ~5 % of that row's
            # elements (measured per case: 3e-7 where no gate flips ... 3e-2).  So this arithmetic is held in the
            # Frobenius norm here (a flipped row is 1/W of the matrix); element-wise it is held to 5e-5 on identical gates
            # and to 2e-3 against the reference's autograd on the 8 x 256 nets / 16 k points (tests/test_gpu_train.py).
            err = float((got.double() - want).norm() / (want.norm() + 1e-30))
            assert err < 1e-2, (k, cfg, n_rays * S, err)
        worst = max(worst, err)
    print(f"gradient fuzz {seed} {train_precision}: worst error {worst:.2e} (fp32: max element / scale; bf16x3: Frobenius)")
    # the same pass again gives the same bits (slab reduction in a fixed order, no atomics; and nothing timing-dependent
    # such as the 16-byte store hazard of DESIGN.md section 9 in any instantiation these cases reach)
    first = {k: dict(net.named_parameters())[k].grad.clone() for k in used}
    net.zero_grad()
    (net.query_points(pts, dirs if cfg["use_viewdirs"] else None, S, L, Lv) * g).sum().backward()
    for k in used:
        assert torch.equal(first[k], dict(net.named_parameters())[k].grad), (k, cfg, "second pass differs")


@pytest.mark.parametrize("seed", range(6))
def test_random_deformation_chain_vs_oracle(seed):
    """DNeRF_Deformation at random depth / width / skip, random times (one ray at t = 0 somewhere), random netchunk --
    the per-netchunk t == 0 shortcut included -- against the oracle's chain: dx and raw in f32, bf16x3 at its bar."""
    import nerf_amd.vanilla_dynamic as vd
    rng = np.random.RandomState(3000 + seed)
    D, W = int(rng.randint(2, 9)), int(rng.choice([128, 256]))
    skip = int(rng.randint(0, D - 1))
    n_rays, S = int(rng.randint(2, 12)), int(rng.randint(1, 50))
    sd = fx.make_dnerf_state(800 + seed, D=D, W=W, D_deform=D, W_deform=W, skips=(skip,), dx_scale=1.0)
    model = vd.DNeRF_Deformation(D=D, W=W, D_deform=D, W_deform=W, input_ch=63, input_ch_views=27, input_ch_time=21,
                                 output_ch=4, skips=[skip], use_viewdirs=True, zero_canonical=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    embed_fn, _ = vd.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = vd.get_embedder(4, 0, input_dims=3)
    embed_time, _ = vd.get_time_embedder(10)
    model.embed_fn = embed_fn
    pts = rng.uniform(-1.5, 1.5, (n_rays, S, 3)).astype(np.float32)
    dirs = fx.blender_rays(n_rays, 60 + seed)[:, -3:]
    times = rng.uniform(0.05, 1.0, (n_rays, 1)).astype(np.float32)
    times[int(rng.randint(0, n_rays)), 0] = 0.0
    netchunk = int(rng.choice([S, 3 * S, 1 << 16]))      # whole rays per chunk, as render_rays_dnerf produces them
    want_raw, want_dx = orc.run_network_dnerf(pts, dirs, times, orc.Net(sd, D=D, W=W, skips=(skip,), prefix="_occ."),
                                              orc.TimeNet(sd, D=D, W=W, skips=(skip,)), netchunk=netchunk)
    run = lambda: vd.run_network_dnerf(dev(pts), dev(dirs), dev(times), model, embed_fn, embeddirs_fn, embed_time,  # noqa: E731
                                       netchunk=netchunk)
    with torch.no_grad():
        raw, dx = run()
        model.precision = "bf16x3"
        raw3, dx3 = run()
    sdx, sraw = max(np.abs(want_dx).max(), 1e-3), np.abs(want_raw).max()
    assert np.abs(dx.cpu().numpy() - want_dx).max() < 2e-5 * sdx
    # raw sees x + dx through the frequency-512 encoding: f32-rounding-sized differences of dx are multiplied
    assert np.abs(raw.cpu().numpy() - want_raw).max() < 2e-4 * sraw, np.abs(raw.cpu().numpy() - want_raw).max() / sraw
    assert np.abs(dx3.cpu().numpy() - want_dx).max() < 5e-5 * sdx
    assert np.abs(raw3.cpu().numpy() - want_raw).max() < 1e-3 * sraw
    zero_rows = (np.abs(want_dx).max(axis=(1, 2)) == 0)
    assert np.array_equal(np.abs(dx.cpu().numpy()).max(axis=(1, 2)) == 0, zero_rows)


@pytest.mark.parametrize("seed", range(12))
def test_random_sampling_and_compositing_sizes(seed):
    """Inverse-CDF resampling + merge (bit-exact) and the compositor (forward 3e-6, backward vs the oracle) at random
    sample counts, with both instantiations of the resampling kernel, empty / spiky / zero weights, given and default
    u."""
    from nerf_amd import ops
    from conftest import bits
    rng = np.random.RandomState(4000 + seed)
    S = int(rng.randint(3, 200))
    Ni = int(rng.randint(1, min(400, 1024 - S)))
    N = int(rng.randint(1, 200))
    near, far = (0.0, 1.0) if rng.randint(0, 2) else (2.0, 6.0)
    rays = fx.blender_rays(N, 70 + seed)
    rays[:, 6], rays[:, 7] = near, far
    z = ops.sample_coarse(dev(rays), S).cpu().numpy()
    w = rng.uniform(0, 1, (N, S)).astype(np.float32) ** float(rng.choice([1, 6, 30]))
    w[0] = 0.0                                           # all-zero weights: the uniform pdf of the 1e-5 floor
    if N > 1:
        w[1] = 0.0
        w[1, S // 2] = 1.0                               # one spike
    u = rng.uniform(0, 1, (N, Ni)).astype(np.float32) if rng.randint(0, 2) else None
    zs, zm, sd = ops.sample_fine(dev(z), dev(w), Ni, dev(u) if u is not None else None)
    o_zs, o_zm, o_sd = orc.sample_fine(z, w, Ni, u)
    np.testing.assert_array_equal(bits(zs.cpu().numpy()), bits(o_zs), err_msg=f"z_samples S={S} Ni={Ni} N={N}")
    np.testing.assert_array_equal(bits(zm.cpu().numpy()), bits(o_zm), err_msg=f"z_merged S={S} Ni={Ni} N={N}")
    np.testing.assert_allclose(sd.cpu().numpy(), o_sd, atol=1e-6, rtol=1e-5)
    # compositor on the merged depths
    St = S + Ni
    raw = (rng.normal(size=(N, St, 4)) * np.array([1, 1, 1, 4.0])).astype(np.float32)
    white = bool(rng.randint(0, 2))
    got = ops.composite_fwd(dev(raw), zm, dev(rays[:, 3:6]), None, white)
    want = orc.raw2outputs(raw, o_zm, rays[:, 3:6], white)
    for g, w_, name in zip(got, want, ("rgb", "disp", "acc", "weights", "depth")):
        if name == "disp":
            continue                                     # 1 / max(1e-10, depth / acc): unbounded where acc -> 0
        np.testing.assert_allclose(g.cpu().numpy(), w_, atol=3e-6 * max(1.0, float(np.abs(w_).max())), rtol=0, err_msg=name)
    g_rgb = rng.normal(size=(N, 3)).astype(np.float32)
    g_raw = ops.composite_bwd(dev(raw), zm, dev(rays[:, 3:6]), dev(g_rgb), white_bkgd=white)
    o_graw = orc.raw2outputs_bwd(raw, o_zm, rays[:, 3:6], g_rgb, white)
    np.testing.assert_allclose(g_raw.cpu().numpy(), o_graw, atol=2e-5 * max(1.0, float(np.abs(o_graw).max())), rtol=0)


def test_encoder_arguments_beyond_the_fast_range():
    """The in-kernel sincos is a short-range routine up to |x * 2^k| = 16384 and the library's beyond (decided per wave):
    coordinates of +-100 at L = 10 (arguments up to 51 200) -- alone, and mixed with ordinary points -- against the
    oracle, f32 and bf16x3; the deformation chain with a far point as well."""
    from nerf_amd.network import NeRF
    sd = fx.make_nerf_state(24)
    net = NeRF(8, 256, 63, 27, 4, [4], True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda()
    rng = np.random.RandomState(9)
    far = rng.uniform(-100, 100, (6, 40, 3)).astype(np.float32)
    mixed = far.copy()
    mixed[::2] = rng.uniform(-2, 2, mixed[::2].shape)
    mixed[1, 3] = [16384.0 / 512, -16384.0 / 512, 32.000004]      # at and just past the switch-over
    d = fx.blender_rays(6, 5)[:, -3:]
    onet = orc.Net(sd)
    for pts in (far, mixed):
        want = orc.run_network(pts, d, onet).reshape(-1, 4)
        scale = np.abs(want).max()
        with torch.no_grad():
            net.precision = "fp32"
            got = net.query_points(dev(pts.reshape(-1, 3)), dev(d), 40, 10, 4).cpu().numpy()
            net.precision = "bf16x3"
            got3 = net.query_points(dev(pts.reshape(-1, 3)), dev(d), 40, 10, 4).cpu().numpy()
        assert np.abs(got - want).max() < 2e-5 * scale, np.abs(got - want).max() / scale
        assert np.abs(got3 - want).max() < 5e-5 * scale, np.abs(got3 - want).max() / scale


@pytest.mark.parametrize("W,D", [(64, 2), (128, 4)])
def test_deformation_chain_with_unequal_widths(W, D):
    """create_dnerf builds the deformation net at 8 x 256 whatever --netwidth is: the chain then runs as two launches
    (time net, canonical net) instead of the fused kernel -- same results against the oracle, t = 0 shortcut included;
    the bf16 modes, which only exist fused, say so."""
    import nerf_amd.vanilla_dynamic as vd
    rng = np.random.RandomState(77 + W)
    sd = fx.make_dnerf_state(900 + W, D=D, W=W, D_deform=8, W_deform=256, skips=(4,), dx_scale=1.0)
    model = vd.DNeRF_Deformation(D=D, W=W, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27, input_ch_time=21,
                                 output_ch=4, skips=[4], use_viewdirs=True, zero_canonical=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    embed_fn, _ = vd.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = vd.get_embedder(4, 0, input_dims=3)
    embed_time, _ = vd.get_time_embedder(10)
    model.embed_fn = embed_fn
    n_rays, S = 5, 33
    pts = rng.uniform(-1.5, 1.5, (n_rays, S, 3)).astype(np.float32)
    dirs = fx.blender_rays(n_rays, 8)[:, -3:]
    times = np.array([[0.6], [0.0], [0.2], [0.9], [0.35]], np.float32)
    run = lambda nc: vd.run_network_dnerf(dev(pts), dev(dirs), dev(times), model, embed_fn, embeddirs_fn, embed_time,  # noqa: E731
                                          netchunk=nc)
    for nc in (S, 1 << 16):
        want_raw, want_dx = orc.run_network_dnerf(pts, dirs, times, orc.Net(sd, D=D, W=W, skips=(4,), prefix="_occ."),
                                                  orc.TimeNet(sd), netchunk=nc)
        with torch.no_grad():
            raw, dx = run(nc)
        assert np.abs(dx.cpu().numpy() - want_dx).max() < 2e-5 * max(np.abs(want_dx).max(), 1e-3)
        assert np.abs(raw.cpu().numpy() - want_raw).max() < 2e-4 * np.abs(want_raw).max()
    # the time-net half runs in slices of whole rays through one reused workspace (bounded memory at a 32768-ray chunk):
    # slices of 2 rays (ragged last slice) must give the unsliced result bit for bit, per-ray and per-point times alike
    from nerf_amd.vanilla_dynamic import network as vnet
    with torch.no_grad():
        whole = [t.clone() for t in run(S)]
        per_point = [t.clone() for t in model.query_points(dev(pts.reshape(-1, 3)), dev(dirs), dev(np.repeat(times, S)), S, True,
                                                            S, 10, 4, 10)]
        cap, vnet._TWO_LAUNCH_POINTS = vnet._TWO_LAUNCH_POINTS, 2 * S + 5
        try:
            sliced = run(S)
            sliced_pp = model.query_points(dev(pts.reshape(-1, 3)), dev(dirs), dev(np.repeat(times, S)), S, True, S, 10, 4, 10)
        finally:
            vnet._TWO_LAUNCH_POINTS = cap
    for a, b in zip(whole + per_point, list(sliced) + list(sliced_pp)):
        assert torch.equal(a.reshape(b.shape), b)
    model.precision = "bf16"
    with torch.no_grad(), pytest.raises(ValueError, match="same width"):
        run(S)
