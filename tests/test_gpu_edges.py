"""Edge cases through the C ABI on the GPU: empty and ragged inputs, one-sample rays, tile-boundary point counts."""
import numpy as np
import pytest
import torch

import nerf_fixtures as fx

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.fixture(scope="module")
def net():
    from nerf_amd.network import NeRF
    n = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(24).items()})
    return n.cuda()


def test_empty_inputs(net):
    from nerf_amd import ops
    rays = torch.zeros((0, 11), device="cuda")
    z = ops.sample_coarse(rays, 64)
    assert z.shape == (0, 64)
    assert ops.points(rays, z).shape == (0, 64, 3)
    rgb, disp, acc, w, depth = ops.composite_fwd(torch.zeros((0, 64, 4), device="cuda"), z, rays[:, 3:6], None, True)
    assert rgb.shape == (0, 3) and w.shape == (0, 64)
    _, zf, zs = ops.sample_fine(z, w, 128, None, want_samples=False)
    assert zf.shape == (0, 192) and zs.shape == (0,)
    with torch.no_grad():
        for prec in ("fp32", "bf16"):
            net.precision = prec
            out = net.query_points(torch.zeros((0, 3), device="cuda"), torch.zeros((0, 3), device="cuda"), 64, 10, 4)
            assert out.shape == (0, 4)
    net.precision = "fp32"


@pytest.mark.parametrize("n_points", [1, 31, 32, 33, 255, 256, 257, 64 * 1024 + 1])
def test_ragged_point_counts(net, n_points):
    """Tile tails (32-point tiles in fp32, 256-point workgroup tiles in bf16) must not disturb the live points:
    the first n points of a longer batch give bit-identical results."""
    rng = np.random.RandomState(n_points)
    pts = dev(rng.uniform(-2, 2, (n_points + 300, 3)))
    d = rng.normal(size=(n_points + 300, 3))
    dirs = dev(d / np.linalg.norm(d, axis=-1, keepdims=True))
    with torch.no_grad():
        for prec in ("fp32", "bf16"):
            net.precision = prec
            full = net.query_points(pts, dirs, 1, 10, 4)
            part = net.query_points(pts[:n_points].contiguous(), dirs[:n_points].contiguous(), 1, 10, 4)
            assert part.shape == (n_points, 4) and torch.equal(part, full[:n_points]), prec
    net.precision = "fp32"


def test_single_sample_and_single_ray():
    from nerf_amd import ops
    rays = dev(fx.blender_rays(1, 3))
    z = ops.sample_coarse(rays, 1)                      # linspace(0, 1, 1) = [0] -> z = near
    assert z.shape == (1, 1) and float(z[0, 0]) == 2.0
    raw = torch.tensor([[[0.3, -0.2, 0.1, 0.7]]], device="cuda")
    rgb, disp, acc, w, depth = ops.composite_fwd(raw, z, rays[:, 3:6], None, False)
    # one sample: dist = 1e10 -> alpha = 1 for sigma > 0 -> weight 1, colour = sigmoid(raw)
    np.testing.assert_allclose(w.cpu().numpy(), [[1.0]], atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy(), torch.sigmoid(raw[0, :, :3]).cpu().numpy(), atol=1e-6)
    assert float(depth[0]) == pytest.approx(2.0, abs=1e-6)


def test_cpu_tensors_are_refused(net):
    from nerf_amd import ops
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.composite_fwd(torch.zeros(2, 4, 4), torch.zeros(2, 4).cuda(), torch.zeros(2, 3).cuda())
    with pytest.raises(RuntimeError, match="GPU tensor"):
        net.query_points(torch.zeros(4, 3), torch.zeros(4, 3).cuda(), 1, 10, 4)


def test_non_finite_rows_still_fill_every_merged_slot():
    """A ray whose coarse weights (or depths) hold NaN / Inf: the resampler's outputs for it are garbage, as in the
    reference, but EVERY slot of its merged row is written (such rows take the sort path; comparisons with NaN are false
    and would otherwise pass the rank-merge's monotonicity checks) and its neighbours are untouched."""
    import ctypes as C
    from nerf_amd import _lib, ops
    rays = dev(fx.blender_rays(6, 5))
    z = ops.sample_coarse(rays, 64)
    rng = np.random.RandomState(2)
    w = dev(np.abs(rng.normal(size=(6, 64))))
    _, ref, _ = ops.sample_fine(z, w, 128, None, want_samples=False)
    w_bad, z_bad = w.clone(), z.clone()
    w_bad[1, 10] = float("nan")
    w_bad[3, 5] = float("inf")
    z_bad[4, 20] = float("nan")
    merged = torch.full((6, 192), 12345.0, device="cuda")
    std = torch.empty(6, device="cuda")
    _lib.check(_lib.lib().nrf_sample_fine(ops._ptr(z_bad), ops._ptr(w_bad), 6, 64, 128, None, None, ops._ptr(merged),
                                          ops._ptr(std), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nrf_sample_fine")
    assert not bool((merged == 12345.0).any())
    for r in (0, 2, 5):
        assert torch.equal(merged[r], ref[r])


# ------------------------------------------------------------------ the documented maxima (supported ranges)
def test_sample_and_composite_at_their_maximum_sizes():
    """N_samples = 257 coarse + 767 fine = 1024 merged samples (the resampler's and the compositor backward's limit), and
    one past it refused: sample_pdf + merge against the oracle bit-for-bit, the compositor against torch autograd."""
    rng = np.random.RandomState(12)
    N, Nc, Nf = 5, 257, 767
    rays = dev(fx.blender_rays(N, 3)[:, :8])
    z = dev(np.sort(rng.uniform(2, 6, (N, Nc)).astype(np.float32), -1))
    w = dev((rng.uniform(0, 1, (N, Nc)).astype(np.float32)) ** 4)
    u = dev(rng.uniform(0, 1, (N, Nf)).astype(np.float32))
    import oracle.oracle as orc
    from nerf_amd import ops
    host = lambda t: t.detach().cpu().numpy()  # noqa: E731
    zs, zm, sd = ops.sample_fine(z, w, Nf, u)
    zs_o, zm_o, sd_o = orc.sample_fine(host(z), host(w), Nf, u=host(u))
    assert np.array_equal(host(zs), zs_o) and np.array_equal(host(zm), zm_o)
    np.testing.assert_allclose(host(sd), sd_o, rtol=1e-6)
    with pytest.raises(RuntimeError):
        ops.sample_fine(z, w, Nf + 1, None)
    S = 1024
    raw = dev(rng.normal(size=(N, S, 4)).astype(np.float32)).requires_grad_(True)
    from nerf_amd.render import raw2outputs
    rgb, disp, acc, weights, depth = raw2outputs(raw, zm, rays[:, 3:6], 0.0, True)
    (rgb.sum() + acc.sum() + (weights * weights).sum()).backward()
    r64 = raw.detach().double().requires_grad_(True)
    d = torch.cat([zm[:, 1:] - zm[:, :-1], torch.full_like(zm[:, :1], 1e10)], -1).double() * rays[:, 3:6].double().norm(dim=-1, keepdim=True)
    al = 1 - torch.exp(-torch.relu(r64[..., 3]) * d)
    T = torch.cumprod(torch.cat([torch.ones_like(al[:, :1]), 1 - al + 1e-10], -1), -1)[:, :-1]
    w64 = al * T
    rgb64 = (w64[..., None] * torch.sigmoid(r64[..., :3])).sum(-2)
    acc64 = w64.sum(-1)
    rgb64 = rgb64 + (1 - acc64[..., None])
    (rgb64.sum() + acc64.sum() + (w64 * w64).sum()).backward()
    assert float((raw.grad.double() - r64.grad).abs().max()) <= 2e-5 * float(r64.grad.abs().max())


def test_widest_heads_and_embedders():
    """output_ch = 8 without view directions, multires = 10 / multires_views = 4 (the encoders' maxima), embedded rows of
    64 + 32 channels: the fused query (forward, and parameter gradients) against the oracle / the torch composition."""
    from nerf_amd.network import NeRF
    torch.manual_seed(13)
    rng = np.random.RandomState(13)
    n = 3 * 37
    pts = dev(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
    net = NeRF(D=3, W=64, input_ch=63, input_ch_views=0, output_ch=8, skips=[1], use_viewdirs=False).cuda()
    raw = net.query_points(pts, None, 1, 10, 0)
    assert raw.shape == (n, 8)
    g = dev(rng.normal(size=(n, 8)).astype(np.float32))
    (raw * g).sum().backward()
    net64 = NeRF(D=3, W=64, input_ch=63, input_ch_views=0, output_ch=8, skips=[1], use_viewdirs=False)
    net64.load_state_dict(net.state_dict())
    net64 = net64.cuda().double()
    x = pts.double()
    xin = torch.cat([x] + [f(x * 2.0 ** j) for j in range(10) for f in (torch.sin, torch.cos)], -1)
    ref = net64._forward_autograd(xin)
    assert float((raw.detach().double() - ref.detach()).abs().max()) <= 2e-5 * float(ref.detach().abs().max())
    (ref * g.double()).sum().backward()
    for (k, p), (_, q) in zip(net.named_parameters(), net64.named_parameters()):
        if q.grad is None:
            continue
        assert float((p.grad.double() - q.grad).norm()) <= 2e-4 * float(q.grad.norm()), k
    with pytest.raises(RuntimeError):          # one past the head's width
        NeRF(D=3, W=64, input_ch=63, input_ch_views=0, output_ch=9, skips=[1], use_viewdirs=False).cuda().query_points(
            pts, None, 1, 10, 0)
    # already-embedded rows at the widest the kernel takes: 64 position + 32 direction channels
    wide = NeRF(D=2, W=128, input_ch=64, input_ch_views=32, skips=[0], use_viewdirs=True).cuda()
    rows = dev(rng.normal(size=(n, 96)).astype(np.float32))
    with torch.no_grad():
        a = wide(rows)
        b = wide._forward_autograd(rows)
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # one channel more: forward() composes GPU torch ops instead (announced), the explicit fused call refuses
    over = NeRF(D=2, W=128, input_ch=66, input_ch_views=32, skips=[0], use_viewdirs=True).cuda()
    with torch.no_grad(), pytest.warns(UserWarning, match="GPU torch composition"):
        o = over(dev(rng.normal(size=(4, 98))))
    assert o.shape == (4, 4) and not over._packed_cache
