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
