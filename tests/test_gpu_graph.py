"""RenderGraph: a whole-frame render() captured once as a HIP graph and replayed per pose (the frames of
render_path share every shape).  The bar is bit-identity with render() -- the graph replays the very same kernels."""
import numpy as np
import pytest
import torch

import nerf_fixtures as fx

pytestmark = pytest.mark.gpu


def _setup(precision="fp32"):
    from nerf_amd.network import NeRF
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import run_network
    nets = []
    for seed in (21, 22):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(seed, peaky=True).items()})
        net.precision = precision
        nets.append(net.cuda())
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
              network_fine=nets[1], N_samples=64, network_fn=nets[0], use_viewdirs=True, white_bkgd=True,
              raw_noise_std=0., ndc=False, lindisp=False, near=2., far=6.)
    return nets, kw


def _pose(a):
    return torch.from_numpy(fx.pose_spherical(a, -30.0, 4.0)[:3, :4]).cuda()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_graph_replay_is_render_bit_for_bit(precision):
    from nerf_amd.render import RenderGraph, render
    nets, kw = _setup(precision)
    H, W = 40, 56                       # 2240 rays, chunk 1000: two full chunks and a ragged one inside the graph
    K, _ = fx.intrinsics(H, W)
    frame = RenderGraph(H, W, K, chunk=1000, **kw)
    with torch.no_grad():
        for a in (0.0, 77.0, -140.0):
            rgb, disp, acc, extras = frame(_pose(a))
            want = render(H, W, K, chunk=1000, c2w=_pose(a), **kw)
            assert torch.equal(rgb, want[0]) and torch.equal(disp, want[1]) and torch.equal(acc, want[2])
            assert torch.equal(extras["rgb0"], want[3]["rgb0"]) and torch.equal(extras["z_std"], want[3]["z_std"])
        # parameters changed after the capture: the replay reads the refreshed packed weights
        before = rgb.clone()
        with torch.no_grad():
            nets[1].rgb_linear.bias.add_(0.5)
        rgb2 = frame(_pose(-140.0))[0]
        want2 = render(H, W, K, chunk=1000, c2w=_pose(-140.0), **kw)[0]
        assert torch.equal(rgb2, want2) and not torch.equal(rgb2, before)
    with pytest.raises(ValueError, match="perturb"):
        RenderGraph(H, W, K, chunk=1000, **dict(kw, perturb=1.0))


def test_render_path_with_graph_equals_without():
    from nerf_amd.render import render_path
    _, kw = _setup()
    H, W = 24, 32
    K, focal = fx.intrinsics(H, W)
    poses = torch.stack([torch.from_numpy(fx.pose_spherical(a, -30.0, 4.0)).cuda() for a in (0.0, 120.0, 240.0, 300.0)])
    with torch.no_grad():
        a_rgb, a_disp = render_path(poses, [H, W, focal], K, 512, kw)
        b_rgb, b_disp = render_path(poses, [H, W, focal], K, 512, kw, graph=True)
    assert np.array_equal(a_rgb, b_rgb) and np.array_equal(a_disp, b_disp)
