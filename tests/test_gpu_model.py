"""create_nerf / create_dnerf (model.py:11-107, vanilla_dynamic/model.py:25-186): the keyword sets render() receives,
which parameters the optimizer owns, checkpoint discovery (newest *.tar, --ft_path, best.tar for D-NeRF, --no_reload)
and a resume that restores weights, optimizer state and the step counter."""
import os
import types

import numpy as np
import pytest
import torch

import nerf_fixtures as fx

pytestmark = pytest.mark.gpu


def _args(tmp, **kw):
    a = dict(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=16, N_samples=8, netdepth=2,
             netwidth=64, netdepth_fine=2, netwidth_fine=64, netchunk=65536, lrate=5e-4, ft_path=None, basedir=str(tmp),
             expname="exp", no_reload=False, perturb=1., white_bkgd=True, raw_noise_std=1., dataset_type='blender',
             no_ndc=False, lindisp=False, multires_time=10, network_type='deformation', zero_canonical=True)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_create_nerf_wiring_and_resume(tmp_path):
    from nerf_amd.checkpoint import save_checkpoint
    from nerf_amd.model import create_nerf
    from nerf_amd.render import render
    os.makedirs(tmp_path / "exp")
    train, test, start, grad_vars, opt = create_nerf(_args(tmp_path))
    assert start == 0
    assert set(train) == {'network_query_fn', 'perturb', 'N_importance', 'network_fine', 'N_samples', 'network_fn',
                          'use_viewdirs', 'white_bkgd', 'raw_noise_std', 'ndc', 'lindisp'}
    assert train['perturb'] == 1. and train['raw_noise_std'] == 1. and train['ndc'] is False
    assert test['perturb'] is False and test['raw_noise_std'] == 0. and test['network_fn'] is train['network_fn']
    coarse, fine = train['network_fn'], train['network_fine']
    assert fine is not None and coarse.output_ch == 5
    assert len(grad_vars) == len(list(coarse.parameters())) + len(list(fine.parameters()))
    # forward-facing LLFF keeps render()'s NDC default: no 'ndc' / 'lindisp' keys (model.py:97-101)
    llff, _, _, _, _ = create_nerf(_args(tmp_path, dataset_type='llff'))
    assert 'ndc' not in llff and 'lindisp' not in llff
    single, _, _, gv1, _ = create_nerf(_args(tmp_path, N_importance=0))
    assert single['network_fine'] is None and single['network_fn'].output_ch == 4 and len(gv1) == len(list(coarse.parameters()))
    # one optimizer step, two checkpoints; the newest by name is resumed
    H = W = 6
    K, _ = fx.intrinsics(H, W)
    c2w = torch.from_numpy(fx.pose_spherical(20.0, -30.0, 4.0)[:3, :4]).cuda()
    rgb = render(H, W, K, chunk=64, c2w=c2w, near=2., far=6., **train)[0]
    rgb.square().mean().backward()
    opt.step()
    save_checkpoint(str(tmp_path / "exp" / "000010.tar"), 10, coarse, fine, opt)
    with torch.no_grad():
        coarse.pts_linears[0].bias.add_(1.0)
    save_checkpoint(str(tmp_path / "exp" / "000020.tar"), 20, coarse, fine, opt)
    tr2, _, start2, _, opt2 = create_nerf(_args(tmp_path))
    assert start2 == 20
    for (k, a), (_, b) in zip(coarse.state_dict().items(), tr2['network_fn'].state_dict().items()):
        assert torch.equal(a, b), k
    s1, s2 = opt.state_dict()['state'], opt2.state_dict()['state']
    assert len(s1) == len(s2) and all(torch.equal(s1[i]['exp_avg'], s2[i]['exp_avg']) for i in s1)
    tr3, _, start3, _, _ = create_nerf(_args(tmp_path, ft_path=str(tmp_path / "exp" / "000010.tar")))
    assert start3 == 10 and not torch.equal(tr3['network_fn'].pts_linears[0].bias, coarse.pts_linears[0].bias)
    _, _, start4, _, _ = create_nerf(_args(tmp_path, no_reload=True))
    assert start4 == 0
    _, _, start5, _, _ = create_nerf(_args(tmp_path, expname="missing"))     # warns, starts fresh
    assert start5 == 0
    with torch.no_grad():   # the test set renders deterministically through the same nets
        a = render(H, W, K, chunk=64, c2w=c2w, near=2., far=6., **test)[0]
        b = render(H, W, K, chunk=64, c2w=c2w, near=2., far=6., **test)[0]
    assert torch.equal(a, b)


def test_create_dnerf_wiring_and_best_checkpoint(tmp_path):
    import nerf_amd.vanilla_dynamic as vd
    from nerf_amd.vanilla_dynamic.model import create_dnerf, save_checkpoint
    os.makedirs(tmp_path / "exp")
    args = _args(tmp_path)
    train, test, start, grad_vars, opt = create_dnerf(args)
    assert start == 0 and train['network_type'] == 'deformation' and train['ndc'] is False
    assert test['perturb'] is False and test['raw_noise_std'] == 0.
    model = train['network_fn']
    assert isinstance(model, vd.DNeRF_Deformation) and train['network_fine'] is None     # one model for both passes
    assert model.embed_fn is not None and len(grad_vars) == len(list(model.parameters()))
    two, _, _, gv2, _ = create_dnerf(_args(tmp_path, use_two_models_for_fine=True))
    assert two['network_fine'] is not None and len(gv2) == 2 * len(grad_vars)
    H = W = 5
    K, _ = fx.intrinsics(H, W)
    c2w = torch.from_numpy(fx.pose_spherical(20.0, -30.0, 4.0)[:3, :4]).cuda()
    with torch.no_grad():
        rgb = vd.render_dnerf(H, W, K, 0.4, chunk=64, c2w=c2w, near=2., far=6., **test)[0]
    assert rgb.shape == (H, W, 3) and torch.isfinite(rgb).all()
    save_checkpoint(str(tmp_path / "exp" / "000100.tar"), 100, model, None, opt, args)
    with torch.no_grad():
        model._time_out.bias.add_(0.25)
    save_checkpoint(str(tmp_path / "exp" / "best.tar"), 70, model, None, opt, args)
    tr2, _, start2, _, _ = create_dnerf(_args(tmp_path))
    assert start2 == 70                                                                   # best.tar wins over newer files
    assert torch.equal(tr2['network_fn']._time_out.bias, model._time_out.bias)


@pytest.mark.parametrize("kw", [dict(netwidth=300, netwidth_fine=300), dict(multires=12, multires_views=5),
                                dict(netwidth=300, netwidth_fine=128, use_viewdirs=False)])
def test_architectures_outside_the_kernels_ranges_still_run(tmp_path, kw):
    """The reference's parser allows any --netwidth / --multires (model.py:11-34); the fused kernels take W <= 256
    (zero-padded to whole 32-channel tiles) and multires <= 10.  Outside that, create_nerf's networks run the same arithmetic as a GPU torch
    composition (announced by a warning) instead of refusing: render + a training step work, and a supported net beside
    them still takes the fused route."""
    from nerf_amd.model import create_nerf
    from nerf_amd.render import render
    from nerf_amd.utils import img2mse
    train, test, _, _, opt = create_nerf(_args(tmp_path, **kw))
    K, _ = fx.intrinsics(20, 20)
    rays = torch.from_numpy(fx.blender_rays(48, 5)[:, :6].astype(np.float32)).cuda()
    batch = torch.stack([rays[:, :3], rays[:, 3:6]])
    with pytest.warns(UserWarning, match="GPU torch composition"):
        rgb, disp, acc, extras = render(20, 20, K, chunk=4096, rays=batch, retraw=True, near=2., far=6., **train)
    loss = img2mse(rgb, torch.rand_like(rgb)) + img2mse(extras["rgb0"], torch.rand_like(rgb))
    opt.zero_grad()
    loss.backward()
    nets = [train["network_fn"], train["network_fine"]]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n in nets for p in n.parameters()
               if n.use_viewdirs or not any(p is q for q in n.views_linears.parameters()))
    opt.step()
    with torch.no_grad():
        rgb2, _, _, _ = render(20, 20, K, chunk=4096, rays=batch, near=2., far=6., **test)
    assert torch.isfinite(rgb2).all() and rgb2.shape == (48, 3)
    fused = [n for n in nets if n.kernel_shape_reason() is None]
    if "multires" not in kw:
        assert len(fused) == (1 if kw.get("netwidth_fine") == 128 else 0)
        assert all(n._packed_cache for n in fused), "a supported net left the fused route"
