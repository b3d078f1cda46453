"""Fused Adam (nrf_adam_step) against torch.optim.Adam on the reference's configuration (model.py:52)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 63), (256,), (256, 256), (1, 256), (3, 128), (5,), (1000, 37)]
    return [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]


@pytest.mark.parametrize("clip", [None, 0.5])
def test_matches_torch_adam(clip):
    from nerf_amd.optim import FusedAdam
    a, b = make_params(0), make_params(0)
    ref = torch.optim.Adam(a, lr=5e-4, betas=(0.9, 0.999))
    opt = FusedAdam(b, lr=5e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(1)
    for step in range(25):
        lr = 5e-4 * 0.1 ** (step / 250.0)              # train.py:247-251
        ref.param_groups[0]["lr"] = lr
        opt.param_groups[0]["lr"] = lr
        for pa, pb in zip(a, b):
            gr = (torch.randn(pa.shape, generator=g) * (10.0 if step % 7 == 0 else 0.1)).cuda()
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if step == 3:
            a[4].grad = None                            # a parameter without a gradient is skipped, like torch
            b[4].grad = None
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(a, clip)
        ref.step()
        opt.step(max_grad_norm=clip)
    for pa, pb in zip(a, b):
        np.testing.assert_allclose(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_state_dict_round_trip_with_torch():
    from nerf_amd.optim import FusedAdam
    a, b = make_params(2), make_params(2)
    ref = torch.optim.Adam(a, lr=1e-3)
    opt = FusedAdam(b, lr=1e-3)
    for pa, pb in zip(a, b):
        pa.grad = torch.ones_like(pa)
        pb.grad = torch.ones_like(pb)
    ref.step()
    opt.step()
    # ours -> torch and torch -> ours
    ref2 = torch.optim.Adam(a, lr=7.0)
    ref2.load_state_dict(opt.state_dict())
    assert ref2.param_groups[0]["lr"] == 1e-3
    opt2 = FusedAdam(b, lr=7.0)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.steps == [1] * len(b) and opt2.param_groups[0]["lr"] == 1e-3
    for pa, pb in zip(a, b):
        pa.grad = torch.full_like(pa, 0.5)
        pb.grad = torch.full_like(pb, 0.5)
    ref2.step()
    opt2.step()
    for pa, pb in zip(a, b):
        np.testing.assert_allclose(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_step_invalidates_packed_weights():
    """The fused step writes parameters through raw pointers; the networks' packed copies must notice."""
    from nerf_amd.network import NeRF
    from nerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True).cuda()
    opt = FusedAdam(list(net.parameters()), lr=1e-2)
    x = torch.rand(256, 90, device="cuda") * 2 - 1
    before = [p._version for p in net.parameters()]
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = (net(x) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(p._version > v for p, v in zip(net.parameters(), before))
    assert losses[1] != losses[0] and losses[-1] < losses[0]
