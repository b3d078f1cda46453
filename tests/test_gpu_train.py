"""Training step on the GPU (train.py:226-243: render -> img2mse(rgb)+img2mse(rgb0) -> backward) against
the gradients the reference's autograd produced (tests/golden/static.npz, `train_*`)."""
import os

import numpy as np
import pytest
import torch

import nerf_fixtures as fx
from conftest import assert_mostly_close

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def grad_summary(g):
    g = g.detach().double().cpu().numpy().reshape(-1)
    idx = np.linspace(0, g.size - 1, min(256, g.size)).astype(np.int64)
    return np.concatenate([[g.sum(), np.sqrt((g * g).sum())], g[idx]])


def make_nets(train_precision="fp32"):
    from nerf_amd.network import NeRF
    nets = []
    for seed in (21, 22):
        net = NeRF(8, 256, 63, 27, 5, [4], True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(seed, peaky=True).items()})
        net.train_precision = train_precision
        nets.append(net.cuda())
    return nets


# the training kernels in the reference's f32 arithmetic, and on the bf16x3 arithmetic (three bf16 MFMAs per product on
# hi/residual operands): both against the SAME reference autograd goldens and the same bars
# "bf16x3_a16": bf16x3 with the saved activations in f16 (half the bytes of the step's largest workspace)
# "bf16x3_w16": that plus the gradient blocks as scaled f16, weight-gradient GEMMs as single f16 MFMAs
TRAIN_PRECISIONS = ["fp32", "bf16x3", "bf16x3_a16", "bf16x3_w16"]


def check_grads(net, gold, tag, tol):
    for k, p in net.named_parameters():
        ref = gold[f"train_grad_{tag}_{k}"]
        assert p.grad is not None, k
        got = grad_summary(p.grad)
        scale = max(ref[1], 1e-12)
        assert abs(got[1] - ref[1]) <= tol * scale, (tag, k, got[1], ref[1])
        np.testing.assert_allclose(got[2:], ref[2:], atol=tol * scale / np.sqrt(max(1, p.numel() / 64)) + 1e-9,
                                   rtol=5 * tol, err_msg=f"{tag} {k}")


@pytest.mark.parametrize("train_precision", TRAIN_PRECISIONS)
def test_train_step_through_render(gold, train_precision):
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render, run_network
    from nerf_amd.utils import img2mse
    coarse, fine = make_nets(train_precision)
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
              network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True,
              raw_noise_std=0., ndc=False, lindisp=False)
    K, _ = fx.intrinsics(400, 400)
    batch_rays = torch.stack([dev(gold["train_ro"]), dev(gold["train_rd"])], 0)
    target = dev(gold["train_target"])
    rgb, disp, acc, extras = render(400, 400, K, chunk=32768, rays=batch_rays, retraw=True, near=2., far=6., **kw)
    assert extras["raw"].shape == (64, 192, 4) and "rgb0" in extras
    loss = img2mse(rgb, target) + img2mse(extras["rgb0"], target)
    loss.backward()
    np.testing.assert_allclose(loss.item(), gold["train_loss"][0], rtol=2e-4)
    np.testing.assert_allclose(extras["rgb0"].detach().cpu().numpy(), gold["train_rgb0"], atol=1e-4, rtol=0)
    if train_precision == "fp32":
        assert_mostly_close(rgb.detach().cpu().numpy(), gold["train_rgb"], what="rgb")      # measured: all within 7.9e-5
    else:
        # sharp test nets (sigma head x30): the 1e-5-relative error of the bf16x3 coarse pass moves fine samples on a few
        # rays (measured: 85 % of the rays within 1e-4, worst 4.1e-3, PSNR 64 dB); well-conditioned nets meet 1e-4 on
        # every ray in this arithmetic too (tests/test_gpu_wellcond.py, tests/test_gpu_bf16.py)
        assert_mostly_close(rgb.detach().cpu().numpy(), gold["train_rgb"], what="rgb", frac=0.8, hard=1e-2, psnr_min=58.0)
    check_grads(coarse, gold, "coarse", 2e-3)   # the coarse pass sees exactly the reference's sample positions
    # fine pass end to end: its own resampled positions (sensitive, see conftest.assert_mostly_close)
    for k, p in fine.named_parameters():
        ref = gold[f"train_grad_fine_{k}"]
        assert abs(grad_summary(p.grad)[1] - ref[1]) <= 5e-2 * max(ref[1], 1e-12), k


@pytest.mark.parametrize("train_precision", TRAIN_PRECISIONS)
def test_fine_grads_on_reference_positions(gold, train_precision):
    """Same step with the fine pass evaluated at the reference's own merged depths: gradients to 2e-3."""
    from nerf_amd import ops
    from nerf_amd.render import raw2outputs
    coarse, fine = make_nets(train_precision)
    ro, rd = dev(gold["train_ro"]), dev(gold["train_rd"])
    n = ro.shape[0]
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    rays = torch.cat([ro, rd, torch.full((n, 1), 2., device="cuda"), torch.full((n, 1), 6., device="cuda"), vd], -1)
    z = dev(gold["train_z_vals"])
    pts = ops.points(rays, z)
    raw = fine.query_points(pts.reshape(-1, 3), vd, 192, 10, 4).reshape(n, 192, 4)
    rgb, _, _, _, _ = raw2outputs(raw, z, rays[:, 3:6], 0, True)
    loss = torch.mean((rgb - dev(gold["train_target"])) ** 2)
    loss.backward()
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), gold["train_rgb"], atol=1e-4, rtol=0)
    check_grads(fine, gold, "fine", 2e-3)


def test_adam_step_and_repack(gold):
    """Two optimizer steps: the packed weight copy must follow the nn.Linear parameters (loss changes)."""
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render, run_network
    from nerf_amd.utils import img2mse
    coarse, fine = make_nets()
    opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=1., N_importance=128,
              network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True,
              raw_noise_std=1., ndc=False, lindisp=False)
    K, _ = fx.intrinsics(400, 400)
    batch_rays = torch.stack([dev(gold["train_ro"]), dev(gold["train_rd"])], 0)
    target = dev(gold["train_target"])
    losses = []
    for _ in range(3):
        rgb, _, _, extras = render(400, 400, K, chunk=32768, rays=batch_rays, retraw=True, near=2., far=6., **kw)
        opt.zero_grad()
        loss = img2mse(rgb, target) + img2mse(extras["rgb0"], target)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        assert all(torch.isfinite(p.grad).all() for p in coarse.parameters())
    assert len(set(losses)) == 3


def test_bucketed_allreduce_streams_on_the_gpu():
    """BucketedGradAllReduce's overlapped path (side stream + async RCCL all-reduce started from the gradient hooks) in a
    one-rank nccl group on this GPU: a real training backward (render -> loss -> backward through both nets), launch
    order fine -> coarse, gradients bit-identical to the same backward without the reducer (average over one rank).
    Runs in a child process so that the test process never holds a process group."""
    import subprocess
    import sys
    script = r"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
import nerf_fixtures as fx
from nerf_amd import parallel
from nerf_amd.network import NeRF
from nerf_amd.positional_encoding import get_embedder
from nerf_amd.render import render, run_network
from nerf_amd.utils import img2mse
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(sys.argv[1]), rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
nets = []
for seed in (21, 22):
    n = NeRF(8, 256, 63, 27, 5, [4], True)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(seed, peaky=True).items()})
    nets.append(n.cuda())
coarse, fine = nets
e10, _ = get_embedder(10, 0); e4, _ = get_embedder(4, 0)
kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
          network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.,
          ndc=False, lindisp=False)
ro, rd, target = fx.train_batch(256, seed=5)
rays = torch.stack([torch.from_numpy(ro), torch.from_numpy(rd)]).cuda()
tgt = torch.from_numpy(target).cuda()
K, _ = fx.intrinsics(400, 400)
params = list(fine.parameters()) + list(coarse.parameters())
def backward():
    for p in params:
        p.grad = None
    rgb, _, _, ex = render(400, 400, K, chunk=32768, rays=rays, retraw=True, near=2., far=6., **kw)
    (img2mse(rgb, tgt) + img2mse(ex["rgb0"], tgt)).backward()
backward()
want = [p.grad.clone() for p in params]
ar = parallel.BucketedGradAllReduce([list(fine.parameters()), list(coarse.parameters())], min_world=1)
for it in range(3):
    backward()
    order = ar.finish()
    assert order == [0, 1], order
    torch.cuda.synchronize()
    for p, w in zip(params, want):
        assert torch.equal(p.grad, w)
dist.destroy_process_group()
print("BUCKETED_OK")
"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-c", script, str(port)], capture_output=True, text=True, timeout=110,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "BUCKETED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def p16_to_t8_index(sections):
    """idx with t8_order = p16_order[idx] for a run of f16 workspace sections [(channels, kind)]: the f32 sections are
    channel-major T8 (element (c, point q) at ((q >> 3) C + c) 8 + (q & 7)); the f16 ones are point-major: per 32-channel
    tile ct two blocks j of [point][lane half h][8 values e], holding accumulator register 8j + e of half h -- channel
    32 ct + ((8j+e) & 3) + 8 ((8j+e) >> 2) + 4 h of a hidden section, k-step-space channel 32 ct + 2 (8j+e) + h of an
    encoded-input section."""
    out, base = [], 0
    for C, kind in sections:
        idx = np.arange(32 * C, dtype=np.int64)
        if kind is not None:
            ct, j, n, h, e = np.meshgrid(np.arange(C // 32), np.arange(2), np.arange(32), np.arange(2), np.arange(8), indexing="ij")
            r = 8 * j + e
            c = 32 * ct + ((r & 3) + 8 * (r >> 2) + 4 * h if kind == "hidden" else 2 * r + h)
            src = ((2 * ct + j) * 32 + n) * 16 + 8 * h + e
            idx[(((n >> 3) * C + c) * 8 + (n & 7)).reshape(-1)] = src.reshape(-1)
        out.append(idx + base)
        base += 32 * C
    return np.concatenate(out)


@pytest.mark.parametrize("cfg", ["views256", "noview128", "views128_ragged"])
def test_bf16x3_training_forward_saves_the_f32_workspace(cfg):
    """nrf_query_fwd_train_bf16x3 against nrf_query_fwd_train on the same net and points: raw within 5e-5 of its scale
    (the bf16x3 inference bar), and the saved workspace -- which the f32 backward kernels consume as it is -- within 5e-5
    of each section's scale, ReLU gate words identical except where a pre-activation is within that error of zero.  Then
    the parameter gradients of the two paths (same f32 backward kernels) within 1e-2 of their norm (gate flips)."""
    import ctypes as C
    from nerf_amd import _lib
    from nerf_amd.network import NeRF
    rng = np.random.RandomState(11)
    if cfg == "views256":
        net, P, ppr, Lv = NeRF(8, 256, 63, 27, 5, [4], True), 40 * 64, 64, 4
        sd = fx.make_nerf_state(24)
    elif cfg == "noview128":
        net, P, ppr, Lv = NeRF(4, 128, 63, 0, 5, [2], False), 1000, 1, 0
        sd = fx.make_nerf_state(41, D=4, W=128, input_ch=63, input_ch_views=0, output_ch=5, skips=(2,), use_viewdirs=False)
    else:
        net, P, ppr, Lv = NeRF(3, 128, 39, 15, 4, [1], True), 37 * 9, 37, 2      # 333 points: a ragged 128-point pass
        sd = fx.make_nerf_state(26, D=3, W=128, input_ch=39, input_ch_views=15, output_ch=4, skips=(1,))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda()
    L = (net.input_ch - 3) // 6
    pts = dev(rng.uniform(-2, 2, (P, 3)))
    dirs = dev(fx.blender_rays(P // ppr, 3)[:, -3:]) if Lv else None
    f32 = net._packed((0, L, Lv))
    x3 = net._packed((0, L, Lv, "bf16x3"))
    raw_a, acts_a = f32.query_train(pts, dirs, ppr)
    raw_b, acts_b = x3.query_train(pts, dirs, ppr)
    scale = float(raw_a.abs().max())
    assert float((raw_a - raw_b).abs().max()) <= 5e-5 * scale
    tl_floats = _lib.lib().nrf_train_acts_floats(C.byref(f32.net), 32)
    A, B = acts_a.reshape(-1, tl_floats), acts_b.reshape(-1, tl_floats)
    n_full = P // 32                                  # tiles whose 32 points are all live
    W, D = net.W, net.D
    a_trunk = 32 * (64 + 32 + 24)
    a_mask = a_trunk + 32 * W * D + 32 * W + 32 * (W // 2)
    keep = torch.ones(a_mask, dtype=torch.bool, device="cuda")
    keep[32 * 96:32 * 120] = False                    # the time-input section belongs to deformation nets: never written here
    if not Lv:          # no view branch: the direction / feature / view-layer sections are never written
        keep[32 * 64:32 * 96] = False
        keep[a_trunk + 32 * W * D:] = False
    fa, fb = A[:n_full, :a_mask][:, keep], B[:n_full, :a_mask][:, keep]
    # encoder channels of frequencies >= L hold finite values the gradient unpack drops; compare where the f32 kernel is live
    live = fa != 0
    err = float(((fa - fb).abs() * live).max())
    assert err <= 5e-5 * float(fa.abs().max()), err
    ga = A[:n_full, a_mask:a_mask + 256 * (D + 1)].view(torch.int32)
    gb = B[:n_full, a_mask:a_mask + 256 * (D + 1)].view(torch.int32)
    if not Lv:
        ga, gb = ga[:, :256 * D], gb[:, :256 * D]
    flips = int(torch.count_nonzero((ga ^ gb)))
    assert flips <= 1e-4 * ga.numel(), (flips, ga.numel())
    g_raw = dev(rng.normal(size=tuple(raw_a.shape)))
    grads_a = f32.query_bwd(g_raw, P, acts_a)
    grads_b = f32.query_bwd(g_raw, P, acts_b)
    # a pre-activation within 1e-5 of zero gates differently in the two arithmetics (counted above): a flipped fraction f
    # of the gates moves a gradient's norm by ~sqrt(f) (measured 3.2e-3 on the 8 x 256 net, flips <= 1e-4 of the gates).
    # Against the reference's autograd both arithmetics meet the same 2e-3 (tests above): this is ReLU, not rounding.
    for k in grads_a:
        d = float((grads_a[k] - grads_b[k]).double().norm())
        assert d <= 1e-2 * max(float(grads_a[k].double().norm()), 1e-12), (k, d)
    # nrf_query_bwd_bf16x3 on the SAME workspace (identical gates): only the weight-gradient GEMMs change arithmetic, so
    # the agreement is the arithmetic's own -- 5e-5 of each gradient's scale
    # ("wgrad": the weight-gradient GEMMs alone; True: the activation-gradient chain on the bf16 pipe as well)
    for mode in ("wgrad", True):
        grads_c = f32.query_bwd(g_raw, P, acts_a, x3=mode)
        for k in grads_a:
            sc = max(float(grads_a[k].abs().max()), 1e-12)
            e = float((grads_a[k] - grads_c[k]).abs().max())
            assert e <= 5e-5 * sc, (mode, k, e, sc)
    # "a16": the same forward saving its activations as f16 (nrf_query_fwd_train_bf16x3_a16) -- raw bit for bit, the
    # workspace the f32 one rounded to f16 (2^-11 relative, point-major sections of half the floats), gate
    # words identical; nrf_query_bwd_bf16x3_a16 then differs from nrf_query_bwd_bf16x3 on the f32 workspace by that
    # rounding of the weight-gradient GEMMs' X operand alone: <= 1e-3 of each gradient's scale (measured ~2e-4)
    raw_c, acts_c = x3.query_train(pts, dirs, ppr, a16=True)
    assert torch.equal(raw_c, raw_b)
    a_mask_h = a_mask // 2
    tile_h = a_mask_h + 256 * (D + 1)
    Cw = acts_c[:(P + 31) // 32 * tile_h].reshape(-1, tile_h)
    # the f16 sections are POINT-major (P16, store_tile_p16 in mlp_bf16.hip): bring them back to the f32 workspace's order
    perm = torch.from_numpy(p16_to_t8_index([(64, "input"), (32, "input"), (24, None)] + [(W, "hidden")] * (D + 1)
                                            + [(W // 2, "hidden")])).cuda()
    fc = Cw[:n_full, :a_mask_h].contiguous().view(torch.float16).float()[:, perm][:, keep]
    assert float(((fc - fb).abs() * live).max()) <= 6e-4 * float(fb.abs().max())
    assert torch.equal(Cw[:n_full, a_mask_h:].view(torch.int32)[:, :gb.shape[1]], gb)
    grads_x3 = f32.query_bwd(g_raw, P, acts_b, x3=True)
    grads_h = f32.query_bwd(g_raw, P, acts_c, x3=True, a16=True)
    worst = 0.0
    for k in grads_x3:
        sc = max(float(grads_x3[k].abs().max()), 1e-12)
        e = float((grads_x3[k] - grads_h[k]).abs().max()) / sc
        worst = max(worst, e)
        assert e <= 1e-3, (k, e)
        if k.endswith("bias"):       # bias gradients are row sums of G alone: untouched by the activations' format
            assert e <= 5e-5, (k, e)
    print(f"a16 {cfg}: weight gradients within {worst:.1e} of their scale of the f32-workspace bf16x3 backward")
    # "w16": the gradient blocks saved as f16 too (scaled by the power of two that brings max |g_raw| into [2^10, 2^11)), one
    # f16 MFMA per tile pair, the reduction undoes the scale -- also when the incoming gradient is tiny or huge.  Both
    # operands of every product carry a 2^-12 rounding: <= 1.5e-3 of a gradient's norm (measured: 2-4e-4 for the
    # layers' matrices, 9.6e-4 for alpha_linear's single row, whose sum over the points cancels the most), <= 3e-3 of its largest
    # element on the worst of 65 536 elements (measured 1.7e-3); against the reference's autograd every golden holds 2e-3
    for mag in (1.0, 3e-7, 5e4):
        grads_ref = f32.query_bwd(g_raw * mag, P, acts_b, x3=True)
        grads_w = f32.query_bwd(g_raw * mag, P, acts_c, x3=True, a16="w16")
        worst, worst_f = 0.0, 0.0
        for k in grads_ref:
            sc = max(float(grads_ref[k].abs().max()), 1e-30)
            e = float((grads_ref[k] - grads_w[k]).abs().max()) / sc
            ef = float((grads_ref[k] - grads_w[k]).double().norm() / max(float(grads_ref[k].double().norm()), 1e-30))
            worst, worst_f = max(worst, e), max(worst_f, ef)
            if grads_ref[k].numel() >= 64:
                assert e <= 3e-3 and ef <= 1.5e-3, (mag, k, e, ef)
            else:       # a handful of sums of 2 560 signed terms (alpha_linear.bias is ONE): relative to a sum that cancels
                assert e <= 1e-2, (mag, k, e)
        print(f"w16 {cfg} (|g_raw| x {mag:g}): gradients within {worst_f:.1e} of their norm ({worst:.1e} of their largest "
              "element) of the f32-workspace bf16x3 backward")
    # an all-zero incoming gradient (scale 1, nothing to divide by): every gradient exactly zero; and ONE non-zero point among
    # zeros: the scale comes from that point, the pass stays finite and agrees with the f32-workspace pass
    grads_0 = f32.query_bwd(torch.zeros_like(g_raw), P, acts_c, x3=True, a16="w16")
    assert all(bool((v == 0).all()) for v in grads_0.values())
    g_one = torch.zeros_like(g_raw)
    g_one[P // 2] = g_raw[P // 2]
    grads_ref = f32.query_bwd(g_one, P, acts_b, x3=True)
    grads_w = f32.query_bwd(g_one, P, acts_c, x3=True, a16="w16")
    for k in grads_ref:
        assert bool(torch.isfinite(grads_w[k]).all()), k
        ef = float((grads_ref[k] - grads_w[k]).double().norm()) / max(float(grads_ref[k].double().norm()), 1e-30)
        assert ef <= 3e-3, (k, ef)


@pytest.mark.parametrize("train_precision", TRAIN_PRECISIONS)
@pytest.mark.parametrize("views,W,D,skips,n,ppr", [(True, 256, 8, [4], 192, 64), (False, 128, 4, [1], 77, 1)])
def test_query_point_gradients(train_precision, views, W, D, skips, n, ppr):
    """d(raw)/d(pts) of the fused query (what the reference's autograd gives a caller that differentiates through
    run_network w.r.t. the sample positions, run_nerf_helpers-style; network.py:46-69 + positional_encoding.py:25-36):
    nrf_query_bwd_input behind NeRF.query_points, against autograd through the same module composed from torch ops in f64."""
    from nerf_amd.network import NeRF
    from nerf_amd.positional_encoding import get_embedder
    torch.manual_seed(11)
    net = NeRF(D=D, W=W, input_ch=39, input_ch_views=15 if views else 0, output_ch=4, skips=skips, use_viewdirs=views).cuda()
    net.train_precision = train_precision
    L, Lv = 6, 2                                   # moderate frequencies: the gradient is then well above f32 noise
    e, _ = get_embedder(L, 0)
    ev, _ = get_embedder(Lv, 0)
    rs = np.random.RandomState(3)
    pts = dev(rs.uniform(-1, 1, (n, 3)).astype(np.float32)).requires_grad_(True)
    dirs = dev(rs.normal(size=(n // ppr, 3)).astype(np.float32))
    g = dev(rs.normal(size=(n, 4)).astype(np.float32))
    raw = net.query_points(pts, dirs if views else None, ppr, L, Lv)
    (raw * g).sum().backward()
    got, got_params = pts.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    # the same graph in float64 torch ops
    net64 = NeRF(D=D, W=W, input_ch=39, input_ch_views=15 if views else 0, output_ch=4, skips=skips, use_viewdirs=views)
    net64.load_state_dict(net.state_dict())
    net64 = net64.cuda().double()
    p64 = pts.detach().double().requires_grad_(True)

    def emb(x, Lx):
        parts = [x] + [f(x * 2.0 ** j) for j in range(Lx) for f in (torch.sin, torch.cos)]
        return torch.cat(parts, -1)
    xin = emb(p64, L)
    if views:
        xin = torch.cat([xin, emb(dirs.double().repeat_interleave(ppr, 0), Lv)], -1)
    else:
        xin = torch.cat([xin, xin[:, :0]], -1)
    ref = net64._forward_autograd(xin)
    np.testing.assert_allclose(raw.detach().cpu().numpy(), ref.detach().cpu().numpy(), atol=5e-5 * float(ref.detach().abs().max()))
    (ref * g.double()).sum().backward()
    scale = float(p64.grad.norm())
    err = float((got.double() - p64.grad).norm()) / scale
    print(f"d/dpts ({train_precision}, W={W}): relative error {err:.2e}")
    # bf16x3: the forward's 1e-5 error flips a few ReLU gates; the point gradient chain itself is f32
    assert err <= (2e-5 if train_precision == "fp32" else 1e-2), err
    for k, p in net64.named_parameters():
        if p.grad is None:
            assert k not in got_params, k
            continue
        e_k = float((got_params[k].double() - p.grad).norm()) / max(float(p.grad.norm()), 1e-12)
        assert e_k <= (2e-4 if train_precision == "fp32" else 1e-2), (k, e_k)
    # view directions: refused, not silently dropped
    if views:
        with pytest.raises(NotImplementedError, match="view directions"):
            net.query_points(pts.detach(), dirs.clone().requires_grad_(True), ppr, L, Lv)


@pytest.mark.parametrize("train_precision,W", [("fp32", 64), ("fp32", 256), ("bf16x3", 128), ("bf16x3_a16", 128),
                                               ("bf16x3_w16", 256), ("bf16x3_w16", 128)])
def test_deep_net_with_several_skips(train_precision, W):
    """D = 16 with three skip layers (the deepest net the kernels take; network.py:29-53): 19 weight-gradient GEMMs and more
    slab reductions than one batched launch holds, so the backward flushes its queues mid-pass.  Parameter gradients against
    the same module composed from torch ops in float64."""
    from nerf_amd.network import NeRF
    from nerf_amd.positional_encoding import get_embedder
    torch.manual_seed(5)
    kw = dict(D=16, W=W, input_ch=63, input_ch_views=27, output_ch=4, skips=[2, 7, 12], use_viewdirs=True)
    net = NeRF(**kw).cuda()
    net.train_precision = train_precision
    rs = np.random.RandomState(9)
    n, ppr = 40 * 33, 33
    pts = dev(rs.uniform(-1, 1, (n, 3)).astype(np.float32))
    dirs = dev(rs.normal(size=(n // ppr, 3)).astype(np.float32))
    g = dev(rs.normal(size=(n, 4)).astype(np.float32))
    raw = net.query_points(pts, dirs, ppr, 10, 4)
    (raw * g).sum().backward()
    net64 = NeRF(**kw)
    net64.load_state_dict(net.state_dict())
    net64 = net64.cuda().double()

    def emb(x, Lx):
        return torch.cat([x] + [f(x * 2.0 ** j) for j in range(Lx) for f in (torch.sin, torch.cos)], -1)
    ref = net64._forward_autograd(torch.cat([emb(pts.double(), 10), emb(dirs.double().repeat_interleave(ppr, 0), 4)], -1))
    assert float((raw.detach().double() - ref.detach()).abs().max()) <= 5e-5 * float(ref.detach().abs().max())
    (ref * g.double()).sum().backward()
    worst = 0.0
    for (k, p), (_, q) in zip(net.named_parameters(), net64.named_parameters()):
        e = float((p.grad.double() - q.grad).norm()) / max(float(q.grad.norm()), 1e-30)
        worst = max(worst, e)
        assert e <= (2e-4 if train_precision == "fp32" else 1e-2), (k, e)
    print(f"D=16 W={W} {train_precision}: worst relative gradient error {worst:.2e}")


def test_deepest_supported_nets_inference():
    """D = 16 (the documented maximum) through the other kernels: bf16 / bf16x3 queries against the f32 kernel, the LayerNorm
    instantiation (115 packed sections: more than one pack launch holds) against its torch composition."""
    from nerf_amd import vanilla_dynamic as vd
    from nerf_amd.network import NeRF
    from nerf_amd.positional_encoding import get_embedder
    torch.manual_seed(6)
    rs = np.random.RandomState(10)
    n, ppr = 64 * 20, 20
    pts = dev(rs.uniform(-1, 1, (n, 3)).astype(np.float32))
    dirs = dev(rs.normal(size=(n // ppr, 3)).astype(np.float32))
    net = NeRF(D=16, W=128, input_ch=63, input_ch_views=27, skips=[3, 9], use_viewdirs=True).cuda()
    with torch.no_grad():
        ref = net.query_points(pts, dirs, ppr, 10, 4)
        net.precision = "bf16x3"
        x3 = net.query_points(pts, dirs, ppr, 10, 4)
        net.precision = "bf16"
        bf = net.query_points(pts, dirs, ppr, 10, 4)
    scale = float(ref.abs().max())
    assert float((x3 - ref).abs().max()) <= 5e-5 * scale
    assert float((bf - ref).abs().max()) <= 3e-2 * scale
    e, _ = get_embedder(10, 0)
    ev, _ = get_embedder(4, 0)
    ln = vd.NeRF(D=16, W=256, input_ch=63, input_ch_views=27, skips=[4, 8, 12], use_viewdirs=True, use_layernorm=True).cuda()
    with torch.no_grad():
        x = torch.cat([e(pts), ev(dirs.repeat_interleave(ppr, 0))], -1)
        a, _ = ln(x)
        assert ln._packed_cache, "the fused LayerNorm kernel did not run"
        b = ln._forward_layernorm(x)
    assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("train_precision", TRAIN_PRECISIONS)
def test_training_trajectory_vs_reference(train_precision):
    """Thirty consecutive steps of the reference's training loop (train.py:226-251: render -> mse(rgb) + mse(rgb0) ->
    backward -> Adam -> learning-rate decay) against the imported reference's own run of the same loop (CPU f32,
    tests/golden/trajectory.npz): the closed loop of every kernel on the path, the fused Adam and the weight repacking.  The
    loss falls from 0.27 to 0.11 over the run; every step's loss is held to the reference's."""
    import importlib.util
    from nerf_amd.optim import FusedAdam
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render, run_network
    from nerf_amd.utils import img2mse, mse2psnr
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(here, "golden", "trajectory.npz"))
    from nerf_amd.network import NeRF
    coarse, fine = NeRF(8, 256, 63, 27, 5, [4], True), NeRF(8, 256, 63, 27, 5, [4], True)
    for net, seed in ((coarse, 24), (fine, 25)):            # the well-conditioned fixtures the generator trained
        net.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(seed).items()})
        net.train_precision = train_precision
        net.cuda()
    nets = (("coarse", coarse), ("fine", fine))
    w0 = {f"{t}.{k}": p.detach().clone() for t, n in nets for k, p in n.named_parameters()}
    lrate, decay_rate, decay_steps = 5e-4, 0.1, 250 * 1000
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=lrate, betas=(0.9, 0.999))
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
              network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.,
              ndc=False, lindisp=False)
    K, _ = fx.intrinsics(400, 400)
    ro, rd, target = mg.traj_batches()
    R = mg.TRAJ_RAYS
    losses, psnrs = [], []
    for i in range(mg.TRAJ_STEPS):
        sl = slice(i * R, (i + 1) * R)
        batch_rays = torch.stack([dev(ro[sl]), dev(rd[sl])], 0)
        rgb, _, _, extras = render(400, 400, K, chunk=32768, rays=batch_rays, retraw=True, near=2., far=6., **kw)
        opt.zero_grad()
        img_loss = img2mse(rgb, dev(target[sl]))
        loss = img_loss + img2mse(extras["rgb0"], dev(target[sl]))
        loss.backward()
        opt.step()
        opt.param_groups[0]["lr"] = lrate * (decay_rate ** ((i + 1) / decay_steps))
        losses.append(loss.item())
        psnrs.append(mse2psnr(img_loss).item())
    losses, psnrs = np.array(losses), np.array(psnrs)
    rel = np.abs(losses - gold["losses"]) / gold["losses"]
    print(f"trajectory ({train_precision}): loss {losses[0]:.4f} -> {losses[-1]:.4f} (reference {gold['losses'][-1]:.4f}); "
          f"worst relative loss difference {rel.max():.2e} at step {int(rel.argmax())}, PSNR difference "
          f"{np.abs(psnrs - gold['psnrs']).max():.2e} dB")
    dn = np.array([float((p.detach() - w0[f"{t}.{k}"]).double().norm()) for t, n in nets for k, p in n.named_parameters()])
    dn_rel = np.abs(dn - gold["delta_norms"]) / np.maximum(gold["delta_norms"], 1e-12)
    print(f"   parameter movement ||w_30 - w_0|| vs the reference's: worst {dn_rel.max():.2e} ({gold['names'][int(dn_rel.argmax())]})")
    with torch.no_grad():
        rgbf, _, accf, ex = render(400, 400, K, chunk=32768, rays=torch.stack([dev(ro[:R]), dev(rd[:R])], 0), retraw=True,
                                   near=2., far=6., **kw)
    d_rgb = float(np.abs(rgbf.cpu().numpy() - gold["final_rgb"]).max())
    print(f"   image of the trained nets vs the reference's trained nets: max |rgb| difference {d_rgb:.2e}")
    # measured: 2.2e-5 (losses), 1.3e-3 (movement of the least-moved tensor), 2.4e-5 (image) for both arithmetics
    assert rel.max() <= 2e-4, rel
    assert dn_rel.max() <= 5e-3
    assert d_rgb <= 2e-4
