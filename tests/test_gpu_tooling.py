"""SURVEY 8f-4 on the GPU: the on-the-fly ray-batch gather against the reference's rays_rgb table, the batcher's row
order semantics (train.py:183-192), and reference-format checkpoints into modules / packed layouts / fused Adam."""
import os

import numpy as np
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLD, "ref_ckpt_small.tar")


@pytest.fixture(scope="module")
def tab():
    return np.load(os.path.join(GOLD, "tooling.npz"))


def _scene(tab):
    H, W = (int(v) for v in tab["table_hw"])
    return (H, W, tab["table_K"], torch.from_numpy(tab["table_images"]).cuda(), torch.from_numpy(tab["table_poses"]).cuda(),
            torch.from_numpy(tab["table_i_train"].astype(np.int32)).cuda())


def test_gather_equals_reference_table_bitwise(tab):
    from nerf_amd import ops
    H, W, K, images, poses, i_train = _scene(tab)
    n = tab["table"].shape[0]
    for rows in (torch.arange(n), torch.randperm(n, generator=torch.Generator().manual_seed(3))[: n // 2]):
        rays, rgb = ops.gather_ray_batch(rows.cuda(), images, poses, H, W, K, img_index=i_train, f64=True)
        want = tab["table"][rows.numpy()]
        np.testing.assert_array_equal(bits(rays[0].cpu().numpy()), bits(want[:, 0]))
        np.testing.assert_array_equal(bits(rays[1].cpu().numpy()), bits(want[:, 1]))
        np.testing.assert_array_equal(bits(rgb.cpu().numpy()), bits(want[:, 2]))


def test_gather_f32_mode_is_get_rays(tab):
    """f64=False is the float32 arithmetic of get_rays (rays.py:19-26): identical to the fused ray-matrix kernel."""
    from nerf_amd import ops
    H, W, K, images, poses, _ = _scene(tab)
    rows = torch.arange(H * W, device="cuda") + 2 * H * W          # every pixel of image 2, identity slot map
    rays, rgb = ops.gather_ray_batch(rows, images, poses, H, W, K, f64=False)
    full = ops.gen_rays(H, W, K, poses[2], 2.0, 6.0, False, False)
    np.testing.assert_array_equal(bits(rays[0].cpu().numpy()), bits(full[:, 0:3].cpu().numpy()))
    np.testing.assert_array_equal(bits(rays[1].cpu().numpy()), bits(full[:, 3:6].cpu().numpy()))
    np.testing.assert_array_equal(rgb.cpu().numpy(), tab["table_images"][2].reshape(-1, 3))


def test_gather_edges(tab):
    from nerf_amd import ops
    H, W, K, images, poses, i_train = _scene(tab)
    rays, rgb = ops.gather_ray_batch(torch.empty(0, dtype=torch.int64, device="cuda"), images, poses, H, W, K, i_train)
    assert rays.shape == (2, 0, 3) and rgb.shape == (0, 3)
    rows = torch.tensor([0, 3 * H * W, -1, 3 * H * W - 1], device="cuda")    # two rows outside the 3-image table
    rays, rgb = ops.gather_ray_batch(rows, images, poses, H, W, K, i_train)
    bad = torch.isnan(rgb).all(dim=1).cpu().numpy()
    np.testing.assert_array_equal(bad, [False, True, True, False])
    assert torch.isnan(rays[:, 1]).all() and torch.isnan(rays[:, 2]).all()
    bad_map = torch.tensor([0, 7, -1], dtype=torch.int32, device="cuda")          # slots 1, 2 name no image
    rays, rgb = ops.gather_ray_batch(torch.tensor([0, H * W, 2 * H * W], device="cuda"), images, poses, H, W, K, bad_map)
    np.testing.assert_array_equal(torch.isnan(rgb).all(dim=1).cpu().numpy(), [False, True, True])
    with pytest.raises(RuntimeError):
        ops.gather_ray_batch(rows.int(), images, poses, H, W, K, i_train)
    with pytest.raises(ValueError):
        ops.gather_ray_batch(rows, images[:, :-1], poses, H, W, K, i_train)


def test_batcher_row_order_and_epochs(tab):
    """batch k = table[perm][k*N : (k+1)*N]; at an epoch end perm <- perm[randperm] (train.py:183-192)."""
    from nerf_amd.raybatch import RayBatcher
    H, W = (int(v) for v in tab["table_hw"])
    torch.manual_seed(11)
    b = RayBatcher(tab["table_images"], tab["table_poses"], H, W, tab["table_K"], tab["table_i_train"], N_rand=32)
    table = tab["table"]
    n = table.shape[0]
    assert b.n_rows == n and sorted(b.perm.cpu().tolist()) == list(range(n))
    first = b.perm.clone()
    seen = []
    for k in range(4):                                   # 105 rows: the 4th batch is ragged and ends the epoch
        perm = b.perm.cpu().numpy()
        lo = b.i_batch
        rays, rgb = b.next()
        want = table[perm[lo:lo + 32]]
        seen.append(perm[lo:lo + 32])
        np.testing.assert_array_equal(bits(rays.cpu().numpy()), bits(np.transpose(want[:, :2], (1, 0, 2))))
        np.testing.assert_array_equal(bits(rgb.cpu().numpy()), bits(want[:, 2]))
    assert sorted(np.concatenate(seen).tolist()) == list(range(n))        # one epoch visits every row once
    assert b.epoch == 1 and b.i_batch == 0 and not torch.equal(b.perm, first)
    assert sorted(b.perm.cpu().tolist()) == list(range(n))


def test_batcher_reference_rng_and_rank_slices(tab):
    from nerf_amd.raybatch import RayBatcher
    H, W = (int(v) for v in tab["table_hw"])
    args = (tab["table_images"], tab["table_poses"], H, W, tab["table_K"], tab["table_i_train"])
    np.random.seed(4)
    order = np.arange(3 * H * W)
    np.random.shuffle(order)                             # what train.py:155 does to the table rows
    np.random.seed(4)
    b = RayBatcher(*args, N_rand=48, reference_rng=True)
    np.testing.assert_array_equal(b.perm.cpu().numpy(), order)
    np.random.seed(4)
    b2 = RayBatcher(*args, N_rand=48, reference_rng=True)
    whole = b.next()
    parts = [b2.next(rank=r, world=2) for r in (0, 1)]   # NB: each call advances; rebuild for rank 1
    np.random.seed(4)
    b3 = RayBatcher(*args, N_rand=48, reference_rng=True)
    p1 = b3.next(rank=1, world=2)
    got = torch.cat([parts[0][0], p1[0]], dim=1), torch.cat([parts[0][1], p1[1]], dim=0)
    assert torch.equal(got[0], whole[0]) and torch.equal(got[1], whole[1])


def test_image_ray_sampler_is_get_rays_indexing(tab):
    """no_batching branch (train.py:194-224): get_rays of the image, then [select_coords]; with and without precrop."""
    from nerf_amd.raybatch import ImageRaySampler
    from nerf_amd.rays import get_rays
    H, W, K, images, poses, _ = _scene(tab)
    s = ImageRaySampler(tab["table_images"], tab["table_poses"], H, W, tab["table_K"])
    ro, rd = get_rays(H, W, K, poses[1, :3, :4])
    for frac in (None, 0.5):
        if frac is None:
            coords = torch.stack(torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij"), -1)
        else:
            dH, dW = int(H // 2 * frac), int(W // 2 * frac)
            coords = torch.stack(torch.meshgrid(torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH),
                                                torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW), indexing="ij"), -1)
        coords = coords.reshape(-1, 2)
        sel = np.random.RandomState(5).choice(coords.shape[0], size=[min(6, coords.shape[0])], replace=False)
        sc = coords[sel].long()
        rays, tgt = s.sample(1, len(sel), precrop_frac=frac, select_inds=sel)
        np.testing.assert_array_equal(bits(rays[0].cpu().numpy()), bits(ro[sc[:, 0], sc[:, 1]].cpu().numpy()))
        # torch's GPU sum may contract differently from the reference's CPU order the kernel follows (<= 1 ulp); the
        # bit-exact statement is test_gather_f32_mode_is_get_rays (against the golden-pinned ray-matrix kernel)
        np.testing.assert_allclose(rays[1].cpu().numpy(), rd[sc[:, 0], sc[:, 1]].cpu().numpy(), rtol=0, atol=5e-7)
        np.testing.assert_array_equal(tgt.cpu().numpy(), tab["table_images"][1][sc[:, 0].numpy(), sc[:, 1].numpy()])
    rays, tgt = s.sample(2, 10)
    assert rays.shape == (2, 10, 3) and tgt.shape == (10, 3) and torch.isfinite(rays).all()


def test_checkpoint_to_modules_and_packed(tab):
    from nerf_amd import checkpoint as ck
    nets = ck.networks_from_checkpoint(CKPT, device="cuda")
    assert nets["global_step"] == 2
    x = torch.from_numpy(tab["ckpt_x"]).cuda()
    with torch.no_grad():
        for name, key in (("network_fn", "ckpt_out_coarse"), ("network_fine", "ckpt_out_fine")):
            out = nets[name](x).cpu().numpy()
            np.testing.assert_allclose(out, tab[key], rtol=0, atol=1e-5)
    # straight to the packed layout: raw points + directions in, the same numbers as the module's fused query
    pts = torch.rand(500, 3, device="cuda") * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(500, 3, device="cuda"), dim=-1)
    pk = ck.packed_from_checkpoint(CKPT, "network_fine")
    raw = pk.query(pts, dirs, 1)
    from nerf_amd.positional_encoding import get_embedder
    e, ev = get_embedder(10, 0)[0], get_embedder(4, 0)[0]
    with torch.no_grad():
        want = nets["network_fine"](torch.cat([e(pts), ev(dirs)], -1))
    np.testing.assert_allclose(raw.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-5)
    pkb = ck.packed_from_checkpoint(CKPT, "network_fine", precision="bf16")
    rawb = pkb.query(pts, dirs, 1)
    assert float((rawb - raw).abs().max()) < 3e-2 * max(1.0, float(raw.abs().max()))
    raw3 = ck.packed_from_checkpoint(CKPT, "network_fine", precision="bf16x3").query(pts, dirs, 1)
    assert float((raw3 - raw).abs().max()) < 5e-5 * max(1.0, float(raw.abs().max()))
    with pytest.raises(ValueError, match="precision"):
        ck.packed_from_checkpoint(CKPT, "network_fine", precision="fp16")


def test_checkpoint_resumes_training_like_torch(tab):
    """Adam state from the reference's file -> FusedAdam; one more step lands where the reference's third step did."""
    from nerf_amd import checkpoint as ck
    from nerf_amd.optim import FusedAdam
    nets = ck.networks_from_checkpoint(CKPT, device="cuda")
    coarse, fine = nets["network_fn"], nets["network_fine"]
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    opt.load_state_dict(nets["optimizer_state_dict"])
    x, tgt = torch.from_numpy(tab["ckpt_x"]).cuda(), torch.from_numpy(tab["ckpt_tgt"]).cuda()
    opt.zero_grad()
    loss = ((coarse(x) - tgt) ** 2).mean() + ((fine(x) - tgt) ** 2).mean()
    loss.backward()
    opt.step()
    assert abs(float(loss.detach()) - float(tab["ckpt_loss3"][0])) < 1e-5
    with torch.no_grad():
        np.testing.assert_allclose(coarse(x).cpu().numpy(), tab["ckpt_out3_coarse"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(fine(x).cpu().numpy(), tab["ckpt_out3_fine"], rtol=0, atol=2e-5)
