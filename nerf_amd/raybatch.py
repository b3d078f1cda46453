"""Random ray batching over all training images -- the `use_batching` branch of train.py:141-192.

The reference materialises rays_rgb [(n_train*H*W), 3, 3] on the host (get_rays_np of every pose stacked with the
pixels, train.py:147-153), shuffles it, uploads it (2.3 GB for 100 800x800 images) and gathers the whole table
through a new permutation at every epoch end (:188-192).  Here only the permutation lives on the GPU (8 B/ray); a batch
is one gather kernel that rebuilds each row's ray from its pose and reads its pixel (nrf_gather_ray_batch), so an epoch
end is one index gather and nothing is ever re-laid out.  Row order semantics are the reference's:
batch k of an epoch = table[perm][k*N_rand : (k+1)*N_rand], and a reshuffle composes perm = perm[randperm(n)].
"""
import numpy as np
import torch

from . import ops


class RayBatcher:
    def __init__(self, images, poses, H, W, K, i_train, N_rand, device="cuda", f64=True, shuffle=True,
                 reference_rng=False):
        """images [M, H, W, 3(+)], poses [M, >=3, >=4] (numpy or torch), i_train: indices of the training views.

        f64: see ops.gather_ray_batch (True = what get_rays_np computes under NumPy >= 2 with the float64 K of
        train.py).  reference_rng: draw permutations exactly where the reference does (np.random.shuffle for the first,
        CPU torch.randperm for later epochs) so that seeded runs visit rows in the reference's order; otherwise they
        are drawn on the device."""
        dev = torch.device(device)
        images = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
        poses = torch.as_tensor(np.asarray(poses) if not isinstance(poses, torch.Tensor) else poses)
        self.images = images[..., :3].to(dev, torch.float32).contiguous()
        self.poses = poses[:, :3, :4].to(dev, torch.float32).contiguous()
        self.H, self.W, self.K = int(H), int(W), [[float(v) for v in r] for r in np.asarray(K)]
        self.i_train = torch.as_tensor(np.asarray(i_train), dtype=torch.int32).to(dev)
        self.N_rand, self.f64, self.reference_rng = int(N_rand), bool(f64), bool(reference_rng)
        self.n_rows = int(self.i_train.numel()) * self.H * self.W
        self.perm = torch.arange(self.n_rows, device=dev, dtype=torch.int64)
        self.i_batch = 0
        self.epoch = 0
        if shuffle:
            if reference_rng:       # train.py:155 shuffles the table rows in place with numpy's global generator
                order = np.arange(self.n_rows)
                np.random.shuffle(order)
                self.perm = torch.from_numpy(order).to(dev)
            else:
                self.perm = torch.randperm(self.n_rows, device=dev)

    def rows(self, start, stop):
        return self.perm[start:stop]

    def next(self, rank=0, world=1):
        """-> (batch_rays [2, B, 3], target_s [B, 3]): train.py:183-192.  With world > 1 every rank advances the same
        permutation and takes its contiguous slice of the global batch (identical stream, disjoint rows)."""
        rows = self.perm[self.i_batch:self.i_batch + self.N_rand]
        if world > 1:
            from .parallel import slice_partition
            lo, cnt = slice_partition(rows.numel(), world)[rank]
            rows = rows[lo:lo + cnt]
        batch_rays, target_s = ops.gather_ray_batch(rows, self.images, self.poses, self.H, self.W, self.K,
                                                    img_index=self.i_train, f64=self.f64)
        self.i_batch += self.N_rand
        if self.i_batch >= self.n_rows:     # train.py:188-192
            if self.reference_rng:
                idx = torch.randperm(self.n_rows).to(self.perm.device)
            else:
                idx = torch.randperm(self.n_rows, device=self.perm.device)
            self.perm = self.perm[idx]
            self.i_batch = 0
            self.epoch += 1
        return batch_rays, target_s


class ImageRaySampler:
    """The `no_batching` branch of train.py:194-224 (and vanilla_dynamic/train.py:205-235): N_rand random pixels of ONE
    training image per step, optionally from the centre crop of the first `precrop_iters` iterations.  The reference
    builds all H*W rays of the image with get_rays and indexes them; here only the chosen pixels' rays are built, by the
    same gather kernel in its float32 mode (bit-identical to get_rays, rays.py:19-26)."""

    def __init__(self, images, poses, H, W, K, device="cuda", reference_rng=False):
        dev = torch.device(device)
        images = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
        poses = torch.as_tensor(np.asarray(poses) if not isinstance(poses, torch.Tensor) else poses)
        self.images = images[..., :3].to(dev, torch.float32).contiguous()
        self.poses = poses[:, :3, :4].to(dev, torch.float32).contiguous()
        self.H, self.W, self.K = int(H), int(W), [[float(v) for v in r] for r in np.asarray(K)]
        self.reference_rng = bool(reference_rng)

    def crop_pixels(self, precrop_frac=None):
        """Flat pixel indices (y*W + x) of the candidate set, in the reference's order (train.py:204-218)."""
        H, W = self.H, self.W
        if precrop_frac is None:
            return None
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        ys = torch.arange(H // 2 - dH, H // 2 + dH, device=self.images.device)
        xs = torch.arange(W // 2 - dW, W // 2 + dW, device=self.images.device)
        return (ys[:, None] * W + xs[None, :]).reshape(-1)

    def sample(self, img_i, N_rand, precrop_frac=None, select_inds=None):
        """-> (batch_rays [2, N_rand, 3], target_s [N_rand, 3]).  select_inds: positions in the candidate set (what
        np.random.choice returns at train.py:219); drawn here when None."""
        cand = self.crop_pixels(precrop_frac)
        n = self.H * self.W if cand is None else cand.numel()
        if select_inds is None:
            if self.reference_rng:
                select_inds = torch.from_numpy(np.random.choice(n, size=[N_rand], replace=False))
            else:
                select_inds = torch.randperm(n, device=self.images.device)[:N_rand]
        select_inds = torch.as_tensor(select_inds).to(self.images.device, torch.int64)
        pix = select_inds if cand is None else cand[select_inds]
        rows = pix + int(img_i) * self.H * self.W
        return ops.gather_ray_batch(rows, self.images, self.poses, self.H, self.W, self.K, f64=False)
