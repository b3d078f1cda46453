"""Multi-process paths on CPU (gloo, world_size 2): the row-sharded render gather and the flat gradient
all-reduce of nerf_amd.parallel are placement logic around torch.distributed; the kernels themselves are
covered by the GPU tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parallel.warm_up(torch.device("cpu"))
        # a deterministic stand-in for "render these rows": value = f(row, col)
        def rows(row0, nrows):
            r = torch.arange(row0, row0 + nrows, dtype=torch.float32)[:, None, None]
            c = torch.arange(W, dtype=torch.float32)[None, :, None]
            ch = torch.arange(5, dtype=torch.float32)[None, None, :]
            return (r * 1000 + c + ch / 10).reshape(-1, 5)
        img = parallel.render_sharded(rows, H, W, rank, world)
        # data-parallel gradient averaging
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(5)),
                  torch.nn.Parameter(torch.zeros(2, 2))]
        params[0].grad = torch.full((7, 3), float(rank + 1))
        params[1].grad = torch.arange(5, dtype=torch.float32) * (rank + 1)
        params[2].grad = None          # a parameter that got no gradient on this rank
        parallel.FlatGradAllReduce(params)()
        if rank == 0:
            out["img"] = img.numpy()
            out["g0"] = params[0].grad.numpy().copy()
            out["g1"] = params[1].grad.numpy().copy()
            out["g2"] = params[2].grad.numpy().copy()
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(8, 6), (7, 5)])
def test_render_sharded_and_grad_allreduce(H, W):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), H, W, out), nprocs=world, join=True)
    r = np.arange(H, dtype=np.float32)[:, None, None]
    c = np.arange(W, dtype=np.float32)[None, :, None]
    ch = np.arange(5, dtype=np.float32)[None, None, :]
    np.testing.assert_array_equal(out["img"], r * 1000 + c + ch / 10)      # bit-identical to the 1-rank image
    np.testing.assert_allclose(out["g0"], np.full((7, 3), 1.5))
    np.testing.assert_allclose(out["g1"], np.arange(5) * 1.5)
    np.testing.assert_allclose(out["g2"], 0.0)


def test_partitions():
    assert parallel.row_partition(800, 8) == [(100 * i, 100) for i in range(8)]
    assert parallel.row_partition(7, 2) == [(0, 4), (4, 3)]
    parts = parallel.slice_partition(4096, 3)
    assert sum(n for _, n in parts) == 4096 and parts[0][0] == 0
    assert all(a[0] + a[1] == b[0] for a, b in zip(parts, parts[1:]))


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)                      # same parameters on every rank
        coarse, fine = torch.nn.Linear(6, 4), torch.nn.Linear(6, 4)
        lonely = torch.nn.Parameter(torch.zeros(3))          # in the coarse bucket, never reached by the loss
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 + rank))     # each rank its own slice of the batch
        # train.py:232-239: loss = mse(rgb) + mse(rgb0); the FINE branch is built last, so autograd finishes it first
        def loss_fn():
            return (coarse(x) ** 2).mean() + (fine(x) ** 2).mean()
        # reference result: per-rank gradients averaged with one flat all-reduce
        loss_fn().backward()
        plain = [p for p in list(fine.parameters()) + list(coarse.parameters())]
        local = [p.grad.clone() for p in plain]
        parallel.FlatGradAllReduce(plain)()
        want = [p.grad.clone() for p in plain]
        for p in plain:
            p.grad = None
        ar = parallel.BucketedGradAllReduce([list(fine.parameters()), list(coarse.parameters()) + [lonely]])
        orders = []
        for _ in range(2):                        # two steps: the hook bookkeeping resets
            for p in plain:
                p.grad = None
            loss_fn().backward()
            orders.append(ar.finish())
        got = [p.grad.clone() for p in plain]
        assert lonely.grad is not None and float(lonely.grad.abs().max()) == 0.0
        ar.remove()
        # rank-dependent graphs: rank 1's loss never reaches the fine net, so there only the coarse bucket (index 1) fires
        # during the backward.  It must wait for bucket 0 (launched at finish() with zeros) -- every rank issues the
        # collectives in index order, or buckets of different sizes would be paired.  Buckets listed coarse-first as well:
        # the later-ready bucket 0 holds back the earlier-ready bucket 1.
        for buckets in ([list(fine.parameters()), list(coarse.parameters())],
                        [list(coarse.parameters()), list(fine.parameters())]):
            ar = parallel.BucketedGradAllReduce(buckets)
            for p in plain:
                p.grad = None
            ((coarse(x) ** 2).mean() + ((fine(x) ** 2).mean() if rank == 0 else 0.0)).backward()
            local_fine = fine.weight.grad.clone() if rank == 0 else torch.zeros_like(fine.weight)   # write-back is finish()'s
            order = ar.finish()
            assert order == [0, 1], order
            dist.all_reduce(local_fine)                  # after finish(): no collective of the test between the buckets'
            assert torch.allclose(fine.weight.grad, local_fine / world, atol=1e-7)
            ar.remove()
        # one backward() per finish(): a second backward meets a bucket already in flight
        ar = parallel.BucketedGradAllReduce([list(fine.parameters()), list(coarse.parameters())])
        loss_fn().backward()
        try:
            loss_fn().backward()
            raised = False
        except RuntimeError as e:
            raised = "finish()" in str(e)
        ar.finish()
        ar.remove()
        assert raised
        if rank == 0:
            out["orders"] = orders
            out["max_err"] = max(float((a - b).abs().max()) for a, b in zip(got, want))
            out["changed"] = max(float((a - b).abs().max()) for a, b in zip(got, local))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_order_and_result():
    """Two buckets (fine net, coarse net): the fine bucket's all-reduce starts first -- during the backward, from the
    gradient hooks -- the coarse bucket (which holds a parameter that never gets a gradient) at finish(); the averaged
    gradients equal one flat all-reduce's."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out["orders"] == [[0, 1], [0, 1]], out["orders"]
    assert out["max_err"] <= 1e-7, out["max_err"]
    assert out["changed"] > 1e-3           # the ranks did have different gradients
