#!/usr/bin/env python3
"""What ONE rank of an N-GPU job does, timed on one GPU (no collectives): its row share of the 800x800 render and its slice
of the 4096-ray training batch, for N = 1, 2, 4, 8.  The N-GPU job's time is this plus the gather / all-reduce; the driver's
SCALE record is the measurement, this is the per-rank compute it should be compared with.  One JSON line."""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402

import nerf_fixtures as fx  # noqa: E402
from bench import build_nets  # noqa: E402
from nerf_amd import parallel  # noqa: E402
from nerf_amd.optim import FusedAdam  # noqa: E402
from nerf_amd.positional_encoding import get_embedder  # noqa: E402
from nerf_amd.render import render, run_network  # noqa: E402
from nerf_amd.utils import img2mse  # noqa: E402

dev = torch.device("cuda:0")
coarse, fine = build_nets(torch, dev)
e10, _ = get_embedder(10, 0)
e4, _ = get_embedder(4, 0)
q = lambda i, v, f: run_network(i, v, f, e10, e4, 65536)  # noqa: E731
H = W = 800
K, _ = fx.intrinsics(H, W)
c2w = torch.from_numpy(fx.pose_spherical(-60.0, -30.0, 4.0)[:3, :4]).to(dev)
kw = dict(network_query_fn=q, perturb=0., N_importance=128, network_fine=fine, N_samples=64, network_fn=coarse,
          white_bkgd=True, raw_noise_std=0., lindisp=False)


def timed(f, n):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {"render_rank_ms": {}, "train_rank_ms": {}}
with torch.no_grad():
    for N in (1, 2, 4, 8):
        row0, nrows = parallel.row_partition(H, N)[N // 2]
        ms = timed(lambda: parallel.render_rows(H, W, K, c2w, row0, nrows, chunk=32768, ndc=False, near=2., far=6.,
                                                use_viewdirs=True, **kw), 3)
        out["render_rank_ms"][str(N)] = {"rows": nrows, "ms": ms, "ideal_ms": None}
one = out["render_rank_ms"]["1"]["ms"]
for N in (1, 2, 4, 8):
    e = out["render_rank_ms"][str(N)]
    e["ideal_ms"] = one / N
    e["efficiency_before_gather"] = one / N / e["ms"]
ro, rd, target = fx.train_batch(4096, seed=1)
for prec in ("fp32", "bf16x3"):
    coarse.train_precision = fine.train_precision = prec
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    res = {}
    for N in (1, 2, 4, 8):
        n = 4096 // N
        rays = torch.stack([torch.from_numpy(ro[:n]), torch.from_numpy(rd[:n])]).to(dev)
        tgt = torch.from_numpy(target[:n]).to(dev)

        def step():
            rgb, _, _, ex = render(400, 400, fx.intrinsics(400, 400)[0], chunk=32768, rays=rays, retraw=True, near=2., far=6.,
                                   use_viewdirs=True, ndc=False, **dict(kw, perturb=1.))
            opt.zero_grad()
            (img2mse(rgb, tgt) + img2mse(ex["rgb0"], tgt)).backward()
            opt.step()
        res[str(N)] = {"rays": n, "ms": timed(step, 20)}
    for N in (1, 2, 4, 8):
        res[str(N)]["efficiency_before_allreduce"] = res["1"]["ms"] / N / res[str(N)]["ms"]
    out["train_rank_ms"][prec] = res
print(json.dumps(out))
