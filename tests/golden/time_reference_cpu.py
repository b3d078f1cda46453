#!/usr/bin/env python3
"""Test infrastructure (it runs the oracle and the imported reference; never part of the product path).  Build-container only (the reference never travels to the GPU box): time the IMPORTED reference's PyTorch CPU path
and the C oracle port (oracle/liborc.so) on the same rays, same weights, same thread count, so that the `cpu_baseline`
of bench.py -- the port, timed on the GPU box's host cores -- can be read as a statement about the reference.
  C1 (BASELINE configs[0]): render() of one 100x100 view, 64 coarse + 128 fine samples, D=8 W=256 coarse+fine, no_grad
  C3: one 4096-ray training step, forward + backward through both nets (no optimizer step), reference only
Import recipe as tests/golden/make_golden.py (imageio -- PNG writing only -- replaced by an empty module object).
    python tests/golden/time_reference_cpu.py > profiles/r1_cpu_reference_timing.json"""
import json
import os
import statistics
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nerf_fixtures as fx  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import network
    import positional_encoding as pe
    import render as rr
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(threads)
    sds = [fx.make_nerf_state(21, peaky=True), fx.make_nerf_state(22, peaky=True)]
    nets = []
    for sd in sds:
        net = network.NeRF(8, 256, 63, 27, 5, [4], True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net)
    embed_fn, _ = pe.get_embedder(10, 0)
    embeddirs_fn, _ = pe.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, fn: rr.run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, 65536)  # noqa: E731
    kw = dict(network_query_fn=qfn, perturb=0., N_importance=128, network_fine=nets[1], N_samples=64,
              network_fn=nets[0], use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)
    H = W = 100
    K, _ = fx.intrinsics(H, W)
    c2w = fx.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    out = {"threads": threads, "torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(),
           "where": "build container (8 CPUs); the GPU box's host cores are a different CPU"}

    def med(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    # ---- C1 through the reference's render()
    def ref_render():
        with torch.no_grad():
            return rr.render(H, W, torch.from_numpy(K), chunk=32768, c2w=torch.from_numpy(c2w), near=2., far=6., **kw)
    t_ref = med(ref_render)
    out["reference_render_100x100"] = {"seconds": t_ref, "rays_per_s": H * W / t_ref}

    # ---- the same rays through the port
    coarse, fine = orc.Net(sds[0]), orc.Net(sds[1])
    pix = np.stack(np.meshgrid(np.arange(H), np.arange(W), indexing="ij"), -1).reshape(-1, 2)
    ro, rd = fx.pinhole_rays_np(H, W, K, fx.pose_spherical(30.0, -30.0, 4.0), pix)
    vd = (rd / np.linalg.norm(rd, axis=-1, keepdims=True)).astype(np.float32)
    n = H * W
    rays = np.concatenate([ro, rd, np.full((n, 1), 2, np.float32), np.full((n, 1), 6, np.float32), vd], -1)
    res = {}
    t_port = med(lambda: res.update(r=orc.render_rays(rays, coarse, fine, 64, 128, False, True)))
    out["port_render_100x100"] = {"seconds": t_port, "rays_per_s": n / t_port}
    rgb_ref = ref_render()[0].reshape(-1, 3).numpy()
    # sharp-density bench nets: sample_pdf amplifies 1e-7 differences on a few rays (DESIGN, conftest.assert_mostly_close)
    d = np.abs(res["r"]["rgb_map"] - rgb_ref)
    out["port_vs_reference_rgb"] = {"frac_within_1e-4": float((d <= 1e-4).mean()), "max_abs": float(d.max())}
    out["reference_over_port_time"] = t_ref / t_port

    # ---- C3: 4096-ray training step through the reference (forward + backward, both nets)
    ro4, rd4, target = fx.train_batch(n_rays=4096, H=400, W=400, seed=7)
    batch = torch.stack([torch.from_numpy(ro4), torch.from_numpy(rd4)])
    K4, _ = fx.intrinsics(400, 400)
    tgt = torch.from_numpy(target)

    def ref_train():
        for net in nets:
            net.zero_grad()
        rgb, _, _, extras = rr.render(400, 400, torch.from_numpy(K4), chunk=32768, rays=batch, near=2., far=6.,
                                      retraw=True, **kw)
        loss = torch.mean((rgb - tgt) ** 2) + torch.mean((extras["rgb0"] - tgt) ** 2)
        loss.backward()
    t_tr = med(ref_train, n=2)
    out["reference_train_step_4096"] = {"seconds": t_tr, "rays_per_s": 4096 / t_tr}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
