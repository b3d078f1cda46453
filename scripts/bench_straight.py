#!/usr/bin/env python3
"""The secondary D-NeRF networks on the GPU, one JSON line: network_type='straightforward' (800x800 render in fp32 and bf16x3,
1024-ray training step on the fused kernels and composed from GPU torch ops) and the LayerNorm variant (1 M embedded rows
through the fused kernel with the normalisation stage vs its torch composition).  Profiled by scripts/profile_gpu.sh
stats_straight -> profiles/<round>_dnerf_straight_kernel_stats.csv."""
import json
import os
import sys
import time
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402

import bench_dnerf  # noqa: E402
from nerf_amd import vanilla_dynamic as vd  # noqa: E402

dev = torch.device("cuda:0")
a = types.SimpleNamespace(size=int(sys.argv[1]) if len(sys.argv) > 1 else 800)
out = {}
for prec in ("fp32", "bf16x3", "bf16"):
    out[f"straightforward_render_{prec}"] = bench_dnerf.run_dnerf_straight(a, torch, dev, precision=prec)
out["straightforward_train_step"] = bench_dnerf.run_dnerf_straight_train(a, torch, dev)
out["straightforward_train_step_composed"] = bench_dnerf.run_dnerf_straight_train(a, torch, dev, fused=False)


def t(f, n=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


net = vd.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True, use_layernorm=True).to(dev)
x = torch.randn(1 << 20, 90, device=dev)
with torch.no_grad():
    out["layernorm_1M_rows_ms"] = {"fused": t(lambda: net(x)), "torch_composition": t(lambda: net._forward_layernorm(x))}


# training pass (forward + backward of every parameter incl. the LayerNorm affine ones) on 256 K embedded rows: the fused
# kernels (round 3: LayerNorm backward inside the dgrad chain) against the torch composition the module used before
xt = torch.randn(1 << 18, 90, device=dev)
gt = torch.randn(1 << 18, 4, device=dev)


def train_pass(fwd):
    net.zero_grad(set_to_none=True)
    (fwd(xt) * gt).sum().backward()


out["layernorm_256K_rows_train_ms"] = {"fused": t(lambda: train_pass(lambda v: net(v)[0])),
                                        "torch_composition": t(lambda: train_pass(net._forward_layernorm))}
plain = vd.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True).to(dev)


def plain_pass():
    plain.zero_grad(set_to_none=True)
    (plain(xt)[0] * gt).sum().backward()


out["layernorm_256K_rows_train_ms"]["plain_net_fused"] = t(plain_pass)
print(json.dumps(out))
