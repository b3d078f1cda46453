"""bench.py --workload train: BASELINE.json configs[2] -- a 4096-ray training step (64 coarse + 128 fine
samples, both D=8/W=256 nets): render() forward, img2mse(rgb)+img2mse(rgb0), backward through the
compositor and both MLPs, (data-parallel: one flat RCCL all-reduce of the gradients), Adam step.
perturb=1 / raw_noise_std=0 as configs/lego.txt trains.  With N ranks each takes 4096/N rays of the batch."""
import json
import os
import time

import numpy as np

FWD_FLOP_PER_POINT = 1186816
STEP_FLOP_PER_POINT = 2 * 1744512   # fwd + dgrad + wgrad, SURVEY 8d


def step_traffic(workload="train"):
    """HBM bytes per training step summed over this library's kernels, from the committed PMC passes."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        t = json.load(open(files[-1]))[workload]
        return sum(v["launches"] * v["hbm_bytes"] for k, v in t.items() if isinstance(v, dict) and "hbm_bytes" in v) \
            / t["_steps_profiled"]
    except (KeyError, ValueError):
        return None


def run_train(args, torch, dist, world, rank, device, emit=True):
    import nerf_fixtures as fx
    from bench import PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, build_nets
    from nerf_amd import parallel
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render, run_network
    from nerf_amd.utils import img2mse

    coarse, fine = build_nets(torch, device)
    tprec = getattr(args, "train_precision", "fp32")
    coarse.train_precision = fine.train_precision = tprec
    params = list(coarse.parameters()) + list(fine.parameters())
    from nerf_amd.optim import FusedAdam
    opt = FusedAdam(params, lr=5e-4, betas=(0.9, 0.999))   # what create_nerf hands out on the GPU
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=1., N_importance=128,
              network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True,
              raw_noise_std=0., ndc=False, lindisp=False)
    n_global = getattr(args, "train_rays", 4096)
    ro, rd, target = fx.train_batch(n_global, seed=1)
    r0, n_local = parallel.slice_partition(n_global, world)[rank]
    batch_rays = torch.stack([torch.from_numpy(ro[r0:r0 + n_local]), torch.from_numpy(rd[r0:r0 + n_local])]).to(device)
    target_t = torch.from_numpy(target[r0:r0 + n_local]).to(device)
    K, _ = fx.intrinsics(400, 400)
    # two buckets in the order the backward finishes them (fine net first: its branch of the graph is the younger one);
    # the fine bucket's all-reduce runs on a side stream under the coarse net's backward
    allreduce = parallel.BucketedGradAllReduce([list(fine.parameters()), list(coarse.parameters())]) if world > 1 else None
    ar_events, step_events = [], []

    def step():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rgb, disp, acc, extras = render(400, 400, K, chunk=32768, rays=batch_rays, retraw=True, near=2., far=6., **kw)
        opt.zero_grad()
        # equal slices: the mean over the GLOBAL batch is the average of the local means, which the all-reduce forms
        loss = (img2mse(rgb, target_t) + img2mse(extras["rgb0"], target_t))
        loss.backward()
        if allreduce is not None:
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            allreduce.finish()       # what is left of the collectives after the backward, + scale + write-back
            a1.record()
            ar_events.append((a0, a1))
        opt.step()
        e1.record()
        step_events.append((e0, e1))
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    from bench import multi_gpu_report
    local_ms = sum(a.elapsed_time(b) for a, b in step_events[-args.steps:]) / max(args.steps, 1)
    mg = multi_gpu_report(torch, dist, world, device, local_ms, ar_events[-args.steps:], "allreduce_tail_ms")
    out = None
    if rank == 0:
        pts = n_global * 256
        achieved = (pts / world) * STEP_FLOP_PER_POINT * args.steps / dt / 1e12      # per GPU: the roofline is one GPU's
        # bf16x3 issues three bf16 MFMAs per f32 product: in f32-equivalent FLOP/s its matrix roofline is a third of the bf16 peak
        peak = PEAK_F32_MFMA_TFLOPS if tprec == "fp32" else PEAK_BF16_MFMA_TFLOPS / 3
        traffic = step_traffic({"fp32": "train", "bf16x3": "train_bf16x3"}.get(tprec, "train_" + tprec))
        out = {
            "metric": "rays/sec at 64+128 samples/ray (train step)",
            "value": n_global * args.steps / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32" if tprec == "fp32" else tprec, "data": "synthetic",
            "config": {"workload": f"lego-shaped training step, {n_global}-ray batch, 64+128 samples, D=8 W=256 coarse+fine, "
                                   f"{tprec}, forward+backward+Adam", "global_batch_rays": n_global,
                       "parallelism": f"dp{world}, two gradient buckets (fine, coarse: 2.38 MB each), the first all-reduce "
                                      "overlapped with the coarse net's backward" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         # the step also moves ~10 KB/point of saved activations and gradients each way: its HBM side
                         "hbm_frac_of_8TBs": (traffic / (dt / args.steps) / 8e12) if traffic and world == 1 else None,
                         "kernel": ("whole step (query_fwd<train> + dgrad + wgrad kernels)" if tprec == "fp32" else
                                    "whole step (query_fwd_bf16x3<train> + dgrad_x3 + wgrad<x3> kernels; FLOP/s f32-equivalent, "
                                    "peak = bf16 peak / 3" + ("; activations saved as f16" if tprec.endswith("a16") else "") + ")") + ", per GPU; algorithmic 3.489 MFLOP/point over wall time; "
                                   "traffic = HBM bytes per 4096-ray STEP over all kernels"},
        }
        out.update(mg)
        if emit:
            print(json.dumps(out), flush=True)
    if world > 1 and emit:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None
