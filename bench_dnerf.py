"""bench.py --workload dnerf: BASELINE.json configs[4] -- D-NeRF deformation chain (vanilla_dynamic): one
800x800 view at time t through render_dnerf(), 64 coarse + 128 fine samples, ONE DNeRF_Deformation model for
both passes (create_dnerf's default, use_two_models_for_fine=False), time net D=8/W=256 on gamma(x) (+) gamma(t),
canonical net D=8/W=256 with view directions.  With N ranks the image rows are sharded as in the render bench."""
import json
import time

import numpy as np

TIME_MAC = (63 + 21) * 256 + 6 * 256 * 256 + (256 + 63) * 256 + 256 * 3          # 497 152
OCC_MAC = 63 * 256 + 6 * 256 * 256 + 319 * 256 + 256 * 256 + 256 + 283 * 128 + 128 * 3   # 593 408
FLOP_PER_POINT = 2 * (TIME_MAC + OCC_MAC)


def run_dnerf(args, torch, dist, world, rank, device, emit=True):
    import nerf_fixtures as fx
    import nerf_amd.vanilla_dynamic as vd
    from bench import PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS
    from nerf_amd import parallel
    from nerf_amd.rays import get_rays

    embed_fn, _ = vd.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = vd.get_embedder(4, 0, input_dims=3)
    embed_time, _ = vd.get_time_embedder(10)
    model = vd.DNeRF_Deformation(D=8, W=256, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                                 input_ch_time=21, output_ch=5, skips=[4], use_viewdirs=True, zero_canonical=True)
    sd = fx.make_dnerf_state(31, peaky=True, dx_scale=3.0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(device)
    model.embed_fn = embed_fn
    model.precision = args.precision
    bf16, x3 = args.precision == "bf16", args.precision == "bf16x3"
    # bf16x3 issues three bf16 MFMAs per f32 product: in f32-equivalent FLOP/s its roofline is a third of the bf16 peak
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else (PEAK_BF16_MFMA_TFLOPS / 3 if x3 else PEAK_F32_MFMA_TFLOPS)
    qfn = lambda i, v, t, f: vd.run_network_dnerf(i, v, t, f, embed_fn, embeddirs_fn, embed_time, 65536)  # noqa: E731
    kw = dict(network_fn=model, network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128, network_fine=None,
              white_bkgd=True, raw_noise_std=0., use_viewdirs=True)
    H = W = args.size
    K, _ = fx.intrinsics(H, W)

    events, enabled, timings = [], [False], {}
    orig = vd.DNeRF_Deformation.query_points

    def timed(self, pts, *a, **k):
        if not enabled[0]:
            return orig(self, pts, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(self, pts, *a, **k)
        e1.record()
        events.append((e0, e1, pts.shape[0]))
        return out
    vd.DNeRF_Deformation.query_points = timed

    def step(i):
        c2w = torch.from_numpy(fx.pose_spherical(-180.0 + 9.0 * i, -30.0, 4.0)[:3, :4]).to(device)
        tval = 0.1 + 0.8 * ((i * 7) % 10) / 10.0
        with torch.no_grad():
            if world == 1 and not getattr(args, "sharded_path", False):
                return vd.render_dnerf(H, W, K, tval, chunk=32768, c2w=c2w, near=2., far=6., **kw)[0]

            def rows(row0, nrows):       # a rank's share: (row range, pose, time, intrinsics) -- no ray data moves
                return parallel.render_rows_dnerf(H, W, K, c2w, tval, row0, nrows, chunk=32768, near=2., far=6., **kw)
            return parallel.render_sharded(rows, H, W, rank, world, timings=timings)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    enabled[0] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    enabled[0] = False
    if world > 1:
        t = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    q_ms = sum(a.elapsed_time(b) for a, b, _ in events)
    from bench import multi_gpu_report
    mg = multi_gpu_report(torch, dist, world, device, q_ms / max(args.steps, 1), timings.get("gather", [])[-args.steps:],
                          "gather_ms")
    out = None
    if rank == 0:
        q_pts = sum(n for _, _, n in events)
        achieved = q_pts * FLOP_PER_POINT / (q_ms * 1e-3) / 1e12 if q_ms > 0 else 0.0
        out = {
            "metric": "rays/sec at 64+128 samples/ray (D-NeRF deformation render)",
            "value": H * W * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if bf16 else ("bf16x3" if x3 else "f32"), "data": "synthetic",
            "config": {"workload": f"D-NeRF bouncingballs-shaped {H}x{W} view, deformation network (time net D=8 W=256 -> "
                                   f"dx -> canonical net D=8 W=256), 64+128 samples, one model for both passes, {args.precision}",
                       "rays_per_step": H * W,
                       "parallelism": f"rows sharded over {world} GPU(s), one gather" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": None,
                         "kernel": ("dnerf_query_fwd_bf16_kernel<8>" if bf16 else
                                    "dnerf_query_fwd_bf16x3_kernel<8>" if x3 else "dnerf_query_fwd_kernel<8>")
                         + " (gamma(x), gamma(t), time net, re-embed, canonical net)",
                         "launches": len(events), "avg_launch_ms": q_ms / max(len(events), 1),
                         "flop_per_point": FLOP_PER_POINT, "query_share_of_step": q_ms * 1e-3 / dt},
        }
        out.update(mg)
        if emit:
            print(json.dumps(out), flush=True)
    if world > 1 and emit:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


STRAIGHT_MAC = 84 * 256 + 6 * 256 * 256 + (256 + 84) * 256 + 256 * 256 + 256 + 283 * 128 + 128 * 3      # 604 160


def run_dnerf_straight(args, torch, device, steps=2, warmup=1, precision="fp32"):
    """One 800x800 D-NeRF view with network_type='straightforward' (eight of the reference's sixteen shipped D-NeRF configs;
    create_dnerf builds a coarse and a fine DNeRF_Straightforward, vanilla_dynamic/model.py:93-99): cat[gamma(x), gamma(t)] ->
    NeRF-shaped MLP on the fused f32 kernel, 64 + 128 samples.  Single GPU; returns the summary dict."""
    import nerf_fixtures as fx
    import nerf_amd.vanilla_dynamic as vd
    from bench import PEAK_F32_MFMA_TFLOPS

    embed_fn, _ = vd.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = vd.get_embedder(4, 0, input_dims=3)
    embed_time, _ = vd.get_time_embedder(10)
    nets = []
    for seed in (61, 62):
        m = vd.DNeRF_Straightforward(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=5, skips=[4],
                                     use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_straightforward_state(seed).items()}, strict=True)
        m.precision = precision
        nets.append(m.to(device))
    qfn = lambda i, v, t, f: vd.run_network_dnerf(i, v, t, f, embed_fn, embeddirs_fn, embed_time, 65536,  # noqa: E731
                                                  network_type='straightforward')
    kw = dict(network_fn=nets[0], network_fine=nets[1], network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128,
              white_bkgd=True, raw_noise_std=0., use_viewdirs=True, network_type='straightforward')
    H = W = args.size
    K, _ = fx.intrinsics(H, W)

    def step(i):
        c2w = torch.from_numpy(fx.pose_spherical(-180.0 + 9.0 * i, -30.0, 4.0)[:3, :4]).to(device)
        with torch.no_grad():
            return vd.render_dnerf(H, W, K, 0.1 + 0.08 * i, chunk=32768, c2w=c2w, near=2., far=6., **kw)[0]
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = H * W * 256 * 2 * STRAIGHT_MAC / dt / 1e12
    from bench import PEAK_BF16_MFMA_TFLOPS
    peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3, "bf16": PEAK_BF16_MFMA_TFLOPS}[precision]
    return {"value": H * W / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt, "dtype": "f32" if precision == "fp32" else precision,
            "roofline_frac": tf / peak, "achieved_TFLOPs": tf,
            "workload": f"D-NeRF {H}x{W} view, network_type='straightforward' (coarse + fine DNeRF_Straightforward D=8 W=256), "
                        f"64+128 samples, {precision}; FLOP/s over the whole frame's wall time (1.208 MFLOP/point"
                        + (", f32-equivalent, peak = bf16 peak / 3)" if precision == "bf16x3" else ")")}


def run_dnerf_straight_train(args, torch, device, n_rays=1024, steps=10, warmup=3, fused=True):
    """One training step with network_type='straightforward' (configs/*_straightforward.txt: N_rand 1024, 64+128 samples,
    coarse + fine DNeRF_Straightforward, grad_clip 1.0): render_dnerf on a ray batch with per-ray times -> mse(rgb) +
    mse(rgb0) -> backward -> clipped fused Adam.  fused=False: the same step with the networks composed from GPU torch
    ops (what nerf_amd ran before the training kernels took this network), for the comparison."""
    import nerf_fixtures as fx
    import nerf_amd.vanilla_dynamic as vd
    from bench import PEAK_F32_MFMA_TFLOPS
    from nerf_amd.optim import FusedAdam
    from nerf_amd.utils import img2mse

    embed_fn, _ = vd.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = vd.get_embedder(4, 0, input_dims=3)
    embed_time, _ = vd.get_time_embedder(10)
    nets = []
    for seed in (61, 62):
        m = vd.DNeRF_Straightforward(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=5, skips=[4],
                                     use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_straightforward_state(seed).items()}, strict=True)
        if not fused:
            m._kernel_takes = lambda: False
        nets.append(m.to(device))
    opt = FusedAdam([p for m in nets for p in m.parameters()], lr=5e-4, betas=(0.9, 0.999))
    qfn = lambda i, v, t, f: vd.run_network_dnerf(i, v, t, f, embed_fn, embeddirs_fn, embed_time, 65536,  # noqa: E731
                                                  network_type='straightforward')
    kw = dict(network_fn=nets[0], network_fine=nets[1], network_query_fn=qfn, N_samples=64, perturb=1., N_importance=128,
              white_bkgd=True, raw_noise_std=0., use_viewdirs=True, network_type='straightforward')
    H = W = 400
    K, _ = fx.intrinsics(H, W)
    ro_np, rd_np, target_np = fx.train_batch(n_rays=n_rays * (steps + warmup), H=H, W=W, seed=7)
    ro_all, rd_all = torch.from_numpy(ro_np).to(device), torch.from_numpy(rd_np).to(device)
    target_all = torch.from_numpy(target_np).to(device)
    times_all = torch.rand(n_rays * (steps + warmup), 1, device=device) * 0.9 + 0.05

    def step(i):
        sl = slice(i * n_rays, (i + 1) * n_rays)
        rays = torch.stack([ro_all[sl], rd_all[sl]])
        rgb, disp, acc, extras = vd.render_dnerf(H, W, K, times_all[sl], chunk=32768, rays=rays, near=2., far=6.,
                                                 retraw=True, **kw)
        opt.zero_grad()
        loss = img2mse(rgb, target_all[sl]) + img2mse(extras['rgb0'], target_all[sl])
        loss.backward()
        opt.step(max_grad_norm=1.0)          # vanilla_dynamic/train.py:273-274

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mac = 3 * STRAIGHT_MAC - (84 * 256 + 84 * 256) - 27 * 128      # no input-side activation gradients
    tf = n_rays * 256 * 2 * mac / dt / 1e12
    return {"value": n_rays / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt, "dtype": "f32",
            "roofline_frac": tf / PEAK_F32_MFMA_TFLOPS, "achieved_TFLOPs": tf,
            "workload": f"D-NeRF straightforward training step, {n_rays}-ray batch with per-ray times, 64+128 samples, coarse + "
                        f"fine DNeRF_Straightforward D=8 W=256, fp32, forward+backward+clipped fused Adam"
                        + ("" if fused else " -- networks composed from GPU torch ops (comparison)")}


# fwd + bwd of the chain per point: both nets' forward, weight and activation gradients, minus the input-side
# activation gradients that nothing consumes (time net: gamma(x), gamma(t) of layer 0 and the skip slice; canonical
# net: the view slice) -- d/d(gamma(x')) of the canonical net IS consumed (it drives the time net).
TRAIN_MAC = 3 * (TIME_MAC + OCC_MAC) - ((63 + 21) * 256 + 63 * 256) - 27 * 128


def run_dnerf_train(args, torch, device, n_rays=1024, steps=10, warmup=3, train_precision="fp32"):
    """One D-NeRF training step as vanilla_dynamic/train.py:237-277 runs it (bouncingballs_deform.txt: N_rand 1024,
    64+128 samples, one DNeRF_Deformation for both passes, grad_clip 1.0): render_dnerf on a ray batch with per-ray
    times -> mse(rgb) + mse(rgb0) -> backward through the canonical net into the deformation net -> clipped fused Adam.
    Single GPU; returns the summary dict."""
    import nerf_fixtures as fx
    import nerf_amd.vanilla_dynamic as vd
    from bench import PEAK_F32_MFMA_TFLOPS
    from nerf_amd.optim import FusedAdam
    from nerf_amd.utils import img2mse

    embed_fn, _ = vd.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = vd.get_embedder(4, 0, input_dims=3)
    embed_time, _ = vd.get_time_embedder(10)
    model = vd.DNeRF_Deformation(D=8, W=256, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                                 input_ch_time=21, output_ch=5, skips=[4], use_viewdirs=True, zero_canonical=True)
    sd = fx.make_dnerf_state(31, peaky=True, dx_scale=3.0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(device)
    model.embed_fn = embed_fn
    model.train_precision = train_precision
    params = list(model.parameters())
    opt = FusedAdam(params, lr=5e-4, betas=(0.9, 0.999))
    qfn = lambda i, v, t, f: vd.run_network_dnerf(i, v, t, f, embed_fn, embeddirs_fn, embed_time, 65536)  # noqa: E731
    kw = dict(network_fn=model, network_query_fn=qfn, N_samples=64, perturb=1., N_importance=128, network_fine=None,
              white_bkgd=True, raw_noise_std=0., use_viewdirs=True)
    H = W = 400
    K, _ = fx.intrinsics(H, W)
    ro_np, rd_np, target_np = fx.train_batch(n_rays=n_rays * (steps + warmup), H=H, W=W, seed=7)
    ro_all, rd_all = torch.from_numpy(ro_np).to(device), torch.from_numpy(rd_np).to(device)
    target_all = torch.from_numpy(target_np).to(device)
    times_all = torch.rand(n_rays * (steps + warmup), 1, device=device) * 0.9 + 0.05

    def step(i):
        sl = slice(i * n_rays, (i + 1) * n_rays)
        rays = torch.stack([ro_all[sl], rd_all[sl]])
        rgb, disp, acc, extras = vd.render_dnerf(H, W, K, times_all[sl], chunk=32768, rays=rays, near=2., far=6.,
                                                 retraw=True, **kw)
        opt.zero_grad()
        loss = img2mse(rgb, target_all[sl]) + img2mse(extras['rgb0'], target_all[sl])
        loss.backward()
        opt.step(max_grad_norm=1.0)          # vanilla_dynamic/train.py:273-274
        opt.param_groups[0]['lr'] = 5e-4 * 0.1 ** (i / 500000.0)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pts = n_rays * (64 + 192)
    tf = pts * 2 * TRAIN_MAC / dt / 1e12
    return {"value": n_rays / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt,
            "dtype": "f32" if train_precision == "fp32" else "f32 chains + bf16x3 canonical forward and weight gradients",
            # mixed arithmetic has no single matrix roofline: the fraction is quoted for the exact-f32 step only
            "roofline_frac": tf / PEAK_F32_MFMA_TFLOPS if train_precision == "fp32" else None, "achieved_TFLOPs": tf,
            "workload": f"D-NeRF deformation training step, {n_rays}-ray batch with per-ray times, 64+128 samples, time net + "
                        "canonical net D=8 W=256, fp32, forward+backward+clipped fused Adam"}
