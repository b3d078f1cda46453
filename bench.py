#!/usr/bin/env python3
"""bench.py -- rays/sec of the NeRF per-ray rendering path at 64 coarse + 128 fine samples per ray.

    python bench.py --gpus N --steps K --warmup W [--workload render|fern|train|dnerf] [--precision fp32|bf16|bf16x3] [--size 800]

A step = one full render of a synthetic 800x800 view (BASELINE.json configs[1]: D=8/W=256 coarse+fine
nets, multires 10/4, view directions, white background, near/far 2/6, chunk 32768, perturb=0) through
nerf_amd.render.render().  Weights are seeded random-init (+ the "peaky" head scaling of SURVEY 8c),
poses come from the spherical-pose formula: data is synthetic, there is no dataset offline.
With N > 1 the image's rows are sharded over N ranks (one process per GPU) and one gather returns (rgb, disp, acc)
to rank 0: strong scaling of one image.  `python bench.py --gpus N` starts the ranks itself (torch.distributed.run
as a child process); under an external launcher (WORLD_SIZE set) it runs as one rank and checks WORLD_SIZE == N.

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel (the fused PE+MLP query,
MFMA-bound) from HIP-event durations of every query launch inside the timed region; `cpu_baseline` is
the C oracle (a port of the reference's CPU path) timed on this box's host cores on a bounded ray sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# the host driver of this pool only supports dmabuf IPC: without this RCCL's peer setup fails in hipIpcGetMemHandle
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_POINT = 2 * (63 * 256 + 6 * 256 * 256 + 319 * 256 + 256 * 256 + 256 + 283 * 128 + 128 * 3)  # 1 186 816
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="render", choices=["render", "fern", "train", "dnerf"])
    ap.add_argument("--size", type=int, default=800, help="image side (render workload)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "f16", "bf16x3"],
                    help="fp32 = the reference's arithmetic (the headline number); bf16 = the optional bf16 MFMA variant; "
                         "f16 = the same 16-bit kernel with f16 hidden operands (saturating)")
    ap.add_argument("--train-rays", type=int, default=4096, help="global ray batch of the train workload")
    ap.add_argument("--train-precision", default="fp32", choices=["fp32", "bf16x3", "bf16x3_a16", "bf16x3_w16"],
                    help="train workload: fp32 = the reference's arithmetic; bf16x3 = three bf16 MFMAs per f32 product "
                         "(f32-grade results, gradients within the same 2e-3 of the reference's autograd)")
    ap.add_argument("--n-importance", type=int, default=128,
                    help="fine samples per ray (render / fern workloads): 128 = the metric; 64 = what configs/fern.txt ships")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-path", action="store_true", help="use the multi-GPU row-sharded code path even with one GPU")
    ap.add_argument("--graph", action="store_true",
                    help="render workload, N=1: replay the frame as one captured HIP graph (nerf_amd.render.RenderGraph)")
    ap.add_argument("--no-extras", action="store_true", help="skip the short secondary measurements (N=1 only)")
    ap.add_argument("--cpu-rays", type=int, default=32768, help="rays in the bounded CPU-baseline sample")
    return ap.parse_args()


def build_nets(torch, device):
    import nerf_fixtures as fx
    from nerf_amd.network import NeRF
    nets = []
    for seed in (21, 22):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        sd = fx.make_nerf_state(seed, peaky=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net.to(device))
    return nets


class QueryTimer:
    """Brackets every fused-query launch with HIP events (on torch's current stream, the one the C ABI
    launches on) while enabled; durations are read after the timed region."""

    def __init__(self, torch):
        self.torch = torch
        self.enabled = False
        self.events = []

    def install(self):
        from nerf_amd import ops
        timer, torch = self, self.torch

        def wrap(cls):
            orig = cls.query

            def timed(pk, pts, viewdirs=None, pts_per_ray=1):
                if not timer.enabled:
                    return orig(pk, pts, viewdirs, pts_per_ray)
                pk.refresh()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig(pk, pts, viewdirs, pts_per_ray)
                e1.record()
                timer.events.append((e0, e1, pts.shape[0]))
                return out
            cls.query = timed
        wrap(ops.PackedNet)
        wrap(ops.PackedNetBf16)     # PackedNetF16 inherits the wrapped query
        wrap(ops.PackedNetBf16x3)

    def summary(self):
        ms = sum(a.elapsed_time(b) for a, b, _ in self.events)
        pts = sum(n for _, _, n in self.events)
        return ms, pts, len(self.events)


def _kernel_entry(table, kernel):
    """table[kernel], or the entry whose name continues `kernel`'s template argument list (kernels gain trailing template
    parameters between rounds: query_fwd_kernel<8, false, false, false> is ...<8, false, false, false, false, false> now)."""
    if kernel in table:
        return table[kernel]
    stem = kernel[:-1] + ","
    for k in sorted(table):
        if k.startswith(stem):
            return table[k]
    raise KeyError(kernel)


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*_pmc_traffic.json, latest round;
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same command, FETCH doubled as the
    gfx950 note in MI355X_MICROARCH.md prescribes).  None when no PMC file is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return _kernel_entry(json.load(open(files[-1]))[workload], kernel)["hbm_bytes"]
    except (KeyError, ValueError):
        return None


def pmc_mfma_util(workload, kernel):
    """Matrix-pipe busy fraction and held shader clock of `kernel` from the committed SQ pass (profiles/*_mfma_util.json,
    latest round; scripts/profile_gpu.sh sq_all -> scripts/collect_profiles.py).  {} when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util.json")))
    if not files:
        return {}
    try:
        e = _kernel_entry(json.load(open(files[-1]))[workload], kernel)
        return {"mfma_busy_frac": round(e["mfma_busy_frac"], 4), "clock_ghz": round(e.get("clock_ghz", 0.0), 3),
                "mfma_util_source": os.path.basename(files[-1])}
    except (KeyError, ValueError):
        return {}


def multi_gpu_report(torch, dist, world, device, local_ms, events, key):
    """What an N > 1 line says about itself: `world` as the process group sees it, each rank's kernel time per step (query
    launches for the render workloads, the whole step for training) and the collective's time per step on rank 0 (HIP
    events around the gather / all-reduce wait).  Collective call on every rank; {} at N = 1."""
    if world == 1:
        return {}
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([local_ms], device=device if on_gpu else "cpu", dtype=torch.float64)
    per = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(per, t)
    ms = [a.elapsed_time(b) for a, b in events]
    return {"world": dist.get_world_size(), "backend": dist.get_backend(),
            "per_rank_ms_per_step": [round(float(x.item()), 3) for x in per],
            key: round(sum(ms) / max(len(ms), 1), 3) if ms else None}


# parameter tensors of one D=8 / W=256 NeRF with view directions (network.py:29-44): weights and biases, 595 844 in all
_NERF_PARAM_SIZES = [63 * 256, 256] + [256 * 256, 256] * 4 + [319 * 256, 256] + [256 * 256, 256] * 2 + \
    [256 * 256, 256, 256, 1, 283 * 128, 128, 128 * 3, 3]


def _us_per_launch(torch, fn, reps):
    """Microseconds per launch of `fn`: `reps` launches captured ONCE as a HIP graph and replayed between two events --
    the kernels' own time.  (An eager Python loop adds ~10-40 us of host time per call -- ctypes tables, torch.empty --
    which is several times what these kernels take: the round-2 lines, timed that way, priced the host.)  The eager loop
    remains the fallback when capture is refused."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps, "graph replay"
    except Exception:      # noqa: BLE001 -- any capture refusal: time the eager loop instead and say so
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps, "eager loop (host-bound)"


def hbm_kernels(torch, device, n_rays=32768, reps=20):
    """Achieved GB/s of the sampling / compositing kernels at the render chunk size: algorithmic bytes per launch
    (inputs read once + outputs written once) over the HIP-event duration of `reps` launches replayed as one graph."""
    import nerf_fixtures as fx
    from nerf_amd import ops
    rays = torch.from_numpy(fx.blender_rays(n_rays, 3)).to(device)
    z64 = ops.sample_coarse(rays, 64)
    w64 = torch.rand(n_rays, 64, device=device)
    _, z192, _ = ops.sample_fine(z64, w64, 128, None, want_samples=False)
    raw192 = torch.randn(n_rays, 192, 4, device=device)
    ncol = rays.shape[1]
    cases = {
        "sample_coarse_kernel (S=64)": (lambda: ops.sample_coarse(rays, 64), 4 * (ncol + 64)),
        "points_kernel (S=192)": (lambda: ops.points(rays, z192), 4 * (ncol + 192 + 3 * 192)),
        "sample_fine_kernel (64 -> 64+128)": (lambda: ops.sample_fine(z64, w64, 128, None, want_samples=False),
                                              4 * (64 + 64 + 192 + 1)),
        "composite_fwd_kernel (S=192)": (lambda: ops.composite_fwd(raw192, z192, rays[:, 3:6], None, True),
                                         4 * (4 * 192 + 192 + 3 + 3 + 1 + 1 + 192 + 1)),
    }
    g_rgb = torch.randn(n_rays, 3, device=device)
    # the backward recomputes the forward's quantities: it reads raw, z, d and the rgb gradient, writes the raw gradient
    cases["composite_bwd_kernel (S=192)"] = (lambda: ops.composite_bwd(raw192, z192, rays[:, 3:6], g_rgb, white_bkgd=True),
                                             4 * (4 * 192 + 192 + 3 + 3 + 4 * 192))
    out = {}
    for name, (fn, bytes_per_ray) in cases.items():
        us, how = _us_per_launch(torch, fn, reps)
        gbs = bytes_per_ray * n_rays / (us * 1e-6) / 1e9
        out[name] = {"us_per_launch": round(us, 2), "bytes_per_ray": bytes_per_ray, "GB/s": round(gbs, 1),
                     "frac_of_8TB/s": round(gbs / 8000.0, 4), "timed_as": how}
    # the fused Adam step over both nets' parameters: reads p, g, m, v and writes p, m, v = 28 B/parameter
    from nerf_amd.optim import FusedAdam
    params = [torch.nn.Parameter(torch.randn(n, device=device)) for n in _NERF_PARAM_SIZES * 2]
    for p in params:
        p.grad = torch.randn_like(p)
    opt = FusedAdam(params, lr=5e-4)
    us, how = _us_per_launch(torch, opt.step, reps)
    nparam = sum(p.numel() for p in params)
    gbs = 28 * nparam / (us * 1e-6) / 1e9
    out["adam_kernel (both nets, 1 191 688 parameters in 48 tensors)"] = {
        "us_per_launch": round(us, 2), "bytes_per_parameter": 28, "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / 8000.0, 4),
        "timed_as": how}
    return out


def extras(args, torch, dist, device, coarse, fine, step, timer):
    """Short secondary measurements on the same GPU, reported beside the headline (never instead of it): the bf16
    variant of the same render, the 4096-ray training step, and the D-NeRF deformation render in fp32 and bf16."""
    import copy
    import bench_dnerf
    import bench_train
    also = {}

    def brief(o):
        return {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "dtype": o["dtype"],
                "roofline_frac": o["roofline"]["frac"], "achieved_TFLOPs": o["roofline"]["achieved"],
                "workload": o["config"]["workload"]}
    # the same 800x800 render with bf16 MFMA
    coarse.precision = fine.precision = "bf16"
    step(0)
    torch.cuda.synchronize()
    timer.events.clear()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(3):
        step(i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    timer.enabled = False
    q_ms, q_pts, q_n = timer.summary()
    tf = q_pts * FLOP_PER_POINT / (q_ms * 1e-3) / 1e12
    also["render_bf16"] = {"value": args.size * args.size / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt, "dtype": "bf16",
                           "roofline_frac": tf / PEAK_BF16_MFMA_TFLOPS, "achieved_TFLOPs": tf,
                           "workload": f"the same {args.size}x{args.size} render, bf16 MFMA variant (tolerance 1e-2)"}
    # the 16-bit kernel with f16 hidden operands (saturating): same rate class, 8x finer rounding
    coarse.precision = fine.precision = "f16"
    step(0)
    torch.cuda.synchronize()
    timer.events.clear()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(3):
        step(i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    timer.enabled = False
    q_ms, q_pts, q_n = timer.summary()
    tf = q_pts * FLOP_PER_POINT / (q_ms * 1e-3) / 1e12
    also["render_f16"] = {"value": args.size * args.size / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt, "dtype": "f16",
                          "roofline_frac": tf / PEAK_BF16_MFMA_TFLOPS, "achieved_TFLOPs": tf,
                          "workload": f"the same {args.size}x{args.size} render, 16-bit kernel with f16 hidden operands "
                                      "(v_mfma_f32_32x32x16_f16, saturating conversion; tolerance 1e-2)"}
    # and at f32-grade accuracy on the bf16 pipe (bf16x3)
    coarse.precision = fine.precision = "bf16x3"
    step(0)
    torch.cuda.synchronize()
    timer.events.clear()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(3):
        step(i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    timer.enabled = False
    q_ms, q_pts, q_n = timer.summary()
    tf = q_pts * FLOP_PER_POINT / (q_ms * 1e-3) / 1e12
    also["render_bf16x3"] = {"value": args.size * args.size / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt, "dtype": "bf16x3",
                             "roofline_frac": tf / (PEAK_BF16_MFMA_TFLOPS / 3), "achieved_TFLOPs": tf,
                             "workload": f"the same {args.size}x{args.size} render, three bf16 MFMAs per f32 product on "
                                         "hi/residual operands (meets the f32 tolerance of 1e-4; FLOP/s f32-equivalent)"}
    # BASELINE.json configs[3]: the LLFF-fern-shaped 1008x756 NDC view, at the metric's 64+128 samples in all three
    # arithmetics and at the 64+64 configs/fern.txt:10-11 ships (SURVEY 8d asks for both)
    for prec, n_imp, key in (("fp32", 128, "fern_ndc_fp32"), ("bf16x3", 128, "fern_ndc_bf16x3"), ("bf16", 128, "fern_ndc_bf16"),
                             ("fp32", 64, "fern_ndc_fp32_64+64")):
        coarse.precision = fine.precision = prec
        fH, fW, fstep = make_render_step(args, torch, device, coarse, fine, True, n_imp)
        fstep(0)
        torch.cuda.synchronize()
        timer.events.clear()
        timer.enabled = True
        t0 = time.perf_counter()
        for i in range(3):
            fstep(i + 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        timer.enabled = False
        q_ms, q_pts, q_n = timer.summary()
        tf = q_pts * FLOP_PER_POINT / (q_ms * 1e-3) / 1e12
        pk = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3}[prec]
        also[key] = {"value": fH * fW / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt, "dtype": "f32" if prec == "fp32" else prec,
                     "roofline_frac": tf / pk, "achieved_TFLOPs": tf, "points_per_launch": q_pts / max(q_n, 1),
                     "workload": f"LLFF-fern-shaped {fW}x{fH} NDC render, 64+{n_imp} samples, D=8 W=256 coarse+fine, {prec}"}
    timer.events.clear()
    coarse.precision = fine.precision = "fp32"
    a = copy.copy(args)
    a.steps, a.warmup = 10, 3
    also["train_step"] = brief(bench_train.run_train(a, torch, dist, 1, 0, device, emit=False))
    a.train_precision = "bf16x3"
    also["train_step_bf16x3"] = brief(bench_train.run_train(a, torch, dist, 1, 0, device, emit=False))
    a.train_precision = "bf16x3_a16"      # the same with the saved activations in f16 (weight gradients within ~2e-4)
    also["train_step_bf16x3_a16"] = brief(bench_train.run_train(a, torch, dist, 1, 0, device, emit=False))
    a.train_precision = "bf16x3_w16"      # that plus scaled f16 gradient blocks: weight-gradient GEMMs as single f16 MFMAs
    also["train_step_bf16x3_w16"] = brief(bench_train.run_train(a, torch, dist, 1, 0, device, emit=False))
    a.train_precision = "fp32"
    a.steps, a.warmup = 2, 1
    for prec in ("fp32", "bf16", "bf16x3"):
        a.precision = prec
        also[f"dnerf_{prec}"] = brief(bench_dnerf.run_dnerf(a, torch, dist, 1, 0, device, emit=False))
    also["dnerf_straightforward_fp32"] = bench_dnerf.run_dnerf_straight(a, torch, device)
    also["dnerf_straightforward_bf16x3"] = bench_dnerf.run_dnerf_straight(a, torch, device, precision="bf16x3")
    also["dnerf_straightforward_bf16"] = bench_dnerf.run_dnerf_straight(a, torch, device, precision="bf16")
    also["dnerf_straightforward_train_step"] = bench_dnerf.run_dnerf_straight_train(a, torch, device)
    also["dnerf_train_step"] = bench_dnerf.run_dnerf_train(a, torch, device)
    also["dnerf_train_step_bf16x3"] = bench_dnerf.run_dnerf_train(a, torch, device, train_precision="bf16x3")
    return also


def make_render_step(args, torch, device, coarse, fine, fern, n_importance=128, world=1, rank=0, timings=None):
    """(H, W, step): step(i) renders pose i of the workload -- the lego-shaped size x size view (configs[1]) or the
    LLFF-fern-shaped 1008x756 NDC view (configs[3]: factor 4, forward-facing pose, no white background) -- through
    nerf_amd.render.render(), or as this rank's row shard + one gather when world > 1."""
    import nerf_fixtures as fx
    from nerf_amd import parallel
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render, run_network
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=n_importance,
              network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True,
              raw_noise_std=0., ndc=False, lindisp=False)
    H = W = args.size
    K, _ = fx.intrinsics(H, W)
    if fern:
        H, W = 756, 1008
        K = np.array([[815.0, 0, 0.5 * W], [0, 815.0, 0.5 * H], [0, 0, 1]], dtype=np.float32)
        kw.update(ndc=True, white_bkgd=False)
    near, far = (0., 1.) if fern else (2., 6.)
    frame_graph = []

    def step(i):
        if fern:   # identity rotation, small lateral motion per step
            c2w = torch.eye(4)[:3, :4].clone()
            c2w[0, 3] = 0.02 * i
            c2w = c2w.to(device)
        else:
            c2w = torch.from_numpy(fx.pose_spherical(-180.0 + 9.0 * i, -30.0, 4.0)[:3, :4]).to(device)
        with torch.no_grad():
            if world == 1 and args.graph:
                if not frame_graph:
                    from nerf_amd.render import RenderGraph
                    frame_graph.append(RenderGraph(H, W, K, chunk=32768, near=near, far=far, **kw))
                return frame_graph[0](c2w)[0]
            if world == 1 and not args.sharded_path:
                rgb, disp, acc, _ = render(H, W, K, chunk=32768, c2w=c2w, near=near, far=far, **kw)
                return rgb

            def rows(row0, nrows):
                kw2 = {k: v for k, v in kw.items() if k not in ("ndc", "use_viewdirs")}
                return parallel.render_rows(H, W, K, c2w, row0, nrows, chunk=32768, ndc=kw["ndc"], near=near, far=far,
                                            use_viewdirs=True, **kw2)
            return parallel.render_sharded(rows, H, W, rank, world, timings=timings)
    return H, W, step


def cpu_baseline(n_rays):
    """The oracle (C restatement of the reference's CPU path) on a bounded sample of the same workload."""
    import nerf_fixtures as fx
    from oracle import oracle as orc
    coarse = orc.Net(fx.make_nerf_state(21, peaky=True))
    fine = orc.Net(fx.make_nerf_state(22, peaky=True))
    H = W = 800
    K, _ = fx.intrinsics(H, W)
    rng = np.random.RandomState(0)
    pix = np.stack([rng.randint(0, H, n_rays), rng.randint(0, W, n_rays)], -1)
    ro, rd = fx.pinhole_rays_np(H, W, K, fx.pose_spherical(30.0, -30.0, 4.0), pix)
    vd = (rd / np.linalg.norm(rd, axis=-1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([ro, rd, np.full((n_rays, 1), 2, np.float32), np.full((n_rays, 1), 6, np.float32), vd], -1)
    orc.render_rays(rays[:64], coarse, fine, 64, 128, False, True)  # warm-up (thread pool, page-in)
    t0 = time.time()
    orc.render_rays(rays, coarse, fine, 64, 128, False, True)
    dt = time.time() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": n_rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} random rays of the 800x800 view, 64+128 samples, oracle/liborc.so (OpenMP), {dt:.1f} s"}


def needs_launch(gpus, env):
    """True when this process is the PARENT of an N > 1 run: --gpus N asked for more than one rank and no launcher has set
    WORLD_SIZE yet (the driver's `python bench.py --gpus N` form)."""
    return gpus > 1 and "WORLD_SIZE" not in env


def launcher_command(gpus, argv, port, script=None):
    """The child command of an N > 1 run: torch.distributed.run with one rank per GPU on this node, same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(gpus, argv, script=None, env=None):
    """`python bench.py --gpus N` without an external launcher: start the N ranks as a CHILD process (never exec: this
    process may not be replaced once anything has touched the GPU, and nothing here has), relay the ranks' stdout
    (rank 0's JSON line) and return the child's exit code -- non-zero as well when no line with n_gpus == N came back,
    so an N-GPU run can never be mistaken for the one-GPU measurement.  No torch.cuda call happens in the parent."""
    import subprocess
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = launcher_command(gpus, argv, _free_port(), script)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    seen = False
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{"):
            try:
                seen = seen or json.loads(line).get("n_gpus") == gpus
            except ValueError:
                pass
    rc = proc.wait()
    if rc == 0 and not seen:
        print(f"bench.py: the {gpus}-rank run ended without a result line carrying n_gpus == {gpus}", file=sys.stderr)
        return 3
    return rc


def check_world(gpus, world):
    """Every rank: the process group's size IS the --gpus argument, or the run stops before it measures anything."""
    if world != gpus:
        print(f"bench.py: --gpus {gpus} but WORLD_SIZE is {world}: refusing to report a {world}-rank run as {gpus} GPUs "
              f"(start `python bench.py --gpus {gpus}` without a launcher, or give the launcher --nproc-per-node {gpus})",
              file=sys.stderr)
        sys.exit(2)


def main():
    args = parse()
    if needs_launch(args.gpus, os.environ):
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    check_world(args.gpus, world)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NRF_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives staged by gloo); the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("NRF_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {torch.cuda.device_count()} "
              "(NRF_BENCH_BACKEND=gloo rehearses the multi-rank path with ranks sharing a GPU)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if world > 1 else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    import nerf_fixtures as fx
    from nerf_amd import parallel
    from nerf_amd.positional_encoding import get_embedder
    parallel.warm_up(device)
    from nerf_amd.rays import get_rays
    from nerf_amd.render import render, run_network

    if args.workload == "train":
        from bench_train import run_train
        return run_train(args, torch, dist, world, rank, device)
    if args.workload == "dnerf":
        from bench_dnerf import run_dnerf
        return run_dnerf(args, torch, dist, world, rank, device)

    coarse, fine = build_nets(torch, device)
    coarse.precision = fine.precision = args.precision
    bf16 = args.precision in ("bf16", "f16")     # the 16-bit kernels: same matrix rate (bf16 / f16 MFMA), same roofline
    x3 = args.precision == "bf16x3"
    # bf16x3 issues three bf16 MFMAs per f32 product: its roofline in f32-equivalent FLOP/s is a third of the bf16 peak
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else (PEAK_BF16_MFMA_TFLOPS / 3 if x3 else PEAK_F32_MFMA_TFLOPS)
    fern = args.workload == "fern"
    timer = QueryTimer(torch)
    timer.install()
    timings = {}
    H, W, step = make_render_step(args, torch, device, coarse, fine, fern, args.n_importance, world, rank, timings)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    mg = multi_gpu_report(torch, dist, world, device, timer.summary()[0] / max(args.steps, 1),
                          timings.get("gather", [])[-args.steps:], "gather_ms")
    if rank == 0:
        n_rays = H * W
        q_ms, q_pts, q_n = timer.summary()
        if args.graph and world == 1:
            # a replayed graph has no per-launch events: price the query kernel against the WHOLE frame time (a lower
            # bound of its rate; the default, ungraphed run measures the kernel itself)
            q_ms, q_pts, q_n = 1e3 * dt, n_rays * (64 + 64 + args.n_importance) * args.steps, args.steps
        achieved = q_pts * FLOP_PER_POINT / (q_ms * 1e-3) / 1e12 if q_ms > 0 else 0.0
        out = {
            "metric": f"rays/sec at 64+{args.n_importance} samples/ray (render)",
            "value": n_rays * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.precision if bf16 else ("bf16x3" if x3 else "f32"),
            "data": "synthetic",
            "config": {"workload": (f"LLFF-fern-shaped {W}x{H} NDC render" if fern else f"lego-shaped {H}x{W} test-view render")
                       + f", 64+{args.n_importance} samples, D=8 W=256 coarse+fine, {args.precision}, chunk 32768",
                       "rays_per_step": n_rays,
                       "parallelism": f"rows sharded over {world} GPU(s), one gather" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "traffic": pmc_traffic("render_bf16" if bf16 else ("render_bf16x3" if x3 else "render"),
                                                "query_fwd_bf16_kernel<8, true>" if bf16
                                                else ("query_fwd_bf16x3_kernel<8, true>" if x3
                                                      else "query_fwd_kernel<8, false, false, false>")),
                         "kernel": ("query_fwd_bf16_kernel<8,true,F16> (fused PE + MLP, f16 MFMA with saturating conversion, "
                                    "LDS-shared weight stream)" if args.precision == "f16" else
                                    "query_fwd_bf16_kernel<8,true> (fused PE + MLP, bf16 MFMA, LDS-shared weight stream)") if bf16
                         else ("query_fwd_bf16x3_kernel<8,true> (fused PE + MLP, 3 bf16 MFMAs per f32 product; FLOP/s are "
                               "f32-equivalent, peak = bf16 peak / 3)" if x3
                               else "query_fwd_kernel<8,false> (fused PE + MLP, f32 MFMA)"),
                         "launches": q_n, "avg_launch_ms": q_ms / max(q_n, 1),
                         "flop_per_point": FLOP_PER_POINT, "points_per_launch": q_pts / max(q_n, 1),
                         "query_share_of_step": q_ms * 1e-3 / dt},
        }
        out.update(mg)
        out["roofline"].update(pmc_mfma_util("render_bf16" if bf16 else ("render_bf16x3" if x3 else "render"),
                                             "query_fwd_bf16_kernel<8, true>" if bf16
                                             else ("query_fwd_bf16x3_kernel<8, true>" if x3
                                                   else "query_fwd_kernel<8, false, false, false>")))
        out["hbm_kernels"] = hbm_kernels(torch, device)
        if world == 1 and not args.no_extras and args.workload == "render" and not bf16 and not x3:
            out["also"] = extras(args, torch, dist, device, coarse, fine, step, timer)
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is a rank-0, N = 1 measurement
            out["cpu_baseline"] = cpu_baseline(args.cpu_rays)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
