#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the reference (THIS container only).

    python tests/golden/make_golden.py static     -> tests/golden/static.npz
    python tests/golden/make_golden.py dnerf      -> tests/golden/dnerf.npz
    python tests/golden/make_golden.py tooling    -> tests/golden/tooling.npz + tests/golden/ref_ckpt_small.tar
    python tests/golden/make_golden.py static2    -> tests/golden/static2.npz  (configs[0] at 100x100; output_ch = 5 nets)
    python tests/golden/make_golden.py dnerf2     -> tests/golden/dnerf2.npz   (well-conditioned D-NeRF, LayerNorm, straightforward)
    python tests/golden/make_golden.py trajectory -> tests/golden/trajectory.npz (30 steps of the reference's training loop)
    python tests/golden/make_golden.py trajectory_dnerf -> tests/golden/trajectory_dnerf.npz (20 steps of the D-NeRF loop)
    python tests/golden/make_golden.py static3    -> tests/golden/static3.npz (configs[0]: the coarse weights sample_pdf saw)
    python tests/golden/make_golden.py trained_scene -> tests/golden/trained_scene.npz (trained weights + the reference's image)

The reference (a pure-Python/PyTorch project at /root/reference) cannot travel to the GPU box, so
its outputs on seeded synthetic inputs are frozen here as data.  Inputs come from
tests/nerf_fixtures.py (numpy, seed-driven) so the tests can rebuild the weights instead of
storing 2.4 MB per network.  `imageio` (PNG/MP4 writing only; render.py:6) is absent from the image
and is replaced by an empty module object before import; nothing on the measured path touches it.
Static and D-NeRF stacks use colliding bare module names (render, network, ...) and therefore run
in separate processes.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import nerf_fixtures as fx  # noqa: E402

REF = "/root/reference"


def _meta(torch):
    return np.array([torch.__version__, torch.backends.cpu.get_cpu_capability(),
                     str(torch.get_num_threads())])


def _load(torch, module, sd):
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return module


def grad_summary(g):
    """Compact, order-sensitive summary of a gradient tensor: [sum, l2, strided subsample(256)]."""
    g = g.detach().reshape(-1).double().numpy()
    idx = np.linspace(0, g.size - 1, min(256, g.size)).astype(np.int64)
    return np.concatenate([[g.sum(), np.sqrt((g * g).sum())], g[idx]])


def gen_static(out):
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import positional_encoding as pe
    import network
    import sampling
    import render as rr
    import rays as rays_mod
    from utils import img2mse
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    rng = np.random.RandomState(1234)

    # ---- A1 positional encoding (positional_encoding.py:5-64)
    x = np.concatenate([rng.uniform(-6, 6, (200, 3)), rng.uniform(-1, 1, (100, 3)),
                        np.array([[0, 0, 0], [6, -6, 2.5], [1e-3, 1e-6, -4.0]])]).astype(np.float32)
    G["pe_x"] = x
    e10, d10 = pe.get_embedder(10, 0)
    e4, d4 = pe.get_embedder(4, 0)
    assert (d10, d4) == (63, 27)
    G["pe10"] = e10(T(x)).numpy()
    G["pe4"] = e4(T(x)).numpy()
    x2 = rng.uniform(0, 1, (64, 2)).astype(np.float32)
    emb2 = pe.Embedder(include_input=True, input_dims=2, max_freq_log2=5, num_freqs=6, log_sampling=True,
                       periodic_fns=[torch.sin, torch.cos])
    G["pe2d_x"] = x2
    G["pe2d_L6"] = emb2.embed(T(x2)).numpy()
    emb_lin = pe.Embedder(include_input=False, input_dims=3, max_freq_log2=3, num_freqs=5, log_sampling=False,
                          periodic_fns=[torch.sin, torch.cos])
    G["pe_lin_noinput"] = emb_lin.embed(T(x[:64])).numpy()

    # ---- A2 NeRF forward (network.py:46-69)
    pts = rng.uniform(-3, 3, (384, 3)).astype(np.float32)
    dirs = rng.normal(size=(384, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    G["mlp_pts"], G["mlp_dirs"] = pts, dirs
    xin = torch.cat([e10(T(pts)), e4(T(dirs))], -1)
    G["mlp_x"] = xin.numpy()
    with torch.no_grad():
        for name, seed, peaky in (("default", 11, False), ("peaky", 12, True)):
            net = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(seed, peaky=peaky))
            G[f"mlp_out_{name}"] = net(xin).numpy()
        # small no-viewdirs variant (model.py:26-34 with use_viewdirs=False: input_ch_views=0, output_ch=5)
        sd = fx.make_nerf_state(13, D=4, W=128, input_ch=63, input_ch_views=0, output_ch=5, skips=(2,),
                                use_viewdirs=False)
        net = _load(torch, network.NeRF(4, 128, 63, 0, 5, [2], False), sd)
        G["mlp_out_noview"] = net(xin[:, :63]).numpy()

    # ---- A6 raw2outputs (render.py:82-131) + autograd grads of a scalar of (rgb_map, ...)
    for name, S, wb in (("s64_white", 64, True), ("s192", 192, False), ("s7", 7, True)):
        n = 48
        raw = (rng.normal(size=(n, S, 4)) * np.array([1, 1, 1, 4.0])).astype(np.float32)
        raw[0, :, 3] = -1.0       # fully transparent ray
        raw[1, S // 2, 3] = 1e4   # an opaque wall: alpha == 1 exactly -> exercises the +1e-10
        z = np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), -1)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        rawt = T(raw).requires_grad_(True)
        rgb, disp, acc, w, depth = rr.raw2outputs(rawt, T(z), T(d), 0, wb)
        g_rgb = rng.normal(size=(n, 3)).astype(np.float32)
        (rgb * T(g_rgb)).sum().backward()
        G[f"r2o_{name}_raw"], G[f"r2o_{name}_z"], G[f"r2o_{name}_d"] = raw, z, d
        G[f"r2o_{name}_rgb"], G[f"r2o_{name}_disp"], G[f"r2o_{name}_acc"] = \
            rgb.detach().numpy(), disp.detach().numpy(), acc.detach().numpy()
        G[f"r2o_{name}_w"], G[f"r2o_{name}_depth"] = w.detach().numpy(), depth.detach().numpy()
        G[f"r2o_{name}_grgb"], G[f"r2o_{name}_graw"] = g_rgb, rawt.grad.numpy()

    # ---- coarse z_vals / pts (render.py:186-210), perturb=0, lindisp both ways
    for name, near, far, lindisp in (("blender", 2.0, 6.0, False), ("ndc", 0.0, 1.0, False),
                                     ("lindisp", 0.5, 9.0, True)):
        n = 40
        nr = torch.full((n, 1), near) + 0 * T(rng.uniform(0, 1, (n, 1)).astype(np.float32))
        fr = torch.full((n, 1), far)
        if name == "lindisp":
            nr = nr + T(rng.uniform(0, 1, (n, 1)).astype(np.float32))
        t_vals = torch.linspace(0., 1., steps=64)
        zv = nr * (1. - t_vals) + fr * t_vals if not lindisp else 1. / (1. / nr * (1. - t_vals) + 1. / fr * t_vals)
        o = T(rng.uniform(-4, 4, (n, 3)).astype(np.float32))
        d = T(rng.normal(size=(n, 3)).astype(np.float32))
        p = o[..., None, :] + d[..., None, :] * zv[..., :, None]
        G[f"zc_{name}_near"], G[f"zc_{name}_far"] = nr.numpy(), fr.numpy()
        G[f"zc_{name}_o"], G[f"zc_{name}_d"] = o.numpy(), d.numpy()
        G[f"zc_{name}_z"], G[f"zc_{name}_pts"] = zv.numpy(), p.numpy()
    for n in (2, 3, 5, 64, 65, 128, 192):
        G[f"linspace_{n}"] = torch.linspace(0., 1., steps=n).numpy()

    # ---- A7 sample_pdf + merge (sampling.py:5-60, render.py:219-224, :238), det=True
    n = 200
    for name, near, far, peak in (("blender_p1", 2, 6, 1.0), ("blender_p6", 2, 6, 6.0), ("ndc_p30", 0, 1, 30.0)):
        t_vals = torch.linspace(0., 1., steps=64)
        zv = (near * (1. - t_vals) + far * t_vals).expand(n, 64)
        w = np.abs(rng.normal(size=(n, 64))) ** peak
        w = (w / w.sum(-1, keepdims=True)).astype(np.float32)
        w[0] = 0.0                      # all-zero weights ray
        w[1] = 0.0; w[1, 17] = 1.0      # one-hot
        w[2] = 0.0; w[2, 1] = 0.5; w[2, 62] = 0.5
        mids = .5 * (zv[..., 1:] + zv[..., :-1])
        for nimp in (128, 64):
            zs = sampling.sample_pdf(mids, T(w)[..., 1:-1], nimp, det=True)
            zm, _ = torch.sort(torch.cat([zv, zs], -1), -1)
            G[f"spdf_{name}_n{nimp}_zs"] = zs.numpy()
            G[f"spdf_{name}_n{nimp}_zm"] = zm.numpy()
            G[f"spdf_{name}_n{nimp}_std"] = torch.std(zs, dim=-1, unbiased=False).numpy()
        G[f"spdf_{name}_z"], G[f"spdf_{name}_w"] = zv.numpy().copy(), w
    # explicit-u variant (the pytest=True route is numpy f64 linspace -> torch.Tensor f32)
    u = rng.uniform(0, 1, (n, 128)).astype(np.float32)
    G["spdf_u"] = u

    # ---- A8 render_rays (render.py:134-244) full dicts
    def query_fn(embed, embeddirs, chunk=65536):
        return lambda inputs, viewdirs, fn: rr.run_network(inputs, viewdirs, fn, embed, embeddirs, chunk)

    coarse = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(21, peaky=True))
    fine = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(22, peaky=True))
    cases = {
        "blender": dict(rays=fx.blender_rays(96, 5), white_bkgd=True, lindisp=False, N_importance=128),
        "ndc": dict(rays=fx.ndc_like_rays(64, 6), white_bkgd=False, lindisp=False, N_importance=64),
        "lindisp": dict(rays=fx.blender_rays(48, 7, near=1.0, far=8.0), white_bkgd=False, lindisp=True,
                        N_importance=128),
    }
    with torch.no_grad():
        for name, c in cases.items():
            ret = rr.render_rays(T(c["rays"]), coarse, query_fn(e10, e4), 64, retraw=True, lindisp=c["lindisp"],
                                 perturb=0., N_importance=c["N_importance"], network_fine=fine,
                                 white_bkgd=c["white_bkgd"], raw_noise_std=0.)
            G[f"rr_{name}_rays"] = c["rays"]
            for k, v in ret.items():
                G[f"rr_{name}_{k}"] = v.numpy()
            # the intermediates render_rays does not return (coarse weights, merged z_vals), recomputed
            # with the reference's own functions in the order of render.py:186-229; checked to
            # reproduce the returned fine `raw` bit-for-bit.
            rb = T(c["rays"])
            o_, d_, vd_ = rb[:, 0:3], rb[:, 3:6], rb[:, -3:]
            nr, fr = rb[:, 6:7], rb[:, 7:8]
            tv = torch.linspace(0., 1., steps=64)
            zv = nr * (1. - tv) + fr * tv if not c["lindisp"] else 1. / (1. / nr * (1. - tv) + 1. / fr * tv)
            p_ = o_[..., None, :] + d_[..., None, :] * zv[..., :, None]
            raw0 = query_fn(e10, e4)(p_, vd_, coarse)
            _, _, _, w0, _ = rr.raw2outputs(raw0, zv, d_, 0, c["white_bkgd"])
            zs = sampling.sample_pdf(.5 * (zv[..., 1:] + zv[..., :-1]), w0[..., 1:-1], c["N_importance"], det=True)
            zf, _ = torch.sort(torch.cat([zv, zs], -1), -1)
            p_ = o_[..., None, :] + d_[..., None, :] * zf[..., :, None]
            raw1 = query_fn(e10, e4)(p_, vd_, fine)
            assert torch.equal(raw1, ret["raw"])
            G[f"rr_{name}_raw0"], G[f"rr_{name}_weights0"] = raw0.numpy(), w0.numpy()
            G[f"rr_{name}_z_vals"] = zf.numpy()
        # default-init ("smooth") nets: a well-conditioned field, where END-TO-END parity holds to 1e-4
        c_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(24))
        f_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(25))
        rs = fx.blender_rays(128, 10)
        ret = rr.render_rays(T(rs), c_s, query_fn(e10, e4), 64, retraw=True, perturb=0., N_importance=128,
                             network_fine=f_s, white_bkgd=True, raw_noise_std=0.)
        G["rr_smooth_rays"] = rs
        for k, v in ret.items():
            G[f"rr_smooth_{k}"] = v.numpy()
        # no viewdirs, no fine net (N_importance=0), 8-column rays
        sd = fx.make_nerf_state(23, D=4, W=128, input_ch=63, input_ch_views=0, output_ch=4, skips=(2,),
                                use_viewdirs=False, peaky=True)
        small = _load(torch, network.NeRF(4, 128, 63, 0, 4, [2], False), sd)
        r8 = fx.blender_rays(32, 8, with_viewdirs=False)
        ret = rr.render_rays(T(r8), small, query_fn(e10, None), 64, retraw=True, N_importance=0, white_bkgd=True)
        G["rr_noview_rays"] = r8
        for k, v in ret.items():
            G[f"rr_noview_{k}"] = v.numpy()
        # fine pass through the coarse net (network_fine=None, render.py:226)
        ret = rr.render_rays(T(r8), small, query_fn(e10, None), 64, N_importance=32, network_fine=None,
                             white_bkgd=False)
        for k, v in ret.items():
            G[f"rr_samefine_{k}"] = v.numpy()

    # ---- A9 render() from a pose (render.py:247-315, rays.py:5-27, :52-83)
    with torch.no_grad():
        H = W = 12
        K, focal = fx.intrinsics(H, W)
        c2w = fx.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
        kw = dict(network_query_fn=query_fn(e10, e4), perturb=0., N_importance=128, network_fine=fine,
                  N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.,
                  ndc=False, lindisp=False)
        rgb, disp, acc, extras = rr.render(H, W, K, chunk=100, c2w=T(c2w), near=2., far=6., **kw)
        G["render_blender_c2w"], G["render_blender_K"] = c2w, K
        G["render_blender_rgb"], G["render_blender_disp"], G["render_blender_acc"] = rgb.numpy(), disp.numpy(), acc.numpy()
        G["render_blender_rgb0"], G["render_blender_z_std"] = extras["rgb0"].numpy(), extras["z_std"].numpy()
        ro, rd = rays_mod.get_rays(H, W, K, T(c2w))
        G["render_blender_rays_o"], G["render_blender_rays_d"] = ro.numpy().copy(), rd.numpy()
        # NDC forward-facing
        H, W = 9, 12
        K, focal = fx.intrinsics(H, W, 0.9)
        c2w = np.array([[1, 0, 0, 0.1], [0, 1, 0, -0.05], [0, 0, 1, 0.2]], np.float32)
        kw.update(white_bkgd=False, ndc=True, N_importance=64)
        rgb, disp, acc, extras = rr.render(H, W, K, chunk=50, c2w=T(c2w), near=0., far=1., **kw)
        G["render_ndc_c2w"], G["render_ndc_K"] = c2w, K
        G["render_ndc_rgb"], G["render_ndc_disp"], G["render_ndc_acc"] = rgb.numpy(), disp.numpy(), acc.numpy()
        ro, rd = rays_mod.get_rays(H, W, K, T(c2w))
        ro, rd = rays_mod.ndc_rays(H, W, K[0][0], 1., ro, rd)
        G["render_ndc_rays_o"], G["render_ndc_rays_d"] = ro.numpy(), rd.numpy()
        # c2w_staticcam branch (render.py:286-288)
        c2w_s = fx.pose_spherical(-60.0, -20.0, 4.0)[:3, :4]
        H = W = 6
        K, focal = fx.intrinsics(H, W)
        kw.update(white_bkgd=True, ndc=False, N_importance=128)
        rgb, _, _, _ = rr.render(H, W, K, chunk=1 << 15, c2w=T(G["render_blender_c2w"]), near=2., far=6.,
                                 c2w_staticcam=T(c2w_s), **kw)
        G["render_static_c2w"], G["render_static_rgb"] = c2w_s, rgb.numpy()

    # ---- training step (train.py:226-243): loss + per-parameter grad summaries, 64-ray batch
    ro, rd, target = fx.train_batch(64, seed=3)
    batch_rays = torch.stack([T(ro), T(rd)], 0)
    kw = dict(network_query_fn=query_fn(e10, e4), perturb=0., N_importance=128, network_fine=fine, N_samples=64,
              network_fn=coarse, use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)
    H = W = 400
    K, _ = fx.intrinsics(H, W)
    for net in (coarse, fine):
        net.zero_grad()
    rgb, disp, acc, extras = rr.render(H, W, K, chunk=32768, rays=batch_rays, retraw=True, near=2., far=6., **kw)
    loss = img2mse(rgb, T(target)) + img2mse(extras["rgb0"], T(target))
    loss.backward()
    G["train_ro"], G["train_rd"], G["train_target"] = ro, rd, target
    with torch.no_grad():   # intermediates (as for the rr_* cases): coarse weights and merged fine depths
        o_, d_ = T(ro), T(rd)
        vd_ = d_ / torch.norm(d_, dim=-1, keepdim=True)
        tv = torch.linspace(0., 1., steps=64)
        zv = (2. * torch.ones_like(d_[:, :1])) * (1. - tv) + (6. * torch.ones_like(d_[:, :1])) * tv
        raw0 = query_fn(e10, e4)(o_[..., None, :] + d_[..., None, :] * zv[..., :, None], vd_, coarse)
        _, _, _, w0, _ = rr.raw2outputs(raw0, zv, d_, 0, True)
        zs = sampling.sample_pdf(.5 * (zv[..., 1:] + zv[..., :-1]), w0[..., 1:-1], 128, det=True)
        zf, _ = torch.sort(torch.cat([zv, zs], -1), -1)
        raw1 = query_fn(e10, e4)(o_[..., None, :] + d_[..., None, :] * zf[..., :, None], vd_, fine)
        assert torch.equal(raw1, extras["raw"].detach())
        G["train_weights0"], G["train_z_vals"] = w0.numpy(), zf.numpy()
    G["train_loss"] = np.array([loss.item()])
    G["train_rgb"], G["train_rgb0"] = rgb.detach().numpy(), extras["rgb0"].detach().numpy()
    for tag, net in (("coarse", coarse), ("fine", fine)):
        for k, p in net.named_parameters():
            G[f"train_grad_{tag}_{k}"] = grad_summary(p.grad)

    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


def gen_static_pytest(out):
    """The reference's fixed-random route (pytest=True, render.py:113-117, :203-206; sampling.py:37-45): stratified
    jitter, uniform density noise and random resampling, all from numpy seeded with 0 -- a deterministic end-to-end
    vector for perturb=1 / raw_noise_std>0."""
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import positional_encoding as pe
    import network
    import render as rr
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    e10, _ = pe.get_embedder(10, 0)
    e4, _ = pe.get_embedder(4, 0)
    q = lambda inputs, viewdirs, fn: rr.run_network(inputs, viewdirs, fn, e10, e4, 65536)  # noqa: E731
    c_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(24))
    f_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(25))
    rays = fx.blender_rays(64, 11)
    G["rays"] = rays
    with torch.no_grad():
        for name, kw in (("jitter_noise", dict(perturb=1., raw_noise_std=1.)), ("jitter", dict(perturb=1., raw_noise_std=0.)),
                         ("noise", dict(perturb=0., raw_noise_std=0.5))):
            ret = rr.render_rays(T(rays), c_s, q, 64, retraw=True, N_importance=128, network_fine=f_s, white_bkgd=True,
                                 pytest=True, **kw)
            for k, v in ret.items():
                G[f"{name}_{k}"] = v.numpy()
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


def gen_dnerf(out):
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, os.path.join(REF, "vanilla_dynamic"))
    import torch
    import positional_encoding as pe
    import network
    import render as rr
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    rng = np.random.RandomState(4321)

    embed_fn, ich = pe.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, ichv = pe.get_embedder(4, 0, input_dims=3)
    embed_time, icht = pe.get_time_embedder(10)
    assert (ich, ichv, icht) == (63, 27, 21)
    tt = rng.uniform(0, 1, (50, 1)).astype(np.float32)
    G["time_x"], G["time_pe"] = tt, embed_time(T(tt)).numpy()
    e_noinp, d_noinp = pe.get_embedder(4, 0, input_dims=3, include_input=False)
    x = rng.uniform(-2, 2, (40, 3)).astype(np.float32)
    G["pe_noinp_x"], G["pe_noinp"] = x, e_noinp(T(x)).numpy()

    model = network.DNeRF_Deformation(D=8, W=256, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                                      input_ch_time=21, output_ch=4, skips=[4], use_viewdirs=True,
                                      zero_canonical=True)
    sd = fx.make_dnerf_state(31, peaky=True, dx_scale=3.0)
    model.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    model.embed_fn = embed_fn

    # D3 forward on points (network.py:307-341)
    pts = rng.uniform(-2, 2, (256, 3)).astype(np.float32)
    dirs = rng.normal(size=(256, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    G["fwd_pts"], G["fwd_dirs"] = pts, dirs
    with torch.no_grad():
        xin = torch.cat([embed_fn(T(pts)), embeddirs_fn(T(dirs))], -1)
        for name, tval in (("t0", 0.0), ("t05", 0.5)):
            t_raw = torch.full((256, 1), tval)
            te = embed_time(t_raw)
            out_, dx = model(xin, [te, te], pts_raw=T(pts), t_raw=t_raw)
            G[f"fwd_{name}_out"], G[f"fwd_{name}_dx"] = out_.numpy(), dx.numpy()
        # canonical net alone (D1 forward returns (out, zeros))
        o, z = model._occ(xin)
        G["occ_out"], G["occ_zeros"] = o.numpy(), z.numpy()

    # D5 render_rays_dnerf (render.py:218-326)
    qfn = lambda inputs, viewdirs, times, fn: rr.run_network_dnerf(  # noqa: E731
        inputs, viewdirs, times, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embed_time_fn=embed_time,
        netchunk=65536, network_type="deformation")
    rays = fx.blender_rays(48, 9)
    G["rr_rays"] = rays
    with torch.no_grad():
        for name, tval in (("t0", 0.0), ("t05", 0.5)):
            tv = torch.full((48, 1), tval)
            ret = rr.render_rays_dnerf(T(rays), tv, model, qfn, 64, retraw=True, perturb=0., N_importance=128,
                                       network_fine=None, white_bkgd=True)
            for k, v in ret.items():
                G[f"rr_{name}_{k}"] = v.numpy()
        # per-ray times (train.py:132-161 route): first ray has t != 0 so the time net runs for all
        tv = T(rng.uniform(0.1, 1, (48, 1)).astype(np.float32))
        ret = rr.render_rays_dnerf(T(rays), tv, model, qfn, 64, retraw=False, perturb=0., N_importance=128,
                                   network_fine=None, white_bkgd=True)
        G["rr_tvar_times"] = tv.numpy()
        for k, v in ret.items():
            G[f"rr_tvar_{k}"] = v.numpy()
        # render_dnerf from a pose with scalar time (render.py:368-433)
        H = W = 8
        K, focal = fx.intrinsics(H, W)
        c2w = fx.pose_spherical(45.0, -30.0, 4.0)[:3, :4]
        rgb, disp, acc, extras = rr.render_dnerf(
            H, W, K, 0.25, chunk=40, c2w=T(c2w), near=2., far=6., use_viewdirs=True, network_fn=model,
            network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128, network_fine=None, white_bkgd=True,
            raw_noise_std=0.)
        G["render_c2w"], G["render_K"] = c2w, K
        G["render_rgb"], G["render_disp"], G["render_acc"], G["render_dx"] = \
            rgb.numpy(), disp.numpy(), acc.numpy(), extras["dx"].numpy()

    # training grads through the deformation chain (train.py:238-276 minus clip/optimizer)
    ro, rd, target = fx.train_batch(32, seed=4)
    vd = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    rays = np.concatenate([ro, rd, np.full((32, 1), 2, np.float32), np.full((32, 1), 6, np.float32), vd], -1).astype(np.float32)
    tv = T(rng.uniform(0.1, 1, (32, 1)).astype(np.float32))
    model.zero_grad()
    ret = rr.render_rays_dnerf(T(rays), tv, model, qfn, 64, retraw=True, perturb=0., N_importance=128,
                               network_fine=None, white_bkgd=True)
    loss = torch.mean((ret["rgb_map"] - T(target)) ** 2) + torch.mean((ret["rgb0"] - T(target)) ** 2)
    loss.backward()
    G["train_rays"], G["train_times"], G["train_target"] = rays, tv.numpy(), target
    G["train_loss"] = np.array([loss.item()])
    for k, p in model.named_parameters():
        G[f"train_grad_{k}"] = grad_summary(p.grad)

    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


def gen_tooling(out):
    """(1) the rays_rgb table of train.py:147-153 built with the reference's get_rays_np; (2) a checkpoint in
    the on-disk format of train.py:260-265 written from the reference's NeRF modules + torch Adam, with the networks'
    outputs before and after one more optimizer step."""
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import network
    import rays as rays_mod
    G = {"meta": _meta(torch), "numpy": np.array([np.__version__])}
    rng = np.random.RandomState(77)

    # ---- (1) train.py:147-153, restated call for call on a small scene
    H, W, focal = 5, 7, 6.5
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])           # train.py builds K like this (float64)
    poses = np.stack([fx.pose_spherical(t, -30.0, 4.0) for t in (-120.0, -10.0, 45.0, 160.0)]).astype(np.float32)
    images = rng.rand(4, H, W, 3).astype(np.float32)
    i_train = np.array([0, 2, 3])
    rays = np.stack([rays_mod.get_rays_np(H, W, K, p) for p in poses[:, :3, :4]], 0)
    rays_rgb = np.concatenate([rays, images[:, None]], 1)
    rays_rgb = np.transpose(rays_rgb, [0, 2, 3, 1, 4])
    rays_rgb = np.stack([rays_rgb[i] for i in i_train], 0)
    rays_rgb = np.reshape(rays_rgb, [-1, 3, 3])
    G["table_dtype_before_cast"] = np.array([str(rays_rgb.dtype)])
    rays_rgb = rays_rgb.astype(np.float32)
    G.update(table=rays_rgb, table_K=K, table_poses=poses, table_images=images, table_i_train=i_train,
             table_hw=np.array([H, W]))

    # ---- (2) checkpoint
    torch.manual_seed(5)
    mk = lambda: network.NeRF(D=3, W=128, input_ch=63, output_ch=5, skips=[1], input_ch_views=27, use_viewdirs=True)  # noqa: E731
    coarse, fine = mk(), mk()
    grad_vars = list(coarse.parameters()) + list(fine.parameters())
    opt = torch.optim.Adam(params=grad_vars, lr=5e-4, betas=(0.9, 0.999))          # model.py:52
    x = torch.from_numpy(rng.uniform(-1, 1, (96, 90)).astype(np.float32))
    tgt = torch.from_numpy(rng.uniform(0, 1, (96, 4)).astype(np.float32))

    def step():
        opt.zero_grad()
        loss = ((coarse(x) - tgt) ** 2).mean() + ((fine(x) - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        return float(loss.detach())

    step(); step()
    path = os.path.join(HERE, "ref_ckpt_small.tar")
    torch.save({'global_step': 2, 'network_fn_state_dict': coarse.state_dict(),
                'network_fine_state_dict': fine.state_dict(), 'optimizer_state_dict': opt.state_dict()}, path)
    with torch.no_grad():
        G["ckpt_out_coarse"], G["ckpt_out_fine"] = coarse(x).numpy(), fine(x).numpy()
    G["ckpt_loss3"] = np.array([step()])
    with torch.no_grad():
        G["ckpt_out3_coarse"], G["ckpt_out3_fine"] = coarse(x).numpy(), fine(x).numpy()
    G.update(ckpt_x=x.numpy(), ckpt_tgt=tgt.numpy())
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), "and", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_static2(out):
    """Round-2 additions, static stack: (1) BASELINE configs[0] at its own size -- render() of the 100x100 view,
    default-init nets, theta = 0 (render.py:247-315); (2) model.py:26-41's no-viewdirs wiring
    (output_ch = 5 when N_importance > 0, render.py:100-103 slices raw[..., :3] / raw[..., 3]) through render_rays with
    a fine pass, and the parameter gradients of train.py:232-242's loss."""
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import positional_encoding as pe
    import network
    import render as rr
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    e10, _ = pe.get_embedder(10, 0)
    e4, _ = pe.get_embedder(4, 0)

    def query_fn(embed, embeddirs, chunk=65536):
        return lambda inputs, viewdirs, fn: rr.run_network(inputs, viewdirs, fn, embed, embeddirs, chunk)

    c_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(24))
    f_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(25))
    H = W = 100
    K, focal = fx.intrinsics(H, W)
    c2w = fx.pose_spherical(0.0, -30.0, 4.0)[:3, :4]
    kw = dict(network_query_fn=query_fn(e10, e4), perturb=0., N_importance=128, network_fine=f_s, N_samples=64,
              network_fn=c_s, use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)
    with torch.no_grad():
        rgb, disp, acc, extras = rr.render(H, W, K, chunk=32768, c2w=T(c2w), near=2., far=6., **kw)
    G["c1_c2w"], G["c1_K"] = c2w, K
    G["c1_rgb"], G["c1_disp"], G["c1_acc"] = rgb.numpy(), disp.numpy(), acc.numpy()
    G["c1_rgb0"], G["c1_acc0"], G["c1_z_std"] = extras["rgb0"].numpy(), extras["acc0"].numpy(), extras["z_std"].numpy()
    # the SAME reference code run in float64 (torch.set_default_dtype): what the algorithm computes without f32 rounding.
    # |f32 run - f64 run| is the reference's own rounding error on this view -- the fine pass amplifies 1e-7 differences of
    # the coarse weights through sample_pdf on near-empty rays -- and bounds what "agrees with the reference" can mean.
    torch.set_default_dtype(torch.float64)
    try:
        c64 = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True).double(),
                    {k: v.astype(np.float64) for k, v in fx.make_nerf_state(24).items()})
        f64 = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True).double(),
                    {k: v.astype(np.float64) for k, v in fx.make_nerf_state(25).items()})
        kw64 = dict(kw, network_fn=c64, network_fine=f64)
        with torch.no_grad():
            rgb64, disp64, acc64, _ = rr.render(H, W, K, chunk=32768, c2w=T(c2w).double(), near=2., far=6., **kw64)
    finally:
        torch.set_default_dtype(torch.float32)
    G["c1_rgb_f64"], G["c1_disp_f64"], G["c1_acc_f64"] = rgb64.numpy(), disp64.numpy(), acc64.numpy()
    for k in ("rgb", "disp", "acc"):
        print(f"c1 {k}: |reference f32 - reference f64| max = {np.abs(G['c1_' + k] - G['c1_' + k + '_f64']).max():.3e}")

    # no view directions, output_ch = 5 (model.py:30: output_ch = 5 if N_importance > 0), fine pass through a second net
    def mk(seed):
        sd = fx.make_nerf_state(seed, D=4, W=128, input_ch=63, input_ch_views=0, output_ch=5, skips=(2,),
                                use_viewdirs=False)
        return _load(torch, network.NeRF(4, 128, 63, 0, 5, [2], False), sd)
    c5, f5 = mk(41), mk(42)
    r8 = fx.blender_rays(40, 13, with_viewdirs=False)
    tgt = np.random.RandomState(14).rand(40, 3).astype(np.float32)
    ret = rr.render_rays(T(r8), c5, query_fn(e10, None), 64, retraw=True, N_importance=32, network_fine=f5,
                         white_bkgd=True, perturb=0.)
    loss = torch.mean((ret["rgb_map"] - T(tgt)) ** 2) + torch.mean((ret["rgb0"] - T(tgt)) ** 2)
    loss.backward()
    G["nv5_rays"], G["nv5_target"], G["nv5_loss"] = r8, tgt, np.array([loss.item()])
    for k, v in ret.items():
        G[f"nv5_{k}"] = v.detach().numpy()
    for tag, net in (("coarse", c5), ("fine", f5)):
        for k, p in net.named_parameters():
            if p.grad is not None:
                G[f"nv5_grad_{tag}_{k}"] = grad_summary(p.grad)
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


def gen_dnerf2(out):
    """Round-2 additions, vanilla_dynamic stack, all on WELL-CONDITIONED models (default-init family, dx_scale = 1), where
    the reference's outputs are reproducible to the north star's 1e-4 end to end: render_rays_dnerf ret-dicts at t = 0.5
    and per-ray times, a 100 x 100 render_dnerf image at t = 0.5 (BASELINE configs[4] at the size of configs[0]), the
    training gradients, the LayerNorm variants (network.py:46-48, :75-76, :88-89, :276-300) forward and backward, and
    DNeRF_Straightforward.forward (network.py:151-196; callable directly although run_network_dnerf cannot reach it)."""
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, os.path.join(REF, "vanilla_dynamic"))
    import torch
    import positional_encoding as pe
    import network
    import render as rr
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    rng = np.random.RandomState(8765)
    embed_fn, _ = pe.get_embedder(10, 0, input_dims=3)
    embeddirs_fn, _ = pe.get_embedder(4, 0, input_dims=3)
    embed_time, _ = pe.get_time_embedder(10)

    def mk_model(sd, **kw):
        m = network.DNeRF_Deformation(input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=4,
                                      use_viewdirs=True, zero_canonical=True, **kw)
        m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
        m.embed_fn = embed_fn
        return m

    model = mk_model(fx.make_dnerf_state(33, peaky=False, dx_scale=1.0), D=8, W=256, D_deform=8, W_deform=256, skips=[4])
    qfn = lambda inputs, viewdirs, times, fn: rr.run_network_dnerf(  # noqa: E731
        inputs, viewdirs, times, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embed_time_fn=embed_time,
        netchunk=65536, network_type="deformation")
    rays = fx.blender_rays(48, 15)
    G["rr_rays"] = rays
    with torch.no_grad():
        ret = rr.render_rays_dnerf(T(rays), torch.full((48, 1), 0.5), model, qfn, 64, retraw=True, perturb=0.,
                                   N_importance=128, network_fine=None, white_bkgd=True)
        for k, v in ret.items():
            G[f"rr_t05_{k}"] = v.numpy()
        tv = T(rng.uniform(0.1, 1, (48, 1)).astype(np.float32))
        ret = rr.render_rays_dnerf(T(rays), tv, model, qfn, 64, retraw=False, perturb=0., N_importance=128,
                                   network_fine=None, white_bkgd=True)
        G["rr_tvar_times"] = tv.numpy()
        for k, v in ret.items():
            G[f"rr_tvar_{k}"] = v.numpy()
        H = W = 100
        K, focal = fx.intrinsics(H, W)
        c2w = fx.pose_spherical(0.0, -30.0, 4.0)[:3, :4]
        rgb, disp, acc, extras = rr.render_dnerf(
            H, W, K, 0.5, chunk=32768, c2w=T(c2w), near=2., far=6., use_viewdirs=True, network_fn=model,
            network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128, network_fine=None, white_bkgd=True,
            raw_noise_std=0.)
        G["img_c2w"], G["img_K"] = c2w, K
        G["img_rgb"], G["img_disp"], G["img_acc"] = rgb.numpy(), disp.numpy(), acc.numpy()
        G["img_rgb0"], G["img_acc0"] = extras["rgb0"].numpy(), extras["acc0"].numpy()
        G["img_dx_absmean"] = np.abs(extras["dx"].numpy()).mean(axis=(2, 3))      # [H, W]: per-pixel mean |dx|, fine pass
    # the last sample's density, fine and coarse pass (its alpha is a step function of the sign: render.py:98), for the
    # bf16 test's stability mask
    with torch.no_grad():
        ret = rr.render_dnerf(H, W, K, 0.5, chunk=32768, c2w=T(c2w), near=2., far=6., use_viewdirs=True, network_fn=model,
                              network_query_fn=qfn, N_samples=64, perturb=0., N_importance=0, network_fine=None,
                              white_bkgd=True, raw_noise_std=0., retraw=True)[3]
        G["img_raw0_last_sigma"] = ret["raw"][..., -1, 3].numpy()
        ret = rr.render_dnerf(H, W, K, 0.5, chunk=32768, c2w=T(c2w), near=2., far=6., use_viewdirs=True, network_fn=model,
                              network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128, network_fine=None,
                              white_bkgd=True, raw_noise_std=0., retraw=True)[3]
        G["img_raw_last_sigma"] = ret["raw"][..., -1, 3].numpy()
    torch.set_default_dtype(torch.float64)       # the same reference code in float64: see gen_static2
    try:
        m64 = network.DNeRF_Deformation(D=8, W=256, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                                        input_ch_time=21, output_ch=4, skips=[4], use_viewdirs=True, zero_canonical=True)
        m64.load_state_dict({k: T(v.astype(np.float64)) for k, v in fx.make_dnerf_state(33, peaky=False, dx_scale=1.0).items()},
                            strict=True)
        m64 = m64.double()
        m64.embed_fn = embed_fn
        with torch.no_grad():
            rgb64, disp64, acc64, _ = rr.render_dnerf(
                H, W, K, 0.5, chunk=32768, c2w=T(c2w).double(), near=2., far=6., use_viewdirs=True, network_fn=m64,
                network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128, network_fine=None, white_bkgd=True,
                raw_noise_std=0.)
    finally:
        torch.set_default_dtype(torch.float32)
    G["img_rgb_f64"], G["img_disp_f64"], G["img_acc_f64"] = rgb64.numpy(), disp64.numpy(), acc64.numpy()
    for k in ("rgb", "disp", "acc"):
        print(f"dnerf img {k}: |reference f32 - reference f64| max = {np.abs(G['img_' + k] - G['img_' + k + '_f64']).max():.3e}")

    # training gradients on the smooth model (train.py:238-276 minus clip/optimizer), incl. the dx regulariser (:252-260)
    ro, rd, target = fx.train_batch(32, seed=6)
    vd = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    trays = np.concatenate([ro, rd, np.full((32, 1), 2, np.float32), np.full((32, 1), 6, np.float32), vd], -1).astype(np.float32)
    tv = T(rng.uniform(0.1, 1, (32, 1)).astype(np.float32))
    model.zero_grad()
    ret = rr.render_rays_dnerf(T(trays), tv, model, qfn, 64, retraw=True, perturb=0., N_importance=128,
                               network_fine=None, white_bkgd=True)
    loss = torch.mean((ret["rgb_map"] - T(target)) ** 2) + torch.mean((ret["rgb0"] - T(target)) ** 2) \
        + 0.1 * torch.mean(ret["dx"] ** 2)
    loss.backward()
    G["train_rays"], G["train_times"], G["train_target"] = trays, tv.numpy(), target
    G["train_loss"] = np.array([loss.item()])
    for k, p in model.named_parameters():
        G[f"train_grad_{k}"] = grad_summary(p.grad)

    # ---- LayerNorm variants
    pts = rng.uniform(-1.5, 1.5, (320, 3)).astype(np.float32)
    dirs = rng.normal(size=(320, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    tt = rng.uniform(0.05, 1, (320, 1)).astype(np.float32)
    G["ln_pts"], G["ln_dirs"], G["ln_times"] = pts, dirs, tt
    xin = torch.cat([embed_fn(T(pts)), embeddirs_fn(T(dirs))], -1)
    for tag, D, W, skips in (("small", 3, 64, [1]), ("full", 8, 256, [4])):
        sd = fx.make_dnerf_layernorm_state(51 if tag == "small" else 52, D=D, W=W, D_deform=D, W_deform=W, skips=tuple(skips))
        m = mk_model(sd, D=D, W=W, D_deform=D, W_deform=W, skips=skips, use_layernorm=True)
        with torch.no_grad():
            o, z = m._occ(xin)
            G[f"ln_{tag}_occ_out"] = o.numpy()
        te = embed_time(T(tt))
        m.zero_grad()
        out_, dx = m(xin, [te, te], pts_raw=T(pts), t_raw=T(tt))
        G[f"ln_{tag}_out"], G[f"ln_{tag}_dx"] = out_.detach().numpy(), dx.detach().numpy()
        gw = rng.normal(size=(320, 4)).astype(np.float32)
        gd = rng.normal(size=(320, 3)).astype(np.float32)
        ((out_ * T(gw)).sum() + (dx * T(gd)).sum()).backward()
        G[f"ln_{tag}_gw"], G[f"ln_{tag}_gd"] = gw, gd
        for k, p in m.named_parameters():
            G[f"ln_{tag}_grad_{k}"] = grad_summary(p.grad)

    # ---- DNeRF_Straightforward.forward(x, ts)
    sf = network.DNeRF_Straightforward(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=4,
                                       skips=[4], use_viewdirs=True)
    sf.load_state_dict({k: T(v) for k, v in fx.make_straightforward_state(61).items()}, strict=True)
    te = embed_time(T(tt))
    out_, dx = sf(xin, [te, te])
    G["sf_out"], G["sf_dx"] = out_.detach().numpy(), dx.detach().numpy()
    gw = rng.normal(size=(320, 4)).astype(np.float32)
    (out_ * T(gw)).sum().backward()
    G["sf_gw"] = gw
    for k, p in sf.named_parameters():
        G[f"sf_grad_{k}"] = grad_summary(p.grad)

    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


TRAJ_STEPS, TRAJ_RAYS = 30, 256


def traj_batches():
    """The fixed schedule of the training-trajectory golden: TRAJ_STEPS batches of TRAJ_RAYS rays (C3-style poses) with a
    smooth, learnable target colour (a function of the ray direction), shared by the generator and the GPU test."""
    ro, rd, _ = fx.train_batch(TRAJ_STEPS * TRAJ_RAYS, seed=21)
    d = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    target = (0.5 + 0.5 * np.sin(3.0 * d + np.array([0.0, 1.0, 2.0]))).astype(np.float32)
    return ro, rd, target


def gen_trajectory(out):
    """The reference's training loop (train.py:226-251: render -> img2mse(rgb) + img2mse(rgb0) -> backward -> Adam ->
    exponential learning-rate decay) run for TRAJ_STEPS steps on the imported reference, CPU f32: the loss of every step, the
    PSNR, and the parameters' total movement.  perturb = 0 and raw_noise_std = 0 (no random draws to reproduce)."""
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import positional_encoding as pe
    import network
    import render as rr
    from utils import img2mse, mse2psnr
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    e10, _ = pe.get_embedder(10, 0)
    e4, _ = pe.get_embedder(4, 0)
    coarse = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(24))
    fine = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(25))
    w0 = {f"{t}.{k}": p.detach().clone() for t, n in (("coarse", coarse), ("fine", fine)) for k, p in n.named_parameters()}
    grad_vars = list(coarse.parameters()) + list(fine.parameters())
    lrate, decay_rate, decay_steps = 5e-4, 0.1, 250 * 1000
    opt = torch.optim.Adam(params=grad_vars, lr=lrate, betas=(0.9, 0.999))          # model.py:52
    kw = dict(network_query_fn=lambda i, v, f: rr.run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
              network_fine=fine, N_samples=64, network_fn=coarse, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.,
              ndc=False, lindisp=False)
    K, _ = fx.intrinsics(400, 400)
    ro, rd, target = traj_batches()
    losses, psnrs = [], []
    for i in range(TRAJ_STEPS):
        sl = slice(i * TRAJ_RAYS, (i + 1) * TRAJ_RAYS)
        batch_rays = torch.stack([T(ro[sl]), T(rd[sl])], 0)
        rgb, disp, acc, extras = rr.render(400, 400, K, chunk=32768, rays=batch_rays, retraw=True, near=2., far=6., **kw)
        opt.zero_grad()
        img_loss = img2mse(rgb, T(target[sl]))
        loss = img_loss + img2mse(extras["rgb0"], T(target[sl]))                    # train.py:232-239
        loss.backward()
        opt.step()
        for g in opt.param_groups:                                                   # train.py:247-251
            g["lr"] = lrate * (decay_rate ** ((i + 1) / decay_steps))
        losses.append(loss.item())
        psnrs.append(mse2psnr(img_loss).item())
        print(i, losses[-1], psnrs[-1], flush=True)
    G["losses"], G["psnrs"] = np.array(losses), np.array(psnrs)
    named = [(f"{t}.{k}", p) for t, n in (("coarse", coarse), ("fine", fine)) for k, p in n.named_parameters()]
    G["names"] = np.array([k for k, _ in named])
    G["delta_norms"] = np.array([float((p.detach() - w0[k]).double().norm()) for k, p in named])
    # the trained nets' image of the first batch: where the whole trajectory ends up
    with torch.no_grad():
        rgb, _, acc, extras = rr.render(400, 400, K, chunk=32768, rays=torch.stack([T(ro[:TRAJ_RAYS]), T(rd[:TRAJ_RAYS])], 0),
                                        retraw=True, near=2., far=6., **kw)
    G["final_rgb"], G["final_acc"], G["final_rgb0"] = rgb.numpy(), acc.numpy(), extras["rgb0"].numpy()
    for k, p in named:
        G[f"final_{k}"] = grad_summary(p)        # [sum, l2, strided subsample] of the trained parameter
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


DTRAJ_STEPS, DTRAJ_RAYS, DTRAJ_REG = 20, 256, 1e-2


def dtraj_batches():
    """Schedule of the D-NeRF training-trajectory golden: DTRAJ_STEPS batches of DTRAJ_RAYS rays with per-ray times and a
    smooth target that moves with time."""
    ro, rd, _ = fx.train_batch(DTRAJ_STEPS * DTRAJ_RAYS, seed=23)
    d = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    times = np.random.RandomState(24).uniform(0.05, 0.95, (DTRAJ_STEPS * DTRAJ_RAYS, 1)).astype(np.float32)
    target = (0.5 + 0.5 * np.sin(3.0 * d + 2.0 * times + np.array([0.0, 1.0, 2.0]))).astype(np.float32)
    return ro, rd, times, target


def gen_trajectory_dnerf(out):
    """vanilla_dynamic/train.py:237-277 for DTRAJ_STEPS steps on the imported reference (CPU): render_dnerf on a ray
    batch with per-ray times -> mse(rgb) + deform_reg_weight * (mean dx^2 + mean dx0^2) + mse(rgb0) -> backward ->
    clip_grad_norm_(1.0) -> Adam -> learning-rate decay.  One DNeRF_Deformation for both passes; perturb = 0.
    Run twice: in float32 (the golden) and, the SAME code, in float64 -- this loop is sensitive (x + dx passes through a
    frequency-512 encoding, Adam normalises every step), so the distance between the two runs is the reference's own rounding
    error along the trajectory and bounds what agreement with its float32 run can mean."""
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, os.path.join(REF, "vanilla_dynamic"))
    import torch
    import torch.nn as nn
    import positional_encoding as pe
    import network
    import render as rr
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    K, _ = fx.intrinsics(400, 400)
    ro, rd, times, target = dtraj_batches()

    def run(dtype):
        torch.set_default_dtype(dtype)
        try:
            T = lambda a: torch.from_numpy(a).to(dtype)  # noqa: E731
            embed_fn, _ = pe.get_embedder(10, 0, input_dims=3)
            embeddirs_fn, _ = pe.get_embedder(4, 0, input_dims=3)
            embed_time, _ = pe.get_time_embedder(10)
            model = network.DNeRF_Deformation(D=8, W=256, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                                              input_ch_time=21, output_ch=4, skips=[4], use_viewdirs=True, zero_canonical=True)
            model.load_state_dict({k: T(v) for k, v in fx.make_dnerf_state(33, peaky=False, dx_scale=1.0).items()}, strict=True)
            model = model.to(dtype)
            model.embed_fn = embed_fn
            w0 = {k: p.detach().clone() for k, p in model.named_parameters()}
            qfn = lambda inputs, viewdirs, tt, fn: rr.run_network_dnerf(  # noqa: E731
                inputs, viewdirs, tt, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embed_time_fn=embed_time,
                netchunk=65536, network_type="deformation")
            kw = dict(network_fn=model, network_query_fn=qfn, N_samples=64, perturb=0., N_importance=128, network_fine=None,
                      white_bkgd=True, raw_noise_std=0., use_viewdirs=True)
            grad_vars = list(model.parameters())
            lrate, decay_rate, decay_steps = 5e-4, 0.1, 250 * 1000
            opt = torch.optim.Adam(params=grad_vars, lr=lrate, betas=(0.9, 0.999))
            losses, gnorms = [], []
            for i in range(DTRAJ_STEPS):
                sl = slice(i * DTRAJ_RAYS, (i + 1) * DTRAJ_RAYS)
                rgb, disp, acc, extras = rr.render_dnerf(400, 400, K, T(times[sl]), chunk=32768,
                                                         rays=torch.stack([T(ro[sl]), T(rd[sl])], 0), retraw=True, near=2.,
                                                         far=6., **kw)
                opt.zero_grad()
                loss = torch.mean((rgb - T(target[sl])) ** 2)
                loss = loss + DTRAJ_REG * torch.mean(extras["dx"] ** 2) + DTRAJ_REG * torch.mean(extras["dx0"] ** 2)
                loss = loss + torch.mean((extras["rgb0"] - T(target[sl])) ** 2)
                loss.backward()
                gnorms.append(float(nn.utils.clip_grad_norm_(grad_vars, 1.0)))
                opt.step()
                for g in opt.param_groups:
                    g["lr"] = lrate * (decay_rate ** ((i + 1) / decay_steps))
                losses.append(loss.item())
                print(dtype, i, losses[-1], gnorms[-1], flush=True)
            named = list(model.named_parameters())
            delta = np.array([float((p.detach() - w0[k]).double().norm()) for k, p in named])
            with torch.no_grad():
                sl = slice(0, DTRAJ_RAYS)
                rgb, _, acc, extras = rr.render_dnerf(400, 400, K, T(times[sl]), chunk=32768,
                                                      rays=torch.stack([T(ro[sl]), T(rd[sl])], 0), retraw=True, near=2., far=6.,
                                                      **kw)
            return (np.array(losses), np.array(gnorms), np.array([k for k, _ in named]), delta, rgb.double().numpy(),
                    acc.double().numpy())
        finally:
            torch.set_default_dtype(torch.float32)

    G["losses"], G["grad_norms"], G["names"], G["delta_norms"], rgb, acc = run(torch.float32)
    G["final_rgb"], G["final_acc"] = rgb.astype(np.float32), acc.astype(np.float32)
    G["losses_f64"], G["grad_norms_f64"], _, G["delta_norms_f64"], rgb64, _ = run(torch.float64)
    G["final_rgb_f64"] = rgb64
    print("reference f32 vs f64: loss", np.abs(G["losses"] - G["losses_f64"]) / G["losses_f64"])
    print("   final image", np.abs(G["final_rgb"] - rgb64).max(), "movement",
          (np.abs(G["delta_norms"] - G["delta_norms_f64"]) / G["delta_norms_f64"]).max())
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


def _import_static_reference():
    sys.modules["imageio"] = types.ModuleType("imageio")
    sys.path.insert(0, REF)
    import torch
    import positional_encoding as pe
    import network
    import render as rr
    return torch, pe, network, rr


def _capture_sample_pdf(rr):
    """Wrap the reference's sample_pdf (as render.py:220 calls it) so that each call's (bins, weights) inputs and its
    output are recorded while the reference renders; the original function does the work."""
    calls = []
    orig = rr.sample_pdf

    def recording(bins, weights, N_samples, det=False, pytest=False):
        out = orig(bins, weights, N_samples, det=det, pytest=pytest)
        calls.append((bins.detach().numpy().copy(), weights.detach().numpy().copy(), out.detach().numpy().copy()))
        return out
    rr.sample_pdf = recording
    return calls, orig


def gen_static3(out):
    """Round-3 addition: BASELINE configs[0] (the static2 `c1_*` view, same nets, same pose) rendered once more by the
    imported reference while recording what render.py:215-223 hands to sample_pdf on every ray -- the coarse weights
    `weights[..., 1:-1]` of all 10 000 rays.  Feeding THESE to the resampling kernel gives the reference's fine sample
    positions bit for bit, so a test can hold the fine pass to 1e-4 on identical positions at configs[0]'s full size and
    separate the kernels' error from the conditioning of sample_pdf (which amplifies 1e-7 differences of the coarse
    weights).  Also stored: a float64 checksum of the reference's merged z_vals per ray, and its images for the check."""
    torch, pe, network, rr = _import_static_reference()
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    e10, _ = pe.get_embedder(10, 0)
    e4, _ = pe.get_embedder(4, 0)
    c_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(24))
    f_s = _load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True), fx.make_nerf_state(25))
    H = W = 100
    K, focal = fx.intrinsics(H, W)
    c2w = fx.pose_spherical(0.0, -30.0, 4.0)[:3, :4]
    kw = dict(network_query_fn=lambda i, v, f: rr.run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
              network_fine=f_s, N_samples=64, network_fn=c_s, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.,
              ndc=False, lindisp=False)
    calls, orig = _capture_sample_pdf(rr)
    try:
        with torch.no_grad():
            rgb, disp, acc, extras = rr.render(H, W, K, chunk=32768, c2w=T(c2w), near=2., far=6., **kw)
    finally:
        rr.sample_pdf = orig
    bins = np.concatenate([c[0] for c in calls])
    w = np.concatenate([c[1] for c in calls])
    zs = np.concatenate([c[2] for c in calls])
    assert bins.shape == (H * W, 63) and w.shape == (H * W, 62) and zs.shape == (H * W, 128)
    G["c1_coarse_weights_mid"] = w.astype(np.float32)                 # weights[..., 1:-1] of the coarse pass
    G["c1_z_samples_sum"] = zs.astype(np.float64).sum(-1)             # per-ray checksum of the 128 fine samples
    G["c1_z_samples_first_last"] = np.stack([zs[:, 0], zs[:, 63], zs[:, 127]], -1)
    G["c1_rgb"], G["c1_disp"], G["c1_acc"] = rgb.numpy(), disp.numpy(), acc.numpy()
    G["c1_z_std"] = extras["z_std"].numpy()
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


def gen_trained_scene(out):
    """Round-3 addition: a TRAINED field.  The weights are the 3000-step run of scripts/train_synthetic_scene.py on the
    GPU box (train.py's loop on an analytic scene: unit sphere with a smooth colour pattern, white background;
    gpurun_out/r2_trained_scene_weights.pt, a file this repository's own code wrote).  Here the imported reference
    renders the held-out 200x200 view (theta 37, phi -35) from those weights, in float32 and -- as the yardstick of its
    own rounding error -- in float64; the analytic image of the same view is stored beside it (the common target of the
    'PSNR delta' of the north star).  The closest available stand-in for 'PSNR delta <= 0.05 dB on lego'
    (/root/reference/README.md:57): datasets and checkpoints are absent, so parity against the README's dataset numbers
    stays unpinned."""
    torch, pe, network, rr = _import_static_reference()
    torch.manual_seed(0)
    G = {"meta": _meta(torch)}
    T = torch.from_numpy
    wts = torch.load(os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "r2_trained_scene_weights.pt"),
                     weights_only=True)
    sd = {tag: {k: v.numpy().astype(np.float32) for k, v in wts[tag].items()} for tag in ("coarse", "fine")}
    for tag in sd:
        for k, v in sd[tag].items():
            G[f"{tag}.{k}"] = v
    e10, _ = pe.get_embedder(10, 0)
    e4, _ = pe.get_embedder(4, 0)
    H = W = 200
    K, _ = fx.intrinsics(H, W)
    c2w = fx.pose_spherical(37.0, -35.0, 4.0)

    def shade(ro, rd):    # the analytic scene of scripts/train_synthetic_scene.py
        d = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
        b = (ro * d).sum(-1)
        disc = b * b - ((ro * ro).sum(-1) - 1.0)
        hit = disc > 0
        t = -b - np.sqrt(np.where(hit, disc, 0.0))
        p = ro + t[..., None] * d
        col = 0.5 + 0.5 * np.sin(4.0 * p + np.array([0.0, 2.0, 4.0]))
        return np.where(hit[..., None], col, 1.0).astype(np.float32)
    o, d = fx.pinhole_rays_np(H, W, K, c2w)
    G["truth"] = shade(o.reshape(-1, 3).astype(np.float64), d.reshape(-1, 3).astype(np.float64)).reshape(H, W, 3)
    G["c2w"], G["K"] = c2w[:3, :4].astype(np.float32), K

    def run(dtype):
        torch.set_default_dtype(dtype)
        try:
            nets = [_load(torch, network.NeRF(8, 256, 63, 27, 5, [4], True).to(dtype),
                          {k: v.astype(np.float64 if dtype == torch.float64 else np.float32) for k, v in sd[tag].items()})
                    for tag in ("coarse", "fine")]
            kw = dict(network_query_fn=lambda i, v, f: rr.run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
                      network_fine=nets[1], N_samples=64, network_fn=nets[0], use_viewdirs=True, white_bkgd=True,
                      raw_noise_std=0., ndc=False, lindisp=False)
            with torch.no_grad():
                rgb, disp, acc, _ = rr.render(H, W, K.astype(np.float64 if dtype == torch.float64 else np.float32),
                                              chunk=32768, c2w=T(c2w[:3, :4]).to(dtype), near=2., far=6., **kw)
        finally:
            torch.set_default_dtype(torch.float32)
        return rgb.numpy(), disp.numpy(), acc.numpy()
    G["rgb"], G["disp"], G["acc"] = run(torch.float32)
    print("reference f32 PSNR vs analytic: %.3f dB" % (-10 * np.log10(np.mean((G["rgb"] - G["truth"]) ** 2))))
    r64, d64, a64 = run(torch.float64)
    G["rgb_f64"], G["disp_f64"], G["acc_f64"] = r64.astype(np.float32), d64.astype(np.float32), a64.astype(np.float32)
    dd = np.abs(G["rgb"] - r64).max(-1)
    print("reference f32 vs f64: rgb max %.3e, within 1e-4: %.4f %%" % (dd.max(), 100 * (dd <= 1e-4).mean()))
    np.savez_compressed(out, **G)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024), len(G), "arrays")


if __name__ == "__main__":
    which = sys.argv[1]
    {"trajectory_dnerf": gen_trajectory_dnerf, "trajectory": gen_trajectory, "static": gen_static, "dnerf": gen_dnerf, "static_pytest": gen_static_pytest,
     "tooling": gen_tooling, "static2": gen_static2, "dnerf2": gen_dnerf2, "static3": gen_static3,
     "trained_scene": gen_trained_scene}[which](os.path.join(HERE, f"{which}.npz"))
