"""Error budget of the bf16 query on the trained sphere scene (CPU emulation): which rounding costs the PSNR."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import nerf_fixtures as fx
from oracle import oracle as orc
torch.set_num_threads(8)
w = torch.load(os.path.join(R, "gpurun_out", "r2_trained_scene_weights.pt"), weights_only=True)
H = W = 200
K, _ = fx.intrinsics(H, W)
c2w = fx.pose_spherical(37.0, -35.0, 4.0)

def shade(ro, rd):
    d = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    b = (ro * d).sum(-1); disc = b * b - ((ro * ro).sum(-1) - 1.0); hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0)); p = ro + t[..., None] * d
    col = 0.5 + 0.5 * np.sin(4.0 * p + np.array([0.0, 2.0, 4.0]))
    return np.where(hit[..., None], col, 1.0).astype(np.float32)

rays = orc.build_rays(H, W, K, c2w=c2w[:3, :4], near=2., far=6., use_viewdirs=True).reshape(H, W, -1)[::2, ::2].reshape(-1, 11)
truth = shade(rays[:, :3].astype(np.float64), rays[:, 3:6].astype(np.float64))
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)

def emb(x, L):
    return torch.cat([x] + [f(x * 2.0 ** j) for j in range(L) for f in (torch.sin, torch.cos)], -1)

def net_fn(sd, cfg):
    rw = (lambda t: bf(t)) if cfg["w"] else (lambda t: t)
    ra = (lambda t: bf(t)) if cfg["a"] else (lambda t: t)
    ri = (lambda t: bf(t)) if cfg["i"] else (lambda t: t)
    rb = (lambda t: bf(t)) if cfg["b"] else (lambda t: t)
    rh = (lambda t: bf(t)) if cfg["h"] else (lambda t: t)          # the head layers' operands (views layer in/out, rgb/alpha inputs)
    rwh = (lambda t: bf(t)) if (cfg["w"] and cfg["h"]) else (lambda t: t)
    def f(pts, viewdirs, _net):
        N, S, _ = pts.shape
        x = ri(emb(torch.from_numpy(pts.reshape(-1, 3)), 10))
        d = ri(emb(torch.from_numpy(np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3).copy()), 4))
        h = x
        for l in range(8):
            W_, b_ = sd[f"pts_linears.{l}.weight"], sd[f"pts_linears.{l}.bias"]
            h = ra(torch.relu(h @ rw(W_).T + rb(b_)))
            if l == 4:
                h = torch.cat([x, h], -1)
        hh = rh(h) if not cfg["a"] else h
        alpha = hh @ rwh(sd["alpha_linear.weight"]).T + sd["alpha_linear.bias"]
        feat = h @ rw(sd["feature_linear.weight"]).T + rb(sd["feature_linear.bias"])
        feat = rh(feat) if cfg["h"] else feat
        hv = torch.relu(torch.cat([feat, d if cfg["h"] else emb(torch.from_numpy(np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3).copy()), 4)], -1)
                        @ rwh(sd["views_linears.0.weight"]).T + (rb(sd["views_linears.0.bias"]) if cfg["h"] else sd["views_linears.0.bias"]))
        hv = rh(hv) if cfg["h"] else hv
        rgb = hv @ rwh(sd["rgb_linear.weight"]).T + sd["rgb_linear.bias"]
        return torch.cat([rgb, alpha], -1).reshape(N, S, 4).numpy()
    return f

def psnr(cfg):
    t0 = time.time()
    imgs = []
    qc, qf = net_fn(w["coarse"], cfg), net_fn(w["fine"], cfg)
    for i in range(0, rays.shape[0], 2500):
        r = rays[i:i + 2500]
        cur = {"n": 0}
        def q(p, v, n, cur=cur):
            cur["n"] += 1
            return (qc if cur["n"] == 1 else qf)(p, v, n)
        with torch.no_grad():
            out = orc.render_rays(r, None, "fine", N_samples=64, N_importance=128, white_bkgd=True, query=q)
        imgs.append(out["rgb_map"])
    img = np.concatenate(imgs)
    return -10 * np.log10(np.mean((img - truth) ** 2)), img, time.time() - t0

ALL = dict(w=1, a=1, i=1, b=1, h=1)
base, img0, dt = psnr(dict(w=0, a=0, i=0, b=0, h=0))
print(f"f32 emulation: {base:.4f} dB ({dt:.0f} s)", flush=True)
for name, cfg in (("all bf16", ALL), ("inputs exact", dict(ALL, i=0)), ("weights exact", dict(ALL, w=0)),
                  ("trunk activations exact", dict(ALL, a=0)), ("biases exact", dict(ALL, b=0)), ("head layers exact", dict(ALL, h=0)),
                  ("only inputs bf16", dict(w=0, a=0, i=1, b=0, h=0)), ("only weights bf16", dict(w=1, a=0, i=0, b=0, h=0)),
                  ("only trunk activations bf16", dict(w=0, a=1, i=0, b=0, h=0))):
    p, img, dt = psnr(cfg)
    print(f"{name:30s} delta {p - base:+.4f} dB   image-vs-f32 PSNR {-10 * np.log10(np.mean((img - img0) ** 2) + 1e-30):.1f} dB", flush=True)
