import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import nerf_fixtures as fx
from nerf_amd.network import NeRF
from nerf_amd import ops
G = np.load("tests/golden/static.npz")
def dev(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
def load(seed):
    net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(seed).items()}); return net.cuda()
coarse = load(24)
rays = dev(G["rr_smooth_rays"])
with torch.no_grad():
    z = ops.sample_coarse(rays, 64); pts = ops.points(rays, z).reshape(-1, 3)
    ref = coarse.query_points(pts, rays[:, -3:], 64, 10, 4)
    coarse.precision = "bf16"
    for rep in range(3):
        out = coarse.query_points(pts, rays[:, -3:], 64, 10, 4)
        d = (out - ref).abs().reshape(-1, 64, 4).amax(dim=(1, 2))
        bad = torch.nonzero(d > 0.05 * ref.abs().max()).flatten().tolist()
        print("rep", rep, "max err per ray: max %.4g median %.4g" % (d.max().item(), d.median().item()), "bad rays", bad, "ref scale", ref.abs().max().item())
    if bad:
        r = bad[0]
        print("pts of bad ray", pts.reshape(-1, 64, 3)[r, :3], "dirs", rays[r, -3:])
        print("out", out.reshape(-1, 64, 4)[r, :4]); print("ref", ref.reshape(-1, 64, 4)[r, :4])
        e = (out - ref).abs().reshape(-1, 64, 4)[r].amax(dim=1); print("per-sample err", e)
print("---- fine pass")
fine = load(25)
with torch.no_grad():
    coarse.precision = "fp32"
    raw0 = coarse.query_points(pts, rays[:, -3:], 64, 10, 4).reshape(-1, 64, 4)
    _, _, _, w0, _ = ops.composite_fwd(raw0, z, rays[:, 3:6], None, True)
    _, zf, _ = ops.sample_fine(z, w0, 128, None, want_samples=False)
    pf = ops.points(rays, zf).reshape(-1, 3)
    ref = fine.query_points(pf, rays[:, -3:], 192, 10, 4)
    fine.precision = "bf16"
    out = fine.query_points(pf, rays[:, -3:], 192, 10, 4)
    d = (out - ref).abs().reshape(-1, 192, 4).amax(dim=2)
    print("max err", d.max().item(), "ref scale", ref.abs().max().item())
    bad = torch.nonzero(d > 0.02)
    print("bad (ray, sample):", bad[:40].tolist(), len(bad))
    rgb_r, _, acc_r, _, _ = ops.composite_fwd(ref.reshape(-1, 192, 4), zf, rays[:, 3:6], None, True)
    rgb_o, _, acc_o, _, _ = ops.composite_fwd(out.reshape(-1, 192, 4), zf, rays[:, 3:6], None, True)
    e = (rgb_r - rgb_o).abs().amax(dim=1)
    print("rgb err max", e.max().item(), "ray", e.argmax().item(), "acc", acc_r[e.argmax()].item(), acc_o[e.argmax()].item())
    r = e.argmax().item()
    print("sigma ref", ref.reshape(-1, 192, 4)[r, :, 3].max().item(), "sigma out", out.reshape(-1, 192, 4)[r, :, 3].max().item())
    print("dists", (zf[r, 1:] - zf[r, :-1]).min().item(), (zf[r, 1:] - zf[r, :-1]).max().item())
