import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import nerf_fixtures as fx
from nerf_amd.network import NeRF
from nerf_amd import _lib
net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in fx.make_nerf_state(21, peaky=True).items()}); net = net.cuda(); net.precision = "bf16"
P = 32768 * 192
pts = (torch.rand(P, 3, device="cuda") * 4 - 2); dirs = torch.nn.functional.normalize(torch.randn(32768, 3, device="cuda"), dim=-1)
lib = _lib.lib(); lib.nrf_bf16_profile_set.argtypes = [C.c_void_p]
cnt = torch.zeros(8, dtype=torch.int64, device='cuda'); lib.nrf_bf16_profile_set(C.c_void_p(cnt.data_ptr()))
with torch.no_grad():
    net.query_points(pts, dirs, 192, 10, 4); torch.cuda.synchronize()
    cnt.zero_()
    net.query_points(pts, dirs, 192, 10, 4); torch.cuda.synchronize()
v = cnt.cpu().numpy().astype(np.float64)[:8] / 256.0
iters = P / 256 / 256
names = ["PE+load", "gemm trunk", "pack trunk", "views head (all)", "kernel total", "alpha gemm", "din load+PE", "rgb gemm"]
for n_, x in zip(names, v): print(f"{n_:20s} {x:12.0f} cycles/WG   {x/iters:9.0f} per tile   {100*x/v[4]:5.1f} %")
