import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import nerf_fixtures as fx
from oracle import oracle as orc
from nerf_amd import ops
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float32)).cuda()
rng=np.random.RandomState(7)
n_rays,S=13,31
pts=rng.uniform(-3,3,(n_rays*S,3)).astype(np.float32)
dirs=rng.normal(size=(n_rays,3)).astype(np.float32); dirs/=np.linalg.norm(dirs,axis=-1,keepdims=True)
g_out=rng.normal(size=(n_rays*S,4)).astype(np.float32)
sd=fx.make_nerf_state(31,peaky=True)
params={k:dev(v) for k,v in sd.items()}
net=ops.PackedNet(params,8,256,63,27,4,(4,),True,mode=0,multires=10,multires_views=4)
raw,acts=net.query_train(dev(pts),dev(dirs),S)
grads=net.query_bwd(dev(g_out),pts.shape[0],acts)
onet=orc.Net(sd)
e=np.concatenate([orc.embed(pts,10),orc.embed(np.repeat(dirs,S,axis=0),4)],-1)
_,ref=onet.forward_backward(e,g_out)
for k,g in ref.items():
    got=grads[k].cpu().numpy().astype(np.float64); sc=np.abs(g).max()+1e-30
    d=np.abs(got-g); print(k, 'err %.2e'%(d.max()/sc), 'argmax', np.unravel_index(d.argmax(), d.shape), 'frac bad %.4f'%((d>1e-3*sc).mean()))
