import sys, os, json, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
np.set_printoptions(linewidth=200, precision=3)
from oracle import nn as onn, linear as ol
from deep_cartograph_amd import hip
f=np.load('tests/golden/features_164x54.npz'); X=np.ascontiguousarray(f['X'])
Xtr = X[:120].copy()
st=ol.feature_stats(Xtr); m,r=ol.prepare_normalization(st,'mean_std'); m=m.astype(np.float32); r=r.astype(np.float32)
Xn=ol.normalize(Xtr,m,r)
dims=[54,16,8,2]; acts=["leaky_relu","leaky_relu",None]
tt=torch.from_numpy(Xtr)
Xd=torch.from_numpy(Xn).cuda()
torch.manual_seed(43)
ref=onn.DeepTICAModel(dims,acts,[0.0,0.0,None],m,r,1e-6)
ref64=copy.deepcopy(ref).double()
lins=[mm for mm in ref.nn if isinstance(mm,torch.nn.Linear)]
lins64=[mm for mm in ref64.nn if isinstance(mm,torch.nn.Linear)]
eng=hip.Mlp("deep_tica",dims,acts,max_batch=32,lag=1,tica_reg=1e-6,lr=1e-3)
eng.set_linears([(l.weight.detach().numpy(),l.bias.detach().numpy()) for l in lins])
eng.reset_log(8)
eng.forward(Xd,row0=0,batch=32); eng.backward(Xd,row0=0,batch=32)
g=eng.grads_view().cpu().numpy().copy()
loss,_=ref.step(tt[0:32],tt[1:33]); loss.backward()
loss,_=ref64.step(tt[0:32].double(),tt[1:33].double()); loss.backward()
for l in range(3):
    wo,bo=eng.offsets[l]
    n=lins[l].bias.numel()
    print('layer',l,'bias grad eng',g[bo:bo+n]); print('layer',l,'bias grad f32',lins[l].bias.grad.numpy()); print('layer',l,'bias grad f64',lins64[l].bias.grad.numpy())
h=ref64.nn[:5](ref64.norm_in(tt[0:33].double())).detach().numpy()
print('hidden-1 units: fraction of rows positive', (h>0).mean(0))
