import sys, os, json, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import nn as onn, linear as ol
from deep_cartograph_amd import hip
f=np.load('tests/golden/features_164x54.npz'); X=np.ascontiguousarray(f['X'])
Xtr, Xva = X[:120].copy(), X[120:].copy()
st=ol.feature_stats(Xtr); m,r=ol.prepare_normalization(st,'mean_std'); m=m.astype(np.float32); r=r.astype(np.float32)
Xn=ol.normalize(Xtr,m,r); Xvn=ol.normalize(Xva,m,r)
dims=[54,16,8,2]; acts=["leaky_relu","leaky_relu",None]
torch.manual_seed(43)
ref=onn.DeepTICAModel(dims,acts,[0.0,0.0,None],m,r,1e-6)
lins=[mm for mm in ref.nn if isinstance(mm,torch.nn.Linear)]
eng=hip.Mlp("deep_tica",dims,acts,max_batch=32,lag=1,tica_reg=1e-6,lr=1e-3)
eng.set_linears([(l.weight.detach().numpy(),l.bias.detach().numpy()) for l in lins])
Xd=torch.from_numpy(Xn).cuda(); Xvd=torch.from_numpy(Xvn).cuda()
tt=torch.from_numpy(Xtr); tv=torch.from_numpy(Xva)
opt=torch.optim.Adam(ref.parameters(),lr=1e-3)
eng.reset_log(64)
for ep in range(2):
    for r0,b in [(0,32),(32,32),(64,32),(96,23)]:
        eng.train_step(Xd,row0=r0,batch=b)
        opt.zero_grad(); loss,_=ref.step(tt[r0:r0+b],tt[r0+1:r0+1+b]); loss.backward(); opt.step()
        print('train',ep,r0,b,float(loss))
    ref.eval()
    for r0,b in [(0,32),(32,11)]:
        eng.eval_step(Xvd,row0=r0,batch=b)
        with torch.no_grad(): loss,_=ref.step(tv[r0:r0+b],tv[r0+1:r0+1+b])
        print('val',ep,r0,b,float(loss))
    ref.train()
print(eng.read_log()[:,:2])
