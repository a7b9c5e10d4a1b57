import sys, os, json, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import nn as onn, linear as ol
from deep_cartograph_amd import hip
f=np.load('tests/golden/features_164x54.npz'); X=np.ascontiguousarray(f['X'])
Xtr = X[:120].copy()
st=ol.feature_stats(Xtr); m,r=ol.prepare_normalization(st,'mean_std'); m=m.astype(np.float32); r=r.astype(np.float32)
Xn=ol.normalize(Xtr,m,r)
dims=[54,16,8,2]; acts=["leaky_relu","leaky_relu",None]
torch.manual_seed(43)
ref=onn.DeepTICAModel(dims,acts,[0.0,0.0,None],m,r,1e-6)
ref64=copy.deepcopy(ref).double()
lins=[mm for mm in ref.nn if isinstance(mm,torch.nn.Linear)]
lins64=[mm for mm in ref64.nn if isinstance(mm,torch.nn.Linear)]
for share in (True, False):
    eng=hip.Mlp("deep_tica",dims,acts,max_batch=32,lag=1,tica_reg=1e-6,lr=1e-3)
    eng.set_row_sharing(share)
    eng.set_linears([(l.weight.detach().numpy(),l.bias.detach().numpy()) for l in lins])
    Xd=torch.from_numpy(Xn).cuda()
    tt=torch.from_numpy(Xtr)
    eng.reset_log(8)
    eng.forward(Xd,row0=0,batch=32); eng.backward(Xd,row0=0,batch=32)
    g=eng.grads_view().cpu().numpy().copy()
    for mdl,ll,name in ((ref,lins,'f32'),(ref64,lins64,'f64')):
        for p in mdl.parameters(): p.grad=None
        t = tt if name=='f32' else tt.double()
        loss,_=mdl.step(t[0:32],t[1:33]); loss.backward()
    for l in range(3):
        wo,bo=eng.offsets[l]
        g64=lins64[l].weight.grad.numpy(); g32=lins[l].weight.grad.numpy(); ge=g[wo:wo+g64.size].reshape(g64.shape)
        print('share',share,'layer',l,'max|g64|',np.abs(g64).max(),'eng-f64',np.abs(ge-g64).max(),'f32-f64',np.abs(g32-g64).max(),
              'sign flips eng',int((np.sign(ge)!=np.sign(g64)).sum()),'f32',int((np.sign(g32)!=np.sign(g64)).sum()), 'n',g64.size)
    eng.close()
