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
tt=torch.from_numpy(Xtr)
Xd=torch.from_numpy(Xn).cuda()
for mode in ('row','idx','row_noshare'):
    torch.manual_seed(43)
    ref=onn.DeepTICAModel(dims,acts,[0.0,0.0,None],m,r,1e-6)
    lins=[mm for mm in ref.nn if isinstance(mm,torch.nn.Linear)]
    eng=hip.Mlp("deep_tica",dims,acts,max_batch=32,lag=1,tica_reg=1e-6,lr=1e-3)
    if mode=='row_noshare': eng.set_row_sharing(False)
    eng.set_linears([(l.weight.detach().numpy(),l.bias.detach().numpy()) for l in lins])
    opt=torch.optim.Adam(ref.parameters(),lr=1e-3)
    eng.reset_log(8)
    for step,(r0,b) in enumerate([(0,32),(32,32),(64,32)]):
        if mode=='idx': eng.train_step(Xd,idx=torch.arange(r0,r0+b).cuda())
        else: eng.train_step(Xd,row0=r0,batch=b)
        g=eng.grads_view().cpu().numpy().copy()
        opt.zero_grad(); loss,_=ref.step(tt[r0:r0+b],tt[r0+1:r0+1+b]); loss.backward()
        for l in range(3):
            wo,bo=eng.offsets[l]
            gw=lins[l].weight.grad.numpy(); gb=lins[l].bias.grad.numpy()
            print(mode,'step',step,'layer',l,'dgW',np.abs(g[wo:wo+gw.size].reshape(gw.shape)-gw).max(),'dgb',np.abs(g[bo:bo+gb.size]-gb).max(),'gb',gb, 'eng gb',g[bo:bo+gb.size] if l==2 else '')
        opt.step()
        got=eng.get_linears()
        for l in range(3):
            dw=np.abs(got[l][0]-lins[l].weight.detach().numpy()); db=np.abs(got[l][1]-lins[l].bias.detach().numpy())
            print(mode,'step',step,'layer',l,'dW',dw.max(),'n>1e-4',int((dw>1e-4).sum()),'db',db.max(),'n>1e-4',int((db>1e-4).sum()))
        print(mode,'loss eng',eng.read_log()[step,0],'oracle',float(loss.detach()))
    eng.close()
