"""Per-layer gradient deviation from float64 autograd at the contract batch, both arithmetic flavours."""
import copy, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_mlp_gpu import ar_features, normalized, linears_of, push_params, rel_err
from oracle import nn as onn
from deep_cartograph_amd import hip

dims, n, lag, batch = [512, 256, 128, 3], 8300, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Xn, _, _ = normalized(ar_features(n, dims[0], 11))
acts = [sys.argv[3] if len(sys.argv) > 3 else "leaky_relu"] * (len(dims) - 2) + [None]
torch.manual_seed(3)
ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
ref64 = copy.deepcopy(ref).double()
xt = torch.from_numpy(Xn).double()
idx = torch.arange(7, 7 + batch)
loss, _ = ref64.step(xt[idx], xt[idx + lag])
loss.backward()
lins = linears_of(ref64.nn)
Xd = torch.from_numpy(Xn).cuda()
res = {}
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("split", "native")):
    hip.set_gemm_mode(mode)
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
    push_params(eng, linears_of(ref.nn))
    eng.reset_log(2)
    eng.forward(Xd, row0=7, batch=batch)
    eng.backward(Xd, row0=7, batch=batch)
    g = eng.grads_view().cpu().numpy().copy()
    res[mode] = g
    for l, lin in enumerate(lins):
        wo, bo = eng.offsets[l]
        gw = lin.weight.grad.numpy(); gb = lin.bias.grad.numpy()
        print(mode, "layer", l, "W %.2e" % rel_err(g[wo:wo + gw.size].reshape(gw.shape), gw), "b %.2e" % rel_err(g[bo:bo + gb.size], gb))
    eng.close()
if "native" in res and "split" in res: print("native vs split max rel", rel_err(res["native"], res["split"]))
