import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from deep_cartograph_amd import hip
from deep_cartograph_amd.synth import synth_features
n, F, lag, B = 4_000_000, 512, 10, 524208
X = synth_features(n, F, k_slow=4, device='cuda')
dims = [F, 256, 128, 4]; acts = ['leaky_relu', 'leaky_relu', None]
torch.manual_seed(0)
lins = [(torch.nn.Linear(dims[i], dims[i+1]).weight.detach().numpy().copy(), np.zeros(dims[i+1], np.float32)) for i in range(3)]
for mode in ('split', 'native'):
    hip.set_gemm_mode(mode)
    eng = hip.Mlp('deep_tica', dims, acts, max_batch=B, lag=lag, tica_reg=1e-6, lr=1e-3)
    eng.set_linears(lins); eng.reset_log(64)
    perm = torch.randperm(n - lag, device='cuda')
    idx = perm[:B].contiguous()
    for kind, kw in (('contiguous (rows shared)', dict(row0=0, batch=B)), ('shuffled gather index', dict(idx=idx))):
        for _ in range(3): eng.train_step(X, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): eng.train_step(X, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(mode, kind, f'{dt*1e3:.2f} ms/step', f'{B/dt/1e6:.1f} M frames/s', flush=True)
    eng.close()
