#!/usr/bin/env python3
"""Timing probe of the fused small-network step: eval steps (staging + forward + loss) vs training steps (+ backward,
partials, reduction + optimiser) at the C2 shape, wall time per step over back-to-back launches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deep_cartograph_amd import hip  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims = [128, 64, 32, 2, 32, 64, 128]
acts = ["leaky_relu", "leaky_relu", None, "leaky_relu", "leaky_relu", None]
torch.manual_seed(0)
X = torch.randn(200000, 128, device="cuda")
eng = hip.Mlp("ae", dims, acts, max_batch=bs, latent_layer=3, lr=1e-3)
lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(6)]
eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in lins])
eng.set_feature_range(np.ones(128, np.float32))
eng.reset_log(20000)
for name, fn in (("eval", eng.eval_step), ("train", eng.train_step)):
    for i in range(50):
        fn(X, row0=(i % 40) * bs, batch=bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for i in range(n):
        fn(X, row0=(i % 40) * bs, batch=bs)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter()
    for i in range(100):   # into an empty queue: the host's own cost per step
        fn(X, row0=(i % 40) * bs, batch=bs)
    t_host = (time.perf_counter() - tt) / 100
    torch.cuda.synchronize()
    print(f"{name}: host cost of 100 enqueues into an empty queue {t_host * 1e6:.1f} us/step")
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us/step (batch {bs}, DCV_SNET_TR={os.environ.get('DCV_SNET_TR', 'auto')}); "
          f"host enqueue alone {t_enq / n * 1e6:.1f} us/step", flush=True)
eng.close()

if os.environ.get("DCV_SNET_STAMPS") == "1":
    import ctypes as C

    from deep_cartograph_amd import _lib

    eng = hip.Mlp("ae", dims, acts, max_batch=bs, latent_layer=3, lr=1e-3)
    eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in lins])
    eng.set_feature_range(np.ones(128, np.float32))
    eng.reset_log(64)
    for i in range(20):
        eng.train_step(X, row0=0, batch=bs)
    st = (C.c_uint64 * 64)()
    fn = _lib.load().dcv_debug_snet_stamps
    fn.argtypes = [C.c_void_p, C.c_void_p]
    rc = fn(eng.h, st)
    v = list(st)
    t0 = v[0]
    names = {0: "start", 1: "weights staged", 2: "input tile staged", 20: "sse reduced", 60: "backward done", 61: "ticket taken"}
    for k in sorted(range(64), key=lambda k: v[k]):
        if v[k]:
            if k in names:
                nm = names[k]
            elif 3 <= k < 20:
                nm = f"forward layer {k - 3} done"
            elif 21 <= k < 40:
                nm = f"layer {(k - 21) // 2}: wgrad + bias partials issued"
            elif 40 <= k < 48:
                nm = f"layer {k - 40}: dgrad MFMAs done"
            else:
                nm = f"layer {k - 48}: wgrad tiles done"
            print(f"  stamp {k:2d} {nm:38s} +{(v[k] - t0) * 0.01:7.2f} us")
    if v[12] and v[13]:
        print(f"  shader clock over the kernel: {(v[13] - v[12]) / ((v[60] - v[0]) * 10.0):.3f} GHz (s_memtime / s_memrealtime)")
    eng.close()
