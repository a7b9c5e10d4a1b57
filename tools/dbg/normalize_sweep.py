"""Developer probe: dcv_normalize (in place) at 5M x 256 and 10M x 512 for the launch knobs in the environment."""
import os, sys, time, torch
sys.path.insert(0, ".")
from deep_cartograph_amd import hip
for n, F in ((5_000_000, 256), (10_000_000, 512)):
    X = torch.randn(n, F, device="cuda")
    m = torch.zeros(F, device="cuda"); r = torch.ones(F, device="cuda")
    for _ in range(3): hip.normalize(X, m, r, out=X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hip.normalize(X, m, r, out=X)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"RPB={os.environ.get('DCV_NORMALIZE_RPB','-')} NT={os.environ.get('DCV_NORMALIZE_NT','-')} {n}x{F}: {dt*1e3:.3f} ms  {8.0*n*F/dt/1e12:.3f} TB/s")
    del X
