import time, torch
torch.cuda.init(); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
n = 160000
idx = torch.randperm(200000)[:n].clone()
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("threads", torch.get_num_threads())
print("randperm           %.3f ms" % t(lambda: torch.randperm(n)))
perm = torch.randperm(n)
print("idx[perm] (cpu)    %.3f ms" % t(lambda: idx[perm]))
g = idx[perm]
print(".to(cuda) pageable %.3f ms" % t(lambda: g.to("cuda")))
pin = torch.empty(n, dtype=torch.int64).pin_memory()
def via_pin():
    pin.copy_(g)
    return pin.to("cuda", non_blocking=True)
print("pinned staging     %.3f ms" % t(via_pin))
idx_d = idx.to("cuda")
def dev_gather():
    pin.copy_(perm)
    return idx_d[pin.to("cuda", non_blocking=True)]
print("perm->pin->gpu gather %.3f ms" % t(dev_gather))
views = lambda: [("idx", gd[i:i + 256], gd) for i in range(0, n, 256)]
gd = g.to("cuda")
print("625 views          %.3f ms" % t(views))
torch.set_num_threads(8)
print("-- 8 threads")
print("randperm           %.3f ms" % t(lambda: torch.randperm(n)))
print("idx[perm] (cpu)    %.3f ms" % t(lambda: idx[perm]))
