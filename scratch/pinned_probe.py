"""Is torch's pinned host memory sometimes slow to READ on the CPU (uncached / write-combined mapping)? Times numpy reads of a
13-MB pinned tensor, of pageable memory, and of malloc'd memory registered with hipHostRegister."""
import time, sys, os
import numpy as np, torch
torch.cuda.init()
n = 13 * 1024 * 1024
def rd(a, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); s = int(a.view(np.uint64).sum()); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True); pin.zero_()
pag = torch.empty(n, dtype=torch.uint8); pag.zero_()
reg = torch.empty(n, dtype=torch.uint8); reg.zero_()
rc = torch.cuda.cudart().cudaHostRegister(reg.data_ptr(), n, 0)
d = torch.zeros(n, dtype=torch.uint8, device='cuda')
pin.copy_(d, non_blocking=True); reg.copy_(d, non_blocking=True); torch.cuda.synchronize()
print('read 13 MB: pinned (hipHostMalloc) %.2f ms, pageable %.2f ms, registered (hipHostRegister rc=%s, is_pinned=%s) %.2f ms' % (
    rd(pin.numpy()), rd(pag.numpy()), rc, reg.is_pinned(), rd(reg.numpy())), flush=True)
