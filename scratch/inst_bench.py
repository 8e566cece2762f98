import sys, torch
sys.path.insert(0, '/root/repo')
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
logits = torch.randn(100, 256, 256, device=dev) * 3
sel = torch.arange(100, dtype=torch.int32, device=dev)
for _ in range(3): ops.instance_masks(logits, sel, (1024, 1024), (1024, 1024), (1024, 1024))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.instance_masks(logits, sel, (1024, 1024), (1024, 1024), (1024, 1024))
e1.record(); torch.cuda.synchronize()
print('instance_masks 100 x 1024^2: %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
