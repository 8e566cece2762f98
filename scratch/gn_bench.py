import sys, torch
sys.path.insert(0, '/root/repo')
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
B, HW, C = 2, 65536, 256
x = torch.randn(B, HW, C, device=dev).bfloat16()
g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
ws = ops.group_norm_nhwc_workspace(B, HW, 32, dev)
z = torch.empty(B, HW, C, device=dev, dtype=torch.bfloat16)
def f(): ops.group_norm_nhwc(x, g, b, 32, 1e-5, ws, relu=True, out16=(z, 0, HW * C))
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): f()
e1.record(); torch.cuda.synchronize()
print('group_norm_nhwc 2x65536x256 (memset + stats + apply): %.1f us' % (e0.elapsed_time(e1) / 30 * 1e3))
