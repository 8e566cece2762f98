"""cgg_msda_backward alone at configs[2] shapes (B = 16, levels 32^2 / 64^2 / 128^2, 8 heads x 32 channels, 4 points): for rocprofv3
kernel-trace / --pmc passes and timing. Sampling locations = reference point + small learned-offset-like noise (the encoder's
regime); argv[1] = offset std in pixels of the level (default 2.0); argv[2] = iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
std = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, H, D, L, P = 16, 8, 32, 3, 4
hw = [(32, 32), (64, 64), (128, 128)]
N = sum(h * w for h, w in hw)
g = torch.Generator(device='cpu').manual_seed(3)
value = torch.randn(B, N, H, D, generator=g).to(dev)
shapes = torch.tensor(hw, dtype=torch.int64, device=dev)
start = torch.tensor([0, 1024, 1024 + 4096], dtype=torch.int64, device=dev)
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij'), -1).flip(-1).reshape(-1, 2)
                 for h, w in hw], 0)                                            # (N, 2) as (x, y)
off = torch.randn(B, N, H, L, P, 2, generator=g) * std
norm = torch.tensor([[w, h] for h, w in hw], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
loc = (ref.view(1, N, 1, 1, 1, 2) + off / norm).contiguous().to(dev)
attw = torch.softmax(torch.randn(B, N, H, L * P, generator=g), -1).view(B, N, H, L, P).contiguous().to(dev)
gout = torch.randn(B, N, H * D, generator=g).to(dev)
starts = [0, 1024, 1024 + 4096]
def single_pass():
    ops.MSDA_BWD_2P = False
    try:
        return ops.msda_backward_hostlevels(value, hw, starts, loc, attw, gout)
    finally:
        ops.MSDA_BWD_2P = True


for name, fn in (('mmcv-contract entry (device level table, accumulate)', lambda: ops.msda_backward(value, shapes, start, loc, attw, gout)),
                 ('host-level entry, single-pass sorted scatter (round 5)', single_pass),
                 ('host-level entry (autograd path: two-pass sorted scatter, grad_loc / grad_attn written)', lambda: ops.msda_backward_hostlevels(value, hw, starts, loc, attw, gout))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    print(f'msda_backward B=16 (offset std {std} px), {name}: {s.elapsed_time(e) / iters * 1e3:.0f} us per call (incl. zero-fills)')
