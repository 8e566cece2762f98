import os, sys, time, torch
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
import torch.nn.functional as F
dev = torch.device('cuda')
torch.backends.cudnn.benchmark = True
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, hw, k in [(64, 64, 256, 3), (128, 128, 128, 3), (256, 256, 64, 3), (64, 256, 256, 1), (128, 512, 128, 1)]:
    x = torch.randn(2, cin, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev).bfloat16()
    z = torch.randn(2, cout, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    pad = [k // 2, k // 2]
    t_plain = timeit(lambda: F.conv2d(x, w, None, padding=k // 2))
    try:
        y1 = torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], pad, [1, 1], 1)
        ref = F.conv2d(x, w, b, padding=k // 2).relu()
        err1 = (y1.float() - ref.float()).abs().max().item()
        t_relu = timeit(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], pad, [1, 1], 1))
    except Exception as e:
        t_relu, err1 = float('nan'), str(e)[:80]
    try:
        y2 = torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, b, [1, 1], pad, [1, 1], 1)
        ref2 = (F.conv2d(x, w, b, padding=k // 2) + z).relu()
        err2 = (y2.float() - ref2.float()).abs().max().item()
        t_add = timeit(lambda: torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, b, [1, 1], pad, [1, 1], 1))
    except Exception as e:
        t_add, err2 = float('nan'), str(e)[:80]
    print('%3d->%3d %3d k%d: conv %.1f us | conv+bias+relu fused %.1f us (err %s, cl=%s) | conv+add+relu fused %.1f us (err %s)' % (
        cin, cout, hw, k, t_plain, t_relu, err1, y1.is_contiguous(memory_format=torch.channels_last) if isinstance(err1, float) else '-', t_add, err2))
