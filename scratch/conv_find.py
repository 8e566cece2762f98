import os, sys, time, torch
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
dev = torch.device('cuda')
bench = int(sys.argv[1])
torch.backends.cudnn.benchmark = bool(bench)
shapes = [(64, 64, 256, 1), (128, 128, 128, 1), (256, 256, 64, 1), (512, 512, 32, 1), (128, 128, 256, 2), (256, 256, 128, 2), (512, 512, 64, 2), (256, 256, 256, 1)]
for cin, cout, hw, st in shapes:
    x = torch.randn(2, cin, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    t0 = time.time()
    for _ in range(3): y = F.conv2d(x, w, None, stride=st, padding=1)
    torch.cuda.synchronize(); t1 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = F.conv2d(x, w, None, stride=st, padding=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2 * 2 * (hw // st) ** 2 * cout * cin * 9
    print('bench=%d  %3d->%3d %3dx%3d s%d: %.1f us  %.0f TF/s  (first calls %.1f s)' % (bench, cin, cout, hw, hw, st, ms * 1e3, fl / ms / 1e9, t1 - t0))
