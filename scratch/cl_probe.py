"""Probe: torch GroupNorm / 1x1 convolution / bilinear up-sampling on contiguous vs channels_last (B, 256, 256, 256) maps, forward +
backward, f32 and bf16 -- does the library path of the FPN's finest level run without layout copies in channels_last? GPU box only."""
import torch, torch.nn.functional as F
dev = 'cuda'


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, C, H, W = 16, 256, 256, 256
for dt in (torch.float32, torch.bfloat16):
    for cl in (False, True):
        x = torch.randn(B, C, H, W, device=dev, dtype=dt)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_()
        gn = torch.nn.GroupNorm(32, C).to(dev)
        conv = torch.nn.Conv2d(C, C, 1, bias=False).to(dev).to(dt)
        if cl:
            conv = conv.to(memory_format=torch.channels_last)

        def f_gn():
            y = gn(x.float())
            y.backward(torch.ones_like(y))
            x.grad = None

        def f_conv():
            y = conv(x)
            y.backward(torch.ones_like(y))
            x.grad = None

        def f_up():
            xs = x[:, :, ::2, ::2]
            xs = xs.contiguous(memory_format=torch.channels_last) if cl else xs.contiguous()
            y = F.interpolate(xs.float(), size=(H, W), mode='bilinear', align_corners=False)
            return y

        y = gn(x.float())
        print(f'{dt} channels_last={cl}: GN out CL={y.is_contiguous(memory_format=torch.channels_last)}  GN f+b {timeit(f_gn):.2f} ms   '
              f'conv1x1 f+b {timeit(f_conv):.2f} ms  conv out CL={conv(x).is_contiguous(memory_format=torch.channels_last)}   '
              f'upsample fwd {timeit(f_up):.2f} ms', flush=True)
        del x, gn, conv, y
