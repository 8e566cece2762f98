"""Wall time vs kernel time of the channel-last FPN level (runtime.fpn_level_x3_train) and of the module path at configs[2] shapes
(B = 16, 256 x 256, 256 channels): is the level host-bound? GPU box only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgg_amd  # noqa
from cgg_amd import runtime, synthetic
from cgg_amd.pixel_decoder import MSDeformAttnPixelDecoder
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
cfg = dict(synthetic.model_config(num_things=8, num_stuff=0, num_unknown=0, num_queries=10, enc_layers=1)['panoptic_head']['pixel_decoder'])
cfg.pop('type')
pd = MSDeformAttnPixelDecoder(in_channels=[256, 512, 1024, 2048], strides=[4, 8, 16, 32], feat_channels=256, out_channels=256, **cfg).to(dev).train()
B, H, W = 16, 256, 256
x = torch.randn(B, 256, H, W, device=dev)
lo = torch.randn(B, (H // 2) * (W // 2), 256, device=dev, requires_grad=True)
gm = torch.randn(B, 256, H, W, device=dev) * 1e-5


def rows():
    with runtime.precision_scope('fp32'):
        y = runtime.fpn_level_x3_train(pd, x, lo, (H // 2, W // 2))
    (y * gm).sum().backward()


def module():
    import torch.nn.functional as F
    with runtime.precision_scope('fp32'):
        cur = pd.lateral_convs[0](x)
        y = cur + F.interpolate(lo.view(B, H // 2, W // 2, 256).permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=False)
        y = pd.output_convs[0](y)
        y = pd.mask_feature(y.contiguous())
    (y * gm).sum().backward()


for name, fn in (('rows', rows), ('module', module)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    kt = sum(e.self_device_time_total for e in prof.key_averages()) / 1e3
    print(f'{name}: wall {wall:.2f} ms per fwd+bwd, kernel time {kt:.2f} ms', flush=True)
    rows_ = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)[:14]
    for e in rows_:
        print(f'    {e.self_device_time_total / 1e3:8.2f} ms  {e.count:4d}  {e.key[:90]}')
    cpu = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)[:8]
    for e in cpu:
        print(f'    cpu {e.self_cpu_time_total / 1e3:8.2f} ms  {e.count:4d}  {e.key[:90]}')
