import sys, time, torch, warnings
sys.path.insert(0, '.')
import cgg_amd
from cgg_amd import registry
cfg = dict(type='ResNet', depth=50, num_stages=4, out_indices=(0,1,2,3), frozen_stages=3, norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch')
m = registry.build_backbone(cfg).cuda().eval()
x = torch.randn(2,3,1024,1024, device='cuda')
import contextlib
def run(dtype, cl, bench):
    torch.backends.cudnn.benchmark = bench
    xx = x.contiguous(memory_format=torch.channels_last) if cl else x
    mm = m.to(memory_format=torch.channels_last) if cl else m.to(memory_format=torch.contiguous_format)
    ctx = torch.autocast('cuda', dtype=dtype) if dtype is not None else contextlib.nullcontext()
    def f():
        with torch.no_grad(), ctx:
            y = mm.maxpool(mm.relu(mm.bn1(mm.conv1(xx))))
            outs = []
            for n in mm.res_layers:
                y = getattr(mm, n)(y); outs.append(y)
        return outs
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/5*1e3
for dtype in (None, torch.bfloat16, torch.float16):
    for cl in (False, True):
        for bench in (False, True):
            try:
                print(dtype, 'CL' if cl else 'NCHW', 'bench' if bench else 'nobench', '%.2f ms' % run(dtype, cl, bench), flush=True)
            except Exception as e:
                print('ERR', dtype, cl, bench, e)
