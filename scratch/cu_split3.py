"""3-stream pipeline: encode on most CUs, the narrow query decoder on a few reserved CUs, post-processing unmasked."""
import sys, os, time, warnings, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from cgg_amd import ops, runtime, synthetic
from cgg_amd.pipeline import StagePipeline
class A: pass
args = A(); args.queries = 100
dev = torch.device('cuda:0')
runtime.set_precision('bf16')
cfg, model = bench.build_model(args, dev)
B, H, W = 2, 1024, 1024
img = torch.randn(B, 3, H, W, device=dev)
metas = synthetic.img_metas(B, H, W)
kw = dict(rescale=True, device_results=True)
def run(name, pattern):
    if pattern is None:
        streams = None
    else:
        res = [1 if pattern(c) else 0 for c in range(256)]
        streams = [ops.masked_stream(dev, [1 - r for r in res]), ops.masked_stream(dev, res), torch.cuda.Stream(dev)]
    fns = [lambda x: model.stage_encode(x, defer_tail=0), lambda enc: model.stage_head(enc, metas, **kw),
           lambda out: model.stage_post(out, metas, **kw)]
    with torch.no_grad():
        pipe = StagePipeline(fns, img, streams=streams)
        for _ in range(5): pipe.submit(img)
        pipe.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 60
        for _ in range(n): pipe.submit(img)
        pipe.flush(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print('%-28s %.3f ms/step  %.1f images/s' % (name, dt * 1e3, B / dt), flush=True)
run('no masks (3 streams)', None)
run('decoder on CUs c%8==0', lambda c: c % 8 == 0)
run('decoder on CUs (c>>3)%8==0', lambda c: (c >> 3) % 8 == 0)
run('decoder on CUs (c>>2)%8==0', lambda c: (c >> 2) % 8 == 0)
run('decoder on CUs c%4==0', lambda c: c % 4 == 0)
