"""Isolated duration of each pipeline stage (its hipGraph replayed back to back on its own stream, nothing else running) next
to the pipelined step: which stage bounds the parity-mode pipeline?  usage: stage_times.py [stages=3]"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import registry, synthetic, runtime
from cgg_amd.pipeline import detector_pipeline
dev = torch.device('cuda')
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=100, depth=50)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = registry.build_detector(cfg)
    torch.manual_seed(0)
    model.init_weights()
model = model.to(dev).eval()
B, H, W = 2, 1024, 1024
img = synthetic.structured_images(B, H, W, seed=5).to(dev)
metas = synthetic.img_metas(B, H, W)
runtime.set_precision('fp32')
with torch.no_grad():
    pipe = detector_pipeline(model, img, metas, stages=nst, defer_tail=0, rescale=True, device_results=True)
def ev():
    return torch.cuda.Event(enable_timing=True)
n = 30
for i, st in enumerate(pipe.streams):
    g = pipe.graphs[i][0]
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for _ in range(3): g.replay()
        s, e = ev(), ev()
        s.record(st)
        for _ in range(n): g.replay()
        e.record(st)
    torch.cuda.synchronize()
    print(f'stage {i} alone: {s.elapsed_time(e) / n:.3f} ms', flush=True)
# pairs of stages concurrently (different slots): how much do they slow each other?
import itertools
for a, b in itertools.combinations(range(len(pipe.streams)), 2):
    torch.cuda.synchronize()
    evs = []
    for i in (a, b):
        st = pipe.streams[i]
        g = pipe.graphs[i][0]
        with torch.cuda.stream(st):
            s, e = ev(), ev()
            s.record(st)
            for _ in range(n): g.replay()
            e.record(st)
        evs.append((s, e))
    torch.cuda.synchronize()
    print(f'stages {a} + {b} together: {evs[0][0].elapsed_time(evs[0][1]) / n:.3f} / {evs[1][0].elapsed_time(evs[1][1]) / n:.3f} ms', flush=True)
torch.cuda.synchronize()
s, e = ev(), ev()
for _ in range(6): pipe.submit(img)
pipe.flush(); torch.cuda.synchronize()
s.record()
for _ in range(n): pipe.submit(img)
pipe.flush()
e.record(); torch.cuda.synchronize()
print(f'pipelined step: {s.elapsed_time(e) / n:.3f} ms')
