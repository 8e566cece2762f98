import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
class A: pass
args = A(); args.queries = 100
dev = torch.device('cuda')
import cgg_amd
from cgg_amd import runtime, synthetic
from cgg_amd.pipeline import detector_pipeline
runtime.set_precision('bf16')
cfg, model = bench.build_model(args, dev)
B, H, W = 2, 1024, 1024
img = torch.randn(B, 3, H, W, device=dev)
metas = synthetic.img_metas(B, H, W)
with torch.no_grad():
    for _ in range(3): model.simple_test(img, metas, rescale=True, device_results=True)
torch.cuda.synchronize()
for npipe in (1, 2, 3):
    pipes = [detector_pipeline(model, img, metas, stages=2, rescale=True, device_results=True) for _ in range(npipe)]
    for p in pipes:
        for _ in range(2): p.submit(img)
        p.flush()
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for k in range(K):
        pipes[k % npipe].submit(img)
    for p in pipes: p.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d interleaved 2-stage pipelines: %.2f ms/step  %.1f images/s' % (npipe, dt / K * 1e3, B * K / dt))
    del pipes
