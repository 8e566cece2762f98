import cProfile, pstats, sys, io, time
sys.path.insert(0, '/root/repo')
import torch, bench
class A: pass
args = A(); args.queries = 100
dev = torch.device('cuda')
import cgg_amd
from cgg_amd import runtime, synthetic
runtime.set_precision('bf16')
cfg, model = bench.build_model(args, dev)
img = torch.randn(2, 3, 1024, 1024, device=dev)
metas = synthetic.img_metas(2, 1024, 1024)
def step():
    with torch.no_grad():
        return model.simple_test(img, metas, rescale=True, device_results=True)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()   # host-only time to enqueue (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue ms/step %.2f, wall ms/step %.2f' % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
