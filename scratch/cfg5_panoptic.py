"""BASELINE configs[4]: COCO-panoptic (80 things + 53 stuff), 800x1344 (1333x800 padded to /32), bf16 stream."""
import sys, time, warnings, torch
sys.path.insert(0, '/root/repo')
import cgg_amd
from cgg_amd import registry, runtime, synthetic
dev = torch.device('cuda')
runtime.set_precision('bf16')
cfg = synthetic.model_config(num_things=80, num_stuff=53, num_unknown=20, num_queries=100, depth=50, panoptic=True)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = registry.build_detector(cfg)
    torch.manual_seed(0)
    model.init_weights()
model = model.to(dev).eval()
B, H, W = 2, 800, 1344
img = torch.randn(B, 3, H, W, device=dev)
metas = synthetic.img_metas(B, H, W)
def step():
    with torch.no_grad():
        return model.simple_test(img, metas, rescale=True, device_results=True)
for _ in range(4): out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('panoptic 800x1344 B=2 eager: %.2f ms/step = %.1f images/s' % (dt * 1e3, B / dt))
print({k: (tuple(v.shape), str(v.dtype)) if torch.is_tensor(v) else type(v) for k, v in out[0].items()})
seg = list(out[0].values())[0]
print('segments', torch.unique(seg).numel())
