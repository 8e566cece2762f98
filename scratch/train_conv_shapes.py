"""Which convolutions of the parity-mode training step run through MIOpen (shapes + device time), via torch.profiler."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from cgg_amd import runtime, synthetic
from cgg_amd.train import build_optimizer, train_step, GradReducer
class A: pass
args = A(); args.size = 1024; args.batch = 16; args.queries = 100; args.precision = 'fp32'
dev = torch.device('cuda', 0)
runtime.set_precision('fp32')
cfg, model = bench.build_model(args, dev)
model.train()
img = torch.randn(16, 3, 1024, 1024, device=dev)
metas = synthetic.img_metas(16, 1024, 1024)
nc = cfg['panoptic_head']['num_things_classes'] + cfg['panoptic_head']['num_stuff_classes']
batch = synthetic.train_batch(16, 1024, 1024, num_classes=nc, seed=77, device=dev)
data = dict(img=img, img_metas=metas, **batch)
opt = build_optimizer(model, dict(type='AdamW', lr=1e-4, weight_decay=0.05))
red = GradReducer(model, bucket_bytes=64 << 20)
def step():
    return train_step(model, opt, red, data, dict(max_norm=0.01, norm_type=2))
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
import sys as _s
pat = _s.argv[1].split(',') if len(_s.argv) > 1 else ['conv']
rows = [e for e in prof.key_averages(group_by_input_shape=True) if any(p_ in e.key.lower() for p_ in pat)]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f'{e.key[:44]:44s} n={e.count:3d} dev {e.device_time_total / 1e3:8.2f} ms  {str(e.input_shapes)[:150]}')
