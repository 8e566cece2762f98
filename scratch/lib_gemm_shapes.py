"""Which library (non-cgg) GEMMs / convolutions remain in the configs[2] parity-mode training step, by input shape (torch profiler,
record_shapes). usage: python scratch/lib_gemm_shapes.py [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
sys.argv = ['bench.py', '--workload', 'cfg2', '--steps', '1', '--warmup', '2', '--precision', prec]
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402


def patched(args, cfg, model, img, metas, dev, rank, world, steps=None, warmup=None):
    from cgg_amd import synthetic
    from cgg_amd.train import GradReducer, build_optimizer, train_step
    B, (H, W) = args.batch, args.hw
    model.train()
    em = dict(lr_mult=1.0, decay_mult=0.0)
    optimizer = build_optimizer(model, dict(type='AdamW', lr=1e-4, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999),
                                            paramwise_cfg=dict(custom_keys={'backbone': dict(lr_mult=0.1, decay_mult=1.0), 'query_embed': em,
                                                                            'query_feat': em, 'level_embed': em}, norm_decay_mult=0.0)))
    reducer = GradReducer(model, bucket_bytes=args.bucket_mb << 20)
    nc = cfg['panoptic_head']['num_things_classes'] + cfg['panoptic_head']['num_stuff_classes']
    batch = synthetic.train_batch(B, H, W, num_classes=nc, seed=77 + rank, device=dev)
    data = dict(img=img, img_metas=metas, **batch)
    clip = dict(max_norm=0.01, norm_type=2)
    for _ in range(3):
        train_step(model, optimizer, reducer, data, clip)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        train_step(model, optimizer, reducer, data, clip)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, 'self_device_time_total', 0) or getattr(e, 'self_cuda_time_total', 0)
        if t > 100:
            rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    for t, c, k, sh in rows[:130]:
        print(f'{t:9.0f} us {c:4d}  {k:45s} {sh}', flush=True)
    print('---- elementwise / copy ops by call site ----', flush=True)
    rows = []
    for e in prof.key_averages(group_by_stack_n=12):
        t = getattr(e, 'self_device_time_total', 0) or getattr(e, 'self_cuda_time_total', 0)
        if t > 150 and e.key.startswith('aten::') and not any(k in e.key for k in ('mm', 'conv', 'linear')):
            st = [f for f in (e.stack or []) if 'captions_amd' in f or 'cgg_amd' in f]
            rows.append((t, e.count, e.key, ' <- '.join(x.split('/')[-1] for x in st[:3])))
    rows.sort(reverse=True)
    for t, c, k, sh in rows[:80]:
        print(f'{t:9.0f} us {c:4d}  {k:28s} {sh}', flush=True)
    return dict(value=0.0, ms_per_step=0.0, loss=0.0)


bench.train_run = patched
bench.main()
