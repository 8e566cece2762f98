"""Which Python call sites issue the small elementwise launches of the training step (configs[2], bf16)? torch.profiler with
stacks, one step; aggregates aten ops by the innermost cgg_amd frame."""
import os, sys, collections, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py', '--mode', 'train', '--steps', '1', '--warmup', '2', '--precision', 'bf16']
import bench
from torch.profiler import profile, ProfilerActivity
orig = bench.train_main
def patched(args, cfg, model, img, metas, dev, rank, world):
    from cgg_amd import synthetic
    from cgg_amd.train import GradReducer, build_optimizer, train_step
    B, H, W = args.batch, args.size, args.size
    model.train()
    embed_multi = dict(lr_mult=1.0, decay_mult=0.0)
    optimizer = build_optimizer(model, dict(type='AdamW', lr=1e-4, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999),
        paramwise_cfg=dict(custom_keys={'backbone': dict(lr_mult=0.1, decay_mult=1.0), 'query_embed': embed_multi,
                                        'query_feat': embed_multi, 'level_embed': embed_multi}, norm_decay_mult=0.0)))
    reducer = GradReducer(model, bucket_bytes=args.bucket_mb << 20)
    nc = cfg['panoptic_head']['num_things_classes'] + cfg['panoptic_head']['num_stuff_classes']
    batch = synthetic.train_batch(B, H, W, num_classes=nc, seed=77 + rank, device=dev)
    data = dict(img=img, img_metas=metas, **batch)
    for _ in range(3):
        train_step(model, optimizer, reducer, data, dict(max_norm=0.01, norm_type=2))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        train_step(model, optimizer, reducer, data, dict(max_norm=0.01, norm_type=2))
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.device_time_total <= 0:
            continue
        if ev.name not in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::fill_', 'aten::mul', 'aten::zero_', 'aten::_to_copy', 'aten::sum', 'aten::cat', 'aten::index', 'aten::clamp', 'aten::div'):
            continue
        site = 'autograd / no cgg frame'
        for fr in ev.stack:
            if 'betrayed-by-captions_amd' in fr or 'cgg_amd' in fr:
                site = fr.split('betrayed-by-captions_amd/')[-1][:90]
                break
        agg[(ev.name, site)][0] += 1
        agg[(ev.name, site)][1] += ev.self_device_time_total
    for (name, site), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print('%8.1f us %5d  %-14s %s' % (t, n, name, site))
bench.train_main = patched
bench.main()
