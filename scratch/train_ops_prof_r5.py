"""Round 5: which ATen / library ops carry the training step's non-`cgg_*` kernel time? torch.profiler over ONE step of
`bench.py --workload cfg2|cfg3 --precision fp32|bf16`, aggregated by (op, input shapes): self device time, launches.
usage: python scratch/train_ops_prof_r5.py [cfg2|cfg3] [fp32|bf16]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
sys.argv = ['bench.py', '--workload', wl, '--steps', '1', '--warmup', '2', '--precision', prec]
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


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
    agg = collections.defaultdict(lambda: [0, 0.0, ''])
    byop = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        t = ev.self_device_time_total
        if t <= 0:
            continue
        site = ''
        for fr in (ev.stack or []):
            if 'betrayed-by-captions_amd' in fr:
                site = fr.split('betrayed-by-captions_amd/')[-1][:70]
                break
        shp = str(ev.input_shapes)[:90] if ev.input_shapes else ''
        k = (ev.name, shp, site)
        agg[k][0] += 1
        agg[k][1] += t
        byop[ev.name][0] += 1
        byop[ev.name][1] += t
    tot = sum(v[1] for v in byop.values())
    print('== %s %s: device time by op (total %.1f ms) ==' % (wl, prec, tot / 1e3))
    for name, (n, t) in sorted(byop.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%9.1f us %5d  %s' % (t, n, name))
    print('== by (op, shapes, call site) ==')
    for (name, shp, site), (n, t, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:260]:
        print('%9.1f us %5d  %-28s %-90s %s' % (t, n, name[:28], shp, site))
    return dict(value=0)


bench.train_run = patched
bench.main()
