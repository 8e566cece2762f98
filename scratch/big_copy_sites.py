"""Where do the large ATen copy / add / cat kernels of the configs[2] training step come from? A TorchDispatchMode prints op, shapes
and the nearest frames of this package for every such op on >= 32 M elements. usage: python scratch/big_copy_sites.py [fp32|bf16]"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
sys.argv = ['bench.py', '--workload', 'cfg2', '--steps', '1', '--warmup', '2', '--precision', prec]
import torch  # noqa: E402
import bench  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

SEEN = collections.Counter()
WATCH = ('copy_', 'clone', 'add', 'add_', 'cat', '_to_copy', 'stack', 'mul', 'sum', 'fill_', 'zero_', 'zeros_like', 'index', 'sub')


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split('.')[0]
        if name in WATCH:
            big = max([a.numel() for a in args if isinstance(a, torch.Tensor)] +
                      [t.numel() for a in args if isinstance(a, (list, tuple)) for t in a if isinstance(t, torch.Tensor)] +
                      ([out.numel()] if isinstance(out, torch.Tensor) else []) + [0])
            if big >= 32 * 1024 * 1024:
                fr = [f for f in traceback.extract_stack() if 'captions_amd' in f.filename or 'cgg_amd' in f.filename]
                site = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in fr[-3:][::-1]) or '(autograd engine)'
                shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
                SEEN[(name, str(shapes)[:90], site)] += 1
        return out


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
    for _ in range(2):
        train_step(model, optimizer, reducer, data, clip)
    torch.cuda.synchronize()
    with Spy():
        train_step(model, optimizer, reducer, data, clip)
    torch.cuda.synchronize()
    for (name, shapes, site), c in sorted(SEEN.items(), key=lambda kv: -kv[1]):
        print(f'SITE {c:3d} x {name:9s} {shapes:92s} {site}', flush=True)
    return dict(value=0.0, ms_per_step=0.0, loss=0.0)


bench.train_run = patched
bench.main()
