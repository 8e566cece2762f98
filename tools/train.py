#!/usr/bin/env python
"""Minimal training driver with the command line of the reference's tools/train.py (:33-113): the callers' side of the
hot path. It does what that script does AROUND the model -- config file + `--cfg-options`, seed, one
process per GPU under `--launcher pytorch` (torchrun; RCCL), detector from `cfg.model` through the registry, AdamW from
`cfg.optimizer`, `--resume-from` / periodic mmcv-layout checkpoints in `--work-dir` -- and nothing of mmcv's runner /
hook machinery. Datasets are out of this build's scope (no COCO on the box): samples come from a user function
(`--data pkg.module:function`, a generator of raw sample dicts that go through OpenFormatBundle + collate) or, by
default, from the synthetic COCO-shaped stream used by bench.py.

    python tools/train.py configs/instance/coco_b48n17.py --work-dir work --max-iters 100 --synthetic 512
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py CONFIG --launcher pytorch
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402

import cgg_amd              # noqa: E402,F401
from cgg_amd import registry, runtime, synthetic                                        # noqa: E402
from cgg_amd.checkpoint import load_checkpoint, save_checkpoint                         # noqa: E402
from cgg_amd.config import Config, parse_option_value                                   # noqa: E402
from cgg_amd.data_contract import OpenFormatBundle, collate, collect, to_forward_kwargs  # noqa: E402
from cgg_amd.train import GradReducer, LrSchedule, build_optimizer, train_step                    # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train a detector')
    p.add_argument('config', help='train config file path')
    p.add_argument('--work-dir', help='the dir to save logs and models')
    p.add_argument('--resume-from', help='the checkpoint file to resume from')
    p.add_argument('--no-validate', action='store_true', help='accepted for compatibility (no evaluation hook here)')
    p.add_argument('--gpu-id', type=int, default=0, help='id of gpu to use (non-distributed training)')
    p.add_argument('--seed', type=int, default=None, help='random seed')
    p.add_argument('--diff-seed', action='store_true', help='different seeds for different ranks')
    p.add_argument('--deterministic', action='store_true', help='deterministic backend options')
    p.add_argument('--cfg-options', nargs='+', default=None, help='key=value overrides merged into the config')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    p.add_argument('--local_rank', type=int, default=0)
    # this build's additions
    p.add_argument('--max-iters', type=int, default=None, help='stop after this many iterations')
    p.add_argument('--samples-per-gpu', type=int, default=None, help='default: cfg.data.samples_per_gpu or 2')
    p.add_argument('--synthetic', type=int, default=512, help='synthetic sample size (H = W) when --data is not given')
    p.add_argument('--data', default=None, help='pkg.module:function -> iterable of raw sample dicts')
    p.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'])
    p.add_argument('--log-interval', type=int, default=10)
    p.add_argument('--save-interval', type=int, default=0, help='iterations between checkpoints (0: only at the end)')
    args = p.parse_args(argv)
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))
    return args


def synthetic_samples(size, num_classes, seed, vocab=30522):
    """Endless stream of raw samples in the layout the dataset pipeline hands to OpenFormatBundle."""
    i = 0
    while True:
        b = synthetic.train_batch(1, size, size, num_classes=num_classes, vocab=vocab, seed=seed + i)
        rng = np.random.RandomState(seed + i)
        shape = (size, size, 3)
        yield dict(img=rng.randn(*shape).astype(np.float32), filename=f'synthetic_{i}.jpg', ori_filename=f'synthetic_{i}.jpg',
                   ori_shape=shape, img_shape=shape, pad_shape=shape, scale_factor=1.0, flip=False,
                   gt_bboxes=b['gt_bboxes'][0].numpy(), gt_labels=b['gt_labels'][0].numpy(), gt_masks=b['gt_masks'][0].numpy(),
                   gt_caption_ids=b['gt_caption_ids'][0].numpy(), gt_caption_mask=b['gt_caption_mask'][0].numpy(),
                   gt_caption_nouns_ids=b['gt_caption_nouns_ids'][0].numpy(),
                   gt_caption_nouns_mask=b['gt_caption_nouns_mask'][0].numpy())
        i += 1


KEYS = ['img', 'gt_bboxes', 'gt_labels', 'gt_masks', 'gt_caption_ids', 'gt_caption_mask', 'gt_caption_nouns_ids',
        'gt_caption_nouns_mask']


def batches(samples, samples_per_gpu, device):
    bundle = OpenFormatBundle()
    group = []
    for raw in samples:
        formatted = bundle(raw)
        group.append(collect(formatted, [k for k in KEYS if k in formatted]))
        if len(group) == samples_per_gpu:
            batch = collate(group, samples_per_gpu)
            kw = to_forward_kwargs(batch, device)
            for meta, im in zip(kw['img_metas'], kw['img']):
                meta.setdefault('batch_input_shape', tuple(kw['img'].shape[-2:]))
            yield kw
            group = []


def main(argv=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    if args.cfg_options:
        cfg.merge_from_dict({k: parse_option_value(v) for k, v in (kv.split('=', 1) for kv in args.cfg_options)})
    distributed = args.launcher == 'pytorch'
    if distributed:
        import torch.distributed as dist
        local = int(os.environ['LOCAL_RANK'])
        torch.cuda.set_device(local)
        dist.init_process_group(backend='nccl' if torch.cuda.is_available() else 'gloo')
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        local, rank, world = args.gpu_id, 0, 1
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
    device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    # the MODEL is initialised from the same seed on every rank (and rank 0's weights are broadcast below, as DDP does
    # at construction); --diff-seed only offsets the data / sampling RNG, which is re-seeded after the model is built
    base_seed = args.seed if args.seed is not None else 0
    seed = base_seed + (rank if args.diff_seed else 0)
    torch.manual_seed(base_seed)
    np.random.seed(base_seed)
    if args.deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    work_dir = args.work_dir or cfg.get('work_dir') or os.path.join('work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    if rank == 0:
        os.makedirs(work_dir, exist_ok=True)

    model = registry.build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    if hasattr(model, 'init_weights'):
        model.init_weights()
    model = model.to(device).train()
    optimizer = build_optimizer(model, cfg.optimizer)
    schedule = LrSchedule(optimizer, cfg.get('lr_config'))      # stateless in the iteration -> resume restores it
    grad_clip = (cfg.get('optimizer_config') or {}).get('grad_clip')
    start_iter = 0
    resume = args.resume_from or cfg.get('resume_from')
    if resume:
        ck = load_checkpoint(model, resume, map_location='cpu')
        if 'optimizer' in ck:
            optimizer.load_state_dict(ck['optimizer'])
        start_iter = int(ck.get('meta', {}).get('iter', 0))
    elif cfg.get('load_from'):
        load_checkpoint(model, cfg.load_from, map_location='cpu')
    reducer = GradReducer(model)
    reducer.broadcast_parameters(model)          # [3P] DDP construction: rank 0's parameters and buffers everywhere
    torch.manual_seed(seed)
    np.random.seed(seed)

    spg = args.samples_per_gpu or (cfg.get('data') or {}).get('samples_per_gpu', 2)
    head = cfg.model['panoptic_head']
    num_classes = head['num_things_classes'] + head['num_stuff_classes']
    if args.data:
        mod, fn = args.data.split(':')
        samples = getattr(importlib.import_module(mod), fn)(cfg, rank, world)
    else:
        vocab = ((head.get('caption_generator') or {}).get('nb_tokens')) or 30522      # token ids must index the table
        samples = synthetic_samples(args.synthetic, num_classes, seed=1000 * rank + seed, vocab=vocab)
    max_iters = args.max_iters or (cfg.get('runner') or {}).get('max_iters')
    if not max_iters:
        # EpochBasedRunner configs (max_epochs) need the dataset length, which this driver does not have
        raise SystemExit('tools/train.py: the config has no runner.max_iters (epoch-based runner); pass --max-iters N')

    log = open(os.path.join(work_dir, 'train.log.json'), 'a') if rank == 0 else None
    t0 = time.perf_counter()
    it = start_iter
    with runtime.precision_scope(args.precision):
        for data in batches(samples, spg, device):
            if it >= max_iters:
                break
            schedule.apply(it)
            logs = train_step(model, optimizer, reducer, data, grad_clip)
            it += 1
            if (it % args.log_interval == 0 or it == max_iters) and args.precision == 'fp32' and runtime.x3_enabled():
                # parity mode's f16 x 3 kernels hold activations up to |a| < 4094 (csrc/x3.h): beyond it the forward turns into
                # inf / NaN; name the cause once per logging interval instead of letting a NaN loss speak for itself
                from cgg_amd import ops
                over = torch.tensor([int(bool(ops.x3_overflow_check(device, reset=True)))], device=device)
                if distributed:
                    # every rank must take the same branch: a rank that raised alone would leave the others waiting in the next
                    # gradient all-reduce
                    torch.distributed.all_reduce(over, op=torch.distributed.ReduceOp.MAX)
                if int(over.item()):
                    # (what the flag covers: the producers of STORED x3a rows -- cgg_gemm_x3s / cgg_conv_x3s epilogues, the x3a
                    # norm / epilogue kernels. The training GEMMs that split f32 operands in-kernel, cgg_gemm_x3* / cgg_wgrad_x3,
                    # do NOT raise it: gradients carry a per-tensor scale from cgg_absmax_f32, but activations take the fixed 2^4
                    # pre-scale there too -- an |a| >= 4094 in those shows up as inf / NaN losses, not as this message.)
                    raise RuntimeError(f'iteration {it}: a stored x3a activation left the range of the f32-class f16 x 3 format '
                                       '(|a| >= 4094, csrc/x3.h: the x3s GEMM / convolution / norm epilogues raise the flag) on at '
                                       'least one rank -- the losses since the last check are invalid; re-run with CGG_X3A=0 '
                                       '(f32 rows, in-kernel split) or --precision bf16')
            if rank == 0 and (it % args.log_interval == 0 or it == max_iters):
                dt = (time.perf_counter() - t0) / max(it - start_iter, 1)
                rec = dict(iter=it, time=round(dt, 4), lr=optimizer.param_groups[0]['lr'],
                           **{k: round(float(v), 5) for k, v in logs.items()})
                print(json.dumps(rec), flush=True)
                log.write(json.dumps(rec) + '\n')
            if rank == 0 and args.save_interval and it % args.save_interval == 0:
                save_checkpoint(model, os.path.join(work_dir, f'iter_{it}.pth'), optimizer, meta=dict(iter=it))
    if rank == 0:
        save_checkpoint(model, os.path.join(work_dir, 'latest.pth'), optimizer, meta=dict(iter=it, config=args.config))
        log.close()
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()
    return it


if __name__ == '__main__':
    main()
