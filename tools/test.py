#!/usr/bin/env python
"""Minimal inference driver with the command line of the reference's tools/test.py: config + checkpoint -> detector from
the registry -> `simple_test` over a stream of images -> results pickle (`--out`), one process per GPU under
`--launcher pytorch` (images are sharded by rank, results gathered on rank 0). COCO evaluation (`--eval`) needs the
dataset classes, which are out of this build's scope: the option is accepted and reported as unavailable.

Both precision modes serve through `pipeline.StagePipeline` (hipGraph per stage, encode of batch k+1 overlapped with decode of
batch k): `--precision fp32` (default) is parity mode -- the reference's f32 semantics on the f32-class x3 kernels --,
`--precision bf16` the faster throughput mode (narrower than the reference's arithmetic).

    python tools/test.py CONFIG CHECKPOINT --out results.pkl --num-images 64 --synthetic 1024
"""
import argparse
import importlib
import json
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402

import cgg_amd              # noqa: E402,F401
from cgg_amd import registry, runtime, synthetic                          # noqa: E402
from cgg_amd.checkpoint import load_checkpoint                            # noqa: E402
from cgg_amd.config import Config, parse_option_value                     # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='test (and eval) a model')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', help='checkpoint file ("none": keep the initialisation)')
    p.add_argument('--work-dir', help='the directory to save the metrics file')
    p.add_argument('--out', help='output result file in pickle format')
    p.add_argument('--fuse-conv-bn', action='store_true', help='accepted: the bf16 path always folds BN into the convs')
    p.add_argument('--gpu-id', type=int, default=0)
    p.add_argument('--eval', type=str, nargs='+', help='evaluation metrics (needs the dataset classes: unavailable here)')
    p.add_argument('--cfg-options', nargs='+', default=None, help='key=value overrides merged into the config')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    # this build's additions
    p.add_argument('--num-images', type=int, default=16)
    p.add_argument('--samples-per-gpu', type=int, default=2)
    p.add_argument('--synthetic', type=int, default=1024, help='synthetic image size (H = W) when --data is not given')
    p.add_argument('--data', default=None, help='pkg.module:function -> iterable of (img (3,H,W) float tensor, img_meta)')
    p.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                   help='fp32 = parity mode: every contraction in f32-class f16 x 3 arithmetic (as accurate as f32 GEMMs; activations '
                        'must satisfy |a| < 4094 -- the run fails loudly otherwise, CGG_X3=0 lifts the limit); bf16 = throughput mode')
    p.add_argument('--no-pipeline', action='store_true', help='plain sequential simple_test (no graphs / overlap)')
    p.add_argument('--rle', action='store_true',
                   help='results in the evaluation format: masks as COCO RLE dicts (what results2json builds with pycocotools from '
                        'the reference\'s arrays), from bit-packed device masks through pinned staging buffers, one asynchronous '
                        'copy per tensor and the C++ host encoder of the extension, overlapped with the next batch')
    p.add_argument('--mask-bits', action='store_true',
                   help='keep the masks bit-packed ((n, H, W / 8) uint8, pixel x = bit x & 7 of byte x >> 3) in the results: '
                        '8x less PCIe traffic and 8x smaller result files; unpacking 315 MB of bool masks per 1024^2 image on '
                        'the host costs more than the copy it saves (28 vs 36 images/s end to end), so it is opt-in')
    args = p.parse_args(argv)
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))
    return args


def synthetic_images(n, size, seed, pool=4):
    """n images from a pool of `pool` distinct pinned N(0, 1) tensors (drawing 3 x 1024^2 normals on the host costs 20 ms,
    which would make this generator -- not the detector -- the serving bottleneck)."""
    g = torch.Generator().manual_seed(seed)
    meta = synthetic.img_metas(1, size, size)[0]
    imgs = [torch.randn(3, size, size, generator=g) for _ in range(min(pool, n))]
    if torch.cuda.is_available():
        imgs = [t.pin_memory() for t in imgs]
    for i in range(n):
        yield imgs[i % len(imgs)], dict(meta, filename=f'synthetic_{i}.jpg')


def to_numpy(result):
    """device results {type: (labels, bboxes (n,5), masks)} -> numpy, one copy per tensor (masks: (n, H, W) bool, or
    (n, H, W / 8) uint8 with --mask-bits)."""
    out = {}
    for key, val in result.items():
        out[key] = tuple(v.detach().cpu().numpy() if torch.is_tensor(v) else v for v in val) \
            if isinstance(val, (tuple, list)) else (val.detach().cpu().numpy() if torch.is_tensor(val) else val)
    return out


def interleave_rank_results(parts):
    """Per-rank result lists (image i was served by rank i % world) -> one list in dataset order: what [3P] mmcv
    `collect_results` does with `zip(*part_list)` (reference tools/test.py -> apis/test.py:79-130), without its
    padding duplicates because the shards here are not padded to equal length."""
    world = len(parts)
    total = sum(len(p) for p in parts)
    out = []
    for i in range(total + world):
        r, j = i % world, i // world
        if j < len(parts[r]):
            out.append(parts[r][j])
    return out[:total]


def _test_stages():
    """CGG_TEST_STAGES (pipeline stage split of the serving loop): validated up front, 2..5 as `pipeline.detector_pipeline` names them
    (4 and 5 add a separate post-processing stage to the 2- / 3-stage split)."""
    v = os.environ.get('CGG_TEST_STAGES', '3')
    if v not in ('2', '3', '4', '5'):
        raise SystemExit(f'CGG_TEST_STAGES={v!r}: expected 2, 3, 4 or 5')
    return int(v)


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
        dist.init_process_group(backend='nccl')
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        local, rank, world = args.gpu_id, 0, 1
        torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    model = registry.build_detector(cfg.model, test_cfg=cfg.get('test_cfg'))
    meta = {}
    if args.checkpoint and args.checkpoint.lower() != 'none':
        meta = load_checkpoint(model, args.checkpoint, map_location='cpu').get('meta', {})
    if 'CLASSES' in meta:
        model.CLASSES = meta['CLASSES']
    model = model.to(device).eval()
    if args.eval:
        print('--eval: dataset classes / COCO evaluation are outside this build; results are written with --out', flush=True)

    if args.data:
        mod, fn = args.data.split(':')
        stream = getattr(importlib.import_module(mod), fn)(cfg, rank, world)
    else:
        stream = (s for i, s in enumerate(synthetic_images(args.num_images, args.synthetic, seed=11)) if i % world == rank)

    B = args.samples_per_gpu
    collector, in_copy = None, []
    if args.rle:
        from cgg_amd.host_results import RleCollector, fusion_class_counts
        args.mask_bits = True
        collector = RleCollector(device, fusion_class_counts(model.panoptic_fusion_head))

    def emit(device_results):
        """device results of one batch -> `results` (numpy now, or a future of the RLE collector)"""
        if collector is None:
            results.extend(to_numpy(r) for r in device_results)
        else:
            fut = collector.submit(device_results)
            results.append(fut)
            in_copy.append(fut)

    def copies_done(keep=0):
        """host-side wait until all but the `keep` newest batches have been copied out of their device buffers"""
        while len(in_copy) > keep:
            RleCollector.wait_copied(in_copy.pop(0))

    results, pending = [], []          # pending: (pipeline, slot) pairs -- a slot is only meaningful for ITS pipeline
    slot_copied = {}                   # (pipeline, slot) -> event: that slot's last results have been copied to the host
    n_img, t0 = 0, None
    pipe = None
    with torch.no_grad(), runtime.precision_scope(args.precision):
        group = []

        def drain():
            """copy out every batch still in flight, in submission order, from the pipeline that produced it"""
            while pending:
                owner, slot = pending.pop(0)
                emit(owner.wait(slot))

        def run(group):
            nonlocal pipe, t0, n_img
            # straight into a device batch (pinned sources copy asynchronously; torch.stack would build a pageable staging tensor)
            imgs = torch.empty((len(group),) + tuple(group[0][0].shape), dtype=group[0][0].dtype, device=device)
            for k, g in enumerate(group):
                imgs[k].copy_(g[0], non_blocking=True)
            metas = [dict(g[1], batch_input_shape=tuple(imgs.shape[-2:])) for g in group]
            use_pipe = (not args.no_pipeline and len(group) == B
                        and all(m['img_shape'] == metas[0]['img_shape'] and m['ori_shape'] == metas[0]['ori_shape'] for m in metas))
            if use_pipe:
                key = (tuple(imgs.shape), metas[0]['img_shape'], metas[0]['ori_shape'])
                if pipe is None or pipe.meta_key != key:
                    drain()            # the old pipeline's results leave before its buffers are dropped
                    pipe = pipes.get(key)
                    if pipe is None:
                        from cgg_amd.pipeline import detector_pipeline
                        pipe = detector_pipeline(model, imgs, metas, stages=_test_stages(), defer_tail=False, rescale=True, device_results=True, mask_bits=args.mask_bits)
                        pipe.meta_key = key
                        pipes[key] = pipe
                        torch.cuda.synchronize()
                    if t0 is None:
                        t0, n_img = time.perf_counter(), 0
                depth = len(pipe.stages) - 1       # batches that may stay in flight behind the one being submitted
                if collector is not None:
                    copies_done(keep=depth + 1)    # bounds the pinned staging in use; ordering is enforced on the GPU below
                    # the slot this submit reuses: its previous batch's device->host copies must have left before the LAST stage
                    # overwrites the result buffers -- a GPU-side wait on that stage's stream, not a host wait
                    ev = slot_copied.pop((id(pipe), pipe._n % pipe.slots), None)
                    if ev is not None:
                        pipe.streams[-1].wait_event(ev)
                slot = pipe.submit(imgs)
                # `imgs` was filled on this stream but is READ by stage 0 on the pipeline's first stream, after `run` has dropped
                # its reference: without this the caching allocator may hand the block to the next batch's host->device copy
                # while stage 0 has not copied the previous batch yet (batch k served with batch k + 1's pixels)
                imgs.record_stream(pipe.streams[0])
                pending.append((pipe, slot))
                # results of the batch `depth` submits back are copied out while the newer ones run
                while len(pending) > depth:
                    owner, old_slot = pending.pop(0)
                    emit(owner.wait(old_slot))
                    if collector is not None:
                        slot_copied[(id(owner), old_slot)] = in_copy[-1].copied
            else:
                drain()                # keep dataset order: earlier batches first
                if t0 is None:
                    t0 = time.perf_counter()
                emit(model.simple_test(imgs, metas, rescale=True, device_results=True, mask_bits=args.mask_bits))
                if collector is not None:
                    copies_done()
            n_img += len(group)

        pipes = {}                     # one captured pipeline per (batch shape, img_shape, ori_shape)
        for sample in stream:
            # a batch holds images of ONE padded shape (the reference pads a batch to its largest image through the
            # dataset's collate; this driver takes pre-sized tensors and starts a new batch when the shape changes)
            if group and tuple(sample[0].shape) != tuple(group[0][0].shape):
                run(group)
                group = []
            group.append(sample)
            if len(group) == B:
                run(group)
                group = []
        if group:
            run(group)                 # trailing short batch: sequential path (len(group) != B)
        drain()
        torch.cuda.synchronize()
        # parity mode stores its GEMM-consumed activations as f16 x 3 pieces of 16 a: a value with |a| >= 4094 (an un-normalised
        # input, a checkpoint with one huge BN-folded scale) becomes inf / NaN. Every producer raises a device flag; ONE read at
        # the end of the run turns it into an error instead of silently wrong masks (the f32 reference has no such limit)
        if args.precision == 'fp32' and runtime.x3a_enabled():
            from cgg_amd import ops
            if ops.x3_overflow_check(device):
                raise SystemExit('tools/test.py: an activation left the f16 x 3 range of parity mode (|a| >= 4094): results may hold '
                                 'inf / NaN. Re-run with CGG_X3=0 (f32 library path, no range limit).')
        if collector is not None:          # futures -> host results, in submission (= dataset) order
            flat = []
            for r in results:
                flat.extend(r.result())
            results = flat
            collector.close()
    dt = time.perf_counter() - (t0 or time.perf_counter())
    if distributed:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(results, gathered, dst=0)
        if rank == 0:
            results = interleave_rank_results(gathered)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(images=len(results), images_per_sec_this_rank=round(n_img / max(dt, 1e-9), 1),
                              precision=args.precision, pipeline=pipe is not None,
                              results='coco-rle' if args.rle else ('bit-packed' if args.mask_bits else 'bool arrays'))), flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            with open(args.out, 'wb') as f:
                pickle.dump(results, f)
    return results


if __name__ == '__main__':
    main()
