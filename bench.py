#!/usr/bin/env python
"""bench.py -- images/sec of the CGG decoder + mask-prediction hot path on MI355X.

A "step" = one forward of `Mask2FormerOpen.simple_test` (R50 backbone -> MSDeformAttn pixel decoder ->
9-layer masked-attention query decoder -> mask logits -> open-vocabulary instance post-processing for
all / novel / base class sets) over one synthetic COCO-shaped batch that is already resident in HBM;
results stay on the device (BASELINE.json configs[1]: R50, 100 queries, 1024x1024, batch 2, forward-only).
N > 1: one process per GPU (torch.distributed / RCCL only for the barrier), every rank runs its own
replica on its own batch -> weak scaling, no data-path collective (inference: "replicas only").

Launch: `python bench.py --gpus N` starts the N ranks ITSELF (child processes created before the parent touches the
GPU; rank r binds device r, RCCL for the barrier / max-over-ranks); under an external launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) RANK / LOCAL_RANK / WORLD_SIZE come from
the environment and `--gpus` must agree with WORLD_SIZE (it fails loudly otherwise).

Prints ONE JSON line (rank 0). Extra objects:
  roofline      -- the mask-logit kernel (einsum 'bqc,bchw->bqhw', cgg_mask_logits, full resolution),
                   timed with events on the launch stream INSIDE the timed steps.
  cpu_baseline  -- the oracle (torch CPU restatement of the reference path, kind "port") on one batch of the
                   same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time
import warnings

# MIOpen's find step otherwise times its naive reference convolutions (15-30 ms each) on the first step
for _k in ('FWD', 'BWD', 'WRW'):
    os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_' + _k, '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0


def build_model(args, dev):
    import cgg_amd
    from cgg_amd import registry, synthetic
    cfg = synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=args.queries, depth=50)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
        torch.manual_seed(0)
        model.init_weights()
    return cfg, model.to(dev).eval()


def cpu_baseline(args, cfg, model, img_cpu):
    """oracle path on the host: the SAME weights; backbone = the same plain-torch ResNet. One un-timed warm-up pass, then
    the median of 3 timed passes over ONE image of the benched workload, plus SURVEY 8(d)'s cfg-1 case (512 x 512,
    batch 1) timed the same way. `cores` = host cores of the box, `threads` = torch intra-op threads actually used."""
    import copy
    import statistics
    from oracle import head as OH
    from cgg_amd import runtime, synthetic
    hc = copy.deepcopy(cfg['panoptic_head'])
    hc.update(train_cfg=cfg['train_cfg'], test_cfg=cfg['test_cfg'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        orc = OH.OracleHead(**hc)
    sd = {k: v.detach().cpu() for k, v in model.panoptic_head.state_dict().items()}
    orc.load_state_dict(sd)
    orc.eval()
    backbone = copy.deepcopy(model.backbone).cpu().eval()
    fh = model.panoptic_fusion_head
    embs = [fh.all_class_embs.cpu(), fh.novel_class_embs.cpu(), fh.base_class_embs.cpu()]
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    torch.set_num_threads(threads)

    def one_pass(x):
        B, _, H, W = x.shape
        metas = synthetic.img_metas(B, H, W)
        t0 = time.perf_counter()
        with torch.no_grad(), runtime.precision_scope('fp32'):      # the CPU reference path is plain f32 torch
            feats = backbone(x)
            _, emb, up = orc.simple_test(list(feats), metas)
            for b in range(B):
                mp = OH.crop_rescale(up[b], metas[b], True)
                for e in embs:
                    OH.instance_postprocess_emb(emb[b], mp, e, 100)
        return time.perf_counter() - t0

    def timed(x, reps=3):
        one_pass(x)                                                  # warm-up: allocator, thread pool, first-touch
        ts = [one_pass(x) for _ in range(reps)]
        return statistics.median(ts), ts

    full = img_cpu[:1]                      # bounded sample: ONE image of the same workload
    H, W = full.shape[-2:]
    dt, ts = timed(full)
    small = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(4321))
    dt1, ts1 = timed(small)
    return dict(value=1.0 / dt, unit='images/sec', cores=cores, threads=threads, kind='port',
                sample=f'1 image {H}x{W} of the benched workload, full detector forward + instance post-processing '
                       f'(torch CPU oracle, fp32, {threads} threads on {cores} host cores): 1 warm-up + median of 3 '
                       f'timed passes ({", ".join("%.2f" % t for t in ts)} s)',
                cfg1_512=dict(value=1.0 / dt1, unit='images/sec',
                              sample='configs[0]: one 512x512 image, same pipeline, 1 warm-up + median of 3 '
                                     f'({", ".join("%.2f" % t for t in ts1)} s)'))


def pmc_traffic(kernel, B, H, W):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r1_pmc_{fetch,write}_*.csv:
    separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of scratch/kernel_only.py at configs[1] shapes). Units are KB;
    FETCH_SIZE is doubled (gfx950 tallies 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md "HBM").
    Counters cannot be collected from inside this process, so the value is only reported for the shapes they were
    collected at; otherwise null."""
    import csv
    if (B, H, W) != (2, 1024, 1024):
        return None, None
    tot = {}
    rnd = 'r2' if os.path.exists(os.path.join(ROOT, 'profiles', 'r2_pmc_fetch_counter_collection.csv')) else 'r1'
    for name, cnt, mult in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
        path = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_{name}_counter_collection.csv')
        if not os.path.exists(path):
            return None, None
        vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
                if kernel in r['Kernel_Name'] and r['Counter_Name'] == cnt]
        if not vals:
            return None, None
        tot[cnt] = sum(vals) / len(vals) * 1024.0 * mult
    return tot['FETCH_SIZE'] + tot['WRITE_SIZE'], (f'profiles/{rnd}_pmc_' + '{fetch,write}_counter_collection.csv: separate '
                                                   'rocprofv3 --pmc passes; FETCH_SIZE x2 (gfx950 correction), KB units')


def rocprof_launch_mean(kernel):
    """Per-launch mean of `kernel`'s full-resolution launches in the committed rocprofv3 kernel trace of this command
    (profiles/r2_hot_kernel_launches.json, written by scratch/publish_profiles.py) -- the cross-check for `launch_ms`."""
    for rnd in ('r2', 'r1'):
        path = os.path.join(ROOT, 'profiles', f'{rnd}_hot_kernel_launches.json')
        if os.path.exists(path):
            try:
                rec = json.load(open(path)).get(kernel)
            except Exception:
                rec = None
            if rec:
                return dict(rec, source=os.path.relpath(path, ROOT))
    return None


def host_results_rate(args, model, img, metas, dev):
    """images/sec of the same step when the results must reach the HOST in the evaluation format: the
    staged pipeline (same stage split as the headline run) with bit-packed masks, ONE asynchronous device->host copy per result tensor into pinned staging
    buffers, COCO RLE on the extension's host threads overlapped with the next batch (host_results.RleCollector).
    PCIe-inclusive; never `value`. Also reports the mean number of runs per mask (encoder cost is per run: the
    random-weight masks of this benchmark are far noisier than a trained model's)."""
    from cgg_amd.host_results import RleCollector, fusion_class_counts
    from cgg_amd.pipeline import detector_pipeline
    nstage = min(max(args.pipeline, 2), 3)
    pipe = detector_pipeline(model, img, metas, stages=nstage, defer_tail=args.defer_tail, rescale=True,
                             device_results=True, mask_bits=True)
    col = RleCollector(dev, fusion_class_counts(model.panoptic_fusion_head))
    steps = max(args.steps, 8)

    def run(n):
        # `depth` batches stay in flight behind the one being submitted; a slot's result buffers are overwritten by the LAST
        # stage of the batch that reuses it, which waits (on the GPU) for the slot's previous device->host copies
        depth = nstage - 1
        futs, in_copy, pending, copied = [], [], [], {}
        for _ in range(n):
            while len(in_copy) > depth + 1:
                RleCollector.wait_copied(in_copy.pop(0))
            ev = copied.pop(pipe._n % pipe.slots, None)
            if ev is not None:
                pipe.streams[-1].wait_event(ev)
            pending.append(pipe.submit(img))
            while len(pending) > depth:
                old = pending.pop(0)
                f = col.submit(pipe.wait(old))
                futs.append(f)
                in_copy.append(f)
                copied[old] = f.copied
        for old in pending:
            futs.append(col.submit(pipe.wait(old)))
        return [r for f in futs for r in f.result()]
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    col.close()
    rles = [r for im in res for key in im for cls in im[key][1] for r in cls]
    nbytes = sum(len(r['counts']) for r in rles)
    return dict(value=len(res) / dt, unit='images/sec (this rank, results on the host as COCO RLE)', steps=steps,
                masks_per_image=len(rles) / max(len(res), 1), rle_bytes_per_mask=nbytes / max(len(rles), 1),
                how=f'{nstage}-stage pipeline, bit-packed masks, pinned async D2H, C++ RLE on host threads overlapped with the next batch')


def parity_mode_rate(args, model, img, metas, dev):
    """images/sec of the SAME step in parity mode (`--precision fp32`: f32 GEMMs / values, 3x-bf16-split MFMA mask
    logits, f32 MFMA attention -- the mode the 1e-3 / bit-exact parity tests run in), timed in this process right after
    the headline region: the step captured once into a hipGraph and replayed K times (eager if capture fails)."""
    from cgg_amd import runtime
    with runtime.precision_scope('fp32'):
        def step():
            with torch.no_grad():
                return model.simple_test(img, metas, rescale=True, device_results=True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        graph, how, pipe = None, 'eager launches', None
        if args.graph and args.pipeline:
            # the same staged pipeline as the headline run (stages captured under the fp32 scope)
            try:
                from cgg_amd.pipeline import detector_pipeline
                nst = min(max(args.pipeline, 2), 3)
                pipe = detector_pipeline(model, img, metas, stages=nst, defer_tail=args.defer_tail, rescale=True,
                                         device_results=True)
                for _ in range(nst):
                    pipe.submit(img)
                pipe.flush()
                torch.cuda.synchronize()
                how = f'{nst}-stage software pipeline across steps, as the headline run'
            except Exception as e:
                print(f'bench.py: parity-mode pipeline setup failed ({type(e).__name__}: {e}); one graph per step', file=sys.stderr)
                pipe = None
                torch.cuda.synchronize()
        if args.graph and pipe is None:
            try:
                graph = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()
                torch.cuda.current_stream().wait_stream(side)
                with torch.cuda.graph(graph):
                    step()
                graph.replay()
                torch.cuda.synchronize()
                how = 'one hipGraph per step, replayed back to back'
            except Exception as e:
                print(f'bench.py: parity-mode hipGraph capture failed ({type(e).__name__}: {e}); eager', file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        steps = max(args.steps // 2, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pipe is not None:
            for _ in range(steps):
                pipe.submit(img)
            pipe.flush()
        else:
            for _ in range(steps):
                graph.replay() if graph is not None else step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pipe = None
    return dict(value=img.shape[0] * steps / dt, unit='images/sec (this rank)', ms_per_step=dt / steps * 1e3, steps=steps,
                precision='fp32', how=how)


def train_main(args, cfg, model, img, metas, dev, rank, world):
    """configs[2]: one optimisation step = forward_train (all 10 layers' losses incl. grounding + caption
    generation) -> backward with bucketed gradient all-reduce over RCCL overlapped -> clip -> AdamW step."""
    import torch.distributed as dist
    from cgg_amd import synthetic
    from cgg_amd.train import GradReducer, build_optimizer, train_step
    B, H, W = args.batch, args.size, args.size
    model.train()
    embed_multi = dict(lr_mult=1.0, decay_mult=0.0)
    optimizer = build_optimizer(model, dict(          # configs/instance/coco_b48n17.py:270-286
        type='AdamW', lr=1e-4, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999),
        paramwise_cfg=dict(custom_keys={'backbone': dict(lr_mult=0.1, decay_mult=1.0), 'query_embed': embed_multi,
                                        'query_feat': embed_multi, 'level_embed': embed_multi},
                           norm_decay_mult=0.0)))
    reducer = GradReducer(model, bucket_bytes=args.bucket_mb << 20)
    nc = cfg['panoptic_head']['num_things_classes'] + cfg['panoptic_head']['num_stuff_classes']
    batch = synthetic.train_batch(B, H, W, num_classes=nc, seed=77 + rank, device=dev)
    data = dict(img=img, img_metas=metas, **batch)
    clip = dict(max_norm=0.01, norm_type=2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        logs = train_step(model, optimizer, reducer, data, clip)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = train_step(model, optimizer, reducer, data, clip)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if args.shared_devices else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        nparam = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(json.dumps(dict(
            metric='images/sec (COCO-instance training step, 1024x1024, 100 queries)',
            value=B * world * args.steps / dt, unit='images/sec', n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak',
            vs_baseline=None, dtype='bf16' if args.precision == 'bf16' else 'f32', data='synthetic',
            config=dict(workload=f'configs[2]: R50 + {args.queries} queries, {H}x{W}, batch {B}/GPU, training step '
                                 '(forward_train with grounding + caption-generation losses, backward, gradient '
                                 'all-reduce, clip, AdamW)',
                        global_batch=B * world, parallelism=f'dp{world}', precision=args.precision,
                        trainable_params=nparam, grad_buckets=len(reducer.buckets), bucket_mb=args.bucket_mb),
            loss=logs.get('loss'), peak_mem_gb=torch.cuda.max_memory_allocated() / 2**30)))
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the
    torch.distributed environment (rendezvous on 127.0.0.1), BEFORE this process makes any GPU call (a process that
    has initialised the GPU must not exec / fork GPU work on this pool; `device_count()` does not initialise it).
    Rank 0's stdout (the JSON line) is passed through; the exit code is the first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    bad = [c for c in codes if c != 0]
    if bad:
        raise SystemExit(f'bench.py: rank exit codes {codes}')
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="infer = configs[1] (the metric's config); train = configs[2] training step")
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (2 infer / 16 train)')
    ap.add_argument('--bucket-mb', type=int, default=64, help='gradient all-reduce bucket size (train)')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--queries', type=int, default=100)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--graph', type=int, default=1,
                    help='1: capture the step once and replay it from a hipGraph (default; the eager step is '
                         'host-bound at ~530 launches); 0: eager launches')
    ap.add_argument('--defer-tail', type=int, default=0, choices=[0, 1, 2],
                    help='pipeline stage balancing: K/V projections + mask-feature packing run in the decode stage')
    ap.add_argument('--pipeline', type=int, default=3, choices=[0, 2, 3, 4, 5],
                    help='(with --graph 1) software pipeline across steps, one HIP stream + hipGraph per stage: '
                         '3 = backbone | pixel decoder + K/V | query decoder + post-processing, 2 = the first two '
                         'merged, 0 = one graph per step replayed back to back')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-results', type=int, default=1,
                    help='1 (default): also time the step with results delivered to the host as COCO RLE (reported as '
                         '`host_results`, never `value`)')
    ap.add_argument('--no-parity-mode', action='store_true',
                    help='skip the fp32 (parity-mode) timing of the same step that is reported as config.parity_mode_value')
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 2 if args.mode == 'infer' else 16

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return spawn_ranks(args)             # parent: starts the ranks, never touches the GPU itself
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; they must agree '
                         '(run `python bench.py --gpus N` alone, or torch.distributed.run --nproc-per-node N ... --gpus N)')
    ndev = torch.cuda.device_count()         # does not initialise the GPU
    if ndev < 1:
        raise SystemExit('bench.py needs a ROCm device (the hot path has no CPU implementation)')
    shared = world > ndev                    # fewer devices than ranks (1-GPU dev box): ranks share devices, dry run only
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device('cuda', local_rank % ndev)
    import torch.distributed as dist
    if world > 1:
        if shared:
            # RCCL refuses two ranks on one device: the barrier / max-over-ranks go over gloo; the line says so
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=dev)
    args.shared_devices = shared

    import cgg_amd
    from cgg_amd import ops, runtime, synthetic
    runtime.set_precision(args.precision)
    cfg, model = build_model(args, dev)
    B, H, W = args.batch, args.size, args.size
    g = torch.Generator().manual_seed(1234 + rank)
    img_cpu = torch.randn(B, 3, H, W, generator=g)
    img = img_cpu.to(dev)
    metas = synthetic.img_metas(B, H, W)

    if args.mode == 'train':
        return train_main(args, cfg, model, img, metas, dev, rank, world)

    def step():
        with torch.no_grad():
            return model.simple_test(img, metas, rescale=True, device_results=True)

    def barrier():
        torch.cuda.synchronize()             # (gloo dry run: the device work must be done before the host barrier)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()

    graph = None
    pipe = None
    if args.graph and args.pipeline:
        try:
            from cgg_amd.pipeline import detector_pipeline
            pipe = detector_pipeline(model, img, metas, stages=args.pipeline, defer_tail=args.defer_tail,
                                     rescale=True, device_results=True)
            for _ in range(args.pipeline):
                pipe.submit(img)
            pipe.flush()
            torch.cuda.synchronize()
        except Exception as e:
            print(f'bench.py: stage pipeline setup failed ({type(e).__name__}: {e}); falling back to one graph '
                  'per step', file=sys.stderr)
            pipe = None
            args.pipeline = 0
            torch.cuda.synchronize()
    if args.graph and pipe is None:
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(graph):
                out = step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as e:      # loud, not silent: the JSON line says hip_graph false and why
            print(f'bench.py: hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches',
                  file=sys.stderr)
            graph = None
            args.graph = 0
            torch.cuda.synchronize()

    ops.KERNEL_EVENTS = {} if (graph is None and pipe is None) else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if pipe is not None:
            pipe.submit(img)
        elif graph is not None:
            graph.replay()
        else:
            out = step()
    if pipe is not None:
        pipe.flush()          # every submitted step is complete before the closing barrier + synchronize
    barrier()
    dt = time.perf_counter() - t0
    events = ops.KERNEL_EVENTS or {}
    ops.KERNEL_EVENTS = None

    tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if args.shared_devices else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- per-batch latency: ONE step alone (no cross-step overlap), submit -> results complete ----
    lat = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if pipe is not None:
            pipe.submit(img)
            pipe.flush()
        elif graph is not None:
            graph.replay()
        else:
            out = step()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    latency_ms = sorted(lat)[len(lat) // 2]

    # ---- roofline of the mask-logit kernel (full resolution) ----
    timed_how = 'HIP events around the launch inside the timed steps'
    if not events.get('mask_logits_full'):
        # graph mode: events cannot be recorded inside a replay -> the SAME K steps are run once more eagerly right
        # after the timed replays (same process, weights, inputs, shapes) with events around the launches; the
        # rocprofv3 kernel trace of the replays (profiles/) is the cross-check
        ops.KERNEL_EVENTS = events
        for _ in range(args.steps):
            out = step()
        ops.KERNEL_EVENTS = None
        timed_how = ('HIP events around the launch in the same %d steps re-run eagerly right after the timed hipGraph '
                     'replays (events cannot be recorded inside a replay)' % args.steps)
    torch.cuda.synchronize()
    # `frac` comes from the RAW event mean. An event pair with nothing between it still measures a few microseconds
    # (event-record latency on the stream); that floor is calibrated in situ and reported beside it as a secondary,
    # overhead-adjusted figure (`*_event_adjusted`), never as `frac`.
    pairs = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    ev_over = min(a.elapsed_time(b) for a, b in pairs)     # the floor: anything above it is queueing noise, not event cost
    ml = [s.elapsed_time(e) for s, e in events['mask_logits_full']]
    ml_ms = sum(ml) / len(ml)                              # raw event mean
    ml_adj_ms = max(ml_ms - ev_over, 1e-6)
    HW4 = (H // 4) * (W // 4)
    Q = args.queries
    in_bytes = 2 if args.precision == 'bf16' else 4       # packed bf16 (hi) or hi+lo = 4 B / element
    alg_bytes = B * (256 * HW4 * in_bytes + Q * 256 * 4 + Q * HW4 * 4)
    flops = 2.0 * B * Q * 256 * HW4
    gbs = alg_bytes / (ml_ms * 1e-3) / 1e9
    tfs = flops / (ml_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic('cgg_mask_logits_kernel', B, H, W)
    roofline = dict(bound='hbm', kernel='cgg_mask_logits_kernel', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=gbs / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src, launch_ms=ml_ms, launches_timed=len(ml),
                    algorithmic_bytes=alg_bytes, tflops=tfs, frac_mfma_bf16_peak=tfs / MFMA_BF16_PEAK_TF,
                    timed=timed_how, event_pair_overhead_ms=ev_over, launch_ms_event_adjusted=ml_adj_ms,
                    frac_event_adjusted=alg_bytes / (ml_adj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    rocprof=rocprof_launch_mean('cgg_mask_logits_kernel'))
    extra = {}
    if events.get('msda_fused'):
        ms = [s.elapsed_time(e) for s, e in events['msda_fused']]
        ms = sum(ms) / len(ms)                              # raw event mean
        N = sum((H // s) * (W // s) for s in (8, 16, 32))
        vb = 2 if args.precision == 'bf16' else 4         # bf16 stream: value, offsets|logits and output are bf16
        mbytes = B * N * (256 * vb + 288 * vb + 256 * vb)
        extra['msda'] = dict(launch_ms=ms, algorithmic_bytes=mbytes, achieved_GBs=mbytes / (ms * 1e-3) / 1e9,
                             frac_hbm_peak=mbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             rocprof=rocprof_launch_mean('cgg_msda_fwd_stream_kernel'))

    if events.get('encoder_tail'):
        # post-attention half of an encoder layer (output_proj + LN + FFN + LN) as one launch: MFMA-bound, and since round 2 the
        # kernel with the largest share of the step (5 launches of ~78 us + the K/V variant) -> it is the `roofline` kernel; the
        # mask-logit launch that held that place in round 1 moves to `kernels.mask_logits` unchanged.
        # flops = 2 M (256*256 + 2 * 256*1024); algorithmic bytes = attention rows + layer input rows in, y rows out (bf16)
        # (+ f32 pos rows in and `y + pos` rows out when the projection kernel does not form x + pos itself).
        ms_l = [s.elapsed_time(e) for s, e in events['encoder_tail']]
        ms = sum(ms_l) / len(ms_l)
        ms_adj = max(ms - ev_over, 1e-6)
        N = sum((H // s) * (W // s) for s in (8, 16, 32))
        M = B * N
        fl = 2.0 * M * (256 * 256 + 2 * 256 * 1024)
        from cgg_amd import pixel_decoder as _pd
        pos_in_proj = _pd.FUSED_PROJ and _pd.POS_IN_PROJ
        tb = M * 256 * 2 * 3 + (0 if pos_in_proj else M * 256 * 2 + N * 256 * 4)
        tname = 'cgg_encoder_ffn_ln_kernel<false, true>'
        ttraffic, ttraffic_src = pmc_traffic(tname, B, H, W)
        tail = dict(bound='mfma', kernel=tname, achieved=fl / (ms * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TF, unit='TFLOP/s',
                    frac=fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, traffic=ttraffic,
                    traffic_source=ttraffic_src,
                    launch_ms=ms, launches_timed=len(ms_l), flops=fl, algorithmic_bytes=tb, achieved_GBs=tb / (ms * 1e-3) / 1e9,
                    timed=timed_how, event_pair_overhead_ms=ev_over, launch_ms_event_adjusted=ms_adj,
                    frac_event_adjusted=fl / (ms_adj * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                    share_of_step='5 launches per step + 1 K/V-emitting variant: ~12 % of the kernel time of a step',
                    replaces='3 library GEMMs + 2 residual-LayerNorm passes per layer: 396 MB of HBM traffic -> 66 MB',
                    rocprof=rocprof_launch_mean(tname))
        extra['mask_logits'] = roofline
        roofline = tail
    if events.get('encoder_proj'):
        ms = [s.elapsed_time(e) for s, e in events['encoder_proj']]
        ms = sum(ms) / len(ms)
        N = sum((H // s) * (W // s) for s in (8, 16, 32))
        pb = B * N * (256 + 256 + 256 + 288) * 2
        extra['encoder_proj'] = dict(kernel='cgg_encoder_proj_kernel', bound='hbm', launch_ms=ms,
                                     launches_timed=len(events['encoder_proj']), algorithmic_bytes=pb,
                                     achieved_GBs=pb / (ms * 1e-3) / 1e9, frac_hbm_peak=pb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     rocprof=rocprof_launch_mean('cgg_encoder_proj_kernel'))

    parity = None
    pipelined = pipe is not None
    if args.precision == 'bf16' and not args.no_parity_mode:
        pipe = graph = None                  # release the captured graphs' buffers before the second mode
        parity = parity_mode_rate(args, model, img, metas, dev)
    host = None
    if args.host_results and rank == 0:
        with runtime.precision_scope(args.precision):
            host = host_results_rate(args, model, img, metas, dev)
    if rank == 0:
        res = dict(metric='images/sec (COCO-shaped 1024x1024, 100 queries, forward-only)',
                   value=B * world * args.steps / dt, unit='images/sec', n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
                   scaling='weak', vs_baseline=None,
                   dtype='bf16' if args.precision == 'bf16' else 'f32', data='synthetic',
                   config=dict(workload=f'configs[1]: R50 + {Q} queries, {H}x{W}, batch {B}/GPU, forward-only '
                                        '(backbone + MSDeformAttn pixel decoder + 9-layer masked-attention '
                                        'decoder + mask logits + instance post-processing, results on device)',
                               global_batch=B * world, parallelism=f'replicas x{world}',
                               precision=args.precision, hip_graph=bool(args.graph),
                               parity_mode_value=None if parity is None else parity['value'],
                               ranks_share_devices=bool(args.shared_devices),
                               pipeline=(f'{args.pipeline}-stage software pipeline across steps (one HIP stream + hipGraph '
                                         'per stage and buffer slot; every timed step completes inside the timed '
                                         'region)') if pipelined else 'none'),
                   latency_ms_per_batch=latency_ms, parity_mode=parity, host_results=host, roofline=roofline,
                   kernels=extra)
        if not args.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(args, cfg, model, img_cpu)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
