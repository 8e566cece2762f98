#!/usr/bin/env python
"""bench.py -- images/sec of the CGG decoder + mask-prediction hot path on MI355X.

`--workload cfg1` (default, BASELINE.json configs[1] = the metric's config): a "step" = one forward of `Mask2FormerOpen.simple_test`
(R50 backbone -> MSDeformAttn pixel decoder -> 9-layer masked-attention query decoder -> mask logits -> open-vocabulary instance
post-processing for all / novel / base class sets) over one synthetic COCO-shaped batch (1024 x 1024, batch 2, 100 queries) that is
already resident in HBM; results stay on the device. N > 1: one process per GPU (torch.distributed / RCCL only for the barrier), every
rank runs its own replica on its own batch -> weak scaling, no data-path collective (inference: "replicas only"); the N ranks then also
take configs[2]'s image-parallel training step over RCCL (`train_step.n_gpus = N`).
Other workloads: cfg2 = configs[2] (R50 training step, batch 16), cfg3 = configs[3] (Swin-B + 200 queries training step, batch 4 = one
GPU's share of the DDP batch 32), cfg4 = configs[4] (COCO-panoptic forward, 1333 x 800 padded to 800 x 1344, batch 2).

Launch: `python bench.py --gpus N` starts the N ranks ITSELF (child processes created before the parent touches the
GPU; rank r binds device r, RCCL for the barrier / max-over-ranks); under an external launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) RANK / LOCAL_RANK / WORLD_SIZE come from
the environment and `--gpus` must agree with WORLD_SIZE (it fails loudly otherwise).

Prints ONE JSON line (rank 0) as the LAST line of stdout: the compact record (`compact_record` / `emit`: < 4 KB of strict JSON -- the
contract keys, `roofline`, `cpu_baseline` and one short object per secondary measurement); the full record with every note goes to
gpurun_out/bench_full.json (`--full-line` prints it instead: what a parent bench.py run parses from its children). `value` is PARITY
mode (`--precision fp32`, the default): every contraction of the path in f32-class arithmetic on the f16 matrix cores (csrc/x3.h), the
mode the 1e-3 / bit-exact parity tests run in. Objects of the full record:
  roofline      -- the dominant kernel family of that step, the x3 GEMM / implicit-GEMM convolution family (cgg_gemm_x3s_kernel: LDS-DMA
                   GEMM over pre-split x3a rows), from HIP events on the launch stream around every launch of the K steps re-run eagerly
                   after the timed region; kernels.* hold the encoder layer tail, the mask-logit einsum and MSDeformAttn the same way;
                   `traffic` / `rocprof` fields come from the committed rocprofv3 runs of this command (labelled as such).
  einsum_mfma_target -- BASELINE.json's kernel target: the mask-logit einsum at (B, Q) = (2, 100), (2, 200), (4, 200), (16, 100) (stored-
                   logits and consumer-fused forms, incl. the query-stationary / LDS-ring kernel), 20 launches in one hipGraph, % of MFMA peak.
  bf16_mode     -- the same step in throughput mode (bf16 MFMA): secondary, never `value`; with the bf16 agreement record of this config.
  train_step    -- configs[2]'s training step, run in child processes: the parity-mode (f32-class) step is the object itself, the
                   bf16 autocast step its `bf16_mode` member; each with `roofline` / `kernels` from live events.
  extra         -- configs[3] (training step, both precisions) and configs[4] (panoptic forward, parity + bf16) measured the same way
                   in child processes (`python bench.py --workload cfg3|cfg4`).
  cpu_baseline  -- the oracle (torch CPU restatement of the reference path, kind "port") on one image of the
                   same workload on this box's host cores.
"""
import argparse
import json
import os
import re
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
MFMA_BF16_PEAK_TF = 2500.0          # dense bf16 / f16 MFMA peak
F32_MFMA_PEAK_TF = 157.3            # f32-input MFMA = f32 vector peak
X3_PEAK_TF = MFMA_BF16_PEAK_TF / 3  # f32-class f16 x 3 arithmetic: three f16 MFMAs per product


# BASELINE.json configs[1..4] (configs[0] is the reference's CPU-only plumbing case: `cpu_baseline.cfg1_512`). `--workload` picks
# one; the default line is configs[1] (the metric's config) with the others attached as `train_step` (configs[2]) and `extra`
# (configs[3], configs[4]) objects measured in child processes of the same run.
WORKLOADS = {
    'cfg1': dict(mode='infer', backbone='r50', queries=100, hw=(1024, 1024), batch=2, panoptic=False,
                 name='configs[1]: R50 + 100 queries, 1024x1024, batch 2/GPU, forward-only'),
    'cfg2': dict(mode='train', backbone='r50', queries=100, hw=(1024, 1024), batch=16, panoptic=False,
                 name='configs[2]: R50 + 100 queries, COCO-instance training step, batch 16/GPU'),
    'cfg3': dict(mode='train', backbone='swin_b', queries=200, hw=(1024, 1024), batch=4, panoptic=False,
                 name='configs[3]: Swin-B + 200 queries, 1024x1024, training step, batch 4/GPU (DDP batch 32 on 8 GPUs)'),
    'cfg4': dict(mode='infer', backbone='r50', queries=100, hw=(800, 1344), batch=2, panoptic=True,
                 name='configs[4]: COCO-panoptic (80 things + 53 stuff), 1333x800 padded to 800x1344, batch 2/GPU, forward-only'),
}


LINE_LIMIT = 4096       # bytes of the final stdout line (the driver keeps an 8 KB tail; round 5's 36 KB line came back `parsed: null`)


def _r(x, n=4):
    """float -> n significant digits (the compact line carries numbers, not prose; the full record keeps full precision)"""
    if isinstance(x, float):
        if x != x or x in (float('inf'), float('-inf')):
            return None
        return float(f'{x:.{n}g}')
    return x


def _finite(o):
    """strict-JSON form of a record: non-finite floats -> None (json.dumps would print NaN / Infinity, which is not JSON)"""
    if isinstance(o, float):
        return o if (o == o and abs(o) != float('inf')) else None
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def _pick(d, *keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _roof_short(r):
    """name-only kernel + the roofline numbers of one kernel / family object of the full record"""
    if not isinstance(r, dict):
        return None
    o = {}
    if r.get('kernel'):
        o['kernel'] = re.split(r'[ <(]', r['kernel'], 1)[0]
    o['bound'] = r.get('bound')
    ach = r.get('achieved', r.get('achieved_GBs') if r.get('bound') == 'hbm' else None)
    peak = r.get('peak', HBM_PEAK_GBS if r.get('bound') == 'hbm' else None)
    frac = r.get('frac', r.get('frac_hbm_peak'))
    o.update(achieved=_r(ach), peak=_r(peak), unit=r.get('unit', 'GB/s' if r.get('bound') == 'hbm' else 'TFLOP/s'), frac=_r(frac),
             traffic=_r(r.get('traffic', r.get('traffic_bytes_per_step'))))
    for src, dst in (('algorithmic_bytes_per_step', 'algorithmic_bytes_per_step'), ('algorithmic_bytes', 'algorithmic_bytes'),
                     ('flops_per_step', 'flops_per_step'), ('flops', 'flops'), ('ms_per_step', 'ms_per_step'), ('launch_ms', 'launch_ms'),
                     ('launches_per_step', 'launches_per_step'), ('share_of_step', 'share_of_step')):
        if r.get(src) is not None:
            o[dst] = _r(r[src])
    return o


def _train_short(t):
    if not isinstance(t, dict):
        return None
    if 'error' in t:
        return dict(error=str(t['error'])[:120])
    o = _pick(t, 'value', 'ms_per_step', 'n_gpus', 'loss', 'launches_per_step', 'hand_written_share_of_kernel_time')
    o['dtype'] = str(t.get('dtype', '')).split(' ')[0]
    if isinstance(t.get('roofline'), dict):
        rr = _roof_short(t['roofline'])
        o['roofline'] = {k: rr[k] for k in ('kernel', 'bound', 'frac', 'ms_per_step', 'share_of_step') if rr.get(k) is not None}
    if isinstance(t.get('bf16_mode'), dict):
        o['bf16'] = _pick(t['bf16_mode'], 'value', 'ms_per_step', 'error')
    return o


def compact_record(res):
    """The driver-readable form of a full bench record: the contract keys + `roofline` + `cpu_baseline` + one short object per
    secondary measurement, numbers only (4 significant digits), every string short. `emit` asserts it stays under LINE_LIMIT."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data')
    out = {k: res.get(k) for k in keep}
    out['dtype'] = str(out['dtype']).split(' ')[0] if out.get('dtype') else out.get('dtype')
    cfg = res.get('config') or {}
    wl = str(cfg.get('workload', ''))
    c = dict(workload=wl.split(' (')[0][:120], global_batch=cfg.get('global_batch'), parallelism=cfg.get('parallelism'))
    if cfg.get('precision'):
        c['precision'] = str(cfg['precision']).split(' ')[0] + (' (parity mode, f16x3 MFMA)' if str(cfg['precision']).startswith('fp32') else '')
    for k in ('hip_graph', 'x3_overflow', 'ranks_share_devices', 'trainable_params', 'grad_buckets'):
        if cfg.get(k) is not None:
            c[k] = cfg[k]
    if isinstance(cfg.get('library_fallbacks'), dict):
        c['library_fallbacks'] = cfg['library_fallbacks'].get('count')
    ag = cfg.get('parity_mode_agreement_without_injection')
    if isinstance(ag, dict) and isinstance(ag.get('record'), dict):
        c['no_injection'] = _pick(ag['record'], 'final_mask_logit_err', 'attn_mask_bit_agreement_min', 'topk_pairs_swapped_max',
                                  'det_score_abs_err_max')
    out['config'] = c
    if res.get('latency_ms_per_batch') is not None:
        out['latency_ms_per_batch'] = _r(res['latency_ms_per_batch'])
    for k in ('value', 'ms_per_step'):
        out[k] = _r(out[k], 6) if out.get(k) is not None else None
    out['roofline'] = _roof_short(res.get('roofline'))
    ks = {}
    for name, r in (res.get('kernels') or {}).items():
        rr = _roof_short(r)
        if rr:
            ks[name] = {k: rr[k] for k in ('bound', 'frac', 'traffic', 'launch_ms', 'ms_per_step', 'algorithmic_bytes', 'flops')
                        if rr.get(k) is not None}
    if ks:
        out['kernels'] = ks
    sw = res.get('einsum_mfma_target')
    if sw:
        rows = [r for r in sw if 'queries' in r]
        if rows:
            best = max(rows, key=lambda r: r['frac_mfma_peak'] * (r['mfma_peak_tf'] / MFMA_BF16_PEAK_TF))
            out['einsum'] = dict(
                target='>=0.40 of 2500 TF bf16 MFMA', kernel='bqc,bchw->bqhw',
                best=dict(B=best.get('batch'), Q=best['queries'], form=best.get('form', best['mode'][:24]),
                          frac_bf16_mfma_peak=_r(best['tflops'] / MFMA_BF16_PEAK_TF), tflops=_r(best['tflops']), launch_ms=_r(best['launch_ms'])),
                rows=[[r.get('batch'), r['queries'], r.get('form', r['mode'][:24]), _r(r['tflops'] / MFMA_BF16_PEAK_TF, 3),
                       _r(r['frac_hbm_peak'], 3)] for r in rows],
                row_keys=['B', 'Q', 'form', 'frac_bf16_mfma_peak', 'frac_hbm_peak'])
            cp = [r for r in sw if 'reference' in r]
            if cp:
                out['einsum']['copy_frac_hbm_peak'] = _r(cp[0]['frac_hbm_peak'], 3)
    b = res.get('bf16_mode')
    if isinstance(b, dict):
        o = _pick(b, 'value', 'ms_per_step')
        rec = (b.get('agreement_with_f32_oracle') or {}).get('record')
        if isinstance(rec, dict):
            if rec.get('attn_mask_bit_agreement_per_layer'):
                o['attn_bit_agreement_min'] = min(rec['attn_mask_bit_agreement_per_layer'])
            o.update(_pick(rec, 'mask_iou_mean', 'topk_pair_jaccard_mean', 'panoptic_pixel_agreement'))
        out['bf16_mode'] = o
    h = res.get('host_results')
    if isinstance(h, dict):
        out['host_results'] = dict(_pick(h, 'value', 'rle_bytes_per_mask'),
                                   trained_like_masks=_r((h.get('trained_like_masks') or {}).get('value')))
    cb = res.get('cpu_baseline')
    if isinstance(cb, dict):
        o = _pick(cb, 'value', 'unit', 'cores', 'threads', 'kind', 'host_logical_cpus')
        o['sample'] = str(cb.get('sample', '')).split(' (')[0][:100]
        if isinstance(cb.get('cfg1_512'), dict):
            o['configs0_512x512'] = _r(cb['cfg1_512'].get('value'))
        out['cpu_baseline'] = o
    if res.get('train_step') is not None:
        out['train_step'] = _train_short(res['train_step'])
    ex = res.get('extra')
    if isinstance(ex, dict):
        e = {}
        c3, c4 = ex.get('configs[3]'), ex.get('configs[4]')
        if c3 is not None:
            e['cfg3'] = _train_short(c3)
        if isinstance(c4, dict):
            if 'error' in c4:
                e['cfg4'] = dict(error=str(c4['error'])[:120])
            else:
                o = _pick(c4, 'value', 'ms_per_step')
                if isinstance(c4.get('roofline'), dict):
                    o['roofline_frac'] = _r(c4['roofline'].get('frac'))
                if isinstance(c4.get('bf16_mode'), dict):
                    o['bf16'] = _r(c4['bf16_mode'].get('value'))
                e['cfg4'] = o
        out['extra'] = e
    for k in ('loss', 'peak_mem_gb'):
        if res.get(k) is not None:
            out[k] = _r(res[k])
    return out


def emit(res, full_path=None, stream=None):
    """Write the full record to `full_path` (default gpurun_out/bench_full.json; also echoed on stderr) and print the compact line --
    ONE line of strict JSON under LINE_LIMIT bytes -- as the last line of stdout. If the compact record ever outgrows the
    limit, the optional objects are dropped largest first (never `roofline` / `cpu_baseline`) and the line says which."""
    stream = stream or sys.stdout
    full_path = full_path or os.path.join(ROOT, 'gpurun_out', 'bench_full.json')
    try:
        os.makedirs(os.path.dirname(full_path), exist_ok=True)
        with open(full_path, 'w') as f:
            json.dump(_finite(res), f, indent=1, allow_nan=False)
    except OSError as e:
        print(f'bench.py: could not write {full_path}: {e}', file=sys.stderr)
        full_path = None
    rec = _finite(compact_record(res))
    if full_path:
        rec['full_record'] = os.path.relpath(full_path, ROOT)
    dropped = []
    line = json.dumps(rec, allow_nan=False, separators=(',', ':'))
    optional = ['einsum', 'kernels', 'extra', 'host_results', 'bf16_mode', 'train_step']
    while len(line) >= LINE_LIMIT and optional:
        k = max(optional, key=lambda k: len(json.dumps(rec.get(k))) if k in rec else -1)
        optional.remove(k)
        if k in rec:
            del rec[k]
            dropped.append(k)
            rec['dropped_for_length'] = dropped
        line = json.dumps(rec, allow_nan=False, separators=(',', ':'))
    assert len(line) < LINE_LIMIT, f'bench line is {len(line)} bytes'
    assert '\n' not in line
    print(line, file=stream, flush=True)
    return line


def workload_config(args):
    """model dict of the selected BASELINE config (reference files: configs/instance/coco_b48n17.py, configs/openset_panoptic/
    coco_panoptic_p20.py; Swin-B = the Mask2Former Swin-B backbone settings)."""
    from cgg_amd import synthetic
    if args.panoptic:
        cfg = synthetic.model_config(panoptic=True, num_things=80, num_stuff=53, num_unknown=16, num_queries=args.queries, depth=50)
    else:
        cfg = synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=args.queries, depth=50)
    if args.backbone == 'swin_b':
        cfg['backbone'] = dict(type='SwinTransformer', embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12,
                               mlp_ratio=4, out_indices=(0, 1, 2, 3), drop_path_rate=0.3, patch_norm=True)
        cfg['panoptic_head']['in_channels'] = [128, 256, 512, 1024]
    return cfg


def workload_metas(args, B):
    from cgg_amd import synthetic
    H, W = args.hw
    if args.panoptic:
        # 1333 x 800 keep-ratio resize + Pad(size_divisor=32) (coco_panoptic_p20.py:221-226): image 800 x 1333 inside an 800 x 1344
        # batch, results rescaled to a COCO-sized 480 x 800 original
        return [dict(img_shape=(H, 1333, 3), ori_shape=(480, 800, 3), pad_shape=(H, W, 3), batch_input_shape=(H, W), scale_factor=1.0,
                     flip=False) for _ in range(B)]
    return synthetic.img_metas(B, H, W)


def build_model(args, dev):
    import cgg_amd
    from cgg_amd import registry, synthetic
    cfg = workload_config(args)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        # seeded BEFORE construction: parameters that `init_weights` does not touch keep their constructor values, which came from
        # the unseeded global generator until round 4 -- the masks (and with them the per-run cost of the host RLE path: 0.4 to
        # 80 KB of RLE per mask) differed from process to process
        torch.manual_seed(0)
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
    quota = runtime.effective_cpu_count()       # what the container may really burn (cgroup CPU quota: 16 of the 256 logical CPUs on the GPU boxes)

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
    small = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(4321))
    # intra-op thread count: torch's CPU kernels stop scaling long before 256 cores (more threads = more fork/join and NUMA
    # traffic per op); a short sweep on the 512 x 512 case picks the count that is then used for both timed cases
    sweep = {}
    for t in sorted({max(1, min(cores, c)) for c in (quota // 2, quota, 2 * quota, 4 * quota)}):
        torch.set_num_threads(t)
        one_pass(small)
        sweep[t] = one_pass(small)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    dt, ts = timed(full)
    dt1, ts1 = timed(small)
    return dict(value=1.0 / dt, unit='images/sec', cores=threads, threads=threads, host_logical_cpus=cores, cgroup_cpu_quota=quota, kind='port',
                thread_sweep_512_s={str(k): round(v, 3) for k, v in sweep.items()},
                sample=f'1 image {H}x{W} of the benched workload, full detector forward + instance post-processing '
                       f'(torch CPU oracle, fp32, {threads} threads -- the fastest of the swept counts; the container\'s cgroup grants '
                       f'{quota} of the host\'s {cores} logical CPUs): '
                       f'1 warm-up + median of 3 timed passes ({", ".join("%.2f" % t for t in ts)} s)',
                cfg1_512=dict(value=1.0 / dt1, unit='images/sec',
                              sample='configs[0]: one 512x512 image, same pipeline, 1 warm-up + median of 3 '
                                     f'({", ".join("%.2f" % t for t in ts1)} s)'))


def committed_profile(name):
    """A summary committed under profiles/ by scratch/collect_profiles_r5.sh + publish_profiles_r5.py (rocprofv3 runs of THIS command on an MI355X;
    counters cannot be collected from inside the process). Every field taken from it is labelled `source: committed profile`."""
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except Exception:
        return None


AGREEMENT_KEYS = ('configs1_fp32_no_injection', 'configs1_bf16', 'configs3_bf16', 'configs4_bf16', 'configs2_train_slice',
                  'configs3_train_slice')


def agreement_records():
    """(records, path) of the newest committed agreement profile (tools/collect_agreement.sh publish)."""
    for name in ('r6_agreement.json', 'r5_agreement.json'):
        d = committed_profile(name)
        if d:
            return d, 'profiles/' + name
    return {}, 'profiles/r6_agreement.json'


def host_results_rate(args, model, img, metas, dev):
    """images/sec of the same step when the results must reach the HOST in the evaluation format: the
    staged pipeline (same stage split as the headline run) with bit-packed masks, ONE asynchronous device->host copy per result tensor into pinned staging
    buffers, COCO RLE on the extension's host threads overlapped with the next batch (host_results.RleCollector).
    PCIe-inclusive; never `value`. Also reports the mean number of runs per mask (encoder cost is per run: the
    random-weight masks of this benchmark are far noisier than a trained model's)."""
    from cgg_amd import runtime
    from cgg_amd.host_results import RleCollector, fusion_class_counts
    from cgg_amd.pipeline import detector_pipeline
    nstage = min(max(args.pipeline, 2), 3)
    pipe = detector_pipeline(model, img, metas, stages=nstage, defer_tail=args.defer_tail, rescale=True,
                             device_results=True, mask_bits=True)
    col = RleCollector(dev, fusion_class_counts(model.panoptic_fusion_head))
    steps = max(args.steps, 8)
    disks = {}

    def with_disks(results):
        # the same result tensors with every mask replaced by a filled disk (bit-packed, device-resident, made once): the run count
        # of a trained model's masks (~0.5-2 KB of RLE) instead of the random-weight model's noise (~100 KB) -- the encoder's cost is
        # per run. The GPU pipeline, the copies and the host path are unchanged.
        out = []
        for res in results:
            per = {}
            for key, val in res.items():
                if isinstance(val, (tuple, list)) and len(val) == 3 and torch.is_tensor(val[2]) and val[2].dtype == torch.uint8:
                    m = val[2]
                    k = (tuple(m.shape), m.device)
                    if k not in disks:
                        n, Hh, Wb = m.shape
                        yy = torch.arange(Hh, device=m.device).view(1, Hh, 1).float()
                        xx = torch.arange(Wb * 8, device=m.device).view(1, 1, Wb * 8).float()
                        g = torch.Generator(device='cpu').manual_seed(11)
                        cy = (torch.rand(n, 1, 1, generator=g) * Hh).to(m.device)
                        cx = (torch.rand(n, 1, 1, generator=g) * Wb * 8).to(m.device)
                        rr = (torch.rand(n, 1, 1, generator=g) * 0.25 + 0.05).to(m.device) * Hh
                        on = ((yy - cy) ** 2 + (xx - cx) ** 2) <= rr ** 2
                        w = (on.view(n, Hh, Wb, 8).to(torch.uint8) << torch.arange(8, device=m.device, dtype=torch.uint8)).sum(-1)
                        disks[k] = w.to(torch.uint8).contiguous()
                    per[key] = (val[0], val[1], disks[k])
                else:
                    per[key] = val
            out.append(per)
        return out

    def run(n, smooth=False):
        # `depth` batches stay in flight behind the one being submitted; a slot's result buffers are overwritten by the LAST
        # stage of the batch that reuses it, which waits (on the GPU) for the slot's previous device->host copies.
        # Finished batches are reduced to counters at once (a serving loop hands them on; holding 40 batches x 600 RLE dicts
        # alive made the cyclic garbage collector stall the submitting thread for tens of milliseconds now and then)
        depth = nstage - 1
        futs, in_copy, pending, copied = [], [], [], {}
        tally = dict(images=0, masks=0, nbytes=0)

        def drain(block):
            while futs and (block or futs[0].done()):
                for im in futs.pop(0).result():
                    tally['images'] += 1
                    for key in im:
                        for cls in im[key][1]:
                            tally['masks'] += len(cls)
                            tally['nbytes'] += sum(len(r['counts']) for r in cls)
        for _ in range(n):
            while len(in_copy) > depth + 1:
                RleCollector.wait_copied(in_copy.pop(0))
            ev = copied.pop(pipe._n % pipe.slots, None)
            if ev is not None:
                pipe.streams[-1].wait_event(ev)
            pending.append(pipe.submit(img))
            while len(pending) > depth:
                old = pending.pop(0)
                r = pipe.wait(old)
                f = col.submit(with_disks(r) if smooth else r)
                futs.append(f)
                in_copy.append(f)
                copied[old] = f.copied
            drain(False)
        for old in pending:
            r = pipe.wait(old)
            futs.append(col.submit(with_disks(r) if smooth else r))
        drain(True)
        return tally
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    run(3, smooth=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res_s = run(steps, smooth=True)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    col.close()
    return dict(value=res['images'] / dt, unit='images/sec (this rank, results on the host as COCO RLE)', steps=steps,
                masks_per_image=res['masks'] / max(res['images'], 1), rle_bytes_per_mask=res['nbytes'] / max(res['masks'], 1),
                note='the encoder\'s cost is per run: the random-weight model\'s masks are noise (rle_bytes_per_mask; a trained model: '
                     '0.5-2 KB). `trained_like_masks` = the same pipeline, copies and host path with every mask replaced by a filled disk',
                trained_like_masks=dict(value=res_s['images'] / dt_s, unit='images/sec',
                                        rle_bytes_per_mask=res_s['nbytes'] / max(res_s['masks'], 1)),
                rle_threads=col.rle_threads, cpu_quota=runtime.effective_cpu_count(),
                how=f'{nstage}-stage pipeline, bit-packed masks, pinned async D2H, C++ RLE on a persistent pool of host threads overlapped with the next batch')


def time_mode(args, model, img, metas, dev, precision, barrier, collect_events):
    """images/sec of the step in one precision mode: the staged pipeline (one HIP stream + hipGraph per stage) or one graph
    per step / eager launches; the K-step region is repeated `args.repeats` times back to back (each bracketed by the barrier)
    and the MEDIAN region is reported, so that the timed span is >= 0.5 s while `steps` stays what the driver passed.
    Then: per-batch latency (one step alone) and, if `collect_events`, the same K steps once more eagerly with HIP events
    around the hot launches on the launch stream (events cannot be recorded inside a graph replay)."""
    import statistics
    from cgg_amd import ops, runtime
    out = {}
    with runtime.precision_scope(precision):
        def step():
            with torch.no_grad():
                return model.simple_test(img, metas, rescale=True, device_results=True)
        for _ in range(max(args.warmup, 1)):
            step()
        torch.cuda.synchronize()
        graph = pipe = None
        how = 'eager launches'
        if args.graph and args.pipeline:
            try:
                from cgg_amd.pipeline import detector_pipeline
                pipe = detector_pipeline(model, img, metas, stages=args.pipeline, defer_tail=args.defer_tail, rescale=True,
                                         device_results=True)
                for _ in range(len(pipe.stages)):
                    pipe.submit(img)
                pipe.flush()
                torch.cuda.synchronize()
                how = (f'{len(pipe.stages)}-stage software pipeline across steps (one HIP stream + hipGraph per stage and buffer '
                       'slot; every timed step completes inside the timed region)')
            except Exception as e:
                print(f'bench.py: stage pipeline setup failed in {precision} mode ({type(e).__name__}: {e}); one graph per step',
                      file=sys.stderr)
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
            except Exception as e:      # loud, not silent: the JSON line says so
                print(f'bench.py: hipGraph capture failed in {precision} mode ({type(e).__name__}: {e}); eager', file=sys.stderr)
                graph = None
                torch.cuda.synchronize()

        def region():
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                if pipe is not None:
                    pipe.submit(img)
                elif graph is not None:
                    graph.replay()
                else:
                    step()
            if pipe is not None:
                pipe.flush()          # every submitted step is complete before the closing barrier + synchronize
            barrier()
            return time.perf_counter() - t0
        regions = [region() for _ in range(max(args.repeats, 1))]
        out['regions_s'] = regions
        out['dt'] = statistics.median(regions)
        out['how'] = how
        out['hip_graph'] = graph is not None or pipe is not None
        out['pipelined'] = pipe is not None
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
                step()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        out['latency_ms'] = sorted(lat)[len(lat) // 2]
        pipe = graph = None
        if collect_events:
            ops.KERNEL_EVENTS, ops.KERNEL_META = {}, {}
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            out['events'], out['meta'] = ops.KERNEL_EVENTS, ops.KERNEL_META
            ops.KERNEL_EVENTS = ops.KERNEL_META = None
    return out


def emit_result(args, res):
    """rank 0's output: a child of another bench.py run (`--full-line`) hands its parent the full record as one stdout line; every
    other run prints the compact line (and leaves the full record in gpurun_out/)."""
    if args.full_line:
        print(json.dumps(_finite(res), allow_nan=False), flush=True)
        return
    name = 'bench_full.json' if args.workload == 'cfg1' else f'bench_full_{args.workload}.json'
    emit(res, full_path=os.path.join(ROOT, 'gpurun_out', name))


def train_main(args, cfg, model, img, metas, dev, rank, world):
    res = train_run(args, cfg, model, img, metas, dev, rank, world)
    if rank == 0:
        emit_result(args, res)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def train_run(args, cfg, model, img, metas, dev, rank, world, steps=None, warmup=None):
    """configs[2] / configs[3]: one optimisation step = forward_train (all 10 layers' losses incl. grounding + caption
    generation) -> backward with bucketed gradient all-reduce over RCCL overlapped -> clip -> AdamW step. Every rank calls this
    (collectives inside); rank 0 gets the result line as a dict."""
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    import torch.distributed as dist
    from cgg_amd import synthetic
    from cgg_amd.train import GradReducer, build_optimizer, train_step
    B, (H, W) = args.batch, args.hw
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

    for _ in range(max(warmup, 1)):
        logs = train_step(model, optimizer, reducer, data, clip)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        logs = train_step(model, optimizer, reducer, data, clip)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if args.shared_devices else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # ---- roofline of the step's hand-written kernel families: ONE more step, eagerly, with HIP events around the launches on the
    #      launch stream (the timed region above carries no events) ----
    roofline, kernels = None, {}
    if rank == 0:
        from cgg_amd import ops
        ops.KERNEL_EVENTS, ops.KERNEL_META = {}, {}
        train_step(model, optimizer, reducer, data, clip)
        torch.cuda.synchronize()
        ev, meta = ops.KERNEL_EVENTS, ops.KERNEL_META
        ops.KERNEL_EVENTS = ops.KERNEL_META = None
        roofline, kernels = train_kernel_summaries(ev, meta, dt / steps * 1e3, args.precision)
    res = None
    if rank == 0:
        nparam = sum(p.numel() for p in model.parameters() if p.requires_grad)
        res = (dict(
            metric=f'images/sec (training step, {H}x{W}, {args.queries} queries)',
            value=B * world * steps / dt, unit='images/sec', n_gpus=world, steps=steps,
            warmup=warmup, ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling='weak',
            vs_baseline=None, dtype='bf16 (bf16 GEMMs / MFMA operands, f32 accumulate; the FPN level and MSDeformAttn in f32-class / f32)' if args.precision == 'bf16' else 'f32 (encoder layers, the FPN level, caption-transformer / Swin linears and the frozen backbone stages on the f32-class f16x3 kernels: forward, grad-input, grad-weight; the rest f32 library GEMMs / convolutions under autograd)', data='synthetic',
            config=dict(workload=f'{WORKLOADS[args.workload]["name"]} '
                                 '(forward_train with grounding + caption-generation losses, backward, gradient '
                                 'all-reduce, clip, AdamW)',
                        global_batch=B * world, parallelism=f'dp{world}', precision=args.precision,
                        trainable_params=nparam, grad_buckets=len(reducer.buckets), bucket_mb=args.bucket_mb),
            loss=logs.get('loss'), peak_mem_gb=torch.cuda.max_memory_allocated() / 2**30, roofline=roofline, kernels=kernels))
    return res


def train_kernel_summaries(events, meta, step_ms, precision):
    """Roofline objects of a training step's hand-written kernel families from raw HIP-event means of ONE eagerly re-run step:
    the f32-class x3 GEMM family (forward + grad-input of the encoder linears, the x3 convolutions), the x3 weight-gradient
    kernel, the MSDeformAttn backward (HBM-bound: value + locations + weights + grad_output in, three gradients out). `roofline` =
    the family with the largest share of the step."""
    out = {}

    def fam(name):
        ev, mt = events.get(name, []), meta.get(name, [])
        if not ev or len(mt) != len(ev):
            return None
        ms = [s.elapsed_time(e) for s, e in ev]
        return ms, mt
    for name, kernel, peak_note in (('gemm_x3', 'cgg_gemm_x3_kernel / cgg_gemm_x3s_kernel (f32-class f16 x 3 GEMM / implicit-GEMM convolution: '
                                                'forward and grad-input of the encoder linears, FPN 3x3, frozen backbone stages)', None),
                                    ('wgrad_x3', 'cgg_wgrad_x3_kernel (dW = dy^T x on the f16 x 3 contraction, transpose reads)', None)):
        f = fam(name)
        if f is None:
            continue
        ms, mt = f
        tot = sum(ms)
        fl = sum(m['flops'] for m in mt)
        by = sum(m['bytes'] for m in mt)
        tf = fl / (tot * 1e-3) / 1e12
        out[name] = dict(bound='mfma', kernel=kernel, achieved=tf, peak=X3_PEAK_TF, unit='TFLOP/s', frac=tf / X3_PEAK_TF,
                         launches_per_step=len(ms), ms_per_step=tot, share_of_step=tot / step_ms, flops_per_step=fl,
                         algorithmic_bytes_per_step=by, achieved_GBs=by / (tot * 1e-3) / 1e9,
                         timed='HIP events around every launch of one step re-run eagerly after the timed region; raw means')
    f = fam('msda_backward')
    if f is not None:
        ms, mt = f
        tot = sum(ms)
        by = sum(m['bytes'] for m in mt)
        gbs = by / (tot * 1e-3) / 1e9
        out['msda_backward'] = dict(bound='hbm', kernel='cgg_msda_backward (gather kernel for grad_loc / grad_attn + destination-tiled '
                                                       'scatter for grad_value)', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
                                    frac=gbs / HBM_PEAK_GBS, launches_per_step=len(ms), ms_per_step=tot, launch_ms=tot / len(ms),
                                    share_of_step=tot / step_ms, algorithmic_bytes_per_step=by, traffic=None,
                                    timed='HIP events around every call of one step re-run eagerly after the timed region')
    if not out:
        return None, {}
    top = max(out, key=lambda k: out[k]['share_of_step'])
    roof = out.pop(top)
    roof['family'] = top
    roof.setdefault('traffic', None)
    return roof, out


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


def kernel_summaries(args, events, meta, B, H, W, Q):
    """Roofline objects of the parity-mode step's kernels from the raw HIP-event means of the launches inside the K re-run steps
    (`frac` is never adjusted for the event-pair floor; it is reported beside it)."""
    pairs = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    ev_over = min(a.elapsed_time(b) for a, b in pairs)     # the floor of an empty event pair on this stream
    timed = ('HIP events around the launches in the same %d steps re-run eagerly right after the timed region (events cannot be '
             'recorded inside a hipGraph replay); raw means' % args.steps)
    out = {}
    pname = 'cfg4_kernels.json' if getattr(args, 'workload', 'cfg1') == 'cfg4' else 'fp32_kernels.json'
    prof = committed_profile('r6_' + pname) or committed_profile('r5_' + pname) or {}

    def ms_list(name):
        return [s.elapsed_time(e) for s, e in events.get(name, [])]

    # ---- the x3 GEMM / implicit-GEMM convolution family: the dominant kernel of the step ----
    g = ms_list('gemm_x3')
    gm = meta.get('gemm_x3', [])
    if g and len(gm) == len(g):
        per_step = len(g) // args.steps
        ms_step = sum(g) / args.steps
        fl_step = sum(m['flops'] for m in gm) / args.steps
        by_step = sum(m['bytes'] for m in gm) / args.steps
        tf = fl_step / (ms_step * 1e-3) / 1e12
        worst = {}
        for t, m in zip(g, gm):
            worst.setdefault(m['shape'], []).append(t)
        shapes = sorted(((sum(v) / len(v), len(v) // args.steps, k) for k, v in worst.items()), reverse=True)
        if os.environ.get('CGG_BENCH_ALL_SHAPES'):     # measurement aid: the whole per-shape table on stderr
            for t, n, k in shapes:
                print('gemm shape %s x%d: %.1f us' % (k, n, t * 1e3), file=sys.stderr)
        shapes = shapes[:6]
        pr = prof.get('cgg_gemm_x3s_kernel', prof.get('cgg_gemm_x3_kernel', {}))
        out['gemm_x3'] = dict(
            bound='mfma', kernel='cgg_gemm_x3s_kernel<CONV, TM, TN, WM, WN, KG, SA, SB> (LDS-DMA GEMM / implicit-GEMM convolution over '
                                 'pre-split x3a rows; all instantiations: %d launches per step -- the BN-folded ResNet convolutions, the '
                                 'pixel decoder\'s 1x1 / 3x3 convolutions, value / offset / merged K-V projections; CGG_X3A=0: round 3\'s '
                                 'cgg_gemm_x3_kernel)' % per_step,
            achieved=tf, peak=X3_PEAK_TF, unit='TFLOP/s', frac=tf / X3_PEAK_TF,
            peak_note='f32-class f16 x 3 arithmetic issues 3 v_mfma_f32_32x32x16_f16 per product: peak = dense f16 MFMA peak '
                      '2500 TF / 3; achieved = ALGORITHMIC flops (2 M N K) / time',
            frac_of_f32_mfma_peak=tf / F32_MFMA_PEAK_TF, mfma_issue_frac_of_f16_peak=3 * tf / MFMA_BF16_PEAK_TF,
            traffic=pr.get('traffic_bytes_per_step'), traffic_source=pr.get('source'),
            launch_ms=ms_step / per_step, launches_timed=len(g), launches_per_step=per_step, ms_per_step=ms_step,
            flops_per_step=fl_step, algorithmic_bytes_per_step=by_step, achieved_GBs=by_step / (ms_step * 1e-3) / 1e9,
            slowest_shapes=[dict(M_N_K=list(k), launches_per_step=n, launch_ms=t) for t, n, k in shapes],
            timed=timed, event_pair_overhead_ms=ev_over, rocprof=pr.get('rocprof'))
    t = ms_list('encoder_tail_x3')
    tm = meta.get('encoder_tail_x3', [])
    if t and tm:
        ms = sum(t) / len(t)
        fl, by = tm[0]['flops'], tm[0]['bytes']
        tf = fl / (ms * 1e-3) / 1e12
        pr = prof.get('cgg_encoder_tail_x3_kernel', {})
        out['encoder_tail_x3'] = dict(
            bound='mfma', kernel='cgg_encoder_tail_x3_kernel', achieved=tf, peak=X3_PEAK_TF, unit='TFLOP/s', frac=tf / X3_PEAK_TF,
            frac_of_f32_mfma_peak=tf / F32_MFMA_PEAK_TF, traffic=pr.get('traffic_bytes'), traffic_source=pr.get('source'),
            launch_ms=ms, launches_timed=len(t), flops=fl, algorithmic_bytes=by, achieved_GBs=by / (ms * 1e-3) / 1e9,
            launch_ms_event_adjusted=max(ms - ev_over, 1e-6), timed=timed, rocprof=pr.get('rocprof'),
            replaces='3 x3 GEMM launches + 2 residual-LayerNorm passes per encoder layer (0.6 GB of f32 intermediates per layer)')
    ml = ms_list('mask_logits_full')
    if ml:
        ms = sum(ml) / len(ml)
        HW4 = (H // 4) * (W // 4)
        fl = 2.0 * B * Q * 256 * HW4
        by = B * (256 * HW4 * 4 + Q * 256 * 4 + Q * HW4 * 4)          # hi + lo pieces = 4 B / element
        pr = prof.get('cgg_mask_logits_kernel', {})
        out['mask_logits'] = dict(
            bound='hbm', kernel='cgg_mask_logits_kernel<true> (einsum bqc,bchw->bqhw, f16 x 3 split operands, f32 logits out)',
            achieved=by / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            traffic=pr.get('traffic_bytes'), traffic_source=pr.get('source'), launch_ms=ms, launches_timed=len(ml),
            algorithmic_bytes=by, flops=fl, tflops=fl / (ms * 1e-3) / 1e12, queries=Q,
            frac_x3_mfma_peak=fl / (ms * 1e-3) / 1e12 / X3_PEAK_TF, timed=timed, rocprof=pr.get('rocprof'))
    if events.get('msda_fused'):
        ms = ms_list('msda_fused')
        ms = sum(ms) / len(ms)
        N = sum((H // s) * (W // s) for s in (8, 16, 32))
        mbytes = B * N * (256 + 288 + 256) * 4
        pr = prof.get('cgg_msda_fwd_stream2_f32_kernel', {})
        out['msda'] = dict(kernel='cgg_msda_fwd_stream2_f32_kernel (f32 values / offsets / output)', bound='hbm', launch_ms=ms,
                           algorithmic_bytes=mbytes, achieved_GBs=mbytes / (ms * 1e-3) / 1e9,
                           frac_hbm_peak=mbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=pr.get('traffic_bytes'),
                           traffic_source=pr.get('source'), rocprof=pr.get('rocprof'))
    return out


def einsum_q_sweep(dev, B, H, W):
    """BASELINE.json's kernel target: the mask-logit einsum bqc,bchw->bqhw (1024 x 1024 -> 256 x 256 mask feature) at the benched
    shape (B, Q = 100), at Q = 200, and at the shapes where the fixed cost amortises -- configs[3]'s per-GPU B = 4 x Q = 200 and
    configs[2]'s B = 16 x Q = 100 -- both arithmetic modes, 20 launches back to back in one hipGraph (cache-warm:
    the 10-call sequence of a forward is)."""
    from cgg_amd import ops, runtime
    HW4 = (H // 4) * (W // 4)
    g = torch.Generator().manual_seed(7)
    res = []
    forms = (('bf16 stored', 'bf16 MFMA, f32 logits out', False, False),
             ('x3 stored', 'f32-class f16 x 3 MFMA, f32 logits out', True, False),
             ('bf16 fused', 'bf16 MFMA, consumer fused: threshold bits out, logits never stored', False, True),
             ('x3 fused', 'f32-class f16 x 3 MFMA, consumer fused: threshold bits out, logits never stored', True, True),
             ('bf16 fused astat', 'bf16 MFMA, consumer fused + query tiles stationary in registers (cgg_mask_logits_bits_astat): '
                                  'threshold bits out, logits never stored', False, 'astat'))
    for Bs, Q in ((B, 100), (B, 200), (4, 200), (16, 100)):
        feat = torch.randn(Bs, 256, H // 4, W // 4, generator=g).to(dev)
        emb = torch.randn(Bs, Q, 256, generator=g).to(dev)
        # `fused` = the consumer in the epilogue, logits NEVER stored: the full-resolution contraction with the attention-mask rule
        # (mask2former_head.py:749-759: threshold) applied to the accumulators, one bit per (query, pixel) out -- the form SURVEY 7
        # names as the only one that can approach the MFMA roofline (the stored-f32-logits form is HBM-bound at AI = 56-100 FLOP/B).
        for form, mode, split, fused in forms:
            if Bs != B and not fused and split:
                continue                                   # the big shapes: fused forms + the bf16 stored form
            with runtime.precision_scope('fp32' if split else 'bf16'):
                packed = ops.pack_mask_feature(feat, 1, split)
                if split:
                    packed.f32 = None                      # the f16 x 3 kernel (the exact-f32 path keeps an un-packed copy)
                call = (lambda: ops.mask_logits_bits_astat(emb, packed)) if fused == 'astat' else \
                    (lambda: ops.mask_logits(emb, packed, want_logits=False, want_bits=True)) if fused else \
                    (lambda: ops.mask_logits(emb, packed))
                for _ in range(5):
                    call()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    call()
                e.record()
                torch.cuda.synchronize()
                ms_eager = s.elapsed_time(e) / 20
                # the same 20 launches replayed from ONE hipGraph (how the forward runs them): an eager Python loop issues a launch
                # every ~8-10 us, which is the floor of `launch_ms_eager` for kernels shorter than that
                ms = ms_eager
                try:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        call()
                    torch.cuda.current_stream().wait_stream(side)
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        for _ in range(20):
                            call()
                    gr.replay()
                    torch.cuda.synchronize()
                    s.record()
                    for _ in range(5):
                        gr.replay()
                    e.record()
                    torch.cuda.synchronize()
                    ms = s.elapsed_time(e) / 100
                    del gr
                except Exception as ex:          # loud: the line says which figure it carries
                    print(f'bench.py: einsum sweep graph capture failed ({type(ex).__name__}: {ex}); eager timing', file=sys.stderr)
                del packed
            fl = 2.0 * Bs * Q * 256 * HW4
            by = Bs * (256 * HW4 * (4 if split else 2) + Q * 256 * 4 + (Q * HW4 // 8 if fused else Q * HW4 * 4))
            peak = X3_PEAK_TF if split else MFMA_BF16_PEAK_TF
            tf = fl / (ms * 1e-3) / 1e12
            ai = fl / by
            res.append(dict(batch=Bs, queries=Q, form=form, mode=mode, launch_ms=ms, launch_ms_eager_loop=ms_eager,
                            timed='20 launches back to back in one hipGraph, 5 replays' if ms != ms_eager else 'eager launches', tflops=tf, frac_mfma_peak=tf / peak, mfma_peak_tf=peak,
                            frac_bf16_mfma_peak=tf / MFMA_BF16_PEAK_TF,
                            frac_hbm_roofline_attainable=tf / min(peak, ai * HBM_PEAK_GBS / 1e3), GBs=by / (ms * 1e-3) / 1e9,
                            frac_hbm_peak=by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, algorithmic_bytes=by, flops=fl,
                            arithmetic_intensity=ai, launches_q_split=ops.mask_logits_launches(Q, split)))
        del feat, emb
        torch.cuda.empty_cache()
    # what this device's memory system delivers to a pure read + write stream of the same size class (torch device copy, 128 MiB
    # in + 128 MiB out): the practical ceiling the `GBs` figures above sit under (the 8 TB/s of `frac_hbm_peak` is the pin rate)
    src = torch.empty(32 << 20, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        dst.copy_(src)
    e.record()
    torch.cuda.synchronize()
    cms = s.elapsed_time(e) / 20
    res.append(dict(reference='device-to-device copy, 128 MiB read + 128 MiB written per launch', launch_ms=cms,
                    GBs=2 * src.numel() * 4 / (cms * 1e-3) / 1e9, frac_hbm_peak=2 * src.numel() * 4 / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS))
    return res


def workload_child(workload, extra_args, timeout=900):
    """Another BASELINE config measured by THIS script in a child process (started after the parent has finished its GPU work,
    never an exec from a GPU-initialised process) -> its parsed JSON line or an error record."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', workload, '--full-line'] + list(extra_args)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode != 0 or not line:
            return dict(error=f'child exited {r.returncode}', stderr_tail=r.stderr[-400:])
        d = json.loads(line[-1])
        d['how'] = 'python bench.py ' + ' '.join(cmd[2:]) + ' in a child process of this run'
        return d
    except Exception as e:
        return dict(error=f'{type(e).__name__}: {e}')


def train_step_child(args, precision='fp32', workload='cfg2'):
    """A training config's step (`--workload cfg2 | cfg3`) in a CHILD process started after this process has finished its GPU work
    (never an exec from a GPU-initialised process); its JSON line is embedded as `train_step` (precision fp32 = the reference's
    training arithmetic, open_set/apis/train.py:182-189) / `train_step.bf16_mode` (bf16 autocast: narrower, secondary)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', workload, '--steps', str(args.train_steps), '--warmup', '3',
           '--precision', precision, '--full-line']
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode != 0 or not line:
            return dict(error=f'child exited {r.returncode}', stderr_tail=r.stderr[-400:])
        d = json.loads(line[-1])
        # launches per step and the dominant kernel come from the committed per-step kernel table of the same command
        # (profiles/r4_train_step_kernels_<precision>.txt: rocprofv3 --kernel-trace of `bench.py --mode train --precision <p>`,
        # last 3 steps) -- labelled as such
        prof = {}
        try:
            table = ('r6_train_step_kernels_%s.txt' if workload == 'cfg2' else 'r6_cfg3_train_step_kernels_%s.txt') % precision
            if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', table)):
                table = table.replace('r6_', 'r5_')
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', table)) as f:
                rows = f.read().splitlines()
            head = [r for r in rows if r.startswith('step (eager')][0]
            hdr = next(i for i, r in enumerate(rows) if r.split()[:2] == ['us/step', 'calls'])
            top = rows[hdr + 1].split(None, 3)
            import re
            mh = re.search(r': ([0-9.]+) launches, ([0-9.]+) us of kernel time', head)
            own = [r for r in rows if r.startswith('hand-written (cgg_*)')]
            mo = re.search(r'= ([0-9.]+) % of the kernel time', own[0]) if own else None
            prof = dict(launches_per_step=float(mh.group(1)), kernel_ms_per_step=float(mh.group(2)) / 1e3,
                        hand_written_share_of_kernel_time=float(mo.group(1)) / 100.0 if mo else None,
                        dominant_kernel=dict(name=top[3].split('(')[0], ms_per_step=float(top[0]) / 1e3, launches_per_step=float(top[1]),
                                             share_of_kernel_time=float(top[2]) / 100.0),
                        profile_source=f'committed profile: profiles/{table} (rocprofv3 --kernel-trace of this command)')
        except Exception:
            pass
        return dict(value=d['value'], unit=d['unit'], n_gpus=d.get('n_gpus', 1), ms_per_step=d['ms_per_step'], steps=d['steps'],
                    warmup=d['warmup'], dtype=d['dtype'], workload=d['config']['workload'], loss=d.get('loss'),
                    peak_mem_gb=d.get('peak_mem_gb'), roofline=d.get('roofline'), kernels=d.get('kernels'),
                    how=f'python bench.py --workload {workload} --precision {precision} in a child process of this run', **prof)
    except Exception as e:
        return dict(error=f'{type(e).__name__}: {e}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=10,
                    help='the K-step timed region is repeated this many times back to back; the median region is reported')
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS),
                    help="BASELINE.json config: cfg1 (default: the metric's config, forward-only), cfg2 (R50 training step), cfg3 "
                         '(Swin-B + 200 queries training step, one GPU\'s share of the DDP batch), cfg4 (COCO-panoptic 1333x800 forward)')
    ap.add_argument('--mode', default=None, choices=['infer', 'train'],
                    help='(older spelling) infer = --workload cfg1, train = --workload cfg2')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default: the workload\'s)')
    ap.add_argument('--bucket-mb', type=int, default=64, help='gradient all-reduce bucket size (train)')
    ap.add_argument('--size', type=int, default=None, help='square input size (default: the workload\'s H x W)')
    ap.add_argument('--queries', type=int, default=None)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="fp32 (default, the headline): parity mode -- every contraction in f32-class arithmetic (f16 x 3 split MFMA, "
                         "f32 accumulate; the mode the 1e-3 / bit-exact parity tests run in); bf16: throughput mode, narrower "
                         "than the reference's arithmetic, reported as the secondary `bf16_mode` object")
    ap.add_argument('--graph', type=int, default=1,
                    help='1: capture the step once and replay it from a hipGraph (default; the eager step is '
                         'host-bound at ~250-500 launches); 0: eager launches')
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
    ap.add_argument('--no-bf16-mode', action='store_true', help='skip the secondary bf16 (throughput-mode) timing')
    ap.add_argument('--train-step', type=int, default=1,
                    help="1 (default, rank 0 of a 1-GPU run): also run configs[2]'s training step in a child process -> `train_step`")
    ap.add_argument('--train-steps', type=int, default=5)
    ap.add_argument('--no-einsum-sweep', action='store_true', help='skip the Q = 100 / 200 mask-logit einsum launches (profile runs)')
    ap.add_argument('--extra-workloads', type=int, default=1,
                    help='1 (default, rank 0 of a 1-GPU cfg1 run): also measure configs[3] and configs[4] in child processes -> `extra`')
    ap.add_argument('--full-line', action='store_true',
                    help='print the FULL record as the stdout line (what a parent bench.py run parses from its children); default: '
                         'the compact line, full record in gpurun_out/bench_full*.json')
    args = ap.parse_args()
    if args.workload is None:
        args.workload = 'cfg2' if args.mode == 'train' else 'cfg1'
    wl = WORKLOADS[args.workload]
    args.mode = wl['mode']
    args.backbone, args.panoptic = wl['backbone'], wl['panoptic']
    args.hw = (args.size, args.size) if args.size else wl['hw']
    if args.queries is None:
        args.queries = wl['queries']
    if args.batch is None:
        args.batch = wl['batch']
    primary = args.workload == 'cfg1'       # the driver's default line: carries the secondary objects (bf16, host results, CPU, ...)

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
    B, (H, W) = args.batch, args.hw
    g = torch.Generator().manual_seed(1234 + rank)
    img_cpu = torch.randn(B, 3, H, W, generator=g)
    img = img_cpu.to(dev)
    metas = workload_metas(args, B)

    if args.mode == 'train':
        return train_main(args, cfg, model, img, metas, dev, rank, world)

    def barrier():
        torch.cuda.synchronize()             # (gloo dry run: the device work must be done before the host barrier)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    Q = args.queries
    main_mode = time_mode(args, model, img, metas, dev, args.precision, barrier, collect_events=(args.precision == 'fp32'))
    dt = main_mode['dt']
    tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if args.shared_devices else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    kernels = {}
    if args.precision == 'fp32' and rank == 0:
        kernels = kernel_summaries(args, main_mode.get('events', {}), main_mode.get('meta', {}), B, H, W, Q)
    roofline = kernels.pop('gemm_x3', None)

    # agreement with the f32 CPU oracle at this config's real shapes, measured by tests/test_fullsize_gpu.py on an MI355X and committed
    # by tools/collect_agreement.sh (run on the GPU box, then `publish`): the bf16 record of THIS config under `bf16_mode`, parity
    # mode's own no-injection record under `config`. A key this line is meant to publish must be there: no `record: null`.
    agree_all, agree_src = agreement_records()
    agree_key = {'cfg1': 'configs1_bf16', 'cfg4': 'configs4_bf16'}.get(args.workload)
    need = ([agree_key] if (agree_key and not args.no_bf16_mode and args.precision == 'fp32') else []) + \
        (['configs1_fp32_no_injection'] if primary else [])
    missing = [k for k in need if not isinstance(agree_all.get(k), dict)]
    if missing:
        raise SystemExit(f'bench.py: {agree_src} lacks the agreement record(s) {missing} this line publishes -- regenerate it with '
                         'tools/collect_agreement.sh (run on the GPU box, then publish)')
    other = None
    if not args.no_bf16_mode and args.precision == 'fp32':
        o = time_mode(args, model, img, metas, dev, 'bf16', barrier, collect_events=False)
        other = dict(value=B * world * args.steps / o['dt'], unit='images/sec', ms_per_step=o['dt'] / args.steps * 1e3, dtype='bf16',
                     latency_ms_per_batch=o['latency_ms'], how=o['how'],
                     note='throughput mode: bf16 MFMA contractions (narrower than the reference\'s f32 arithmetic; outside the '
                          '1e-3 / bit-exact parity clause -- never `value`)' +
                          ('; NOT a usable panoptic path: one flipped segment-level threshold decision moves up to a third of an '
                           'image\'s pixels (panoptic_pixel_agreement below)' if args.panoptic else ''),
                     agreement_with_f32_oracle=dict(record=agree_all.get(agree_key), key=agree_key,
                                                    source=f'committed profile: {agree_src} (tests/test_fullsize_gpu.py, bf16 mode, no mask injection)'))
    host = None
    if args.host_results and rank == 0 and primary:
        with runtime.precision_scope(args.precision):
            host = host_results_rate(args, model, img, metas, dev)
    sweep = None
    if rank == 0 and (H, W) == (1024, 1024) and not args.no_einsum_sweep and primary:
        sweep = einsum_q_sweep(dev, B, H, W)
    # the x3a range guard: a value outside +-4094 anywhere in the timed steps raised the device flag (tools/test.py aborts on it)
    overflow = bool(ops.x3_overflow_check(dev, reset=True)) if args.precision == 'fp32' else None
    if overflow:
        print('bench.py: x3a overflow flag raised -- an activation left the representable range; the result is INVALID',
              file=sys.stderr)
    if rank == 0:
        f32 = args.precision == 'fp32'
        res = dict(metric=f'images/sec (COCO-shaped {W}x{H}, {Q} queries, forward-only)' if not primary else
                   'images/sec (COCO-shaped 1024x1024, 100 queries, forward-only)',
                   value=B * world * args.steps / dt, unit='images/sec', n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
                   scaling='weak', vs_baseline=None,
                   dtype='f32 (f16x3 split MFMA, f32 accumulate)' if f32 else 'bf16', data='synthetic',
                   config=dict(workload=(f'configs[1]: R50 + {Q} queries, {H}x{W}, batch {B}/GPU, forward-only '
                                         '(backbone + MSDeformAttn pixel decoder + 9-layer masked-attention '
                                         'decoder + mask logits + instance post-processing, results on device)') if primary else
                               (WORKLOADS[args.workload]['name'] + f' [{Q} queries, {H}x{W}, batch {B}/GPU] (backbone + MSDeformAttn pixel '
                                'decoder + 9-layer masked-attention decoder + mask logits + ' +
                                ('panoptic' if args.panoptic else 'instance') + ' post-processing, results on device)'),
                               global_batch=B * world, parallelism=f'replicas x{world}',
                               precision=args.precision + (' = parity mode: every contraction of the path in f32-class arithmetic '
                                                           '(two f16 pieces per f32 operand, three f16 MFMAs per product, f32 '
                                                           'accumulation: as accurate as an f32 GEMM, tests/test_x3_gpu.py, '
                                                           'tests/test_x3s_gpu.py); softmax / norms / sampling f32. Parity bar: '
                                                           '1e-3 per layer with the oracle\'s attention masks injected / 2e-2 end '
                                                           'to end without injection (mask-threshold flips feed back through 9 '
                                                           'layers), integer outputs bit-exact' if f32 else ''),
                               activation_format=('x3a: activations stored pre-split (8 x f16 hi | 8 x f16 lo of 16 a per 8 channels, '
                                                  '|a| < 4094, device overflow flag checked after the timed region)'
                                                  if (f32 and runtime.x3a_enabled()) else 'f32'),
                               parity_mode_agreement_without_injection=dict(
                                   record=agree_all.get('configs1_fp32_no_injection'),
                                   source=f'committed profile: {agree_src} (tests/test_fullsize_gpu.py::'
                                          'test_configs1_fp32_mode_end_to_end_without_injection, fp32 = parity mode)') if primary else None,
                               x3_overflow=overflow, library_fallbacks=dict(count=runtime.library_fallbacks(), sites=dict(runtime.FALLBACKS)),
                               hip_graph=main_mode['hip_graph'], ranks_share_devices=bool(args.shared_devices),
                               pipeline=main_mode['how'],
                               timed_region=f'{args.steps} steps, repeated {args.repeats}x back to back; median region '
                                            f'{dt * 1e3:.1f} ms (min {min(main_mode["regions_s"]) * 1e3:.1f}, max '
                                            f'{max(main_mode["regions_s"]) * 1e3:.1f})'),
                   latency_ms_per_batch=main_mode['latency_ms'], bf16_mode=other, host_results=host, roofline=roofline,
                   kernels=kernels, einsum_mfma_target=sweep)
        if not args.no_cpu_baseline and world == 1 and primary:
            res['cpu_baseline'] = cpu_baseline(args, cfg, model, img_cpu)
        if (args.train_step or args.extra_workloads) and world == 1 and primary:
            torch.cuda.synchronize()
            del main_mode
            torch.cuda.empty_cache()
        if args.train_step and world == 1 and primary:
            # parity-mode (f32-class) training step first -- the arithmetic the reference trains in -- then the bf16 autocast one
            ts = train_step_child(args, 'fp32')
            ts['bf16_mode'] = train_step_child(args, 'bf16')
            res['train_step'] = ts
        if args.extra_workloads and world == 1 and primary:
            # the other BASELINE configs, each by `bench.py --workload ...` in a child process: configs[3] = the Swin-B / 200-query
            # training step (one GPU's share of the DDP batch; parity mode, bf16 nested), configs[4] = the panoptic forward at
            # 1333 x 800 (parity mode = `value`, bf16 nested) -- each with its own `roofline` from live HIP events
            c3 = train_step_child(args, 'fp32', workload='cfg3')
            c3['bf16_mode'] = train_step_child(args, 'bf16', workload='cfg3')
            c4 = workload_child('cfg4', ['--steps', str(args.steps), '--warmup', str(args.warmup), '--repeats', '3'])
            res['extra'] = {'configs[3]': c3, 'configs[4]': c4}
    if world > 1 and primary and args.train_step and not args.shared_devices:
        # N ranks (driver's scaling runs): the SAME processes then take configs[2]'s image-parallel training step -- batch 16 per rank,
        # gradients all-reduced over RCCL in 64-MiB buckets -- so that the N-GPU line shows RCCL ranks, not only inference replicas
        # (VERDICT r4 next 8). Every rank runs it; rank 0 attaches the line as `train_step` (n_gpus = N, parity mode).
        import copy
        torch.cuda.synchronize()
        main_mode = None
        model = None
        torch.cuda.empty_cache()
        targs = copy.copy(args)
        targs.workload, targs.mode, targs.batch = 'cfg2', 'train', WORKLOADS['cfg2']['batch']
        tcfg, tmodel = build_model(targs, dev)
        timg = torch.randn(targs.batch, 3, H, W, generator=torch.Generator().manual_seed(99 + rank)).to(dev)
        ts = train_run(targs, tcfg, tmodel, timg, synthetic.img_metas(targs.batch, H, W), dev, rank, world, steps=args.train_steps,
                       warmup=2)
        if rank == 0:
            ts['how'] = f'the {world} ranks of this run, in-process after the inference region (RCCL gradient all-reduce)'
            res['train_step'] = ts
    if rank == 0:
        emit_result(args, res)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
