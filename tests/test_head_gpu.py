"""-m gpu: the product head / fusion head (HIP path, through the C ABI) against the oracle on the same
seeded inputs and the same weights.

Tolerances: mask logits 1e-3 absolute (north_star) in fp32 ('split') mode; class / embedding outputs 1e-3;
assignment indices (argmax class, top-k (query, class) sets, panoptic ids) exact.
"""
import os
import warnings

import pytest
import torch

import cgg_amd  # noqa: F401
from cgg_amd import ops, registry, runtime, synthetic
from oracle import head as OH

from util import MaskTeacher, build_heads, small_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def heads(dev):
    cfg = small_cfg()
    prod, orc = build_heads(cfg)
    return cfg, prod.to(dev), orc


def _feats(B, H, W, seed):
    return synthetic.backbone_feats(B, H, W, channels=(64, 128, 256, 512), seed=seed)


@pytest.mark.parametrize('B,H,W', [(2, 128, 128), (1, 160, 224)])
def test_head_forward_vs_oracle(dev, heads, B, H, W):
    cfg, prod, orc = heads
    feats = _feats(B, H, W, seed=3)
    metas = synthetic.img_metas(B, H, W)
    teacher = MaskTeacher(orc)
    with torch.no_grad():
        oc, oe, om = teacher.run_oracle(lambda: orc.forward(feats, metas))
        prod.attn_mask_hook = teacher.hook
        try:
            pc, pe, pm = prod.forward([f.to(dev) for f in feats], metas)
        finally:
            prod.attn_mask_hook = None
    teacher.check()
    assert len(pc) == len(oc) == cfg['panoptic_head']['transformer_decoder']['num_layers'] + 1
    for li in range(len(oc)):
        assert (pc[li].cpu() - oc[li]).abs().max().item() <= 1e-3, li
        assert (pe[li].cpu() - oe[li]).abs().max().item() <= 1e-3, li
        err = (pm[li].cpu() - om[li]).abs().max().item()
        assert err <= 1e-3, (li, err)  # north_star: mask logits within 1e-3


def test_pixel_decoder_vs_oracle(dev, heads):
    cfg, prod, orc = heads
    feats = _feats(2, 96, 160, seed=5)
    with torch.no_grad():
        omf, omem = orc.pixel_decoder(feats)
        pmf, pmem = prod.pixel_decoder([f.to(dev) for f in feats])
    assert (pmf.cpu() - omf).abs().max().item() <= 2e-4
    for a, b in zip(pmem, omem):
        assert (a.cpu() - b).abs().max().item() <= 2e-4


def test_head_simple_test_and_instance_postprocess(dev, heads):
    cfg, prod, orc = heads
    B, H, W = 2, 128, 160
    feats = _feats(B, H, W, seed=7)
    metas = synthetic.img_metas(B, H, W, ori=(150, 200))
    for m in metas:
        m['img_shape'] = (120, 150, 3)  # padded batch: crop then rescale
    fcfg = dict(cfg['panoptic_fusion_head'])
    fcfg.update(test_cfg=cfg['test_cfg'])
    fusion = registry.build_head(fcfg).to(dev)
    teacher = MaskTeacher(orc)
    with torch.no_grad():
        ocls, oemb, oup = teacher.run_oracle(lambda: orc.simple_test(feats, metas))
        prod.attn_mask_hook = teacher.hook
        try:
            pcls, pemb, pmasks, _, _ = prod.simple_test([f.to(dev) for f in feats], metas)
        finally:
            prod.attn_mask_hook = None
        teacher.check()
        up = pmasks.upsampled().cpu()
        assert (up - oup).abs().max().item() <= 1e-3
        res = fusion.simple_test(pcls, pemb, pmasks, metas, rescale=True)
    for b in range(B):
        omp = OH.crop_rescale(oup[b], metas[b], True)
        for key, embs in (('all_results', fusion.all_class_embs), ('novel_results', fusion.novel_class_embs),
                          ('base_results', fusion.base_class_embs)):
            k = min(100, oemb[b].shape[0] * (embs.shape[0] - 1))
            olab, obox, omask, oqi, osc = OH.instance_postprocess_emb(oemb[b], omp, embs.cpu(), k)
            plab, pbox, pmask = res[b][key]
            ncls = embs.shape[0] - 1
            # (query, class) assignment indices: exact as a set (topk(sorted=False) order is unspecified)
            okeys = sorted((oqi * ncls + olab).tolist())
            # recover the product's query index from its boxes is not possible -> compare labels multiset
            assert sorted(plab.cpu().tolist()) == sorted(olab.tolist())
            # per detection: pair every oracle detection with the unused product detection of the same
            # label whose mask differs least (top-k order is unspecified and near-zero scores tie)
            plab_c, pbox_c, pmask_c = plab.cpu(), pbox.cpu(), pmask.cpu()
            used = set()
            for j in range(olab.numel()):
                cand = [i for i in range(plab_c.numel()) if i not in used and plab_c[i] == olab[j]]
                assert cand
                d = torch.stack([(pmask_c[i] != omask[j]).sum() for i in cand])
                sd = torch.stack([(pbox_c[i, 4] - obox[j, 4]).abs() for i in cand])
                best = int(torch.argmin(d.float() + sd * 1e6))
                i = cand[best]
                used.add(i)
                # masks identical except pixels whose logit sits within f32 rounding of the threshold
                assert d[best].item() <= 8, (key, j, d[best].item())
                assert sd[best].item() <= 1e-3  # scores inherit the 1e-3 logit tolerance
                if d[best].item() == 0:
                    assert torch.equal(pbox_c[i, :4], obox[j, :4])
            assert len(okeys) == plab.numel()


def test_panoptic_postprocess_vs_oracle(dev):
    g = torch.Generator().manual_seed(11)
    Q, h, w, ncls, nth = 30, 40, 56, 12, 8
    emb = torch.randn(Q, 64, generator=g)
    cls_embs = torch.randn(ncls + 1, 64, generator=g)
    cls_embs[-1] = 0
    # one compact positive disk per query (overlapping neighbours are resolved by the argmax)
    ys = torch.arange(h).view(1, h, 1).float()
    xs = torch.arange(w).view(1, 1, w).float()
    cy = torch.rand(Q, 1, 1, generator=g) * h
    cx = torch.rand(Q, 1, 1, generator=g) * w
    r = 4 + torch.rand(Q, 1, 1, generator=g) * 6
    logits = 6 - ((ys - cy)**2 + (xs - cx)**2) / r**2 * 6 + torch.randn(Q, h, w, generator=g) * 0.3
    fcfg = dict(type='MaskFormerFusionHeadOpen', num_things_classes=nth, num_stuff_classes=ncls - nth,
                panoptic_mode=True, test_cfg=dict(eval_types=['all_results'], object_mask_thr=0.2, iou_thr=0.5,
                                                  filter_low_score=True, stuff_area_limit=64))
    fusion = registry.build_head(fcfg).to(dev)
    up = (h * 4, w * 4)
    meta = dict(img_shape=(up[0] - 8, up[1] - 12, 3), ori_shape=(up[0] - 8, up[1] - 12, 3))
    want_in = torch.nn.functional.interpolate(logits[None], up, mode='bilinear', align_corners=False)[0]
    want_in = OH.crop_rescale(want_in, meta, False)
    want = OH.panoptic_postprocess_emb(emb, want_in, cls_embs, ncls, nth, 0.2, 0.5, True, 64)
    from cgg_amd.mask2former_head import LowResMasks
    got = fusion.panoptic_postprocess_emb(emb.to(dev), LowResMasks(logits.to(dev), up), cls_embs.to(dev), meta,
                                          False).cpu()
    assert got.shape == want.shape and got.dtype == torch.int32
    mism = (got != want).float().mean().item()
    assert mism <= 1e-4, mism  # argmax ties at f32 rounding only
    assert len(torch.unique(want)) > 2


def test_bf16_mode_runs_and_is_close(dev, heads):
    cfg, prod, orc = heads
    feats = _feats(1, 128, 128, seed=9)
    metas = synthetic.img_metas(1, 128, 128)
    # throughput mode (bf16 operands everywhere): with the oracle's attention masks injected (a bf16 logit
    # near 0 may flip a mask bit, which is a decision change, not an arithmetic error) the logits must stay
    # within a few bf16 ulps of the logit scale: max <= 5 % of the scale, mean <= 1 %.
    teacher = MaskTeacher(orc, margin=0.5)
    with torch.no_grad():
        _, _, om = teacher.run_oracle(lambda: orc.forward(feats, metas))
        prod.attn_mask_hook = teacher.hook
        try:
            with runtime.precision_scope('bf16'):
                _, _, pm = prod.forward([f.to(dev) for f in feats], metas)
        finally:
            prod.attn_mask_hook = None
    scale = om[-1].abs().max().item()
    err = (pm[-1].cpu() - om[-1]).abs()
    assert err.max().item() <= 0.05 * scale + 0.05, (err.max().item(), scale)
    assert err.mean().item() <= 0.01 * scale, (err.mean().item(), scale)
    assert all(ok for ok, _ in teacher.seen), teacher.seen   # own bits agree away from |logit| < 0.5


def test_stream_path_matches_module_path(dev):
    """Throughput-mode inference stream (channel-last bf16 features -> packed mask feature, GEMM 1x1 convs, HIP
    NHWC GroupNorm) against the module-by-module bf16 path on the same weights and features."""
    from cgg_amd import registry
    from util import head_cfg, randomize
    cfg = small_cfg(num_queries=20, depth=50)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        head = registry.build_head(head_cfg(cfg))
    randomize(head, seed=5)
    head = head.to(dev).eval()
    B, H, W = 2, 128, 160
    feats = synthetic.backbone_feats(B, H, W, channels=(256, 512, 1024, 2048), seed=3)
    feats16 = [f.to(dev).bfloat16().contiguous(memory_format=torch.channels_last) for f in feats]
    feats32 = [f.float() for f in feats16]
    metas = synthetic.img_metas(B, H, W)
    with torch.no_grad(), runtime.precision_scope('bf16'):
        assert head.pixel_decoder.stream_ready(feats16)
        assert not head.pixel_decoder.stream_ready(feats32)
        # a bf16 logit that lands on the other side of 0 flips an attention-mask bit and the two decoders drift
        # apart, so the module path's masks are replayed into the stream path (as util.MaskTeacher does)
        rec = {}
        head.attn_mask_hook = lambda i, bits: rec.setdefault(i, bits.clone())
        c2, e2, m2 = head._forward(feats32, metas, all_masks=False)
        head.attn_mask_hook = lambda i, bits: rec[i].clone()
        c1, e1, m1 = head._forward(feats16, metas, all_masks=False)
        head.attn_mask_hook = None
        # pixel decoder outputs on their own
        mf, mems, sizes = head.pixel_decoder.forward_stream(feats16)
        mf2, mems2 = head.pixel_decoder(feats32)
    a, b = mf.float().permute(0, 3, 1, 2), mf2
    assert (a - b).abs().max().item() <= 0.05 * b.abs().max().item()
    assert (a - b).abs().mean().item() <= 0.01 * b.abs().max().item()
    for x, y in zip(mems, mems2):
        y = y.flatten(2).transpose(1, 2)
        assert (x - y).abs().max().item() <= 0.05 * y.abs().max().item()
    s = m2[-1].abs().max().item()
    assert (m1[-1] - m2[-1]).abs().mean().item() <= 0.02 * s
    assert (e1[-1] - e2[-1]).abs().mean().item() <= 0.02 * e2[-1].abs().max().item()


@pytest.mark.parametrize('stages', [2, 3])
def test_stage_pipeline(dev, stages):
    """pipeline.StagePipeline (stage i of batch k overlapped with stage i-1 of batch k+1, one hipGraph per stage and
    slot, one stream per stage) returns what the plain sequential `simple_test` returns, batch after batch."""
    from cgg_amd.pipeline import detector_pipeline
    from util import randomize
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
    randomize(model, seed=9)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    model = model.to(dev).eval()
    B, H, W = 2, 128, 192
    metas = synthetic.img_metas(B, H, W)
    g = torch.Generator().manual_seed(11)
    imgs = [torch.randn(B, 3, H, W, generator=g).to(dev) for _ in range(5)]
    with torch.no_grad(), runtime.precision_scope('bf16'):
        want = []
        for im in imgs:
            res = model.simple_test(im, metas, rescale=True, device_results=True)
            want.append([{k: tuple(t.clone() for t in v) for k, v in r.items()} for r in res])
        torch.cuda.synchronize()
        pipe = detector_pipeline(model, imgs[0], metas, stages=stages, rescale=True, device_results=True)
        got = []
        for im in imgs:
            slot = pipe.submit(im)
            res = pipe.wait(slot)          # stream-ordered: the clones below run after this batch's decode
            got.append([{k: tuple(t.clone() for t in v) for k, v in r.items()} for r in res])
        pipe.flush()
        torch.cuda.synchronize()
    for w_b, g_b in zip(want, got):
        for w_img, g_img in zip(w_b, g_b):
            assert set(w_img) == set(g_img)
            for k in w_img:
                wl, wb, wm = w_img[k]
                gl, gb, gm = g_img[k]
                # the forward is deterministic (no atomics on its data path; only the mask-score reduction uses f32
                # atomics): same detections, same masks, scores equal to rounding
                assert sorted(wl.cpu().tolist()) == sorted(gl.cpu().tolist())
                assert abs(wb[:, 4].sum().item() - gb[:, 4].sum().item()) <= 1e-4 * max(1.0, wb[:, 4].sum().item())
                assert int(wm.sum()) == int(gm.sum())


def test_head_200_queries(dev):
    """BASELINE configs[3] uses 200 queries: the attention / split mask-logit kernels tile 128 queries per launch and
    the wrappers split larger query sets. Parity mode vs the oracle with its masks injected, then a throughput-mode run."""
    cfg = small_cfg(num_queries=200, depth=50)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prod, orc = build_heads(cfg)
    prod = prod.to(dev).eval()
    orc.eval()
    B, H, W = 1, 96, 128
    feats = synthetic.backbone_feats(B, H, W, channels=(256, 512, 1024, 2048), seed=13)
    metas = synthetic.img_metas(B, H, W)
    teacher = MaskTeacher(orc)
    with torch.no_grad():
        oc, oe, om = teacher.run_oracle(lambda: orc.forward(feats, metas))
        prod.attn_mask_hook = teacher.hook
        pc, pe, pm = prod.forward([f.to(dev) for f in feats], metas)
        prod.attn_mask_hook = None
    teacher.check()
    assert pm[-1].shape == (B, 200, H // 4, W // 4)
    assert (pm[-1].cpu() - om[-1]).abs().max().item() <= 1e-3
    assert (pe[-1].cpu() - oe[-1]).abs().max().item() <= 1e-3
    with torch.no_grad(), runtime.precision_scope('bf16'):
        f16 = [f.to(dev).bfloat16().contiguous(memory_format=torch.channels_last) for f in feats]
        c, e, m = prod._forward(f16, metas, all_masks=False)
    assert torch.isfinite(m[-1]).all() and m[-1].shape == (B, 200, H // 4, W // 4)


def test_simple_test_with_caption_beam_search(dev, heads):
    """`with_caption=True` runs the caption beam search on the device (B = 1, as the reference requires) and equals the
    same search run on the CPU copy of the head (golden G8 pins the algorithm against the reference's function)."""
    import copy
    from cgg_amd.caption_search import beam_search
    _, prod, orc = heads
    prod = prod.eval()
    B, H, W = 1, 64, 96
    feats = _feats(B, H, W, seed=17)
    metas = synthetic.img_metas(B, H, W)
    with torch.no_grad():
        out = prod.simple_test([f.to(dev) for f in feats], metas, with_caption=True)
        emb = out[1]
        ids_dev = beam_search(prod, emb, 101, 102, max_len=35, beam_width=7, return_ids=True)
        cpu_head = copy.deepcopy(prod).cpu()
        ids_cpu = beam_search(cpu_head, emb.cpu(), 101, 102, max_len=35, beam_width=7, return_ids=True)
    assert out[3] is not None
    assert isinstance(ids_dev, list) and ids_dev == ids_cpu


def test_swin_b_200_queries_detector(dev):
    """BASELINE configs[3] plumbing: Swin-B backbone (in_channels 128..1024) + 200 queries through the throughput-mode
    stream (channel-last bf16 hand-over, packed mask feature, bf16 attention) and through parity mode."""
    import copy
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=200, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    cfg = copy.deepcopy(cfg)
    cfg['backbone'] = dict(type='SwinTransformer', embed_dims=128, depths=(2, 2, 6, 2), num_heads=(4, 8, 16, 32),
                           window_size=7, mlp_ratio=4, out_indices=(0, 1, 2, 3), drop_path_rate=0.3, patch_norm=True)
    cfg['panoptic_head']['in_channels'] = [128, 256, 512, 1024]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
        torch.manual_seed(0)
        model.init_weights()
    model = model.to(dev).eval()
    B, H, W = 1, 224, 288
    img = torch.randn(B, 3, H, W, device=dev)
    metas = synthetic.img_metas(B, H, W)
    with torch.no_grad():
        f32 = model.extract_feat(img)
        assert [tuple(f.shape[1:]) for f in f32] == [(128, 56, 72), (256, 28, 36), (512, 14, 18), (1024, 7, 9)]
        res32 = model.simple_test(img, metas, rescale=True, device_results=True)
        with runtime.precision_scope('bf16'):
            f16 = model.extract_feat(img)
            assert all(f.dtype == torch.bfloat16 for f in f16) and model.panoptic_head.pixel_decoder.stream_ready(f16)
            for a, b in zip(f16, f32):
                assert (a.float() - b).abs().max().item() <= 0.08 * b.abs().max().item()
            res16 = model.simple_test(img, metas, rescale=True, device_results=True)
    for r in (res32, res16):
        labels, boxes, masks = r[0]['all_results']
        assert masks.shape[1:] == (H, W) and torch.isfinite(boxes).all()


def test_lean_decode_equals_full_decode(dev):
    """Inference-only decode (cgg_decoder_tail_bf16 between layers, cached weight-only prologue, intermediate class
    heads skipped) returns the final predictions of the full stream decode on the same encoding."""
    from cgg_amd import registry
    from util import head_cfg, randomize
    cfg = small_cfg(num_queries=20, depth=50)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        head = registry.build_head(head_cfg(cfg))
    randomize(head, seed=6)
    head = head.to(dev).eval()
    B, H, W = 2, 128, 160
    feats = synthetic.backbone_feats(B, H, W, channels=(256, 512, 1024, 2048), seed=4)
    feats16 = [f.to(dev).bfloat16().contiguous(memory_format=torch.channels_last) for f in feats]
    with torch.no_grad(), runtime.precision_scope('bf16'):
        enc = head._encode(feats16)
        assert enc['stream'] and head._lean_decode_ok(256)
        full = head._decode_stream(B, enc['kvs'], enc['sizes'], enc['packed_full'], enc['pooled'], True)
        lean = head._decode_stream(B, enc['kvs'], enc['sizes'], enc['packed_full'], enc['pooled'], False)
    assert all(v is None for v in lean[2][:-1]) and len(lean[2]) == len(full[2])
    for k in range(3):
        assert torch.equal(lean[k][-1], full[k][-1])


def test_test_driver_pipeline_equals_sequential(dev, tmp_path):
    """tools/test.py: config + mmcv-layout checkpoint -> results pickle; the pipelined bf16 serving loop returns what
    the sequential loop returns (same detections / masks per image)."""
    import importlib.util
    import pickle
    import sys
    from cgg_amd.checkpoint import save_checkpoint
    from util import randomize
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
    randomize(model, seed=21)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    ck = save_checkpoint(model, str(tmp_path / 'w.pth'), meta=dict(CLASSES=tuple(f'c{i}' for i in range(10))))
    cfg_file = tmp_path / 'tiny.py'
    cfg_file.write_text('model = ' + repr(cfg) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    spec = importlib.util.spec_from_file_location('cgg_tools_test', os.path.join(root, 'tools', 'test.py'))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    out = str(tmp_path / 'r.pkl')
    a = drv.main([str(cfg_file), ck, '--out', out, '--num-images', '7', '--synthetic', '128'])
    b = drv.main([str(cfg_file), ck, '--num-images', '7', '--synthetic', '128', '--no-pipeline'])
    assert len(a) == 7 and len(b) == 7 and len(pickle.load(open(out, 'rb'))) == 7
    for ra, rb in zip(a, b):
        assert set(ra) == set(rb)
        for k in ra:
            assert sorted(ra[k][0].tolist()) == sorted(rb[k][0].tolist())
            assert int(ra[k][2].sum()) == int(rb[k][2].sum())


def mixed_shape_stream(cfg, rank, world):
    """`--data` hook for the driver test: 2 images at 128x128, 3 at 128x160 (a shape change in the middle of a
    batch), 2 at 128x128 again (the first pipeline is reused) and one trailing 128x160 image (odd batch)."""
    g = torch.Generator().manual_seed(5)
    shapes = [(128, 128)] * 2 + [(128, 160)] * 3 + [(128, 128)] * 2 + [(128, 160)]
    for i, (h, w) in enumerate(shapes):
        if i % world != rank:
            continue
        meta = synthetic.img_metas(1, h, w)[0]
        yield torch.randn(3, h, w, generator=g), dict(meta, filename=f'mixed_{i}.jpg')


def test_test_driver_mixed_shapes_keep_order(dev, tmp_path):
    """ADVICE r1 (tools/test.py): when the image shape changes, the batch in flight is drained from the pipeline that
    produced it BEFORE a new pipeline is used, non-pipelined batches do not overtake pending ones, and the trailing
    odd batch is kept -- results arrive in dataset order and equal the sequential loop's."""
    import importlib.util
    import sys
    from util import randomize
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
    randomize(model, seed=21)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    from cgg_amd.checkpoint import save_checkpoint
    ck = save_checkpoint(model, str(tmp_path / 'w.pth'))
    cfg_file = tmp_path / 'tiny.py'
    cfg_file.write_text('model = ' + repr(cfg) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    sys.path.insert(0, os.path.join(root, 'tests'))
    spec = importlib.util.spec_from_file_location('cgg_tools_test2', os.path.join(root, 'tools', 'test.py'))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    a = drv.main([str(cfg_file), ck, '--data', 'test_head_gpu:mixed_shape_stream'])
    b = drv.main([str(cfg_file), ck, '--data', 'test_head_gpu:mixed_shape_stream', '--no-pipeline'])
    widths = [128] * 2 + [160] * 3 + [128] * 2 + [160]
    assert len(a) == 8 and len(b) == 8
    for i, (ra, rb) in enumerate(zip(a, b)):
        for k in ra:
            assert ra[k][2].shape[-1] == widths[i] and rb[k][2].shape[-1] == widths[i], (i, k)   # dataset order
            assert sorted(ra[k][0].tolist()) == sorted(rb[k][0].tolist()), (i, k)
            assert int(ra[k][2].sum()) == int(rb[k][2].sum()), (i, k)


def four_shape_stream(cfg, rank, world):
    """`--data` hook: batches of four different pyramids, then every one of them AGAIN -- the captured pipelines of the first visits are
    replayed after the model's shape-keyed caches (one positional encoding, two pyramids of projection tables) have long moved on."""
    g = torch.Generator().manual_seed(6)
    shapes = [(128, 128)] * 2 + [(128, 160)] * 2 + [(160, 128)] * 2 + [(160, 160)] * 2
    shapes = shapes + shapes
    for i, (h, w) in enumerate(shapes):
        if i % world != rank:
            continue
        if i == len(shapes) // 2 and torch.cuda.is_available():
            # before the second visits: every block the caches have dropped since is overwritten with NaN (allocator churn over the
            # size classes of the encodings / tables / biases) -- a graph that still read such a block would poison its results
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            ns = sorted({sum((h2 // s_) * (w2 // s_) for s_ in (8, 16, 32)) for h2, w2 in shapes})
            sizes = [n * c for n in ns for c in (544, 512, 288, 256, 2)] + [3 << 10, 24 << 10, 400 << 10]
            junk = [torch.full((n,), float('nan'), device='cuda') for _ in range(24) for n in sizes]
            torch.cuda.synchronize()
            del junk
        meta = synthetic.img_metas(1, h, w)[0]
        yield torch.randn(3, h, w, generator=g), dict(meta, filename=f'four_{i}.jpg')


def test_test_driver_replays_old_pipelines_after_the_caches_moved_on(dev, tmp_path):
    """A hipGraph replays raw device addresses. The model's shape-keyed caches are bounded (`_pos_cached` holds ONE pyramid's
    encoding, `_proj_pos_table` two pyramids' tables), so a pipeline captured for an earlier shape must keep the tensors its graphs
    read alive itself (`runtime.keepalive_scope` around StagePipeline's warm-up + capture; round 6 -- before, a revisited shape
    replayed freed memory and was only right while the allocator had not reused the blocks). Four pyramids, each visited twice, with
    allocator churn in between: pipelined results == the sequential loop's."""
    import importlib.util
    import sys
    from util import randomize
    from cgg_amd.checkpoint import save_checkpoint
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
    randomize(model, seed=23)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    ck = save_checkpoint(model, str(tmp_path / 'w.pth'))
    cfg_file = tmp_path / 'tiny.py'
    cfg_file.write_text('model = ' + repr(cfg) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    sys.path.insert(0, os.path.join(root, 'tests'))
    spec = importlib.util.spec_from_file_location('cgg_tools_test3', os.path.join(root, 'tools', 'test.py'))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    b = drv.main([str(cfg_file), ck, '--data', 'test_head_gpu:four_shape_stream', '--no-pipeline'])
    torch.cuda.empty_cache()                                   # freed blocks really go back: stale addresses would not survive
    a = drv.main([str(cfg_file), ck, '--data', 'test_head_gpu:four_shape_stream'])
    assert len(a) == 16 and len(b) == 16
    for i, (ra, rb) in enumerate(zip(a, b)):
        for k in ra:
            assert ra[k][2].shape == rb[k][2].shape, (i, k)
            assert sorted(ra[k][0].tolist()) == sorted(rb[k][0].tolist()), (i, k)
            assert int(ra[k][2].sum()) == int(rb[k][2].sum()), (i, k)


def test_test_driver_rle_results_equal_bool_masks(dev, tmp_path):
    """tools/test.py --rle (pinned staging + asynchronous copies + C++ COCO-RLE encoder, overlapped with the pipeline):
    every detection's decoded RLE equals the bool mask the plain driver returns, per class, in dataset order."""
    import importlib.util
    import sys
    import numpy as np
    from cgg_amd.checkpoint import save_checkpoint
    from cgg_amd.host_results import rle_to_mask
    from util import randomize
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
    randomize(model, seed=21)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    ck = save_checkpoint(model, str(tmp_path / 'w.pth'))
    cfg_file = tmp_path / 'tiny.py'
    cfg_file.write_text('model = ' + repr(cfg) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    spec = importlib.util.spec_from_file_location('cgg_tools_test3', os.path.join(root, 'tools', 'test.py'))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    a = drv.main([str(cfg_file), ck, '--num-images', '7', '--synthetic', '128', '--rle'])
    b = drv.main([str(cfg_file), ck, '--num-images', '7', '--synthetic', '128', '--no-pipeline'])
    assert len(a) == 7 and len(b) == 7
    n_det = 0
    for ra, rb in zip(a, b):
        assert set(ra) == set(rb)
        for k in ra:
            bbox_results, segm = ra[k]
            labels, bboxes, masks = rb[k]                       # plain driver: device-layout numpy (labels, boxes, bool masks)
            assert sum(len(c) for c in segm) == masks.shape[0] == sum(len(c) for c in bbox_results)
            for c in range(len(segm)):
                want = sorted((int(m.sum()), m.tobytes()) for m in masks[labels == c])
                got = sorted((int(x.sum()), x.tobytes()) for x in (rle_to_mask(r) for r in segm[c]))
                assert [w[0] for w in want] == [g[0] for g in got] and all(w[1] == g[1] for w, g in zip(want, got)), (k, c)
                n_det += len(got)
    assert n_det > 0


def test_simple_test_bitpacked_masks_equal_bool_masks(dev):
    """`mask_bits=True` (opt-in: 8x fewer mask bytes for consumers that work on packed bits): the unpacked masks are the
    bool masks, and the reference-format host results (per-class mask lists) agree with the device results."""
    import numpy as np
    from util import randomize
    cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2,
                                 dec_layers=3, vocab=500, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = registry.build_detector(cfg)
    randomize(model, seed=23)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    model = model.to(dev).eval()
    model.panoptic_fusion_head.test_cfg['max_per_image'] = 20      # k <= Q * n_classes for every type: the picks path
    B, H, W = 2, 128, 192
    metas = synthetic.img_metas(B, H, W)
    img = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad(), runtime.precision_scope('bf16'):
        a = model.simple_test(img, metas, rescale=True, device_results=True)
        b = model.simple_test(img, metas, rescale=True, device_results=True, mask_bits=True)
        host = model.simple_test(img, [dict(m) for m in metas], rescale=True)
    for ra, rb, rh in zip(a, b, host):
        for k in ra:
            assert rb[k][2].dtype == torch.uint8 and rb[k][2].shape[-1] == W // 8
            un = np.unpackbits(rb[k][2].cpu().numpy(), axis=-1, bitorder='little').astype(bool)
            assert np.array_equal(un, ra[k][2].cpu().numpy())
            assert torch.equal(ra[k][0], rb[k][0])
            bbox_results, mask_results = rh[k]
            n_host = sum(len(c) for c in mask_results)
            assert n_host == ra[k][2].shape[0]
            assert sum(int(m.sum()) for c in mask_results for m in c) == int(ra[k][2].sum())
