"""One configuration of tests/test_env_switches_gpu.py::test_training_switch, in a FRESH process (the switches are read at import): a
parity-mode forward_train + backward of the head, large enough that every training node is taken (>= 8192 encoder rows, >= 65536
FPN pixels: batch 4 at 512 x 512), against the float32 oracle's losses and gradients on the same weights and inputs (attention masks
and Hungarian solutions pinned tie-aware). The oracle run is computed by the first worker and cached in the file argv[1]."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

KEYS = ['pixel_decoder.encoder.layers.0.attentions.0.sampling_offsets.weight', 'pixel_decoder.encoder.layers.0.attentions.0.value_proj.weight',
        'pixel_decoder.encoder.layers.1.attentions.0.output_proj.weight', 'pixel_decoder.encoder.layers.0.ffns.0.layers.0.0.weight',
        'pixel_decoder.encoder.layers.1.norms.1.weight', 'pixel_decoder.input_convs.0.conv.weight',
        'pixel_decoder.lateral_convs.0.conv.weight', 'pixel_decoder.output_convs.0.conv.weight', 'pixel_decoder.mask_feature.weight',
        'pixel_decoder.level_encoding.weight', 'transformer_decoder.layers.0.attentions.0.attn.in_proj_weight',
        'transformer_decoder.layers.2.attentions.1.attn.in_proj_weight', 'transformer_decoder.layers.1.ffns.0.layers.1.weight',
        'mask_embed.0.weight', 'v2l_transform.weight', 'query_feat.weight', 'query_embed.weight', 'caption_generator.generator.weight']


def main():
    cache = sys.argv[1]
    import cgg_amd  # noqa: F401
    from cgg_amd import runtime, synthetic
    from util import AssignTeacher, Bank, MaskTeacher, build_heads, small_cfg
    dev = torch.device('cuda', 0)
    cfg = small_cfg(num_queries=40, num_points=2048, depth=50, enc_layers=2, dec_layers=3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prod, orc = build_heads(cfg, seed=5)
    prod = prod.to(dev).train()
    orc.train()
    for m in list(prod.modules()) + list(orc.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    B, H, W = 4, 512, 512
    feats = synthetic.backbone_feats(B, H, W, channels=(256, 512, 1024, 2048), seed=41)
    metas = synthetic.img_metas(B, H, W)
    batch = synthetic.train_batch(B, H, W, num_classes=cfg['panoptic_head']['num_things_classes'], max_inst=6, vocab=500, seed=42)
    teacher = MaskTeacher(orc)
    matcher = AssignTeacher([b for b in range(B) if len(batch['gt_labels'][b])])
    if os.path.exists(cache):
        c = torch.load(cache, weights_only=False)
        olosses, ograds, ofg, teacher.logits, matcher.recorded = c['losses'], c['grads'], c['fgrads'], c['logits'], c['assign']
    else:
        orc.point_hook = Bank(9)
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        ofeats = [f.clone().requires_grad_(True) for f in feats]
        oc, oe, om = teacher.run_oracle(lambda: orc.forward(ofeats, metas))
        with matcher.record():
            ol = orc.loss(oc, oe, om, batch['gt_labels'], [m.long() for m in batch['gt_masks']], batch['gt_caption_ids'],
                          batch['gt_caption_mask'], batch['gt_caption_nouns_ids'], batch['gt_caption_nouns_mask'])
        sum(ol.values()).backward()
        olosses = {k: float(v) for k, v in ol.items()}
        ograds = {k: p.grad.clone() for k, p in orc.named_parameters() if k in KEYS}
        ofg = [f.grad.clone() for f in ofeats]
        tmp = cache + '.%d' % os.getpid()
        torch.save(dict(losses=olosses, grads=ograds, fgrads=ofg, logits=teacher.logits, assign=matcher.recorded), tmp)
        os.replace(tmp, cache)
    prod.point_hook = Bank(9)
    prod.attn_mask_hook, prod.assign_hook = teacher.hook, matcher.hook
    to = lambda lst: [t.to(dev) for t in lst]   # noqa: E731
    pfeats = [f.to(dev).requires_grad_(True) for f in feats]
    with runtime.precision_scope('fp32'):
        losses = prod.forward_train(pfeats, metas, to(batch['gt_bboxes']), to(batch['gt_labels']), to(batch['gt_masks']), None,
                                    to(batch['gt_caption_ids']), to(batch['gt_caption_mask']), to(batch['gt_caption_nouns_ids']),
                                    to(batch['gt_caption_nouns_mask']))
        sum(losses.values()).backward()
    teacher.check()
    matcher.check(len(prod.transformer_decoder.layers) + 1)
    assert set(losses) == set(olosses)
    for k in sorted(losses):
        a, b = float(losses[k]), olosses[k]
        assert abs(a - b) <= 2e-3 * (1 + abs(b)), (k, a, b)
    named = dict(prod.named_parameters())
    worst = ('', 0.0)
    for k in KEYS:
        g, og = named[k].grad, ograds[k]
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, k
        err = float((g.cpu() - og).abs().max()) / max(float(og.abs().max()), 1e-30)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err <= 2e-2, (k, err)
    for i, (pf, og) in enumerate(zip(pfeats, ofg)):
        err = float((pf.grad.cpu() - og).abs().max()) / max(float(og.abs().max()), 1e-30)
        worst = max(worst, ('feat%d' % i, err), key=lambda t: t[1])
        assert err <= 5e-2, (i, err)
    print('env switch train worker OK: %d losses within 2e-3, %d gradients, worst %.2e (%s);' % (len(losses), len(KEYS) + 4, worst[1], worst[0]),
          ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('CGG_')), flush=True)


if __name__ == '__main__':
    main()
