"""-m gpu: the no-class-embedding / class-agnostic family (configs/instance/coco_ag_pretrain_3x.py:97-133) on the HIP path --
`use_class_emb=False, pred_emb_norm=True`, `loss_cls` weight 2.0 with the classification cost in the assigner, no caption heads,
closed-set fusion head (`ins_results`, maskformer_fusion_head.py:161-295). Checked against the G10 fixture (outputs of the
reference's own head / fusion head, tests/golden/make_golden.py) and against the oracle on the same seeded inputs."""
import os
import warnings

import numpy as np
import pytest
import torch

import cgg_amd  # noqa: F401
from cgg_amd import registry, synthetic
from oracle import head as OH

from test_train_gpu import Bank
from util import MaskTeacher, ag_cfg, build_heads, g4_inputs, g6_inputs, g10_inputs, head_cfg, randomize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
pytestmark = pytest.mark.gpu


def gold(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) if z[k].shape != () else z[k] for k in z.files}


class Replay:
    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def __call__(self, kind, shape, device):
        d = self.draws[self.i]
        self.i += 1
        assert tuple(d.shape) == tuple(shape), (kind, tuple(d.shape), tuple(shape))
        return d.to(device)


def test_g10_head_forward_losses_targets_vs_fixture(dev):
    z = gold('g10_no_class_emb.npz')
    cfg = ag_cfg(num_queries=8, vocab=120)
    _, B, H, W, feats, metas, _, _ = g4_inputs()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prod = registry.build_head(head_cfg(cfg))
        orc = OH.OracleHead(**head_cfg(cfg)).eval()
    randomize(prod, seed=3)
    randomize(orc, seed=3)
    prod = prod.to(dev).eval()
    teacher = MaskTeacher(orc)
    with torch.no_grad():
        teacher.run_oracle(lambda: orc.forward(feats, metas))
        prod.attn_mask_hook = teacher.hook
        try:
            pc, pe, pm = prod.forward([f.to(dev) for f in feats], metas)
        finally:
            prod.attn_mask_hook = None
    teacher.check()
    assert (torch.stack(pc).cpu() - z['cls']).abs().max().item() <= 1e-3
    assert (torch.stack(pe).cpu() - z['emb']).abs().max().item() <= 1e-3          # raw embeddings: pred_emb_norm acts in the logits
    assert (torch.stack(pm).cpu() - z['mask']).abs().max().item() <= 1e-3         # north_star: mask logits within 1e-3
    # losses from the FIXTURE's predictions with the reference's captured point draws, on the device
    li = int(z['layer'])
    cls, emb, mask = z['cls'][li].to(dev), z['emb'][li].to(dev), z['mask'][li].to(dev)
    gt_labels, gt_masks = g6_inputs(H, W)[:2]
    gl, gm = [t.to(dev) for t in gt_labels], [t.to(dev) for t in gt_masks]
    prod.train()
    prod.point_hook = Replay([z[f'draw{i}'] for i in range(int(z['n_draws']))])
    with torch.no_grad():
        got = prod.loss_single(cls, emb, mask, gl, gm, None, None, None, None, None, None, metas)
    got = torch.stack([g.reshape(()).cpu() for g in got])
    want = z['losses']
    assert (got - want).abs().max().item() <= 1e-4 * (1 + want.abs().max().item()), (got, want)
    prod.point_hook = Replay([z['t_points']])
    t = prod._get_target_single(cls[1], None, mask[1], gl[1], gm[1], metas)
    assert torch.equal(t[0].cpu(), z['t_labels']) and torch.equal(t[4].cpu(), z['t_pos']) and torch.equal(t[5].cpu(), z['t_neg'])


def test_g10_closed_set_postprocess_vs_fixture(dev):
    z = gold('g10_no_class_emb.npz')
    cfg = ag_cfg(num_queries=8, vocab=120)
    mcls, mpred = g10_inputs()
    fcfg = dict(cfg['panoptic_fusion_head'])
    fcfg.update(test_cfg=dict(cfg['test_cfg'], max_per_image=15))
    fusion = registry.build_head(fcfg).to(dev)
    nc = int(z['ins_num_classes'])
    assert fusion.num_classes == nc
    lab, box, msk = fusion.instance_postprocess(mcls[:, :nc + 1].to(dev), mpred.to(dev))
    lab, box, msk = lab.cpu(), box.cpu(), msk.cpu()
    o1 = torch.argsort(box[:, 4].double() * 1e3 + lab, stable=True)
    o2 = torch.argsort(z['ins_bboxes'][:, 4].double() * 1e3 + z['ins_labels'], stable=True)
    assert torch.equal(lab[o1], z['ins_labels'][o2])
    assert torch.equal(box[o1][:, :4], z['ins_bboxes'][o2][:, :4])
    assert (box[o1][:, 4] - z['ins_bboxes'][o2][:, 4]).abs().max().item() <= 1e-5
    assert torch.equal(msk[o1].bool(), z['ins_masks'][o2].bool())
    pfus = registry.build_head(dict(type='MaskFormerFusionHeadOpen', num_things_classes=8, num_stuff_classes=4,
                                    panoptic_mode=True, use_class_emb=False,
                                    test_cfg=dict(object_mask_thr=0.3, iou_thr=0.5, filter_low_score=True))).to(dev)
    pan = pfus.panoptic_postprocess(mcls.to(dev), mpred.to(dev)).cpu()
    assert pan.dtype == torch.int32 and torch.equal(pan, z['pan_seg'].to(torch.int32))


def test_g10_family_training_step_and_ins_results_vs_oracle(dev):
    """one training step (loss_cls live with weight 2.0 and the classification cost in the matcher -> non-zero `cls_embed`
    gradient, no caption / grounding terms) and the `ins_results` inference branch end to end against the oracle."""
    cfg = ag_cfg(num_queries=12, num_points=512)
    prod, orc = build_heads(cfg)
    prod = prod.to(dev).train()
    orc.train()
    for m in list(prod.modules()) + list(orc.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    B, H, W = 2, 128, 160
    feats = synthetic.backbone_feats(B, H, W, channels=(64, 128, 256, 512), seed=31)
    metas = synthetic.img_metas(B, H, W)
    batch = synthetic.train_batch(B, H, W, num_classes=cfg['panoptic_head']['num_things_classes'], max_inst=5, vocab=500, seed=32)
    teacher = MaskTeacher(orc)
    orc.point_hook = Bank(9)
    oc, oe, om = teacher.run_oracle(lambda: orc.forward(feats, metas))
    olosses = orc.loss(oc, oe, om, batch['gt_labels'], [m.long() for m in batch['gt_masks']], None, None, None, None)
    sum(olosses.values()).backward()
    ograds = {k: (None if p.grad is None else p.grad.clone()) for k, p in orc.named_parameters()}
    prod.point_hook = Bank(9)
    prod.attn_mask_hook = teacher.hook
    to = lambda lst: [t.to(dev) for t in lst]   # noqa: E731
    losses = prod.forward_train([f.to(dev) for f in feats], metas, to(batch['gt_bboxes']), to(batch['gt_labels']),
                                to(batch['gt_masks']), None, None, None, None, None)
    prod.attn_mask_hook = None
    teacher.check()
    assert set(losses) == set(olosses)
    for k in sorted(losses):
        a, b = float(losses[k]), float(olosses[k])
        assert abs(a - b) <= 2e-3 * (1 + abs(b)), (k, a, b)
    assert float(olosses['loss_cls']) > 0
    sum(losses.values()).backward()
    named = dict(prod.named_parameters())
    for key in ['cls_embed.weight', 'cls_embed.bias', 'mask_embed.0.weight', 'query_feat.weight',
                'transformer_decoder.layers.1.attentions.0.attn.in_proj_weight',
                'pixel_decoder.encoder.layers.0.attentions.0.sampling_offsets.weight', 'pixel_decoder.mask_feature.weight']:
        g, og = named[key].grad, ograds[key]
        assert g is not None and og is not None and og.abs().max().item() > 0, key
        scale = og.abs().max().item()
        # 2e-3 of the gradient's scale: the sampling-offset gradient is a difference of neighbouring bilinear taps
        assert (g.cpu() - og).abs().max().item() <= 2e-3 * scale + 1e-7, (key, (g.cpu() - og).abs().max().item(), scale)
    # ---- inference: ins_results ----
    prod.eval()
    orc.eval()
    fcfg = dict(cfg['panoptic_fusion_head'])
    fcfg.update(test_cfg=dict(cfg['test_cfg'], max_per_image=20))
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
        res = fusion.simple_test(pcls, pemb, pmasks, metas, rescale=True)
    nc = fusion.num_classes
    assert ocls.shape[-1] == nc + 1                    # :5-9 of the config: nothing held out, head and fusion head agree
    for b in range(B):
        assert set(res[b]) == {'ins_results'}
        omp = OH.crop_rescale(oup[b], metas[b], True)
        olab, obox, omask = OH.instance_postprocess(ocls[b], omp, nc, fusion.num_things_classes, 20)
        plab, pbox, pmask = [t.cpu() for t in res[b]['ins_results']]
        assert sorted(plab.tolist()) == sorted(olab.tolist())
        o1 = torch.argsort(pbox[:, 4].double() * 1e3 + plab, stable=True)
        o2 = torch.argsort(obox[:, 4].double() * 1e3 + olab, stable=True)
        assert torch.equal(plab[o1], olab[o2])
        assert (pbox[o1][:, 4] - obox[o2][:, 4]).abs().max().item() <= 1e-3
        assert (pmask[o1].bool() != omask[o2]).flatten(1).sum(1).max().item() <= 8
