#!/usr/bin/env python
"""Generate the golden fixtures tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN FILES.

Runs only in the build container (needs /root/reference); the fixtures (inputs + expected outputs, no
reference source) are committed and are what travels. Usage:  python tests/golden/make_golden.py

How the reference code is made to run without mmcv / mmdet / clip (none installed, no network):
  * Tier A  -- open_set/models/losses/grounding_loss.py, transformers/{transformers,caption_tranformer}.py,
    utils/bert_embeddings.py, losses/cross_entropy_loss.py, assigners/mask_hungarian_assigner.py,
    maskformer_fusion_head.py and mask2former_head.py are imported BY FILE PATH under stub package objects
    (their package __init__ files, which import mmcv/mmdet wholesale, are never executed).
  * The mmcv / mmdet symbols those files import are provided by a shim built from the oracle's restatement
    of the [3P] leaf ops (oracle/modules.py, oracle/head.py) -- so control flow, einsum, attention-mask rule,
    loss weighting, matching and index logic in the fixtures are the REFERENCE'S, the leaf arithmetic is the
    oracle's (torch primitives).
  * Weights are not stored: both sides call tests/util.randomize(module, seed) (deterministic CPU RNG).
  * G9 -- open_set/datasets/pipelines/formatting.py (OpenFormatBundle) is executed by path with a recording stand-in for
    mmcv.parallel.DataContainer; the fixture holds the raw samples and what the reference wrapped.
  * `torch.rand` is wrapped while the reference runs so the random point coordinates it draws are captured
    into the fixture (CPU and GPU RNG streams differ; parity tests replay the captured coordinates).
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import head as OH          # noqa: E402
from oracle import modules as OM       # noqa: E402


# ------------------------------------------------------------------------------------------------
# the import shim
# ------------------------------------------------------------------------------------------------
class _Registry:
    def __init__(self, name):
        self.name, self.d = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.d[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def build(self, cfg, **kw):
        cfg = dict(cfg)
        t = cfg.pop('type')
        return self.d[t](**cfg, **kw)


HEADS, LOSSES, DETECTORS = _Registry('head'), _Registry('loss'), _Registry('detector')
BBOX_ASSIGNERS, MATCH_COST, BBOX_SAMPLERS = _Registry('assigner'), _Registry('cost'), _Registry('sampler')


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class AnchorFreeHead(BaseModule):
    pass


class MaskFormerHead(AnchorFreeHead):
    def preprocess_gt(self, gt_labels_list, gt_masks_list, gt_semantic_segs, img_metas):
        # instance case of [3P] preprocess_panoptic_gt (process_gt_open.py:38-46): masks -> long tensors
        return list(gt_labels_list), [m.long() for m in gt_masks_list]


class BasePanopticFusionHead(BaseModule):
    def __init__(self, num_things_classes=80, num_stuff_classes=53, test_cfg=None, loss_panoptic=None,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.test_cfg = OM.attrify(test_cfg or {})


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class _Sampling:
    def __init__(self, assign_result, masks, gt_masks):
        self.pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        self.neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        self.pos_assigned_gt_inds = assign_result.gt_inds[self.pos_inds] - 1


class MaskPseudoSampler:
    def __init__(self, **kw):
        pass

    def sample(self, assign_result, masks, gt_masks, **kw):
        return _Sampling(assign_result, masks, gt_masks)


class _Cost:
    def __init__(self, weight=1., **kw):
        self.weight, self.kw = weight, kw


class ClassificationCost(_Cost):
    def __call__(self, pred, labels):
        return -pred.softmax(-1)[:, labels] * self.weight


class CrossEntropyLossCost(_Cost):
    def __call__(self, pred, gt):
        return OH.match_cost(None, pred, None, gt, w_emb=0, w_mask=self.weight, w_dice=0)


class DiceCost(_Cost):
    def __call__(self, pred, gt):
        return OH.match_cost(None, pred, None, gt, w_emb=0, w_mask=0, w_dice=self.weight,
                             dice_eps=self.kw.get('eps', 1e-3))


for _c in (ClassificationCost, CrossEntropyLossCost, DiceCost):
    MATCH_COST.register_module(module=_c)
BBOX_SAMPLERS.register_module(module=MaskPseudoSampler)


class DiceLoss(nn.Module):
    def __init__(self, use_sigmoid=True, activate=True, reduction='mean', naive_dice=False, loss_weight=1.0,
                 eps=1e-3):
        super().__init__()
        self.loss_weight, self.eps = loss_weight, eps

    def forward(self, pred, target, weight=None, reduction_override=None, avg_factor=None):
        return OH.dice_loss(pred, target, avg_factor=avg_factor, eps=self.eps, loss_weight=self.loss_weight)


def multi_apply(func, *args, **kwargs):
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def get_uncertain_point_coords_with_randomness(mask_pred, labels, num_points, oversample_ratio,
                                               importance_sample_ratio):
    n = mask_pred.shape[0]
    ns = int(num_points * oversample_ratio)
    coords = torch.rand(n, ns, 2)
    unc = -torch.abs(OH.point_sample(mask_pred, coords))
    nu = int(importance_sample_ratio * num_points)
    idx = torch.topk(unc[:, 0, :], k=nu, dim=1)[1] + ns * torch.arange(n, dtype=torch.long)[:, None]
    coords = coords.view(-1, 2)[idx.view(-1), :].view(n, nu, 2)
    if num_points - nu > 0:
        coords = torch.cat((coords, torch.rand(n, num_points - nu, 2)), dim=1)
    return coords


class _FileClient:
    def get_text(self, path):
        with open(path) as f:
            return f.read()


def install_shim():
    def force_fp32(*a, **k):
        return lambda f: f

    mmcv = _mod('mmcv', FileClient=_FileClient, load=lambda p: json.load(open(p)))
    _mod('mmcv.cnn', Conv2d=nn.Conv2d, build_plugin_layer=OM.build_plugin_layer,
         caffe2_xavier_init=lambda *a, **k: None)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.transformer', build_positional_encoding=OM.build_positional_encoding,
         build_transformer_layer_sequence=OM.build_transformer_layer_sequence)
    _mod('mmcv.ops', point_sample=lambda inp, pts, **k: OH.point_sample(inp, pts), RoIPool=object)
    _mod('mmcv.runner', ModuleList=nn.ModuleList, force_fp32=force_fp32, get_dist_info=lambda: (0, 1),
         BaseModule=BaseModule)
    _mod('mmcv.parallel', collate=None, scatter=None)
    _mod('mmdet')
    _mod('mmdet.core', build_assigner=lambda cfg, **k: BBOX_ASSIGNERS.build(cfg),
         build_sampler=lambda cfg, **k: BBOX_SAMPLERS.build(cfg), multi_apply=multi_apply,
         reduce_mean=lambda t: t, INSTANCE_OFFSET=1000, bbox2result=None)
    _mod('mmdet.core.bbox')
    _mod('mmdet.core.bbox.builder', BBOX_ASSIGNERS=BBOX_ASSIGNERS)
    _mod('mmdet.core.bbox.match_costs')
    _mod('mmdet.core.bbox.match_costs.builder', build_match_cost=lambda cfg: MATCH_COST.build(cfg))
    _mod('mmdet.core.bbox.assigners')
    _mod('mmdet.core.bbox.assigners.assign_result', AssignResult=AssignResult)
    _mod('mmdet.core.bbox.assigners.base_assigner', BaseAssigner=object)
    _mod('mmdet.core.evaluation')
    _mod('mmdet.core.evaluation.panoptic_utils', INSTANCE_OFFSET=1000)
    _mod('mmdet.core.mask', mask2bbox=OH.mask2bbox)
    _mod('mmdet.models')
    _mod('mmdet.models.utils', preprocess_panoptic_gt=None,
         get_uncertain_point_coords_with_randomness=get_uncertain_point_coords_with_randomness)
    _mod('mmdet.models.builder', HEADS=HEADS, LOSSES=LOSSES, DETECTORS=DETECTORS,
         build_loss=lambda cfg: LOSSES.build(cfg), build_head=lambda cfg: HEADS.build(cfg),
         build_backbone=None, build_neck=None)
    _mod('mmdet.models.losses')
    _mod('mmdet.models.losses.utils', weight_reduce_loss=OH.weight_reduce)
    _mod('mmdet.models.dense_heads')
    _mod('mmdet.models.dense_heads.anchor_free_head', AnchorFreeHead=AnchorFreeHead)
    _mod('mmdet.models.dense_heads.maskformer_head', MaskFormerHead=MaskFormerHead)
    _mod('mmdet.models.seg_heads')
    _mod('mmdet.models.seg_heads.panoptic_fusion_heads')
    _mod('mmdet.models.seg_heads.panoptic_fusion_heads.base_panoptic_fusion_head',
         BasePanopticFusionHead=BasePanopticFusionHead)
    _mod('mmdet.datasets', replace_ImageToTensor=None)
    _mod('mmdet.datasets.pipelines', Compose=None)
    _mod('clip', load=None)
    # reference packages as empty shells whose __path__ points at the reference tree
    for pkg in ('open_set', 'open_set.models', 'open_set.models.utils', 'open_set.models.losses',
                'open_set.models.transformers', 'open_set.utils', 'open_set.utils.eval', 'open_set.assigners'):
        m = _mod(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split('.'))]
    return mmcv


def ref_import(name):
    return importlib.import_module(name)


class StubBert:
    """what BertEmbeddings(bert_model) reads: .config and .embeddings.{word_embeddings,LayerNorm}."""

    def __init__(self, vocab=30522, hidden=768):
        self.config = types.SimpleNamespace(vocab_size=vocab, hidden_size=hidden, pad_token_id=0,
                                            layer_norm_eps=1e-12)
        self.embeddings = types.SimpleNamespace(
            word_embeddings=nn.Embedding(vocab, hidden, padding_idx=0), LayerNorm=nn.LayerNorm(hidden, eps=1e-12))

    def eval(self):
        return self


class RandCapture:
    """records every torch.rand() result drawn while active (kind decided by the shape)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = torch.rand

        def rec(*a, **k):
            k.pop('device', None)
            out = self._orig(*a, **k)
            self.draws.append(out.clone())
            return out
        torch.rand = rec
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f'wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB  ({", ".join(out)})')


# ------------------------------------------------------------------------------------------------
def main():
    from util import g4_inputs, g6_inputs, g7_inputs, head_cfg, randomize
    import transformers
    install_shim()
    torch.set_num_threads(4)

    # ---- G1: grounding loss (tier A, verbatim) ----
    gl = ref_import('open_set.models.losses.grounding_loss')
    g = torch.Generator().manual_seed(100)
    out = {}
    for B in (1, 2, 4):
        preds = torch.randn(B, 10, 64, generator=g)
        embs = torch.randn(B, 7, 64, generator=g)
        mask = (torch.rand(B, 7, generator=g) < 0.6).long()
        mask[:, 0] = 1
        if B == 4:
            mask[2] = 0                                   # a caption with zero object nouns
        out[f'preds{B}'], out[f'embs{B}'], out[f'mask{B}'] = preds, embs, mask
        out[f'loss{B}'] = gl.grounding_loss(preds, embs, mask, 10.0)
    npz('g1_grounding_loss.npz', **out)

    # ---- G2: caption transformer + bert embeddings (tier A, verbatim) ----
    ct = ref_import('open_set.models.transformers.caption_tranformer')
    cap_cfg = dict(nb_layers=2, input_dim=64, hidden_dim=64, ff_dim=32, nb_heads=4, drop_val=0.1,
                   pre_norm=False, seq_length=12, nb_tokens=50)
    model = ct.CaptionTransformer(**cap_cfg).eval()
    randomize(model, seed=7)
    g = torch.Generator().manual_seed(101)
    tgt = torch.randn(3, 11, 64, generator=g)
    mem = torch.randn(3, 9, 64, generator=g)
    kpm = torch.zeros(3, 11, dtype=torch.bool)
    kpm[0, 7:] = True
    kpm[2, 4:] = True
    with torch.no_grad():
        outs, logits = model(tgt=tgt, memory=mem, tgt_key_padding_mask=kpm)
    be = ref_import('open_set.models.utils.bert_embeddings')
    sb = StubBert(100, 32)
    randomize(sb.embeddings.word_embeddings, seed=8)
    randomize(sb.embeddings.LayerNorm, seed=9)
    bemb = be.BertEmbeddings(sb)
    ids = torch.tensor([[5, 0, 99, 17], [1, 2, 3, 0]])
    with torch.no_grad():
        bout = bemb.LayerNorm(bemb.word_embeddings(ids))
    npz('g2_caption_transformer.npz', cfg=json.dumps(cap_cfg), seed=7, tgt=tgt, mem=mem, kpm=kpm, logits=logits,
        last=outs[-1], first=outs[0], psne=model.position_encoder.psne_layer, bert_ids=ids, bert_out=bout,
        bert_table=sb.embeddings.word_embeddings.weight, bert_ln_w=sb.embeddings.LayerNorm.weight,
        bert_ln_b=sb.embeddings.LayerNorm.bias)

    # ---- reference losses / assigner registered under the names the configs use ----
    cel = ref_import('open_set.models.losses.cross_entropy_loss')          # registers CrossEntropyLossOpen
    LOSSES.register_module(name='CrossEntropyLoss', module=cel.CrossEntropyLossOpen)
    LOSSES.register_module(name='DiceLoss', module=DiceLoss)
    ref_import('open_set.assigners.mask_hungarian_assigner')               # registers MaskHungarianAssignerOpen
    transformers.BertModel.from_pretrained = staticmethod(lambda *a, **k: StubBert())
    m2f = ref_import('open_set.models.mask2former_head')
    fus = ref_import('open_set.models.maskformer_fusion_head')

    # ---- G3/G4: the reference head: forward (all decoder layers) ----
    cfg, B, H, W, feats, metas, qf, mf = g4_inputs()
    hc = OM.attrify(head_cfg(cfg))
    ref_head = m2f.Mask2FormerHeadOpen(**hc).eval()
    randomize(ref_head, seed=0)
    with torch.no_grad():
        cls_l, emb_l, mask_l = ref_head.forward(feats, metas)
        # G3: one forward_head call in isolation (einsum + attention-mask rule)
        c3, e3, m3, a3 = ref_head.forward_head(qf, mf, (4, 6))
    # inputs are regenerated by tests/util.g4_inputs(); only expected outputs are stored
    npz('g4_head_forward.npz', seed=0, cls=torch.stack(cls_l), emb=torch.stack(emb_l), mask=torch.stack(mask_l),
        fh_cls=c3, fh_emb=e3, fh_mask=m3, fh_attn=a3)

    # ---- G5/G6: targets + losses of one decoder layer, with captured random points ----
    ref_head.train()
    for mod in ref_head.modules():            # dropout off (caption generator drop_val=0.1) -> deterministic
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask = g6_inputs(H, W)
    torch.manual_seed(1234)
    with RandCapture() as rc:
        cap_embs, _ = ref_head.extract_word_embeddings(cap_ids, cap_mask, 'bert')
        noun_embs, _ = ref_head.extract_word_embeddings(noun_ids, noun_mask, 'bert')
        li = 2
        losses = ref_head.loss_single(cls_l[li], emb_l[li], mask_l[li], gt_labels, gt_masks, cap_ids, cap_embs,
                                      cap_mask, noun_ids, noun_embs, noun_mask, metas)
    draws = rc.draws
    with RandCapture() as rc2:
        torch.manual_seed(99)
        emb_logit = ref_head._get_cls_emb_logits(emb_l[li])
        tgt_single = ref_head._get_target_single(cls_l[li][0], emb_logit[0], mask_l[li][0], gt_labels[0],
                                                 gt_masks[0], metas)
    npz('g6_loss_single.npz', layer=li, losses=torch.stack([x.detach().reshape(()) for x in losses]),
        **{f'draw{i}': d for i, d in enumerate(draws)}, n_draws=len(draws),
        t_points=rc2.draws[0], t_labels=tgt_single[0], t_mask_weights=tgt_single[3], t_pos=tgt_single[4],
        t_neg=tgt_single[5])

    # ---- G11: the head's non-default caption-target / loss-assembly branches no shipped config sets (VERDICT r3 missing 6):
    #      gen_only / gen_mask / gen_replace_obj_nouns (mask2former_head.py:562-580: the in-place edit of the caption ids the
    #      generator is trained on), learnable_temperature (:228), loss_only_last (:448), freeze_v2l (:242-244). The reference's
    #      own `loss_single` / `loss` / `init_weights` run with each flag; stored: the edited ids, the caption-generation loss,
    #      the loss-dict keys, the temperature parameter and the requires_grad pattern ----
    g11 = {}
    for flag in ('gen_only_obj_nouns', 'gen_mask_obj_nouns', 'gen_replace_obj_nouns'):
        hf = OM.attrify(dict(head_cfg(cfg), **{flag: True}))
        rh = m2f.Mask2FormerHeadOpen(**hf).train()
        rh.load_state_dict(ref_head.state_dict())
        for mod in rh.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
        ids = [t.clone() for t in cap_ids]
        torch.manual_seed(1234)
        ce, _ = rh.extract_word_embeddings(ids, cap_mask, 'bert')
        ne, _ = rh.extract_word_embeddings(noun_ids, noun_mask, 'bert')
        try:
            ls = rh.loss_single(cls_l[li], emb_l[li], mask_l[li], gt_labels, gt_masks, ids, ce, cap_mask, noun_ids, ne, noun_mask, metas)
            g11[f'{flag}_loss'] = ls[3].detach().reshape(())
        except IndexError:
            # gen_replace writes token 4874 ('object' in BERT's vocabulary); the fixture's toy generator has fewer tokens, so the
            # reference's cross-entropy refuses the target -- the in-place edit (what is pinned here) has already happened
            g11[f'{flag}_loss'] = torch.tensor(float('nan'))
        g11[f'{flag}_ids'] = torch.stack(ids)                 # edited IN PLACE by the reference
    hf = OM.attrify(dict(head_cfg(cfg), learnable_temperature=True, softmax_temperature=7.0, loss_only_last=True, freeze_v2l=True))
    rh = m2f.Mask2FormerHeadOpen(**hf).train()
    rh.init_weights()
    g11['frozen'] = json.dumps(sorted(n for n, p_ in rh.named_parameters() if not p_.requires_grad and not n.startswith('bert')
                                      and 'class_embs' not in n))
    g11['temperature_is_param'] = int(isinstance(rh.softmax_temperature, nn.Parameter) and rh.softmax_temperature.requires_grad)
    g11['temperature'] = rh.softmax_temperature.detach().reshape(-1)
    rh.load_state_dict(dict(ref_head.state_dict(), softmax_temperature=rh.softmax_temperature.detach()))
    for mod in rh.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    with torch.no_grad():
        g11['temp_logits'] = rh._get_cls_emb_logits(emb_l[li])
    torch.manual_seed(1234)
    ids = [t.clone() for t in cap_ids]
    ce, _ = rh.extract_word_embeddings(ids, cap_mask, 'bert')
    ne, _ = rh.extract_word_embeddings(noun_ids, noun_mask, 'bert')
    ld = rh.loss(cls_l, emb_l, mask_l, gt_labels, gt_masks, ids, ce, cap_mask, noun_ids, ne, noun_mask, metas)
    g11['last_only_keys'] = json.dumps(sorted(ld.keys()))
    npz('g11_head_flags.npz', layer=li, **g11)

    # ---- G7: fusion head post-processing (instance + panoptic) ----
    fcfg = dict(cfg['panoptic_fusion_head'])
    fcfg.pop('type')
    fh = fus.MaskFormerFusionHeadOpen(test_cfg=cfg['test_cfg'], **fcfg)
    emb, mp, cls_embs, pemb = g7_inputs()
    fh.test_cfg['max_per_image'] = 20
    with torch.no_grad():
        lab, box, msk = fh.instance_postprocess_emb(emb, mp, fh.all_class_embs)
        labn, boxn, mskn = fh.instance_postprocess_emb(emb, mp, fh.novel_class_embs)
    pcfg = dict(num_things_classes=8, num_stuff_classes=4, panoptic_mode=True,
                test_cfg=dict(object_mask_thr=0.2, iou_thr=0.5, filter_low_score=True, stuff_area_limit=16))
    ph = fus.MaskFormerFusionHeadOpen(**pcfg)
    with torch.no_grad():
        pan = ph.panoptic_postprocess_emb(pemb, mp, cls_embs)
    npz('g7_postprocess.npz', labels=lab, bboxes=box, masks=msk, labels_novel=labn, bboxes_novel=boxn, masks_novel=mskn,
        pan_seg=pan)

    # ---- G10: the no-class-embedding family (configs/instance/coco_ag_pretrain_3x.py:97-133): the head with
    #      use_class_emb=False / pred_emb_norm=True (forward, targets with cls_cost 2.0, loss_cls with weight 2.0 and
    #      class_weight) and the closed-set post-processing of the fusion head (maskformer_fusion_head.py:161-295) ----
    from util import ag_cfg, g10_inputs
    cfgA = ag_cfg(num_queries=8, vocab=120)
    hcA = OM.attrify(head_cfg(cfgA))
    ref_ag = m2f.Mask2FormerHeadOpen(**hcA).eval()
    randomize(ref_ag, seed=3)
    with torch.no_grad():
        cls_a, emb_a, mask_a = ref_ag.forward(feats, metas)
    ref_ag.train()
    torch.manual_seed(4321)
    with RandCapture() as rca:
        li = 1
        losses_a = ref_ag.loss_single(cls_a[li], emb_a[li], mask_a[li], gt_labels, gt_masks, None, None, None, None, None, None,
                                      metas)
    with RandCapture() as rca2:
        torch.manual_seed(98)
        tgt_a = ref_ag._get_target_single(cls_a[li][1], None, mask_a[li][1], gt_labels[1], gt_masks[1], metas)
    fa = dict(cfgA['panoptic_fusion_head'])
    fa.pop('type')
    fha = fus.MaskFormerFusionHeadOpen(test_cfg=dict(cfgA['test_cfg'], max_per_image=15), **fa)
    mcls, mpred = g10_inputs()
    with torch.no_grad():
        labc, boxc, mskc = fha.instance_postprocess(mcls[:, :fha.num_classes + 1], mpred)
    pha = fus.MaskFormerFusionHeadOpen(num_things_classes=8, num_stuff_classes=4, panoptic_mode=True,
                                       test_cfg=dict(object_mask_thr=0.3, iou_thr=0.5, filter_low_score=True))
    with torch.no_grad():
        panc = pha.panoptic_postprocess(mcls, mpred)
    npz('g10_no_class_emb.npz', seed=3, layer=li, cls=torch.stack(cls_a), emb=torch.stack(emb_a), mask=torch.stack(mask_a),
        losses=torch.stack([x.detach().reshape(()) for x in losses_a]), n_draws=len(rca.draws),
        **{f'draw{i}': d for i, d in enumerate(rca.draws)}, t_points=rca2.draws[0], t_labels=tgt_a[0], t_mask_weights=tgt_a[3],
        t_pos=tgt_a[4], t_neg=tgt_a[5], ins_labels=labc, ins_bboxes=boxc, ins_masks=mskc, pan_seg=panc,
        ins_num_classes=fha.num_classes)

    g8_beam_search()


class _StubTokenizer:
    """stands in for BertTokenizer (no vocabulary file offline): ids -> 'i0 i1 ...'."""

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()

    def decode(self, ids):
        return ' '.join(str(int(i)) for i in ids)


def g8_beam_search():
    """G8: the reference's beam_search (open_set/utils/eval/inference.py) over the reference's CaptionTransformer."""
    import transformers
    from util import randomize
    ct = ref_import('open_set.models.transformers.caption_tranformer')
    inf = ref_import('open_set.utils.eval.inference')
    transformers.BertTokenizer = _StubTokenizer
    cap_cfg = dict(nb_layers=2, input_dim=32, hidden_dim=32, ff_dim=32, nb_heads=4, drop_val=0.1, pre_norm=False,
                   seq_length=12, nb_tokens=30)
    out = dict(cfg=json.dumps(cap_cfg))
    case = 0
    # NB the reference crashes when exactly ONE live sequence remains (`get_ids_embedding` squeezes the batch
    # dimension away, inference.py:80); such seeds are skipped -- the fixture holds runs the reference completes
    for seed in range(21, 80):
        if case == 5:
            break
        beam, max_len = [(3, 8), (4, 10), (2, 6), (5, 12), (3, 10)][case]
        gen = ct.CaptionTransformer(**cap_cfg).eval()
        randomize(gen, seed=seed)
        with torch.no_grad():
            gen.generator.bias[2] += 1.0 + 0.5 * case        # make EOS reachable so that sequences finish
        sb = StubBert(30, 32)
        randomize(sb.embeddings.word_embeddings, seed=seed + 100)
        randomize(sb.embeddings.LayerNorm, seed=seed + 200)
        be = ref_import('open_set.models.utils.bert_embeddings').BertEmbeddings(sb)
        model = types.SimpleNamespace(bert_embeddings=be, caption_generator=gen)
        mem = torch.randn(1, 7, 32, generator=torch.Generator().manual_seed(seed + 300))
        try:
            with torch.no_grad():
                sent = inf.beam_search(model, mem, 1, 2, max_len=max_len, beam_width=beam)
        except ValueError:
            continue
        if not sent:
            continue
        out[f'mem{case}'] = mem
        out[f'params{case}'] = np.array([seed, beam, max_len, case])
        out[f'sentence{case}'] = np.array(sent)
        print('G8 case', case, 'seed', seed, repr(sent))
        case += 1
    out['n_cases'] = np.array(case)
    npz('g8_beam_search.npz', **out)


class _RecDC:
    """stand-in for mmcv.parallel.DataContainer: records what the reference asks for."""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self.data, self.stack, self.padding_value, self.cpu_only, self.pad_dims = data, stack, padding_value, cpu_only, pad_dims


def g9_samples():
    """seeded raw samples (what the pipeline hands to the bundle); shared with tests/test_host_logic.py."""
    out = []
    for case, (H, W, n, gray, seg, caps) in enumerate([(11, 13, 3, False, True, True), (8, 9, 0, False, False, True),
                                                       (7, 5, 2, True, True, False)]):
        rng = np.random.RandomState(900 + case)
        img = rng.randint(0, 256, (H, W) if gray else (H, W, 3)).astype(np.uint8)
        if case == 1:
            img = rng.rand(H, W, 3).astype(np.float32)            # already float: no cast
        r = dict(img=img, filename=f'im{case}.jpg', ori_shape=img.shape, img_shape=img.shape, flip=False,
                 gt_bboxes=(rng.rand(n, 4) * 10).astype(np.float32), gt_labels=rng.randint(0, 48, (n,)).astype(np.int64),
                 gt_masks=rng.randint(0, 2, (n, H, W)).astype(np.uint8))
        if seg:
            r['gt_semantic_seg'] = rng.randint(0, 134, (H, W)).astype(np.uint8)
        if caps:
            r['gt_caption_ids'] = rng.randint(0, 30522, (35,)).astype(np.int64)
            r['gt_caption_mask'] = (rng.rand(35) < 0.5).astype(np.int64)
            r['gt_caption_nouns_ids'] = [int(v) for v in rng.randint(0, 30522, (35,))]      # a python list
            r['gt_caption_nouns_mask'] = (rng.rand(35) < 0.3).astype(np.int64)
        if case == 2:
            r['pad_shape'] = (16, 16)                               # a pipeline that already padded: default not applied
            r['gt_bboxes_ignore'] = np.zeros((0, 4), dtype=np.float32)
        out.append(r)
    return out


def g9_format_bundle():
    """G9: the reference's OpenFormatBundle (open_set/datasets/pipelines/formatting.py) on seeded samples."""
    import importlib.util
    mmcv = sys.modules['mmcv']
    mmcv.is_str = lambda x: isinstance(x, str)
    sys.modules['mmcv.parallel'].DataContainer = _RecDC
    _mod('mmdet.datasets.builder', PIPELINES=_Registry('pipeline'))
    spec = importlib.util.spec_from_file_location('ref_formatting', os.path.join(REF, 'open_set/datasets/pipelines/formatting.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for case, raw in enumerate(g9_samples()):
        # inputs travel with the fixture
        meta = {}
        for k, v in raw.items():
            if isinstance(v, np.ndarray):
                out[f'c{case}_raw_{k}'] = v
            else:
                meta[k] = v
        out[f'c{case}_raw_meta'] = np.array(json.dumps(meta))
        res = ref.OpenFormatBundle()(dict(raw))
        keys = []
        for k, v in res.items():
            if isinstance(v, _RecDC):
                keys.append(k)
                d = v.data
                out[f'c{case}_{k}_data'] = d.numpy() if torch.is_tensor(d) else np.asarray(d)
                out[f'c{case}_{k}_dtype'] = np.array(str(d.dtype) if torch.is_tensor(d) else 'object')
                out[f'c{case}_{k}_attrs'] = np.array([int(v.stack), int(v.padding_value), int(v.cpu_only), int(v.pad_dims)])
        out[f'c{case}_keys'] = np.array(keys)
        out[f'c{case}_pad_shape'] = np.array(res['pad_shape'])
        out[f'c{case}_scale_factor'] = np.array(res['scale_factor'])
        out[f'c{case}_norm_mean'] = res['img_norm_cfg']['mean']
        out[f'c{case}_norm_std'] = res['img_norm_cfg']['std']
        print('G9 case', case, keys)
    out['n_cases'] = np.array(3)
    npz('g9_format_bundle.npz', **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'g8':
        install_shim()
        g8_beam_search()
    elif len(sys.argv) > 1 and sys.argv[1] == 'g9':
        install_shim()
        g9_format_bundle()
    else:
        main()
