"""`MaskFormerFusionHeadOpen` -- open-vocabulary post-processing of the reference
(open_set/models/maskformer_fusion_head.py:15-464) on the HIP inference-tail kernels.

Class scores are `softmax(emb @ class_embs^T)` WITHOUT temperature (:312-313).
Differences in execution, not in results:
  * masks arrive as low-resolution logits + resize geometry (`LowResMasks`); "upsample to
    batch_input_shape -> crop img_shape -> (rescale to ori_shape)" is evaluated on the fly inside
    `cgg_instance_masks` / `cgg_panoptic_argmax`, so the (Q, H_img, W_img) f32 tensor is never stored;
  * the panoptic segment loop (:122-157) needs three areas per kept query; they come back from the
    device in ONE copy instead of three `.item()` syncs per query, and the decisions are painted by a
    LUT kernel.
"""
import json

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import to_config_dict
from .mask2former_head import LowResMasks
from .registry import HEADS

INSTANCE_OFFSET = 1000  # [3P] mmdet.core.evaluation.panoptic_utils.INSTANCE_OFFSET


def _read_lines(path):
    with open(path, 'r', encoding='utf-8') as f:
        return f.read().split('\n')


@HEADS.register_module()
class MaskFormerFusionHeadOpen(nn.Module):

    def __init__(self, num_things_classes=65, num_stuff_classes=0, panoptic_mode=False, test_cfg=None,
                 loss_panoptic=None, init_cfg=None, **kwargs):
        super().__init__()
        # [3P] BasePanopticFusionHead
        self.num_things_classes = num_things_classes
        self.num_stuff_classes = num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.test_cfg = to_config_dict(test_cfg) if test_cfg is not None else to_config_dict({})
        self.loss_panoptic = None
        if loss_panoptic:
            raise NotImplementedError('MaskFormerFusionHeadOpen has no training loss (forward_train -> {})')
        self.kwargs = kwargs
        self.panoptic_mode = panoptic_mode
        self.use_class_emb = kwargs.get('use_class_emb', False)
        self.known_file = kwargs.get('known_file', None)
        self.unknown_file = kwargs.get('unknown_file', None)
        if self.known_file is not None:
            self.known_cat_names = _read_lines(self.known_file)
        self.unknown_cat_names = _read_lines(self.unknown_file) if self.unknown_file is not None else []
        if self.use_class_emb:
            with open(kwargs['class_to_emb_file'], 'r') as f:
                class_to_emb = json.load(f)
            d = len(class_to_emb[0]['emb'])
            # buffer sizes exactly as :46-48 (incl. the trailing-'' quirk of unknown files)
            all_embs = torch.zeros((self.num_classes + 1, d), dtype=torch.float)
            novel_embs = torch.zeros((len(self.unknown_cat_names) + 1, d), dtype=torch.float)
            base_embs = torch.zeros((len(all_embs) - len(novel_embs) + 1, d), dtype=torch.float)
            names = []
            i = j = k = 0
            for cd in class_to_emb:
                if self.known_file and cd['name'] not in self.known_cat_names:
                    continue
                e = torch.FloatTensor(cd['emb'])
                if self.unknown_file:
                    if cd['name'] in self.unknown_cat_names:
                        novel_embs[j] = e
                        j += 1
                    else:
                        base_embs[k] = e
                        k += 1
                all_embs[i] = e
                names.append(cd['name'])
                i += 1
            self.register_buffer('all_class_embs', all_embs)
            self.register_buffer('novel_class_embs', novel_embs)
            self.register_buffer('base_class_embs', base_embs)
            self.all_classes = len(all_embs) - 1
            self.novel_classes = len(novel_embs) - 1
            self.base_classes = len(base_embs) - 1
            self.ordered_class_names = names

    @property
    def with_loss(self):
        return self.loss_panoptic is not None

    def forward_train(self, **kwargs):
        """MaskFormerFusionHead has no training loss."""
        return dict()

    # ------------------------------------------------------------------------------------------
    def get_cls_emb_scores(self, cls_emb_preds, gt_cls_embs):
        """:297-315 -- softmax(emb @ E^T), one wavefront per query row."""
        dots = torch.matmul(cls_emb_preds, gt_cls_embs.t()).contiguous()
        prob, _, _ = ops.rowwise_softmax_argmax(dots, want_prob=True)
        return prob

    @staticmethod
    def _geom(mask_pred, meta, rescale):
        """(logits (Q,h,w), up, crop, out) for one image."""
        if isinstance(mask_pred, LowResMasks):
            logits, up = mask_pred.logits, mask_pred.up_size
        else:  # an already-upsampled tensor: stage 1 degenerates to the identity
            logits, up = mask_pred, tuple(mask_pred.shape[-2:])
        crop = tuple(int(v) for v in meta['img_shape'][:2])
        crop = (min(crop[0], up[0]), min(crop[1], up[1]))
        out = tuple(int(v) for v in meta['ori_shape'][:2]) if rescale else crop
        return logits.contiguous(), up, crop, out

    def _topk(self, scores):
        """:340-347 given per-query class scores (Q, n) (background column already dropped):
        (labels, class scores, query indices) of the `max_per_image` best (query, class) pairs."""
        max_per_image = self.test_cfg.get('max_per_image', 100)
        n_cls = scores.shape[-1]
        flat = scores.flatten(0, 1)
        k = min(max_per_image, flat.numel())
        scores_per_image, top_indices = flat.topk(k, sorted=False)
        labels_per_image = top_indices % n_cls
        query_indices = torch.div(top_indices, n_cls, rounding_mode='floor')
        return labels_per_image, scores_per_image, query_indices

    def _batch_picks(self, emb_results, tables):
        """:297-347 for the whole batch and all evaluation types: ONE GEMM against the concatenated class tables and one
        `cgg_class_topk` launch -> (labels, class scores, query indices), each (B, T, k); None when the shapes are
        outside the kernel's limits (the per-type torch path then runs)."""
        if not (torch.is_tensor(emb_results) and emb_results.dim() == 3 and emb_results.is_cuda):
            return None
        B, Q, D = emb_results.shape
        ncols = [int(e.shape[0]) for e in tables]
        k = self.test_cfg.get('max_per_image', 100)
        if not ops.class_topk_supported(Q, ncols, k):
            return None
        from . import runtime
        cat = runtime.derived_cached('fusion_cls_cat', tuple(tables), lambda: torch.cat(list(tables), 0).float().contiguous())
        x = emb_results.reshape(B * Q, D).float()
        # 200 x 134 x 768: a BLAS library serves this with ONE workgroup (119 us measured); the skinny-linear kernel
        # (3 bf16 MFMAs on hi/lo operands, f32-class accuracy) spreads it over 10
        dots = ops.linear_rows(x, cat, split=True) if D % 16 == 0 else torch.matmul(x, cat.t())
        col0 = [sum(ncols[:t]) for t in range(len(ncols))]
        return ops.class_topk(dots, B, col0, ncols, k)

    def _instances_multi(self, picks, geom):
        """:349-363 for SEVERAL evaluation types of one image at once: picks = [(labels, class scores, query indices)];
        binary mask / mask score / bbox depend on the QUERY only, so every picked query's mask is interpolated once
        (`cgg_instance_masks_multi`) and stored straight into each type's detection slots -- no per-type gather pass
        over (n, H, W) masks. Returns [(labels, bboxes (n,5), masks (n,H,W) bool)] in the order of `picks`."""
        logits, up, crop, out = geom
        masks_l, qscores, qboxes = ops.instance_masks_multi(logits, [p[2] for p in picks], up, crop, out)
        res = []
        for (labels, cls_scores, qidx), masks in zip(picks, masks_l):
            det_scores = cls_scores * qscores.index_select(0, qidx)
            res.append((labels, torch.cat([qboxes.index_select(0, qidx), det_scores[:, None]], dim=-1), masks))
        return res

    def _geom_or_identity(self, mask_pred, meta, rescale):
        if meta is not None:
            return self._geom(mask_pred, meta, rescale)
        hw = tuple(mask_pred.shape[-2:])
        return mask_pred.contiguous(), hw, hw, hw

    def instance_postprocess_emb(self, mask_cls_emb, mask_pred, gt_cls_embs, meta=None, rescale=False):
        """:317-366 -> (labels (n,), bboxes (n,5) [x0,y0,x1,y1,score], masks (n,H,W) bool)."""
        geom = self._geom_or_identity(mask_pred, meta, rescale)
        scores = self.get_cls_emb_scores(mask_cls_emb, gt_cls_embs)[:, :-1]
        return self._instances_multi([self._topk(scores)], geom)[0]

    def instance_postprocess(self, mask_cls, mask_pred, meta=None, rescale=False):
        """:245-295 (closed-set variant on the classification logits)."""
        geom = self._geom_or_identity(mask_pred, meta, rescale)
        prob, _, _ = ops.rowwise_softmax_argmax(mask_cls.contiguous(), want_prob=True)
        labels, bboxes, masks = self._instances_multi([self._topk(prob[:, :-1])], geom)[0]
        is_thing = labels < self.num_things_classes
        return labels[is_thing], bboxes[is_thing], masks[is_thing]

    def _panoptic_from_scores(self, scores, labels, geom, defer_stuff):
        """shared body of :77-159 (defer_stuff=True) and :161-225 (False)."""
        logits, up, crop, out = geom
        thr = self.test_cfg.get('object_mask_thr', 0.8)
        iou_thr = self.test_cfg.get('iou_thr', 0.8)
        filter_low_score = self.test_cfg.get('filter_low_score', False)
        stuff_area_limit = self.test_cfg.get('stuff_area_limit', 4096)
        keep = labels.ne(self.num_classes) & (scores > thr)
        keep_idx = torch.nonzero(keep, as_tuple=False).squeeze(1)
        n = int(keep_idx.numel())                                   # host sync #1 (data-dependent size)
        if n == 0:
            return torch.full(out, self.num_classes, dtype=torch.int32, device=logits.device)
        cur_scores = scores[keep_idx].contiguous()
        ids, win_half, counts = ops.panoptic_argmax(logits, keep_idx.to(torch.int32).contiguous(),
                                                    cur_scores, up, crop, out)
        host = torch.cat([counts.flatten(), labels[keep_idx].to(torch.int32)]).cpu().tolist()  # sync #2
        cnt, classes = host[:3 * n], host[3 * n:]
        lut_val = [-1] * n
        lut_half = [0] * n
        instance_id = 1
        stuff = []
        for k in range(n):
            pred_class = classes[k]
            isthing = pred_class < self.num_things_classes
            mask_area, original_area = cnt[3 * k], cnt[3 * k + 1]
            if mask_area > 0 and original_area > 0:
                if mask_area / original_area < iou_thr:
                    continue
                if not isthing:
                    if defer_stuff:
                        stuff.append(k)
                        continue
                    lut_val[k] = pred_class
                    lut_half[k] = 1 if filter_low_score else 0
                else:
                    lut_val[k] = pred_class + instance_id * INSTANCE_OFFSET
                    lut_half[k] = 1 if filter_low_score else 0
                    instance_id += 1
        for k in stuff:
            # pasted onto still-void pixels only: the argmax partition is disjoint, so that is all of
            # (ids == k); the area test uses the UNFILTERED region (:151-157)
            if cnt[3 * k] < stuff_area_limit:
                continue
            lut_val[k] = classes[k]
            lut_half[k] = 0
        dev = logits.device
        return ops.panoptic_paint(ids, win_half, torch.tensor(lut_val, dtype=torch.int32, device=dev),
                                  torch.tensor(lut_half, dtype=torch.int32, device=dev), self.num_classes)

    def panoptic_postprocess_emb(self, mask_cls_emb, mask_pred, gt_cls_embs, meta=None, rescale=False):
        """:77-159 -> (H,W) int32 map, `cls + instance_id * INSTANCE_OFFSET`, void = num_classes."""
        geom = self._geom(mask_pred, meta, rescale) if meta is not None else \
            (mask_pred.contiguous(), tuple(mask_pred.shape[-2:]), tuple(mask_pred.shape[-2:]),
             tuple(mask_pred.shape[-2:]))
        dots = torch.matmul(mask_cls_emb, gt_cls_embs.t()).contiguous()
        _, scores, labels = ops.rowwise_softmax_argmax(dots, want_prob=False)
        return self._panoptic_from_scores(scores, labels, geom, defer_stuff=True)

    def panoptic_postprocess(self, mask_cls, mask_pred, meta=None, rescale=False):
        """:161-225."""
        geom = self._geom(mask_pred, meta, rescale) if meta is not None else \
            (mask_pred.contiguous(), tuple(mask_pred.shape[-2:]), tuple(mask_pred.shape[-2:]),
             tuple(mask_pred.shape[-2:]))
        _, scores, labels = ops.rowwise_softmax_argmax(mask_cls.contiguous(), want_prob=False)
        return self._panoptic_from_scores(scores, labels, geom, defer_stuff=False)

    def semantic_postprocess(self, mask_cls, mask_pred):
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------
    def simple_test(self, mask_cls_results, mask_cls_emb_results, mask_pred_results, img_metas, **kwargs):
        """:369-464 -> list (one dict per image) keyed by eval type."""
        eval_types = self.test_cfg.get('eval_types', [])
        rescale = kwargs.get('rescale', False)
        # this build's extension for index-parity checks: the (query, class) picks behind every detection, i.e. the
        # `top_indices // n_cls` of :344-347 that the reference computes and drops
        want_idx = bool(kwargs.get('with_query_indices', False))
        results = []
        todo = [(key, embs) for key, embs in (('all_results', getattr(self, 'all_class_embs', None)),
                                              ('novel_results', getattr(self, 'novel_class_embs', None)),
                                              ('base_results', getattr(self, 'base_class_embs', None)))
                if key in eval_types and not (key == 'all_results' and self.panoptic_mode)]
        batch_picks = self._batch_picks(mask_cls_emb_results, [e for _, e in todo]) if todo else None
        for b, meta in enumerate(img_metas):
            mask_cls_result = mask_cls_results[b]
            emb = mask_cls_emb_results[b]
            mp = mask_pred_results[b]
            result = dict()
            if 'all_results' in eval_types and self.panoptic_mode:
                result['panoptic_all_results'] = self.panoptic_postprocess_emb(
                    emb, mp, self.all_class_embs, meta, rescale)
            # the embedding-based instance types of this image share ONE mask pass (same reference semantics as three
            # instance_postprocess_emb calls, :385-400)
            if todo and batch_picks is not None:
                # 4 launches per image: slot plan, one mask pass for all types, per-detection boxes / scores
                labels, cls_scores, qidx = batch_picks
                logits, up, crop, out = self._geom(mp, meta, rescale)
                # mask_bits (this build's extension): masks come back bit-packed, (n, H, W / 8) uint8 with pixel x in bit
                # x & 7 -- 8x fewer bytes for a host-side consumer to copy; callers unpack (detectors.simple_test does)
                bits = bool(kwargs.get('mask_bits', False)) and ops.instance_masks_bitpack_ok(logits.shape[-2:], up, crop, out)
                masks, bboxes = ops.instance_masks_picks(logits, qidx[b].reshape(-1), cls_scores[b].reshape(-1), up, crop,
                                                         out, bitpack=bits)
                k = labels.shape[-1]
                for t, (key, _) in enumerate(todo):
                    result[key] = (labels[b, t], bboxes[t * k:(t + 1) * k], masks[t * k:(t + 1) * k])
                if want_idx:
                    result['query_indices'] = {key: qidx[b, t].long() for t, (key, _) in enumerate(todo)}
                    result['class_scores'] = {key: cls_scores[b, t] for t, (key, _) in enumerate(todo)}
            elif todo:
                geom = self._geom(mp, meta, rescale)
                picks = [self._topk(self.get_cls_emb_scores(emb, embs)[:, :-1]) for _, embs in todo]
                for (key, _), r in zip(todo, self._instances_multi(picks, geom)):
                    result[key] = r
                if want_idx:
                    result['query_indices'] = {key: p[2].long() for (key, _), p in zip(todo, picks)}
                    result['class_scores'] = {key: p[1] for (key, _), p in zip(todo, picks)}
            if 'ins_results' in eval_types:
                result['ins_results'] = self.instance_postprocess(mask_cls_result, mp, meta, rescale)
            if 'pan_results' in eval_types:
                result['pan_results'] = self.panoptic_postprocess(mask_cls_result, mp, meta, rescale)
            results.append(result)
        return results
