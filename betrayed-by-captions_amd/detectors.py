"""Detectors registered by the reference: `Mask2FormerOpen` (open_set/models/mask2former.py:6-27) and
`MaskFormerOpen` (open_set/models/maskformer.py:14-381): backbone -> panoptic_head ->
panoptic_fusion_head, plus the [3P] mmdet BaseDetector plumbing they inherit (`forward(return_loss=)`,
`train_step`, `_parse_losses`) that tools/train.py / tools/test.py call.
"""
import copy
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .config import to_config_dict
from .registry import DETECTORS, build_backbone, build_head, build_neck


def bbox2result(bboxes, labels, num_classes):
    """[3P] mmdet.core.bbox2result: (n,5) boxes + labels -> list of per-class numpy arrays."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


@DETECTORS.register_module()
class MaskFormerOpen(nn.Module):

    def __init__(self, backbone, neck=None, panoptic_head=None, panoptic_fusion_head=None, train_cfg=None,
                 test_cfg=None, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        panoptic_head_ = copy.deepcopy(dict(panoptic_head))
        panoptic_head_.update(train_cfg=train_cfg)
        panoptic_head_.update(test_cfg=test_cfg)
        self.panoptic_head = build_head(panoptic_head_)
        fusion_ = copy.deepcopy(dict(panoptic_fusion_head))
        fusion_.update(test_cfg=test_cfg)
        self.panoptic_fusion_head = build_head(fusion_)
        self.num_things_classes = self.panoptic_fusion_head.num_things_classes
        self.num_stuff_classes = self.panoptic_fusion_head.num_stuff_classes
        self.num_classes = self.panoptic_fusion_head.num_classes
        self.train_cfg = to_config_dict(train_cfg) if train_cfg is not None else None
        self.test_cfg = to_config_dict(test_cfg) if test_cfg is not None else to_config_dict({})

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    def init_weights(self):
        if hasattr(self.backbone, 'init_weights'):
            self.backbone.init_weights()
        self.panoptic_head.init_weights()

    def extract_feat(self, img):
        # x3a hand-over (parity-mode ResNet -> pixel decoder, csrc/x3.h) only when this detector's head is the direct consumer:
        # a neck, or any other user of `self.backbone`, gets plain float32 maps
        if hasattr(self.backbone, 'x3a_outputs'):
            self.backbone.x3a_outputs = (not self.with_neck and hasattr(self, 'panoptic_head')
                                         and hasattr(getattr(self.panoptic_head, 'pixel_decoder', None), 'forward_stream_x3'))
        try:
            x = self.backbone(img)
        finally:
            if hasattr(self.backbone, 'x3a_outputs'):
                self.backbone.x3a_outputs = False
        if self.with_neck:
            x = self.neck(x)
        return x

    # ---- [3P] BaseDetector plumbing --------------------------------------------------------------
    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        if not isinstance(imgs, (list, tuple)):
            imgs, img_metas = [imgs], [img_metas]
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        for img, img_meta in zip(imgs, img_metas):
            for img_id in range(len(img_meta)):
                img_meta[img_id]['batch_input_shape'] = tuple(img.size()[-2:])
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """[3P] BaseDetector._parse_losses; the per-key scalar all-reduces of the reference (71 per step,
        SURVEY.md C4) are ONE vector all-reduce here."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        keys = list(log_vars.keys())
        vec = torch.stack([log_vars[k].detach().float().reshape(()) for k in keys])
        if dist.is_available() and dist.is_initialized():
            n = torch.tensor([float(len(keys))], device=vec.device)
            dist.all_reduce(n)
            assert int(n.item()) == len(keys) * dist.get_world_size(), 'loss log variables are different across GPUs!'
            dist.all_reduce(vec.div_(dist.get_world_size()))
        vals = vec.tolist()
        return loss, OrderedDict((k, v) for k, v in zip(keys, vals))

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer=None):
        return self.train_step(data, optimizer)

    # ---- the reference's own methods ----------------------------------------------------------
    def forward_dummy(self, img):
        """maskformer.py:53-78 (FLOPs hook: head forward + one caption-generator pass)."""
        img_metas = [{'img_shape': [1280, 800], 'batch_input_shape': tuple(img.shape[-2:])}
                     for _ in range(img.shape[0])]
        x = self.extract_feat(img)
        outs = self.panoptic_head(x, img_metas)
        if getattr(self.panoptic_head, 'use_caption_generation', False):
            emb = torch.randn([1, 35, 768], device=img.device)
            masks = torch.ones([1, 35], device=img.device).bool()
            preds = torch.randn([1, 100, 768], device=img.device)
            self.panoptic_head.caption_generator(tgt=emb[:, :-1, :], memory=preds,
                                                 tgt_key_padding_mask=torch.logical_not(masks[:, :-1]))
        return outs

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_masks, gt_caption_ids=None,
                      gt_caption_mask=None, gt_caption_nouns_ids=None, gt_caption_nouns_mask=None,
                      gt_semantic_seg=None, gt_bboxes_ignore=None, **kwargs):
        """maskformer.py:80-133."""
        batch_input_shape = tuple(img[0].size()[-2:])
        for img_meta in img_metas:
            img_meta['batch_input_shape'] = batch_input_shape
        x = self.extract_feat(img)
        return self.panoptic_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_masks, gt_semantic_seg,
                                                gt_caption_ids, gt_caption_mask, gt_caption_nouns_ids,
                                                gt_caption_nouns_mask, gt_bboxes_ignore, **kwargs)

    def simple_test(self, imgs, img_metas, **kwargs):
        """maskformer.py:135-219. `device_results=True` (this build's extension) returns the fusion
        head's device tensors and skips the per-mask `.cpu().numpy()` conversion of :205-208."""
        feats = None if kwargs.get('encoded') is not None else self.extract_feat(imgs)
        assigned_labels, mask_cls_emb_results, mask_pred_results, caption_results, att = \
            self.panoptic_head.simple_test(feats, img_metas, **kwargs)
        host = not kwargs.get('device_results', False)
        results = self.panoptic_fusion_head.simple_test(assigned_labels, mask_cls_emb_results,
                                                        mask_pred_results, img_metas, **kwargs)
        if not host:
            return results
        fh = self.panoptic_fusion_head
        for i in range(len(results)):
            for res_type in self.test_cfg.get('eval_types', []):
                if res_type == 'cap_results':
                    results[i][res_type] = caption_results
                    continue
                pred_classes = {'all_results': getattr(fh, 'all_classes', None),
                                'novel_results': getattr(fh, 'novel_classes', None),
                                'base_results': getattr(fh, 'base_classes', None),
                                'ins_results': fh.num_classes}.get(res_type)
                if 'pan' in list(results[i].keys())[0]:
                    key = list(results[i].keys())[0]
                    if isinstance(results[i][key], torch.Tensor):
                        results[i][key] = results[i][key].detach().cpu().numpy()
                else:
                    labels_per_image, bboxes, mask_pred_binary = results[i][res_type]
                    bbox_results = bbox2result(bboxes, labels_per_image, pred_classes)
                    masks_np = mask_pred_binary.detach().cpu().numpy()      # ONE copy for all masks
                    if masks_np.dtype == np.uint8:      # `mask_bits=True` was requested: (n, H, W / 8) -> the reference's bool
                        masks_np = np.unpackbits(masks_np, axis=-1, bitorder='little').view(np.bool_)
                    mask_results = [[] for _ in range(pred_classes)]
                    for j, label in enumerate(labels_per_image.detach().cpu().tolist()):
                        mask_results[label].append(masks_np[j])
                    results[i][res_type] = bbox_results, mask_results
            if kwargs.get('with_mask', False):
                results[i]['mask'] = mask_pred_results.upsampled().cpu().numpy()
            if kwargs.get('with_att', False):
                results[i]['att'] = att.cpu().numpy()
            if kwargs.get('gt_labels', None) is not None:
                results[i]['visual'] = (mask_cls_emb_results.squeeze().cpu().numpy(),
                                        assigned_labels.cpu().numpy())
        return results

    # ---- two-stage serving split (this build's extension; see pipeline.TwoStagePipeline) ----
    def stage_encode(self, imgs, defer_tail=False):
        """backbone + pixel decoder + K/V projections + packed mask feature: the throughput-bound, query-independent
        part of `simple_test`. `defer_tail` leaves the K/V projections and the mask-feature packing to `stage_decode`
        (stage balancing for the pipeline)."""
        return self.panoptic_head._encode(self.extract_feat(imgs), defer_tail=defer_tail)

    def stage_decode(self, encoded, img_metas, **kwargs):
        """query decoder + mask logits + post-processing on the output of `stage_encode`."""
        return self.simple_test(None, img_metas, encoded=encoded, **kwargs)

    def stage_head(self, encoded, img_metas, **kwargs):
        """query decoder only (the narrow, latency-bound part): the head's `simple_test` outputs."""
        return self.panoptic_head.simple_test(None, img_metas, encoded=encoded, **kwargs)

    def stage_post(self, head_out, img_metas, **kwargs):
        """fusion-head post-processing (wide kernels again) on `stage_head`'s outputs; device results."""
        assigned_labels, mask_cls_emb_results, mask_pred_results, _, _ = head_out
        return self.panoptic_fusion_head.simple_test(assigned_labels, mask_cls_emb_results, mask_pred_results,
                                                     img_metas, **kwargs)

    def aug_test(self, imgs, img_metas, **kwargs):
        raise NotImplementedError

    def onnx_export(self, img, img_metas):
        raise NotImplementedError(f'{self.__class__.__name__} does not support ONNX EXPORT')


@DETECTORS.register_module()
class Mask2FormerOpen(MaskFormerOpen):
    """open_set/models/mask2former.py:6-27."""

    def __init__(self, backbone, neck=None, panoptic_head=None, panoptic_fusion_head=None, train_cfg=None,
                 test_cfg=None, init_cfg=None):
        super().__init__(backbone, neck=neck, panoptic_head=panoptic_head,
                         panoptic_fusion_head=panoptic_fusion_head, train_cfg=train_cfg, test_cfg=test_cfg,
                         init_cfg=init_cfg)
