"""Synthetic stand-ins for everything the benchmark cannot fetch offline:
class-embedding JSON / known / unknown files with the reference's schema, a model config with the
reference's keys (the `model = dict(type='Mask2FormerOpen', ...)` block of
configs/instance/coco_b48n17.py:16-188 and configs/openset_panoptic/coco_panoptic_p20.py), and
COCO-shaped seeded input batches.
"""
import json
import os
import tempfile

import torch

_TMP = {}


def write_class_files(num_classes, num_unknown, dim=768, seed=0, root=None):
    """-> dict(class_to_emb_file, known_file, unknown_file). Embeddings ~ N(-0.03, 0.77^2) like the
    measured statistics of datasets/embeddings/*.json; names 'class_i'."""
    key = (num_classes, num_unknown, dim, seed, root)
    if key in _TMP and all(os.path.exists(p) for p in _TMP[key].values()):
        return _TMP[key]
    root = root or tempfile.mkdtemp(prefix='cgg_synth_')
    os.makedirs(root, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    embs = (torch.randn(num_classes, dim, generator=g) * 0.77 - 0.03).tolist()
    names = [f'class_{i}' for i in range(num_classes)]
    # unknown classes: every (num_classes // num_unknown)-th name
    unknown = names[::max(1, num_classes // max(num_unknown, 1))][:num_unknown] if num_unknown else []
    paths = dict(class_to_emb_file=os.path.join(root, f'class_emb_{num_classes}.json'),
                 known_file=os.path.join(root, f'known_{num_classes}.txt'),
                 unknown_file=os.path.join(root, f'unknown_{num_unknown}.txt'))
    with open(paths['class_to_emb_file'], 'w') as f:
        json.dump([dict(id=i, name=n, emb=e) for i, (n, e) in enumerate(zip(names, embs))], f)
    with open(paths['known_file'], 'w') as f:
        f.write('\n'.join(names))
    with open(paths['unknown_file'], 'w') as f:
        f.write('\n'.join(unknown))
    _TMP[key] = paths
    return paths


def model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=100, depth=50, panoptic=False,
                 num_points=12544, use_caption=True, use_caption_generation=True, files=None,
                 eval_types=None, enc_layers=6, dec_layers=9, vocab=30522, seed=0):
    """A `model` dict with the keys of the reference configs (Mask2FormerOpen / R50 / 100 queries).
    The head is trained on the known classes only (num_things_classes = known), the fusion head scores
    all classes -- as in coco_b48n17.py:30-36 and :154-163."""
    num_classes = num_things + num_stuff
    num_known = num_classes - num_unknown
    files = files or write_class_files(num_classes, num_unknown, seed=seed)
    if eval_types is None:
        eval_types = ['all_results'] if panoptic else ['all_results', 'novel_results', 'base_results']
    head_things = num_known if not panoptic else num_things - num_unknown
    head_classes = head_things + num_stuff
    in_channels = [256, 512, 1024, 2048] if depth >= 50 else [64, 128, 256, 512]
    return dict(
        type='Mask2FormerOpen',
        backbone=dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=3,
                      norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch',
                      init_cfg=None),
        panoptic_head=dict(
            type='Mask2FormerHeadOpen', in_channels=in_channels, strides=[4, 8, 16, 32], feat_channels=256,
            out_channels=256, num_things_classes=head_things, num_stuff_classes=num_stuff,
            num_queries=num_queries, num_transformer_feat_level=3,
            pixel_decoder=dict(
                type='MSDeformAttnPixelDecoder', num_outs=3, norm_cfg=dict(type='GN', num_groups=32),
                act_cfg=dict(type='ReLU'),
                encoder=dict(
                    type='DetrTransformerEncoder', num_layers=enc_layers,
                    transformerlayers=dict(
                        type='BaseTransformerLayer',
                        attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256, num_heads=8,
                                       num_levels=3, num_points=4, im2col_step=64, dropout=0.0,
                                       batch_first=False, norm_cfg=None, init_cfg=None),
                        ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, num_fcs=2,
                                      ffn_drop=0.0, act_cfg=dict(type='ReLU', inplace=True)),
                        operation_order=('self_attn', 'norm', 'ffn', 'norm')),
                    init_cfg=None),
                positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True),
                init_cfg=None),
            enforce_decoder_input_project=False,
            positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True),
            transformer_decoder=dict(
                type='DetrTransformerDecoder', return_intermediate=True, num_layers=dec_layers,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=dict(type='MultiheadAttention', embed_dims=256, num_heads=8, attn_drop=0.0,
                                   proj_drop=0.0, dropout_layer=None, batch_first=False),
                    ffn_cfgs=dict(embed_dims=256, feedforward_channels=2048, num_fcs=2,
                                  act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0.0, dropout_layer=None,
                                  add_identity=True),
                    feedforward_channels=2048,
                    operation_order=('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm')),
                init_cfg=None),
            caption_generator=dict(type='CaptionTransformer', nb_layers=4, input_dim=768, hidden_dim=768,
                                   ff_dim=512, nb_heads=8, drop_val=0.1, pre_norm=False, seq_length=35,
                                   nb_tokens=vocab),
            loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.0, reduction='mean',
                          class_weight=[1.0] * head_classes + [0.1]),
            loss_cls_emb=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0, reduction='mean',
                              class_weight=[1.0] * head_classes + [0.1]),
            loss_grounding=dict(type='GroundingLoss', loss_weight=2.0),
            loss_caption_generation=dict(type='CrossEntropyLoss', ignore_index=0, loss_weight=2.0),
            loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='mean', loss_weight=5.0),
            loss_dice=dict(type='DiceLoss', use_sigmoid=True, activate=True, reduction='mean',
                           naive_dice=True, eps=1.0, loss_weight=5.0),
            class_agnostic=False, use_caption=use_caption, use_class_emb=True,
            use_caption_generation=use_caption_generation,
            class_to_emb_file=files['class_to_emb_file'], known_file=files['known_file'],
            unknown_file=files['unknown_file'], softmax_temperature=10, pred_emb_norm=False,
            text_emb_norm=True, caption_emb_type='bert', caption_gen_emb_type='bert',
            synthetic_text_encoder=True),
        panoptic_fusion_head=dict(
            type='MaskFormerFusionHeadOpen', num_things_classes=num_things if panoptic else num_classes,
            num_stuff_classes=num_stuff if panoptic else 0, panoptic_mode=panoptic, loss_panoptic=None,
            init_cfg=None, use_class_emb=True, class_to_emb_file=files['class_to_emb_file'],
            known_file=files['known_file'], unknown_file=files['unknown_file']),
        train_cfg=dict(
            num_points=num_points, oversample_ratio=3.0, importance_sample_ratio=0.75,
            assigner=dict(type='MaskHungarianAssignerOpen',
                          cls_cost=dict(type='ClassificationCost', weight=0.0),
                          cls_emb_cost=dict(type='ClassificationCost', weight=2.0),
                          mask_cost=dict(type='CrossEntropyLossCost', weight=5.0, use_sigmoid=True),
                          dice_cost=dict(type='DiceCost', weight=5.0, pred_act=True, eps=1.0)),
            sampler=dict(type='MaskPseudoSampler')),
        test_cfg=dict(eval_types=eval_types, max_per_image=100, iou_thr=0.8, filter_low_score=True,
                      use_class_emb=True),
        init_cfg=None)


def backbone_feats(B, H, W, channels=(256, 512, 1024, 2048), seed=0, device='cpu'):
    """N(0,1) feature maps at strides 4/8/16/32 (used when the backbone is bypassed)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, c, H // s, W // s, generator=g).to(device) for c, s in zip(channels, (4, 8, 16, 32))]


def structured_images(B, H, W, seed=0, shapes=24, noise=0.15):
    """Normalised-image-like batch WITH spatial structure: a smooth background plus `shapes` random rectangles /
    ellipses of random colour per image and a little noise. White-noise images average out in a stride-4 backbone and
    give spatially constant features (every mask all-on or all-off); parity tests need masks with real boundaries."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(-1, 1, H).view(1, H, 1)
    xs = torch.linspace(-1, 1, W).view(1, 1, W)
    out = torch.empty(B, 3, H, W)
    for b in range(B):
        a = torch.randn(3, 1, 1, generator=g)
        img = a * ys + torch.randn(3, 1, 1, generator=g) * xs + 0.3 * torch.randn(3, 1, 1, generator=g)
        for i in range(shapes):
            cy, cx = (torch.rand(2, generator=g) * 2 - 1).tolist()
            ry, rx = (torch.rand(2, generator=g) * 0.35 + 0.04).tolist()
            col = torch.randn(3, 1, 1, generator=g) * 1.2
            if i % 2 == 0:
                m = ((ys - cy).abs() <= ry) & ((xs - cx).abs() <= rx)
            else:
                m = ((ys - cy) / ry)**2 + ((xs - cx) / rx)**2 <= 1
            img = torch.where(m, col.expand(3, H, W), img)
        out[b] = img + noise * torch.randn(3, H, W, generator=g)
    return out


def img_metas(B, H, W, ori=None):
    ori = ori or (H, W)
    return [dict(img_shape=(H, W, 3), ori_shape=(ori[0], ori[1], 3), pad_shape=(H, W, 3),
                 batch_input_shape=(H, W), scale_factor=1.0, flip=False) for _ in range(B)]


def train_batch(B, H, W, num_classes, max_inst=20, T=35, vocab=30522, seed=0, device='cpu', stuff_range=None):
    """COCO-shaped synthetic training batch: labels, rectangle/ellipse masks at image
    resolution, caption ids [101, tokens..., 102, 0...], noun ids."""
    g = torch.Generator().manual_seed(seed)
    out = dict(gt_bboxes=[], gt_labels=[], gt_masks=[], gt_caption_ids=[], gt_caption_mask=[],
               gt_caption_nouns_ids=[], gt_caption_nouns_mask=[])
    ys = torch.arange(H).view(H, 1).float()
    xs = torch.arange(W).view(1, W).float()
    for _ in range(B):
        n = int(torch.randint(1, max_inst + 1, (1,), generator=g))
        labels = torch.randint(0, num_classes, (n,), generator=g)
        masks = torch.zeros(n, H, W, dtype=torch.uint8)
        boxes = torch.zeros(n, 4)
        for i in range(n):
            cy, cx = (torch.rand(2, generator=g) * torch.tensor([H, W])).tolist()
            hh, ww = (torch.rand(2, generator=g) * torch.tensor([H / 3, W / 3]) + 4).tolist()
            if i % 2 == 0:
                m = ((ys - cy).abs() <= hh / 2) & ((xs - cx).abs() <= ww / 2)
            else:
                m = ((ys - cy) / (hh / 2))**2 + ((xs - cx) / (ww / 2))**2 <= 1
            masks[i] = m.to(torch.uint8)
            boxes[i] = torch.tensor([max(cx - ww / 2, 0), max(cy - hh / 2, 0), min(cx + ww / 2, W), min(cy + hh / 2, H)])
        Ltok = int(torch.randint(5, 21, (1,), generator=g))
        ids = torch.zeros(T, dtype=torch.long)
        ids[0] = 101
        ids[1:1 + Ltok] = torch.randint(min(1000, vocab // 2), vocab, (Ltok,), generator=g)
        ids[1 + Ltok] = 102
        cmask = (ids != 0).long()
        nn_ = int(torch.randint(1, 7, (1,), generator=g))
        nouns = torch.zeros(T, dtype=torch.long)
        nouns[:nn_] = ids[1:1 + nn_]
        out['gt_bboxes'].append(boxes.to(device))
        out['gt_labels'].append(labels.to(device))
        out['gt_masks'].append(masks.to(device))
        out['gt_caption_ids'].append(ids.to(device))
        out['gt_caption_mask'].append(cmask.to(device))
        out['gt_caption_nouns_ids'].append(nouns.to(device))
        out['gt_caption_nouns_mask'].append((nouns != 0).long().to(device))
    return out
