"""CPU tests of the host-side mirror of the reference plug-in API: registry / config contract, constructor
signatures and state_dict key layout, buffers built from the class-embedding files, forward refusing to run
without the HIP path."""
import copy
import os
import warnings

import pytest
import torch

import cgg_amd
from cgg_amd import registry, synthetic
from cgg_amd.config import Config, ConfigDict
from cgg_amd.query_decoder import pack_bool_mask
from cgg_amd import ops

from util import head_cfg, small_cfg


def test_registry_contract():
    R = registry.Registry('thing')

    @R.register_module()
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y

    a = R.build(dict(type='A', x=1))
    assert (a.x, a.y) == (1, 2)
    with pytest.raises(KeyError, match='is not in the thing registry'):
        R.build(dict(type='Nope'))
    with pytest.raises(TypeError):
        R.build(['not', 'a', 'dict'])
    with pytest.raises(KeyError, match='already registered'):
        R.register_module(module=A)
    R.register_module(module=A, force=True)
    with pytest.raises(KeyError):
        R.build(dict(x=1))
    # the names the reference registers, under the registries it registers them in
    for reg, names in ((registry.DETECTORS, ['Mask2FormerOpen', 'MaskFormerOpen']),
                       (registry.HEADS, ['Mask2FormerHeadOpen', 'MaskFormerFusionHeadOpen', 'CaptionTransformer',
                                        'V2lTranformHead']),
                       (registry.LOSSES, ['GroundingLoss', 'CrossEntropyLossOpen', 'CrossEntropyLoss', 'DiceLoss']),
                       (registry.BBOX_ASSIGNERS, ['MaskHungarianAssignerOpen']),
                       (registry.PLUGIN_LAYERS, ['MSDeformAttnPixelDecoder']),
                       (registry.ATTENTION, ['MultiScaleDeformableAttention', 'MultiheadAttention']),
                       (registry.TRANSFORMER_LAYER_SEQUENCE, ['DetrTransformerEncoder', 'DetrTransformerDecoder']),
                       (registry.TRANSFORMER_LAYER, ['BaseTransformerLayer', 'DetrTransformerDecoderLayer']),
                       (registry.MATCH_COST, ['ClassificationCost', 'CrossEntropyLossCost', 'DiceCost']),
                       (registry.BACKBONES, ['ResNet'])):
        for n in names:
            assert n in reg, (reg.name, n)


def test_config_base_inheritance_delete_and_dotted_overrides(tmp_path):
    (tmp_path / 'base.py').write_text("a = dict(x=1, y=dict(z=2, w=3))\nlst = [dict(type='A'), dict(type='B')]\nkeep = 5\n")
    (tmp_path / 'child.py').write_text(
        "_base_ = ['base.py']\na = dict(y=dict(_delete_=True, q=9), n=4)\nimport os\nval = os.path.basename('x/y')\n")
    c = Config.fromfile(str(tmp_path / 'child.py'))
    assert c.a.x == 1 and c.a.n == 4 and dict(c.a.y) == {'q': 9} and c.keep == 5 and c.val == 'y'
    assert isinstance(c.a, ConfigDict)
    c.merge_from_dict({'a.x': 7, 'lst.1.type': 'C', 'new.k': 1})
    assert c.a.x == 7 and c.lst[1].type == 'C' and c.new.k == 1
    with pytest.raises(AttributeError):
        c.a.nope
    with pytest.raises(FileNotFoundError):
        Config.fromfile(str(tmp_path / 'missing.py'))


@pytest.fixture(scope='module')
def detector():
    cfg = small_cfg()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = registry.build_detector(copy.deepcopy(cfg))
        m.init_weights()
    return cfg, m


def test_state_dict_layout_matches_reference_checkpoints(detector):
    cfg, m = detector
    sd = m.state_dict()
    expect = [
        'panoptic_head.pixel_decoder.input_convs.0.conv.weight', 'panoptic_head.pixel_decoder.input_convs.0.conv.bias',
        'panoptic_head.pixel_decoder.input_convs.2.gn.weight',
        'panoptic_head.pixel_decoder.encoder.layers.0.attentions.0.sampling_offsets.weight',
        'panoptic_head.pixel_decoder.encoder.layers.1.attentions.0.attention_weights.bias',
        'panoptic_head.pixel_decoder.encoder.layers.0.attentions.0.value_proj.weight',
        'panoptic_head.pixel_decoder.encoder.layers.0.attentions.0.output_proj.bias',
        'panoptic_head.pixel_decoder.encoder.layers.0.ffns.0.layers.0.0.weight',
        'panoptic_head.pixel_decoder.encoder.layers.0.ffns.0.layers.1.bias',
        'panoptic_head.pixel_decoder.encoder.layers.0.norms.1.weight',
        'panoptic_head.pixel_decoder.level_encoding.weight',
        'panoptic_head.pixel_decoder.lateral_convs.0.conv.weight', 'panoptic_head.pixel_decoder.lateral_convs.0.gn.bias',
        'panoptic_head.pixel_decoder.output_convs.0.conv.weight', 'panoptic_head.pixel_decoder.mask_feature.bias',
        'panoptic_head.transformer_decoder.layers.0.attentions.0.attn.in_proj_weight',
        'panoptic_head.transformer_decoder.layers.2.attentions.1.attn.out_proj.bias',
        'panoptic_head.transformer_decoder.layers.0.ffns.0.layers.0.0.weight',
        'panoptic_head.transformer_decoder.layers.0.norms.2.bias', 'panoptic_head.transformer_decoder.post_norm.weight',
        'panoptic_head.query_embed.weight', 'panoptic_head.query_feat.weight', 'panoptic_head.level_embed.weight',
        'panoptic_head.cls_embed.weight', 'panoptic_head.mask_embed.0.weight', 'panoptic_head.mask_embed.4.bias',
        'panoptic_head.class_embs', 'panoptic_head.v2l_transform.weight',
        'panoptic_head.bert_embeddings.word_embeddings.weight', 'panoptic_head.bert_embeddings.LayerNorm.bias',
        'panoptic_head.caption_generator.position_encoder.psne_layer',
        'panoptic_head.caption_generator.transformer_decoder.decoders.0.mha_layer.qkv_layer.weight',
        'panoptic_head.caption_generator.transformer_decoder.decoders.1.crx_layer.to_key.bias',
        'panoptic_head.caption_generator.transformer_decoder.decoders.0.ffn_layer.linears.1.0.weight',
        'panoptic_head.caption_generator.transformer_decoder.decoders.0.layer_normalz.ffn.1.bias',
        'panoptic_head.caption_generator.generator.weight',
        'panoptic_fusion_head.all_class_embs', 'panoptic_fusion_head.novel_class_embs',
        'panoptic_fusion_head.base_class_embs', 'backbone.conv1.weight', 'backbone.layer2.0.downsample.0.weight',
    ]
    for k in expect:
        assert k in sd, k
    assert 'panoptic_head.pixel_decoder.lateral_convs.0.conv.bias' not in sd     # no bias when normed
    assert sd['panoptic_head.pixel_decoder.encoder.layers.0.attentions.0.sampling_offsets.weight'].shape == (192, 256)
    assert sd['panoptic_head.transformer_decoder.layers.0.attentions.0.attn.in_proj_weight'].shape == (768, 256)
    h = m.panoptic_head
    nk = cfg['panoptic_head']['num_things_classes']
    assert sd['panoptic_head.class_embs'].shape == (nk + 1, 768)
    assert sd['panoptic_head.class_embs'][-1].abs().sum() == 0                  # background row = zeros
    assert sd['panoptic_head.class_embs'][:-1].abs().sum(1).min() > 0
    f = m.panoptic_fusion_head
    assert (f.all_classes, f.novel_classes, f.base_classes) == (10, 3, 7)
    # deformable attention init: zero offset weights, per-head directional bias ([3P] init_weights)
    att = h.pixel_decoder.encoder.layers[0].attentions[0]
    assert att.sampling_offsets.weight.abs().sum() == 0 and att.attention_weights.weight.abs().sum() == 0
    b = att.sampling_offsets.bias.view(8, 3, 4, 2)
    assert torch.allclose(b[0, 0, :, 0], torch.tensor([1., 2., 3., 4.])) and b[0, 0, :, 1].abs().max() < 1e-6


def test_frozen_stages_and_bn_eval(detector):
    _, m = detector
    m.train()
    bb = m.backbone
    assert not bb.conv1.weight.requires_grad and not bb.layer3[0].conv1.weight.requires_grad
    assert bb.layer4[0].conv1.weight.requires_grad
    assert all(not mod.training for mod in bb.modules() if isinstance(mod, torch.nn.BatchNorm2d))
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        outs = bb(x)
    assert [tuple(o.shape[1:]) for o in outs] == [(64, 16, 16), (128, 8, 8), (256, 4, 4), (512, 2, 2)]
    m.eval()


def test_head_forward_has_no_cpu_fallback(detector):
    _, m = detector
    feats = synthetic.backbone_feats(1, 64, 64, channels=(64, 128, 256, 512))
    with pytest.raises(cgg_amd._lib.CggError, match='ROCm device'):
        with torch.no_grad():
            m.panoptic_head.forward(feats, synthetic.img_metas(1, 64, 64))


def test_unsupported_config_values_raise():
    cfg = small_cfg()
    hc = head_cfg(cfg)
    hc['transformer_decoder']['transformerlayers']['attn_cfgs']['attn_drop'] = 0.1
    with pytest.raises(NotImplementedError):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            registry.build_head(hc)
    hc = head_cfg(cfg)
    hc['caption_emb_type'] = 'clip'
    with pytest.raises(NotImplementedError):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            registry.build_head(hc)
    hc = head_cfg(cfg)
    hc['pixel_decoder']['type'] = 'PixelDecoder'
    with pytest.raises(KeyError):
        registry.build_head(hc)


def test_pack_unpack_bool_mask_roundtrip():
    g = torch.Generator().manual_seed(0)
    m = torch.rand(3, 7, 77, generator=g) < 0.5
    bits = pack_bool_mask(m)
    assert bits.dtype == torch.int32 and bits.shape == (3, 7, 3)
    assert torch.equal(ops.unpack_bits(bits, 77), m)


def test_synthetic_batch_contract():
    b = synthetic.train_batch(2, 64, 96, num_classes=7, seed=3)
    assert len(b['gt_labels']) == 2 and b['gt_masks'][0].shape[1:] == (64, 96)
    assert b['gt_caption_ids'][0].shape == (35,) and int(b['gt_caption_ids'][0][0]) == 101
    assert int((b['gt_caption_ids'][0] == 102).sum()) == 1
    assert torch.equal(b['gt_caption_mask'][0], (b['gt_caption_ids'][0] != 0).long())


def test_build_optimizer_paramwise(detector):
    """configs/instance/coco_b48n17.py:270-285: backbone lr x0.1, embeddings / norms without weight decay."""
    from cgg_amd.train import build_optimizer
    _, m = detector
    embed_multi = dict(lr_mult=1.0, decay_mult=0.0)
    opt = build_optimizer(m, dict(type='AdamW', lr=1e-4, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999),
                                  paramwise_cfg=dict(custom_keys={'backbone': dict(lr_mult=0.1, decay_mult=1.0),
                                                                  'query_embed': embed_multi,
                                                                  'query_feat': embed_multi,
                                                                  'level_embed': embed_multi},
                                                     norm_decay_mult=0.0)))
    where = {}
    for g in opt.param_groups:
        for p in g['params']:
            where[id(p)] = (g['lr'], g['weight_decay'])
    names = dict(m.named_parameters())
    h = 'panoptic_head.'
    assert where[id(names['backbone.layer4.0.conv1.weight'])] == (1e-5, 0.05)
    assert where[id(names[h + 'query_embed.weight'])] == (1e-4, 0.0)
    assert where[id(names[h + 'level_embed.weight'])] == (1e-4, 0.0)
    assert where[id(names[h + 'transformer_decoder.post_norm.weight'])] == (1e-4, 0.0)
    assert where[id(names[h + 'mask_embed.0.weight'])] == (1e-4, 0.05)
    assert all(id(p) in where for p in m.parameters() if p.requires_grad)
    assert not any(id(p) in where for p in m.parameters() if not p.requires_grad)


def test_cast_cached_views_and_versions():
    """The low-precision weight cache is keyed by the slice's address, not by id() of a temporary view."""
    from cgg_amd import runtime
    w = torch.nn.Parameter(torch.arange(24, dtype=torch.float32).view(6, 4))
    for _ in range(50):                      # temporaries are freed and their ids recycled between calls
        a = runtime.cast_cached(w[:2])
        b = runtime.cast_cached(w[2:4])
        c = runtime.cast_cached(w[4:])
        assert torch.equal(a.float(), w[:2]) and torch.equal(b.float(), w[2:4]) and torch.equal(c.float(), w[4:])
    assert runtime.cast_cached(w[:2]) is a              # cached
    with torch.no_grad():
        w.add_(1.0)                                     # optimizer-style in-place update bumps the version
    assert torch.equal(runtime.cast_cached(w[:2]).float(), w[:2].detach())
    w2 = torch.nn.Parameter(torch.zeros(6, 4))
    assert torch.equal(runtime.cast_cached(w2[:2]).float(), torch.zeros(2, 4))


def test_swin_backbone_contract():
    """[3P] mmdet SwinTransformer restated for BASELINE configs[3]: output pyramid, upstream parameter names, window
    partition round trip, shifted-window mask blocks cross-region attention."""
    from cgg_amd.swin import ShiftWindowMSA
    bb = registry.build_backbone(dict(type='SwinTransformer', embed_dims=32, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16),
                                      window_size=7, mlp_ratio=4, out_indices=(0, 1, 2, 3), drop_path_rate=0.1,
                                      patch_norm=True))
    bb.init_weights()
    bb.eval()
    x = torch.randn(1, 3, 90, 128)           # not a multiple of the patch / window size: exercises the padding paths
    with torch.no_grad():
        outs = bb(x)
    assert [tuple(o.shape) for o in outs] == [(1, 32, 23, 32), (1, 64, 12, 16), (1, 128, 6, 8), (1, 256, 3, 4)]
    keys = set(bb.state_dict())
    for k in ['patch_embed.projection.weight', 'patch_embed.norm.weight',
              'stages.0.blocks.1.attn.w_msa.relative_position_bias_table',
              'stages.0.blocks.1.attn.w_msa.relative_position_index', 'stages.0.blocks.0.attn.w_msa.qkv.bias',
              'stages.1.blocks.0.attn.w_msa.proj.weight', 'stages.2.blocks.1.ffn.layers.0.0.weight',
              'stages.2.blocks.1.ffn.layers.1.bias', 'stages.0.downsample.norm.weight',
              'stages.0.downsample.reduction.weight', 'norm0.weight', 'norm3.bias']:
        assert k in keys, k
    assert 'stages.3.downsample.norm.weight' not in keys
    assert bb.state_dict()['stages.0.downsample.reduction.weight'].shape == (64, 128)
    assert bb.state_dict()['stages.1.blocks.0.attn.w_msa.relative_position_bias_table'].shape == (169, 4)
    idx = bb.state_dict()['stages.0.blocks.0.attn.w_msa.relative_position_index']
    assert idx.shape == (49, 49) and int(idx.min()) == 0 and int(idx.max()) == 168
    assert int(idx[0, 0]) == 84 and torch.equal(idx.diagonal(), torch.full((49,), 84))     # zero displacement
    msa = ShiftWindowMSA(32, 2, 7, shift_size=3)
    t = torch.randn(2, 14, 21, 32)
    assert torch.equal(msa._reverse(msa._partition(t), 14, 21), t)
    # a pure translation-equivariance check of the un-shifted block: permuting windows permutes the output
    blk = bb.stages[0].blocks[0].eval()
    tok = torch.randn(1, 14 * 14, 32)
    with torch.no_grad():
        y = blk(tok, (14, 14)).view(1, 14, 14, 32)
        tok2 = tok.view(1, 14, 14, 32).roll(7, dims=2).reshape(1, 196, 32)
        y2 = blk(tok2, (14, 14)).view(1, 14, 14, 32)
    assert torch.allclose(y.roll(7, dims=2), y2, atol=1e-5)


def test_collate_follows_datacontainer_rules():
    """data_contract.collate: stacked fields are padded at the end to the group maximum with their padding value,
    plain fields become per-sample lists, cpu-only ones stay objects; to_forward_kwargs unwraps one GPU group."""
    import numpy as np
    from cgg_amd.data_contract import OpenFormatBundle, collate, collect, to_forward_kwargs
    rng = np.random.RandomState(3)
    samples = []
    for (H, W, n) in [(6, 9, 2), (8, 7, 1), (5, 5, 3), (8, 9, 0)]:
        r = dict(img=rng.randint(0, 255, (H, W, 3)).astype(np.uint8), img_shape=(H, W, 3), ori_shape=(H, W, 3),
                 gt_bboxes=rng.rand(n, 4).astype(np.float32), gt_labels=rng.randint(0, 5, (n,)).astype(np.int64),
                 gt_masks=rng.randint(0, 2, (n, H, W)).astype(np.uint8),
                 gt_semantic_seg=rng.randint(0, 5, (H, W)).astype(np.uint8),
                 gt_caption_ids=rng.randint(0, 100, (35,)).astype(np.int64))
        r = OpenFormatBundle()(r)
        samples.append(collect(r, ['img', 'gt_bboxes', 'gt_labels', 'gt_masks', 'gt_semantic_seg', 'gt_caption_ids']))
    batch = collate(samples, samples_per_gpu=2)
    img = batch['img']
    assert img.stack and len(img.data) == 2
    assert tuple(img.data[0].shape) == (2, 3, 8, 9) and tuple(img.data[1].shape) == (2, 3, 8, 9)
    assert img.data[0].dtype == torch.float32
    assert torch.equal(img.data[0][0, :, :6, :9], samples[0]['img'].data) and float(img.data[0][0, :, 6:, :].abs().sum()) == 0
    seg = batch['gt_semantic_seg'].data[0]
    assert tuple(seg.shape) == (2, 1, 8, 9) and int(seg[1, 0, 0, 8]) == 255 and int(seg[0, 0, 7, 0]) == 255
    assert [tuple(t.shape) for t in batch['gt_bboxes'].data[1]] == [(3, 4), (0, 4)]
    assert batch['gt_masks'].cpu_only and batch['gt_masks'].data[0][1].shape == (1, 8, 7)
    assert batch['img_metas'].data[1][0]['img_shape'] == (5, 5, 3) and batch['img_metas'].data[0][0]['pad_shape'] == (6, 9, 3)
    kw = to_forward_kwargs(batch, device='cpu', group=1)
    assert tuple(kw['img'].shape) == (2, 3, 8, 9) and len(kw['gt_labels']) == 2 and len(kw['img_metas']) == 2
    assert kw['gt_masks'][0].dtype == torch.uint8 and tuple(kw['gt_masks'][0].shape) == (3, 5, 5)
    assert tuple(kw['gt_caption_ids'][1].shape) == (35,)


def test_checkpoint_roundtrip_and_key_compat(tmp_path):
    """checkpoint.load_checkpoint: mmcv-runner layout, DataParallel prefix, mismatch reporting."""
    from cgg_amd.checkpoint import load_checkpoint, load_state_dict, save_checkpoint
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(4, 2))
    other = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(4, 2))
    f = save_checkpoint(net, str(tmp_path / 'w' / 'latest.pth'), meta=dict(CLASSES=('a', 'b')))
    ck = load_checkpoint(other, f, strict=True)
    assert ck['meta']['CLASSES'] == ('a', 'b')
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), other.state_dict().values()))
    wrapped = {'state_dict': {'module.' + k: v + 1 for k, v in net.state_dict().items() if v.dtype.is_floating_point}}
    load_checkpoint(other, wrapped)
    assert torch.equal(other[0].weight, net[0].weight + 1)
    bad = dict(net.state_dict())
    bad['2.weight'] = torch.zeros(3, 4)
    bad['extra'] = torch.zeros(1)
    del bad['0.bias']
    msgs = []
    missing, unexpected, mismatched = load_state_dict(other, bad, logger=type('L', (), {'warning': staticmethod(msgs.append)}))
    assert missing == ['0.bias'] and unexpected == ['extra'] and mismatched[0][0] == '2.weight' and msgs
    with pytest.raises(RuntimeError):
        load_state_dict(other, bad, strict=True)


def test_cpp_linear_sum_assignment_equals_scipy():
    """cgg_linear_sum_assignment_f32 (host code in the C-ABI library) returns scipy's INDICES: random rectangular
    matrices both ways, heavy ties (integer costs), constant matrices, single rows / columns, +inf entries."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    from cgg_amd import ops
    rng = np.random.RandomState(0)
    mats = []
    for _ in range(60):
        nr, nc = rng.randint(1, 40), rng.randint(1, 40)
        mats.append(rng.randn(nr, nc).astype(np.float32))
    for _ in range(60):                                   # ties everywhere
        nr, nc = rng.randint(1, 25), rng.randint(1, 25)
        mats.append(rng.randint(0, 3, (nr, nc)).astype(np.float32))
    mats += [np.zeros((5, 9), np.float32), np.ones((9, 5), np.float32), np.zeros((1, 7), np.float32),
             np.zeros((7, 1), np.float32), rng.rand(100, 13).astype(np.float32), rng.rand(100, 1).astype(np.float32)]
    m = rng.rand(6, 8).astype(np.float32)
    m[m < 0.3] = np.inf                                   # forbidden pairs, still feasible
    m[:, 0] = 0.5
    mats.append(m)
    got = ops.linear_sum_assignment_batch([torch.from_numpy(x) for x in mats])
    for x, (r, c) in zip(mats, got):
        wr, wc = linear_sum_assignment(x)
        assert r.tolist() == wr.tolist() and c.tolist() == wc.tolist(), x.shape
    assert ops.linear_sum_assignment_batch([]) == []
    with pytest.raises(Exception):
        ops.linear_sum_assignment_batch([torch.tensor([[float('nan'), 1.0]])])
    with pytest.raises(Exception):
        ops.linear_sum_assignment_batch([torch.full((2, 2), float('inf'))])


def test_cfg_option_values():
    from cgg_amd.config import Config, parse_option_value
    assert parse_option_value('3') == 3 and parse_option_value('2e-4') == 2e-4 and parse_option_value('True') is True
    assert parse_option_value('[1,2]') == [1, 2] and parse_option_value('(0.9,0.999)') == [0.9, 0.999]
    assert parse_option_value('a,b') == ['a', 'b'] and parse_option_value('AdamW') == 'AdamW'
    cfg = Config(dict(optimizer=dict(lr=1e-4, betas=(0.9, 0.999)), model=dict(layers=[dict(k=1), dict(k=2)])))
    cfg.merge_from_dict({'optimizer.lr': parse_option_value('2e-4'), 'model.layers.1.k': parse_option_value('7')})
    assert cfg.optimizer.lr == 2e-4 and cfg.model.layers[1].k == 7


def test_test_driver_reinterleaves_rank_shards():
    """tools/test.py --launcher pytorch: rank r serves images r, r + world, ...; rank 0 restores dataset order."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('cgg_tools_test_cpu', os.path.join(root, 'tools', 'test.py'))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    for n, world in ((7, 2), (8, 2), (10, 4), (3, 4), (0, 2)):
        parts = [[i for i in range(n) if i % world == r] for r in range(world)]
        assert drv.interleave_rank_results(parts) == list(range(n))


def test_lr_schedule_step_with_linear_warmup():
    """configs/instance/coco_b48n17.py:289-297 semantics ([3P] mmcv StepLrUpdaterHook, by_epoch=False)."""
    from cgg_amd.train import LrSchedule
    w = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.AdamW([dict(params=[w[0]], lr=1e-4), dict(params=[w[1]], lr=1e-5)])
    sch = LrSchedule(opt, dict(policy='step', gamma=0.1, by_epoch=False, step=[70, 80], warmup='linear',
                               warmup_by_epoch=False, warmup_ratio=0.5, warmup_iters=10))
    want = {0: 0.5, 5: 0.75, 9: 0.95, 10: 1.0, 69: 1.0, 70: 0.1, 79: 0.1, 80: 0.01, 1000: 0.01}
    for it, f in want.items():
        sch.apply(it)
        assert abs(opt.param_groups[0]['lr'] - 1e-4 * f) < 1e-12 and abs(opt.param_groups[1]['lr'] - 1e-5 * f) < 1e-13, it
    # warmup_ratio = 1.0 (the shipped config: "no warmup") leaves the step policy alone
    sch = LrSchedule(opt, dict(policy='step', gamma=0.1, step=[3], warmup='linear', warmup_ratio=1.0, warmup_iters=10))
    sch.apply(0)
    assert abs(opt.param_groups[0]['lr'] - 1e-4) < 1e-12
    sch.apply(3)
    assert abs(opt.param_groups[0]['lr'] - 1e-5) < 1e-12
    # resumed optimizer state (decayed rates in the groups) does not change the base rates
    opt.param_groups[0]['lr'] = 123.0
    sch.apply(0)
    assert abs(opt.param_groups[0]['lr'] - 1e-4) < 1e-12


def _coco_rle_reference(mask):
    """Published COCO RLE (pycocotools maskApi.c rleEncode + rleToString) restated with plain Python loops: column-major
    runs starting with zeros; counts[i] delta-coded against counts[i-2] for i > 2; 5-bit groups, continuation bit 0x20,
    offset 48."""
    import numpy as np
    flat = np.asarray(mask, dtype=np.uint8).flatten(order='F')
    counts, p, c = [], 0, 0
    for v in flat.tolist():
        if v != p:
            counts.append(c)
            c, p = 0, v
        c += 1
    counts.append(c)
    s = bytearray()
    for i, cnt in enumerate(counts):
        x = cnt - counts[i - 2] if i > 2 else cnt
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            s.append(ch + 48)
    return bytes(s), counts


def test_rle_encode_bitmasks_equals_coco_reference():
    """cgg_rle_encode_bitmasks (host threads, 64x64 bit transposes) == the published COCO RLE on bit-packed masks:
    blobs, empty / full masks, a checkerboard (one run per pixel), a mask starting with a 1, sizes that are not
    multiples of 64 (H = 37, W = 80) and the 1024 x 1024 serving size."""
    import numpy as np
    rng = np.random.RandomState(3)
    for H, W in ((37, 80), (64, 64), (130, 192), (1024, 1024)):
        ys, xs = np.mgrid[0:H, 0:W]
        masks = [np.zeros((H, W), bool), np.ones((H, W), bool), (ys + xs) % 2 == 0, (ys + xs) % 2 == 1]
        for _ in range(4):
            cy, cx, r = rng.rand() * H, rng.rand() * W, 3 + rng.rand() * min(H, W) / 3
            masks.append((ys - cy) ** 2 + (xs - cx) ** 2 <= r * r)
        masks.append(rng.rand(H, W) < 0.3)
        if H * W > 200000:
            masks = masks[:2] + masks[4:8]                 # the Python reference is slow: no per-pixel patterns at 1024^2
        m = np.stack(masks)
        bits = np.packbits(m, axis=-1, bitorder='little')
        got = ops.rle_encode_bitmasks(torch.from_numpy(bits), W, threads=3)
        assert len(got) == len(masks)
        for g, mk in zip(got, masks):
            want, counts = _coco_rle_reference(mk)
            assert g['size'] == [H, W] and g['counts'] == want, (H, W, len(counts))
    assert ops.rle_encode_bitmasks(np.zeros((0, 8, 2), np.uint8), 16) == []


def test_rle_decoder_round_trip():
    import numpy as np
    from cgg_amd.host_results import rle_to_mask
    rng = np.random.RandomState(5)
    m = np.stack([rng.rand(50, 48) < 0.5, np.zeros((50, 48), bool), np.ones((50, 48), bool)])
    m[0, 0, 0] = True
    bits = np.packbits(m, axis=-1, bitorder='little')
    for g, mk in zip(ops.rle_encode_bitmasks(bits, 48, threads=1), m):
        assert np.array_equal(rle_to_mask(g), mk)


def test_splitk_linear_function_matches_autograd_on_cpu():
    """runtime._SplitKLinearFn (training linears of the encoder stream: dW as ONE batched GEMM over row slabs + an f32 sum) against
    plain autograd of the same bf16 linear on the CPU: forward identical, grad_input identical (same GEMM), grad_weight / grad_bias
    equal up to the bf16 rounding of the per-slab partial products (2^-8 relative to the gradient's scale)."""
    import torch
    import torch.nn.functional as F
    from cgg_amd import runtime
    g = torch.Generator().manual_seed(0)
    M, K, N = 32768, 64, 48                      # 32 768 rows -> 8 slabs of 4 096
    x = torch.randn(M, K, generator=g, requires_grad=True)
    w = (torch.randn(N, K, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    gy = torch.randn(M, N, generator=g).bfloat16()
    y = runtime._SplitKLinearFn.apply(x, w, b)
    y.backward(gy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    x16, w16, b16 = x.detach().bfloat16().requires_grad_(True), w.detach().bfloat16().requires_grad_(True), b.detach().bfloat16().requires_grad_(True)
    yr = F.linear(x16, w16, b16)
    yr.backward(gy)
    assert torch.equal(got[0], yr.detach())
    assert got[1].dtype == torch.float32 and torch.allclose(got[1], x16.grad.float(), atol=2e-2, rtol=2e-2)
    gw64 = gy.double().t() @ x.detach().bfloat16().double()
    assert (got[2].double() - gw64).abs().max().item() <= 2 ** -8 * gw64.abs().max().item() + 1e-3
    assert (got[3].double() - gy.double().sum(0)).abs().max().item() <= 1e-2
    assert got[2].dtype == torch.float32 and got[3].dtype == torch.float32


def test_batched_targets_cost_formulation_equals_oracle_on_cpu():
    """Host logic of `Mask2FormerHeadOpen._targets_batched` on CPU tensors (no kernel involved): the cost matrices built from the
    prediction-only halves -- sum_p softplus(x) - x . t for mmdet's pos . t + neg . (1 - t), one sigmoid for the dice numerator --
    against the oracle's per-(layer, image) `get_target_single` (reference: open_set/models/mask2former_head.py:320-390,
    assigners/mask_hungarian_assigner.py:100-143) with the same pinned random points: costs within 1e-5, labels equal; one image
    without ground truth."""
    import warnings
    from cgg_amd import synthetic
    from util import build_heads, small_cfg

    class Bank:
        def __init__(self, seed):
            self.seed, self.count = seed, {}

        def __call__(self, kind, shape, device):
            i = self.count.get(kind, 0)
            self.count[kind] = i + 1
            g = torch.Generator().manual_seed(self.seed + 1000 * i + {'target': 1, 'oversample': 2, 'random': 3}[kind])
            return torch.rand(*shape, generator=g).to(device)

    cfg = small_cfg(num_queries=12, num_points=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prod, orc = build_heads(cfg)
    prod.train()
    B, n, Q, h, H = 3, len(prod.transformer_decoder.layers) + 1, 12, 32, 128
    K1 = prod.class_embs.shape[0]
    g = torch.Generator().manual_seed(5)
    batch = synthetic.train_batch(B, H, H, num_classes=cfg['panoptic_head']['num_things_classes'], max_inst=4, vocab=500, seed=3)
    gt_labels, gt_masks = batch['gt_labels'], [m.long() for m in batch['gt_masks']]
    gt_labels[1], gt_masks[1] = gt_labels[1][:0], gt_masks[1][:0]
    cls = [torch.randn(B, Q, K1, generator=g) for _ in range(n)]
    emb = [torch.randn(B, Q, K1, generator=g) * 2 for _ in range(n)]
    masks = [torch.randn(B, Q, h, h, generator=g) * 2 for _ in range(n)]
    orc.point_hook = Bank(11)
    ref = [[orc.get_target_single(cls[li][b], emb[li][b], masks[li][b], gt_labels[b], gt_masks[b]) for b in range(B)]
           for li in range(n)]
    prod.point_hook = Bank(11)
    prod.cost_trace = []
    out = prod._targets_batched(cls, emb, masks, gt_labels, [m.float() for m in gt_masks])
    costs = dict(prod.cost_trace)
    prod.cost_trace = None
    for li in range(n):
        labels = out[li][0]
        for b in range(B):
            r_labels, r_cost = ref[li][b][0], ref[li][b][6]
            if gt_labels[b].numel() == 0:
                assert (labels[b] == prod.num_classes).all()
                continue
            assert torch.allclose(costs[b][li], r_cost, atol=1e-5, rtol=1e-5), (li, b)
            assert torch.equal(labels[b], r_labels), (li, b)


def test_channel_last_hand_over_is_dropped_after_in_place_writes():
    """`runtime.hand_nhwc` / `handed_nhwc` (frozen ResNet stages hand their channel-last f32 maps to the FPN rows path): the original
    is returned only for the very tensor it was attached to, and not after either tensor was written in place."""
    from cgg_amd import runtime
    f = torch.randn(2, 4, 5, 8)
    t = runtime.hand_nhwc(f.permute(0, 3, 1, 2).contiguous(), f)
    assert runtime.handed_nhwc(t) is f
    assert runtime.handed_nhwc(t.clone()) is None and runtime.handed_nhwc(torch.randn(2, 8, 4, 5)) is None
    t.add_(1)
    assert runtime.handed_nhwc(t) is None
    t = runtime.hand_nhwc(f.permute(0, 3, 1, 2).contiguous(), f)
    f.mul_(2)
    assert runtime.handed_nhwc(t) is None
    t = runtime.hand_nhwc(f.permute(0, 3, 1, 2).contiguous(), f)
    assert runtime.handed_nhwc(t.requires_grad_(True)) is None


def test_keepalive_scope_holds_what_the_caches_hand_out():
    """runtime.keepalive_scope (what pipeline.StagePipeline wraps its warm-up + capture in): every value `derived_cached` returns
    inside the scope -- a fresh entry or a hit -- is referenced by the scope's dict, once; outside a scope nothing is recorded; the
    value stays alive through the scope's references after its cache entry is gone."""
    import gc
    import weakref
    from cgg_amd import runtime
    w = torch.nn.Parameter(torch.randn(4, 4))
    outside = runtime.derived_cached('t_keepalive', (w,), lambda: w.detach() * 2)
    with runtime.keepalive_scope() as refs:
        a = runtime.derived_cached('t_keepalive', (w,), lambda: w.detach() * 2)       # a hit
        b = runtime.derived_cached('t_keepalive2', (w,), lambda: w.detach() * 3)      # a new entry
        a2 = runtime.derived_cached('t_keepalive', (w,), lambda: w.detach() * 2)
        assert a is outside and a2 is a and len(refs) == 2 and set(map(id, refs.values())) == {id(a), id(b)}
    c = runtime.derived_cached('t_keepalive3', (w,), lambda: w.detach() * 4)          # after the scope: not recorded
    assert len(refs) == 2 and not runtime._KEEPALIVE
    wr = weakref.ref(b)
    for k in [k for k in runtime._DCACHE if k[0] == 't_keepalive2']:
        del runtime._DCACHE[k]
    del b
    gc.collect()
    assert wr() is not None                                                           # the scope's dict still holds it
    refs.clear()
    gc.collect()
    assert wr() is None and c is not None


def test_round6_training_paths_are_gated_off_without_a_device():
    """The channel-last training paths of round 6 (`runtime.x3_resnet_stage_ok` / `input_level_x3_train`, the x3 point-logit sampler)
    never claim a CPU tensor, a stage that is not frozen-BatchNorm Bottlenecks, or a level without a handed-over channel-last map:
    the caller keeps the module path (the product has no CPU implementation; these are the host-side gates)."""
    import torch
    from cgg_amd import ops, runtime
    from cgg_amd.backbones import BasicBlock, Bottleneck
    from cgg_amd.pixel_decoder import ConvModule
    stage = torch.nn.Sequential(Bottleneck(64, 32, 1, torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, bias=False), torch.nn.BatchNorm2d(128))),
                                Bottleneck(128, 32)).eval()
    x = torch.randn(2, 8, 8, 64)
    with runtime.precision_scope('fp32'):
        assert not runtime.x3_resnet_stage_ok(stage, x)                                   # CPU tensor
        assert not runtime.x3_resnet_stage_ok(torch.nn.Sequential(BasicBlock(64, 64)).eval(), x)
        cm = ConvModule(64, 256, kernel_size=1, norm_cfg=dict(type='GN', num_groups=32), act_cfg=None, bias=True)
        assert runtime.input_level_x3_train(cm, torch.randn(2, 64, 8, 8)) is None          # CPU map, nothing handed over
    # the gates of the convolution + frozen-BatchNorm node: trainable filter, frozen affine BatchNorm in eval mode, 1 x 1 or 3 x 3 / pad 1
    blk = stage[1]
    assert runtime._x3_convbn_ok(blk.conv1, blk.bn1) is False                             # BatchNorm affine still requires grad
    for p in blk.bn1.parameters():
        p.requires_grad = False
    assert runtime._x3_convbn_ok(blk.conv1, blk.bn1) and not runtime._x3_convbn_ok(blk.conv1, blk.bn1.train())
    assert not runtime._x3_convbn_ok(torch.nn.Conv2d(64, 64, 3, padding=2, dilation=2, bias=False), blk.bn1.eval())
    assert not runtime._x3_convbn_ok(torch.nn.Conv2d(48, 64, 1, bias=False), blk.bn1)     # C % 32
    pts = torch.rand(2, 3 * 64, 2)
    assert not ops.point_sample_nhwc_x3_ok(torch.randn(2, 8, 8, 64), pts, 3)              # CPU tensor
