"""CPU tests (-m "not gpu"): the oracle -- and the product's pure-host logic (losses, assigner, caption
transformer, target / loss assembly) -- against the golden vectors produced by executing the reference's own
files (tests/golden/make_golden.py). This is what PINS the oracle."""
import copy
import json
import os
import warnings

import numpy as np
import pytest
import torch

import cgg_amd  # noqa: F401
from cgg_amd import caption_transformer as P_ct
from cgg_amd import losses as P_losses
from cgg_amd import registry
from cgg_amd.bert_embeddings import BertEmbeddings as P_Bert
from oracle import head as OH

from util import g4_inputs, g6_inputs, g7_inputs, head_cfg, randomize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def gold(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) if z[k].dtype.kind in 'fiub' and z[k].shape != () else z[k] for k in z.files}


# ---- G1 ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B', [1, 2, 4])
def test_g1_grounding_loss(B):
    z = gold('g1_grounding_loss.npz')
    want = float(z[f'loss{B}'])
    got_o = OH.grounding_loss(z[f'preds{B}'], z[f'embs{B}'], z[f'mask{B}'], 10.0)
    got_p = P_losses.grounding_loss(z[f'preds{B}'], z[f'embs{B}'], z[f'mask{B}'], 10.0)
    assert abs(float(got_o) - want) <= 1e-6 * max(1, abs(want))
    assert abs(float(got_p) - want) <= 2e-5 * max(1, abs(want))   # one fused contraction vs B^2 bmm's
    if B == 4:
        assert int(z['mask4'][2].sum()) == 0                      # the zero-noun caption edge case is in there


# ---- G2 ---------------------------------------------------------------------------------------------
def test_g2_caption_transformer_and_bert_embeddings():
    z = gold('g2_caption_transformer.npz')
    cfg = json.loads(str(z['cfg']))
    for cls in (OH.CaptionTransformer, P_ct.CaptionTransformer):
        m = cls(**cfg).eval()
        randomize(m, seed=int(z['seed']))
        assert torch.allclose(m.position_encoder.psne_layer, z['psne'], atol=1e-6)
        with torch.no_grad():
            outs, logits = m(tgt=z['tgt'], memory=z['mem'], tgt_key_padding_mask=z['kpm'].bool())
        assert (logits - z['logits']).abs().max().item() <= 1e-5, cls
        assert (outs[-1] - z['last']).abs().max().item() <= 1e-5
        assert (outs[0] - z['first']).abs().max().item() <= 1e-5
    be = P_Bert(None, vocab_size=100, hidden_size=32)
    with torch.no_grad():
        be.word_embeddings.weight.copy_(z['bert_table'])
        be.LayerNorm.weight.copy_(z['bert_ln_w'])
        be.LayerNorm.bias.copy_(z['bert_ln_b'])
        assert (be(z['bert_ids']) - z['bert_out']).abs().max().item() <= 1e-6
    sd = P_ct.CaptionTransformer(**cfg).state_dict()
    assert 'transformer_decoder.decoders.0.layer_normalz.mha.1.weight' in sd      # LN lives at index 1
    assert sd['transformer_decoder.decoders.0.mha_layer.qkv_layer.weight'].shape == (3 * 64, 64)


# ---- G3 / G4 ------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def oracle_head():
    cfg, B, H, W, feats, metas, qf, mf = g4_inputs()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        orc = OH.OracleHead(**head_cfg(cfg)).eval()
    randomize(orc, seed=0)
    return cfg, orc


def test_g4_oracle_head_forward_matches_reference(oracle_head):
    cfg, orc = oracle_head
    _, B, H, W, feats, metas, qf, mf = g4_inputs()
    z = gold('g4_head_forward.npz')
    with torch.no_grad():
        c, e, m = orc.forward(feats, metas)
        fc, fe, fm, fa = orc.forward_head(qf, mf, (4, 6))
    assert (torch.stack(c) - z['cls']).abs().max().item() <= 1e-5
    assert (torch.stack(e) - z['emb']).abs().max().item() <= 1e-5
    assert (torch.stack(m) - z['mask']).abs().max().item() <= 1e-4
    assert (fm - z['fh_mask']).abs().max().item() <= 1e-5
    assert (fc - z['fh_cls']).abs().max().item() <= 1e-6 and (fe - z['fh_emb']).abs().max().item() <= 1e-6
    assert torch.equal(fa, z['fh_attn'].bool())                   # (B*heads, Q, hw) attention mask, bit exact
    assert fa.shape == (2 * 8, 8, 24)


# ---- G5 / G6 ------------------------------------------------------------------------------------------
class Replay:
    """point_hook that replays the coordinates the reference drew (captured in the fixture)."""

    def __init__(self, draws):
        self.draws, self.i = draws, 0

    def __call__(self, kind, shape, device):
        d = self.draws[self.i]
        self.i += 1
        assert tuple(d.shape) == tuple(shape), (kind, d.shape, shape)
        return d.to(device)


def _product_head(cfg):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ph = registry.build_head(head_cfg(cfg))
    randomize(ph, seed=0)
    return ph


def test_g6_targets_and_losses_match_reference(oracle_head):
    cfg, orc = oracle_head
    _, B, H, W, feats, metas, _, _ = g4_inputs()
    z = gold('g6_loss_single.npz')
    g4 = gold('g4_head_forward.npz')
    li = int(z['layer'])
    cls, emb, mask = g4['cls'][li], g4['emb'][li], g4['mask'][li]
    gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask = g6_inputs(H, W)
    draws = [z[f'draw{i}'] for i in range(int(z['n_draws']))]
    want = z['losses']

    # oracle
    orc.point_hook = Replay(draws)
    orc.train()
    for mod in orc.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    with torch.no_grad():
        cap_embs, noun_embs = orc.word_embeddings(cap_ids), orc.word_embeddings(noun_ids)
        got = orc.loss_single(cls, emb, mask, gt_labels, gt_masks, cap_ids, cap_embs, cap_mask, noun_embs, noun_mask)
    got = torch.stack([g.reshape(()) for g in got])
    assert (got - want).abs().max().item() <= 1e-5 * (1 + want.abs().max().item()), (got, want)
    # one-image targets: indices bit exact
    orc.point_hook = Replay([z['t_points']])
    t = orc.get_target_single(cls[0], orc.cls_emb_logits(emb)[0], mask[0], gt_labels[0], gt_masks[0])
    assert torch.equal(t[0], z['t_labels']) and torch.equal(t[4], z['t_pos']) and torch.equal(t[5], z['t_neg'])
    assert torch.equal(t[3], z['t_mask_weights'])
    orc.point_hook = None
    orc.eval()

    # product host logic (torch ops only: assigner, costs, losses, caption head) on the same tensors
    ph = _product_head(cfg).train()
    for mod in ph.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    ph.point_hook = Replay(draws)
    with torch.no_grad():
        pe, _ = ph.extract_word_embeddings(cap_ids, cap_mask, 'bert')
        ne, _ = ph.extract_word_embeddings(noun_ids, noun_mask, 'bert')
        pl = ph.loss_single(cls, emb, mask, gt_labels, gt_masks, [c.clone() for c in cap_ids], pe, cap_mask, noun_ids,
                            ne, noun_mask, metas)
    pl = torch.stack([g.reshape(()) for g in pl])
    assert (pl - want).abs().max().item() <= 2e-5 * (1 + want.abs().max().item()), (pl, want)
    ph.point_hook = Replay([z['t_points']])
    pt = ph._get_target_single(cls[0], ph._get_cls_emb_logits(emb)[0], mask[0], gt_labels[0], gt_masks[0], metas)
    assert torch.equal(pt[0], z['t_labels']) and torch.equal(pt[4], z['t_pos']) and torch.equal(pt[5], z['t_neg'])


def test_loss_dict_keys_and_layer_batched_assignment(oracle_head):
    """loss(): same keys / values as per-layer loss_single, with ONE batched Hungarian transfer."""
    cfg, orc = oracle_head
    _, B, H, W, feats, metas, _, _ = g4_inputs()
    g4 = gold('g4_head_forward.npz')
    gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask = g6_inputs(H, W)
    ph = _product_head(cfg).train()
    for mod in ph.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    n = g4['cls'].shape[0]
    gen = torch.Generator().manual_seed(5)
    P = ph.num_points
    bank = {'target': [torch.rand(1, P, 2, generator=gen) for _ in range(n * B)],
            'oversample': [torch.rand(3 + 2, 3 * P, 2, generator=gen) for _ in range(n)],
            'random': [torch.rand(3 + 2, P - int(0.75 * P), 2, generator=gen) for _ in range(n)]}

    def hook_factory():
        idx = {k: 0 for k in bank}

        def hook(kind, shape, device):
            t = bank[kind][idx[kind]]
            idx[kind] += 1
            assert tuple(t.shape) == tuple(shape)
            return t
        return hook

    with torch.no_grad():
        ce, _ = ph.extract_word_embeddings(cap_ids, cap_mask, 'bert')
        ne, _ = ph.extract_word_embeddings(noun_ids, noun_mask, 'bert')
        ph.point_hook = hook_factory()
        d = ph.loss(list(g4['cls']), list(g4['emb']), list(g4['mask']), gt_labels, gt_masks, cap_ids, ce, cap_mask,
                    noun_ids, ne, noun_mask, metas)
        ph.point_hook = hook_factory()
        orc.point_hook = hook_factory()
        orc.train()
        for mod in orc.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        od = orc.loss(list(g4['cls']), list(g4['emb']), list(g4['mask']), gt_labels, gt_masks, cap_ids, cap_mask,
                      noun_ids, noun_mask)
        orc.point_hook = None
        orc.eval()
    names = ['loss_cls', 'loss_cls_emb', 'loss_grounding', 'loss_caption_generation', 'loss_caption_align',
             'loss_mask', 'loss_dice']
    assert list(d.keys())[:7] == names
    assert set(d.keys()) == set(names) | {f'd{i}.{k}' for i in range(n - 1) for k in names}
    for k in d:
        assert abs(float(d[k]) - float(od[k])) <= 2e-5 * (1 + abs(float(od[k]))), (k, float(d[k]), float(od[k]))


# ---- G7 ---------------------------------------------------------------------------------------------
def test_g7_postprocess_oracle_matches_reference(oracle_head):
    cfg, _ = oracle_head
    z = gold('g7_postprocess.npz')
    emb, mp, cls_embs, pemb = g7_inputs()
    fcfg = dict(cfg['panoptic_fusion_head'])
    fcfg.update(test_cfg=cfg['test_cfg'])
    fh = registry.build_head(fcfg)            # buffers (class embedding tables) are host logic
    for embs, suffix in ((fh.all_class_embs, ''), (fh.novel_class_embs, '_novel')):
        lab, box, msk, qi, sc = OH.instance_postprocess_emb(emb, mp, embs, 20)
        # top-k (sorted=False) order is unspecified: compare as sets keyed by (label, score)
        o1 = torch.argsort(box[:, 4] * 1e3 + lab, stable=True)
        o2 = torch.argsort(z['bboxes' + suffix][:, 4] * 1e3 + z['labels' + suffix], stable=True)
        assert torch.equal(lab[o1], z['labels' + suffix][o2])
        assert torch.allclose(box[o1], z['bboxes' + suffix][o2], atol=1e-6)
        assert torch.equal(msk[o1], z['masks' + suffix][o2].bool())
    pan = OH.panoptic_postprocess_emb(pemb, mp, cls_embs, 12, 8, 0.2, 0.5, True, 16)
    assert torch.equal(pan, z['pan_seg'].to(torch.int32))
    assert len(torch.unique(pan)) > 2


def test_caption_decode_step_equals_full_forward():
    """`CaptionTransformer.decode_step` (one new position against cached key / value prefixes, beams re-gathered by parent)
    == the last position of the full causal forward of the same sequences (transformers.py:187-267)."""
    from cgg_amd import registry
    torch.manual_seed(5)
    gen = registry.build_head(dict(type='CaptionTransformer', nb_layers=3, input_dim=48, hidden_dim=48, ff_dim=96, nb_heads=4,
                                   drop_val=0.1, pre_norm=False, seq_length=12, nb_tokens=50)).eval()
    mem = torch.randn(1, 9, 48)
    toks = torch.randn(5, 6, 48)                       # 5 beams, 6 embedded positions
    parents = [None, torch.tensor([0, 0, 0, 0, 0]), torch.tensor([0, 1, 2, 3, 4]), torch.tensor([4, 3, 3, 0, 1]),
               torch.tensor([2, 2, 1, 0, 4]), torch.tensor([1, 0, 4, 4, 3])]
    with torch.no_grad():
        state = gen.begin_decode(mem)
        seqs = toks[:1, :1]                              # the sequences as a full re-run would see them
        for t in range(6):
            if t == 0:
                new = toks[:1, :1]
            else:
                new = toks[:, t:t + 1]
                seqs = torch.cat([seqs.index_select(0, parents[t]) if seqs.shape[0] > 1 else seqs.expand(5, -1, -1), new], 1)
            outs = gen.decode_step(new, state, parents[t])
            full = gen(seqs, mem.expand(seqs.shape[0], -1, -1))[0]
            for a, b in zip(outs, full):
                assert (a - b[:, -1]).abs().max().item() <= 2e-5, t
    assert state['length'] == 6 and state['k'][0].shape[:2] == (5, 6)


# ---- G10: the no-class-embedding family (coco_ag_pretrain_3x.py:97-133) ------------------------------------------------
def test_g10_no_class_emb_head_targets_losses_and_closed_set_postprocess():
    """`use_class_emb=False, pred_emb_norm=True`, `loss_cls` weight 2.0, assigner `cls_cost` 2.0: oracle AND product host logic
    against the reference's own head / fusion head run by path (forward of all layers, loss_single with the captured points,
    one image's target indices bit exact, `instance_postprocess` / `panoptic_postprocess` of maskformer_fusion_head.py:161-295)."""
    from util import ag_cfg, g10_inputs
    z = gold('g10_no_class_emb.npz')
    cfg = ag_cfg(num_queries=8, vocab=120)
    _, B, H, W, feats, metas, _, _ = g4_inputs()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        orc = OH.OracleHead(**head_cfg(cfg)).eval()
        ph = registry.build_head(head_cfg(cfg))
    randomize(orc, seed=3)
    randomize(ph, seed=3)
    with torch.no_grad():
        c, e, m = orc.forward(feats, metas)
    assert (torch.stack(c) - z['cls']).abs().max().item() <= 1e-5 and (torch.stack(m) - z['mask']).abs().max().item() <= 1e-4
    assert (torch.stack(e) - z['emb']).abs().max().item() <= 1e-5
    li = int(z['layer'])
    cls, emb, mask = z['cls'][li], z['emb'][li], z['mask'][li]
    gt_labels, gt_masks = g6_inputs(H, W)[:2]
    draws = [z[f'draw{i}'] for i in range(int(z['n_draws']))]
    want = z['losses']
    assert float(want[0]) > 0 and float(want[1]) == 0          # loss_cls is live, loss_cls_emb off
    orc.train()
    orc.point_hook = Replay(draws)
    with torch.no_grad():
        got = orc.loss_single(cls, emb, mask, gt_labels, gt_masks, None, None, None, None, None)
    got = torch.stack([g.reshape(()) for g in got])
    assert (got - want).abs().max().item() <= 1e-5 * (1 + want.abs().max().item()), (got, want)
    orc.point_hook = Replay([z['t_points']])
    t = orc.get_target_single(cls[1], None, mask[1], gt_labels[1], gt_masks[1])
    assert torch.equal(t[0], z['t_labels']) and torch.equal(t[4], z['t_pos']) and torch.equal(t[5], z['t_neg'])
    # product host logic
    ph.train()
    ph.point_hook = Replay(draws)
    with torch.no_grad():
        pl = ph.loss_single(cls, emb, mask, gt_labels, gt_masks, None, None, None, None, None, None, metas)
    pl = torch.stack([g.reshape(()) for g in pl])
    assert (pl - want).abs().max().item() <= 2e-5 * (1 + want.abs().max().item()), (pl, want)
    ph.point_hook = Replay([z['t_points']])
    pt = ph._get_target_single(cls[1], None, mask[1], gt_labels[1], gt_masks[1], metas)
    assert torch.equal(pt[0], z['t_labels']) and torch.equal(pt[4], z['t_pos']) and torch.equal(pt[5], z['t_neg'])
    # closed-set post-processing (oracle restatement vs the reference)
    mcls, mpred = g10_inputs()
    nc = int(z['ins_num_classes'])
    lab, box, msk = OH.instance_postprocess(mcls[:, :nc + 1], mpred, nc, nc, 15)
    o1 = torch.argsort(box[:, 4] * 1e3 + lab, stable=True)
    o2 = torch.argsort(z['ins_bboxes'][:, 4] * 1e3 + z['ins_labels'], stable=True)
    assert torch.equal(lab[o1], z['ins_labels'][o2]) and torch.allclose(box[o1], z['ins_bboxes'][o2], atol=1e-6)
    assert torch.equal(msk[o1], z['ins_masks'][o2].bool())
    pan = OH.panoptic_postprocess(mcls, mpred, 12, 8, 0.3, 0.5, True)
    assert torch.equal(pan, z['pan_seg'].to(torch.int32)) and len(torch.unique(pan)) > 2


# ---- G8 -------------------------------------------------------------------------------------------------
class _StubTokenizer:
    def decode(self, ids):
        return ' '.join(str(int(i)) for i in ids)


def test_g8_beam_search_matches_reference():
    """cgg_amd.caption_search.beam_search == the reference's beam_search (run by path in make_golden.py) on the runs
    the reference completes (it crashes when a single live sequence remains; the product does not)."""
    import types
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
        head = types.SimpleNamespace(bert_embeddings=be, caption_generator=gen)
        got = beam_search(head, z[f'mem{case}'], 1, 2, max_len=max_len, beam_width=beam, tokenizer=_StubTokenizer())
        assert got == str(z[f'sentence{case}']), (case, got, str(z[f'sentence{case}']))
        ids = beam_search(head, z[f'mem{case}'], 1, 2, max_len=max_len, beam_width=beam, return_ids=True)
        # the incremental (key / value prefix) decode is the default; the full re-run of every sequence is the reference's shape
        assert ids == beam_search(head, z[f'mem{case}'], 1, 2, max_len=max_len, beam_width=beam, return_ids=True, kv_cache=False)
        assert ids[0] == 1 and ids[-1] == 2 and ' '.join(map(str, ids))[1:-1] == got


def test_g9_open_format_bundle_matches_reference():
    """G9: OpenFormatBundle (data contract on the input side) vs the reference's own class run by path."""
    import json
    from cgg_amd.data_contract import DataContainer, OpenFormatBundle
    z = np.load(os.path.join(GOLD, 'g9_format_bundle.npz'), allow_pickle=False)
    for case in range(int(z['n_cases'])):
        raw = {}
        meta = json.loads(str(z[f'c{case}_raw_meta']))
        for k in z.files:
            pre = f'c{case}_raw_'
            if k.startswith(pre) and k != pre + 'meta':
                raw[k[len(pre):]] = z[k]
        for k, v in meta.items():
            raw[k] = tuple(v) if k in ('ori_shape', 'img_shape', 'pad_shape') else v
        res = OpenFormatBundle()(dict(raw))
        keys = [k for k, v in res.items() if isinstance(v, DataContainer)]
        assert sorted(keys) == sorted(str(k) for k in z[f'c{case}_keys'])   # (dict order follows the raw sample's)
        for k in keys:
            v = res[k]
            want = z[f'c{case}_{k}_data']
            got = v.data.numpy() if torch.is_tensor(v.data) else np.asarray(v.data)
            assert got.shape == want.shape and np.array_equal(got, want), (case, k)
            if torch.is_tensor(v.data):
                assert str(v.data.dtype) == str(z[f'c{case}_{k}_dtype']), (case, k)
            assert [int(v.stack), int(v.padding_value), int(v.cpu_only), int(v.pad_dims)] == z[f'c{case}_{k}_attrs'].tolist()
        assert tuple(res['pad_shape']) == tuple(z[f'c{case}_pad_shape'].tolist())
        assert float(res['scale_factor']) == float(z[f'c{case}_scale_factor'])
        assert np.array_equal(res['img_norm_cfg']['mean'], z[f'c{case}_norm_mean'])
        assert np.array_equal(res['img_norm_cfg']['std'], z[f'c{case}_norm_std'])


def test_swin_product_module_equals_dense_oracle():
    """`cgg_amd.swin.SwinTransformer` (roll + window partition + SDPA + nn.Unfold merging) against `oracle.swin.OracleSwin`
    (dense per-pair window membership / region mask / relative-position lookup, explicit 2x2 gather, patch matmul): the
    two formulations share only the parameter tensors. Sizes that are NOT multiples of the patch / window size exercise
    the padding paths of both; depths (2, 2, 2, 2) put a shifted block and a merge in every stage."""
    import torch
    import cgg_amd  # noqa: F401
    from cgg_amd import registry
    from oracle.swin import OracleSwin
    from util import randomize
    kw = dict(embed_dims=32, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), window_size=7, mlp_ratio=4,
              out_indices=(0, 1, 2, 3), patch_norm=True)
    bb = registry.build_backbone(dict(type='SwinTransformer', drop_path_rate=0.1, **kw))
    randomize(bb, seed=3)
    with torch.no_grad():
        for n, p in bb.named_parameters():
            if n.endswith('relative_position_bias_table'):
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))))     # O(1) biases
    bb.eval()
    orc = OracleSwin(**kw)
    missing, unexpected = orc.load_state_dict(bb.state_dict(), strict=False)
    assert not missing and all(k.endswith('relative_position_index') for k in unexpected), (missing, unexpected)
    orc.eval()
    for shape in ((1, 3, 90, 128), (2, 3, 112, 84)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[2]))
        with torch.no_grad():
            want = orc(x)
            got = bb(x)
        assert len(want) == len(got) == 4
        for a, b in zip(got, want):
            assert a.shape == b.shape
            assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item()), (a - b).abs().max().item()


def test_g11_non_default_head_flags_match_reference():
    """Fixture G11 (tests/golden/make_golden.py, produced by the reference's own `loss_single` / `loss` / `init_weights`): the
    head branches no shipped config sets -- gen_only / gen_mask / gen_replace_obj_nouns (mask2former_head.py:562-580: the in-place
    edit of the caption ids the generator is trained on, and the caption-generation loss that follows), learnable_temperature
    (:228, :645), loss_only_last (:448), freeze_v2l (:242-244). Product host logic on the CPU."""
    import json
    cfg, B, H, W, feats, metas, _, _ = g4_inputs()
    z = gold('g11_head_flags.npz')
    g4 = gold('g4_head_forward.npz')
    g6 = gold('g6_loss_single.npz')
    li = int(z['layer'])
    cls, emb, mask = g4['cls'][li], g4['emb'][li], g4['mask'][li]
    gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask = g6_inputs(H, W)
    draws = [g6[f'draw{i}'] for i in range(int(g6['n_draws']))]

    def build(**flags):
        c = copy.deepcopy(cfg)
        c['panoptic_head'].update(flags)
        ph = _product_head(c).train()
        for mod in ph.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return ph

    for flag in ('gen_only_obj_nouns', 'gen_mask_obj_nouns', 'gen_replace_obj_nouns'):
        ph = build(**{flag: True})
        ids = [c.clone() for c in cap_ids]
        edited = ph._caption_targets(ids, noun_ids)
        assert torch.equal(edited, z[f'{flag}_ids']), flag                     # the reference's in-place edit, id for id
        assert torch.equal(torch.stack(ids), z[f'{flag}_ids'])                 # ... and in place, as upstream
        want = float(z[f'{flag}_loss'])
        if want == want:                                                       # (gen_replace: token 4874 exceeds the toy vocabulary)
            ph.point_hook = Replay(draws)
            with torch.no_grad():
                ids = [c.clone() for c in cap_ids]
                pe, _ = ph.extract_word_embeddings(ids, cap_mask, 'bert')
                ne, _ = ph.extract_word_embeddings(noun_ids, noun_mask, 'bert')
                pl = ph.loss_single(cls, emb, mask, gt_labels, gt_masks, ids, pe, cap_mask, noun_ids, ne, noun_mask, metas)
            assert abs(float(pl[3]) - want) <= 2e-5 * (1 + abs(want)), (flag, float(pl[3]), want)
    # learnable temperature + last-layer-only loss dict + frozen v2l
    c = copy.deepcopy(cfg)
    c['panoptic_head'].update(learnable_temperature=True, softmax_temperature=7.0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fresh = registry.build_head(head_cfg(c))                                   # as built: the temperature is a parameter = 7.0
    assert isinstance(fresh.softmax_temperature, torch.nn.Parameter) and fresh.softmax_temperature.requires_grad == bool(z['temperature_is_param'])
    assert torch.equal(fresh.softmax_temperature.detach().reshape(-1), z['temperature'])
    ph = build(learnable_temperature=True, softmax_temperature=7.0, loss_only_last=True, freeze_v2l=True)
    with torch.no_grad():
        ph.softmax_temperature.copy_(z['temperature'])                             # (`randomize` touched it with the other parameters)
        got = ph._get_cls_emb_logits(emb)
    assert (got - z['temp_logits']).abs().max().item() <= 1e-5 * (1 + z['temp_logits'].abs().max().item())
    ph.init_weights()
    randomize(ph, seed=0)
    frozen = sorted(n for n, p in ph.named_parameters() if not p.requires_grad and not n.startswith('bert') and 'class_embs' not in n)
    assert frozen == json.loads(str(z['frozen'])), (frozen, str(z['frozen']))
    ph.point_hook = None                       # (the key set does not depend on the random points)
    with torch.no_grad():
        ids = [c.clone() for c in cap_ids]
        pe, _ = ph.extract_word_embeddings(ids, cap_mask, 'bert')
        ne, _ = ph.extract_word_embeddings(noun_ids, noun_mask, 'bert')
        ph.force_reference_targets = True
        ld = ph.loss(list(g4['cls']), list(g4['emb']), list(g4['mask']), gt_labels, gt_masks, ids, pe, cap_mask, noun_ids, ne,
                     noun_mask, metas)
    assert sorted(ld.keys()) == json.loads(str(z['last_only_keys']))
