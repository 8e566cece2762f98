"""Oracle restatement of the reference's own model code on the hot path. TEST INFRASTRUCTURE ONLY.

  OracleHead        <- open_set/models/mask2former_head.py (Mask2FormerHeadOpen)
  OracleFusionHead  <- open_set/models/maskformer_fusion_head.py (MaskFormerFusionHeadOpen)
  grounding_loss    <- open_set/models/losses/grounding_loss.py:9-77
  CaptionTransformer<- open_set/models/transformers/{transformers,caption_tranformer}.py
  hungarian_assign  <- open_set/assigners/mask_hungarian_assigner.py:47-144
Each function cites the lines it follows. Plain torch on the CPU, seq-first, boolean masks repeated over
heads, F.interpolate / einsum / grid_sample -- i.e. the reference's formulation, not the product's.
PINNED by tests/golden/*.npz, which are produced by executing the reference's own files
(tests/golden/make_golden.py).
"""
import json
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import modules as M

_EPS32 = torch.finfo(torch.float32).eps
INSTANCE_OFFSET = 1000


# ---- [3P] helpers -------------------------------------------------------------------
def point_sample(inp, points):
    add = points.dim() == 3
    if add:
        points = points.unsqueeze(2)
    if inp.dtype != points.dtype:         # (a float64 run of the oracle, tests: the float32 GT masks follow the points' dtype)
        inp = inp.to(points.dtype)
    out = F.grid_sample(inp, 2.0 * points - 1.0, align_corners=False)
    return out.squeeze(3) if add else out


def weight_reduce(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    assert reduction == 'mean'
    return loss.sum() / (avg_factor + _EPS32)


def ce_loss(pred, label, weight=None, avg_factor=None, class_weight=None, ignore_index=None, loss_weight=1.0):
    """mmdet CrossEntropyLoss, softmax form (documented by losses/cross_entropy_loss.py:63-112)."""
    ii = -100 if ignore_index is None else ignore_index
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none', ignore_index=ii)
    return loss_weight * weight_reduce(loss, None if weight is None else weight.float(), 'mean', avg_factor)


def bce_loss(pred, label, avg_factor=None, loss_weight=1.0):
    """mmdet CrossEntropyLoss(use_sigmoid=True) on equal-shape pred/label (cross_entropy_loss.py:136-196)."""
    valid = ((label >= 0) & (label != -100)).float()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), reduction='none')
    return loss_weight * weight_reduce(loss, valid, 'mean', avg_factor)


def dice_loss(pred, target, avg_factor=None, eps=1.0, loss_weight=1.0):
    """mmdet DiceLoss(use_sigmoid, activate, naive_dice=True)."""
    x = pred.sigmoid().flatten(1)
    t = target.flatten(1).float()
    d = (2 * (x * t).sum(1) + eps) / (x.sum(1) + t.sum(1) + eps)
    return loss_weight * weight_reduce(1 - d, None, 'mean', avg_factor)


def match_cost(cls_emb_logit, mask_pts_pred, gt_labels, gt_pts_mask, w_emb=2.0, w_mask=5.0, w_dice=5.0,
               dice_eps=1.0, cls_pred=None, w_cls=0.0):
    """cost of mask_hungarian_assigner.py:100-123 with the mmdet match costs of SURVEY A8."""
    cost = 0
    if w_cls != 0 and cls_pred is not None:
        cost = cost + (-cls_pred.softmax(-1)[:, gt_labels] * w_cls)
    if w_emb != 0 and cls_emb_logit is not None:
        cost = cost + (-cls_emb_logit.softmax(-1)[:, gt_labels] * w_emb)
    x = mask_pts_pred.flatten(1).float()
    t = gt_pts_mask.flatten(1).float()
    if w_mask != 0:
        pos = F.binary_cross_entropy_with_logits(x, torch.ones_like(x), reduction='none')
        neg = F.binary_cross_entropy_with_logits(x, torch.zeros_like(x), reduction='none')
        cost = cost + (torch.einsum('nc,mc->nm', pos, t) + torch.einsum('nc,mc->nm', neg, 1 - t)) / x.shape[1] * w_mask
    if w_dice != 0:
        s = x.sigmoid()
        num = 2 * torch.einsum('nc,mc->nm', s, t)
        den = s.sum(-1)[:, None] + t.sum(-1)[None, :]
        cost = cost + (1 - (num + dice_eps) / (den + dice_eps)) * w_dice
    return cost


def hungarian_assign(cost, num_query, gt_labels):
    """mask_hungarian_assigner.py:126-144 -> (assigned_gt_inds (0 = bg, k = gt k-1), assigned_labels)."""
    gt_inds = torch.zeros(num_query, dtype=torch.long)
    labels = torch.full((num_query, ), -1, dtype=torch.long)
    if gt_labels.numel() == 0:
        return gt_inds, labels
    rows, cols = linear_sum_assignment(cost.detach().cpu())
    rows, cols = torch.from_numpy(rows), torch.from_numpy(cols)
    gt_inds[rows] = cols + 1
    labels[rows] = gt_labels[cols]
    return gt_inds, labels


def grounding_loss(cls_emb_pred, gt_caption_embs, gt_caption_mask, temperature):
    """losses/grounding_loss.py:9-77, literal (with the B^2 repeat)."""
    B, Q, d = cls_emb_pred.shape
    T = gt_caption_mask.shape[1]
    ntok = gt_caption_mask.sum(dim=1)
    p = cls_emb_pred[None].repeat(B, 1, 1, 1).reshape(B * B, Q, d)
    e = gt_caption_embs[:, None].repeat(1, B, 1, 1).reshape(B * B, T, d)
    m = gt_caption_mask[:, None].repeat(1, B, 1).reshape(B * B, T)
    nt = ntok[:, None].repeat(1, B).reshape(B * B)
    sim = torch.bmm(e, p.transpose(1, 2))
    dist = -sim / temperature
    sim = sim / temperature
    a_l2v = F.softmax(sim, dim=2) * m[:, :, None]
    a_v2l = F.softmax(sim, dim=1)
    g_l2v = (a_l2v * dist).sum(2).sum(1) / torch.max(nt, other=torch.ones_like(nt))
    g_v2l = (a_v2l * dist).sum(2).sum(1) / Q
    g_l2v = torch.where(nt > 0, g_l2v, g_l2v.max().detach() + 100.0)
    g_v2l = torch.where(nt > 0, g_v2l, g_v2l.max().detach() + 100.0)
    tot = 0.
    for g in (g_l2v, g_v2l):
        c = g.reshape(B, B)
        tot = tot + torch.diag(-torch.log_softmax(-c, dim=0)).mean() + torch.diag(-torch.log_softmax(-c, dim=1)).mean()
    return tot / 4


# ---- caption transformer (transformers.py / caption_tranformer.py) -----------------------------------
class _SelfAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.h, self.d = heads, dim // heads
        self.qkv_layer = nn.Linear(dim, 3 * dim)
        self.out_layer = nn.Linear(dim, dim)

    def forward(self, x, mask=None, kpm=None):
        B, L, _ = x.shape
        qkv = self.qkv_layer(x).reshape(B, L, self.h, 3 * self.d).permute(0, 2, 1, 3)
        q, k, v = torch.chunk(qkv, 3, dim=-1)          # per-head interleaved layout (transformers.py:114-117)
        w = q @ k.transpose(-2, -1) / np.sqrt(self.d)
        if mask is not None:
            w = w.masked_fill(mask, float('-inf'))
        if kpm is not None:
            w = w.masked_fill(kpm[:, None, None, :], float('-inf'))
        r = (torch.softmax(w, dim=-1) @ v).permute(0, 2, 1, 3).flatten(2)
        return self.out_layer(r)


class _CrossAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.h, self.d = heads, dim // heads
        self.to_qry, self.to_key = nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.to_val, self.to_out = nn.Linear(dim, dim), nn.Linear(dim, dim)

    def _r(self, s):
        B, L, _ = s.shape
        return s.reshape(B, L, self.h, self.d).permute(0, 2, 1, 3)

    def forward(self, q, k, v, mask=None, kpm=None):
        q, k, v = self._r(self.to_qry(q)), self._r(self.to_key(k)), self._r(self.to_val(v))
        w = q @ k.transpose(-2, -1) / np.sqrt(self.d)
        if mask is not None:
            w = w.masked_fill(mask, float('-inf'))
        if kpm is not None:
            w = w.masked_fill(kpm[:, None, None, :], float('-inf'))
        r = (torch.softmax(w, dim=-1) @ v).permute(0, 2, 1, 3).flatten(2)
        return self.to_out(r)


class _FFNc(nn.Module):
    def __init__(self, dim, ff, drop):
        super().__init__()
        self.linears = nn.ModuleList([
            nn.Sequential(nn.Linear(dim, ff), nn.Dropout(drop) if drop > 0 else nn.Identity(), nn.ReLU()),
            nn.Sequential(nn.Linear(ff, dim), nn.Identity(), nn.Identity())])

    def forward(self, x):
        return self.linears[1](self.linears[0](x))


class _DecBlock(nn.Module):
    def __init__(self, dim, ff, heads, drop, pre_norm):
        super().__init__()
        assert not pre_norm
        self.mha_layer, self.crx_layer, self.ffn_layer = _SelfAttn(dim, heads), _CrossAttn(dim, heads), _FFNc(dim, ff, drop)
        self.dropout_layer = nn.ModuleDict({k: nn.Dropout(drop) for k in ('mha', 'crx', 'ffn')})
        self.layer_normalz = nn.ModuleDict({k: nn.ModuleList([nn.Identity(), nn.LayerNorm(dim)])
                                            for k in ('mha', 'crx', 'ffn')})

    def forward(self, tgt, mem, tgt_mask, mem_mask, tgt_kpm, mem_kpm):
        n, d = self.layer_normalz, self.dropout_layer
        a = n['mha'][1](tgt + d['mha'](self.mha_layer(tgt, tgt_mask, tgt_kpm)))
        b = n['crx'][1](a + d['crx'](self.crx_layer(a, mem, mem, mem_mask, mem_kpm)))
        return n['ffn'][1](b + d['ffn'](self.ffn_layer(b)))


class _TDec(nn.Module):
    def __init__(self, n, dim, ff, heads, drop, pre_norm):
        super().__init__()
        self.decoders = nn.ModuleList([_DecBlock(dim, ff, heads, drop, pre_norm) for _ in range(n)])


class _PosEnc(nn.Module):
    def __init__(self, L, dim, drop=0.1):
        super().__init__()
        pos = np.arange(0, L)[:, None]
        idx = np.fromfunction(lambda _, j: j - j % 2, shape=(1, dim))
        even = np.fromfunction(lambda _, j: j % 2 == 0, shape=(1, dim))
        pnt = pos / (10000**(idx / dim))
        self.drop_layer = nn.Dropout(drop)
        self.register_buffer('psne_layer', torch.tensor(np.sin(pnt) * even + np.cos(pnt) * (1 - even)).float())


class CaptionTransformer(nn.Module):
    """caption_tranformer.py:17-44 -> (list of layer outputs, logits of the last layer)."""

    def __init__(self, nb_layers, input_dim, hidden_dim, ff_dim, nb_heads, drop_val, pre_norm, seq_length,
                 nb_tokens, type=None):
        super().__init__()
        self.adapter = nn.Linear(input_dim, hidden_dim) if input_dim != hidden_dim else nn.Identity()
        self.position_encoder = _PosEnc(seq_length, hidden_dim)
        self.transformer_decoder = _TDec(nb_layers, hidden_dim, ff_dim, nb_heads, drop_val, pre_norm)
        self.generator = nn.Linear(hidden_dim, nb_tokens)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        memory = self.adapter(memory)
        L = tgt.shape[1]
        tgt = self.position_encoder.drop_layer(tgt + self.position_encoder.psne_layer[:L][None])
        if tgt_mask is None:
            tgt_mask = torch.triu(torch.ones(L, L, dtype=torch.bool), 1)
        outs = []
        for blk in self.transformer_decoder.decoders:
            tgt = blk(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask)
            outs.append(tgt)
        return outs, self.generator(outs[-1])


class BertEmbeddings(nn.Module):
    """utils/bert_embeddings.py:4-14 (container only; weights are loaded from a state_dict)."""

    def __init__(self, vocab=30522, dim=768, eps=1e-12):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, dim, padding_idx=0)
        self.LayerNorm = nn.LayerNorm(dim, eps=eps)


# ---- the head ---------------------------------------------------------------------------------------
class OracleHead(nn.Module):

    def __init__(self, in_channels, feat_channels, out_channels, num_things_classes=80, num_stuff_classes=53,
                 num_queries=100, num_transformer_feat_level=3, pixel_decoder=None,
                 enforce_decoder_input_project=False, transformer_decoder=None, positional_encoding=None,
                 v2l_head=None, caption_generator=None, loss_cls=None, loss_cls_emb=None, loss_grounding=None,
                 loss_caption_generation=None, loss_caption_align=None, loss_mask=None, loss_dice=None,
                 train_cfg=None, test_cfg=None, init_cfg=None, type=None, **kw):
        super().__init__()
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.num_queries = num_queries
        self.num_transformer_feat_level = num_transformer_feat_level
        self.num_heads = transformer_decoder['transformerlayers']['attn_cfgs']['num_heads']
        self.num_transformer_decoder_layers = transformer_decoder['num_layers']
        pd = dict(pixel_decoder)
        pd.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = M.build_plugin_layer(pd)[1]
        self.transformer_decoder = M.build_transformer_layer_sequence(transformer_decoder)
        self.decoder_input_projs = nn.ModuleList([nn.Identity() for _ in range(num_transformer_feat_level)])
        assert self.transformer_decoder.embed_dims == feat_channels and not enforce_decoder_input_project
        self.decoder_positional_encoding = M.build_positional_encoding(positional_encoding)
        self.query_embed = nn.Embedding(num_queries, feat_channels)
        self.query_feat = nn.Embedding(num_queries, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, self.num_classes + 1)
        self.mask_embed = nn.Sequential(nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
                                        nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
                                        nn.Linear(feat_channels, out_channels))
        self.train_cfg, self.test_cfg = train_cfg, test_cfg or {}
        if train_cfg:
            self.num_points = train_cfg.get('num_points', 12544)
            self.oversample_ratio = train_cfg.get('oversample_ratio', 3.0)
            self.importance_sample_ratio = train_cfg.get('importance_sample_ratio', 0.75)
            a = train_cfg['assigner']
            self.cost_w = dict(w_cls=a['cls_cost']['weight'], w_emb=a['cls_emb_cost']['weight'],
                               w_mask=a['mask_cost']['weight'], w_dice=a['dice_cost']['weight'],
                               dice_eps=a['dice_cost'].get('eps', 1e-3))
        self.class_weight = loss_cls['class_weight']
        self.lw = dict(cls=loss_cls['loss_weight'], emb=(loss_cls_emb or {}).get('loss_weight', 0.),
                       ground=(loss_grounding or {}).get('loss_weight', 0.),
                       cap=(loss_caption_generation or {}).get('loss_weight', 0.),
                       cap_ignore=(loss_caption_generation or {}).get('ignore_index', None),
                       mask=loss_mask['loss_weight'], dice=loss_dice['loss_weight'], dice_eps=loss_dice.get('eps', 1e-3))
        g = kw.get
        self.use_class_emb, self.use_caption = g('use_class_emb', False), g('use_caption', False)
        self.use_caption_generation = g('use_caption_generation', False)
        self.softmax_temperature = g('softmax_temperature', 10.0)
        self.pred_emb_norm, self.text_emb_norm = g('pred_emb_norm', False), g('text_emb_norm', True)
        self.loss_only_last, self.loss_aux_weight = g('loss_only_last', False), g('loss_aux_weight', 1.0)
        if self.use_class_emb:
            known = open(kw['known_file']).read().split('\n') if g('known_file') else None
            unknown = open(kw['unknown_file']).read().split('\n') if g('unknown_file') else None
            table = json.load(open(kw['class_to_emb_file']))
            ce = torch.zeros((self.num_classes + 1, len(table[0]['emb'])))
            i = 0
            for c in table:                                   # mask2former_head.py:205-215
                if known and c['name'] not in known:
                    continue
                if unknown and c['name'] in unknown:
                    continue
                ce[i] = torch.FloatTensor(c['emb'])
                i += 1
            self.register_buffer('class_embs', ce)
            self.v2l_transform = nn.Linear(feat_channels, ce.shape[1])
        self.bert_embeddings = BertEmbeddings() if (self.use_caption or self.use_caption_generation) else None
        if self.use_caption_generation:
            self.caption_generator = CaptionTransformer(**dict(caption_generator))
        self.point_hook = None

    # -- forward (mask2former_head.py:711-849) --
    def forward_head(self, decoder_out, mask_feature, size):
        decoder_out = self.transformer_decoder.post_norm(decoder_out).transpose(0, 1)
        cls_pred = self.cls_embed(decoder_out)
        emb = cls_pred
        if self.use_class_emb:
            emb = self.v2l_transform(decoder_out)
            if self.pred_emb_norm:
                emb = emb / emb.norm(dim=-1, keepdim=True)
        mask_pred = torch.einsum('bqc,bchw->bqhw', self.mask_embed(decoder_out), mask_feature)
        am = F.interpolate(mask_pred, size, mode='bilinear', align_corners=False)
        if getattr(self, 'trace', None) is not None:      # test hook: resized logits + mask before the fix-up
            self.trace['attn_logits'].append(am.flatten(2).detach().clone())
        am = am.flatten(2).unsqueeze(1).repeat((1, self.num_heads, 1, 1)).flatten(0, 1)
        blocked = (am.sigmoid() < 0.5).detach()
        if getattr(self, 'inject', None):                 # test hook: the mask decisions of ANOTHER run of the same model (a float64 run
            blocked = self.inject.pop(0).to(blocked.device)   # that must follow the float32 run's near-zero threshold decisions)
        return cls_pred, emb, mask_pred, blocked

    def forward(self, feats, img_metas):
        B = len(img_metas)
        mask_features, mems = self.pixel_decoder(feats)
        dec_in, dec_pos = [], []
        for i in range(self.num_transformer_feat_level):
            x = self.decoder_input_projs[i](mems[i]).flatten(2).permute(2, 0, 1) + self.level_embed.weight[i].view(1, 1, -1)
            m = x.new_zeros((B, ) + mems[i].shape[-2:], dtype=torch.bool)
            dec_in.append(x)
            dec_pos.append(self.decoder_positional_encoding(m).flatten(2).permute(2, 0, 1))
        qf = self.query_feat.weight.unsqueeze(1).repeat((1, B, 1))
        qe = self.query_embed.weight.unsqueeze(1).repeat((1, B, 1))
        cls_l, emb_l, mask_l = [], [], []
        c, e, m, am = self.forward_head(qf, mask_features, mems[0].shape[-2:])
        cls_l.append(c), emb_l.append(e), mask_l.append(m)
        for i in range(self.num_transformer_decoder_layers):
            lvl = i % self.num_transformer_feat_level
            am[torch.where(am.sum(-1) == am.shape[-1])] = False
            qf = self.transformer_decoder.layers[i](query=qf, key=dec_in[lvl], value=dec_in[lvl], query_pos=qe,
                                                    key_pos=dec_pos[lvl], attn_masks=[am, None],
                                                    query_key_padding_mask=None, key_padding_mask=None)
            c, e, m, am = self.forward_head(qf, mask_features,
                                            mems[(i + 1) % self.num_transformer_feat_level].shape[-2:])
            cls_l.append(c), emb_l.append(e), mask_l.append(m)
        return cls_l, emb_l, mask_l

    # -- training (mask2former_head.py:273-629) --
    def _rand(self, kind, shape):
        return self.point_hook(kind, shape, 'cpu') if self.point_hook is not None else torch.rand(*shape)

    def cls_emb_logits(self, emb):
        return torch.matmul(emb, self.class_embs.t()) / self.softmax_temperature

    def get_target_single(self, cls_score, cls_emb_logit, mask_pred, gt_labels, gt_masks):
        Q, G = cls_score.shape[0], gt_labels.shape[0]
        pts = self._rand('target', (1, self.num_points, 2))
        mp = point_sample(mask_pred.unsqueeze(1), pts.repeat(Q, 1, 1)).squeeze(1)
        gp = point_sample(gt_masks.unsqueeze(1).float(), pts.repeat(G, 1, 1)).squeeze(1)
        if G == 0:
            gt_inds = torch.zeros(Q, dtype=torch.long)
            cost = None
        else:
            cost = match_cost(cls_emb_logit, mp, gt_labels, gp, cls_pred=cls_score, **self.cost_w)
            gt_inds, _ = hungarian_assign(cost, Q, gt_labels)
        pos = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        pos_gt = gt_inds[pos] - 1
        labels = gt_labels.new_full((self.num_queries, ), self.num_classes, dtype=torch.long)
        labels[pos] = gt_labels[pos_gt]
        label_weights = gt_labels.new_ones((self.num_queries, ))
        mask_weights = mask_pred.new_zeros((self.num_queries, ))
        mask_weights[pos] = 1.0
        return labels, label_weights, gt_masks[pos_gt], mask_weights, pos, neg, cost

    def uncertain_points(self, mask_pred):
        """mmdet get_uncertain_point_coords_with_randomness."""
        n = mask_pred.shape[0]
        ns = int(self.num_points * self.oversample_ratio)
        coords = self._rand('oversample', (n, ns, 2))
        unc = -torch.abs(point_sample(mask_pred, coords))
        nu = int(self.importance_sample_ratio * self.num_points)
        idx = torch.topk(unc[:, 0, :], k=nu, dim=1)[1] + ns * torch.arange(n, dtype=torch.long)[:, None]
        coords = coords.view(-1, 2)[idx.view(-1), :].view(n, nu, 2)
        nr = self.num_points - nu
        if nr > 0:
            coords = torch.cat((coords, self._rand('random', (n, nr, 2))), dim=1)
        return coords

    def loss_single(self, cls_scores, cls_emb_preds, mask_preds, gt_labels_list, gt_masks_list,
                    gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list,
                    gt_nouns_embs_list, gt_nouns_mask_list):
        B = cls_scores.size(0)
        emb_logits = self.cls_emb_logits(cls_emb_preds) if self.use_class_emb else None
        tg = [self.get_target_single(cls_scores[i], emb_logits[i] if emb_logits is not None else None,
                                     mask_preds[i], gt_labels_list[i], gt_masks_list[i]) for i in range(B)]
        labels = torch.stack([t[0] for t in tg]).flatten(0, 1)
        label_weights = torch.stack([t[1] for t in tg]).flatten(0, 1)
        mask_targets = torch.cat([t[2] for t in tg], dim=0)
        mask_weights = torch.stack([t[3] for t in tg])
        num_total_pos = sum(t[4].numel() for t in tg)
        cw = cls_scores.new_tensor(self.class_weight)
        avg = cw[labels].sum()
        loss_cls = ce_loss(cls_scores.flatten(0, 1), labels, label_weights, avg, cw, loss_weight=self.lw['cls'])
        zero = loss_cls.new_tensor(0.0)
        loss_emb = zero
        if self.use_class_emb:
            loss_emb = ce_loss(emb_logits.flatten(0, 1), labels, label_weights.float(), avg, cw,
                               loss_weight=self.lw['emb'])
        loss_ground = zero
        if self.use_caption:
            loss_ground = self.lw['ground'] * grounding_loss(
                cls_emb_preds, torch.stack(gt_nouns_embs_list), torch.stack(gt_nouns_mask_list),
                self.softmax_temperature)
        loss_cap = zero
        if self.use_caption_generation:
            ce_ = torch.stack(gt_caption_embs_list)
            cm = torch.stack(gt_caption_mask_list).bool()
            logits = self.caption_generator(tgt=ce_[:, :-1, :], memory=cls_emb_preds,
                                            tgt_key_padding_mask=torch.logical_not(cm[:, :-1]))[1].flatten(0, 1)
            ids = torch.stack(gt_caption_ids_list)[:, 1:].flatten(0, 1)
            loss_cap = ce_loss(logits, ids, ignore_index=self.lw['cap_ignore'], loss_weight=self.lw['cap'])
        num_total_masks = max(cls_scores.new_tensor([num_total_pos]), 1)
        mp = mask_preds[mask_weights > 0]
        if mask_targets.shape[0] == 0:
            return loss_cls, loss_emb, loss_ground, loss_cap, zero, mp.sum(), mp.sum()
        with torch.no_grad():
            pts = self.uncertain_points(mp.unsqueeze(1))
            tp = point_sample(mask_targets.unsqueeze(1).float(), pts).squeeze(1)
        pp = point_sample(mp.unsqueeze(1), pts).squeeze(1)
        loss_dice = dice_loss(pp, tp, avg_factor=num_total_masks, eps=self.lw['dice_eps'], loss_weight=self.lw['dice'])
        loss_mask = bce_loss(pp.reshape(-1), tp.reshape(-1), avg_factor=num_total_masks * self.num_points,
                             loss_weight=self.lw['mask'])
        return loss_cls, loss_emb, loss_ground, loss_cap, zero, loss_mask, loss_dice

    def word_embeddings(self, ids_list):
        out = []
        for ids in ids_list:
            e = self.bert_embeddings.word_embeddings(ids)
            out.append(self.bert_embeddings.LayerNorm(e) if self.text_emb_norm else e)
        return out

    def loss(self, all_cls, all_emb, all_mask, gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask):
        """mask2former_head.py:393-462 (+ :906-914 word embeddings)."""
        cap_embs = self.word_embeddings(cap_ids) if self.use_caption_generation else None
        noun_embs = self.word_embeddings(noun_ids) if self.use_caption else None
        names = ('loss_cls', 'loss_cls_emb', 'loss_grounding', 'loss_caption_generation',
                 'loss_caption_align', 'loss_mask', 'loss_dice')
        res = [self.loss_single(all_cls[i], all_emb[i], all_mask[i], gt_labels, gt_masks, cap_ids, cap_embs,
                                cap_mask, noun_embs, noun_mask) for i in range(len(all_cls))]
        out = {k: v for k, v in zip(names, res[-1])}
        if not self.loss_only_last:
            for li, r in enumerate(res[:-1]):
                for k, v in zip(names, r):
                    out[f'd{li}.{k}'] = v * self.loss_aux_weight
        return out

    # -- inference (mask2former_head.py:923-980) --
    def simple_test(self, feats, img_metas):
        c, e, m = self.forward(feats, img_metas)
        shp = img_metas[0]['batch_input_shape']
        up = F.interpolate(m[-1], size=(shp[0], shp[1]), mode='bilinear', align_corners=False)
        return c[-1], e[-1], up


# ---- fusion head (maskformer_fusion_head.py) --------------------------------------------------------
def mask2bbox(masks):
    """mmdet mask2bbox."""
    n = masks.shape[0]
    b = masks.new_zeros((n, 4), dtype=torch.float32)
    xa, ya = torch.any(masks, dim=1), torch.any(masks, dim=2)
    for i in range(n):
        x, y = torch.where(xa[i])[0], torch.where(ya[i])[0]
        if len(x) > 0 and len(y) > 0:
            b[i] = b.new_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1])
    return b


def cls_emb_scores(emb, gt_embs):
    return F.softmax(torch.matmul(emb, gt_embs.t()), -1)          # :297-315, no temperature


def crop_rescale(mask_pred, meta, rescale):
    """:414-425."""
    h, w = meta['img_shape'][:2]
    mask_pred = mask_pred[:, :h, :w]
    if rescale:
        oh, ow = meta['ori_shape'][:2]
        mask_pred = F.interpolate(mask_pred[:, None], size=(oh, ow), mode='bilinear', align_corners=False)[:, 0]
    return mask_pred


def instance_postprocess_emb(emb, mask_pred, gt_embs, max_per_image=100):
    """:317-366 -> labels, bboxes(n,5), masks, plus (query_indices, scores) for index-parity checks."""
    scores = cls_emb_scores(emb, gt_embs)[:, :-1]
    ncls = scores.shape[-1]
    labels = torch.arange(ncls).unsqueeze(0).repeat(emb.shape[0], 1).flatten(0, 1)
    sc, top = scores.flatten(0, 1).topk(max_per_image, sorted=False)
    lab = labels[top]
    qi = top // ncls
    mp = mask_pred[qi]
    binary = (mp > 0).float()
    ms = (mp.sigmoid() * binary).flatten(1).sum(1) / (binary.flatten(1).sum(1) + 1e-6)
    det = sc * ms
    binary = binary.bool()
    return lab, torch.cat([mask2bbox(binary), det[:, None]], dim=-1), binary, qi, sc


def instance_postprocess(mask_cls, mask_pred, num_classes, num_things, max_per_image=100):
    """maskformer_fusion_head.py:245-295 (closed-set variant on the classification logits, `use_class_emb=False`)."""
    Q = mask_cls.shape[0]
    scores = F.softmax(mask_cls, dim=-1)[:, :-1]
    labels = torch.arange(num_classes).unsqueeze(0).repeat(Q, 1).flatten(0, 1)
    sc, top = scores.flatten(0, 1).topk(max_per_image, sorted=False)
    lab = labels[top]
    mp = mask_pred[top // num_classes]
    thing = lab < num_things
    sc, lab, mp = sc[thing], lab[thing], mp[thing]
    binary = (mp > 0).float()
    ms = (mp.sigmoid() * binary).flatten(1).sum(1) / (binary.flatten(1).sum(1) + 1e-6)
    binary = binary.bool()
    return lab, torch.cat([mask2bbox(binary), (sc * ms)[:, None]], dim=-1), binary


def panoptic_postprocess(mask_cls, mask_pred, num_classes, num_things, object_mask_thr=0.8, iou_thr=0.8, filter_low_score=False):
    """maskformer_fusion_head.py:161-225 (stuff painted in query order, no area limit -- unlike the `_emb` variant)."""
    scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
    mask_pred = mask_pred.sigmoid()
    keep = labels.ne(num_classes) & (scores > object_mask_thr)
    cs, cc, cm = scores[keep], labels[keep], mask_pred[keep]
    h, w = mask_pred.shape[-2:]
    seg = torch.full((h, w), num_classes, dtype=torch.int32)
    if cm.shape[0] > 0:
        ids = (cs.view(-1, 1, 1) * cm).argmax(0)
        inst = 1
        for k in range(cc.shape[0]):
            pc = int(cc[k].item())
            mask = ids == k
            area = mask.sum().item()
            orig = (cm[k] >= 0.5).sum().item()
            if filter_low_score:
                mask = mask & (cm[k] >= 0.5)
            if area > 0 and orig > 0:
                if area / orig < iou_thr:
                    continue
                if pc >= num_things:
                    seg[mask] = pc
                else:
                    seg[mask] = pc + inst * INSTANCE_OFFSET
                    inst += 1
    return seg


def panoptic_postprocess_emb(emb, mask_pred, gt_embs, num_classes, num_things, object_mask_thr=0.8,
                             iou_thr=0.8, filter_low_score=False, stuff_area_limit=4096, debug=None):
    """:77-159. `debug` (a dict, tests only) receives the decision margins: per-pixel top1 - top2 probability, the
    winner's |sigmoid - 0.5|, per-query |score - thr|, per-segment |area ratio - iou_thr| and stuff |area - limit|."""
    scores, labels = cls_emb_scores(emb, gt_embs).max(-1)
    mask_pred = mask_pred.sigmoid()
    keep = labels.ne(num_classes) & (scores > object_mask_thr)
    cs, cc, cm = scores[keep], labels[keep], mask_pred[keep]
    prob = cs.view(-1, 1, 1) * cm
    h, w = mask_pred.shape[-2:]
    seg = torch.full((h, w), num_classes, dtype=torch.int32)
    stuff = []
    if debug is not None:
        debug.update(score_margin=(scores - object_mask_thr).abs(), kept=int(keep.sum()), ratio_margin=[], stuff_margin=[])
    if cm.shape[0] > 0:
        ids = prob.argmax(0)
        if debug is not None:
            top2 = prob.topk(min(2, prob.shape[0]), dim=0)[0]
            debug['pixel_margin'] = (top2[0] - top2[1]) if prob.shape[0] > 1 else torch.full_like(top2[0], float('inf'))
            debug['half_margin'] = (cm.gather(0, ids[None])[0] - 0.5).abs()
        inst = 1
        for k in range(cc.shape[0]):
            pc = int(cc[k].item())
            mask = ids == k
            area = mask.sum().item()
            orig = (cm[k] >= 0.5).sum().item()
            if filter_low_score:
                mask = mask & (cm[k] >= 0.5)
            if area > 0 and orig > 0:
                if debug is not None:
                    debug['ratio_margin'].append(abs(area / orig - iou_thr))
                if area / orig < iou_thr:
                    continue
                if pc >= num_things:
                    stuff.append(k)
                    continue
                seg[mask] = pc + inst * INSTANCE_OFFSET
                inst += 1
        for k in stuff:
            mask = (ids == k) & (seg == num_classes)
            if debug is not None:
                debug['stuff_margin'].append(abs(mask.sum().item() - stuff_area_limit))
            if mask.sum().item() < stuff_area_limit:
                continue
            seg[mask] = int(cc[k].item())
    return seg
