"""`Mask2FormerHeadOpen` -- the reference's head (open_set/models/mask2former_head.py:33-980) on the
MI355X kernels. Constructor signature, config keys, `state_dict` layout, method names and returned
structures follow the reference; the execution is re-designed:

  forward():  batch-first activations; mask_feature packed once per forward into the MFMA-operand
              image (+ three 2x2-pooled images); every decoder layer's attention mask comes from a
              GEMM over the POOLED image (interpolation is linear -> interp(E.F) == E.interp(F)),
              bit-packed and shared by the 8 heads; full-resolution mask logits are produced only
              for the outputs that are consumed (all 10 in training, the last one at test time);
              K/V projections are single GEMMs with a position-bias matrix.
  loss():     one batched device->host copy for all (layer x image) Hungarian problems, layer-invariant
              all_gathers hoisted, the 10 `reduce_mean` scalars folded into one all-reduce.

Line references in the method docstrings point at the reference implementation being restated.
"""
import copy
import os
import json
import warnings

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops, runtime
from .assigner import get_uncertain_point_coords_with_randomness, point_sample
from .bert_embeddings import BertEmbeddings
from .config import ConfigDict, to_config_dict
from .registry import (HEADS, build_assigner, build_head, build_loss, build_plugin_layer,
                       build_positional_encoding, build_sampler, build_transformer_layer_sequence)

# throughput mode: the decoder's per-level K / V projections as one HIP launch (ops.decoder_kv_proj); CGG_FUSED_KV=0 = library GEMMs
FUSED_KV = os.environ.get('CGG_FUSED_KV', '1') != '0'
# parity-mode training: the matching costs' point logits on the x3 einsum over sampler-written x3 images (CGG_POINT_LOGITS_X3=0: f32 rows + library bmm)
POINT_LOGITS_X3 = os.environ.get('CGG_POINT_LOGITS_X3', '1') != '0'
BOS_TOKEN = 101
EOS_TOKEN = 102


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _read_lines(path):
    with open(path, 'r', encoding='utf-8') as f:
        return f.read().split('\n')


class LowResMasks:
    """Mask logits kept at mask-feature resolution plus the size the reference would upsample them to
    (mask2former_head.py:957-964). `upsampled()` materialises the reference tensor (HIP bilinear)."""

    def __init__(self, logits, up_size):
        self.logits = logits
        self.up_size = (int(up_size[0]), int(up_size[1]))

    def upsampled(self):
        return ops.upsample_bilinear(self.logits.contiguous(), self.up_size)

    def __len__(self):
        return self.logits.shape[0]

    def __getitem__(self, i):
        return LowResMasks(self.logits[i], self.up_size)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


_PINNED, _COPY_STREAMS = {}, {}


def _pinned_like(t):
    """Pinned host view of t's shape / dtype on ONE growing buffer per dtype, reused across steps (the number of ground-truth
    instances changes every step; allocating pinned memory synchronises the device)."""
    n = t.numel()
    b = _PINNED.get(t.dtype)
    if b is None or b.numel() < n:
        cap = 1 << max(int(n - 1).bit_length(), 10)
        b = _PINNED[t.dtype] = torch.empty(cap, dtype=t.dtype, pin_memory=True)
    return b[:n].view(t.shape)


def _copy_stream(dev):
    s = _COPY_STREAMS.get(dev)
    if s is None:
        s = _COPY_STREAMS[dev] = torch.cuda.Stream(dev)
    return s


class LazyMasks:
    """Training-time stand-in for one layer's (B, Q, h, w) `mask_pred`: keeps the factors of the einsum
    (mask2former_head.py:748) instead of its 100-query result. The loss only ever reads mask logits (a) at 12 544 shared
    random points for the matching cost and (b) at full resolution for the handful of MATCHED queries, and sampling is
    linear -- sample(E F) = E sample(F) -- so neither needs the full tensor: (a) is one grid_sample of
    the mask feature + a (Q x C) x (C x P) product, (b) an einsum over the positives only. The forward / backward of
    the 10 full einsums (2 x 53.7 GFLOP f32 and 0.7 GB of gradient traffic per layer at configs[2]) disappears."""

    def __init__(self, mask_embed, mask_feature, packed=None):
        self.mask_embed, self.mask_feature = mask_embed, mask_feature        # (B, Q, C), (B, C, h, w)
        self.packed = packed          # PackedFeature of mask_feature: the positives' logits then run on the MFMA kernels

    @staticmethod
    def _contract(ep, mf, packed):
        """(B, n, C) x (B, C, h, w) -> (B, n, h*w): the einsum of mask2former_head.py:748 restricted to the matched queries."""
        if packed is not None and ep.is_cuda and ep.shape[-1] == 256 and ep.dtype == torch.float32:
            return _MaskLogitsFn.apply(ep, mf, packed).flatten(2)
        return torch.bmm(ep, mf.flatten(2).to(ep.dtype))

    @property
    def shape(self):
        B, Q, _ = self.mask_embed.shape
        return (B, Q) + tuple(self.mask_feature.shape[-2:])

    def sample_points(self, points):
        """points (B, P, 2) in [0, 1] -> logits of all queries at the points, (B, Q, P); no autograd."""
        with torch.no_grad():
            fs = point_sample(self.mask_feature.detach(), points)             # (B, C, P)
            return torch.bmm(self.mask_embed.detach().float(), fs.float())

    def select(self, pos):
        """`mask_pred[weights > 0]` (mask2former_head.py:597) for the positives described by `pos` (device index tensors
        b / q / r = slot inside the image, host int pm = max positives per image): full-resolution logits (n_pos, h, w),
        image-major like boolean indexing; differentiable w.r.t. mask_embed and mask_feature. The positives are packed
        into a zero-padded (B, pm, C) operand so that ONE batched product serves the whole layer."""
        pre = getattr(self, '_pre', None)
        if pre is not None and pre[0] is pos:
            return pre[1]
        B, Q, C = self.mask_embed.shape
        h, w = self.mask_feature.shape[-2:]
        pm = int(pos['pm'])
        if pm == 0:
            return self.mask_embed.new_zeros((0, h, w)) + 0 * self.mask_embed.sum() + 0 * self.mask_feature.sum()
        bi, qi, ri = pos['b'], pos['q'], pos['r']
        ep = self.mask_embed.new_zeros((B, pm, C)).index_put((bi, ri), self.mask_embed[bi, qi])
        full = self._contract(ep, self.mask_feature, self.packed)              # (B, pm, h*w)
        return full[bi, ri].view(-1, h, w)

    @staticmethod
    def preselect(lazies, pos_list):
        """`select` of ALL decoder layers as one batched product: the layers share the mask feature, so their zero-padded
        positive embeddings are concatenated along the slot axis -- (B, sum pm_l, C) x (C, h*w) -- instead of 10 skinny
        products with 10-20 rows each, and autograd produces ONE mask-feature gradient instead of accumulating ten 1-GB
        ones. Each layer's `select(pos)` then returns its slice."""
        mf = lazies[0].mask_feature
        if not all(l.mask_feature is mf for l in lazies):
            return
        B, _, C = lazies[0].mask_embed.shape
        h, w = mf.shape[-2:]
        eps, offs, off = [], [], 0
        for lz, pos in zip(lazies, pos_list):
            pm = int(pos['pm'])
            offs.append(off)
            if pm:
                eps.append(lz.mask_embed.new_zeros((B, pm, C)).index_put((pos['b'], pos['r']), lz.mask_embed[pos['b'], pos['q']]))
                off += pm
        if off == 0:
            return
        full = LazyMasks._contract(torch.cat(eps, 1), mf, lazies[0].packed)              # (B, sum pm, h*w)
        # ONE gather for the positives of all layers, then per-layer views by `split`: autograd then builds ONE zero-filled
        # (B, sum pm, h*w) gradient and one cat of the layers' (n_pos, h*w) gradients -- indexing `full` once per layer made every
        # layer's backward a full-size zero tensor + scatter and nine full-size adds (839 MB each at configs[2]: 5 ms per step)
        live = [(lz, pos, o) for lz, pos, o in zip(lazies, pos_list, offs) if int(pos['pm'])]
        bi = torch.cat([pos['b'] for _, pos, _ in live])
        ri = torch.cat([o + pos['r'] for _, pos, o in live])
        rows = full[bi, ri]                                                              # (sum n_pos, h*w)
        for (lz, pos, _), part in zip(live, rows.split([int(pos['b'].numel()) for _, pos, _ in live], 0)):
            lz._pre = (pos, part.view(-1, h, w))

    def select_by_weights(self, weights):
        """same from a (B, Q) weight map (host round trip for the index set)."""
        sel = weights > 0
        bi, qi = torch.nonzero(sel, as_tuple=True)
        rank = (torch.cumsum(sel.long(), 1) - 1)[bi, qi]
        pm = int(sel.sum(1).max()) if sel.numel() else 0
        return self.select(dict(b=bi, q=qi, r=rank, pm=pm))


class _MaskLogitsFn(torch.autograd.Function):
    """einsum('bqc,bchw->bqhw') on the HIP MFMA kernels, forward (`cgg_mask_logits` on the packed feature) and backward
    (`cgg_mask_logits_backward`: the two transposed contractions; shapes outside the kernels' limits keep torch.einsum)."""

    @staticmethod
    def forward(ctx, embed, feat, packed):
        embed = embed.contiguous()
        ctx.save_for_backward(embed, feat)
        ctx.split = packed.lo is not None
        out = torch.cat([ops.mask_logits(embed[:, s:s + 256].contiguous(), packed, want_logits=True)[0]
                         for s in range(0, embed.shape[1], 256)], 1) if embed.shape[1] > 256 else \
            ops.mask_logits(embed, packed, want_logits=True)[0]
        return out

    @staticmethod
    def backward(ctx, go):
        embed, feat = ctx.saved_tensors
        if ops.mask_logits_backward_ok(embed, feat):
            g_e, g_f = ops.mask_logits_backward(embed, feat, go, ctx.split, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            return g_e, g_f, None
        g_e = torch.einsum('bqhw,bchw->bqc', go, feat)
        g_f = torch.einsum('bqc,bqhw->bchw', embed, go)
        return g_e, g_f, None


@HEADS.register_module()
class Mask2FormerHeadOpen(nn.Module):

    def __init__(self, in_channels, feat_channels, out_channels, num_things_classes=80,
                 num_stuff_classes=53, num_queries=100, num_transformer_feat_level=3, pixel_decoder=None,
                 enforce_decoder_input_project=False, transformer_decoder=None, positional_encoding=None,
                 v2l_head=None, caption_generator=None, loss_cls=None, loss_cls_emb=None,
                 loss_grounding=None, loss_caption_generation=None, loss_caption_align=None,
                 loss_mask=None, loss_dice=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        pixel_decoder = to_config_dict(pixel_decoder)
        transformer_decoder = to_config_dict(transformer_decoder)
        self.num_things_classes = num_things_classes
        self.num_stuff_classes = num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.num_queries = num_queries
        self.num_transformer_feat_level = num_transformer_feat_level
        self.num_heads = transformer_decoder.transformerlayers.attn_cfgs.num_heads
        self.num_transformer_decoder_layers = transformer_decoder.num_layers
        assert pixel_decoder.encoder.transformerlayers.attn_cfgs.num_levels == num_transformer_feat_level
        pixel_decoder_ = copy.deepcopy(pixel_decoder)
        pixel_decoder_.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = build_plugin_layer(pixel_decoder_)[1]
        self.transformer_decoder = build_transformer_layer_sequence(transformer_decoder)
        self.decoder_embed_dims = self.transformer_decoder.embed_dims
        self.decoder_input_projs = nn.ModuleList()
        for _ in range(num_transformer_feat_level):
            if self.decoder_embed_dims != feat_channels or enforce_decoder_input_project:
                self.decoder_input_projs.append(nn.Conv2d(feat_channels, self.decoder_embed_dims, kernel_size=1))
            else:
                self.decoder_input_projs.append(nn.Identity())
        self.decoder_positional_encoding = build_positional_encoding(positional_encoding)
        self.query_embed = nn.Embedding(self.num_queries, feat_channels)
        self.query_feat = nn.Embedding(self.num_queries, feat_channels)
        self.level_embed = nn.Embedding(self.num_transformer_feat_level, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, self.num_classes + 1)
        self.mask_embed = nn.Sequential(
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, out_channels))
        self.feat_channels = feat_channels
        self.v2l_head_cfg = v2l_head
        self.caption_generator_cfg = caption_generator
        self.test_cfg = to_config_dict(test_cfg) if test_cfg is not None else test_cfg
        self.train_cfg = to_config_dict(train_cfg) if train_cfg is not None else train_cfg
        if train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            self.sampler = build_sampler(self.train_cfg.sampler, context=self)
            self.num_points = self.train_cfg.get('num_points', 12544)
            self.oversample_ratio = self.train_cfg.get('oversample_ratio', 3.0)
            self.importance_sample_ratio = self.train_cfg.get('importance_sample_ratio', 0.75)
        self.class_weight = loss_cls['class_weight']
        self.loss_cls = build_loss(loss_cls)
        if loss_cls_emb is not None:
            self.loss_cls_emb = build_loss(loss_cls_emb)
        if loss_grounding is not None:
            self.loss_grounding = build_loss(loss_grounding)
        if loss_caption_generation is not None:
            self.loss_caption_generation = build_loss(loss_caption_generation)
        if loss_caption_align is not None:
            self.loss_caption_align = build_loss(loss_caption_align)
        self.loss_mask = build_loss(loss_mask)
        self.loss_dice = build_loss(loss_dice)
        self.point_hook = None  # callable(kind, shape, device) -> coords; pins the random draws in tests
        self.attn_mask_hook = None  # callable(layer_idx, bits) -> bits; tests inject the oracle's masks
        # callable(layer_idx, image_idx, rows, cols, cost) -> (rows, cols); tests check the Hungarian solution against the oracle's
        # cost matrix and inject the oracle's own (a near-tie may legitimately resolve either way in float32)
        self.assign_hook = None
        self.init_kwargs(**kwargs)

    # ------------------------------------------------------------------------------------------
    def init_kwargs(self, **kwargs):
        """mask2former_head.py:175-229."""
        self.kwargs = kwargs
        g = kwargs.get
        self.class_agnostic = g('class_agnostic', False)
        self.use_class_emb = g('use_class_emb', False)
        self.use_caption = g('use_caption', False)
        self.use_caption_generation = g('use_caption_generation', False)
        self.use_caption_align = g('use_caption_align', False)
        self.known_file = g('known_file', None)
        self.unknown_file = g('unknown_file', None)
        self.softmax_temperature = g('softmax_temperature', 10.0)
        self.learnable_temperature = g('learnable_temperature', False)
        self.pred_emb_norm = g('pred_emb_norm', False)
        self.text_emb_norm = g('text_emb_norm', True)
        self.freeze_pretrained = g('freeze_pretrained', False)
        self.freeze_v2l = g('freeze_v2l', False)
        self.loss_only_last = g('loss_only_last', False)
        self.loss_aux_weight = g('loss_aux_weight', 1.0)
        self.gen_only_obj_nouns = g('gen_only_obj_nouns', False)
        self.gen_mask_obj_nouns = g('gen_mask_obj_nouns', False)
        self.gen_replace_obj_nouns = g('gen_replace_obj_nouns', False)
        if self.known_file is not None:
            self.known_cat_names = _read_lines(self.known_file)
        if self.unknown_file is not None:
            self.unknown_cat_names = _read_lines(self.unknown_file)
        if self.use_class_emb:
            with open(kwargs['class_to_emb_file'], 'r') as f:
                class_to_emb = json.load(f)
            class_embs = torch.zeros((self.num_classes + 1, len(class_to_emb[0]['emb'])), dtype=torch.float)
            i = 0
            for class_dict in class_to_emb:
                if self.known_file and class_dict['name'] not in self.known_cat_names:
                    continue
                if self.unknown_file and class_dict['name'] in self.unknown_cat_names:
                    continue
                class_embs[i, :] = torch.FloatTensor(class_dict['emb'])
                i += 1
            self.register_buffer('class_embs', class_embs)
            self.v2l_transform = nn.Linear(self.feat_channels, class_embs.shape[1])
        self.bert_embeddings = self.clip = None
        if self.use_caption:
            self.caption_emb_type = g('caption_emb_type', 'clip')
            self.build_text_encoders(self.caption_emb_type)
        if self.use_caption_generation:
            self.caption_gen_emb_type = g('caption_gen_emb_type', 'bert')
            self.caption_generator = build_head(self.caption_generator_cfg)
            self.build_text_encoders(self.caption_gen_emb_type)
        if self.learnable_temperature:
            self.softmax_temperature = nn.Parameter(torch.tensor([self.softmax_temperature]), requires_grad=True)

    def init_weights(self):
        """mask2former_head.py:231-247."""
        for m in self.decoder_input_projs:
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1, mode='fan_in', nonlinearity='leaky_relu')
                nn.init.constant_(m.bias, 0)
        self.pixel_decoder.init_weights()
        for p in self.transformer_decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        if self.freeze_v2l:
            for p in self.v2l_transform.parameters():
                p.requires_grad = False
        if self.freeze_pretrained:
            self.freeze_params()

    def build_text_encoders(self, emb_type):
        """mask2former_head.py:249-260. BERT weights come from HF `bert-base-uncased` when they are
        available locally; offline a synthetic table with the same shapes is used (benchmarks)."""
        if emb_type == 'bert' and self.bert_embeddings is None:
            bert = None
            if not self.kwargs.get('synthetic_text_encoder', False):
                try:
                    import transformers
                    bert = transformers.BertModel.from_pretrained('bert-base-uncased', local_files_only=True).eval()
                except Exception as e:  # offline container / GPU box
                    warnings.warn(f'bert-base-uncased weights unavailable ({type(e).__name__}); using a '
                                  'synthetic embedding table of the same shape')
            self.bert_embeddings = BertEmbeddings(bert) if bert is not None else BertEmbeddings.synthetic()
            for p in self.bert_embeddings.parameters():
                p.requires_grad = False
        if emb_type == 'clip' and self.clip is None:
            raise NotImplementedError("caption_emb_type='clip' needs the `clip` package (not shipped); "
                                      "every reference config uses 'bert'")

    def freeze_params(self):
        self.decoder_input_projs.eval()
        self.pixel_decoder.eval()
        self.transformer_decoder.eval()
        for mod in (self.decoder_input_projs, self.pixel_decoder, self.transformer_decoder):
            for p in mod.parameters():
                p.requires_grad = False

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward_head(self, decoder_out, mask_feature, attn_mask_target_size, packed=None, pooled=None,
                     want_mask=True, want_attn=True):
        """mask2former_head.py:711-761 on batch-first `decoder_out` (B,Q,C).
        Returns (cls_pred, cls_emb_pred, mask_pred | None, attn bits (B,Q,words) int32 | None)."""
        from .query_decoder import residual_layernorm, small_linear
        decoder_out = residual_layernorm(decoder_out.contiguous(), None, self.transformer_decoder.post_norm)
        cls_pred = small_linear(decoder_out, self.cls_embed.weight, self.cls_embed.bias)
        cls_emb_pred = cls_pred
        if self.use_class_emb:
            cls_emb_pred = small_linear(decoder_out, self.v2l_transform.weight, self.v2l_transform.bias)
            if self.pred_emb_norm:
                cls_emb_pred = cls_emb_pred / cls_emb_pred.norm(dim=-1, keepdim=True)
        me = self.mask_embed
        h = small_linear(decoder_out, me[0].weight, me[0].bias, relu=True)
        h = small_linear(h, me[2].weight, me[2].bias, relu=True)
        mask_embed = small_linear(h, me[4].weight, me[4].bias).contiguous()
        split = not runtime.is_bf16()
        mask_pred = None
        if want_mask:
            if packed is None:
                packed = ops.pack_mask_feature(mask_feature.detach().contiguous(), 1, split)
            if mask_feature is not None and torch.is_grad_enabled() and (mask_embed.requires_grad or
                                                                         mask_feature.requires_grad):
                if getattr(self, '_lazy_masks', False):
                    mask_pred = LazyMasks(mask_embed, mask_feature, packed)
                else:
                    mask_pred = _MaskLogitsFn.apply(mask_embed, mask_feature, packed)
            else:
                mask_pred, _ = ops.mask_logits(mask_embed, packed, want_logits=True)
        bits = None
        if want_attn:
            H, W = (packed.h, packed.w) if mask_feature is None else mask_feature.shape[-2:]
            h, w = int(attn_mask_target_size[0]), int(attn_mask_target_size[1])
            s = H // h if h > 0 else 0
            if h * s == H and w * s == W and s in (2, 4, 8):
                if pooled is None:
                    pooled = ops.pack_mask_feature(mask_feature.detach().contiguous(), s, split)
                _, bits = ops.mask_logits(mask_embed.detach(), pooled, want_logits=False, want_bits=True)
            else:  # generic size: resize the stored logits (still no x num_heads repeat)
                if mask_pred is None or isinstance(mask_pred, LazyMasks):
                    if packed is None:
                        packed = ops.pack_mask_feature(mask_feature.detach().contiguous(), 1, split)
                    full, _ = ops.mask_logits(mask_embed.detach(), packed, want_logits=True)
                else:
                    full = mask_pred.detach()
                bits = ops.attn_mask_from_logits(full.contiguous(), (h, w))
        return cls_pred, cls_emb_pred, mask_pred, bits

    def _head_stream(self, d, sizes_next, packed_full, pooled_next, want_mask, want_attn, last):
        """forward_head (mask2former_head.py:711-761) on post-normed rows d (M = B*Q, C): cls_embed, v2l_transform and
        mask_embed[0] run as ONE GEMM over concatenated packed weights (ReLU on the mask-MLP columns only)."""
        B, Q = packed_full.B, d.shape[0] // packed_full.B
        me = self.mask_embed
        lr = ops.linear_rows_bf16
        ws = (me[0].weight,) + ((self.v2l_transform.weight,) if self.use_class_emb else ()) + (self.cls_embed.weight,)
        bs = (me[0].bias,) + ((self.v2l_transform.bias,) if self.use_class_emb else ()) + (self.cls_embed.bias,)
        wcat, bcat, Nc = runtime.packed_cached(ws, bs)
        C = me[0].weight.shape[0]
        ld = (Nc + 31) // 32 * 32
        big = torch.empty((d.shape[0], ld), dtype=torch.float32, device=d.device)[:, :Nc]
        lr(d, wcat, Nc, bcat, relu_cols=C, out=big)
        o = C
        cls_emb_pred = None
        if self.use_class_emb:
            De = self.v2l_transform.weight.shape[0]
            cls_emb_pred = big[:, o:o + De].reshape(B, Q, De) if last else big[:, o:o + De].unflatten(0, (B, Q))
            o += De
        cls_pred = big[:, o:].reshape(B, Q, -1) if last else big[:, o:].unflatten(0, (B, Q))
        if cls_emb_pred is None:
            cls_emb_pred = cls_pred
        elif self.pred_emb_norm:
            cls_emb_pred = cls_emb_pred / cls_emb_pred.norm(dim=-1, keepdim=True)
        w1, b1, _ = runtime.packed_cached((me[2].weight,), (me[2].bias,))
        w2, b2, N2 = runtime.packed_cached((me[4].weight,), (me[4].bias,))
        h = lr(big[:, :C], w1, me[2].weight.shape[0], b1, relu_cols=me[2].weight.shape[0])
        mask_embed = lr(h, w2, N2, b2).view(B, Q, N2)
        mask_pred = None
        if want_mask:
            mask_pred, _ = ops.mask_logits(mask_embed, packed_full, want_logits=True)
        bits = None
        if want_attn:
            if pooled_next is not None:
                _, bits = ops.mask_logits(mask_embed, pooled_next, want_logits=False, want_bits=True)
            else:
                full = mask_pred if mask_pred is not None else ops.mask_logits(mask_embed, packed_full)[0]
                bits = ops.attn_mask_from_logits(full.contiguous(), (int(sizes_next[0]), int(sizes_next[1])))
        return cls_pred, cls_emb_pred, mask_pred, bits

    def _decode_stream(self, B, kvs, sizes, packed_full, pooled, all_masks):
        """The 1 + 9 forward_head calls and 9 decoder layers of mask2former_head.py:808-847 on (B*Q, C) rows."""
        layers = self.transformer_decoder.layers
        nl = self.num_transformer_decoder_layers
        L = self.num_transformer_feat_level
        pn = self.transformer_decoder.post_norm
        pos = self.query_embed.weight.detach()
        Q, C = pos.shape
        qf = self.query_feat.weight.detach()
        if not all_masks and nl > 0 and self._lean_decode_ok(C):
            return self._decode_stream_lean(B, kvs, sizes, packed_full, pooled)
        x = qf.unsqueeze(0).expand(B, -1, -1).reshape(B * Q, C)
        xp = (qf + pos).unsqueeze(0).expand(B, -1, -1).reshape(B * Q, C)
        d = ops.layernorm_chain(x, (pn.weight, pn.bias, pn.eps))[0]
        outs = ([], [], [])
        res = self._head_stream(d, sizes[0], packed_full, pooled[0], all_masks or nl == 0, nl > 0, nl == 0)
        bits = res[3]
        for k in range(3):
            outs[k].append(res[k])
        for i in range(nl):
            li = i % L
            if self.attn_mask_hook is not None:
                bits = self.attn_mask_hook(i, bits)
            ops.attn_mask_fix_full_rows(bits, sizes[li][0] * sizes[li][1])
            x, xp, d = layers[i].forward_stream(x, xp, pos, kvs[i], bits, pn)
            last = i == nl - 1
            nxt = (i + 1) % L
            res = self._head_stream(d, sizes[nxt], packed_full, pooled[nxt], all_masks or last, not last, last)
            bits = res[3]
            for k in range(3):
                outs[k].append(res[k])
        return outs

    def _lean_decode_ok(self, C):
        me = self.mask_embed
        layers = self.transformer_decoder.layers
        return (C == 256 and self.attn_mask_hook is None and len(me) == 5
                and all(tuple(m.weight.shape) == (C, C) for m in (me[0], me[2], me[4]))
                and all(l.attentions[0].embed_dims == C for l in layers))

    def _decode_stream_lean(self, B, kvs, sizes, packed_full, pooled):
        """Inference-only decode (`all_masks=False`): the intermediate layers' class / caption-embedding predictions
        are never read (simple_test takes [-1], maskformer_open.py:161-178), so between two layers only the attention
        mask of the next one is produced: FFN planes -> `cgg_decoder_tail_bf16` (norm, post_norm, mask MLP, next query
        projection) -> mask bits. Everything before the first cross-attention depends on the weights alone (query_feat,
        query_embed, post_norm, mask_embed, layer 0's query projection) and is cached. Returns the same lists as
        `_decode_stream` with None for the skipped intermediate entries."""
        layers = self.transformer_decoder.layers
        nl = self.num_transformer_decoder_layers
        L = self.num_transformer_feat_level
        pn = self.transformer_decoder.post_norm
        pos = self.query_embed.weight.detach()
        Q, C = pos.shape
        qf = self.query_feat.weight
        me = self.mask_embed
        pk = runtime.packed_cached
        mlp = pk((me[0].weight,), (me[0].bias,))[:2] + pk((me[2].weight,), (me[2].bias,))[:2] + \
            pk((me[4].weight,), (me[4].bias,))[:2]
        pnorm = (pn.weight, pn.bias, pn.eps)

        def qproj(i):
            a = layers[i].attentions[0].attn
            return pk((a.in_proj_weight[:C],), (a.in_proj_bias[:C],))[:2]

        def consts():
            x = qf.detach().unsqueeze(0).expand(B, -1, -1).reshape(B * Q, C).contiguous()
            xp = (qf.detach() + pos).unsqueeze(0).expand(B, -1, -1).reshape(B * Q, C).contiguous()
            d = ops.layernorm_chain(x, pnorm)[0]
            lr = ops.linear_rows_bf16
            h = lr(d, mlp[0], C, mlp[1], relu_cols=C)
            h = lr(h, mlp[2], C, mlp[3], relu_cols=C)
            m0 = lr(h, mlp[4], C, mlp[5])
            wq0, bq0 = qproj(0)
            q0 = lr(xp, wq0, C, bq0)
            return x, m0, q0

        a0 = layers[0].attentions[0].attn
        x, m0, q = runtime.derived_cached(
            'lean_decode_consts_%d_%s' % (B, runtime.precision()),     # computed by the mode's own kernels
            (qf, self.query_embed.weight, pn.weight, pn.bias, me[0].weight, me[0].bias, me[2].weight, me[2].bias,
             me[4].weight, me[4].bias, a0.in_proj_weight, a0.in_proj_bias), consts)
        outs = ([None], [None], [None])
        _, bits = ops.mask_logits(m0.view(B, Q, C), pooled[0], want_logits=False, want_bits=True) \
            if pooled[0] is not None else (None, None)
        if bits is None:
            full = ops.mask_logits(m0.view(B, Q, C), packed_full)[0]
            bits = ops.attn_mask_from_logits(full.contiguous(), (int(sizes[0][0]), int(sizes[0][1])))
        for i in range(nl):
            # rows that mask every key are un-masked (:825-826) inside the attention kernels (fix_rows)
            last = i == nl - 1
            n2 = layers[i].norms[2]
            if last:
                x, _, d = layers[i].forward_stream(x, None, pos, kvs[i], bits, pn, q=q, fix_rows=True)
                res = self._head_stream(d, sizes[0], packed_full, pooled[0], True, False, True)
                for k in range(3):
                    outs[k].append(res[k])
                break
            t = layers[i].forward_stream(x, None, pos, kvs[i], bits, pn, q=q, raw=True, fix_rows=True)
            x, _, m, q = ops.decoder_tail(t, (n2.weight, n2.bias, n2.eps), pos, pnorm, mlp, qproj(i + 1))
            nxt = (i + 1) % L
            if pooled[nxt] is not None:
                _, bits = ops.mask_logits(m.view(B, Q, C), pooled[nxt], want_logits=False, want_bits=True)
            else:
                full = ops.mask_logits(m.view(B, Q, C), packed_full)[0]
                bits = ops.attn_mask_from_logits(full.contiguous(), (int(sizes[nxt][0]), int(sizes[nxt][1])))
            for k in range(3):
                outs[k].append(None)
        return outs

    def _forward(self, feats, img_metas, all_masks=True, encoded=None):
        enc = encoded if encoded is not None else self._encode(feats)
        return self._decode(enc, len(img_metas), all_masks)

    @staticmethod
    def _pack_stream(mf, sizes):
        """Packed full-resolution mask feature + the pooled images of the decoder levels, one launch when possible."""
        H4, W4 = int(mf.shape[1]), int(mf.shape[2])
        pools = []
        for (h, w) in sizes:
            s = H4 // h
            pools.append(s if (h * s == H4 and w * s == W4 and s in (2, 4, 8)) else None)
        uniq = [1] + sorted({p for p in pools if p is not None})
        if len(uniq) <= 4:
            packed = dict(zip(uniq, ops.pack_mask_feature_nhwc_multi(mf, uniq)))
        else:
            packed = {p: ops.pack_mask_feature_nhwc(mf, p) for p in uniq}
        return packed[1], [packed[p] if p is not None else None for p in pools]

    def _finish_encode(self, enc):
        """The tail of `_encode` for a deferred stream encoding: packed (full + pooled) mask feature and K / V."""
        if enc.get('x3a'):
            sizes, memorys = enc['sizes'], enc['memorys']
            mf = enc['mf'] if 'mf' in enc else self.pixel_decoder.stream_fpn_x3a(*enc['fpn'])
            H4, W4 = int(mf.shape[1]), int(mf.shape[2])
            pools = []
            for h, w in sizes:
                s = H4 // h
                pools.append(s if (h * s == H4 and w * s == W4 and s in (2, 4, 8)) else None)
            uniq = [1] + sorted({p for p in pools if p is not None})
            packed = dict(zip(uniq, ops.pack_mask_feature_nhwc_x3(mf, uniq)))
            return dict(stream=True, kvs=self._project_kv_x3a(memorys, sizes), sizes=sizes, packed_full=packed[1],
                        pooled=[packed[p] if p is not None else None for p in pools], mask_features=None)
        kv16, sizes = enc['kv16'], enc['sizes']
        mf = enc['mf'] if 'mf' in enc else self.pixel_decoder.stream_fpn(*enc['fpn'])
        L = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers
        H4, W4 = int(mf.shape[1]), int(mf.shape[2])
        packed_full, pooled = self._pack_stream(mf, sizes)
        kvs = self._project_kv_levels(kv16)
        return dict(stream=True, kvs=kvs, sizes=sizes, packed_full=packed_full, pooled=pooled, mask_features=None)

    def _project_kv_levels(self, kv16):
        """K / V of all decoder layers from the per-level bf16 operands (m16, mp16): the layers that read level l
        (l, l + L, ...) share their input, so their key projections are ONE GEMM against the stacked [Wk_i] (and the
        transposed value projections one batched GEMM against the stacked [Wv_i]); each layer then gets a column slice
        of k and a row block of vt (`cgg_masked_xattn_forward_bf16` takes the strides). 6 launches instead of 18."""
        L = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers
        nl = self.num_transformer_decoder_layers
        E = layers[0].attentions[0].embed_dims
        kvs = [None] * nl
        for l in range(min(L, nl)):
            idx = list(range(l, nl, L))
            ws = tuple(layers[i].attentions[0].attn.in_proj_weight for i in idx)
            bs = tuple(layers[i].attentions[0].attn.in_proj_bias for i in idx)
            wk = runtime.derived_cached('kv_levels_wk', ws, lambda: torch.cat([w[E:2 * E] for w in ws], 0).to(torch.bfloat16).contiguous())
            bk = runtime.derived_cached('kv_levels_bk', bs, lambda: torch.cat([b[E:2 * E] for b in bs], 0).to(torch.bfloat16).contiguous())
            wv = runtime.derived_cached('kv_levels_wv', ws, lambda: torch.cat([w[2 * E:] for w in ws], 0).to(torch.bfloat16).contiguous())
            m16, mp16 = kv16[l]
            if FUSED_KV and E == 256 and m16.is_contiguous() and mp16.is_contiguous():
                # both projections of the level in ONE launch (k row-major, v transposed straight from the MFMA tiles)
                wkp = runtime.derived_cached('kv_levels_wkp', ws, lambda: ops.pack_decoder_k_weight(torch.cat([w[E:2 * E] for w in ws], 0)))
                wvp = runtime.derived_cached('kv_levels_wvp', ws, lambda: ops.pack_linear_weight(torch.cat([w[2 * E:] for w in ws], 0)))
                bkf = runtime.derived_cached('kv_levels_bkf', bs, lambda: torch.cat([b[E:2 * E] for b in bs], 0).float().contiguous())
                k_all, vt_all = ops.decoder_kv_proj(m16, mp16, wkp, bkf, wvp)
            else:
                k_all = F.linear(mp16, wk, bk)                               # (B, hw, n E)
                vt_all = torch.matmul(wv, m16.transpose(1, 2))               # (B, n E, hw)
            for j, i in enumerate(idx):
                kvs[i] = (k_all[:, :, j * E:(j + 1) * E], vt_all[:, j * E:(j + 1) * E, :])
        return kvs

    def _kv_bf16_ok(self):
        """Throughput-mode decode (bf16 K / V, `forward_stream` layers) is available."""
        layers = self.transformer_decoder.layers
        return (runtime.is_bf16() and not torch.is_grad_enabled() and self.transformer_decoder.post_norm is not None
                and all(l.stream_ready() for l in layers) and self.query_embed.weight.shape[1] % 32 == 0)

    def _kv_tables(self, level_hw, dev):
        """(shift, pos) (N, C) f32 for `cgg_add_layernorm_kv`: level_embed_l broadcast over level l's rows and the
        decoder's sine encoding of every level (:795-812); rebuilt when level_embed changes."""
        w = self.level_embed.weight
        key = (tuple(level_hw), str(dev), w._version, w.data_ptr())
        hit = self.__dict__.get('_kv_table_cache')
        if hit is None or hit[0] != key:
            with torch.no_grad():
                shift = torch.cat([w[i].view(1, -1).expand(h * wd, -1) for i, (h, wd) in enumerate(level_hw)], 0)
                pos = torch.cat([self.decoder_positional_encoding.flat_unpadded(h, wd, dev) for h, wd in level_hw], 0)
                hit = (key, shift.float().contiguous(), pos.float().contiguous())
            self.__dict__['_kv_table_cache'] = hit
        return hit[1], hit[2]

    def _encode(self, feats, defer_tail=False):
        """The query-INDEPENDENT half of mask2former_head.py:763-849: pixel decoder, per-level memories, the packed
        mask feature (full + pooled images) and the K / V projections of all decoder layers. Everything here is
        throughput-bound (GEMMs, convolutions, gathers); `_decode` is the latency-bound query side. Splitting them lets
        a serving loop overlap `_decode` of batch k with `_encode` of batch k+1 (pipeline.TwoStagePipeline)."""
        L = self.num_transformer_feat_level
        pd = self.pixel_decoder
        stream = (hasattr(pd, 'stream_ready') and pd.stream_ready(feats) and pd.num_outs >= L
                  and all(isinstance(p, nn.Identity) for p in self.decoder_input_projs))
        mems, poss, sizes, pooled = [], [], [], []
        if stream:
            # throughput-mode inference: channel-last bf16 all the way, mask_feature only ever exists packed
            layers = self.transformer_decoder.layers
            lv = [(int(f.shape[2]), int(f.shape[3])) for f in list(feats)[::-1][:L]]
            kv_fused = (self._kv_bf16_ok() and pd.num_encoder_levels == L and len(feats) == pd.num_input_levels
                        and all(h * w % 4 == 0 and h * w >= 8 for h, w in lv)
                        and all(l.attentions[0].embed_dims // l.attentions[0].num_heads == 32 for l in layers))
            kv16 = None
            if kv_fused:
                mf, memorys, level_hw, kv16 = pd.forward_stream(feats, kv_tables=self._kv_tables,
                                                                defer_fpn=defer_tail == 2)
                if defer_tail == 2:
                    return dict(stream=True, deferred=True, fpn=mf, kv16=kv16, sizes=[level_hw[i] for i in range(L)])
            else:
                mf, memorys, level_hw = pd.forward_stream(feats)
            mask_features = None
            H4, W4 = int(mf.shape[1]), int(mf.shape[2])
            if kv16 is not None and defer_tail:
                # pipeline balancing: the packed mask features and the 18 K / V projections (0.25 ms of throughput-type
                # work) run at the head of the decode stage instead (`_finish_encode`)
                return dict(stream=True, deferred=True, mf=mf, kv16=kv16, sizes=[level_hw[i] for i in range(L)])
            for i in range(L):
                sizes.append(level_hw[i])
                if kv16 is None:
                    mems.append(memorys[i] + self.level_embed.weight[i].view(1, 1, -1))
                    poss.append(self.decoder_positional_encoding.flat_unpadded(level_hw[i][0], level_hw[i][1], mf.device))
            packed_full, pooled = self._pack_stream(mf, sizes)
        elif (hasattr(pd, 'stream_ready_x3') and pd.stream_ready_x3(feats) and pd.num_outs >= L
              and all(isinstance(p, nn.Identity) for p in self.decoder_input_projs)
              and self.transformer_decoder.post_norm is not None and self.query_embed.weight.shape[1] % 32 == 0
              and all(l.stream_ready() for l in self.transformer_decoder.layers)):
            # parity-mode inference: channel-last f32 all the way on the f32-class x3 kernels; the mask feature only ever
            # exists as its packed x3 images (full + pooled, one launch)
            if defer_tail and runtime.x3a_enabled():
                # pipeline balancing (parity mode): mask-feature packing + K / V projections (defer_tail = 1), and the FPN too
                # (defer_tail = 2), run at the head of the decode stage (`_finish_encode`)
                mf, memorys, level_hw = pd.forward_stream_x3(feats, defer_fpn=defer_tail == 2)
                return dict(stream=True, deferred=True, x3a=True, memorys=memorys, sizes=[level_hw[i] for i in range(L)],
                            **({'fpn': mf} if defer_tail == 2 else {'mf': mf}))
            mf, memorys, level_hw = pd.forward_stream_x3(feats)
            mask_features = None
            H4, W4 = int(mf.shape[1]), int(mf.shape[2])
            pools = []
            x3a_mem = all(ops.is_x3a(m) for m in memorys[:L])
            for i in range(L):
                h, w = level_hw[i]
                sizes.append((h, w))
                if not x3a_mem:
                    mems.append(memorys[i] + self.level_embed.weight[i].view(1, 1, -1))
                    poss.append(self.decoder_positional_encoding.flat_unpadded(h, w, mf.device))
                s = H4 // h
                pools.append(s if (h * s == H4 and w * s == W4 and s in (2, 4, 8)) else None)
            uniq = [1] + sorted({p for p in pools if p is not None})
            packed = dict(zip(uniq, ops.pack_mask_feature_nhwc_x3(mf, uniq)))
            packed_full, pooled = packed[1], [packed[p] if p is not None else None for p in pools]
            if x3a_mem:
                # round 4: the memories are x3a rows -- ONE LDS-DMA GEMM per level projects the K / V of every decoder layer that
                # reads it (level embedding, key position and biases folded into a per-token table)
                kvs = self._project_kv_x3a(memorys, sizes)
                return dict(stream=True, kvs=kvs, sizes=sizes, packed_full=packed_full, pooled=pooled, mask_features=None)
        else:
            if runtime.x3_enabled() and not torch.is_grad_enabled() and feats[0].is_cuda:
                runtime.note_fallback('pixel decoder / head', 'stream_ready_x3() is false for this configuration: module path on f32 '
                                      'library GEMMs')
            feats = [ops.x3a_to_f32(f) if ops.is_x3a(f) else f for f in feats]       # x3a backbone maps outside the x3 stream
            feats = [f.float().contiguous() if f.dtype != torch.float32 else f for f in feats]
            mask_features, memorys = pd(feats)
            mask_features = mask_features.contiguous()
            split = not runtime.is_bf16()
            H4, W4 = mask_features.shape[-2:]
            for i in range(L):
                m = self.decoder_input_projs[i](memorys[i])
                h, w = m.shape[-2:]
                sizes.append((int(h), int(w)))
                mems.append((m.flatten(2).transpose(1, 2) + self.level_embed.weight[i].view(1, 1, -1)).contiguous())
                poss.append(self.decoder_positional_encoding.flat_unpadded(int(h), int(w), m.device))
            feat_d = mask_features.detach()
            packed_full = ops.pack_mask_feature(feat_d, 1, split)
            for (h, w) in sizes:
                s = H4 // h
                ok = h * s == H4 and w * s == W4 and s in (2, 4, 8)
                pooled.append(ops.pack_mask_feature(feat_d, s, split) if ok else None)
        layers = self.transformer_decoder.layers
        if self._kv_bf16_ok() and packed_full.lo is None:
            if stream and kv16 is not None:
                kvs = self._project_kv_levels(kv16)
            elif all(h * w % 4 == 0 and h * w >= 8 for h, w in sizes) and \
                    all(l.attentions[0].embed_dims // l.attentions[0].num_heads == 32 for l in layers):
                m16 = [m.to(torch.bfloat16) for m in mems]
                mp16 = [(m + p[None]).to(torch.bfloat16) for m, p in zip(mems, poss)]
                kvs = [layers[i].attentions[0].project_kv_bf16(m16[i % L], mp16[i % L])
                       for i in range(self.num_transformer_decoder_layers)]
            else:
                kvs = [layers[i].attentions[0].project_kv(mems[i % L], poss[i % L])
                       for i in range(self.num_transformer_decoder_layers)]
            return dict(stream=True, kvs=kvs, sizes=sizes, packed_full=packed_full, pooled=pooled, mask_features=None)
        # K/V of every decoder layer (layer i reads level i % L) -- independent of the queries
        kvs = [layers[i].attentions[0].project_kv(mems[i % L], poss[i % L])
               for i in range(self.num_transformer_decoder_layers)]
        # parity mode: the query side runs the same row-stream kernels as throughput mode on f32-class (x3) contractions,
        # with the f32 [K | V] tensors and the f32-MFMA attention kernels
        x3_stream = (runtime.x3_enabled() and not torch.is_grad_enabled() and packed_full.hi.is_cuda
                     and self.transformer_decoder.post_norm is not None and all(l.stream_ready() for l in layers)
                     and self.query_embed.weight.shape[1] % 32 == 0)
        return dict(stream=x3_stream, kvs=kvs, sizes=sizes, packed_full=packed_full, pooled=pooled,
                    mask_features=mask_features)

    def _project_kv_x3a(self, memorys, sizes):
        """[K | V] of every decoder layer from the x3a encoder memories (`MSDeformAttnPixelDecoder._forward_stream_x3a`): per
        level ONE `ops.gemm_x3s` over the (B, hw, C) rows against the stacked in_proj key / value weights of the layers that read
        the level (layer i reads level i % L, mask2former_head.py:829-840). What the reference adds to the memory before the
        projection -- the level embedding (:806) and, for the keys, the positional encoding (key_pos, :834) -- and the biases are
        input-independent: they are folded through the weights into a per-token table (hw, n_layers * 2E) that rides in the GEMM's
        row-periodic residual. -> list over layers of (B, hw, 2E) f32 column slices of the level's output (read through
        `cgg_masked_xattn_forward_strided`)."""
        layers = self.transformer_decoder.layers
        L, nl = self.num_transformer_feat_level, self.num_transformer_decoder_layers
        kvs = [None] * nl
        for lvl in range(L):
            idx = list(range(lvl, nl, L))
            attns = [layers[i].attentions[0] for i in idx]
            h, w = sizes[lvl]
            mem = memorys[lvl].as_subclass(torch.Tensor)
            B = mem.shape[0]
            pos = self.decoder_positional_encoding.flat_unpadded(h, w, mem.device)
            E = attns[0].embed_dims

            def make(attns=attns, pos=pos, lvl=lvl, E=E):
                le = self.level_embed.weight[lvl].double()
                ws, tabs = [], []
                for a in attns:
                    w_kv, b_kv = a.kv_weight()
                    wd, bd = w_kv.double(), b_kv.double()
                    kb = (pos.double() + le) @ wd[:E].t() + bd[:E]
                    vb = (le @ wd[E:].t() + bd[E:]).expand(pos.shape[0], E)
                    ws.append(w_kv.float())
                    tabs.append(torch.cat([kb, vb], 1))
                return ops.pack_linear_weight_x3(torch.cat(ws, 0).contiguous()), torch.cat(tabs, 1).float().contiguous()

            params = tuple(p for a in attns for p in (a.attn.in_proj_weight, a.attn.in_proj_bias)) + (self.level_embed.weight, pos)
            wk, table = runtime.derived_cached('kv_stack_x3a', params, make)
            nout = table.shape[1]
            kv_all = ops.gemm_x3s(mem, wk, nout, None, res=table, res_mod=h * w).view(B, h * w, nout)
            for j, i in enumerate(idx):
                kvs[i] = kv_all[:, :, j * 2 * E:(j + 1) * 2 * E]
        return kvs

    def _decode(self, enc, B, all_masks=True):
        """The query side: 1 + 9 `forward_head` calls and the 9 decoder layers on the encoded memories."""
        if enc.get('deferred'):
            enc = self._finish_encode(enc)
        kvs, sizes, packed_full, pooled = enc['kvs'], enc['sizes'], enc['packed_full'], enc['pooled']
        mask_features = enc['mask_features']
        L = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers
        if enc['stream']:
            return self._decode_stream(B, kvs, sizes, packed_full, pooled, all_masks)
        query_feat = self.query_feat.weight.unsqueeze(0).expand(B, -1, -1)
        query_embed = self.query_embed.weight.unsqueeze(0).expand(B, -1, -1)

        cls_pred_list, cls_emb_pred_list, mask_pred_list = [], [], []
        nl = self.num_transformer_decoder_layers
        cls_pred, cls_emb_pred, mask_pred, bits = self.forward_head(
            query_feat, mask_features, sizes[0], packed_full, pooled[0], want_mask=all_masks or nl == 0)
        cls_pred_list.append(cls_pred)
        cls_emb_pred_list.append(cls_emb_pred)
        mask_pred_list.append(mask_pred)
        for i in range(nl):
            level_idx = i % L
            if self.attn_mask_hook is not None:
                bits = self.attn_mask_hook(i, bits)
            # rows that mask every key are un-masked (mask2former_head.py:825-826)
            ops.attn_mask_fix_full_rows(bits, sizes[level_idx][0] * sizes[level_idx][1])
            layer = layers[i]
            if tuple(layer.operation_order) == ('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm'):
                query_feat = layer.forward_fast(query_feat, query_embed, kvs[i], bits)
            else:
                raise NotImplementedError(f'operation_order {layer.operation_order} has no MI355X fast path')
            last = i == nl - 1
            nxt = (i + 1) % L
            cls_pred, cls_emb_pred, mask_pred, bits = self.forward_head(
                query_feat, mask_features, sizes[nxt], packed_full, pooled[nxt],
                want_mask=all_masks or last, want_attn=not last)
            cls_pred_list.append(cls_pred)
            cls_emb_pred_list.append(cls_emb_pred)
            mask_pred_list.append(mask_pred)
        return cls_pred_list, cls_emb_pred_list, mask_pred_list

    def forward(self, feats, img_metas):
        """mask2former_head.py:763-849: 3 lists of (num_layers + 1) tensors:
        cls (B,Q,K+1), emb (B,Q,d_l), mask logits (B,Q,h,w)."""
        return self._forward(feats, img_metas, all_masks=True)

    # ------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------
    def preprocess_gt(self, gt_labels_list, gt_masks_list, gt_semantic_segs, img_metas):
        """[3P] MaskFormerHead.preprocess_gt -> preprocess_panoptic_gt.
        gt_masks: tensors (n,H,W) or BitmapMasks-like objects with .pad(shape).to_tensor()."""
        if gt_semantic_segs is None:
            gt_semantic_segs = [None] * len(gt_labels_list)
        labels_out, masks_out = [], []
        for gt_labels, gt_masks, sem, meta in zip(gt_labels_list, gt_masks_list, gt_semantic_segs, img_metas):
            pad_h, pad_w = meta['pad_shape'][:2]
            if torch.is_tensor(gt_masks):
                tm = gt_masks.to(device=gt_labels.device, dtype=torch.bool)
                dh, dw = pad_h - tm.shape[-2], pad_w - tm.shape[-1]
                if dh or dw:
                    tm = F.pad(tm, (0, dw, 0, dh), value=False)
            else:
                tm = gt_masks.pad((pad_h, pad_w), pad_val=0).to_tensor(dtype=torch.bool, device=gt_labels.device)
            if sem is None:
                labels_out.append(gt_labels)
                masks_out.append(tm.long())
                continue
            sem = sem.squeeze(0)
            stuff_masks, stuff_labels = [], []
            for label in torch.unique(sem, sorted=False):
                if label < self.num_things_classes or label >= self.num_classes:
                    continue
                stuff_masks.append(sem == label)
                stuff_labels.append(label)
            if stuff_masks:
                labels = torch.cat([gt_labels, torch.stack(stuff_labels, 0)], 0)
                masks = torch.cat([tm, torch.stack(stuff_masks, 0)], 0)
            else:
                labels, masks = gt_labels, tm
            labels_out.append(labels)
            masks_out.append(masks.long())
        return labels_out, masks_out

    def _get_cls_emb_logits(self, cls_emb_preds):
        """mask2former_head.py:631-648."""
        return torch.matmul(cls_emb_preds, self.class_embs.t()) / self.softmax_temperature

    def _rand(self, kind, shape, device):
        if self.point_hook is not None:
            return self.point_hook(kind, shape, device)
        return torch.rand(*shape, device=device)

    def _draw_points(self, device):
        return self._rand('target', (1, self.num_points, 2), device)

    def _target_inputs(self, cls_score, cls_emb_logit, mask_pred, gt_labels, gt_masks):
        """mask2former_head.py:353-366: random points, sampled predictions and sampled GT masks."""
        num_queries, num_gts = cls_score.shape[0], gt_labels.shape[0]
        point_coords = self._draw_points(cls_score.device)
        mask_points_pred = point_sample(mask_pred.unsqueeze(1), point_coords.repeat(num_queries, 1, 1)).squeeze(1)
        gt_points_masks = point_sample(gt_masks.unsqueeze(1).float(), point_coords.repeat(num_gts, 1, 1)).squeeze(1)
        return (cls_score, cls_emb_logit, mask_points_pred, gt_labels, gt_points_masks)

    def _targets_from_assign(self, assign_result, mask_pred, gt_labels, gt_masks):
        """mask2former_head.py:372-390."""
        sampling_result = self.sampler.sample(assign_result, mask_pred, gt_masks)
        pos_inds, neg_inds = sampling_result.pos_inds, sampling_result.neg_inds
        labels = gt_labels.new_full((self.num_queries, ), self.num_classes, dtype=torch.long)
        labels[pos_inds] = gt_labels[sampling_result.pos_assigned_gt_inds]
        label_weights = gt_labels.new_ones((self.num_queries, ))
        mask_targets = gt_masks[sampling_result.pos_assigned_gt_inds]
        mask_weights = mask_pred.new_zeros((self.num_queries, ))
        mask_weights[pos_inds] = 1.0
        return labels, label_weights, mask_targets, mask_weights, pos_inds, neg_inds

    def _get_target_single(self, cls_score, cls_emb_logit, mask_pred, gt_labels, gt_masks, img_metas):
        """mask2former_head.py:320-390 (one image, one decoder layer)."""
        item = self._target_inputs(cls_score, cls_emb_logit, mask_pred, gt_labels, gt_masks)
        assign_result = self.assigner.assign(*item, img_metas)
        return self._targets_from_assign(assign_result, mask_pred, gt_labels, gt_masks)

    def get_targets(self, cls_scores_list, cls_emb_logits_list, mask_preds_list, gt_labels_list,
                    gt_masks_list, img_metas):
        """mask2former_head.py:273-317, with ONE device->host copy for the whole image list."""
        items = [self._target_inputs(c, e, m, gl, gm) for c, e, m, gl, gm in
                 zip(cls_scores_list, cls_emb_logits_list, mask_preds_list, gt_labels_list, gt_masks_list)]
        assigns = self.assigner.assign_batch(items)
        res = [self._targets_from_assign(a, m, gl, gm)
               for a, m, gl, gm in zip(assigns, mask_preds_list, gt_labels_list, gt_masks_list)]
        labels_list, label_weights_list, mask_targets_list, mask_weights_list, pos_l, neg_l = map(list, zip(*res))
        num_total_pos = sum(int(i.numel()) for i in pos_l)
        num_total_neg = sum(int(i.numel()) for i in neg_l)
        return labels_list, label_weights_list, mask_targets_list, mask_weights_list, num_total_pos, num_total_neg

    # ---- batched targets: the whole (layers x images) matching problem with O(images) launches ----
    def _fast_targets_ok(self):
        from .assigner import (ClassificationCost, CrossEntropyLossCost, DiceCost, MaskHungarianAssignerOpen,
                               MaskPseudoSampler)
        a = self.assigner
        return (type(a) is MaskHungarianAssignerOpen and type(self.sampler) is MaskPseudoSampler
                and type(a.cls_cost) is ClassificationCost and type(a.cls_emb_cost) is ClassificationCost
                and type(a.mask_cost) is CrossEntropyLossCost and type(a.dice_cost) is DiceCost)

    def _targets_batched(self, all_cls_scores, emb_logits, all_mask_preds, gt_labels_list, gt_f, overlap=None):
        """`_get_target_single` (mask2former_head.py:320-390) for every (layer, image) at once.
        The reference runs 2 point-samplings, 4 cost terms, a device->host sync and ~10 indexing kernels per
        (layer, image): 160 times at configs[2]. Here: predictions are sampled per LAYER (all images in one
        grid_sample), ground truth per IMAGE (all layers' points in one grid_sample on the float masks that are
        converted once per step), the cost matrices of an image's 10 layers are batched matmuls, ONE transfer
        brings all costs to the host for scipy's Hungarian solver, ONE transfer takes the targets back.
        Random points are drawn in the reference's order (layer-major, then image) so pinned draws line up.
        Returns per layer: (labels (B,Q) long, mask_weights (B,Q) f32, pos_b, pos_g (npos,) long, num_pos).
        overlap: optional callable run AFTER the cost matrices' device->host copy is enqueued (on a side stream, pinned target) and
        BEFORE the host waits for it: device work that does not depend on the targets (the caption branch) then executes while the
        host waits, solves the assignments and enqueues the loss kernels -- otherwise the device idles from the copy to the first
        loss kernel. Its return value is stored in `self._overlap_result`."""
        n, B, Q = len(all_cls_scores), all_cls_scores[0].shape[0], all_cls_scores[0].shape[1]
        dev = all_cls_scores[0].device
        P = self.num_points
        a = self.assigner
        if self.point_hook is None:
            pts = torch.rand((n, B, P, 2), device=dev)
        else:
            pts = torch.stack([torch.cat([self._draw_points(dev) for _ in range(B)], 0) for _ in range(n)], 0)
        with torch.no_grad():
            mf0 = all_mask_preds[0].mask_feature if isinstance(all_mask_preds[0], LazyMasks) else None
            if (mf0 is not None and mf0.is_cuda and mf0.dtype == torch.float32 and mf0.shape[1] % 4 == 0
                    and all(isinstance(m, LazyMasks) and m.mask_feature is mf0 for m in all_mask_preds)):
                # all layers share the mask feature: ONE channel-last sampling pass at the n * P points of every image,
                # then a (Q x C) x (C x P) product per layer (sample(E F) = E sample(F))
                # (the channel-last FPN path hands its own channel-last copy along: taken only if shape / dtype match and neither
                # tensor was written in place since -- runtime.handed_nhwc; the permute copy otherwise)
                nhwc = runtime.handed_nhwc(mf0, allow_grad=True)
                nhwc = nhwc if nhwc is not None else mf0.detach().permute(0, 2, 3, 1).contiguous()
                all_pts = pts.permute(1, 0, 2, 3).reshape(B, n * P, 2)
                pred_pts = torch.empty((n, B, Q, P), dtype=torch.float32, device=dev)     # (no stack: each product lands in its slab)
                if POINT_LOGITS_X3 and runtime.x3_enabled() and ops.point_sample_nhwc_x3_ok(nhwc, all_pts, n) \
                        and all(m.mask_embed.shape[-1] == nhwc.shape[-1] for m in all_mask_preds):
                    # parity mode: the sampler writes the samples as x3 images (no (B, n P, C) f32 tensor: 2 GB at configs[2]) and
                    # each layer's (Q x C) x (C x P) product is the f32-class MFMA einsum kernel on them (the f32 library bmm ran
                    # these at 44 TF/s: 2.3 ms per step)
                    packs = ops.point_sample_nhwc_x3(nhwc, all_pts, n)
                    for li in range(n):
                        ops.mask_logits(all_mask_preds[li].mask_embed.detach().float().contiguous(), packs[li], out=pred_pts[li])
                else:
                    fs = ops.point_sample_nhwc(nhwc, all_pts)                                      # (B, n*P, C)
                    for li in range(n):
                        torch.bmm(all_mask_preds[li].mask_embed.detach().float(), fs[:, li * P:(li + 1) * P].transpose(1, 2),
                                  out=pred_pts[li])
            else:
                pred_pts = torch.stack([all_mask_preds[li].sample_points(pts[li]) if isinstance(all_mask_preds[li], LazyMasks)
                                        else point_sample(all_mask_preds[li].detach(), pts[li]) for li in range(n)], 0)
            # the prediction-only halves of the mask / dice costs ONCE for all layers and images (mmdet CrossEntropyLossCost forms
            # pos . t + neg . (1 - t) with pos = softplus(-x), neg = softplus(x); pos - neg = -x, so the pair of contractions is
            # sum_p softplus(x) - x . t: per image one contraction with the targets instead of ~12 passes over its (n, Q, P) logits)
            x_all = pred_pts.float()                                                          # (n, B, Q, P)
            xx_all = xx_sum = sp_sum = None
            if x_all.is_cuda and x_all.is_contiguous() and P % 4 == 0 and a.dice_cost.weight != 0 and a.dice_cost.pred_act:
                # one pass (cgg_match_cost_rows): sigmoid(x), sum softplus(x), sum sigmoid(x) [^2]
                xx_all, sp_sum, xx_sum = ops.match_cost_rows(x_all, square=not a.dice_cost.naive_dice)
            else:
                sp_sum = F.softplus(x_all).sum(-1) if a.mask_cost.weight != 0 else None       # (n, B, Q)
                if a.dice_cost.weight != 0:
                    xx_all = x_all.sigmoid() if a.dice_cost.pred_act else x_all
                    xx_sum = xx_all.sum(-1) if a.dice_cost.naive_dice else xx_all.pow(2).sum(-1)
            costs, shapes = [], []
            for b in range(B):
                G = int(gt_labels_list[b].shape[0])
                shapes.append(G)
                if G == 0:
                    continue
                gl = gt_labels_list[b]
                t = point_sample(gt_f[b][None], pts[:, b].reshape(1, n * P, 2))[0]          # (G, n*P)
                t = t.view(G, n, P).permute(1, 0, 2).contiguous()                             # (n, G, P)
                x = x_all[:, b]                                                               # (n, Q, P)
                tt = t.transpose(1, 2)
                cost = 0
                if a.cls_cost.weight != 0:
                    cs = torch.stack([c[b] for c in all_cls_scores], 0)
                    cost = cost + (-cs.detach().softmax(-1)[..., gl] * a.cls_cost.weight)
                if a.cls_emb_cost.weight != 0 and emb_logits[0] is not None:
                    es = torch.stack([e[b] for e in emb_logits], 0)
                    cost = cost + (-es.detach().softmax(-1)[..., gl] * a.cls_emb_cost.weight)
                if a.mask_cost.weight != 0:
                    c = sp_sum[:, b][:, :, None] - torch.bmm(x, tt)
                    cost = cost + c / P * a.mask_cost.weight
                if a.dice_cost.weight != 0:
                    dc = a.dice_cost
                    num = 2 * torch.bmm(xx_all[:, b], tt)
                    den = xx_sum[:, b][:, :, None] + (t.sum(-1) if dc.naive_dice else t.pow(2).sum(-1))[:, None, :]
                    cost = cost + (1 - (num + dc.eps) / (den + dc.eps)) * dc.weight
                if getattr(self, 'cost_trace', None) is not None:      # test hook: the (n, Q, G) cost matrices of image b
                    self.cost_trace.append((b, cost.detach().float().clone()))
                costs.append(cost.float().reshape(-1))                                        # (n*Q*G,)
            flat = None
            if costs and overlap is not None and costs[0].is_cuda:
                flat_dev = torch.cat(costs)
                flat = _pinned_like(flat_dev)
                lab_dev = torch.cat([g.reshape(-1) for g in gt_labels_list]).to(torch.int64)
                lab_host = _pinned_like(lab_dev)          # (a `.cpu()` per image after `overlap` would wait for the work it enqueued)
                cur, side = torch.cuda.current_stream(dev), _copy_stream(dev)
                side.wait_stream(cur)                     # AFTER both sources are enqueued on the caller's stream
                with torch.cuda.stream(side):
                    flat.copy_(flat_dev, non_blocking=True)
                    lab_host.copy_(lab_dev, non_blocking=True)
                    arrived = torch.cuda.Event()
                    arrived.record(side)
                flat_dev.record_stream(side)
                lab_dev.record_stream(side)
        if overlap is not None:
            self._overlap_result = overlap()              # (with autograd: the caption branch is part of the graph)
        with torch.no_grad():
            if flat is not None:
                arrived.synchronize()                                                         # the ONE sync
            elif costs:
                flat = torch.cat(costs).cpu()                                                 # the ONE sync
        import numpy as np
        labels_np = np.full((n, B, Q), self.num_classes, dtype=np.int64)
        weights_np = np.zeros((n, B, Q), dtype=np.float32)
        if costs and overlap is not None and costs[0].is_cuda:
            gl_host, o = [], 0
            for s_ in shapes:
                gl_host.append(lab_host[o:o + s_].numpy() if s_ else None)
                o += s_
        else:
            gl_host = [g.cpu().numpy() if s else None for g, s in zip(gt_labels_list, shapes)] if costs else []
        pos_b = [[] for _ in range(n)]
        pos_g = [[] for _ in range(n)]
        pos_q = [[] for _ in range(n)]
        pos_r = [[] for _ in range(n)]          # rank of the positive inside its image (padded-batch slot)
        # every (image, layer) problem of the step in ONE call of the C++ solver (scipy-identical indices)
        mats, off = [], 0
        for b in range(B):
            G = shapes[b]
            if G:
                cm = flat[off:off + n * Q * G].view(n, Q, G)
                off += n * Q * G
                mats.extend(cm[li] for li in range(n))
        solved = iter(ops.linear_sum_assignment_batch(mats))
        mi = 0
        for b in range(B):
            G = shapes[b]
            if G == 0:
                continue
            for li in range(n):
                rows, cols = (t.numpy() for t in next(solved))
                if self.assign_hook is not None:
                    rows, cols = self.assign_hook(li, b, rows, cols, mats[mi])
                mi += 1
                order = np.argsort(rows)                     # positives in ascending query order (sampler: unique())
                rows, cols = rows[order], cols[order]
                labels_np[li, b, rows] = gl_host[b][cols]
                weights_np[li, b, rows] = 1.0
                pos_b[li].extend([b] * len(rows))
                pos_g[li].extend(cols.tolist())
                pos_q[li].extend(rows.tolist())
                pos_r[li].extend(range(len(rows)))
        labels = torch.from_numpy(labels_np).to(dev, non_blocking=True)
        weights = torch.from_numpy(weights_np).to(dev, non_blocking=True)
        # positives of all layers as ONE index tensor (rows: image, query, gt, slot) -> one H2D
        flat_idx = np.array([sum(pos_b, []), sum(pos_q, []), sum(pos_g, []), sum(pos_r, [])], dtype=np.int64).reshape(4, -1)
        idx_dev = torch.from_numpy(flat_idx).to(dev, non_blocking=True)
        out, o = [], 0
        for li in range(n):
            k = len(pos_b[li])
            pm = max((pos_b[li].count(b) for b in set(pos_b[li])), default=0)
            devi = dict(b=idx_dev[0, o:o + k], q=idx_dev[1, o:o + k], g=idx_dev[2, o:o + k], r=idx_dev[3, o:o + k], pm=pm)
            out.append((labels[li], weights[li], pos_b[li], pos_g[li], k, devi))
            o += k
        return out

    def gather_captions_and_preds(self, gt_caption_embs_list, gt_caption_mask_list, cls_emb_preds):
        """mask2former_head.py:650-684: all_gather of captions / masks / predictions; the local slice of
        the predictions is re-inserted so it keeps its gradient (remote slices are constants)."""
        batch_size = len(gt_caption_embs_list)
        rank, world_size = get_dist_info()
        embs = torch.stack(gt_caption_embs_list, dim=0)
        mask = torch.stack(gt_caption_mask_list, dim=0)
        if world_size == 1:
            return embs, mask, cls_emb_preds
        emb_l = [torch.zeros_like(embs) for _ in range(world_size)]
        mask_l = [torch.zeros_like(mask) for _ in range(world_size)]
        pred_l = [torch.zeros_like(cls_emb_preds) for _ in range(world_size)]
        _all_gather(emb_l, embs.contiguous())
        _all_gather(mask_l, mask.contiguous())
        _all_gather(pred_l, cls_emb_preds.detach().contiguous())
        all_preds = torch.cat(pred_l, dim=0)
        all_preds[rank * batch_size:(rank + 1) * batch_size] = cls_emb_preds
        return torch.cat(emb_l, dim=0), torch.cat(mask_l, dim=0), all_preds

    def _gather_all_layers(self, embs_list, mask_list, all_preds):
        """Same result per layer as `gather_captions_and_preds`, with 3 collectives per STEP."""
        rank, world_size = get_dist_info()
        embs = torch.stack(embs_list, dim=0)
        mask = torch.stack(mask_list, dim=0)
        if world_size == 1:
            return [(embs, mask, p) for p in all_preds]
        bs = embs.shape[0]
        emb_l = [torch.zeros_like(embs) for _ in range(world_size)]
        mask_l = [torch.zeros_like(mask) for _ in range(world_size)]
        stacked = torch.stack([p.detach() for p in all_preds], dim=0).contiguous()   # (n,B,Q,d)
        pred_l = [torch.zeros_like(stacked) for _ in range(world_size)]
        _all_gather(emb_l, embs.contiguous())
        _all_gather(mask_l, mask.contiguous())
        _all_gather(pred_l, stacked)
        all_embs, all_mask = torch.cat(emb_l, dim=0), torch.cat(mask_l, dim=0)
        out = []
        for li, p in enumerate(all_preds):
            ap = torch.cat([pl[li] for pl in pred_l], dim=0)
            ap[rank * bs:(rank + 1) * bs] = p
            out.append((all_embs, all_mask, ap))
        return out

    def extract_word_embeddings(self, ids_list, mask_list, emb_type='bert'):
        """mask2former_head.py:686-709."""
        if emb_type != 'bert':
            raise NotImplementedError("only emb_type='bert' (as in every reference config)")
        embs_list = [self.bert_embeddings(ids, normalize=self.text_emb_norm) for ids in ids_list]
        return embs_list, list(mask_list)

    def loss_single(self, cls_scores, cls_emb_preds, mask_preds, gt_labels_list, gt_masks_list,
                    gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list,
                    gt_caption_nouns_ids_list, gt_caption_nouns_embs_list, gt_caption_nouns_mask_list,
                    img_metas, num_total_masks=None, gathered=None, targets=None, fast=None):
        """mask2former_head.py:464-629 for one decoder layer. `num_total_masks` / `gathered` carry the
        coalesced collectives prepared by `loss()`; when None they are computed here as the reference does."""
        num_imgs = cls_scores.size(0)
        pos_b = pos_g = gt_f = cap_loss = None
        if fast is not None:
            (labels2, mask_weights, pos_b, pos_g, num_total_pos, pos_dev), gt_f, cls_emb_logits, cap_loss = fast[:4]
            labels = labels2.flatten(0, 1)
            label_weights = torch.ones_like(labels)
            mask_targets = None
        else:
            cls_scores_list = [cls_scores[i] for i in range(num_imgs)]
            if self.use_class_emb:
                cls_emb_logits = self._get_cls_emb_logits(cls_emb_preds)
                cls_emb_logits_list = [cls_emb_logits[i] for i in range(num_imgs)]
            else:
                cls_emb_logits_list = [None] * num_imgs
            mask_preds_list = [mask_preds[i] for i in range(num_imgs)]
            if targets is None:
                targets = self.get_targets(cls_scores_list, cls_emb_logits_list, mask_preds_list,
                                           gt_labels_list, gt_masks_list, img_metas)
            labels_list, label_weights_list, mask_targets_list, mask_weights_list, num_total_pos, _ = targets
            labels = torch.stack(labels_list, dim=0).flatten(0, 1)
            label_weights = torch.stack(label_weights_list, dim=0).flatten(0, 1)
            mask_targets = torch.cat(mask_targets_list, dim=0)
            mask_weights = torch.stack(mask_weights_list, dim=0)

        cls_scores = cls_scores.flatten(0, 1)
        class_weight = runtime.const_tensor(self.class_weight, cls_scores)
        avg = class_weight[labels].sum()
        loss_cls = self.loss_cls(cls_scores, labels, label_weights, avg_factor=avg)
        zero = loss_cls.new_zeros(())

        loss_cls_emb = zero
        if self.use_class_emb:
            loss_cls_emb = self.loss_cls_emb(cls_emb_logits.flatten(0, 1), labels, label_weights.float(),
                                             avg_factor=avg)

        loss_grounding = zero
        if self.use_caption:
            if gathered is None:
                all_embs, all_mask, all_preds = self.gather_captions_and_preds(
                    gt_caption_nouns_embs_list, gt_caption_nouns_mask_list, cls_emb_preds)
            else:
                all_embs, all_mask, all_preds = gathered
            loss_grounding = self.loss_grounding(all_preds, all_embs, all_mask, self.softmax_temperature)

        loss_caption_generation = zero
        if self.use_caption_generation and cap_loss is not None:
            loss_caption_generation = cap_loss
        elif self.use_caption_generation:
            gt_caption_embs = torch.stack(gt_caption_embs_list, dim=0)
            gt_caption_masks = torch.stack(gt_caption_mask_list, dim=0).bool()
            ids = self._caption_targets(gt_caption_ids_list, gt_caption_nouns_ids_list)[:, 1:].flatten(0, 1)
            cg, lcg = self.caption_generator, self.loss_caption_generation
            kw = dict(tgt=gt_caption_embs[:, :-1, :], memory=cls_emb_preds,
                      tgt_key_padding_mask=torch.logical_not(gt_caption_masks[:, :-1]))
            if gt_caption_embs.is_cuda and hasattr(cg, 'generator_ce_rows') and getattr(lcg, 'rows_ok', lambda: False)():
                rows = cg.generator_ce_rows(cg.forward_hidden(**kw).flatten(0, 1), ids, lcg.ignore_index)
                loss_caption_generation = lcg.forward_rows(rows, ids)
            else:
                loss_caption_generation = lcg(cg(**kw)[1].flatten(0, 1), ids)

        loss_caption_align = zero
        if self.use_caption_align:
            loss_caption_align = self.loss_caption_align(
                cls_emb_preds, torch.stack(gt_caption_nouns_embs_list, dim=0),
                torch.stack(gt_caption_nouns_mask_list, dim=0).bool())

        if num_total_masks is None:
            num_total_masks = reduce_mean(cls_scores.new_tensor([num_total_pos]))
            num_total_masks = max(num_total_masks, 1)

        if isinstance(mask_preds, LazyMasks):
            mask_preds = mask_preds.select(pos_dev) if fast is not None else mask_preds.select_by_weights(mask_weights)
        else:
            mask_preds = mask_preds[mask_weights > 0]
        if (len(pos_b) if fast is not None else mask_targets.shape[0]) == 0:
            loss_dice = mask_preds.sum()
            loss_mask = mask_preds.sum()
            return (loss_cls, loss_cls_emb, loss_grounding, loss_caption_generation, loss_caption_align,
                    loss_mask, loss_dice)
        with torch.no_grad():
            points_coords = get_uncertain_point_coords_with_randomness(
                mask_preds.unsqueeze(1), None, self.num_points, self.oversample_ratio,
                self.importance_sample_ratio, rand_fn=self._rand)
            if fast is not None:
                # positives are ordered image-major (boolean indexing above): sample each image's float GT masks
                # at its positives' points and pick the assigned mask -- no (npos, H, W) gather / float pass
                gt_cat = fast[4] if len(fast) > 4 else None
                if gt_cat is not None and points_coords.is_cuda:
                    # ONE launch for all images: every positive samples only ITS assigned ground-truth plane
                    planes, g_off = gt_cat
                    idx = (g_off[pos_dev['b']] + pos_dev['g']).to(torch.int32)
                    mask_point_targets = ops.point_sample_planes(planes, idx.contiguous(), points_coords)
                else:
                    Pn = points_coords.shape[1]
                    chunks, j0 = [], 0
                    while j0 < len(pos_b):
                        b = pos_b[j0]
                        j1 = j0
                        while j1 < len(pos_b) and pos_b[j1] == b:
                            j1 += 1
                        g_idx = pos_dev['g'][j0:j1]
                        smp = point_sample(gt_f[b][None], points_coords[j0:j1].reshape(1, (j1 - j0) * Pn, 2))[0]
                        smp = smp.view(-1, j1 - j0, Pn)                                   # (G, npos_b, P)
                        chunks.append(smp[g_idx, torch.arange(j1 - j0, device=smp.device)])
                        j0 = j1
                    mask_point_targets = torch.cat(chunks, 0)
            else:
                mask_point_targets = point_sample(mask_targets.unsqueeze(1).float(), points_coords).squeeze(1)
        if ops.point_sample_rows_ok(mask_preds, points_coords):
            # one gather kernel forward, one scatter kernel backward (no gradient wrt the constant points)
            mask_point_preds = ops.point_sample_rows(mask_preds, points_coords)
        else:
            mask_point_preds = point_sample(mask_preds.unsqueeze(1), points_coords).squeeze(1)
        loss_dice = self.loss_dice(mask_point_preds, mask_point_targets, avg_factor=num_total_masks)
        loss_mask = self.loss_mask(mask_point_preds.reshape(-1), mask_point_targets.reshape(-1),
                                   avg_factor=num_total_masks * self.num_points)
        return (loss_cls, loss_cls_emb, loss_grounding, loss_caption_generation, loss_caption_align,
                loss_mask, loss_dice)

    def _caption_targets(self, gt_caption_ids_list, gt_caption_nouns_ids_list):
        """mask2former_head.py:561-577. With the default flags the reference's B x 35 `int(tensor)` host
        loop is a no-op and is skipped; with a gen_* flag set it is reproduced (in place, as upstream)."""
        if self.gen_only_obj_nouns or self.gen_mask_obj_nouns or self.gen_replace_obj_nouns:
            for i in range(len(gt_caption_ids_list)):
                ids = gt_caption_ids_list[i]
                nouns = gt_caption_nouns_ids_list[i].cpu().numpy().tolist()
                for j in range(len(ids)):
                    if int(ids[j]) not in nouns:
                        if self.gen_only_obj_nouns:
                            ids[j] = 0
                    else:
                        if self.gen_mask_obj_nouns:
                            ids[j] = 0
                            break
                        if self.gen_replace_obj_nouns:
                            ids[j] = 4874  # 'object'
        return torch.stack(gt_caption_ids_list, dim=0)

    def loss(self, all_cls_scores, all_cls_emb_preds, all_mask_preds, gt_labels_list, gt_masks_list,
             gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list, gt_caption_nouns_ids_list,
             gt_caption_nouns_embs_list, gt_caption_nouns_mask_list, img_metas):
        """mask2former_head.py:393-462: 7 losses for the last layer + `d{i}.` copies for the others."""
        n = len(all_cls_scores)
        num_imgs = all_cls_scores[0].size(0)
        emb_logits = [self._get_cls_emb_logits(e) if self.use_class_emb else None for e in all_cls_emb_preds]
        if self._fast_targets_ok() and not getattr(self, 'force_reference_targets', False):
            return self._loss_batched(all_cls_scores, all_cls_emb_preds, emb_logits, all_mask_preds, gt_labels_list,
                                      gt_masks_list, gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list,
                                      gt_caption_nouns_ids_list, gt_caption_nouns_embs_list,
                                      gt_caption_nouns_mask_list, img_metas)
        # (1) Hungarian targets of ALL layers and images: one device->host copy (reference: n x B syncs)
        items = []
        for li in range(n):
            for b in range(num_imgs):
                items.append(self._target_inputs(
                    all_cls_scores[li][b], emb_logits[li][b] if emb_logits[li] is not None else None,
                    all_mask_preds[li][b], gt_labels_list[b], gt_masks_list[b]))
        assigns = self.assigner.assign_batch(items)
        targets, pos_counts = [], []
        for li in range(n):
            res = [self._targets_from_assign(assigns[li * num_imgs + b], all_mask_preds[li][b],
                                             gt_labels_list[b], gt_masks_list[b]) for b in range(num_imgs)]
            labels_l, lw_l, mt_l, mw_l, pos_l, neg_l = map(list, zip(*res))
            targets.append((labels_l, lw_l, mt_l, mw_l, sum(int(i.numel()) for i in pos_l),
                            sum(int(i.numel()) for i in neg_l)))
            pos_counts.append(float(targets[-1][4]))
        # (2) the n `reduce_mean` scalars (mask2former_head.py:591) as ONE all-reduce
        if dist.is_available() and dist.is_initialized():
            ntm = reduce_mean(all_cls_scores[0].new_tensor(pos_counts)).clamp(min=1).tolist()
        else:       # single process: the counts are already host numbers -> no device round trip
            ntm = [max(c, 1.0) for c in pos_counts]
        # (3) caption all_gathers: nouns/mask are layer-invariant -> once; predictions of all layers in
        #     one all_gather (reference: 3 all_gathers per layer, :671-673)
        gathered = [None] * n
        if self.use_caption:
            gathered = self._gather_all_layers(gt_caption_nouns_embs_list, gt_caption_nouns_mask_list,
                                               all_cls_emb_preds)
        results = []
        for i in range(n):
            results.append(self.loss_single(
                all_cls_scores[i], all_cls_emb_preds[i], all_mask_preds[i], gt_labels_list, gt_masks_list,
                gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list, gt_caption_nouns_ids_list,
                gt_caption_nouns_embs_list, gt_caption_nouns_mask_list, img_metas,
                num_total_masks=ntm[i], gathered=gathered[i], targets=targets[i]))
        names = ('loss_cls', 'loss_cls_emb', 'loss_grounding', 'loss_caption_generation',
                 'loss_caption_align', 'loss_mask', 'loss_dice')
        loss_dict = {k: v for k, v in zip(names, results[-1])}
        if self.loss_only_last:
            return loss_dict
        for li, res in enumerate(results[:-1]):
            for k, v in zip(names, res):
                loss_dict[f'd{li}.{k}'] = v * self.loss_aux_weight
        return loss_dict

    def _caption_generation_losses(self, n, B, all_cls_emb_preds, gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list,
                                   gt_caption_nouns_ids_list):
        """loss_caption_generation of all n layers from ONE pass of the caption generator (see `_loss_batched`)."""
        cap_losses = [None] * n
        if not self.use_caption_generation:
            return cap_losses
        emb = torch.stack(gt_caption_embs_list, dim=0)
        msk = torch.stack(gt_caption_mask_list, dim=0).bool()
        T1 = emb.shape[1] - 1
        ids = self._caption_targets(gt_caption_ids_list, gt_caption_nouns_ids_list)[:, 1:].flatten(0, 1)
        cg, lcg = self.caption_generator, self.loss_caption_generation
        kw = dict(tgt=emb[:, :-1, :].repeat(n, 1, 1), memory=torch.cat(list(all_cls_emb_preds), 0),
                  tgt_key_padding_mask=torch.logical_not(msk[:, :-1]).repeat(n, 1))
        if emb.is_cuda and hasattr(cg, 'generator_ce_rows') and getattr(lcg, 'rows_ok', lambda: False)():
            # generator + cross-entropy without the (n*B*34, 30522) logits: HIP row kernels over GEMM row chunks
            hidden = cg.forward_hidden(**kw).flatten(0, 1)                                   # (n*B*(T-1), hidden)
            rows = cg.generator_ce_rows(hidden, ids.repeat(n), lcg.ignore_index).view(n, B * T1)
            cap_losses = [lcg.forward_rows(rows[li], ids) for li in range(n)]
        else:
            logits = cg(**kw)[1].view(n, B * T1, -1)                                         # (n, B*(T-1), V)
            cap_losses = [lcg(logits[li], ids) for li in range(n)]
        return cap_losses

    def _loss_batched(self, all_cls_scores, all_cls_emb_preds, emb_logits, all_mask_preds, gt_labels_list,
                      gt_masks_list, gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list,
                      gt_caption_nouns_ids_list, gt_caption_nouns_embs_list, gt_caption_nouns_mask_list, img_metas):
        """`loss` (mask2former_head.py:393-462) with the per-(layer, image) work batched: targets by
        `_targets_batched`, the caption generator ONCE for all layers (the reference runs its 4-block transformer and
        the 30 522-way generator 10 times on the same targets), ground-truth masks sampled at the loss points per
        image instead of gathered at full resolution per positive. Loss values are the reference's."""
        n = len(all_cls_scores)
        B = all_cls_scores[0].size(0)
        gt_f = [gm.float() for gm in gt_masks_list]        # (G, H, W) once per step
        gt_cat = None
        if gt_f and gt_f[0].is_cuda and len({tuple(g.shape[-2:]) for g in gt_f}) == 1 and sum(g.shape[0] for g in gt_f) > 0:
            counts = torch.tensor([0] + [g.shape[0] for g in gt_f[:-1]], device=gt_f[0].device)
            gt_cat = (torch.cat(gt_f, 0).contiguous(), torch.cumsum(counts, 0))     # planes of all images + first plane of image b

        def caption_branch():
            # independent of the Hungarian targets: enqueued while the cost matrices travel to the host (see `_targets_batched`)
            gathered = [None] * n
            if self.use_caption:
                gathered = self._gather_all_layers(gt_caption_nouns_embs_list, gt_caption_nouns_mask_list,
                                                   all_cls_emb_preds)
            return gathered, self._caption_generation_losses(n, B, all_cls_emb_preds, gt_caption_ids_list, gt_caption_embs_list,
                                                             gt_caption_mask_list, gt_caption_nouns_ids_list)

        self._overlap_result = None
        targets = self._targets_batched(all_cls_scores, emb_logits, all_mask_preds, gt_labels_list, gt_f, overlap=caption_branch)
        gathered, cap_losses = self._overlap_result
        self._overlap_result = None
        pos_counts = [float(t[4]) for t in targets]
        if dist.is_available() and dist.is_initialized():
            ntm = reduce_mean(all_cls_scores[0].new_tensor(pos_counts)).clamp(min=1).tolist()
        else:
            ntm = [max(c, 1.0) for c in pos_counts]
        if all(isinstance(m, LazyMasks) for m in all_mask_preds):
            LazyMasks.preselect(list(all_mask_preds), [t[5] for t in targets])
        results = []
        for li in range(n):
            results.append(self.loss_single(
                all_cls_scores[li], all_cls_emb_preds[li], all_mask_preds[li], gt_labels_list, gt_masks_list,
                gt_caption_ids_list, gt_caption_embs_list, gt_caption_mask_list, gt_caption_nouns_ids_list,
                gt_caption_nouns_embs_list, gt_caption_nouns_mask_list, img_metas, num_total_masks=ntm[li],
                gathered=gathered[li], fast=(targets[li], gt_f, emb_logits[li], cap_losses[li], gt_cat)))
        names = ('loss_cls', 'loss_cls_emb', 'loss_grounding', 'loss_caption_generation',
                 'loss_caption_align', 'loss_mask', 'loss_dice')
        loss_dict = {k: v for k, v in zip(names, results[-1])}
        if self.loss_only_last:
            return loss_dict
        for li, res in enumerate(results[:-1]):
            for k, v in zip(names, res):
                loss_dict[f'd{li}.{k}'] = v * self.loss_aux_weight
        return loss_dict

    def forward_train(self, feats, img_metas, gt_bboxes, gt_labels, gt_masks, gt_semantic_seg,
                      gt_caption_ids, gt_caption_mask, gt_caption_nouns_ids, gt_caption_nouns_mask,
                      gt_bboxes_ignore=None, **kwargs):
        """mask2former_head.py:851-921."""
        assert gt_bboxes_ignore is None
        self._lazy_masks = self._fast_targets_ok() and not getattr(self, 'force_reference_targets', False)
        try:
            all_cls_scores, all_cls_emb_preds, all_mask_preds = self(feats, img_metas)
        finally:
            self._lazy_masks = False
        gt_labels, gt_masks = self.preprocess_gt(gt_labels, gt_masks, gt_semantic_seg, img_metas)
        gt_caption_embs = gt_caption_nouns_embs = None
        if self.use_caption_generation:
            gt_caption_embs, gt_caption_mask = self.extract_word_embeddings(
                gt_caption_ids, gt_caption_mask, self.caption_gen_emb_type)
        if self.use_caption:
            gt_caption_nouns_embs, gt_caption_nouns_mask = self.extract_word_embeddings(
                gt_caption_nouns_ids, gt_caption_nouns_mask, self.caption_emb_type)
        return self.loss(all_cls_scores, all_cls_emb_preds, all_mask_preds, gt_labels, gt_masks,
                         gt_caption_ids, gt_caption_embs, gt_caption_mask, gt_caption_nouns_ids,
                         gt_caption_nouns_embs, gt_caption_nouns_mask, img_metas)

    # ------------------------------------------------------------------------------------------
    # inference
    # ------------------------------------------------------------------------------------------
    def simple_test(self, feats, img_metas, **kwargs):
        """mask2former_head.py:923-980. Returns (assigned_labels | cls logits, emb (B,Q,d_l),
        LowResMasks, caption results, att). The mask logits stay at mask-feature resolution together
        with the upsample target (`batch_input_shape`); the fusion head's HIP kernels resize on the fly,
        `.upsampled()` gives the reference's (B,Q,H,W) tensor."""
        all_cls_scores, all_cls_emb_preds, all_mask_preds = self._forward(feats, img_metas, all_masks=False,
                                                                          encoded=kwargs.get('encoded'))
        mask_cls_results = all_cls_scores[-1]
        mask_cls_emb_results = all_cls_emb_preds[-1]
        mask_pred_results = all_mask_preds[-1]
        assigned_labels = mask_cls_results
        if kwargs.get('gt_labels', None) is not None:
            cls_emb_logits = self._get_cls_emb_logits(mask_cls_emb_results)
            gm = kwargs['gt_masks'][0][0]
            pad = img_metas[0]['pad_shape'][:2]
            if not torch.is_tensor(gm):
                gm = gm.pad(pad, pad_val=0).to_tensor(dtype=torch.long, device=cls_emb_logits.device)
            assigned_labels = self._get_target_single(mask_cls_results[0], cls_emb_logits[0],
                                                      mask_pred_results[0], kwargs['gt_labels'][0][0],
                                                      gm.long(), img_metas)[0]
        img_shape = img_metas[0]['batch_input_shape']
        if kwargs.get('img_shape', None):
            img_shape = kwargs['img_shape']
        masks = LowResMasks(mask_pred_results, (img_shape[0], img_shape[1]))
        eval_types = self.test_cfg.get('eval_types', []) if self.test_cfg else []
        with_caption = kwargs.get('with_caption', False) or ('cap_results' in eval_types)
        caption_generation_results = None
        if with_caption:
            from .caption_search import beam_search
            caption_generation_results = beam_search(self, mask_cls_emb_results, BOS_TOKEN, EOS_TOKEN,
                                                     max_len=35, beam_width=7,
                                                     logging=kwargs.get('logging', False))
        att = None
        if kwargs.get('with_att', False):
            nouns_embs = self.bert_embeddings(kwargs['nouns_ids']).squeeze(0)
            att = torch.matmul(mask_cls_emb_results[0], nouns_embs.t())
        return assigned_labels, mask_cls_emb_results, masks, caption_generation_results, att


def _all_gather(out_list, tensor):
    """dist.all_gather; over gloo (debug / single-device multi-rank runs: RCCL refuses two ranks on one GPU) device tensors are
    staged through the host, because gloo implements only broadcast and all_reduce for them."""
    if tensor.is_cuda and dist.get_backend() == 'gloo':
        host = [torch.empty(o.shape, dtype=o.dtype) for o in out_list]
        dist.all_gather(host, tensor.cpu())
        for o, h in zip(out_list, host):
            o.copy_(h)
        return
    dist.all_gather(out_list, tensor)


def reduce_mean(tensor):
    """[3P] mmdet reduce_mean: all_reduce(x / world); identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor
