"""Oracle restatement (CPU, plain torch, seq-first exactly like upstream) of the [3P] mmcv 1.7.1 /
mmdet 2.28.2 modules the reference builds by `type=` string. TEST INFRASTRUCTURE ONLY.

Parameter names equal upstream's, so a product `state_dict` loads into these
modules and vice versa. Semantics per SURVEY.md Appendix A ("parity unpinned" by the reference: its
source tree holds neither these modules nor tests of them; they are pinned against PyTorch primitives
-- nn.MultiheadAttention, F.grid_sample, F.interpolate -- which is what they are written with).
These classes also serve as the leaf ops of the import shim that lets the reference's OWN
`Mask2FormerHeadOpen` code run in tests/golden/make_golden.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class AttrDict(dict):
    """minimal attribute dict (the reference reads cfg.transformerlayers.attn_cfgs.num_heads)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def attrify(o):
    if isinstance(o, dict):
        return AttrDict({k: attrify(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return type(o)(attrify(v) for v in o)
    return o


# ---- SinePositionalEncoding (mmdet) ----------------------------------------------------
class SinePositionalEncoding(nn.Module):

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0., init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask):
        not_mask = 1 - mask.to(torch.int)
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature**(2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# ---- FFN (mmcv) -------------------------------------------------------------------------------------
class FFN(nn.Module):

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=None, ffn_drop=0.,
                 dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__()
        layers, c = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(c, feedforward_channels), nn.ReLU(inplace=True),
                                        nn.Dropout(ffn_drop)))
            c = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out


# ---- MultiScaleDeformableAttention (mmcv) -------------------------------------------------
class MultiScaleDeformableAttention(nn.Module):

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = \
            embed_dims, num_heads, num_levels, num_points
        self.batch_first = batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nq, _ = query.shape
        nv = value.shape[1]
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        H, L, P = self.num_heads, self.num_levels, self.num_points
        off = self.sampling_offsets(query).view(bs, nq, H, L, P, 2)
        aw = self.attention_weights(query).view(bs, nq, H, L * P).softmax(-1).view(bs, nq, H, L, P)
        norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        out = ops.msda_core(value, spatial_shapes, loc, aw)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


# ---- MultiheadAttention wrapper (mmcv) ---------------------------------------------------
class MultiheadAttention(nn.Module):

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None, init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__()
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


_ATTN = {'MultiScaleDeformableAttention': MultiScaleDeformableAttention, 'MultiheadAttention': MultiheadAttention}


# ---- BaseTransformerLayer / sequences (mmcv / mmdet) ---------------------------------------------------
class BaseTransformerLayer(nn.Module):

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None, norm_cfg=None, init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__()
        ffn_cfgs = dict(ffn_cfgs or dict(embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0.))
        for old, new in (('feedforward_channels', 'feedforward_channels'), ('ffn_dropout', 'ffn_drop'),
                         ('ffn_num_fcs', 'num_fcs')):
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        self.operation_order = tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        n_attn = sum(op in ('self_attn', 'cross_attn') for op in operation_order)
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [dict(attn_cfgs) for _ in range(n_attn)]
        self.num_attn = n_attn
        self.attentions = nn.ModuleList()
        for c in attn_cfgs:
            c = dict(c)
            c.setdefault('batch_first', batch_first)
            self.attentions.append(_ATTN[c.pop('type')](**c))
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count('ffn')):
            c = dict(ffn_cfgs)
            c.pop('type', None)
            c.setdefault('embed_dims', self.embed_dims)
            self.ffns.append(FFN(**c))
        self.norms = nn.ModuleList([nn.LayerNorm(self.embed_dims) for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        ni = ai = fi = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [attn_masks.clone() for _ in range(self.num_attn)]
        for op in self.operation_order:
            if op == 'self_attn':
                query = self.attentions[ai](query, query, query, identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[ai],
                                            key_padding_mask=query_key_padding_mask, **kwargs)
                ai += 1
                identity = query
            elif op == 'norm':
                query = self.norms[ni](query)
                ni += 1
            elif op == 'cross_attn':
                query = self.attentions[ai](query, key, value, identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=key_pos, attn_mask=attn_masks[ai],
                                            key_padding_mask=key_padding_mask, **kwargs)
                ai += 1
                identity = query
            elif op == 'ffn':
                query = self.ffns[fi](query, identity if self.pre_norm else None)
                fi += 1
        return query


class DetrTransformerDecoderLayer(BaseTransformerLayer):

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None, act_cfg=None,
                 norm_cfg=None, ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, ffn_num_fcs=ffn_num_fcs,
                         **kwargs)


_LAYER = {'BaseTransformerLayer': BaseTransformerLayer, 'DetrTransformerDecoderLayer': DetrTransformerDecoderLayer}


class TransformerLayerSequence(nn.Module):

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None, post_norm_cfg=dict(type='LN'),
                 return_intermediate=False, type=None):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(num_layers):
            c = dict(transformerlayers)
            self.layers.append(_LAYER[c.pop('type')](**c))
        self.num_layers = num_layers
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, **kwargs):
        for layer in self.layers:
            query = layer(query, key, value, **kwargs)
        return query


class DetrTransformerEncoder(TransformerLayerSequence):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.post_norm = nn.LayerNorm(self.embed_dims) if self.pre_norm else None


class DetrTransformerDecoder(TransformerLayerSequence):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.post_norm = nn.LayerNorm(self.embed_dims)


_SEQ = {'DetrTransformerEncoder': DetrTransformerEncoder, 'DetrTransformerDecoder': DetrTransformerDecoder}


def build_transformer_layer_sequence(cfg):
    c = dict(cfg)
    return _SEQ[c.pop('type')](**c)


def build_positional_encoding(cfg):
    c = dict(cfg)
    assert c.pop('type') == 'SinePositionalEncoding'
    return SinePositionalEncoding(**c)


# ---- MSDeformAttnPixelDecoder (mmdet) -------------------------------------------------------
class ConvModule(nn.Module):

    def __init__(self, cin, cout, k, padding=0, bias=True, norm_groups=None, act=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=bias)
        self.gn = nn.GroupNorm(norm_groups, cout) if norm_groups else None
        self.act = act

    def forward(self, x):
        x = self.conv(x)
        if self.gn is not None:
            x = self.gn(x)
        return F.relu(x) if self.act else x


class MSDeformAttnPixelDecoder(nn.Module):

    def __init__(self, in_channels=(256, 512, 1024, 2048), strides=(4, 8, 16, 32), feat_channels=256,
                 out_channels=256, num_outs=3, norm_cfg=None, act_cfg=None, encoder=None,
                 positional_encoding=None, init_cfg=None, type=None):
        super().__init__()
        groups = (norm_cfg or dict(num_groups=32))['num_groups']
        self.strides = list(strides)
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = encoder['transformerlayers']['attn_cfgs']['num_levels']
        n_in, n_enc = self.num_input_levels, self.num_encoder_levels
        self.input_convs = nn.ModuleList(
            [ConvModule(in_channels[i], feat_channels, 1, bias=True, norm_groups=groups)
             for i in range(n_in - 1, n_in - n_enc - 1, -1)])
        self.encoder = build_transformer_layer_sequence(encoder)
        self.postional_encoding = build_positional_encoding(positional_encoding)
        self.level_encoding = nn.Embedding(n_enc, feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        for i in range(n_in - n_enc - 1, -1, -1):
            self.lateral_convs.append(ConvModule(in_channels[i], feat_channels, 1, bias=False, norm_groups=groups))
            self.output_convs.append(ConvModule(feat_channels, feat_channels, 3, padding=1, bias=False,
                                                norm_groups=groups, act=True))
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, 1)
        self.num_outs = num_outs

    def init_weights(self):  # the reference head calls it (mask2former_head.py:236); weights are loaded after
        pass

    def forward(self, feats):
        bs = feats[0].shape[0]
        inputs, masks, poss, shapes, refs = [], [], [], [], []
        for i in range(self.num_encoder_levels):
            lvl = self.num_input_levels - i - 1
            feat = feats[lvl]
            proj = self.input_convs[i](feat)
            h, w = feat.shape[-2:]
            pm = feat.new_zeros((bs, h, w), dtype=torch.bool)
            pos = self.postional_encoding(pm) + self.level_encoding.weight[i].view(1, -1, 1, 1)
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                    indexing='ij')
            s = self.strides[lvl]
            pts = torch.stack([(xs.reshape(-1) + 0.5) * s, (ys.reshape(-1) + 0.5) * s], -1)
            refs.append(pts / (feat.new_tensor([[w, h]]) * s))
            inputs.append(proj.flatten(2).permute(2, 0, 1))
            poss.append(pos.flatten(2).permute(2, 0, 1))
            masks.append(pm.flatten(1))
            shapes.append((h, w))
        padding_masks = torch.cat(masks, dim=1)
        enc_in = torch.cat(inputs, dim=0)
        lvl_pos = torch.cat(poss, dim=0)
        spatial_shapes = torch.as_tensor(shapes, dtype=torch.long)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1, )), spatial_shapes.prod(1).cumsum(0)[:-1]))
        reference_points = torch.cat(refs, dim=0)[None, :, None].repeat(bs, 1, self.num_encoder_levels, 1)
        memory = self.encoder(query=enc_in, key=None, value=None, query_pos=lvl_pos, key_pos=None,
                              attn_masks=None, key_padding_mask=None, query_key_padding_mask=padding_masks,
                              spatial_shapes=spatial_shapes, reference_points=reference_points,
                              level_start_index=level_start_index)
        memory = memory.permute(1, 2, 0)
        outs = torch.split(memory, [h * w for h, w in shapes], dim=-1)
        outs = [x.reshape(bs, -1, shapes[i][0], shapes[i][1]) for i, x in enumerate(outs)]
        for i in range(self.num_input_levels - self.num_encoder_levels - 1, -1, -1):
            cur = self.lateral_convs[i](feats[i])
            y = cur + F.interpolate(outs[-1], size=cur.shape[-2:], mode='bilinear', align_corners=False)
            outs.append(self.output_convs[i](y))
        return self.mask_feature(outs[-1]), outs[:self.num_outs]


def build_plugin_layer(cfg, postfix='', **kwargs):
    c = dict(cfg)
    t = c.pop('type')
    assert t == 'MSDeformAttnPixelDecoder', t
    return t.lower() + str(postfix), MSDeformAttnPixelDecoder(**c, **kwargs)
