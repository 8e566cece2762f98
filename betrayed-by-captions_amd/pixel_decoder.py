"""MSDeformAttn pixel decoder (the `pixel_decoder=dict(type='MSDeformAttnPixelDecoder', ...)` of
configs/instance/coco_b48n17.py:38-70, built at open_set/models/mask2former_head.py:112-117 and called
at :787) and the `[3P]` building blocks its config names: `DetrTransformerEncoder`,
`BaseTransformerLayer`, `MultiScaleDeformableAttention`, `FFN`, `SinePositionalEncoding`.

Parameter names follow the upstream mmcv 1.7.1 / mmdet 2.28.2 layout so a
reference checkpoint's `state_dict` loads unchanged. The execution is MI355X-first:
  * activations are batch-first (B, N, C) end to end (no seq-first permutes);
  * per encoder layer the `sampling_offsets` and `attention_weights` linears run as ONE 256->288 GEMM
    whose raw output feeds `cgg_msda_forward_hostlevels(fused=1)`: reference-point add, offset
    normalisation and the softmax over levels*points happen in the HIP kernel prologue, so the
    (B,N,8,3,4,2) locations / (B,N,8,3,4) weights are never written to HBM;
  * sine positional encodings and reference points depend only on the level shapes -> cached.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, runtime
from .registry import (ATTENTION, FEEDFORWARD_NETWORK, PLUGIN_LAYERS, POSITIONAL_ENCODING,
                       TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_attention,
                       build_feedforward_network, build_positional_encoding,
                       build_transformer_layer, build_transformer_layer_sequence)

# throughput mode: encoder FFN + residual LayerNorm as one HIP launch (ops.encoder_ffn_ln); CGG_FUSED_FFN=0 restores the
# two library GEMMs + LayerNorm pass for A/B measurements
FUSED_FFN = os.environ.get('CGG_FUSED_FFN', '1') != '0'
FUSED_TAIL = os.environ.get('CGG_FUSED_TAIL', '1') != '0'   # ... preceded by output_proj + its residual LayerNorm
FUSED_PROJ = os.environ.get('CGG_FUSED_PROJ', '1') != '0'
FUSED_TRAIN_MSDA = os.environ.get('CGG_FUSED_TRAIN_MSDA', '1') != '0'   # training: MSDeformAttn prologue inside the kernels (fwd + bwd)
FUSED_TRAIN_LN = os.environ.get('CGG_FUSED_TRAIN_LN', '1') != '0'   # training: residual + LayerNorm as one-pass HIP fwd / bwd
VALUE_HEAD_MAJOR = os.environ.get('CGG_VALUE_HEAD_MAJOR', '1') != '0'   # value written (B, 8, N, 32) for the MSDeformAttn gather
# x3a encoder stream: value / offsets / logits of a layer from ONE GEMM (needs the strided-value sampling kernel: not with the
# generic MSDeformAttn kernels forced)
MERGED_PROJ = os.environ.get('CGG_MERGED_PROJ', '1') != '0' and not os.environ.get('CGG_MSDA_GENERIC')
POS_IN_PROJ = os.environ.get('CGG_POS_IN_PROJ', '1') != '0'   # the projection kernel forms x + pos from a bf16 pos table   # value_proj + offsets/weights GEMMs as one HIP launch


# ------------------------------------------------------------------------------------------------
def build_norm(cfg, num_features):
    """(name, module) for the norm types the shipped configs use: GN / BN / LN."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    if t == 'GN':
        m = nn.GroupNorm(cfg.pop('num_groups'), num_features, **cfg)
        name = 'gn'
    elif t in ('BN', 'BN2d'):
        m = nn.BatchNorm2d(num_features, **cfg)
        name = 'bn'
    elif t == 'LN':
        m = nn.LayerNorm(num_features, **cfg)
        name = 'ln'
    else:
        raise KeyError(f'Unrecognized norm type {t}')
    for p in m.parameters():
        p.requires_grad = requires_grad
    return name, m


def build_act(cfg):
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t == 'ReLU':
        return nn.ReLU(**cfg)
    if t == 'GELU':
        cfg.pop('inplace', None)
        return nn.GELU(**cfg)
    raise KeyError(f'Unrecognized activation type {t}')


class ConvModule(nn.Module):
    """conv (+ norm) (+ act) with the [3P] mmcv ConvModule attribute names: `.conv`, `.gn` / `.bn`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias='auto',
                 norm_cfg=None, act_cfg=dict(type='ReLU')):
        super().__init__()
        if bias == 'auto':
            bias = norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              bias=bias)
        self.norm_name = None
        if norm_cfg is not None:
            self.norm_name, norm = build_norm(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        self.activate = build_act(dict(act_cfg, inplace=True)) if act_cfg is not None else None

    def forward(self, x):
        if runtime.x3_train_conv3x3_ok(self.conv, x):
            x = runtime.conv3x3_x3_train(x, self.conv)          # parity-mode training: x3 forward / grad-input / grad-weight
        else:
            x = self.conv(x.contiguous())
        norm = getattr(self, self.norm_name) if self.norm_name is not None else None
        if isinstance(norm, nn.GroupNorm) and x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad):
            # inference: HIP GroupNorm (+ fused ReLU), full-chip two-pass reduction
            relu = isinstance(self.activate, nn.ReLU)
            x = ops.group_norm(x.float(), norm.weight, norm.bias, norm.num_groups, norm.eps, relu)
            if self.activate is not None and not relu:
                x = self.activate(x)
            return x
        if norm is not None:
            x = norm(x.float())
        if self.activate is not None:
            x = self.activate(x)
        return x


def _kaiming_uniform_a1(conv):
    """[3P] caffe2_xavier_init: kaiming uniform, a=1, fan_in, leaky_relu; bias 0."""
    nn.init.kaiming_uniform_(conv.weight, a=1, mode='fan_in', nonlinearity='leaky_relu')
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


# ------------------------------------------------------------------------------------------------
@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding(nn.Module):
    """[3P] mmdet SinePositionalEncoding (DETR formula).
    forward(mask (B,H,W) bool, True = padded) -> (B, 2*num_feats, H, W)."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0., init_cfg=None):
        super().__init__()
        if normalize:
            assert isinstance(scale, (float, int)), \
                f'when normalize is set, scale should be provided and in float or int type, found {type(scale)}'
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._cache = {}

    def forward(self, mask):
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature**(2 * torch.div(dim_t, 2, rounding_mode='floor') / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)

    def flat_unpadded(self, h, w, device):
        """(h*w, 2*num_feats) encoding of an all-valid (h, w) map (batch independent); cached."""
        key = (h, w, str(device))
        pe = self._cache.get(key)
        if pe is None:
            m = torch.zeros((1, h, w), dtype=torch.bool, device=device)
            pe = self.forward(m)[0].flatten(1).t().contiguous()
            self._cache[key] = pe
        return pe


# ------------------------------------------------------------------------------------------------
@FEEDFORWARD_NETWORK.register_module()
class FFN(nn.Module):
    """[3P] mmcv FFN: Linear-act-drop x (num_fcs-1), Linear, drop; `identity + out`."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2, f'num_fcs should be no less than 2. got {num_fcs}.'
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, feedforward_channels), build_act(act_cfg),
                                        nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Dropout(dropout_layer['drop_prob']) if dropout_layer else nn.Identity()
        self.add_identity = add_identity

    def forward_noidentity(self, x):
        """Training, parity mode: W2 relu(W1 x + b1) + b2 WITHOUT the residual (it is added inside the fused LayerNorm); the two
        linears go through `runtime.linear` (x3 forward / grad-input / grad-weight for the encoder's row counts)."""
        l0, l1 = self.layers[0][0], self.layers[1]
        if runtime.x3_train_ffn_ok(x, l0.weight, l0.bias, l1.weight, l1.bias):
            return runtime.ffn_x3_train(x, l0.weight, l0.bias, l1.weight, l1.bias)      # one autograd node: ReLU / its backward / amax fused
        h = torch.relu_(runtime.linear(x, l0.weight, l0.bias))
        return runtime.linear(h, l1.weight, l1.bias)

    def forward_bf16_noidentity(self, x):
        """Training, bf16 mode: W2 relu(W1 x + b1) + b2 in bf16 WITHOUT the residual (it is added inside the fused LayerNorm)."""
        h = F.relu(runtime.linear_bf16_train(x, self.layers[0][0].weight, self.layers[0][0].bias))
        return runtime.linear_bf16_train(h, self.layers[1].weight, self.layers[1].bias)

    def plain_relu_ffn(self):
        return (len(self.layers) == 3 and isinstance(self.layers[0][1], nn.ReLU) and self.layers[0][2].p == 0
                and self.layers[2].p == 0 and self.add_identity
                and (isinstance(self.dropout_layer, nn.Identity) or getattr(self.dropout_layer, 'p', 1) == 0))

    def forward(self, x, identity=None):
        if len(self.layers) == 3 and isinstance(self.layers[0][1], nn.ReLU) and self.layers[0][2].p == 0 \
                and self.layers[2].p == 0:
            if runtime.is_bf16() and torch.is_grad_enabled() and x.is_cuda:
                # training in throughput mode: ONE autocast region for both projections, so the hidden activation
                # (B x N x 1024: 1.4 GB in f32 at configs[2]) stays bf16 between them -- `runtime.linear` would widen it
                # to f32, apply the ReLU there and narrow it again: ~5.6 GB of extra traffic per layer and direction
                h = F.relu(runtime.linear_bf16_train(x, self.layers[0][0].weight, self.layers[0][0].bias))
                out = runtime.linear_bf16_train(h, self.layers[1].weight, self.layers[1].bias).float()
            else:
                h = torch.relu_(runtime.linear(x, self.layers[0][0].weight, self.layers[0][0].bias))
                out = runtime.linear(h, self.layers[1].weight, self.layers[1].bias)
        else:
            with runtime.autocast():
                out = self.layers(x)
            out = out.float()
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


@ATTENTION.register_module()
class MultiScaleDeformableAttention(nn.Module):
    """[3P] mmcv MultiScaleDeformableAttention on the HIP MSDeformAttn kernels."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}')
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels, self.num_heads, self.num_points = \
            embed_dims, num_levels, num_heads, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid_init.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    # -- fast path used by MSDeformAttnPixelDecoder (batch-first, fused prologue, forward-only) ----
    def forward_fused(self, src, src_pos, ref_points, level_hw, level_start, add_identity=True, src16=None, src_pos16=None):
        """src (B,N,C) f32 (value input and identity), src_pos = src + pos (query input),
        ref_points (N,2). Returns identity + dropout(output_proj(msda)); add_identity=False (training, bf16 mode, no dropout):
        the bf16 output projection alone, for the fused residual + LayerNorm (`ops.add_layernorm_train`)."""
        B, N, C = src.shape
        H, D = self.num_heads, C // self.num_heads
        w_cat = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0)
        b_cat = torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0)
        if src16 is not None:
            # training, bf16 mode: the bf16 copies of src / src + pos come from the previous LayerNorm kernel (no cast passes)
            value = runtime.linear_bf16_train(src16, self.value_proj.weight, self.value_proj.bias).float()
            offs_logits = runtime.linear_bf16_train(src_pos16, w_cat, b_cat).float()
        elif runtime.is_bf16() and not torch.is_grad_enabled():
            value = F.linear(src.to(torch.bfloat16), runtime.cast_cached(self.value_proj.weight),
                             runtime.cast_cached(self.value_proj.bias))      # bf16 values for the gather
        else:
            value = runtime.linear(src, self.value_proj.weight, self.value_proj.bias)
        if src16 is not None:
            pass
        elif runtime.is_bf16() and torch.is_grad_enabled():
            # training in throughput mode: bf16 operands like the inference stream (whose kernel reads bf16 offset rows);
            # the f32 GEMM and its two backward GEMMs cost 11 ms per step at configs[2]
            offs_logits = runtime.linear(src_pos, w_cat, b_cat)
        elif runtime.x3_train_linear_ok(src_pos, w_cat):
            # parity-mode training: f32-class on the x3 kernels (forward, grad-input, grad-weight) like the other encoder linears
            offs_logits = runtime.linear(src_pos, w_cat, b_cat)
        else:
            # offsets / logits decide WHERE to sample: f32 in parity mode and in the f32-value inference path
            offs_logits = F.linear(src_pos, w_cat, b_cat)
        value = value.view(B, N, H, D)
        if (torch.is_grad_enabled() and (value.requires_grad or offs_logits.requires_grad) and FUSED_TRAIN_MSDA
                and value.is_cuda and self.num_levels * self.num_points <= 16
                and offs_logits.shape[-1] == 3 * H * self.num_levels * self.num_points):
            # training: the prologue (loc = ref + off / (W, H), softmax) stays inside the kernels, forward and backward
            out = ops.MSDeformAttnRowsFunction.apply(value.float(), offs_logits.float(), ref_points, level_hw, level_start,
                                                     self.num_points)
        elif torch.is_grad_enabled() and (value.requires_grad or offs_logits.requires_grad):
            # training: un-fused prologue in torch so autograd reaches the linears; the sampling core
            # and its backward are still the HIP kernels (MultiScaleDeformableAttnFunction)
            L, P = self.num_levels, self.num_points
            n_off = H * L * P * 2
            offs = offs_logits[..., :n_off].view(B, N, H, L, P, 2)
            aw = offs_logits[..., n_off:].view(B, N, H, L * P).softmax(-1).view(B, N, H, L, P)
            norm = offs_logits.new_tensor([[w, h] for h, w in level_hw])
            loc = ref_points[None, :, None, None, None, :] + offs / norm[None, None, None, :, None, :]
            shapes = torch.tensor(level_hw, dtype=torch.int64, device=src.device)
            starts = torch.tensor(level_start, dtype=torch.int64, device=src.device)
            out = ops.MultiScaleDeformableAttnFunction.apply(value.float().contiguous(), shapes, starts,
                                                             loc.contiguous(), aw.contiguous(),
                                                             self.im2col_step)
        else:
            out = ops.msda_forward_fused(value.contiguous(), level_hw, level_start,
                                         offs_logits.contiguous(), ref_points, self.num_points)
        if not add_identity:
            if runtime.is_bf16():
                return runtime.linear_bf16_train(out, self.output_proj.weight, self.output_proj.bias)
            return runtime.linear(out, self.output_proj.weight, self.output_proj.bias)
        out = runtime.linear(out, self.output_proj.weight, self.output_proj.bias)
        return src + self.dropout(out)

    # -- [3P] signature (seq-first unless batch_first); differentiable through the HIP autograd op --
    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == num_value
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        sampling_offsets = self.sampling_offsets(query).view(
            bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).view(
            bs, num_query, self.num_heads, self.num_levels * self.num_points)
        attention_weights = attention_weights.softmax(-1).view(
            bs, num_query, self.num_heads, self.num_levels, self.num_points)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / self.num_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f'Last dim of reference_points must be 2 or 4, but get '
                             f'{reference_points.shape[-1]} instead.')
        output = ops.MultiScaleDeformableAttnFunction.apply(
            value.contiguous(), spatial_shapes, level_start_index, sampling_locations.contiguous(),
            attention_weights.contiguous(), self.im2col_step)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return self.dropout(output) + identity


# ------------------------------------------------------------------------------------------------
@TRANSFORMER_LAYER.register_module()
class BaseTransformerLayer(nn.Module):
    """[3P] mmcv BaseTransformerLayer: attentions / ffns / norms run in `operation_order`."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024,
                                                     num_fcs=2, ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        ffn_cfgs = dict(ffn_cfgs)
        for ori_name, new_name in dict(feedforward_channels='feedforward_channels', ffn_dropout='ffn_drop',
                                       ffn_num_fcs='num_fcs').items():
            if ori_name in kwargs:
                ffn_cfgs[new_name] = kwargs[ori_name]
        assert set(operation_order) & {'self_attn', 'norm', 'ffn', 'cross_attn'} == set(operation_order), \
            f"The operation_order of {self.__class__.__name__} should contains all four operation type " \
            f"{['self_attn', 'norm', 'ffn', 'cross_attn']}"
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [dict(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs), \
                f'The length of attn_cfg {num_attn} is not consistent with the number of attention' \
                f'in operation_order {operation_order}.'
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.batch_first = batch_first
        self.attentions = nn.ModuleList()
        index = 0
        for op in operation_order:
            if op in ('self_attn', 'cross_attn'):
                cfg = dict(attn_cfgs[index])
                if 'batch_first' in cfg:
                    assert self.batch_first == cfg['batch_first']
                else:
                    cfg['batch_first'] = self.batch_first
                attention = build_attention(cfg)
                attention.operation_name = op
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [dict(ffn_cfgs) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for i in range(num_ffns):
            cfg = dict(ffn_cfgs[i])
            cfg.setdefault('embed_dims', self.embed_dims)
            assert cfg['embed_dims'] == self.embed_dims
            self.ffns.append(build_feedforward_network(cfg, dict(type='FFN')))
        self.norms = nn.ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(build_norm(norm_cfg, self.embed_dims)[1])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [attn_masks.clone() for _ in range(self.num_attn)]
        else:
            assert len(attn_masks) == self.num_attn, \
                f'The length of attn_masks {len(attn_masks)} must be equal to the number of ' \
                f'attention in operation_order {self.num_attn}'
        for layer in self.operation_order:
            if layer == 'self_attn':
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[attn_index], key_padding_mask=key_padding_mask,
                    **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


class TransformerLayerSequence(nn.Module):
    """[3P] mmcv TransformerLayerSequence: `num_layers` copies of `transformerlayers`."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [dict(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList([build_transformer_layer(cfg) for cfg in transformerlayers])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        for layer in self.layers:
            query = layer(query, key, value, query_pos=query_pos, key_pos=key_pos, attn_masks=attn_masks,
                          query_key_padding_mask=query_key_padding_mask, key_padding_mask=key_padding_mask,
                          **kwargs)
        return query


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetrTransformerEncoder(TransformerLayerSequence):
    """[3P] mmdet DetrTransformerEncoder: post_norm only when the layers are pre-norm."""

    def __init__(self, *args, post_norm_cfg=dict(type='LN'), **kwargs):
        super().__init__(*args, **kwargs)
        if post_norm_cfg is not None:
            self.post_norm = build_norm(post_norm_cfg, self.embed_dims)[1] if self.pre_norm else None
        else:
            assert not self.pre_norm, f'Use prenorm in {self.__class__.__name__},Please specify post_norm_cfg'
            self.post_norm = None

    def forward(self, *args, **kwargs):
        x = super().forward(*args, **kwargs)
        if self.post_norm is not None:
            x = self.post_norm(x)
        return x


# ------------------------------------------------------------------------------------------------
@PLUGIN_LAYERS.register_module()
class MSDeformAttnPixelDecoder(nn.Module):
    """[3P] mmdet MSDeformAttnPixelDecoder.
    forward(feats: 4 maps, strides 4..32) -> (mask_feature (B,C,H/4,W/4), [3 memories low->high res])."""

    def __init__(self, in_channels=[256, 512, 1024, 2048], strides=[4, 8, 16, 32], feat_channels=256,
                 out_channels=256, num_outs=3, norm_cfg=dict(type='GN', num_groups=32),
                 act_cfg=dict(type='ReLU'), encoder=None, positional_encoding=dict(
                     type='SinePositionalEncoding', num_feats=128, normalize=True), init_cfg=None):
        super().__init__()
        self.strides = strides
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = encoder['transformerlayers']['attn_cfgs']['num_levels']
        assert self.num_encoder_levels >= 1, 'num_levels in attn_cfgs must be at least one'
        self.input_convs = nn.ModuleList()
        for i in range(self.num_input_levels - 1, self.num_input_levels - self.num_encoder_levels - 1, -1):
            self.input_convs.append(ConvModule(in_channels[i], feat_channels, kernel_size=1,
                                               norm_cfg=norm_cfg, act_cfg=None, bias=True))
        self.encoder = build_transformer_layer_sequence(encoder)
        self.postional_encoding = build_positional_encoding(positional_encoding)  # (sic) upstream name
        self.level_encoding = nn.Embedding(self.num_encoder_levels, feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        self.use_bias = norm_cfg is None
        for i in range(self.num_input_levels - self.num_encoder_levels - 1, -1, -1):
            self.lateral_convs.append(ConvModule(in_channels[i], feat_channels, kernel_size=1,
                                                 bias=self.use_bias, norm_cfg=norm_cfg, act_cfg=None))
            self.output_convs.append(ConvModule(feat_channels, feat_channels, kernel_size=3, stride=1,
                                                padding=1, bias=self.use_bias, norm_cfg=norm_cfg,
                                                act_cfg=act_cfg))
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, kernel_size=1, stride=1, padding=0)
        self.num_outs = num_outs
        # throughput-mode (bf16) encoder stream: keep the residual between LayerNorms in bf16 (the rows the GEMMs read
        # anyway) instead of a separate f32 copy; False restores the f32 residual stream
        import os
        self.stream_residual_bf16 = not os.environ.get('CGG_STREAM_RES_F32')
        self._ref_cache = {}

    def init_weights(self):
        for i in range(self.num_encoder_levels):
            nn.init.xavier_uniform_(self.input_convs[i].conv.weight, gain=1)
            nn.init.constant_(self.input_convs[i].conv.bias, 0)
        for i in range(self.num_input_levels - self.num_encoder_levels):
            _kaiming_uniform_a1(self.lateral_convs[i].conv)
            _kaiming_uniform_a1(self.output_convs[i].conv)
        _kaiming_uniform_a1(self.mask_feature)
        nn.init.normal_(self.level_encoding.weight, mean=0, std=1)
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        for layer in self.encoder.layers:
            for attn in layer.attentions:
                if isinstance(attn, MultiScaleDeformableAttention):
                    attn.init_weights()

    def _reference_points(self, level_hw, device):
        """pixel centres ((x+.5)/w, (y+.5)/h), levels concatenated low->high res; (N,2)."""
        key = (tuple(level_hw), str(device))
        ref = self._ref_cache.get(key)
        if ref is None:
            pts = []
            for h, w in level_hw:
                ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                                        torch.arange(w, device=device, dtype=torch.float32), indexing='ij')
                pts.append(torch.stack([(xs.flatten() + 0.5) / w, (ys.flatten() + 0.5) / h], -1))
            ref = torch.cat(pts, 0).contiguous()
            self._ref_cache[key] = ref
        return runtime.keepalive(ref)

    @staticmethod
    def _stream_ok(layer):
        f = layer.ffns[0]
        return (len(f.layers) == 3 and isinstance(f.layers[0][1], nn.ReLU) and f.add_identity
                and isinstance(layer.attentions[0], MultiScaleDeformableAttention)
                and layer.attentions[0].dropout.p == 0)

    def _proj_fused(self, C):
        """True when every encoder layer's input projections fit `ops.encoder_proj` (256 -> 256 + 256..384 columns)."""
        def nc(layer):
            a = layer.attentions[0]
            return a.sampling_offsets.out_features + a.attention_weights.out_features
        return FUSED_PROJ and C == 256 and all(nc(l) % 32 == 0 and 256 <= nc(l) <= 384 for l in self.encoder.layers)

    def _encoder_stream_bf16(self, src, pos, ref, level_hw, level_start, x16=None, xp16=None, kv_tables=None):
        """Throughput-mode encoder over the (B, 21504, 256) bf16 stream, THREE launches per layer (round 2):
        `ops.encoder_proj` (value_proj + offsets / attention-weight projections; forms `x + pos` from a bf16 pos table),
        `ops.msda_forward_fused_bf16` (softmax / sampling locations in the prologue) and `ops.encoder_layer_tail`
        (output_proj + residual LayerNorm + FFN + residual LayerNorm; the last layer also emits the query decoder's
        K / V operands). Each stage falls back to its library-GEMM form (4 GEMMs + 2 fused residual-LayerNorm passes, the
        round-1 path) when its shape is not built or its CGG_FUSED_* switch is off. `src` (f32 stream) is only read
        with the f32 residual option (CGG_STREAM_RES_F32)."""
        bf = torch.bfloat16
        cc = runtime.cast_cached
        if x16 is None:
            x16 = src.to(bf)
            xp16 = (src + pos[None]).to(bf)
        B, N, C = (src if src is not None else x16).shape
        n_layers = len(self.encoder.layers)
        proj_ok = self._proj_fused(C)
        # with the projection kernel forming `x + pos` itself (bf16 pos table), the layer tails stop writing those rows
        pos16 = runtime.derived_cached('enc_pos16', (pos,), lambda: pos.to(bf).contiguous()) if proj_ok and POS_IN_PROJ else None
        for li, layer in enumerate(self.encoder.layers):
            attn = layer.attentions[0]
            H, D = attn.num_heads, C // attn.num_heads
            so, aw = attn.sampling_offsets, attn.attention_weights
            w_cat = runtime.derived_cached('msda_wcat', (so.weight, aw.weight),
                                           lambda: torch.cat([so.weight, aw.weight], 0).to(bf).contiguous())
            b_cat = runtime.derived_cached('msda_bcat', (so.bias, aw.bias),
                                           lambda: torch.cat([so.bias, aw.bias], 0).to(bf).contiguous())
            if proj_ok:
                # value_proj + [sampling_offsets; attention_weights] as one launch over the bf16 rows
                vp = attn.value_proj
                wvp = runtime.derived_cached('msda_wvp', (vp.weight,), lambda: ops.pack_encoder_proj_weight(vp.weight))
                wcp = runtime.derived_cached('msda_wcp', (so.weight, aw.weight),
                                             lambda: ops.pack_encoder_proj_weight(torch.cat([so.weight, aw.weight], 0)))
                bcf = runtime.derived_cached('msda_bcf', (so.bias, aw.bias),
                                             lambda: torch.cat([so.bias, aw.bias], 0).float().contiguous())
                hm = VALUE_HEAD_MAJOR and H == 8 and D == 32 and len(level_hw) == 3 and attn.num_points == 4
                value, offs = ops.encoder_proj(x16, xp16, wvp, vp.bias, wcp, bcf, pos16=pos16 if xp16 is None else None,
                                               value_head_major=hm)
                if not hm:
                    value = value.view(B, N, H, D)
            else:
                hm = False
                value = F.linear(x16, cc(attn.value_proj.weight), cc(attn.value_proj.bias)).view(B, N, H, D)
                offs = F.linear(xp16, w_cat, b_cat)
            a16 = ops.msda_forward_fused_bf16(value, level_hw, level_start, offs, ref, attn.num_points, head_major=hm)
            n0, n1 = layer.norms
            ffn = layer.ffns[0]
            last = li == n_layers - 1
            fused_ffn = (self.stream_residual_bf16 and FUSED_FFN and ffn.layers[0][0].out_features % 256 == 0
                         and (not last or kv_tables is not None))
            if fused_ffn and FUSED_TAIL and C == 256:
                # output_proj + residual LayerNorm + FFN + residual LayerNorm as ONE launch: neither the projection output, nor
                # the first LayerNorm's rows, nor the (B, N, 1024) hidden activation reach memory
                op, fc1, fc2 = attn.output_proj, ffn.layers[0][0], ffn.layers[1]
                wop = runtime.derived_cached('msda_wop', (op.weight,), lambda: ops.pack_linear_weight(op.weight))
                w1p = runtime.derived_cached('ffn_w1p', (fc1.weight,), lambda: ops.pack_linear_weight(fc1.weight))
                w2p = runtime.derived_cached('ffn_w2p', (fc2.weight,), lambda: ops.pack_linear_weight(fc2.weight))
                norm0, norm1 = (n0.weight, n0.bias, n0.eps), (n1.weight, n1.bias, n1.eps)
                if last:
                    src, m16, mp16 = ops.encoder_layer_tail(a16, x16, wop, op.bias, norm0, w1p, fc1.bias, w2p, fc2.bias, norm1,
                                                            kv=(kv_tables[0], kv_tables[1], level_start), want_f32=True)
                    return src, (m16, mp16)
                _, x16, xp16 = ops.encoder_layer_tail(a16, x16, wop, op.bias, norm0, w1p, fc1.bias, w2p, fc2.bias, norm1,
                                                      pos=pos, want_bf16=True, want_pos=pos16 is None)
                src = x16
                continue
            o16 = F.linear(a16, cc(attn.output_proj.weight), cc(attn.output_proj.bias))
            if self.stream_residual_bf16:
                # residual stream in bf16: the LayerNorm reads the same bf16 rows the GEMMs read (66 instead of 132 MB)
                _, x16, _ = ops.add_layernorm_stream(x16, o16, n0.weight, n0.bias, n0.eps, want_f32=False)
                src = x16
            else:
                src, x16, _ = ops.add_layernorm_stream(src, o16, n0.weight, n0.bias, n0.eps)
            if fused_ffn:
                # FFN + residual LayerNorm as one launch: the (B, N, 1024) hidden activation stays on chip
                fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
                w1p = runtime.derived_cached('ffn_w1p', (fc1.weight,), lambda: ops.pack_linear_weight(fc1.weight))
                w2p = runtime.derived_cached('ffn_w2p', (fc2.weight,), lambda: ops.pack_linear_weight(fc2.weight))
                if last:
                    # ... whose LayerNorm also emits the query decoder's bf16 K / V operands (level-major)
                    src, m16, mp16 = ops.encoder_ffn_ln_kv(x16, w1p, fc1.bias, w2p, fc2.bias, n1.weight, n1.bias, n1.eps,
                                                           kv_tables[0], kv_tables[1], level_start)
                    return src, (m16, mp16)
                _, x16, xp16 = ops.encoder_ffn_ln(x16, w1p, fc1.bias, w2p, fc2.bias, n1.weight, n1.bias, n1.eps, pos=pos,
                                                  want_bf16=True, want_pos=True)
                src = x16
                continue
            # bias + ReLU in the GEMM epilogue (hipBLASLt) instead of a separate pass over the (B, N, 1024) hidden
            h16 = torch._addmm_activation(cc(ffn.layers[0][0].bias), x16.view(B * N, C),
                                          cc(ffn.layers[0][0].weight).t()).view(B, N, -1)
            f16 = F.linear(h16, cc(ffn.layers[1].weight), cc(ffn.layers[1].bias))
            if last and kv_tables is not None:
                # the memory's last LayerNorm also emits the query decoder's bf16 K / V operands (level-major)
                src, m16, mp16 = ops.add_layernorm_kv(src, f16, n1.weight, n1.bias, n1.eps, kv_tables[0], kv_tables[1],
                                                      level_start)
                return src, (m16, mp16)
            src, x16, xp16 = ops.add_layernorm_stream(src, f16, n1.weight, n1.bias, n1.eps, pos=pos,
                                                      want_f32=last or not self.stream_residual_bf16,
                                                      want_bf16=not last, want_pos=not last)
        return src if kv_tables is None else (src, None)

    # ---- throughput-mode inference stream: channel-last bf16 from the backbone to the packed mask feature ----
    def _pos_cached(self, level_hw, dev):
        """(N, C) sine encoding + level embedding of the all-valid pyramid; rebuilt when level_encoding changes."""
        key = (tuple(level_hw), str(dev), self.level_encoding.weight._version, self.level_encoding.weight.data_ptr())
        hit = self.__dict__.get('_pos_cache')
        if hit is None or hit[0] != key:
            pos = torch.cat([self.postional_encoding.flat_unpadded(h, w, dev) + self.level_encoding.weight[i][None]
                             for i, (h, w) in enumerate(level_hw)], 0).detach().contiguous()
            hit = (key, pos)
            self.__dict__['_pos_cache'] = hit
        return runtime.keepalive(hit[1])

    def stream_ready(self, feats):
        """True when `forward_stream` applies: throughput mode, no autograd, channel-last bf16 features (what the
        BN-folded ResNet hands over), GN-32 over 256 channels, post-norm ReLU encoder layers, one FPN level."""
        if not runtime.is_bf16() or torch.is_grad_enabled():
            return False
        if self.num_input_levels - self.num_encoder_levels != 1 or self.mask_feature.out_channels != 256:
            return False
        for f in feats:
            if not (f.is_cuda and f.dtype == torch.bfloat16 and f.dim() == 4
                    and f.permute(0, 2, 3, 1).is_contiguous()):
                return False
        for cm in list(self.input_convs) + list(self.lateral_convs) + list(self.output_convs):
            gn = getattr(cm, cm.norm_name, None) if cm.norm_name else None
            if not isinstance(gn, nn.GroupNorm) or gn.num_channels != 256 or gn.num_groups != 32:
                return False
        if not isinstance(self.output_convs[0].activate, nn.ReLU) or self.input_convs[0].activate is not None:
            return False
        return all(tuple(l.operation_order) == ('self_attn', 'norm', 'ffn', 'norm') and self._stream_ok(l)
                   for l in self.encoder.layers)

    @staticmethod
    def _gemm1x1(x2, conv):
        w = runtime.cast_cached(conv.weight).flatten(1)
        if conv.bias is not None:
            return torch.addmm(runtime.cast_cached(conv.bias), x2, w.t())
        return torch.mm(x2, w.t())

    def forward_stream(self, feats, kv_tables=None, defer_fpn=False):
        """-> (mask_feature (B, H4, W4, C) bf16 channel-last, [memories (B, hw_l, C) f32 low->high res], level sizes).
        `kv_tables(level_hw, device) -> (shift, pos)` ((N, C) f32 each): when given, a 4th value is returned, the
        per-level bf16 pairs (memory_l + shift_l, memory_l + shift_l + pos_l), each (B, hw_l, C) contiguous, written by
        the last encoder LayerNorm (`cgg_add_layernorm_kv`) instead of by 12 add / cast passes afterwards.
        1x1 convolutions are GEMMs on the (B*H*W, C) views, every GroupNorm is the channel-last HIP kernel, the three
        encoder inputs are normalised straight INTO the (B, N, C) residual stream (plus the bf16 `x`, `x + pos` copies
        the first layer's GEMMs read), the FPN `cur + up-sample(out)` is the GroupNorm's epilogue, and only the 3x3
        output convolution goes through MIOpen."""
        B = feats[0].shape[0]
        dev = feats[0].device
        C = 256
        level_hw = []
        for i in range(self.num_encoder_levels):
            f = feats[self.num_input_levels - i - 1]
            level_hw.append((int(f.shape[2]), int(f.shape[3])))
        level_start, N = [], 0
        for h, w in level_hw:
            level_start.append(N)
            N += h * w
        pos = self._pos_cached(level_hw, dev)
        ref = self._reference_points(level_hw, dev)
        # bf16 residual stream: the f32 copy of the encoder input is never read; projection kernel with the bf16 pos table:
        # neither are the `x + pos` rows
        need32 = not self.stream_residual_bf16
        needp = not (self._proj_fused(C) and POS_IN_PROJ)
        src = torch.empty((B, N, C), dtype=torch.float32, device=dev) if need32 else None
        x16 = torch.empty((B, N, C), dtype=torch.bfloat16, device=dev)
        xp16 = torch.empty((B, N, C), dtype=torch.bfloat16, device=dev) if needp else None
        ws = ops.group_norm_nhwc_workspace(B, int(feats[0].shape[2]) * int(feats[0].shape[3]), 32, dev)   # largest map
        for i in range(self.num_encoder_levels):
            f = feats[self.num_input_levels - i - 1]
            h, w = level_hw[i]
            cm = self.input_convs[i]
            y = self._gemm1x1(f.permute(0, 2, 3, 1).reshape(B * h * w, f.shape[1]), cm.conv).view(B, h * w, C)
            gn = getattr(cm, cm.norm_name)
            off = level_start[i] * C
            ops.group_norm_nhwc(y, gn.weight, gn.bias, 32, gn.eps, ws, out32=(src, off, N * C) if need32 else None,
                                out16=(x16, off, N * C), pos=(pos, off) if needp else None,
                                outp16=(xp16, off, N * C) if needp else None)
        kv16 = None
        if kv_tables is not None:
            src, kv = self._encoder_stream_bf16(src, pos, ref, level_hw, level_start, x16, xp16,
                                                kv_tables(level_hw, dev))
            if kv is not None:
                kv16 = [(kv[0][B * s0:B * (s0 + h * w)].view(B, h * w, C), kv[1][B * s0:B * (s0 + h * w)].view(B, h * w, C))
                        for s0, (h, w) in zip(level_start, level_hw)]
        else:
            src = self._encoder_stream_bf16(src, pos, ref, level_hw, level_start, x16, xp16)
        mems = [src[:, s0:s0 + h * w, :] for s0, (h, w) in zip(level_start, level_hw)]
        if defer_fpn:        # the caller runs `stream_fpn(*fpn_args)` later (pipeline stage balancing)
            fpn_args = (feats[0], src, level_hw[-1], level_start[-1], N)
            return fpn_args, mems, level_hw, kv16
        mf = self.stream_fpn(feats[0], src, level_hw[-1], level_start[-1], N)
        if kv_tables is not None:
            return mf, mems, level_hw, kv16
        return mf, mems, level_hw

    def stream_fpn(self, f, src, last_hw, last_start, N):
        """The FPN half of `forward_stream`: stride-4 feature `f` + the finest encoder level of `src` (B, N, C) f32 ->
        mask_feature (B, H4, W4, C) bf16 channel-last."""
        B, dev, C = f.shape[0], f.device, 256
        ws = ops.group_norm_nhwc_workspace(B, int(f.shape[2]) * int(f.shape[3]), 32, dev)
        # FPN: lateral 1x1 + GN on the stride-4 map, + bilinear up-sample of the finest encoder level, 3x3 + GN + ReLU
        H4, W4 = int(f.shape[2]), int(f.shape[3])
        lat, outc = self.lateral_convs[0], self.output_convs[0]
        y = self._gemm1x1(f.permute(0, 2, 3, 1).reshape(B * H4 * W4, f.shape[1]), lat.conv).view(B, H4 * W4, C)
        gn = getattr(lat, lat.norm_name)
        z = torch.empty((B, H4, W4, C), dtype=torch.bfloat16, device=dev)
        hl, wl = last_hw
        ops.group_norm_nhwc(y, gn.weight, gn.bias, 32, gn.eps, ws, up=(src, last_start * C, N * C, hl, wl), W=W4,
                            out16=(z, 0, H4 * W4 * C))
        w3 = runtime.cast_cached(outc.conv.weight)
        if not w3.is_contiguous(memory_format=torch.channels_last):
            w3 = w3.contiguous(memory_format=torch.channels_last)
            runtime.cast_cache_replace(outc.conv.weight, w3)
        y = F.conv2d(z.permute(0, 3, 1, 2), w3, None if outc.conv.bias is None else runtime.cast_cached(outc.conv.bias),
                     padding=1)
        y = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(B, H4 * W4, C)
        gn = getattr(outc, outc.norm_name)
        ops.group_norm_nhwc(y, gn.weight, gn.bias, 32, gn.eps, ws, relu=True, out16=(z, 0, H4 * W4 * C))
        return self._gemm1x1(z.view(B * H4 * W4, C), self.mask_feature).view(B, H4, W4, -1)

    # ---- parity-mode inference stream: channel-last F32 from the backbone to the mask feature, every contraction on the
    #      f32-class x3 kernels (ops.gemm_x3 / conv_x3_nhwc: f16 x 3 MFMA, f32 accumulate), norms / sampling in f32 ----
    def stream_ready_x3(self, feats):
        """True when `forward_stream_x3` applies: parity mode on the x3 kernels, no autograd, channel-last f32 features (what the
        ResNet's parity-mode path hands over), GN-32 over 256 channels, post-norm ReLU encoder layers, one FPN level."""
        if not runtime.x3_enabled() or torch.is_grad_enabled():
            return False
        if self.num_input_levels - self.num_encoder_levels != 1 or self.mask_feature.out_channels != 256:
            return False
        for f in feats:
            # channel-last views (the parity-mode ResNet's hand-over, x3a-tagged or plain) are read as they are; NCHW-contiguous
            # maps (any other backbone, tests, `smoke()`) are transposed once by `ops.nchw_to_nhwc` on the way in
            if not (f.is_cuda and f.dtype == torch.float32 and f.dim() == 4 and f.shape[1] % 32 == 0
                    and (f.permute(0, 2, 3, 1).is_contiguous() or f.is_contiguous())):
                return False
        for cm in list(self.input_convs) + list(self.lateral_convs) + list(self.output_convs):
            gn = getattr(cm, cm.norm_name, None) if cm.norm_name else None
            if not isinstance(gn, nn.GroupNorm) or gn.num_channels != 256 or gn.num_groups != 32:
                return False
        if not isinstance(self.output_convs[0].activate, nn.ReLU) or self.input_convs[0].activate is not None:
            return False
        oc = self.output_convs[0].conv
        if tuple(oc.kernel_size) != (3, 3) or tuple(oc.stride) != (1, 1) or tuple(oc.padding) != (1, 1):
            return False
        # the stream runs the input / lateral / mask-feature convolutions as row GEMMs on `weight.flatten(1)`: 1 x 1, stride 1,
        # ungrouped only (a custom config with anything else takes the module path)
        for c in [cm.conv for cm in list(self.input_convs) + list(self.lateral_convs)] + [self.mask_feature, oc]:
            if c.groups != 1 or tuple(c.dilation) != (1, 1):
                return False
            if c is not oc and (tuple(c.kernel_size) != (1, 1) or tuple(c.stride) != (1, 1) or tuple(c.padding) != (0, 0)):
                return False
        return all(tuple(l.operation_order) == ('self_attn', 'norm', 'ffn', 'norm') and self._stream_ok(l)
                   and isinstance(l.norms[0], nn.LayerNorm) and l.attentions[0].embed_dims == 256 for l in self.encoder.layers)

    def _encoder_stream_x3(self, src, pos, ref, level_hw, level_start):
        """The 6 encoder layers on the (B, N, 256) f32 stream: value / offsets+weights / output / FFN projections on the x3 GEMM
        (residuals and the ReLU in its epilogue), the f32 MSDeformAttn kernel with the softmax / location prologue inside."""
        B, N, C = src.shape
        lx = runtime.linear_x3
        x3w = lambda lin: runtime.derived_cached('x3_image', (lin.weight,), lambda: ops.pack_linear_weight_x3(lin.weight))
        srcp = None
        for layer in self.encoder.layers:
            attn = layer.attentions[0]
            H = attn.num_heads
            so, aw = attn.sampling_offsets, attn.attention_weights
            w_cat = runtime.derived_cached('msda_wcat32', (so.weight, aw.weight),
                                           lambda: torch.cat([so.weight, aw.weight], 0).float().contiguous())
            b_cat = runtime.derived_cached('msda_bcat32', (so.bias, aw.bias),
                                           lambda: torch.cat([so.bias, aw.bias], 0).float().contiguous())
            # (value_proj and the offsets | weights projection as ONE split-output GEMM with a per-token table -- `ops.gemm_x3_split`
            # -- was measured slower twice, 266 vs 280 images/s: two launches)
            value = lx(src, attn.value_proj.weight, attn.value_proj.bias).view(B, N, H, C // H)
            offs = lx(srcp if srcp is not None else src + pos[None], w_cat, b_cat)
            a = ops.msda_forward_fused(value, level_hw, level_start, offs, ref, attn.num_points)
            n0, n1 = layer.norms
            fc1, fc2 = layer.ffns[0].layers[0][0], layer.ffns[0].layers[1]
            if FUSED_TAIL and fc1.out_features % 256 == 0:
                # output_proj + LayerNorm + FFN + LayerNorm as ONE launch (x1 and the hidden activation stay on chip); it also
                # writes the next layer's `x + pos` rows
                last = layer is self.encoder.layers[-1]
                src, srcp = ops.encoder_layer_tail_x3(a, src.contiguous(), x3w(attn.output_proj), attn.output_proj.bias,
                                                      (n0.weight, n0.bias, n0.eps), x3w(fc1), fc1.bias, x3w(fc2), fc2.bias,
                                                      (n1.weight, n1.bias, n1.eps), pos=pos,
                                                      want_pos=not last)
                continue
            srcp = None
            y = lx(a, attn.output_proj.weight, attn.output_proj.bias, res=src)
            x1 = ops.add_layernorm_stream(y, None, n0.weight, n0.bias, n0.eps, want_bf16=False)[0]
            h = lx(x1, fc1.weight, fc1.bias, relu=True)
            y = lx(h, fc2.weight, fc2.bias, res=x1)
            src = ops.add_layernorm_stream(y, None, n1.weight, n1.bias, n1.eps, want_bf16=False)[0]
        return src

    def _encoder_stream_x3a(self, src, srcp, pos, ref, level_hw, level_start):
        """Round 4 form of `_encoder_stream_x3`: the residual stream `src` and `srcp = src + pos` are x3a rows (csrc/x3.h). Per
        layer: value / offsets+weights projections on the LDS-DMA GEMM (A = the rows as stored, f32 out for the gather), the f32
        MSDeformAttn kernel, `encoder_layer_tail_x3` reading / writing x3a rows."""
        B, N, C = src.shape
        x3w = lambda lin: runtime.derived_cached('x3_image', (lin.weight,), lambda: ops.pack_linear_weight_x3(lin.weight))
        for layer in self.encoder.layers:
            attn = layer.attentions[0]
            H = attn.num_heads
            so, aw, vp = attn.sampling_offsets, attn.attention_weights, attn.value_proj
            n0, n1 = layer.norms
            fc1, fc2 = layer.ffns[0].layers[0][0], layer.ffns[0].layers[1]
            last = layer is self.encoder.layers[-1]
            n_cat = so.weight.shape[0] + aw.weight.shape[0]
            if self._merged_proj_ok(layer):
                # ONE projection GEMM per layer (round 6): rows [value | offsets | logits] = src [Wv; Woff; Watt]^T + [bv; boff; batt]
                # + T with the batch-independent table T = [0 | pos [Woff; Watt]^T] in the GEMM's residual input ((src + pos) W =
                # src W + pos W: `srcp` rows are neither written by the layer tail nor read here); the sampling kernel takes its
                # value operand as the first 256 columns of those rows.
                w_all = runtime.derived_cached('msda_wall32', (vp.weight, so.weight, aw.weight),
                                               lambda: torch.cat([vp.weight, so.weight, aw.weight], 0).float().contiguous())
                b_all = runtime.derived_cached('msda_ball32', (vp.bias, so.bias, aw.bias),
                                               lambda: torch.cat([vp.bias, so.bias, aw.bias], 0).float().contiguous())

                tab = self._proj_pos_table(layer, pos, so, aw, tuple(level_hw), N, C, n_cat)
                rows_all = runtime.linear_x3s(src.view(B * N, C), w_all, b_all, res=tab, res_mod=N).view(B, N, C + n_cat)
                a = ops.msda_forward_fused_rows(rows_all, level_hw, level_start, ref, attn.num_points, H, C // H)
                src, srcp = ops.encoder_layer_tail_x3(a, src, x3w(attn.output_proj), attn.output_proj.bias,
                                                      (n0.weight, n0.bias, n0.eps), x3w(fc1), fc1.bias, x3w(fc2), fc2.bias,
                                                      (n1.weight, n1.bias, n1.eps), pos=pos, want_pos=False, x3a=True)
                continue
            w_cat = runtime.derived_cached('msda_wcat32', (so.weight, aw.weight),
                                           lambda: torch.cat([so.weight, aw.weight], 0).float().contiguous())
            b_cat = runtime.derived_cached('msda_bcat32', (so.bias, aw.bias),
                                           lambda: torch.cat([so.bias, aw.bias], 0).float().contiguous())
            if srcp is None:             # (a merged layer in front of this one did not produce src + pos)
                raise ops.CggError('encoder stream: mixed merged / split projection layers are not supported')
            value = runtime.linear_x3s(src.view(B * N, C), attn.value_proj.weight, attn.value_proj.bias).view(B, N, H, C // H)
            offs = runtime.linear_x3s(srcp.view(B * N, C), w_cat, b_cat).view(B, N, -1)
            a = ops.msda_forward_fused(value, level_hw, level_start, offs, ref, attn.num_points)
            src, srcp = ops.encoder_layer_tail_x3(a, src, x3w(attn.output_proj), attn.output_proj.bias,
                                                  (n0.weight, n0.bias, n0.eps), x3w(fc1), fc1.bias, x3w(fc2), fc2.bias,
                                                  (n1.weight, n1.bias, n1.eps), pos=pos, want_pos=not last, x3a=True)
        return src

    def _proj_pos_table(self, layer, pos, so, aw, level_hw, N, C, n_cat):
        """T = [0 | pos [Woff; Watt]^T] (N, C + n_cat) f32 of a layer's merged projection, cached per (layer, pyramid) in a SMALL
        LRU on this module (46.8 MB per layer at 1024^2: a cache keyed on the `pos` tensor object would keep one table per image
        shape ever seen -- `_pos_cached` rebuilds `pos` whenever the shape changes) and rebuilt when the weights or the level
        encoding change."""
        import collections
        cache = self.__dict__.setdefault('_proj_tables', collections.OrderedDict())
        key = (id(layer), level_hw, str(pos.device))
        ver = (so.weight._version, aw.weight._version, so.weight.data_ptr(), aw.weight.data_ptr(), self.level_encoding.weight._version)
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            cache.move_to_end(key)
            return runtime.keepalive(hit[1])
        with torch.no_grad():
            w_cat = torch.cat([so.weight, aw.weight], 0).float().contiguous()
            t = torch.zeros((N, C + n_cat), dtype=torch.float32, device=pos.device)
            t[:, C:] = runtime.linear_x3(pos.float().contiguous(), w_cat)
        cache[key] = (ver, t)
        while len(cache) > 2 * len(self.encoder.layers):        # two pyramids' worth of tables (a captured pipeline of an older
            cache.popitem(last=False)                           # pyramid keeps ITS tables alive itself: runtime.keepalive_scope)
        return runtime.keepalive(t)

    @staticmethod
    def _merged_proj_ok(layer):
        """the layer's value / offsets / logits projections as one GEMM + `cgg_msda_forward_fused_vld` (H = 8, D = 32, L = 3, P = 4)"""
        attn = layer.attentions[0]
        so, aw, vp = attn.sampling_offsets, attn.attention_weights, attn.value_proj
        C, H = vp.weight.shape[1], attn.num_heads
        n_cat = so.weight.shape[0] + aw.weight.shape[0]
        return (MERGED_PROJ and H == 8 and C == 256 and vp.weight.shape[0] == 256 and attn.num_levels == 3 and attn.num_points == 4
                and n_cat == 3 * H * attn.num_levels * attn.num_points and (C + n_cat) % 32 == 0
                and all(m.bias is not None for m in (vp, so, aw)))

    def _forward_stream_x3a(self, feats, defer_fpn=False):
        """`forward_stream_x3` on x3a rows: the backbone maps arrive as x3a (`ops.X3ATensor`; plain f32 maps are encoded), every
        GEMM-consumed tensor of the stream stays x3a -- GroupNorm / the encoder tail write it, the LDS-DMA GEMMs read it -- and
        only the tensors a non-GEMM kernel gathers from (MSDeformAttn's value / offsets, the mask feature) are f32. The memories
        come back as x3a-tagged (B, hw_l, C) views."""
        B = feats[0].shape[0]
        dev = feats[0].device
        C = 256
        rows = lambda f: (f if ops.is_x3a(f) else ops.x3a_encode(ops.nchw_to_nhwc(f).contiguous()).permute(0, 3, 1, 2)) \
            .as_subclass(torch.Tensor).permute(0, 2, 3, 1).reshape(-1, f.shape[1])
        level_hw = []
        for i in range(self.num_encoder_levels):
            f = feats[self.num_input_levels - i - 1]
            level_hw.append((int(f.shape[2]), int(f.shape[3])))
        level_start, N = [], 0
        for h, w in level_hw:
            level_start.append(N)
            N += h * w
        pos = self._pos_cached(level_hw, dev)
        ref = self._reference_points(level_hw, dev)
        src = torch.empty((B, N, C), dtype=torch.float32, device=dev)      # x3a rows
        # src + pos, x3a rows (the first layer's offsets input) -- not needed when every layer takes the merged projection
        need_srcp = not all(self._merged_proj_ok(l) for l in self.encoder.layers)
        srcp = torch.empty((B, N, C), dtype=torch.float32, device=dev) if need_srcp else None
        ws = ops.group_norm_nhwc_workspace(B, int(feats[0].shape[2]) * int(feats[0].shape[3]), 32, dev)   # largest map
        for i in range(self.num_encoder_levels):
            f = feats[self.num_input_levels - i - 1]
            h, w = level_hw[i]
            cm = self.input_convs[i]
            y = runtime.linear_x3s(rows(f), cm.conv.weight.flatten(1), cm.conv.bias)
            gn = getattr(cm, cm.norm_name)
            ops.group_norm_nhwc_x3a(y.view(B, h * w, C), gn.weight, gn.bias, 32, gn.eps, ws, out=(src, level_start[i] * C, N * C),
                                    pos=(pos, level_start[i] * C) if need_srcp else None,
                                    outp=(srcp, level_start[i] * C) if need_srcp else None)
        src = self._encoder_stream_x3a(src, srcp, pos, ref, level_hw, level_start)
        mems = [ops.as_x3a(src[:, s0:s0 + h * w, :]) for s0, (h, w) in zip(level_start, level_hw)]
        fpn = (rows(feats[0]), int(feats[0].shape[2]), int(feats[0].shape[3]), src, level_start[-1], N, level_hw[-1], ws)
        if defer_fpn:        # pipeline balancing: the FPN (0.75 ms of throughput kernels at configs[1]) runs in the next stage
            return fpn, mems, level_hw
        return self.stream_fpn_x3a(*fpn), mems, level_hw

    def stream_fpn_x3a(self, frows, H4, W4, src, fine_start, N, fine_hw, ws):
        """FPN of the x3a stream: lateral 1x1 + GN on the stride-4 map, + bilinear up-sample of the finest encoder level, 3x3 + GN +
        ReLU, mask_feature -> (B, H4, W4, C) f32."""
        C = 256
        B = src.shape[0]
        if True:
            lat, outc = self.lateral_convs[0], self.output_convs[0]
            hl, wl = fine_hw
            level_start = [fine_start]
            y = runtime.linear_x3s(frows, lat.conv.weight.flatten(1), lat.conv.bias).view(B, H4 * W4, C)
            gn = getattr(lat, lat.norm_name)
            ops.group_norm_nhwc_x3a(y, gn.weight, gn.bias, 32, gn.eps, ws, out=(y, 0, H4 * W4 * C),
                                    up=(src, level_start[-1] * C, N * C, hl, wl), W=W4)                # y: f32 -> x3a in place
            w3 = runtime.derived_cached('x3_conv_image', (outc.conv.weight,), lambda: ops.pack_conv_weight_x3(outc.conv.weight))
            z = ops.conv_x3s_nhwc(y.view(B, H4, W4, C), w3, C, 3, 1, 1, outc.conv.bias, out_split=False).view(B, H4 * W4, C)
            gn = getattr(outc, outc.norm_name)
            ops.group_norm_nhwc_x3a(z, gn.weight, gn.bias, 32, gn.eps, ws, out=(z, 0, H4 * W4 * C), relu=True)
            mf = runtime.linear_x3s(z.view(B * H4 * W4, C), self.mask_feature.weight.flatten(1), self.mask_feature.bias)
            return mf.view(B, H4, W4, -1)

    def forward_stream_x3(self, feats, defer_fpn=False):
        """-> (mask_feature (B, H4, W4, C) f32 channel-last, [memories (B, hw_l, C) f32 low->high res], level sizes). 1x1
        convolutions are x3 GEMMs on the (B*H*W, C) views, the 3x3 output convolution the x3 implicit GEMM, every GroupNorm the
        channel-last kernel on f32 input (the three encoder inputs normalised straight into the (B, N, C) stream, the FPN's
        `cur + up-sample(out)` in the GroupNorm's epilogue)."""
        if runtime.x3a_enabled():
            return self._forward_stream_x3a(feats, defer_fpn=defer_fpn)
        feats = [ops.x3a_to_f32(f) if ops.is_x3a(f) else f for f in feats]
        B = feats[0].shape[0]
        dev = feats[0].device
        C = 256
        level_hw = []
        for i in range(self.num_encoder_levels):
            f = feats[self.num_input_levels - i - 1]
            level_hw.append((int(f.shape[2]), int(f.shape[3])))
        level_start, N = [], 0
        for h, w in level_hw:
            level_start.append(N)
            N += h * w
        pos = self._pos_cached(level_hw, dev)
        ref = self._reference_points(level_hw, dev)
        src = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        ws = ops.group_norm_nhwc_workspace(B, int(feats[0].shape[2]) * int(feats[0].shape[3]), 32, dev)   # largest map
        for i in range(self.num_encoder_levels):
            f = feats[self.num_input_levels - i - 1]
            h, w = level_hw[i]
            cm = self.input_convs[i]
            y = runtime.linear_x3(ops.nchw_to_nhwc(f).reshape(B * h * w, f.shape[1]), cm.conv.weight.flatten(1), cm.conv.bias)
            gn = getattr(cm, cm.norm_name)
            ops.group_norm_nhwc(y.view(B, h * w, C), gn.weight, gn.bias, 32, gn.eps, ws, out32=(src, level_start[i] * C, N * C))
        src = self._encoder_stream_x3(src, pos, ref, level_hw, level_start)
        mems = [src[:, s0:s0 + h * w, :] for s0, (h, w) in zip(level_start, level_hw)]
        # FPN: lateral 1x1 + GN on the stride-4 map, + bilinear up-sample of the finest encoder level, 3x3 + GN + ReLU, mask_feature
        f = feats[0]
        H4, W4 = int(f.shape[2]), int(f.shape[3])
        lat, outc = self.lateral_convs[0], self.output_convs[0]
        y = runtime.linear_x3(ops.nchw_to_nhwc(f).reshape(B * H4 * W4, f.shape[1]), lat.conv.weight.flatten(1), lat.conv.bias)
        gn = getattr(lat, lat.norm_name)
        hl, wl = level_hw[-1]
        y = y.view(B, H4 * W4, C)
        ops.group_norm_nhwc(y, gn.weight, gn.bias, 32, gn.eps, ws, up=(src, level_start[-1] * C, N * C, hl, wl), W=W4,
                            out32=(y, 0, H4 * W4 * C))
        w3 = runtime.derived_cached('x3_conv_image', (outc.conv.weight,), lambda: ops.pack_conv_weight_x3(outc.conv.weight))
        z = ops.conv_x3_nhwc(y.view(B, H4, W4, C), w3, C, 3, 1, 1, outc.conv.bias).view(B, H4 * W4, C)
        gn = getattr(outc, outc.norm_name)
        ops.group_norm_nhwc(z, gn.weight, gn.bias, 32, gn.eps, ws, relu=True, out32=(z, 0, H4 * W4 * C))
        mf = runtime.linear_x3(z.view(B * H4 * W4, C), self.mask_feature.weight.flatten(1), self.mask_feature.bias)
        return mf.view(B, H4, W4, -1), mems, level_hw

    def forward(self, feats):
        feats = [ops.x3a_to_f32(f) if ops.is_x3a(f) else f for f in feats]           # x3a backbone maps (parity-mode ResNet)
        B = feats[0].shape[0]
        dev = feats[0].device
        srcs, poss, level_hw = [], [], []
        for i in range(self.num_encoder_levels):
            feat = feats[self.num_input_levels - i - 1]
            h, w = feat.shape[-2:]
            level_hw.append((int(h), int(w)))
            rows = runtime.input_level_x3_train(self.input_convs[i], feat)      # parity-mode training: channel-last rows end to end
            if rows is not None:
                srcs.append(rows)
            else:
                with runtime.autocast():
                    proj = self.input_convs[i](feat)
                srcs.append(proj.float().flatten(2).transpose(1, 2))                 # (B, hw, C)
            poss.append(self.postional_encoding.flat_unpadded(int(h), int(w), dev)
                        + self.level_encoding.weight[i][None])                      # (hw, C)
        level_start, s = [], 0
        for h, w in level_hw:
            level_start.append(s)
            s += h * w
        src = torch.cat(srcs, 1).contiguous()      # (B, N, C)
        pos = torch.cat(poss, 0)                   # (N, C)
        ref = self._reference_points(level_hw, dev)
        for layer in self.encoder.layers:
            assert tuple(layer.operation_order) == ('self_attn', 'norm', 'ffn', 'norm'), \
                'MSDeformAttnPixelDecoder fast path expects post-norm (self_attn, norm, ffn, norm) layers'
        if runtime.is_bf16() and not torch.is_grad_enabled() and src.shape[-1] == 256 and \
                all(self._stream_ok(l) for l in self.encoder.layers):
            src = self._encoder_stream_bf16(src, pos, ref, level_hw, level_start)
        else:
            fused_ln = (FUSED_TRAIN_LN and runtime.is_bf16() and torch.is_grad_enabled() and src.is_cuda and src.shape[-1] == 256
                        and all(l.attentions[0].dropout.p == 0 and l.ffns[0].plain_relu_ffn()
                                and isinstance(l.norms[0], nn.LayerNorm) and isinstance(l.norms[1], nn.LayerNorm)
                                for l in self.encoder.layers))
            if fused_ln:
                # training: the branch outputs stay bf16, the residual add happens INSIDE one-pass LayerNorm kernels (forward and
                # backward) instead of cast + add + layer_norm (+ their three backward kernels) per norm, and the same kernels
                # emit the bf16 copies (x, x + pos) the next GEMMs read -- whose bf16 gradients they sum again on the way back
                pos_c = pos.contiguous()                      # differentiable: the level embedding is part of it
                src16, srcp16 = src.to(torch.bfloat16), (src + pos[None]).to(torch.bfloat16)
                n_layers = len(self.encoder.layers)
                for li, layer in enumerate(self.encoder.layers):
                    attn, ffn = layer.attentions[0], layer.ffns[0]
                    out16 = attn.forward_fused(src, None, ref, level_hw, level_start, add_identity=False, src16=src16,
                                               src_pos16=srcp16)
                    mid, mid16, _ = ops.add_layernorm_train(src, out16, layer.norms[0], want_bf16=True)
                    last = li == n_layers - 1
                    nxt = ops.add_layernorm_train(mid, ffn.forward_bf16_noidentity(mid16), layer.norms[1], pos=pos_c,
                                                  want_bf16=not last, want_pos=not last)
                    src, src16, srcp16 = nxt if not last else (nxt, None, None)
            # parity-mode training: the same one-pass residual + LayerNorm kernels (forward and backward) on f32 branch outputs
            fused_ln32 = (not fused_ln and FUSED_TRAIN_LN and not runtime.is_bf16() and torch.is_grad_enabled() and src.is_cuda
                          and src.dtype == torch.float32 and src.shape[-1] == 256
                          and all(l.attentions[0].dropout.p == 0 and l.ffns[0].plain_relu_ffn()
                                  and isinstance(l.norms[0], nn.LayerNorm) and isinstance(l.norms[1], nn.LayerNorm)
                                  and ops.add_layernorm_train_ok(src, src, l.norms[0]) and ops.add_layernorm_train_ok(src, src, l.norms[1])
                                  for l in self.encoder.layers))
            for layer in ([] if fused_ln else self.encoder.layers):
                attn = layer.attentions[0]
                if fused_ln32 and FUSED_TRAIN_MSDA and runtime.x3_layer_nodes_ok(layer, src):
                    # one autograd node per half layer: residuals and gradient fan-in inside GEMM epilogues, LayerNorm of one tensor
                    src = runtime.encoder_layer_x3_train(layer, src, pos, ref, level_hw, level_start)
                    continue
                if fused_ln32:
                    out = attn.forward_fused(src, src + pos[None], ref, level_hw, level_start, add_identity=False)
                    mid = ops.add_layernorm_train(src, out.float(), layer.norms[0])
                    src = ops.add_layernorm_train(mid, layer.ffns[0].forward_noidentity(mid).float(), layer.norms[1])
                    continue
                src = attn.forward_fused(src, src + pos[None], ref, level_hw, level_start)
                src = layer.norms[0](src)
                src = layer.ffns[0](src)
                src = layer.norms[1](src)
        level_rows = src.split([h * w for h, w in level_hw], dim=1)
        outs = [x.transpose(1, 2).reshape(B, -1, h, w) for x, (h, w) in zip(level_rows, level_hw)]
        if (self.num_input_levels - self.num_encoder_levels == 1 and len(outs) >= self.num_outs
                and runtime.x3_fpn_level_ok(self, feats[0], level_hw[-1])):
            # parity-mode training: the FPN level stays channel-last on own kernels (lateral / mask-feature 1 x 1 as x3 row GEMMs, both
            # GroupNorms, the up-sample + add, the ReLU and the x3 3 x 3 convolution in one autograd node: `runtime._X3FpnLevelFn`)
            return runtime.fpn_level_x3_train(self, feats[0], level_rows[-1], level_hw[-1]), outs[:self.num_outs]
        for i in range(self.num_input_levels - self.num_encoder_levels - 1, -1, -1):
            x = feats[i]
            with runtime.autocast():
                cur = self.lateral_convs[i](x)
            y = cur + F.interpolate(outs[-1], size=cur.shape[-2:], mode='bilinear', align_corners=False)
            with runtime.autocast():
                y = self.output_convs[i](y)
            # an FPN level that is not among the returned outputs only feeds the next convolution: it stays in the autocast
            # dtype (bf16 -> f32 -> bf16 is exact, but two 1-GB passes forward and two backward at configs[2])
            outs.append(y.float() if len(outs) < self.num_outs or i > 0 else y)
        with runtime.autocast():
            mask_feature = self.mask_feature(outs[-1].contiguous())
        return mask_feature.float(), outs[:self.num_outs]
