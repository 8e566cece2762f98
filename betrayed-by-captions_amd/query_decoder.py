"""Transformer query decoder blocks named by the reference config
(configs/instance/coco_b48n17.py:72-99): `DetrTransformerDecoder`, `DetrTransformerDecoderLayer`,
`MultiheadAttention` -- [3P] mmcv/mmdet classes, restated with upstream parameter names
(`attentions.N.attn.in_proj_weight`, ...) and executed on the HIP masked
cross-attention kernel. The decoder loop itself lives in Mask2FormerHeadOpen.forward
(open_set/models/mask2former_head.py:822-847).

MI355X-first execution (used by the head's fast path):
  * batch-first (B, Q, C) / (B, S, C) activations;
  * K/V projection: ONE GEMM `mem @ [Wk; Wv]^T` + a batch-independent (S, 2C) bias matrix
    (pos @ Wk^T + bk | bv) -- algebraically (mem + pos) Wk^T + bk, without materialising mem + pos;
  * attention core: `cgg_masked_xattn_forward` with the bit-packed, head-shared mask.
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, runtime
from .pixel_decoder import BaseTransformerLayer, TransformerLayerSequence, build_norm
from .registry import ATTENTION, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def small_linear(x, weight, bias=None, relu=False, res=None, out=None):
    """Query-side linear (M = B*Q rows): the HIP skinny-GEMM kernel at inference; torch (autograd) in training."""
    if _needs_grad(x, weight, bias, res):
        y = F.linear(x, weight, bias)
        if relu:
            y = F.relu(y)
        if res is not None:
            y = y + res
        if out is not None:
            out.copy_(y)
            return out
        return y
    return ops.linear_rows(x.float(), weight, bias, relu=relu, res=res, split=not runtime.is_bf16(), out=out)


def residual_layernorm(x, res, norm):
    """LayerNorm(x (+ res)) with nn.LayerNorm parameters."""
    if _needs_grad(x, res, norm.weight):
        return norm(x if res is None else x + res)
    return ops.add_layernorm(x, res, norm.weight, norm.bias, norm.eps)


def pack_bool_mask(mask):
    """bool (..., S) -> int32 bits (..., ceil(S/32)); bit i of word w = mask[32*w + i]. (torch ops)"""
    S = mask.shape[-1]
    words = (S + 31) // 32
    pad = words * 32 - S
    m = F.pad(mask, (0, pad)) if pad else mask
    m = m.reshape(*mask.shape[:-1], words, 32).to(torch.int64)
    w = (m << torch.arange(32, device=mask.device, dtype=torch.int64)).sum(-1)
    return torch.where(w >= 2**31, w - 2**32, w).to(torch.int32)


@ATTENTION.register_module()
class MultiheadAttention(nn.Module):
    """[3P] mmcv MultiheadAttention wrapper (`identity + dropout(proj_drop(attn(q+pos, k+pos, v)))`).
    `.attn` is an nn.MultiheadAttention used as the PARAMETER CONTAINER (checkpoint key layout);
    the arithmetic runs on the HIP kernel."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__()
        if 'dropout' in kwargs:
            warnings.warn('The arguments `dropout` in MultiheadAttention has been deprecated, now you can '
                          'separately set `attn_drop`(float), proj_drop(float), and `dropout_layer`(dict) ',
                          DeprecationWarning)
            attn_drop = kwargs['dropout']
            dropout_layer = dict(dropout_layer or {}, drop_prob=kwargs.pop('dropout'))
        if attn_drop != 0.:
            raise NotImplementedError('attention dropout is not implemented on the HIP attention kernel '
                                      '(every shipped CGG config uses attn_drop=0.0)')
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(dropout_layer.get('drop_prob', 0.)) if dropout_layer else nn.Identity()

    # ---- fast-path pieces (batch-first) ----
    def kv_weight(self):
        E = self.embed_dims
        return self.attn.in_proj_weight[E:], self.attn.in_proj_bias[E:]

    def project_kv(self, mem, key_pos):
        """mem (B,S,C) = value input, key input = mem + key_pos (S,C) -> kv (B,S,2C) f32 [K|V]."""
        E = self.embed_dims
        w_kv, b_kv = self.kv_weight()
        if key_pos is not None:
            def make_bias():
                kbias = F.linear(key_pos, w_kv[:E])                   # (S,C): pos @ Wk^T
                return torch.cat([kbias + b_kv[:E], b_kv[E:].expand(kbias.shape[0], E)], 1).contiguous()  # (S,2C)
            if _needs_grad(key_pos, w_kv, b_kv):
                bias = make_bias()
            else:   # inference: depends on the weights and the level's (cached) encoding alone
                bias = runtime.derived_cached('kv_pos_bias', (key_pos, self.attn.in_proj_weight, self.attn.in_proj_bias), make_bias)
        else:
            bias = b_kv
        if bias.dim() == 2 and runtime.x3_linear_ok(mem, w_kv) and mem.dim() == 3:
            # parity mode: the batch-independent (S, 2C) bias rides in the x3 GEMM's residual input, one image at a time
            kv = torch.empty((mem.shape[0], mem.shape[1], 2 * E), dtype=torch.float32, device=mem.device)
            for b in range(mem.shape[0]):
                runtime.linear_x3(mem[b], w_kv, None, res=bias, out=kv[b])
            return kv
        if (bias.dim() == 2 and mem.dim() == 3 and runtime.x3_train_linear_ok(mem, w_kv) and not runtime.is_bf16()
                and bias.shape == (mem.shape[1], w_kv.shape[0])):
            # parity-mode training: the same table in the x3 node's residual input (no broadcast add over the (B, S, 2C) result)
            return runtime._X3LinearTableFn.apply(mem, w_kv, bias)
        kv = runtime.linear(mem, w_kv)
        return (kv + bias).contiguous()

    def project_kv_bf16(self, mem16, mempos16):
        """Throughput-mode K / V for `cgg_masked_xattn_forward_bf16`: k = (mem + pos) Wk^T + b_k as (B,S,E) bf16 and
        the value projection TRANSPOSED, vt = Wv mem^T (B,E,S) bf16, with b_v left to the output projection."""
        E = self.embed_dims
        w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        cc = runtime.cast_cached
        k = F.linear(mempos16, cc(w[E:2 * E]), cc(b[E:2 * E]))
        vt = torch.matmul(cc(w[2 * E:]), mem16.transpose(1, 2))
        return k, vt

    def attend(self, query, query_pos, kv, bits):
        """query (B,Q,C) (+ query_pos) against projected kv; returns identity + out_proj(core)."""
        E = self.embed_dims
        q_in = query if query_pos is None else query + query_pos
        q = small_linear(q_in, self.attn.in_proj_weight[:E], self.attn.in_proj_bias[:E])
        core = _xattn(q.contiguous(), kv, bits, self.num_heads)
        out = small_linear(core, self.attn.out_proj.weight, self.attn.out_proj.bias, res=query)
        return out

    def self_attend(self, query, query_pos):
        E = self.embed_dims
        qk_in = query if query_pos is None else query + query_pos
        w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        B, Q, _ = query.shape
        q = small_linear(qk_in, w[:E], b[:E])
        if _needs_grad(query, w):
            kv = torch.cat([small_linear(qk_in, w[E:2 * E], b[E:2 * E]), small_linear(query, w[2 * E:], b[2 * E:])], -1)
        else:   # the two projections write the halves of one [K | V] buffer directly
            kv = torch.empty((B, Q, 2 * E), dtype=torch.float32, device=query.device)
            small_linear(qk_in, w[E:2 * E], b[E:2 * E], out=kv.view(B * Q, 2 * E)[:, :E])
            small_linear(query, w[2 * E:], b[2 * E:], out=kv.view(B * Q, 2 * E)[:, E:])
        core = _xattn(q.contiguous(), kv, None, self.num_heads)
        return small_linear(core, self.attn.out_proj.weight, self.attn.out_proj.bias, res=query)

    # ---- [3P] signature ----
    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None:
            if query_pos.shape == key.shape:
                key_pos = query_pos
            else:
                warnings.warn(f'position encoding of key is missing in {self.__class__.__name__}.')
        if key_padding_mask is not None:
            raise NotImplementedError('key_padding_mask is not supported (CGG passes None: '
                                      'open_set/models/mask2former_head.py:838-840)')
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if not self.batch_first:  # (S,B,C) -> (B,S,C)
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        E = self.embed_dims
        w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        q = F.linear(query, w[:E], b[:E])
        kv = torch.cat([F.linear(key, w[E:2 * E], b[E:2 * E]), F.linear(value, w[2 * E:], b[2 * E:])], -1)
        bits = None
        if attn_mask is not None:
            B, Q = q.shape[0], q.shape[1]
            if attn_mask.dtype == torch.int32:
                bits = attn_mask
            else:
                if attn_mask.dtype != torch.bool:
                    raise NotImplementedError('only boolean attention masks (True = blocked)')
                if attn_mask.dim() == 3:  # (B*H,Q,S): CGG repeats one mask over heads -> take head 0
                    attn_mask = attn_mask.view(B, -1, Q, attn_mask.shape[-1])[:, 0]
                else:
                    attn_mask = attn_mask[None].expand(B, -1, -1)
                bits = pack_bool_mask(attn_mask).contiguous()
        core = _xattn(q.contiguous(), kv.contiguous(), bits, self.num_heads)
        out = F.linear(core, self.attn.out_proj.weight, self.attn.out_proj.bias)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class _XAttnFn(torch.autograd.Function):
    """HIP forward (saves the log-sum-exp rows) and HIP backward (`cgg_masked_xattn_backward`: probabilities recomputed
    tile by tile from the bit mask; the reference's (B*8, Q, S) score / probability / gradient tensors never exist)."""

    @staticmethod
    def forward(ctx, q, kv, bits, num_heads):
        out, lse = ops.masked_xattn(q, kv, bits, num_heads, return_lse=True)
        ctx.save_for_backward(q, kv, bits if bits is not None else torch.empty(0, device=q.device), out, lse)
        ctx.num_heads = num_heads
        ctx.has_bits = bits is not None
        return out

    @staticmethod
    def backward(ctx, go):
        q, kv, bits, out, lse = ctx.saved_tensors
        gq, gkv = ops.masked_xattn_backward(q, kv, bits if ctx.has_bits else None, out, lse, go, ctx.num_heads)
        return gq, gkv, None, None


def xattn_backward_torch(q, kv, bits, go, num_heads):
    """The same gradients with torch ops (materialises the (B, H, Q, S) scores): the formulation the HIP kernel is
    tested against, not used by the product path."""
    H = num_heads
    B, Q, E = q.shape
    S = kv.shape[1]
    D = E // H
    with torch.enable_grad():
        q_ = q.detach().requires_grad_(True)
        kv_ = kv.detach().requires_grad_(True)
        qh = (q_ * D**-0.5).view(B, Q, H, D).transpose(1, 2)
        kh = kv_[..., :E].view(B, S, H, D).transpose(1, 2)
        vh = kv_[..., E:].view(B, S, H, D).transpose(1, 2)
        att = qh @ kh.transpose(-1, -2)
        if bits is not None:
            att = att.masked_fill(ops.unpack_bits(bits, S)[:, None], float('-inf'))
        out = (att.softmax(-1) @ vh).transpose(1, 2).reshape(B, Q, E)
        return torch.autograd.grad(out, (q_, kv_), go)


def _xattn(q, kv, bits, num_heads):
    if torch.is_grad_enabled() and (q.requires_grad or kv.requires_grad):
        return _XAttnFn.apply(q, kv, bits, num_heads)
    return ops.masked_xattn(q, kv, bits, num_heads)


@TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """[3P] mmdet DetrTransformerDecoderLayer."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == set(['self_attn', 'norm', 'cross_attn', 'ffn'])

    def stream_ready(self):
        ffn = self.ffns[0]
        return (tuple(self.operation_order) == ('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm')
                and len(ffn.layers) == 3 and isinstance(ffn.layers[0][1], nn.ReLU) and ffn.add_identity
                and self.embed_dims <= 256 and self.embed_dims % 32 == 0)

    def forward_stream(self, x, xp, pos, kv, bits, post_norm=None, q=None, raw=False, fix_rows=False):
        """Throughput-mode layer on 2-D rows: x, xp = x + pos (M = B*Q, C) f32; pos (Q, C). Every projection is
        `cgg_linear_rows_bf16` on pre-packed bf16 weights; the three post-norm LayerNorms, the residual adds and the
        `+ query_pos` adds run in GEMM epilogues / one LayerNorm-chain pass. Returns (x', x' + pos, post_norm(x')).
        `q` = the cross-attention query projection if the previous layer's tail kernel already made it (xp is then
        unused); `raw` returns the FFN output (+ residual) BEFORE the last norm -- split-K planes (n, M, C) -- for
        `ops.decoder_tail` to finish."""
        ca, sa = self.attentions
        E = self.embed_dims
        M = x.shape[0]
        B = (kv[0] if isinstance(kv, tuple) else kv).shape[0]
        Q = M // B
        H = ca.num_heads
        pk = runtime.packed_cached
        lr = ops.linear_rows_bf16
        n0, n1, n2 = self.norms
        w, b = ca.attn.in_proj_weight, ca.attn.in_proj_bias
        if q is None:
            wq, bq, _ = pk((w[:E],), (b[:E],))
            q = lr(xp, wq, E, bq)
        wo, bo, _ = pk((ca.attn.out_proj.weight,), (ca.attn.out_proj.bias,))
        if isinstance(kv, tuple):
            # bf16 K and transposed V (see project_kv_bf16): the value bias is folded through the softmax
            # (rows sum to 1) into the output projection's bias, bo' = bo + Wo b_v
            core = ops.masked_xattn_bf16(q.view(B, Q, E), kv[0], kv[1], bits, H, fix_full_rows=fix_rows).view(M, E)
            ow, ob = ca.attn.out_proj.weight, ca.attn.out_proj.bias
            bo = runtime.derived_cached('xattn_bo', (ow, ob, b), lambda: (ob + ow @ b[2 * E:]).float().contiguous())
        else:
            if fix_rows:
                ops.attn_mask_fix_full_rows(bits, kv.shape[1])
            core = ops.masked_xattn(q.view(B, Q, E), kv, bits, H).view(M, E)
        w, b = sa.attn.in_proj_weight, sa.attn.in_proj_bias
        fused_mid = E == 256 and core.stride(1) == 1
        if fused_mid:
            # out-proj + residual + LayerNorm + the self-attention's q | k | v projection: one launch
            wqkv, bqkv, _ = pk((w,), (b,))
            x1, q2, kv2 = ops.decoder_mid(core, wo, bo, x, (n0.weight, n0.bias, n0.eps), pos, (wqkv, bqkv))
        else:
            x1, x1p = lr(core, wo, E, bo, res=x, ln=(n0.weight, n0.bias, n0.eps), pos=pos, want_pos=True)
        if fused_mid:
            pass
        elif E % 256 == 0:
            # in_proj_weight is already [Wq; Wk; Wv]: one launch, q | k read x1 + pos, v reads x1
            wqkv, bqkv, _ = pk((w,), (b,))
            q2, kv2 = ops.linear_rows_bf16_qkv(x1p, x1, wqkv, bqkv, E)
        else:
            wq, bq, _ = pk((w[:E],), (b[:E],))
            wk, bk, _ = pk((w[E:2 * E],), (b[E:2 * E],))
            wv, bv, _ = pk((w[2 * E:],), (b[2 * E:],))
            q2 = lr(x1p, wq, E, bq)
            kv2 = torch.empty((M, 2 * E), dtype=torch.float32, device=x.device)
            lr(x1p, wk, E, bk, out=kv2[:, :E])
            lr(x1, wv, E, bv, out=kv2[:, E:])
        if Q <= 128 and E // sa.num_heads == 32 and runtime.is_bf16():
            core2 = ops.self_attn_rows_bf16(q2, kv2, B, sa.num_heads)
        else:
            core2 = ops.masked_xattn(q2.view(B, Q, E), kv2.view(B, Q, 2 * E), None, sa.num_heads).view(M, E)
        wo, bo, _ = pk((sa.attn.out_proj.weight,), (sa.attn.out_proj.bias,))
        if fused_mid:
            x2 = ops.decoder_mid(core2, wo, bo, x1, (n1.weight, n1.bias, n1.eps))[0]
        else:
            x2 = lr(core2, wo, E, bo, res=x1, ln=(n1.weight, n1.bias, n1.eps))
        ffn = self.ffns[0]
        w1, b1, F1 = pk((ffn.layers[0][0].weight,), (ffn.layers[0][0].bias,))
        w2, b2, _ = pk((ffn.layers[1].weight,), (ffn.layers[1].bias,))
        if E == 256 and F1 % 256 == 0 and x2.stride(1) == 1:
            t = ops.decoder_ffn(x2, w1, b1, w2, b2, F1)          # both projections, hidden block in LDS
        else:
            h = lr(x2, w1, F1, b1, relu_cols=F1)
            t = lr(h, w2, E, b2, res=x2, ksplit=8 if F1 >= 1024 else 1)
        if raw:
            return t if t.dim() == 3 else t.unsqueeze(0)
        return ops.layernorm_chain(t, (n2.weight, n2.bias, n2.eps), pos,
                                   None if post_norm is None else (post_norm.weight, post_norm.bias, post_norm.eps))

    def forward_fast(self, query, query_pos, kv, bits):
        """('cross_attn','norm','self_attn','norm','ffn','norm') on batch-first tensors (dropouts are 0)."""
        query = residual_layernorm(self.attentions[0].attend(query, query_pos, kv, bits), None, self.norms[0])
        query = residual_layernorm(self.attentions[1].self_attend(query, query_pos), None, self.norms[1])
        ffn = self.ffns[0]
        if len(ffn.layers) == 3 and isinstance(ffn.layers[0][1], nn.ReLU) and ffn.add_identity:
            h = small_linear(query, ffn.layers[0][0].weight, ffn.layers[0][0].bias, relu=True)
            x = small_linear(h, ffn.layers[1].weight, ffn.layers[1].bias, res=query)
        else:
            x = ffn(query)
        return residual_layernorm(x, None, self.norms[2])


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetrTransformerDecoder(TransformerLayerSequence):
    """[3P] mmdet DetrTransformerDecoder (post_norm applied by the head, mask2former_head.py:734)."""

    def __init__(self, *args, post_norm_cfg=dict(type='LN'), return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.post_norm = build_norm(post_norm_cfg, self.embed_dims)[1] if post_norm_cfg is not None else None

    def forward(self, query, *args, **kwargs):
        if not self.return_intermediate:
            x = super().forward(query, *args, **kwargs)
            if self.post_norm:
                x = self.post_norm(x)[None]
            return x
        intermediate = []
        for layer in self.layers:
            query = layer(query, *args, **kwargs)
            if self.return_intermediate:
                intermediate.append(self.post_norm(query) if self.post_norm is not None else query)
        return torch.stack(intermediate)
