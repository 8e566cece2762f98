"""Caption generation head: `type='CaptionTransformer'`
(open_set/models/transformers/caption_tranformer.py:17-44 + transformers.py:9-267).

A 4-layer post-norm transformer decoder over the query embeddings with a 768 -> 30522 generator.
Parameter / buffer names reproduce the reference module tree:
  position_encoder.psne_layer, transformer_decoder.decoders.N.{mha_layer.{qkv_layer,out_layer},
  crx_layer.{to_qry,to_key,to_val,to_out}, ffn_layer.linears.{0,1}.0, layer_normalz.{mha,crx,ffn}.1},
  generator  -- note `qkv_layer` packs per-head interleaved [q_h | k_h | v_h] (transformers.py:114-117).
The arithmetic is a compact functional forward over those parameters (one fused attention helper for
self and cross attention); GEMMs go through hipBLASLt (bf16 autocast in throughput mode).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import runtime
from .registry import HEADS


def sine_table(seq_length, dim):
    """PositionalEncoding table of transformers.py:12-20: sin on even, cos on odd, 10000^((j - j%2)/dim)."""
    pos = torch.arange(seq_length, dtype=torch.float64)[:, None]
    j = torch.arange(dim, dtype=torch.float64)[None, :]
    even = (j % 2 == 0)
    ang = pos / (10000.0 ** ((j - j % 2) / dim))
    return torch.where(even, ang.sin(), ang.cos()).float()


class PositionalEncoding(nn.Module):

    def __init__(self, seq_length, in_dim, drop_val=0.1):
        super().__init__()
        self.drop_layer = nn.Dropout(drop_val)
        self.register_buffer('psne_layer', sine_table(seq_length, in_dim))

    def forward(self, src):
        return self.drop_layer(src + self.psne_layer[:src.shape[1]][None])


def _attend(q, k, v, nb_heads, mask=None, key_padding_mask=None):
    """q (B,Lq,H,d) k,v (B,Lk,H,d); mask (Lq,Lk) bool True=blocked; key_padding_mask (B,Lk) bool."""
    d = q.shape[-1]
    w = torch.einsum('bqhd,bkhd->bhqk', q, k) / math.sqrt(d)
    if mask is not None:
        w = w.masked_fill(mask, float('-inf'))
    if key_padding_mask is not None:
        w = w.masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
    w = torch.softmax(w.float(), dim=-1).to(v.dtype)
    return torch.einsum('bhqk,bkhd->bqhd', w, v).flatten(2)


class MultiHeadSelfAttention(nn.Module):

    def __init__(self, in_dim, nb_heads):
        super().__init__()
        self.nbr_heads, self.heads_dim = nb_heads, in_dim // nb_heads
        self.qkv_layer = runtime.ParityLinear(in_dim, 3 * in_dim)
        self.out_layer = runtime.ParityLinear(in_dim, in_dim)

    def forward(self, src, mask=None, key_padding_mask=None):
        B, L, _ = src.shape
        qkv = self.qkv_layer(src).view(B, L, self.nbr_heads, 3, self.heads_dim)
        out = _attend(qkv[..., 0, :], qkv[..., 1, :], qkv[..., 2, :], self.nbr_heads, mask, key_padding_mask)
        return self.out_layer(out)


class MultiHeadCrossAttention(nn.Module):

    def __init__(self, in_dim, nb_heads):
        super().__init__()
        self.nbr_heads, self.heads_dim = nb_heads, in_dim // nb_heads
        self.to_qry = runtime.ParityLinear(in_dim, in_dim)
        self.to_key = runtime.ParityLinear(in_dim, in_dim)
        self.to_val = runtime.ParityLinear(in_dim, in_dim)
        self.to_out = runtime.ParityLinear(in_dim, in_dim)

    def forward(self, qry, key, val, mask=None, key_padding_mask=None):
        B, Lq, _ = qry.shape
        Lk = key.shape[1]
        H, d = self.nbr_heads, self.heads_dim
        out = _attend(self.to_qry(qry).view(B, Lq, H, d), self.to_key(key).view(B, Lk, H, d),
                      self.to_val(val).view(B, Lk, H, d), H, mask, key_padding_mask)
        return self.to_out(out)


_ACTS = {0: nn.Identity, 1: nn.ReLU, 2: nn.GELU, 3: nn.Sigmoid, 4: nn.Tanh}


class FeedForwardNetwork(nn.Module):
    """linears[i] = Sequential(Linear, Dropout|Identity, activation) -- dropout BEFORE the activation
    (transformers.py:43-47)."""

    def __init__(self, layer_cfg, activations, drop_vals):
        super().__init__()
        self.linears = nn.ModuleList()
        for i, (a, b) in enumerate(zip(layer_cfg[:-1], layer_cfg[1:])):
            act = nn.Softmax(dim=-1) if activations[i] == 5 else _ACTS.get(activations[i], nn.Identity)()
            self.linears.append(nn.Sequential(
                runtime.ParityLinear(a, b), nn.Dropout(drop_vals[i]) if drop_vals[i] > 0.0 else nn.Identity(), act))

    def forward(self, x):
        for blk in self.linears:
            x = blk(x)
        return x


def _norm_pair(in_dim, pre_norm):
    return nn.ModuleList([nn.LayerNorm(in_dim) if pre_norm else nn.Identity(),
                          nn.LayerNorm(in_dim) if not pre_norm else nn.Identity()])


class DecoderBlock(nn.Module):

    def __init__(self, in_dim, ff_dim, nb_heads, drop_val=0.1, pre_norm=False):
        super().__init__()
        assert in_dim % nb_heads == 0
        self.mha_layer = MultiHeadSelfAttention(in_dim, nb_heads)
        self.crx_layer = MultiHeadCrossAttention(in_dim, nb_heads)
        self.ffn_layer = FeedForwardNetwork([in_dim, ff_dim, in_dim], [1, 0], [drop_val, 0.0])
        self.dropout_layer = nn.ModuleDict({k: nn.Dropout(drop_val) for k in ('mha', 'crx', 'ffn')})
        self.layer_normalz = nn.ModuleDict({k: _norm_pair(in_dim, pre_norm) for k in ('mha', 'crx', 'ffn')})

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        n, dr = self.layer_normalz, self.dropout_layer
        x = n['mha'][0](tgt)
        x = n['mha'][1](x + dr['mha'](self.mha_layer(x, tgt_mask, tgt_key_padding_mask)))
        y = n['crx'][0](x)
        y = n['crx'][1](y + dr['crx'](self.crx_layer(y, memory, memory, memory_mask, memory_key_padding_mask)))
        z = n['ffn'][0](y)
        # the FFN reads the un-normalised residual stream `y` (transformers.py:228-229); identical to
        # `z` in post-norm mode, which is the only mode the configs use
        return n['ffn'][1](z + dr['ffn'](self.ffn_layer(y)))


class TransformerDecoder(nn.Module):

    def __init__(self, nb_layers, in_dim, ff_dim, nb_heads, drop_val=0.1, pre_norm=False):
        super().__init__()
        self.decoders = nn.ModuleList(
            [DecoderBlock(in_dim, ff_dim, nb_heads, drop_val, pre_norm) for _ in range(nb_layers)])

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        outs = []
        for blk in self.decoders:
            tgt = blk(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask)
            outs.append(tgt)
        return outs


_CAUSAL = {}


def build_mask(seq):
    """Causal mask (True = blocked), cached per (length, device)."""
    L = seq.shape[1]
    key = (L, str(seq.device))
    m = _CAUSAL.get(key)
    if m is None:
        m = torch.ones(L, L, dtype=torch.bool).triu(1).to(seq.device)
        _CAUSAL[key] = m
    return m


def build_key_padding_mask(seq, pad_idx):
    return seq == pad_idx


class _GeneratorCEFn(torch.autograd.Function):
    """Per-row cross-entropy of `generator(hidden)` without ever holding the (M, 30522) logits (mask2former_head.py:551-565):
    row chunks of logits come from a library GEMM against the padded generator weight into ONE reused buffer and are
    consumed by `cgg_ce_rows_forward` (log-sum-exp + row loss). Backward recomputes a chunk, `cgg_ce_rows_backward` turns
    it in place into g_row (softmax - onehot), and the two gradient GEMMs read it. bf16 logits / f32 statistics in
    throughput mode (what bf16 autocast of the reference formulation computes), f32 in parity mode -- there the logits GEMM (forward
    and the backward's recomputation) and the weight / bias gradient run on the f32-class x3 kernels (`ops.gemm_x3`, `ops.wgrad_x3`:
    (2048, 768) x (768, 30528) in ~0.35 ms instead of the f32 library's 0.84); the grad-input GEMM (768 output columns: 96 tiles,
    too few workgroups for the x3 kernel) stays on the library. CGG_X3_GENERATOR=0 restores the library GEMMs."""

    CHUNK = 2048
    X3 = os.environ.get('CGG_X3_GENERATOR', '1') != '0'

    @staticmethod
    def _x3_ok(x, w):
        return (_GeneratorCEFn.X3 and runtime._X3_TRAIN and runtime.x3_enabled() and x.dtype == torch.float32 and x.is_cuda
                and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and x.shape[0] >= 256)

    @staticmethod
    def _operands(weight, bias, dt):
        from . import ops
        N, K = weight.shape
        pad = (-N) % 8

        def make():
            w = torch.cat([weight.detach(), weight.new_zeros((pad, K))], 0) if pad else weight.detach()
            b = torch.cat([bias.detach(), bias.new_full((pad,), -1e30)], 0) if pad else bias.detach()
            return w.to(dt).contiguous(), b.to(dt).contiguous()
        return runtime.derived_cached('generator_ce_%s' % dt, (weight, bias), make)

    @staticmethod
    def forward(ctx, hidden, weight, bias, target, ignore_index):
        from . import ops
        dt = torch.bfloat16 if runtime.is_bf16() else torch.float32
        w, b = _GeneratorCEFn._operands(weight, bias, dt)
        x = hidden.detach().to(dt).contiguous()
        M = x.shape[0]
        C = min(_GeneratorCEFn.CHUNK, M)
        buf = torch.empty((C, w.shape[0]), dtype=dt, device=x.device)
        loss = torch.empty(M, dtype=torch.float32, device=x.device)
        lse = torch.empty(M, dtype=torch.float32, device=x.device)
        wk = runtime.derived_cached('generator_ce_x3', (weight, bias), lambda: ops.pack_linear_weight_x3(w)) \
            if _GeneratorCEFn._x3_ok(x, w) else None
        for r0 in range(0, M, C):
            r1 = min(r0 + C, M)
            if wk is not None:
                lg = ops.gemm_x3(x[r0:r1], wk, w.shape[0], b, out=buf[:r1 - r0])
            else:
                lg = torch.addmm(b, x[r0:r1], w.t(), out=buf[:r1 - r0])
            l, s = ops.ce_rows_forward(lg, target[r0:r1], ignore_index)
            loss[r0:r1] = l
            lse[r0:r1] = s
        ctx.save_for_backward(x, w, b, target, lse)
        ctx.wk = wk
        ctx.ignore_index = ignore_index
        ctx.n_out = weight.shape[0]
        ctx.in_dtype = hidden.dtype
        return loss

    @staticmethod
    def backward(ctx, grad_rows):
        from . import ops
        x, w, b, target, lse = ctx.saved_tensors
        M = x.shape[0]
        C = min(_GeneratorCEFn.CHUNK, M)
        buf = torch.empty((C, w.shape[0]), dtype=x.dtype, device=x.device)
        gx = torch.empty_like(x)
        gw = torch.zeros(w.shape, dtype=torch.float32, device=x.device)
        gb = torch.zeros(w.shape[0], dtype=torch.float32, device=x.device)
        wk = ctx.wk
        gwc = torch.empty(w.shape, dtype=x.dtype, device=x.device) if wk is None else None   # one chunk's weight gradient, reused
        g = grad_rows.float().contiguous()
        for r0 in range(0, M, C):
            r1 = min(r0 + C, M)
            if wk is not None:
                lg = ops.gemm_x3(x[r0:r1], wk, w.shape[0], b, out=buf[:r1 - r0])
            else:
                lg = torch.addmm(b, x[r0:r1], w.t(), out=buf[:r1 - r0])
            dl = ops.ce_rows_backward_(lg, target[r0:r1], lse[r0:r1], g[r0:r1], ctx.ignore_index)
            torch.mm(dl, w, out=gx[r0:r1])
            if wk is not None:
                # weight and bias gradient of the chunk from one transpose-read pass over g_row (pre-scaled by its exact maximum)
                gwi, gbi = ops.wgrad_x3(dl, x[r0:r1], want_bias=True, amax=ops.absmax(dl))
                gw += gwi
                gb += gbi
                continue
            # summed in f32 (one rounding of each chunk's partial to the GEMM dtype: 3 chunks at configs[2], below the bf16
            # GEMM's own rounding)
            gw += torch.mm(dl.t(), x[r0:r1], out=gwc)
            gb += dl.sum(0, dtype=torch.float32)
        n = ctx.n_out
        return gx.to(ctx.in_dtype), gw[:n], gb[:n], None, None


@HEADS.register_module()
class CaptionTransformer(nn.Module):
    """forward(tgt (B,L,in), memory (B,Q,in), ...) -> (list of per-layer outputs, last-layer logits (B,L,V))."""

    def __init__(self, nb_layers, input_dim, hidden_dim, ff_dim, nb_heads, drop_val, pre_norm, seq_length,
                 nb_tokens):
        super().__init__()
        self.adapter = runtime.ParityLinear(input_dim, hidden_dim) if input_dim != hidden_dim else nn.Identity()
        self.position_encoder = PositionalEncoding(seq_length, hidden_dim)
        self.transformer_decoder = TransformerDecoder(nb_layers=nb_layers, in_dim=hidden_dim, ff_dim=ff_dim,
                                                      nb_heads=nb_heads, drop_val=drop_val, pre_norm=pre_norm)
        self.generator = runtime.ParityLinear(hidden_dim, nb_tokens)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        with runtime.autocast():
            memory = self.adapter(memory)
            tgt = self.position_encoder(tgt)
            if tgt_mask is None:
                tgt_mask = build_mask(tgt).to(tgt.device)
            output = self.transformer_decoder(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask,
                                              memory_key_padding_mask)
            logits = self.generator(output[-1])
        return output, logits.float()

    def forward_hidden(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                       memory_key_padding_mask=None):
        """The decoder stack WITHOUT the generator: last layer's hidden states (B, L, hidden) f32."""
        with runtime.autocast():
            memory = self.adapter(memory)
            tgt = self.position_encoder(tgt)
            if tgt_mask is None:
                tgt_mask = build_mask(tgt).to(tgt.device)
            output = self.transformer_decoder(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask,
                                              memory_key_padding_mask)
        return output[-1].float()

    # ---- incremental decoding (caption_search.beam_search): each step runs ONE new position per beam ----
    def begin_decode(self, memory):
        """State of an incremental decode over `memory` (1, Q, in): the cross-attention keys / values of every block
        (computed once -- the reference recomputes them for every beam at every step, inference.py:108-113) and empty
        per-block self-attention key / value prefixes."""
        with runtime.autocast():
            mem = self.adapter(memory)
            cross = []
            for blk in self.transformer_decoder.decoders:
                c = blk.crx_layer
                Q = mem.shape[1]
                cross.append((c.to_key(mem).view(1, Q, c.nbr_heads, c.heads_dim),
                              c.to_val(mem).view(1, Q, c.nbr_heads, c.heads_dim)))
        n = len(self.transformer_decoder.decoders)
        return dict(cross=cross, k=[None] * n, v=[None] * n, length=0)

    def decode_step(self, tok, state, parents=None):
        """One position for every live sequence: `tok` (nb, 1, in) = the embedded newest token of each, `parents` (nb,)
        long = the row of the previous step each sequence continues (None: same rows). Causal attention makes position
        t of every block a function of positions <= t only, so the cached prefix keys / values are exactly the ones a
        full re-run of the sequence would compute. Returns the per-block outputs at the new position, each (nb, hidden)
        -- `TransformerDecoder.forward(...)[i][:, -1]` of the whole sequences."""
        nb = tok.shape[0]
        with runtime.autocast():
            x = tok + self.position_encoder.psne_layer[state['length']][None, None]
            outs = []
            for i, blk in enumerate(self.transformer_decoder.decoders):
                n, m, c = blk.layer_normalz, blk.mha_layer, blk.crx_layer
                H, d = m.nbr_heads, m.heads_dim
                h = n['mha'][0](x)
                qkv = m.qkv_layer(h).view(nb, 1, H, 3, d)
                k, v = qkv[..., 1, :], qkv[..., 2, :]
                if state['k'][i] is not None:
                    pk, pv = state['k'][i], state['v'][i]
                    if parents is not None:
                        pk, pv = pk.index_select(0, parents), pv.index_select(0, parents)
                    k, v = torch.cat([pk, k], 1), torch.cat([pv, v], 1)
                state['k'][i], state['v'][i] = k, v
                h = n['mha'][1](h + m.out_layer(_attend(qkv[..., 0, :], k, v, H)))
                y = n['crx'][0](h)
                ck, cv = state['cross'][i]
                att = _attend(c.to_qry(y).view(nb, 1, H, d), ck.expand(nb, -1, -1, -1), cv.expand(nb, -1, -1, -1), H)
                y = n['crx'][1](y + c.to_out(att))
                x = n['ffn'][1](n['ffn'][0](y) + blk.ffn_layer(y))
                outs.append(x[:, 0])
            state['length'] += 1
        return outs

    def generator_ce_rows(self, hidden, target, ignore_index=None):
        """`F.cross_entropy(generator(hidden), target, reduction='none', ignore_index=...)` for hidden (M, hidden) and
        target (M,) without materialising the (M, nb_tokens) logits (HIP row kernels + chunked library GEMMs)."""
        return _GeneratorCEFn.apply(hidden, self.generator.weight, self.generator.bias, target.contiguous(), ignore_index)
