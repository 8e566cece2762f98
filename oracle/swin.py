"""Oracle restatement of the [3P] mmdet 2.28 `SwinTransformer` backbone (BASELINE configs[3]). TEST INFRASTRUCTURE ONLY.

The source is not under /root/reference (no shipped CGG config selects Swin; BASELINE.json configs[3] names it), so this
restates the PUBLISHED algorithm (Liu et al., "Swin Transformer", 2021; upstream mmdet/models/backbones/swin.py
semantics as listed in SURVEY.md 8(b)) in a deliberately different formulation from the product's `swin.py`, so that
the two only agree if both are right -- "parity unpinned by the reference", pinned by independence:

  * window attention is computed DENSELY over all tokens of the padded map: token i attends token j iff, after the
    cyclic shift, they fall into the same window; the additive -100 of shifted windows is derived per pair from the
    three-slice region labels; the relative-position bias is looked up from per-pair coordinate differences
    ((dy + ws - 1) * (2 ws - 1) + dx + ws - 1) -- no roll, no window partition / reverse, no precomputed index buffer;
  * patch merging gathers the 2x2 neighbours by explicit indexing into channel order c*4 + kh*2 + kw -- no nn.Unfold;
  * patch embedding is an explicit patch gather + matmul -- no convolution.

O(N^2) memory: for small maps only (tests use <= ~1000 tokens at stage 0). Same parameter names as upstream, so the
product's `state_dict` loads directly.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attn(nn.Module):
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), heads))
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _WMSA(nn.Module):
    def __init__(self, dim, heads, ws, shift):
        super().__init__()
        self.w_msa = _Attn(dim, heads, ws)
        self.ws, self.shift, self.heads = ws, shift, heads

    def forward(self, x, hw):
        B, L, C = x.shape
        H, W = hw
        ws, s, nh = self.ws, self.shift, self.heads
        Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
        # zero-padded map (padding tokens take part in attention as keys, as upstream pads AFTER norm1)
        xp = x.new_zeros(B, Hp, Wp, C)
        xp[:, :H, :W] = x.view(B, H, W, C)
        xp = xp.view(B, Hp * Wp, C)
        ys, xs = torch.meshgrid(torch.arange(Hp), torch.arange(Wp), indexing='ij')
        ys, xs = ys.flatten(), xs.flatten()
        ysh, xsh = (ys - s) % Hp, (xs - s) % Wp                 # coordinates after the cyclic shift by -s
        win = (ysh // ws) * (Wp // ws) + xsh // ws

        def region(c, n):                                       # slices (0, -ws), (-ws, -s), (-s, None)
            return torch.where(c < n - ws, 0, torch.where(c < n - s, 1, 2)) if s > 0 else torch.zeros_like(c)
        reg = region(ysh, Hp) * 3 + region(xsh, Wp)
        same_win = win[:, None] == win[None, :]
        dy = (ysh % ws)[:, None] - (ysh % ws)[None, :]
        dx = (xsh % ws)[:, None] - (xsh % ws)[None, :]
        ridx = (dy + ws - 1) * (2 * ws - 1) + (dx + ws - 1)
        ridx = torch.where(same_win, ridx, torch.zeros_like(ridx))
        a = self.w_msa
        bias = a.relative_position_bias_table[ridx.flatten()].view(Hp * Wp, Hp * Wp, nh).permute(2, 0, 1)
        add = torch.where(reg[:, None] != reg[None, :], -100.0, 0.0).to(x.dtype)
        qkv = a.qkv(xp).view(B, Hp * Wp, 3, nh, C // nh)
        q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
        logits = (q * (C // nh) ** -0.5) @ k.transpose(-2, -1) + bias[None] + add[None, None]
        logits = logits.masked_fill(~same_win[None, None], float('-inf'))
        out = (logits.softmax(-1) @ v).transpose(1, 2).reshape(B, Hp * Wp, C)
        out = a.proj(out).view(B, Hp, Wp, C)[:, :H, :W]
        return out.reshape(B, H * W, C)


class _FFN(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Identity()),
                                    nn.Linear(hidden, dim), nn.Identity())


class _Block(nn.Module):
    def __init__(self, dim, heads, hidden, ws, shift):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WMSA(dim, heads, ws, ws // 2 if shift else 0)
        self.norm2 = nn.LayerNorm(dim)
        self.ffn = _FFN(dim, hidden)

    def forward(self, x, hw):
        x = x + self.attn(self.norm1(x), hw)
        return x + self.ffn.layers(self.norm2(x))


class _Merge(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.LayerNorm(4 * cin)
        self.reduction = nn.Linear(4 * cin, cout, bias=False)

    def forward(self, x, hw):
        B, L, C = x.shape
        H, W = hw
        g = x.new_zeros(B, H + H % 2, W + W % 2, C)
        g[:, :H, :W] = x.view(B, H, W, C)
        Ho, Wo = g.shape[1] // 2, g.shape[2] // 2
        out = x.new_zeros(B, Ho, Wo, 4 * C)
        for kh in range(2):
            for kw in range(2):
                out[..., kh * 2 + kw::4] = g[:, kh::2, kw::2, :]          # channel c -> c*4 + kh*2 + kw
        return self.reduction(self.norm(out.view(B, Ho * Wo, 4 * C))), (Ho, Wo)


class _Stage(nn.Module):
    def __init__(self, dim, heads, hidden, depth, ws, down):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(dim, heads, hidden, ws, i % 2 == 1) for i in range(depth)])
        self.downsample = down


class _Embed(nn.Module):
    def __init__(self, cin, dim, p, norm):
        super().__init__()
        self.p = p
        self.projection = nn.Conv2d(cin, dim, p, stride=p)      # container for the upstream parameter names only
        self.norm = nn.LayerNorm(dim) if norm else None


class OracleSwin(nn.Module):
    def __init__(self, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), out_indices=(0, 1, 2, 3), patch_norm=True, in_channels=3, **unused):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.patch_embed = _Embed(in_channels, embed_dims, patch_size, patch_norm)
        self.stages = nn.ModuleList()
        ch = embed_dims
        self.dims = []
        for i, d in enumerate(depths):
            down = _Merge(ch, 2 * ch) if i < len(depths) - 1 else None
            self.stages.append(_Stage(ch, num_heads[i], int(mlp_ratio * ch), d, window_size, down))
            self.dims.append(ch)
            if down is not None:
                ch *= 2
        for i in self.out_indices:
            self.add_module(f'norm{i}', nn.LayerNorm(self.dims[i]))

    def forward(self, img):
        pe = self.patch_embed
        p = pe.p
        B, Cin, H, W = img.shape
        img = F.pad(img, (0, (p - W % p) % p, 0, (p - H % p) % p))
        Hn, Wn = img.shape[2] // p, img.shape[3] // p
        patches = img.view(B, Cin, Hn, p, Wn, p).permute(0, 2, 4, 1, 3, 5).reshape(B, Hn * Wn, Cin * p * p)
        x = patches @ pe.projection.weight.view(pe.projection.weight.shape[0], -1).t() + pe.projection.bias
        if pe.norm is not None:
            x = pe.norm(x)
        hw = (Hn, Wn)
        outs = []
        for i, st in enumerate(self.stages):
            for blk in st.blocks:
                x = blk(x, hw)
            if i in self.out_indices:
                o = getattr(self, f'norm{i}')(x)
                outs.append(o.view(B, hw[0], hw[1], -1).permute(0, 3, 1, 2).contiguous())
            if st.downsample is not None:
                x, hw = st.downsample(x, hw)
        return tuple(outs)
