"""`SwinTransformer` backbone for BASELINE configs[3] (Swin-B + 200 queries). No reference config selects it, so the
contract is the upstream one ([3P] mmdet 2.28 `mmdet/models/backbones/swin.py`): constructor keys
(`embed_dims, depths, num_heads, window_size, mlp_ratio, out_indices, patch_norm, drop_path_rate, ...`), parameter
names (`patch_embed.projection`, `stages.N.blocks.M.attn.w_msa.{qkv,proj,relative_position_bias_table}`,
`ffn.layers.0.0 / 1`, `stages.N.downsample.{norm,reduction}`, `normN`) and the unfold-ordered patch merging, so an
mmdet Swin checkpoint loads unchanged. Plain PyTorch (scaled_dot_product_attention for the window attention, bf16
autocast in throughput mode): the backbone is outside the hand-written-kernel scope of the hot path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import runtime
from .registry import BACKBONES


def _double_step_seq(step1, len1, step2, len2):
    s1 = torch.arange(0, step1 * len1, step1)
    s2 = torch.arange(0, step2 * len2, step2)
    return (s1[:, None] + s2[None, :]).reshape(1, -1)


class WindowMSA(nn.Module):

    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None, attn_drop_rate=0.,
                 proj_drop_rate=0.):
        super().__init__()
        self.embed_dims, self.window_size, self.num_heads = embed_dims, window_size, num_heads
        head = embed_dims // num_heads
        self.scale = qk_scale or head**-0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        idx = _double_step_seq(2 * Ww - 1, Wh, 1, Ww)
        idx = (idx + idx.T).flip(1).contiguous()
        self.register_buffer('relative_position_index', idx)
        self.qkv = runtime.ParityLinear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop_rate)
        self.proj = runtime.ParityLinear(embed_dims, embed_dims)
        self.proj_drop = nn.Dropout(proj_drop_rate)

    def forward(self, x, mask=None):
        """x (nW*B, N, C); mask (nW, N, N) additive or None."""
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)]
        bias = bias.view(N, N, -1).permute(2, 0, 1).unsqueeze(0)                 # (1, heads, N, N)
        if mask is not None:
            nW = mask.shape[0]
            bias = (bias.unsqueeze(1) + mask.view(1, nW, 1, N, N)).expand(B_ // nW, nW, -1, N, N)
            bias = bias.reshape(B_, self.num_heads, N, N)
        x = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.to(q.dtype), scale=self.scale,
                                           dropout_p=self.attn_drop.p if self.training else 0.0)
        x = x.transpose(1, 2).reshape(B_, N, C)
        return self.proj_drop(self.proj(x))


class ShiftWindowMSA(nn.Module):

    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None,
                 attn_drop_rate=0., proj_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        assert 0 <= shift_size < window_size
        self.w_msa = WindowMSA(embed_dims, num_heads, (window_size, window_size), qkv_bias, qk_scale,
                               attn_drop_rate, proj_drop_rate)
        self.drop_path_rate = drop_path_rate

    def _partition(self, x):
        B, H, W, C = x.shape
        ws = self.window_size
        x = x.view(B, H // ws, ws, W // ws, ws, C)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)

    def _reverse(self, windows, H, W):
        ws = self.window_size
        B = int(windows.shape[0] / (H * W / ws / ws))
        x = windows.view(B, H // ws, W // ws, ws, ws, -1)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)

    def forward(self, query, hw_shape):
        B, L, C = query.shape
        H, W = hw_shape
        assert L == H * W, 'input feature has wrong size'
        ws = self.window_size
        query = query.view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        query = F.pad(query, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = query.shape[1], query.shape[2]
        mask = None
        if self.shift_size > 0:
            query = torch.roll(query, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
            img_mask = torch.zeros((1, Hp, Wp, 1), device=query.device)
            slices = (slice(0, -ws), slice(-ws, -self.shift_size), slice(-self.shift_size, None))
            cnt = 0
            for h in slices:
                for w in slices:
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = self._partition(img_mask).view(-1, ws * ws)
            mask = mw.unsqueeze(1) - mw.unsqueeze(2)
            mask = mask.masked_fill(mask != 0, float(-100.0)).masked_fill(mask == 0, float(0.0))
        windows = self._partition(query).view(-1, ws * ws, C)
        attn = self.w_msa(windows, mask=mask).view(-1, ws, ws, C)
        x = self._reverse(attn, Hp, Wp)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        if pad_r > 0 or pad_b > 0:
            x = x[:, :H, :W, :].contiguous()
        x = x.view(B, H * W, C)
        if self.training and self.drop_path_rate > 0:          # stochastic depth, per sample
            keep = 1 - self.drop_path_rate
            x = x * (torch.rand((B, 1, 1), device=x.device, dtype=x.dtype) < keep).to(x.dtype) / keep
        return x


class _SwinFFN(nn.Module):
    """[3P] mmcv FFN with GELU: `layers.0.0`, `layers.1`, identity added by the caller's argument."""

    def __init__(self, embed_dims, feedforward_channels, drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(runtime.ParityLinear(embed_dims, feedforward_channels), nn.GELU(),
                                                  nn.Dropout(drop_rate)),
                                    runtime.ParityLinear(feedforward_channels, embed_dims), nn.Dropout(drop_rate))
        self.drop_path_rate = drop_path_rate

    def forward(self, x, identity):
        out = self.layers(x)
        if self.training and self.drop_path_rate > 0:
            keep = 1 - self.drop_path_rate
            out = out * (torch.rand((x.shape[0], 1, 1), device=x.device, dtype=x.dtype) < keep).to(x.dtype) / keep
        return identity + out


class SwinBlock(nn.Module):

    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias,
                                   qk_scale, attn_drop_rate, drop_rate, drop_path_rate)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = _SwinFFN(embed_dims, feedforward_channels, drop_rate, drop_path_rate)

    def forward(self, x, hw_shape):
        x = x + self.attn(self.norm1(x), hw_shape)
        return self.ffn(self.norm2(x), identity=x)


class PatchMerging(nn.Module):
    """2x2 unfold (channel order c*4 + kh*2 + kw, as upstream) -> LN(4C) -> Linear(4C, 2C, bias=False)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.sampler = nn.Unfold(kernel_size=2, stride=2)
        self.norm = nn.LayerNorm(4 * in_channels)
        self.reduction = runtime.ParityLinear(4 * in_channels, out_channels, bias=False)

    def forward(self, x, input_size):
        B, L, C = x.shape
        H, W = input_size
        x = x.view(B, H, W, C).permute(0, 3, 1, 2)
        x = F.pad(x, (0, W % 2, 0, H % 2))                     # 'corner' padding to an even size
        Ho, Wo = (H + H % 2) // 2, (W + W % 2) // 2
        x = self.sampler(x).transpose(1, 2)                    # (B, Ho*Wo, 4C)
        return self.reduction(self.norm(x)), (Ho, Wo)


class SwinBlockSequence(nn.Module):

    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size, qkv_bias, qk_scale,
                 drop_rate, attn_drop_rate, drop_path_rates, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, shift=(i % 2 == 1),
                      qkv_bias=qkv_bias, qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                      drop_path_rate=drop_path_rates[i]) for i in range(depth)])
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for blk in self.blocks:
            x = blk(x, hw_shape)
        if self.downsample is not None:
            x_down, down_hw = self.downsample(x, hw_shape)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


class PatchEmbed(nn.Module):

    def __init__(self, in_channels, embed_dims, patch_size, norm):
        super().__init__()
        self.patch_size = patch_size
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dims) if norm else None

    def forward(self, x):
        p = self.patch_size
        H, W = x.shape[-2:]
        x = F.pad(x, (0, (p - W % p) % p, 0, (p - H % p) % p))
        x = self.projection(x)
        hw = (x.shape[2], x.shape[3])
        x = x.flatten(2).transpose(1, 2)
        if self.norm is not None:
            x = self.norm(x)
        return x, hw


@BACKBONES.register_module()
class SwinTransformer(nn.Module):

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 use_abs_pos_embed=False, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'), with_cp=False,
                 pretrained=None, convert_weights=False, frozen_stages=-1, init_cfg=None):
        super().__init__()
        if use_abs_pos_embed:
            raise NotImplementedError('use_abs_pos_embed=True is not used by any CGG / BASELINE config')
        assert strides[0] == patch_size, 'Use non-overlapping patch embed.'
        self.out_indices, self.frozen_stages = tuple(out_indices), frozen_stages
        self.patch_embed = PatchEmbed(in_channels, embed_dims, patch_size, patch_norm)
        self.drop_after_pos = nn.Dropout(p=drop_rate)
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        self.stages = nn.ModuleList()
        ch = embed_dims
        self.num_features = []
        for i, depth in enumerate(depths):
            down = PatchMerging(ch, 2 * ch) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(ch, num_heads[i], int(mlp_ratio * ch), depth, window_size, qkv_bias,
                                                 qk_scale, drop_rate, attn_drop_rate,
                                                 dpr[sum(depths[:i]):sum(depths[:i + 1])], down))
            self.num_features.append(ch)
            if down is not None:
                ch = 2 * ch
        for i in self.out_indices:
            self.add_module(f'norm{i}', nn.LayerNorm(self.num_features[i]))
        self._freeze_stages()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
            elif isinstance(m, WindowMSA):
                nn.init.trunc_normal_(m.relative_position_bias_table, std=.02)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
            self.drop_after_pos.eval()
        for i in range(1, self.frozen_stages + 1):
            if (i - 1) in self.out_indices:
                norm = getattr(self, f'norm{i - 1}')
                norm.eval()
                for p in norm.parameters():
                    p.requires_grad = False
            m = self.stages[i - 1]
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    def forward(self, x):
        with runtime.autocast():
            x, hw = self.patch_embed(x)
            x = self.drop_after_pos(x)
            outs = []
            for i, stage in enumerate(self.stages):
                x, hw, out, out_hw = stage(x, hw)
                if i in self.out_indices:
                    out = getattr(self, f'norm{i}')(out)
                    outs.append(out.view(-1, out_hw[0], out_hw[1], self.num_features[i]).permute(0, 3, 1, 2))
        if runtime.is_bf16() and not torch.is_grad_enabled() and x.is_cuda:
            # throughput mode: hand the pixel decoder channel-last bf16 views (what its inference stream consumes)
            return tuple(o.to(torch.bfloat16) for o in outs)
        return tuple(o.float().contiguous() for o in outs)
