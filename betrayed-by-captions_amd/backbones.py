"""Backbone named by the reference configs: `type='ResNet', depth=50, out_indices=(0,1,2,3),
frozen_stages, norm_cfg=BN(requires_grad=False), norm_eval=True, style='pytorch'`
(configs/instance/coco_b48n17.py:17-26) -- [3P] mmdet ResNet restated in plain torch with the upstream
parameter names (`conv1`, `bn1`, `layer{1-4}.N.{conv,bn}{1-3}`, `downsample.{0,1}`), so torchvision /
mmdet ResNet checkpoints load unchanged. Convolutions run on MIOpen (bf16 autocast in throughput mode);
the backbone is outside the hand-written-kernel scope of the hot path.
"""
import os

import torch
import torch.nn as nn

from . import ops, runtime
from .registry import BACKBONES

# throughput mode: ResNet layer1 identity Bottlenecks as one HIP launch (ops.bottleneck64); CGG_FUSED_BOTTLENECK=0 = three library calls
FUSED_BOTTLENECK = os.environ.get('CGG_FUSED_BOTTLENECK', '1') != '0'
# training: frozen stem + stages on the BN-folded inference path under no_grad (CGG_FROZEN_FOLDED=0 = autograd-recorded torch path)
FROZEN_FOLDED = os.environ.get('CGG_FROZEN_FOLDED', '1') != '0'
FROZEN_NHWC_BF16 = os.environ.get('CGG_FROZEN_NHWC_BF16', '1') != '0'   # throughput mode: frozen stages' maps through the transpose kernel
# parity mode: the 7x7 stem on the x3 MFMA kernel (CGG_X3_STEM=0 = MIOpen f32 convolution, A/B)
X3_STEM = os.environ.get('CGG_X3_STEM', '1') != '0'


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, style='pytorch'):
        super().__init__()
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, style='pytorch'):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}
    # folded inference path: 3x3 convolutions with at most this many output pixels (per batch) run as im2col + GEMM
    IM2COL_MAX_PIXELS = 16384

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if deep_stem or avg_down or dcn is not None or plugins is not None or any(d != 1 for d in dilations):
            raise NotImplementedError('only the plain ResNet variants used by the CGG configs')
        block, stage_blocks = self.arch_settings[depth]
        self.depth, self.num_stages, self.out_indices = depth, num_stages, out_indices
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.init_cfg = init_cfg
        # parity-mode inference: hand the maps over as x3a rows (`ops.X3ATensor`, csrc/x3.h) instead of plain float32. OFF by
        # default -- only a consumer that reads x3a (the detector, when its head's pixel decoder does) switches it on
        self.x3a_outputs = False
        stem = stem_channels or base_channels
        self.conv1 = nn.Conv2d(in_channels, stem, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(stem)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = stem
        self.res_layers = []
        for i in range(num_stages):
            planes = base_channels * 2**i
            layers = []
            for j in range(stage_blocks[i]):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                         nn.BatchNorm2d(planes * block.expansion))
                layers.append(block(inplanes, planes, stride, down, style))
                inplanes = planes * block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        if not norm_cfg.get('requires_grad', True):
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    for p in m.parameters():
                        p.requires_grad = False
        self._freeze_stages()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)
            elif isinstance(m, BasicBlock):
                nn.init.constant_(m.bn2.weight, 0)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() replace buffer objects: drop the folded-weight caches that reference them
        for k in [k for k in self.__dict__ if k.startswith('_fold_')]:
            self.__dict__.pop(k, None)
        return super()._apply(fn, *args, **kwargs)

    # ---- throughput-mode inference: BN folded into the convolutions, bf16 NHWC filters kept resident ----
    def _folded(self, upto=None):
        """[(w, b)] per conv in execution order, bf16 channels_last, frozen BN folded in
        (w' = w * g / sqrt(var + eps), b' = beta - mean * g / sqrt(var + eps)). Rebuilt when any
        parameter / buffer version changes. `upto` (training with frozen stages): the stem and the first `upto` res layers only,
        keyed on THEIR tensors -- the optimiser bumps the versions of the trainable tail every step, which would otherwise re-fold
        (and re-pack every derived image of) all ~53 convolutions per training step."""
        ck, tk = ('_fold_cache', '_fold_tensors') if upto is None else ('_fold_cache_upto%d' % upto, '_fold_tensors_upto%d' % upto)
        layers = self.res_layers if upto is None else self.res_layers[:upto]
        tensors = self.__dict__.get(tk)
        if tensors is None:       # walking the module tree costs ~1 ms per forward; the tensor objects are stable
            mods = [self] if upto is None else [self.conv1, self.bn1] + [getattr(self, n) for n in layers]
            tensors = [t for m in mods for t in list(m.parameters()) + list(m.buffers())]
            self.__dict__[tk] = tensors
        key = sum(t._version for t in tensors)
        hit = self.__dict__.get(ck)
        if hit is not None and hit[0] == key and hit[1][0][0].device == self.conv1.weight.device:
            return hit[1]

        def fold(conv, bn):
            s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            w = (conv.weight.detach().float() * s.view(-1, 1, 1, 1)).to(torch.bfloat16)
            b = (bn.bias.detach().float() - bn.running_mean.detach().float() * s).to(torch.bfloat16).contiguous()
            if conv.kernel_size == (1, 1) and conv.groups == 1:
                return None, b, w.flatten(1).contiguous()      # 1x1: a plain (Cout, Cin) GEMM weight
            return w.contiguous(memory_format=torch.channels_last), b, None

        seq = [fold(self.conv1, self.bn1)]
        for name in layers:
            for blk in getattr(self, name):
                if blk.downsample is not None:
                    seq.append(fold(blk.downsample[0], blk.downsample[1]))
                seq.append(fold(blk.conv1, blk.bn1))
                seq.append(fold(blk.conv2, blk.bn2))
                if isinstance(blk, Bottleneck):
                    seq.append(fold(blk.conv3, blk.bn3))
        self.__dict__[ck] = (key, seq)
        return seq

    @staticmethod
    def _conv_nhwc(x, conv, folded, relu, res=None):
        """One BN-folded convolution on a channel-last bf16 activation x (B, H, W, Cin) -> (B, H', W', Cout).
        1x1 convolutions are plain GEMMs over the B*H*W rows (hipBLASLt with the bias / ReLU epilogue, instead of
        MIOpen's split-K implicit GEMM + f32->bf16 cast pass); k x k ones go through MIOpen without bias and the
        epilogue `act(y + bias + res)` is ONE in-place HIP pass (cgg_bias_act_nhwc)."""
        import torch.nn.functional as F
        w4, b, w2 = folded
        if w2 is not None:
            sh, sw = conv.stride
            if sh > 1 or sw > 1:
                x = ops.subsample_nhwc(x, sh) if (sh == sw and x.is_cuda) else x[:, ::sh, ::sw, :].contiguous()
            x2 = x.reshape(-1, x.shape[-1])
            r2 = res.reshape(-1, res.shape[-1]) if res is not None else None
            if r2 is not None and x2.is_contiguous() and w2.is_contiguous() and r2.is_contiguous():
                # relu(x W^T + b + identity) as one hipBLASLt call (residual via beta = 1)
                y = ops.gemm_bias_res_act_bf16(x2, w2, b, r2, relu)
                return y.view(x.shape[0], x.shape[1], x.shape[2], -1)
            if relu and res is None:
                y = torch._addmm_activation(b, x2, w2.t())
            else:
                y = torch.addmm(b, x2, w2.t())
            y = y.view(x.shape[0], x.shape[1], x.shape[2], -1)
            if res is not None:
                ops.bias_act_nhwc_(y, None, res, relu)
            return y
        B, H, W, C = x.shape
        st = conv.stride[0]
        if (tuple(conv.kernel_size) == (3, 3) and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1)
                and conv.groups == 1 and conv.stride[0] == conv.stride[1] and st in (1, 2) and C % 8 == 0 and res is None
                and B * ((H - 1) // st + 1) * ((W - 1) // st + 1) <= ResNet.IM2COL_MAX_PIXELS and x.is_contiguous()):
            # deep stages: too few output tiles for the implicit-GEMM convolution kernels -> explicit patch matrix +
            # one library GEMM with the bias / ReLU epilogue (18-20 us instead of 45-54 + 5 us at configs[1])
            wk = runtime.derived_cached('conv3x3_as_gemm', (w4,),
                                        lambda: w4.permute(0, 2, 3, 1).reshape(w4.shape[0], -1).contiguous())
            cols, Ho, Wo = ops.im2col3x3_nhwc(x, st)
            return ops.gemm_bias_res_act_bf16(cols, wk, b, None, relu).view(B, Ho, Wo, -1)
        y = F.conv2d(x.permute(0, 3, 1, 2), w4, None, stride=conv.stride, padding=conv.padding,
                     dilation=conv.dilation, groups=conv.groups)
        y = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        return ops.bias_act_nhwc_(y, b, res, relu)

    def _forward_folded(self, x, upto=None):
        import torch.nn.functional as F
        seq = iter(self._folded(upto))
        mp = self.maxpool
        pool_ok = (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False)
        c1 = self.conv1
        if (pool_ok and x.dtype == torch.float32 and x.is_contiguous() and tuple(c1.weight.shape) == (64, 3, 7, 7)
                and tuple(c1.stride) == (2, 2) and tuple(c1.padding) == (3, 3) and tuple(c1.dilation) == (1, 1)):
            # stem: hand-written MFMA convolution straight from the f32 NCHW image + (bias, ReLU, max-pool) pass
            w4, b, _ = next(seq)
            packed = runtime.derived_cached('stem_packed', (w4,), lambda: ops.pack_stem_weight(w4))
            x = ops.bias_relu_maxpool_nhwc(ops.stem_conv7x7(x, packed), b)
            return self._forward_folded_layers(x, seq, upto)
        x = x.to(dtype=torch.bfloat16, memory_format=torch.channels_last).permute(0, 2, 3, 1)
        if pool_ok:
            # conv -> (+bias, ReLU, 3x3/s2 max-pool) in one HIP pass over the raw convolution output
            w4, b, _ = next(seq)
            y = F.conv2d(x.permute(0, 3, 1, 2), w4, None, stride=self.conv1.stride, padding=self.conv1.padding)
            y = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
            x = ops.bias_relu_maxpool_nhwc(y, b)
        else:
            x = self._conv_nhwc(x, self.conv1, next(seq), True)
            x = mp(x.permute(0, 3, 1, 2)).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        return self._forward_folded_layers(x, seq, upto)

    def _forward_folded_layers(self, x, seq, upto=None):
        """`upto` = number of leading res layers to run (training: the frozen stages): returns (outs so far, x) instead."""
        outs = []
        for i, name in enumerate(self.res_layers):
            if upto is not None and i >= upto:
                return outs, x
            for blk in getattr(self, name):
                identity = x
                if blk.downsample is not None:
                    identity = self._conv_nhwc(x, blk.downsample[0], next(seq), False)
                if (FUSED_BOTTLENECK and isinstance(blk, Bottleneck) and blk.downsample is None and blk.conv1.in_channels == 256
                        and blk.conv1.out_channels == 64 and tuple(blk.conv2.stride) == (1, 1) and blk.conv2.groups == 1
                        and tuple(blk.conv2.dilation) == (1, 1) and tuple(blk.conv2.padding) == (1, 1) and ops.bottleneck64_ok(x)):
                    # layer1 identity block: conv1 -> conv2 -> conv3 + residual in ONE launch (the 64-channel intermediates stay in LDS)
                    f1, f2, f3 = next(seq), next(seq), next(seq)
                    packed = runtime.derived_cached('bottleneck64', (f1[2], f2[0], f3[2], f1[1], f2[1], f3[1]),
                                                    lambda: ops.pack_bottleneck64(f1[2], f1[1], f2[0], f2[1], f3[2], f3[1]))
                    x = ops.bottleneck64(x, packed)
                    continue
                y = self._conv_nhwc(x, blk.conv1, next(seq), True)
                if isinstance(blk, Bottleneck):
                    y = self._conv_nhwc(y, blk.conv2, next(seq), True)
                    x = self._conv_nhwc(y, blk.conv3, next(seq), True, identity)
                else:
                    x = self._conv_nhwc(y, blk.conv2, next(seq), True, identity)
            if i in self.out_indices:
                outs.append(x)
        if upto is not None:
            return outs, x
        # (B, C, H, W)-shaped views of the channel-last bf16 activations: no copy, no cast. The pixel decoder's
        # inference stream consumes them as they are; anything else can `.float().contiguous()` them.
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    # ---- parity-mode inference: BN folded in f32, channel-last f32 activations, every convolution after the stem on the
    #      f32-class implicit-GEMM kernel (ops.conv_x3_nhwc: f16 x 3 MFMA, csrc/x3_gemm.hip) with bias / residual / ReLU fused ----
    def _folded_x3(self, upto=None):
        """[(x3 image | folded f32 filter for the stem, bias f32)] per conv in execution order, for the stem and the first `upto`
        stages (all when None). Cached PER STAGE against the versions of that stage's parameters / buffers: under training with
        frozen stages only the trainable tail changes, and the frozen prefix is never re-folded."""
        def fold(conv, bn, stem=False):
            s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            w = conv.weight.detach().float() * s.view(-1, 1, 1, 1)
            b = (bn.bias.detach().float() - bn.running_mean.detach().float() * s).contiguous()
            return (w.contiguous() if stem else ops.pack_conv_weight_x3(w)), b

        def stage_seq(name):
            if name == 'stem':
                return [fold(self.conv1, self.bn1, stem=True)]
            seq = []
            for blk in getattr(self, name):
                if blk.downsample is not None:
                    seq.append(fold(blk.downsample[0], blk.downsample[1]))
                seq.append(fold(blk.conv1, blk.bn1))
                seq.append(fold(blk.conv2, blk.bn2))
                if isinstance(blk, Bottleneck):
                    seq.append(fold(blk.conv3, blk.bn3))
            return seq

        cache = self.__dict__.setdefault('_fold_cache_x3', {})
        names = ['stem'] + list(self.res_layers if upto is None else self.res_layers[:upto])
        out = []
        for name in names:
            mods = [self.conv1, self.bn1] if name == 'stem' else [getattr(self, name)]
            tensors = [t for m in mods for t in list(m.parameters()) + list(m.buffers())]
            key = (sum(t._version for t in tensors), str(self.conv1.weight.device))
            hit = cache.get(name)
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    hit = (key, stage_seq(name))
                cache[name] = hit
            out.extend(hit[1])
        return out

    def _x3_ok(self):
        def conv_ok(c):
            return (c.groups == 1 and tuple(c.dilation) == (1, 1) and c.kernel_size[0] == c.kernel_size[1]
                    and c.stride[0] == c.stride[1] and c.padding[0] == c.padding[1] and c.in_channels % 32 == 0 and c.bias is None
                    and c.out_channels % 8 == 0 and c.kernel_size[0] * c.kernel_size[1] <= 32)
        for name in self.res_layers:
            for blk in getattr(self, name):
                convs = [blk.conv1, blk.conv2] + ([blk.conv3] if isinstance(blk, Bottleneck) else [])
                if blk.downsample is not None:
                    if not (isinstance(blk.downsample[0], nn.Conv2d) and isinstance(blk.downsample[1], nn.BatchNorm2d)):
                        return False
                    convs.append(blk.downsample[0])
                if not all(conv_ok(c) for c in convs):
                    return False
        return self.conv1.bias is None

    def _forward_x3(self, x, upto=None):
        """`upto` = number of leading stages to run (frozen-stage prefix under training): returns (outs, x) channel-last."""
        import torch.nn.functional as F
        x3a = runtime.x3a_enabled()
        seq = iter(self._folded_x3(upto))
        w, b = next(seq)
        mp, c1 = self.maxpool, self.conv1
        if ((mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False) and x.dtype == torch.float32
                and x.is_contiguous() and tuple(c1.weight.shape) == (64, 3, 7, 7) and tuple(c1.stride) == (2, 2)
                and tuple(c1.padding) == (3, 3) and tuple(c1.dilation) == (1, 1) and X3_STEM):
            # stem: the MFMA convolution straight from the f32 NCHW image on the f32-class contraction + (bias, ReLU, max-pool) pass
            pk, sc = runtime.derived_cached('stem_packed_x3', (w,), lambda: ops.pack_stem_weight_x3(w))
            x = ops.bias_relu_maxpool_nhwc_f32(ops.stem_conv7x7_x3(x, pk, sc), b, x3a=x3a)
        else:
            # (other stems: 3 input channels are not an implicit-GEMM shape) MIOpen f32 with the folded filter, then channel-last
            x = F.conv2d(x.float(), w, b, stride=c1.stride, padding=c1.padding)
            x = mp(torch.relu_(x)).permute(0, 2, 3, 1).contiguous()
            if x3a:
                x = ops.x3a_encode(x)

        def conv(x, c, relu, res=None):
            wk, bias = next(seq)
            if x3a:     # x3a rows in, x3a rows out (residual too): the LDS-DMA implicit GEMM (csrc/x3s_gemm.hip)
                return ops.conv_x3s_nhwc(x, wk, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], bias, res=res, relu=relu)
            return ops.conv_x3_nhwc(x, wk, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], bias, res=res, relu=relu)

        outs = []
        for i, name in enumerate(self.res_layers):
            if upto is not None and i >= upto:
                break
            for blk in getattr(self, name):
                identity = x if blk.downsample is None else conv(x, blk.downsample[0], False)
                y = conv(x, blk.conv1, True)
                if isinstance(blk, Bottleneck):
                    y = conv(y, blk.conv2, True)
                    x = conv(y, blk.conv3, True, identity)
                else:
                    x = conv(y, blk.conv2, True, identity)
            if i in self.out_indices:
                outs.append(x)
        if upto is not None:
            # frozen prefix of a training step: plain f32 channel-last maps for the autograd part that follows
            dec = (lambda t: ops.x3a_decode(t)) if x3a else (lambda t: t)
            return [dec(o) for o in outs], dec(x)
        # (B, C, H, W)-shaped views of the channel-last f32 activations (no copy): the pixel decoder's parity-mode stream reads
        # them as they are; anything else can `.contiguous()` them. Round 4: the maps are x3a rows, tagged `ops.X3ATensor` -- the
        # pixel decoder's x3 GEMMs consume them as stored, `ops.x3a_to_f32` gives the values
        if x3a and self.x3a_outputs:
            return tuple(ops.as_x3a(o.permute(0, 3, 1, 2)) for o in outs)
        if x3a:
            # module boundary: a caller that has not opted in gets plain float32 values -- x3a storage holds f16 pairs,
            # arithmetic on the raw tensor would be garbage without an error
            return tuple(ops.x3a_decode(o).permute(0, 3, 1, 2) for o in outs)
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    def forward(self, x):
        frozen_bn = all(not m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d))
        if runtime.is_bf16() and not torch.is_grad_enabled() and frozen_bn and x.is_cuda:
            return self._forward_folded(x)
        if runtime.x3_enabled() and not torch.is_grad_enabled() and frozen_bn and x.is_cuda:
            if self._x3_ok():
                return self._forward_x3(x)
            runtime.note_fallback('ResNet', 'a convolution outside the implicit-GEMM rules (groups, dilation, C % 32, N % 8): MIOpen f32')
        outs = []
        frozen_nhwc = []
        first = 0
        xn = None
        if (FROZEN_FOLDED and runtime.x3_enabled() and torch.is_grad_enabled() and frozen_bn and x.is_cuda and self.frozen_stages >= 1
                and not x.requires_grad and self._x3_ok() and x.dtype == torch.float32
                and not any(p.requires_grad for n in self.res_layers[:self.frozen_stages] for p in getattr(self, n).parameters())
                and not any(p.requires_grad for p in list(self.conv1.parameters()) + list(self.bn1.parameters()))):
            # parity-mode training with frozen stages: the frozen prefix on the BN-folded f32-class x3 inference path (implicit-GEMM
            # kernels instead of MIOpen f32 conv + BN + ReLU launches); only the trainable tail goes through autograd
            with torch.no_grad():
                fouts, xf = self._forward_x3(x, upto=self.frozen_stages)
            outs = [o.permute(0, 3, 1, 2) for o in fouts]
            frozen_nhwc = list(fouts)
            x = xf.permute(0, 3, 1, 2)
            xn = xf                                          # the same map channel-last: what the x3 training stages read
            first = self.frozen_stages
        if (FROZEN_FOLDED and runtime.is_bf16() and frozen_bn and x.is_cuda and self.frozen_stages >= 1 and not x.requires_grad
                and not any(p.requires_grad for n in self.res_layers[:self.frozen_stages] for p in getattr(self, n).parameters())
                and not any(p.requires_grad for p in list(self.conv1.parameters()) + list(self.bn1.parameters()))):
            # training with frozen stages (configs: frozen_stages=3, norm_eval=True): nothing before the first trainable layer is
            # recorded by autograd anyway, so the stem and the frozen layers run on the BN-folded bf16 channel-last inference
            # path (GEMM / fused kernels instead of conv + BN + ReLU launches); only the trainable tail goes through autograd
            with torch.no_grad():
                fouts, xf = self._forward_folded(x, upto=self.frozen_stages)
            outs = [o.permute(0, 3, 1, 2) for o in fouts]
            frozen_nhwc = list(fouts)
            x = xf.permute(0, 3, 1, 2)                       # channels-last strided (B, C, H, W) bf16 view
            first = self.frozen_stages
        with runtime.autocast():
            if runtime.is_bf16():
                # MIOpen's bf16 solvers want NHWC activations AND NHWC filters (a mixed pair falls back
                # to the naive kernels): convert the filters once, then feed NHWC input
                if not getattr(self, '_channels_last', False):
                    self.to(memory_format=torch.channels_last)
                    self._channels_last = True
                x = x.contiguous(memory_format=torch.channels_last)
            if first == 0:
                x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
            for i, name in enumerate(self.res_layers):
                if i < first:
                    continue
                stage = getattr(self, name)
                if xn is not None and runtime.x3_resnet_stage_ok(stage, xn):
                    # parity-mode training: the trainable stage stays channel-last on the x3 kernels (one autograd node per
                    # convolution + frozen BatchNorm + ReLU / residual: `runtime._X3ConvBnFn`); x = None until somebody needs NCHW
                    xn = runtime.resnet_stage_x3_train(stage, xn)
                    x = None
                else:
                    if x is None:
                        x = runtime.nhwc_to_nchw_train(xn)
                    x = stage(x)
                    xn = None
                if i in self.out_indices:
                    if x is None:
                        x = runtime.nhwc_to_nchw_train(xn)
                    outs.append(x)
        res = []
        for k, o in enumerate(outs):
            src = frozen_nhwc[k] if k < len(frozen_nhwc) else None
            if (src is not None and (src.dtype == torch.float32 or (src.dtype == torch.bfloat16 and FROZEN_NHWC_BF16))
                    and src.is_cuda and src.is_contiguous() and not src.requires_grad):
                # a frozen stage's channel-last map (f32 in parity mode, bf16 in throughput mode: one contiguous cast pass first): NCHW
                # by the tiled transpose kernel (ATen's strided clone of the 1-GB stride-4 map took 1.1 ms, this 0.35), and the
                # channel-last f32 original rides along for consumers that read rows (`runtime.fpn_level_x3_train`)
                f = src if src.dtype == torch.float32 else src.float()
                res.append(runtime.hand_nhwc(ops.nhwc_to_nchw(f), f))
            else:
                res.append(o.float().contiguous())
        return tuple(res)
