"""-m gpu tests of parity mode's f32-class contractions (csrc/x3.h: two f16 pieces per f32 operand, three f16 MFMAs per
product, f32 accumulation) against float64 on the same inputs. The bar is "as accurate as an f32 GEMM": each check compares
the kernel's error with the error of the plain f32 torch product of the same operands (and states an absolute bound).
Reference arithmetic being matched: the f32 linears / einsum of open_set/models/mask2former_head.py:711-761, 829-840."""
import pytest
import torch

import cgg_amd  # noqa: F401
from cgg_amd import ops, runtime
from oracle import ops as ref

pytestmark = pytest.mark.gpu


def _err(got, want64):
    return (got.detach().cpu().double() - want64).abs().max().item()


@pytest.mark.parametrize('M,N,K,xs,ws', [(200, 256, 256, 1.0, 1.0), (37, 49, 256, 30.0, 1e-3), (200, 2048, 256, 1e-2, 1.0),
                                          (130, 1073, 256, 1.0, 40.0), (200, 256, 2048, 3.0, 1.0)])
def test_linear_rows_x3_is_f32_class(dev, M, N, K, xs, ws):
    """y = x W^T + b (+ ReLU columns, + residual) over operand scales from 1e-3 to 40: error vs float64 within 4x the f32
    GEMM's own error (+ 1 ulp), i.e. 2-3 orders of magnitude below the bf16 kernel."""
    g = torch.Generator().manual_seed(147)
    x = torch.randn(M, K, generator=g) * xs
    x[:, ::5] *= 1e-3                                   # mixed magnitudes inside a row
    w = torch.randn(N, K, generator=g) * ws / K**0.5
    w[::3] *= 1e-2                                      # rows of different scale (per-row pre-scaling)
    b = torch.randn(N, generator=g) * xs * ws
    res = torch.randn(M, N, generator=g)
    want = x.double() @ w.double().t() + b.double()
    f32_err = ((x @ w.t() + b).double() - want).abs().max().item()
    packed = ops.pack_linear_weight_x3(w.to(dev))
    assert ops.is_x3(packed)
    y = ops.linear_rows_bf16(x.to(dev), packed, N, b.to(dev))
    scale = want.abs().max().item()
    assert _err(y, want) <= 4 * f32_err + 2e-7 * scale, (_err(y, want), f32_err, scale)
    # ReLU on the first 32 columns + residual; strided output view
    out = torch.zeros(M, N + 5, device=dev)[:, :N]
    ops.linear_rows_bf16(x.to(dev), packed, N, b.to(dev), res=res.to(dev), relu_cols=32, out=out)
    w2 = want.clone()
    w2[:, :32] = w2[:, :32].relu()
    assert _err(out, w2 + res.double()) <= 4 * f32_err + 2e-7 * (scale + 4)
    if K >= 512:                                        # split-K planes: deterministic, sum = the product
        planes = ops.linear_rows_bf16(x.to(dev), packed, N, b.to(dev), res=res.to(dev), ksplit=8)
        assert _err(planes.sum(0), want + res.double()) <= 4 * f32_err + 4e-7 * (scale + 4)
        assert torch.equal(planes, ops.linear_rows_bf16(x.to(dev), packed, N, b.to(dev), res=res.to(dev), ksplit=8))
    if N <= 256:                                        # LayerNorm epilogue + `y + pos`
        ln = torch.nn.LayerNorm(N)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(N, generator=g))
            ln.bias.copy_(torch.randn(N, generator=g))
        pos = torch.randn(7, N, generator=g)
        yl, yp = ops.linear_rows_bf16(x.to(dev), packed, N, b.to(dev), res=res.to(dev),
                                      ln=(ln.weight.to(dev), ln.bias.to(dev), ln.eps), pos=pos.to(dev), want_pos=True)
        wl = torch.nn.functional.layer_norm(want + res.double(), (N,), ln.weight.double(), ln.bias.double(), ln.eps)
        assert _err(yl, wl) <= 2e-5
        assert torch.equal(yp.cpu(), yl.cpu() + pos[torch.arange(M) % 7])


def test_x3_image_is_refused_by_nothing_and_mixed_kinds_are(dev):
    w = torch.randn(256, 256).to(dev)
    with pytest.raises(ops.CggError):
        ops.decoder_ffn(torch.randn(8, 256, device=dev), ops.pack_linear_weight_x3(w), torch.zeros(256, device=dev),
                        ops.pack_linear_weight(w), torch.zeros(256, device=dev), 256)


def _decoder_operands(dev, seed):
    g = torch.Generator().manual_seed(seed)
    M, C, Q = 200, 256, 100
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev)
    return g, M, C, Q, r


def test_decoder_tail_x3_vs_float64(dev):
    """cgg_decoder_tail_x3: LN_a(sum planes), + pos, LN_b, 3-layer mask MLP, next query projection -- every output vs float64
    at f32-class accuracy (the bf16 twin is only good to ~5e-2 of the output scale)."""
    g, M, C, Q, r = _decoder_operands(dev, 168)
    nsum = 8
    planes = r(nsum, M, C, k=0.5)
    pos = r(Q, C)
    na = (r(C), r(C), 1e-5)
    nb = (r(C), r(C), 1e-5)
    ws = [r(C, C, k=1 / 16) for _ in range(4)]
    bs = [r(C) for _ in range(4)]
    pk = [ops.pack_linear_weight_x3(w) for w in ws]
    y, yp, me, qn = ops.decoder_tail(planes, na, pos, nb, (pk[0], bs[0], pk[1], bs[1], pk[2], bs[2]), (pk[3], bs[3]),
                                     want_pos=True)
    d = lambda t: t.detach().cpu().double()
    F = torch.nn.functional
    yr = F.layer_norm(d(planes).sum(0), (C,), d(na[0]), d(na[1]), 1e-5)
    zr = F.layer_norm(yr, (C,), d(nb[0]), d(nb[1]), 1e-5)
    mr = torch.relu(torch.relu(zr @ d(ws[0]).t() + d(bs[0])) @ d(ws[1]).t() + d(bs[1])) @ d(ws[2]).t() + d(bs[2])
    qr = (yr + d(pos).repeat(M // Q, 1)) @ d(ws[3]).t() + d(bs[3])
    assert _err(y, yr) <= 2e-5 and _err(yp, yr + d(pos).repeat(M // Q, 1)) <= 2e-5
    assert _err(me, mr) <= 2e-5 * (1 + mr.abs().max().item()), _err(me, mr)
    assert _err(qn, qr) <= 2e-5 * (1 + qr.abs().max().item()), _err(qn, qr)
    y2, _, me2, qn2 = ops.decoder_tail(planes, na, pos, nb, (pk[0], bs[0], pk[1], bs[1], pk[2], bs[2]), (pk[3], bs[3]))
    assert torch.equal(y, y2) and torch.equal(me, me2) and torch.equal(qn, qn2)
    # without the query projection
    y3, _, me3, qn3 = ops.decoder_tail(planes, na, pos, nb, (pk[0], bs[0], pk[1], bs[1], pk[2], bs[2]))
    assert qn3 is None and torch.equal(me3, me)


def test_decoder_mid_x3_vs_float64(dev):
    g, M, C, Q, r = _decoder_operands(dev, 170)
    core, res, pos = r(M, C), r(M, C), r(Q, C)
    norm = (r(C), r(C), 1e-5)
    wo, bo = r(C, C, k=1 / 16), r(C)
    wqkv, bqkv = r(3 * C, C, k=1 / 16), r(3 * C)
    pwo, pqkv = ops.pack_linear_weight_x3(wo), ops.pack_linear_weight_x3(wqkv)
    x1, q, kv = ops.decoder_mid(core, pwo, bo, res, norm, pos, (pqkv, bqkv))
    d = lambda t: t.detach().cpu().double()
    x1r = torch.nn.functional.layer_norm(d(core) @ d(wo).t() + d(bo) + d(res), (C,), d(norm[0]), d(norm[1]), 1e-5)
    xpr = x1r + d(pos).repeat(M // Q, 1)
    qr = xpr @ d(wqkv[:C]).t() + d(bqkv[:C])
    kr = xpr @ d(wqkv[C:2 * C]).t() + d(bqkv[C:2 * C])
    vr = x1r @ d(wqkv[2 * C:]).t() + d(bqkv[2 * C:])
    assert _err(x1, x1r) <= 2e-5
    assert _err(q, qr) <= 3e-5 and _err(kv[:, :C], kr) <= 3e-5 and _err(kv[:, C:], vr) <= 3e-5
    only = ops.decoder_mid(core, pwo, bo, res, norm)
    assert only[1] is None and torch.equal(only[0], x1)


def test_decoder_ffn_x3_vs_float64(dev):
    g, M, C, Q, r = _decoder_operands(dev, 171)
    F = 2048
    x = r(M, C)
    w1, b1 = r(F, C, k=1 / 16), r(F)
    w2, b2 = r(C, F, k=1 / 45), r(C)
    p1, p2 = ops.pack_linear_weight_x3(w1), ops.pack_linear_weight_x3(w2)
    planes = ops.decoder_ffn(x, p1, b1, p2, b2, F)
    assert planes.shape == (F // 256, M, C)
    d = lambda t: t.detach().cpu().double()
    want = d(x) + torch.relu(d(x) @ d(w1).t() + d(b1)) @ d(w2).t() + d(b2)
    assert _err(planes.sum(0), want) <= 2e-5 * (1 + want.abs().max().item())
    # the same planes as the two x3 launches it fuses (same operands, same per-plane K ranges)
    h = ops.linear_rows_bf16(x, p1, F, b1, relu_cols=F)
    two = ops.linear_rows_bf16(h, p2, C, b2, res=x, ksplit=8)
    assert torch.allclose(planes, two, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('B,Q,H,W,pool', [(2, 100, 32, 32, 1), (1, 100, 20, 28, 1), (2, 37, 16, 24, 2), (1, 128, 64, 64, 4)])
def test_mask_logits_x3_is_f32_class(dev, B, Q, H, W, pool):
    """einsum('bqc,bchw->bqhw') (mask2former_head.py:748) in parity mode: f16 x 3 on both operands; error vs float64 within 4x
    the f32 einsum's own error, bits == (float64 logit < 0) wherever |logit| > 1e-4."""
    g = torch.Generator().manual_seed(110)
    embed = torch.randn(B, Q, 256, generator=g) * 2
    feat = torch.randn(B, 256, H, W, generator=g) * 3
    with runtime.precision_scope('fp32'):
        packed = ops.pack_mask_feature(feat.to(dev), pool=pool, split=True)
        assert packed.f32 is None or not runtime.x3_enabled()
        got, bits = ops.mask_logits(embed.to(dev), packed, want_logits=True, want_bits=True)
    fp = feat.double()
    if pool > 1:
        o = pool // 2 - 1
        fp = ((fp[:, :, o::pool, o::pool] + fp[:, :, o::pool, o + 1::pool])
              + (fp[:, :, o + 1::pool, o::pool] + fp[:, :, o + 1::pool, o + 1::pool])) * 0.25
    want = ref.mask_logits(embed.double(), fp)
    f32_err = (ref.mask_logits(embed, fp.float()).double() - want).abs().max().item()
    err = _err(got, want)
    assert err <= 4 * f32_err + 1e-6 * want.abs().max().item(), (err, f32_err)
    assert err <= 1e-3
    ub = ops.unpack_bits(bits, packed.npix).cpu().view(B, Q, -1)
    w = want.view(B, Q, -1)
    sure = w.abs() > 1e-4
    assert torch.equal(ub[sure], (w < 0)[sure])


@pytest.mark.parametrize('M,N,K', [(43008, 256, 256), (4071, 288, 256), (300, 1024, 256), (2048, 200, 1024), (129, 50, 64),
                                   (1000, 2048, 512)])
def test_gemm_x3_vs_float64(dev, M, N, K):
    """cgg_gemm_x3 (the large parity-mode linear): error vs float64 within 4x the f32 GEMM's; ragged M / N tiles; bias,
    residual and ReLU epilogue; strided input rows and output view; bit-reproducible."""
    g = torch.Generator().manual_seed(300 + N)
    x = torch.randn(M, K, generator=g) * 2
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    want = x.double() @ w.double().t() + b.double()
    f32_err = ((x @ w.t() + b).double() - want).abs().max().item()
    pk = ops.pack_linear_weight_x3(w.to(dev))
    xs = torch.zeros(M, K + 4, device=dev)[:, :K]
    xs.copy_(x)
    y = ops.gemm_x3(xs, pk, N, b.to(dev))
    scale = want.abs().max().item()
    assert _err(y, want) <= 4 * f32_err + 2e-7 * scale, (_err(y, want), f32_err)
    out = torch.zeros(M, N + 3, device=dev)[:, :N]
    ops.gemm_x3(x.to(dev), pk, N, b.to(dev), res=res.to(dev), relu=True, out=out)
    assert _err(out, (want + res.double()).relu()) <= 4 * f32_err + 2e-7 * (scale + 4)
    assert torch.equal(y, ops.gemm_x3(xs, pk, N, b.to(dev)))
    nobias = ops.gemm_x3(x.to(dev), pk, N)
    assert _err(nobias, want - b.double()) <= 4 * f32_err + 2e-7 * scale


@pytest.mark.parametrize('B,H,W,C,N,k,stride', [(2, 64, 64, 64, 64, 3, 1), (1, 33, 47, 32, 96, 3, 2), (2, 32, 32, 256, 256, 1, 1),
                                                (1, 40, 24, 128, 200, 1, 2), (2, 16, 16, 512, 512, 3, 1)])
def test_conv_x3_nhwc_vs_float64(dev, B, H, W, C, N, k, stride):
    """cgg_conv_x3_nhwc (implicit GEMM over a channel-last f32 map) == F.conv2d in float64 on the NCHW view, incl. the zero
    padding at the borders, odd sizes, stride 2, bias + residual + ReLU."""
    g = torch.Generator().manual_seed(400 + C)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(N, C, k, k, generator=g) / (C * k * k)**0.5
    b = torch.randn(N, generator=g)
    pad = k // 2
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    f32_err = (torch.nn.functional.conv2d(x, w, b, stride=stride, padding=pad).double() - want).abs().max().item()
    pk = ops.pack_conv_weight_x3(w.to(dev))
    xl = x.permute(0, 2, 3, 1).contiguous().to(dev)
    y = ops.conv_x3_nhwc(xl, pk, N, k, stride, pad, b.to(dev))
    assert tuple(y.shape) == (B, want.shape[2], want.shape[3], N)
    assert _err(y.permute(0, 3, 1, 2), want) <= 4 * f32_err + 2e-7 * want.abs().max().item()
    res = torch.randn(*y.shape, generator=g)
    y2 = ops.conv_x3_nhwc(xl, pk, N, k, stride, pad, b.to(dev), res=res.to(dev), relu=True)
    w2 = (want + res.permute(0, 3, 1, 2).double()).relu()
    assert _err(y2.permute(0, 3, 1, 2), w2) <= 4 * f32_err + 2e-7 * (want.abs().max().item() + 4)


@pytest.mark.parametrize('M,N,FF', [(43008, 21504, 1024), (4071, 1357, 1024), (100, 50, 512), (64, 64, 256)])
def test_encoder_layer_tail_x3_vs_float64(dev, M, N, FF):
    """cgg_encoder_layer_tail_x3 (output_proj + LN + FFN + LN in one launch, f32-class arithmetic) vs float64 on the same f32
    inputs: 2e-5 of the unit-scale outputs (the bf16 twin: one bf16 ulp = 4e-3); ragged row counts; y + pos; reproducible."""
    g = torch.Generator().manual_seed(500 + FF)
    C = 256
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k)
    a, x, pos = r(M, C), r(M, C), r(N, C)
    wo, bo = r(C, C, k=1 / 16), r(C, k=0.1)
    w1, b1 = r(FF, C, k=1 / 16), r(FF, k=0.1)
    w2, b2 = r(C, FF, k=1 / 32), r(C, k=0.1)
    n0 = (1 + 0.1 * r(C), 0.1 * r(C), 1e-5)
    n1 = (1 + 0.1 * r(C), 0.1 * r(C), 1e-5)
    d = lambda t: t.double()
    F = torch.nn.functional
    x1 = F.layer_norm(d(x) + d(a) @ d(wo).t() + d(bo), (C,), d(n0[0]), d(n0[1]), 1e-5)
    want = F.layer_norm(x1 + torch.relu(x1 @ d(w1).t() + d(b1)) @ d(w2).t() + d(b2), (C,), d(n1[0]), d(n1[1]), 1e-5)
    t = lambda v: v.to(dev)
    pk = [ops.pack_linear_weight_x3(t(w)) for w in (wo, w1, w2)]
    y, yp = ops.encoder_layer_tail_x3(t(a), t(x), pk[0], t(bo), (t(n0[0]), t(n0[1]), 1e-5), pk[1], t(b1), pk[2], t(b2),
                                      (t(n1[0]), t(n1[1]), 1e-5), pos=t(pos), want_pos=True)
    assert _err(y, want) <= 2e-5, _err(y, want)
    assert _err(yp, want + d(pos)[torch.arange(M) % N]) <= 2e-5
    y2, none = ops.encoder_layer_tail_x3(t(a), t(x), pk[0], t(bo), (t(n0[0]), t(n0[1]), 1e-5), pk[1], t(b1), pk[2], t(b2),
                                         (t(n1[0]), t(n1[1]), 1e-5))
    assert none is None and torch.equal(y, y2)


def test_gemm_x3_split_outputs_and_periodic_residual(dev):
    """cgg_gemm_x3_ex: one launch = two column blocks with a per-token residual table (the MSDeformAttn value / offsets
    projections): == float64 of x W^T + table[row % R]."""
    g = torch.Generator().manual_seed(321)
    M, K, N, col2, R = 3 * 1357, 256, 544, 256, 1357
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 16
    table = torch.randn(R, N, generator=g)
    want = x.double() @ w.double().t() + table.double()[torch.arange(M) % R]
    y1, y2 = ops.gemm_x3_split(x.to(dev), ops.pack_linear_weight_x3(w.to(dev)), N, col2, res_table=table.to(dev))
    assert y1.shape == (M, col2) and y2.shape == (M, N - col2)
    assert _err(y1, want[:, :col2]) <= 2e-5 and _err(y2, want[:, col2:]) <= 2e-5


@pytest.mark.parametrize('B,H,W', [(2, 128, 160), (1, 70, 91)])
def test_stem_conv7x7_x3_and_f32_maxpool_vs_float64(dev, B, H, W):
    """parity mode's ResNet stem (mmdet ResNet.conv1 + bn1 folded + relu + maxpool): the x3 MFMA convolution straight from the
    f32 NCHW image + the f32 (bias, ReLU, max-pool) pass == float64 conv2d / relu / max_pool2d, incl. odd sizes (ragged tiles)."""
    g = torch.Generator().manual_seed(77)
    img = torch.randn(B, 3, H, W, generator=g) * 2
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    w[::5] *= 1e-2
    b = torch.randn(64, generator=g)
    F = torch.nn.functional
    raw = F.conv2d(img.double(), w.double(), None, stride=2, padding=3)
    want = F.max_pool2d(torch.relu(raw + b.double().view(1, -1, 1, 1)), 3, 2, 1)
    f32_err = (F.conv2d(img, w, None, stride=2, padding=3).double() - raw).abs().max().item()
    pk, sc = ops.pack_stem_weight_x3(w.to(dev))
    y = ops.stem_conv7x7_x3(img.to(dev), pk, sc)
    assert _err(y.permute(0, 3, 1, 2), raw) <= 4 * f32_err + 1e-6
    z = ops.bias_relu_maxpool_nhwc_f32(y, b.to(dev))
    assert tuple(z.shape) == (B, want.shape[2], want.shape[3], 64)
    assert _err(z.permute(0, 3, 1, 2), want) <= 4 * f32_err + 1e-6
