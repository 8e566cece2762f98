"""-m gpu tests of round 4's pre-split activation rows ("x3a", csrc/x3.h) and the LDS-DMA GEMM / implicit-GEMM convolution that
consumes them (csrc/x3s_gemm.hip), through the C ABI: against float64 on the same (decoded) operands at f32-GEMM accuracy, against
round 3's kernel (cgg_gemm_x3 on the f32 rows: same arithmetic, so the f32 outputs must agree to rounding of the epilogue), for
EVERY tile configuration (forced), ragged row / column counts, zero padding of the convolution by out-of-range LDS-DMA, f32 and
x3a residuals / outputs, the row-periodic residual, and the overflow flag. Reference arithmetic being matched: the f32 linears /
convolutions under open_set/models/mask2former_head.py:787, 829-840 ([3P] MSDeformAttnPixelDecoder, ResNet)."""
import math

import pytest
import torch
import torch.nn.functional as F

import cgg_amd  # noqa: F401
from cgg_amd import ops
from cgg_amd._lib import load

pytestmark = pytest.mark.gpu

CONFIGS = list(range(18))


def _err(got, want64):
    return (got.detach().cpu().double() - want64).abs().max().item()


def test_x3a_roundtrip_and_flag(dev):
    g = torch.Generator().manual_seed(401)
    x = torch.randn(64, 256, generator=g) * 3
    x[:, ::7] *= 1e-4
    x[0, :8] = torch.tensor([0.0, -0.0, 4093.0, -4093.0, 1e-7, -1e-7, 1.0, -1.0])
    xd = x.to(dev)
    ops.x3_overflow_check(dev)                            # clear
    e = ops.x3a_encode(xd)
    assert e.shape == xd.shape and e.dtype == torch.float32
    back = ops.x3a_decode(e).cpu()
    # 22 significant bits (two 11-bit pieces); pieces below f16's subnormal step (2^-24 of the pre-scaled value) are dropped
    tol = x.abs() * 2.0**-21 + 2.0**-24 / 16
    assert ((back - x).abs() <= tol).all(), (back - x).abs().max()
    assert not ops.x3_overflow_check(dev)
    # the raw words are [8 x f16 hi | 8 x f16 lo] of 16 x per 8 channels
    raw = e.cpu().view(torch.float16).view(64, 32, 16)
    hi, lo = raw[..., :8].reshape(64, 256), raw[..., 8:].reshape(64, 256)
    assert torch.equal(hi.float(), (x * 16).half().float())
    assert torch.equal(lo.float(), (x * 16 - hi.float()).half().float())
    # out of range: flagged (and cleared by the check)
    y = xd.clone()
    y[3, 17] = 5000.0
    ops.x3a_encode(y)
    assert ops.x3_overflow_check(dev)
    assert not ops.x3_overflow_check(dev)


def _operands(M, N, K, seed, xs=1.0, ws=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * xs
    x[:, ::5] *= 1e-3
    w = torch.randn(N, K, generator=g) * ws / K**0.5
    w[::3] *= 1e-2
    b = torch.randn(N, generator=g) * xs * ws
    return g, x, w, b


@pytest.mark.parametrize('cfg', CONFIGS)
@pytest.mark.parametrize('M,N,K', [(1000, 288, 256), (257, 64, 64), (300, 520, 96)])
def test_gemm_x3s_every_config_vs_float64_and_round3(dev, cfg, M, N, K):
    g, x, w, b = _operands(M, N, K, 410 + cfg)
    res = torch.randn(M, N, generator=g)
    xe = ops.x3a_encode(x.to(dev))
    xdec = ops.x3a_decode(xe).cpu()                      # what the GEMM multiplies (x to 22 bits)
    packed = ops.pack_linear_weight_x3(w.to(dev))
    want = xdec.double() @ w.double().t() + b.double()
    f32_err = ((xdec @ w.t() + b).double() - want).abs().max().item()
    scale = want.abs().max().item()
    y = ops.gemm_x3s(xe, packed, N, b.to(dev), cfg=cfg)
    assert _err(y, want) <= 4 * f32_err + 2e-7 * scale, (cfg, _err(y, want), f32_err)
    # round 3's kernel on the f32 rows does the same split in its loop: same accumulators; epilogues differ by power-of-two scaling
    old = ops.gemm_x3(x.to(dev), packed, N, b.to(dev))
    assert (y - old).abs().max().item() <= 1e-6 * scale
    # ReLU + f32 residual, x3a output into a strided view; then that output as an x3a residual of a second call
    out = torch.zeros(M, N + 8, device=dev)[:, :N]
    ops.gemm_x3s(xe, packed, N, b.to(dev), res=res.to(dev), relu=True, out=out, out_split=True, cfg=cfg)
    w2 = (want + res.double()).relu()
    got = ops.x3a_decode(out.contiguous())
    assert _err(got, w2) <= 4 * f32_err + 1e-6 * (scale + 4)
    y3 = ops.gemm_x3s(xe, packed, N, b.to(dev), res=out, res_split=True, cfg=cfg)
    assert _err(y3, want + got.cpu().double()) <= 4 * f32_err + 1e-6 * (scale + 4)
    # row-periodic f32 residual (the K / V projections' per-token table)
    tab = torch.randn(7, N, generator=g)
    y4 = ops.gemm_x3s(xe, packed, N, None, res=tab.to(dev), res_mod=7, cfg=cfg)
    assert _err(y4, want - b.double() + tab[torch.arange(M) % 7].double()) <= 4 * f32_err + 1e-6 * (scale + 4)
    assert not ops.x3_overflow_check(dev)


@pytest.mark.parametrize('cfg', CONFIGS)
@pytest.mark.parametrize('B,H,W,C,N,k,s', [(2, 19, 23, 64, 96, 3, 1), (1, 32, 32, 32, 64, 3, 2), (2, 17, 16, 96, 40, 1, 2),
                                           (1, 40, 24, 64, 256, 1, 1)])
def test_conv_x3s_every_config_vs_float64(dev, cfg, B, H, W, C, N, k, s):
    """Implicit GEMM incl. the zero padding (out-of-range LDS-DMA pieces must arrive as zeros), strides, ragged M."""
    g = torch.Generator().manual_seed(430 + cfg)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(N, C, k, k, generator=g) / (C * k * k)**0.5
    b = torch.randn(N, generator=g)
    xe = ops.x3a_encode(x.to(dev))
    xdec = ops.x3a_decode(xe).cpu()
    pad = k // 2
    want = F.conv2d(xdec.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=s, padding=pad).permute(0, 2, 3, 1)
    f32_err = (F.conv2d(xdec.permute(0, 3, 1, 2), w, b, stride=s, padding=pad).permute(0, 2, 3, 1).double() - want).abs().max().item()
    packed = ops.pack_conv_weight_x3(w.to(dev))
    # poison the LDS-visible neighbourhood: a previous launch with large values must not leak into the padding
    ops.conv_x3s_nhwc(ops.x3a_encode(torch.full_like(x, 1000.0).to(dev)), packed, N, k, s, pad, b.to(dev), out_split=False, cfg=cfg)
    y = ops.conv_x3s_nhwc(xe, packed, N, k, s, pad, b.to(dev), out_split=False, cfg=cfg)
    scale = want.abs().max().item()
    assert _err(y, want) <= 4 * f32_err + 2e-7 * scale, (cfg, _err(y, want), f32_err)
    old = ops.conv_x3_nhwc(x.to(dev), packed, N, k, s, pad, b.to(dev))
    assert (y - old).abs().max().item() <= 1e-6 * scale
    # residual + ReLU, all in x3a
    res = torch.randn(*want.shape, generator=g)
    rese = ops.x3a_encode(res.to(dev))
    ye = ops.conv_x3s_nhwc(xe, packed, N, k, s, pad, b.to(dev), res=rese, relu=True, cfg=cfg)
    w2 = (want + ops.x3a_decode(rese).cpu().double()).relu()
    assert _err(ops.x3a_decode(ye), w2) <= 4 * f32_err + 1e-6 * (scale + 4)
    assert not ops.x3_overflow_check(dev)


def test_gemm_x3s_overflow_raises_the_flag(dev):
    g, x, w, b = _operands(256, 64, 64, 470)
    xe = ops.x3a_encode(x.to(dev))
    packed = ops.pack_linear_weight_x3(w.to(dev))
    big = b.clone()
    big[5] = 6000.0
    ops.x3_overflow_check(dev)
    ops.gemm_x3s(xe, packed, 64, big.to(dev), out_split=False)          # f32 output: no stored x3a value, no flag
    assert not ops.x3_overflow_check(dev)
    ops.gemm_x3s(xe, packed, 64, big.to(dev), out_split=True)
    assert ops.x3_overflow_check(dev)


def test_gemm_x3s_large_k_and_default_config(dev):
    """K = 2304 (72 chunks through the stage ring), default tile choice, chunk counts that are not multiples of the ring."""
    for M, N, K in [(2048, 256, 2304), (777, 128, 1120), (512, 512, 32)]:
        g, x, w, b = _operands(M, N, K, 480)
        xe = ops.x3a_encode(x.to(dev))
        xdec = ops.x3a_decode(xe).cpu()
        packed = ops.pack_linear_weight_x3(w.to(dev))
        want = xdec.double() @ w.double().t() + b.double()
        f32_err = ((xdec @ w.t() + b).double() - want).abs().max().item()
        y = ops.gemm_x3s(xe, packed, N, b.to(dev))
        assert _err(y, want) <= 4 * f32_err + 2e-7 * want.abs().max().item(), (M, N, K)
        assert torch.equal(y, ops.gemm_x3s(xe, packed, N, b.to(dev)))   # bit-reproducible


def test_x3_training_linear_forward_and_gradients_vs_float64(dev):
    """runtime._X3LinearFn (parity-mode training: forward and grad-input on the x3 GEMM, weight gradient as a split-K f32 library
    GEMM) vs float64 autograd of F.linear at an encoder-stream row count: f32-GEMM accuracy for y, dx, dW, db."""
    from cgg_amd import runtime
    g = torch.Generator().manual_seed(490)
    M, K, N = 16384, 256, 288
    x = torch.randn(2, M // 2, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    gy = torch.randn(2, M // 2, N, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    yd = F.linear(xd, wd, bd)
    yd.backward(gy.double())
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    with runtime.precision_scope('fp32'):
        assert runtime.x3_train_linear_ok(xg, wg)
        y = runtime.linear(xg, wg, bg)
        assert y.grad_fn is not None and 'X3Linear' in type(y.grad_fn).__name__
        y.relu_()                                                        # callers apply activations in place: y must not be a view
        y2 = runtime.linear(xg, wg, bg)
        y2.backward(gy.to(dev))
    for got, want, name in ((y2, yd, 'y'), (xg.grad, xd.grad, 'dx'), (wg.grad, wd.grad, 'dW'), (bg.grad, bd.grad, 'db')):
        scale = want.abs().max().item()
        err = (got.detach().cpu().double() - want.detach()).abs().max().item()
        assert err <= 2e-5 * scale, (name, err, scale)


@pytest.mark.parametrize('B,S,K,N', [(2, 4096, 256, 512), (3, 1000, 256, 512), (2, 5000, 256, 96)])
def test_x3_training_linear_with_a_per_token_table_vs_float64(dev, B, S, K, N):
    """runtime._X3LinearTableFn (the query decoder's [K | V] projection in parity-mode training, query_decoder.project_kv): y[b] =
    x[b] W^T + table with the (S, N) table in the GEMM's row-periodic residual input (`cgg_gemm_x3_ex`, res_mod = S; S not a
    multiple of the tile height: a tile then spans two images) vs float64 autograd -- y, dx, dW and d table = sum over images."""
    from cgg_amd import runtime
    g = torch.Generator().manual_seed(B * S + N)
    x = torch.randn(B, S, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    t = torch.randn(S, N, generator=g)
    gy = torch.randn(B, S, N, generator=g) * 1e-3
    xd, wd, td = (v.double().requires_grad_(True) for v in (x, w, t))
    yd = F.linear(xd, wd) + td[None]
    yd.backward(gy.double())
    xg, wg, tg = (v.to(dev).requires_grad_(True) for v in (x, w, t))
    with runtime.precision_scope('fp32'):
        y = runtime._X3LinearTableFn.apply(xg, wg, tg)
        y.backward(gy.to(dev))
    for got, want, name in ((y, yd, 'y'), (xg.grad, xd.grad, 'dx'), (wg.grad, wd.grad, 'dW'), (tg.grad, td.grad, 'dtable')):
        scale = want.abs().max().item()
        err = (got.detach().cpu().double() - want.detach()).abs().max().item()
        assert err <= 2e-5 * scale, (name, err, scale)


def test_overflow_in_the_stream_is_reported_not_silent(dev):
    """A stored activation beyond the f16 x 3 range (|a| >= 4094) must not pass silently as inf / NaN masks: the
    ResNet's x3a producers raise the device flag, `ops.x3_overflow_check` reports it (tools/test.py turns it into an error)."""
    import warnings
    from cgg_amd import registry, runtime
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        bb = registry.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                                          norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'))
        bb.init_weights()
    bb = bb.to(dev).eval()
    img = torch.randn(1, 3, 128, 128, device=dev)
    with torch.no_grad(), runtime.precision_scope('fp32'):
        ops.x3_overflow_check(dev)
        # module boundary: without the consumer's opt-in the maps are plain float32 values ...
        plain_feats = bb(img)
        assert not any(ops.is_x3a(f) for f in plain_feats)
        bb.x3a_outputs = True                        # ... the detector switches the x3a hand-over on for its own head
        feats = bb(img)
        torch.cuda.synchronize()
        assert all(torch.equal(ops.x3a_to_f32(f), p) for f, p in zip(feats, plain_feats))
        assert ops.is_x3a(feats[0]) and not ops.x3_overflow_check(dev)
        assert all(torch.isfinite(ops.x3a_to_f32(f)).all() for f in feats)
        # a BN-folded scale that drives layer1's output past the range (fill_, not mul_: a block's last BN is zero-initialised)
        bb.layer1[0].bn3.weight.fill_(1e5)
        feats = bb(img)
        torch.cuda.synchronize()
        assert ops.x3_overflow_check(dev)


@pytest.mark.parametrize('B,Q,S', [(2, 100, 16384), (1, 100, 1050), (2, 37, 200), (1, 128, 4096)])
def test_masked_xattn_x3_vs_float64_and_f32_kernel(dev, B, Q, S, monkeypatch):
    """cgg_masked_xattn_forward_x3 (S^T = K Q^T and O^T = V^T P^T on the f16 x 3 contraction, V^T by ds_read_b64_tr_b16 transpose
    reads in accumulator key order) vs the float64 definition of the masked attention core (mask2former_head.py:829-840) and vs the
    f32-MFMA kernel: both within 2e-5 of the O(1) outputs (the f32 kernel's own error vs float64 is the yardstick); ragged key
    counts, an un-masked row, a single visible key, kv as a strided column slice of a wider projection."""
    g = torch.Generator().manual_seed(33 + S)
    E, H = 256, 8
    q = torch.randn(B, Q, E, generator=g)
    kv_wide = torch.randn(B, S, 3 * 2 * E, generator=g)                  # three layers' [K | V] side by side
    kv = kv_wide[..., 2 * E:4 * E]                                       # layer 1's slice: strided rows
    mask = torch.rand(B, Q, S, generator=g) < 0.6
    mask[0, 1] = False
    mask[0, 2] = True
    mask[0, 2, S - 1] = False
    d = lambda t: t.double()
    k64, v64 = d(kv[..., :E]), d(kv[..., E:])
    logits = torch.einsum('bqhd,bshd->bhqs', d(q).view(B, Q, H, 32), k64.reshape(B, S, H, 32)) / 32 ** 0.5
    logits = logits.masked_fill(mask[:, None], float('-inf'))
    want = torch.einsum('bhqs,bshd->bqhd', logits.softmax(-1), v64.reshape(B, S, H, 32)).reshape(B, Q, E)
    words = (S + 31) // 32
    padded = torch.zeros(B, Q, words * 32, dtype=torch.bool)
    padded[..., :S] = mask
    w = (padded.view(B, Q, words, 32).long() << torch.arange(32)).sum(-1)
    bits = w.where(w < 2 ** 31, w - 2 ** 32).to(torch.int32).to(dev)
    kvd = kv_wide.to(dev)[..., 2 * E:4 * E]
    assert not kvd.is_contiguous()
    ops.x3_overflow_check(dev, reset=True)
    got = ops.masked_xattn(q.to(dev), kvd, bits, H)
    monkeypatch.setattr(ops, 'XATTN_X3', False)
    f32 = ops.masked_xattn(q.to(dev), kvd, bits, H)
    e_x3, e_f32 = _err(got, want), _err(f32, want)
    assert e_x3 <= max(2e-5, 4 * e_f32), (e_x3, e_f32)
    assert not ops.x3_overflow_check(dev, reset=True)
    monkeypatch.setattr(ops, 'XATTN_X3', True)
    assert torch.equal(got, ops.masked_xattn(q.to(dev), kvd, bits, H))          # reproducible
    big = kvd.clone()
    big[0, 0, 5] = 5000.0
    ops.masked_xattn(q.to(dev), big, bits, H)
    assert ops.x3_overflow_check(dev, reset=True)


@pytest.mark.parametrize('M,N,K', [(43008, 256, 256), (8192, 288, 256), (5000, 1024, 256), (4099, 256, 1024), (300, 36, 68)])
def test_wgrad_x3_vs_float64(dev, M, N, K):
    """cgg_wgrad_x3 (dW = dy^T x through LDS transpose reads, x3 arithmetic, row ranges split over the grid) vs float64: the error of
    an f32 GEMM of the same operands is the yardstick (<= 4x + a rounding floor); ragged row counts (partial chunks, partial
    splits), column counts that are not multiples of the 128-wide tile, strided rows, reproducible (fixed-order sum)."""
    g = torch.Generator().manual_seed(M + N)
    dy_w = torch.randn(M, N + 8, generator=g) * 0.3
    x_w = torch.randn(M, K + 4, generator=g)
    dy, x = dy_w[:, :N], x_w[:, 4:]                      # strided rows (x starts 16 B into its row)
    want = dy.double().t() @ x.double()
    f32_err = ((dy.t() @ x).double() - want).abs().max().item()
    dyd, xd = dy_w.to(dev)[:, :N], x_w.to(dev)[:, 4:]
    got = ops.wgrad_x3(dyd, xd)
    assert tuple(got.shape) == (N, K)
    err = _err(got, want)
    assert err <= 4 * f32_err + 2e-7 * want.abs().max().item(), (err, f32_err)
    assert torch.equal(got, ops.wgrad_x3(dyd, xd))
    gw2, gb = ops.wgrad_x3(dyd, xd, want_bias=True)                 # bias gradient from the same pass
    assert torch.equal(gw2, got)
    wb = dy.double().sum(0)
    assert _err(gb, wb) <= 4 * (dy.sum(0).double() - wb).abs().max().item() + 1e-6 * dy.abs().double().sum(0).max().item()


@pytest.mark.parametrize('mag', [1.0, 1e-4, 1e-6, 1e-8, 1e-10])
def test_x3_training_linear_gradients_at_real_gradient_magnitudes(dev, mag):
    """grad_output behind a normalised loss is not unit scale. `runtime._X3LinearFn.backward` pre-scales it per
    tensor (ops.absmax -> 2^(9 - floor(log2 amax)), csrc/x3.h) instead of by the activations' fixed 2^4, so dx = dy W and
    dW = dy^T x keep f32-GEMM accuracy relative to float64 at EVERY magnitude (with 2^4: 1e-3 relative at 1e-6, 1e-1 at 1e-8).
    Rows of mixed magnitude inside the tensor (x 1 .. x 1e-3) as a normalised loss over many tokens produces."""
    from cgg_amd import runtime
    M, N, K = 8192, 256, 256
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    gy = torch.randn(M, N, generator=g) * mag * torch.logspace(0, -3, M).view(M, 1)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    (F.linear(x64, w64, b64) * gy.double()).sum().backward()
    x32, w32, b32 = (t.clone().requires_grad_(True) for t in (x, w, b))
    (F.linear(x32, w32, b32) * gy).sum().backward()
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max()).item()
    f32 = (rel(x32.grad, x64.grad), rel(w32.grad, w64.grad), rel(b32.grad, b64.grad))
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    with runtime.precision_scope('fp32'):
        y = runtime._X3LinearFn.apply(xd, wd, bd)
        y.backward(gy.to(dev))
    got = (rel(xd.grad, x64.grad), rel(wd.grad, w64.grad), rel(bd.grad, b64.grad))
    for e, r in zip(got, f32):
        assert e <= 4 * r + 1e-6, (mag, got, f32)
    # and the scale really is what carries it: the fixed 2^4 loses the small magnitudes (documents why the pass over dy is paid)
    if mag <= 1e-8:
        fixed = ops.wgrad_x3(gy.to(dev), xd.detach())
        assert rel(fixed, w64.grad) > 10 * got[1]


@pytest.mark.parametrize('mag', [1.0, 1e-6])
def test_x3_training_ffn_node_vs_float64(dev, mag):
    """runtime._X3FfnFn (round 5: the encoder FFN branch W2 relu(W1 x + b1) + b2 as one autograd node -- ReLU in the first GEMM's
    epilogue, its backward as a mask in the grad-input GEMM's epilogue, max |grad_hidden| from that epilogue for the next
    contractions' pre-scale) vs float64 autograd, at unit and at real gradient magnitudes; the f32 autograd of the same formulation
    is the yardstick."""
    from cgg_amd import runtime
    M, K, Fh = 8192, 256, 1024
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g)
    w1, b1 = torch.randn(Fh, K, generator=g) / K ** 0.5, torch.randn(Fh, generator=g) * 0.1
    w2, b2 = torch.randn(K, Fh, generator=g) / Fh ** 0.5, torch.randn(K, generator=g) * 0.1
    gy = torch.randn(M, K, generator=g) * mag

    def run(dt, device):
        ts = [t.detach().clone().to(device=device, dtype=dt).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = F.linear(torch.relu(F.linear(ts[0], ts[1], ts[2])), ts[3], ts[4])
        y.backward(gy.to(device=device, dtype=dt))
        return y, [t.grad for t in ts]
    y64, g64 = run(torch.float64, 'cpu')
    y32, g32 = run(torch.float32, 'cpu')
    ts = [t.detach().clone().to(dev).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    with runtime.precision_scope('fp32'):
        assert runtime.x3_train_ffn_ok(*ts)
        y = runtime.ffn_x3_train(*ts)
        y.backward(gy.to(dev))
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max()).item()
    assert rel(y.detach(), y64) <= 4 * rel(y32.detach(), y64) + 1e-6
    for got, want, f32 in zip([t.grad for t in ts], g64, g32):
        assert rel(got, want) <= 4 * rel(f32, want) + 2e-6, (mag, rel(got, want), rel(f32, want))


def test_absmax_strided_and_zero(dev):
    x = torch.randn(300, 72, device=dev)
    x[17, 40] = -123.5
    assert ops.absmax(x).item() == 123.5
    assert ops.absmax(x[:, 8:40]).item() == x[:, 8:40].abs().max().item()          # strided rows
    assert ops.absmax(torch.zeros(64, 64, device=dev)).item() == 0.0


def test_x3_image_shape_is_checked(dev):
    """An x3 image carries no (N, K): the GEMM / convolution wrappers compare its byte size with what the call's shape needs
     instead of letting the kernel read past it."""
    w = torch.randn(256, 512, device=dev)
    pk = ops.pack_linear_weight_x3(w)
    a = torch.randn(64, 256, device=dev)
    with pytest.raises(Exception, match='x3 image holds'):
        ops.gemm_x3(a, pk, 256)                                   # K = 256, the image was packed for K = 512
    with pytest.raises(Exception, match='x3 image holds'):
        ops.gemm_x3s(ops.x3a_encode(a), pk, 512)                  # N, K swapped
    x = ops.x3a_encode(torch.randn(1, 8, 8, 64, device=dev))
    with pytest.raises(Exception, match='x3 image holds'):
        ops.conv_x3s_nhwc(x, pk, 256, 3, 1, 1)


@pytest.mark.parametrize('gmag', [1.0, 1e-7])
@pytest.mark.parametrize('B,C,N,H,W', [(2, 64, 96, 20, 28), (1, 32, 32, 9, 5), (2, 256, 256, 32, 32)])
def test_x3_training_conv3x3_forward_and_gradients_vs_float64(dev, B, C, N, H, W, gmag):
    """runtime._X3Conv3x3Fn (parity-mode training: forward and grad-input on the x3 implicit GEMM, grad-weight as nine x3
    transpose-read contractions over the zero-padded channel-last maps) vs float64 autograd of F.conv2d: errors of the order of an
    f32 convolution's; NCHW in, NCHW-shaped (channel-last strided) out."""
    from cgg_amd import runtime
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(N, C, 3, 3, generator=g) / (3 * C ** 0.5)
    go = torch.randn(B, N, H, W, generator=g) * gmag          # 1e-7: a real gradient magnitude (per-tensor pre-scale, csrc/x3.h)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(go.double())
    x32, w32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y32 = F.conv2d(x32, w32, None, 1, 1)
    y32.backward(go)
    f = lambda a, b: (a.double() - b).abs().max().item()
    ref_err = (f(y32, y64), f(x32.grad, x64.grad), f(w32.grad, w64.grad))
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    with runtime.precision_scope('fp32'):
        y = runtime._X3Conv3x3Fn.apply(xd, wd)
        assert tuple(y.shape) == (B, N, H, W)
        y.backward(go.to(dev))
    got = (f(y.detach().cpu(), y64), f(xd.grad.cpu(), x64.grad), f(wd.grad.cpu(), w64.grad))
    scale = (y64.abs().max().item(), x64.grad.abs().max().item(), w64.grad.abs().max().item())
    for e, r, s in zip(got, ref_err, scale):
        assert e <= 4 * r + 3e-7 * s, (got, ref_err, scale)


@pytest.mark.parametrize('B,R,C', [(2, 256, 4096), (3, 70, 129), (1, 64, 64), (2, 5, 1000), (16, 256, 1024)])
def test_transpose_f32_is_exact(dev, B, R, C):
    """cgg_transpose_f32 (the NCHW <-> NHWC layout changes around the x3 training convolution): bit-exact, ragged tiles, the scalar
    edge path (R or C not a multiple of 4)."""
    x = torch.randn(B, R, C, device=dev)
    assert torch.equal(ops.transpose_f32(x), x.transpose(1, 2).contiguous())
    if C % 16 == 0:                                      # the (B, C, H, W) <-> (B, H, W, C) wrappers on the same data
        nchw = x.view(B, R, 16, C // 16)
        nhwc = ops.nchw_to_nhwc(nchw)
        assert torch.equal(nhwc, nchw.permute(0, 2, 3, 1).contiguous())
        assert torch.equal(ops.nhwc_to_nchw(nhwc), nchw)


@pytest.mark.gpu
@pytest.mark.parametrize('mag', [1.0, 1e-3, 1e-5, 3e-7, 100.0])
def test_in_kernel_split_is_the_stored_split_bit_for_bit(dev, mag):
    """The kernels that take f32 rows split them with mixed-precision fmas (v_fma_mixlo / mixhi_f16, x3.h `cgg_x3_split2_s`); the
    stored x3a form is made with convert / subtract / convert. Through an identity weight the GEMM returns hi + lo of every element
    exactly, so both must agree bit for bit -- including magnitudes whose low piece is an f16 subnormal -- and the per-tensor
    pre-scale form must equal its definition (s = 2^(9 - floor(log2 amax)), hi = f16(s a), lo = f16(s a - hi))."""
    g = torch.Generator().manual_seed(int(mag * 1e7) % 9973 + 1)
    a = (torch.randn(256, 64, generator=g) * mag).to(dev)
    pk = ops.pack_linear_weight_x3(torch.eye(64, device=dev))
    y = ops.gemm_x3(a, pk, 64)
    assert torch.equal(y, ops.x3a_decode(ops.x3a_encode(a)))
    amax = ops.absmax(a)
    ys = ops.gemm_x3(a, pk, 64, amax=amax)
    s = 2.0 ** (9 - math.floor(math.log2(float(a.abs().max()))))
    t = a.cpu().double() * s
    hi = t.float().half()
    lo = (t.float() - hi.float()).half()
    ref = ((hi.double() + lo.double()) / s).float()
    assert torch.equal(ys.cpu(), ref)


@pytest.mark.gpu
def test_encoder_block_nodes_match_the_unfused_nodes_and_float64(dev, monkeypatch):
    """PARITY-mode training of the pixel decoder's encoder layers on the two per-block autograd nodes (`runtime._X3MsdaBlockFn`,
    `_X3FfnBlockFn`: residuals and gradient fan-in in GEMM epilogues, LayerNorm of one tensor, max |dz| from the LayerNorm backward)
    against the previous node structure (linear / FFN / add-LayerNorm nodes, CGG_X3_LAYER_NODES=0): same outputs, and every parameter
    / input gradient within f32 reassociation of each other at real gradient magnitudes (the upstream gradient is 1e-5-scale)."""
    from cgg_amd import runtime, synthetic
    from cgg_amd.pixel_decoder import MSDeformAttnPixelDecoder
    cfg = dict(synthetic.model_config(num_things=8, num_stuff=0, num_unknown=0, num_queries=10, enc_layers=2)['panoptic_head']['pixel_decoder'])
    cfg.pop('type')
    torch.manual_seed(3)
    pd = MSDeformAttnPixelDecoder(in_channels=[32, 64, 96, 128], strides=[4, 8, 16, 32], feat_channels=256, out_channels=256, **cfg)
    pd.init_weights()
    with torch.no_grad():          # (the reference initialises the offset / logit weights to zero: the `src + pos` path would carry no gradient)
        for layer in pd.encoder.layers:
            a = layer.attentions[0]
            a.sampling_offsets.weight.normal_(0, 0.02)
            a.attention_weights.weight.normal_(0, 0.05)
    pd = pd.to(dev).train()
    B = 8
    g = torch.Generator().manual_seed(4)
    feats = [torch.randn(B, c, 128 // s, 128 // s, generator=g).to(dev) for c, s in zip((32, 64, 96, 128), (1, 2, 4, 8))]
    params = [p for n, p in pd.named_parameters() if n.startswith('encoder.') or n.startswith('level_encoding')]
    gm = (torch.randn(B, 256, 128, 128, generator=g) * 1e-5).to(dev)

    def run(nodes):
        monkeypatch.setattr(runtime, '_X3_LAYER_NODES', nodes)
        for p in pd.parameters():
            p.grad = None
        with runtime.precision_scope('fp32'):
            mf, outs = pd(feats)
        (mf * gm).sum().backward()
        return mf.detach().clone(), [p.grad.detach().clone() for p in params]

    calls = []
    real = runtime._X3MsdaBlockFn.apply
    monkeypatch.setattr(runtime._X3MsdaBlockFn, 'apply', staticmethod(lambda *a: (calls.append(1), real(*a))[1]))
    mf1, g1 = run(True)
    assert len(calls) == 2, 'the block nodes did not run'
    mf0, g0 = run(False)
    assert len(calls) == 2
    assert (mf1 - mf0).abs().max().item() <= 2e-5 * mf0.abs().max().item()
    for (n, _), a, b in zip([(n, p) for n, p in pd.named_parameters() if n.startswith('encoder.') or n.startswith('level_encoding')], g1, g0):
        ref = b.abs().max().item()
        assert ref > 0 and torch.isfinite(a).all(), n
        assert (a - b).abs().max().item() <= 2e-4 * ref, (n, (a - b).abs().max().item(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,H,W', [(1, 64, 8, 8), (2, 96, 5, 7), (3, 256, 32, 20)])
def test_nchw_to_nhwc_pad1_is_the_padded_permutation(dev, B, C, H, W):
    """`ops.nchw_to_nhwc_pad1` (tiled transpose straight into the padded channel-last map of the x3 training convolution) ==
    F.pad(x.permute(0, 2, 3, 1), one zero pixel around), exactly."""
    x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(B + H)).to(dev)
    got = ops.nchw_to_nhwc_pad1(x)
    want = F.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('relu,up', [(True, False), (False, True), (False, False)])
def test_group_norm_rows_forward_backward_vs_float64(dev, relu, up):
    """Channel-last f32 GroupNorm for training (`ops.GroupNormRowsFn`: forward with the fused ReLU / bilinear up-sample + add,
    backward `cgg_group_norm_nhwc_f32_backward` incl. the up-sample's transpose) against torch in float64 on NCHW copies."""
    B, C, H, W, G = 3, 64, 24, 20, 8
    g = torch.Generator().manual_seed(17 + relu + 2 * up)
    x = (torch.randn(B, H * W, C, generator=g) * 2 + 0.5).to(dev).requires_grad_()
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev).requires_grad_()
    beta = torch.randn(C, generator=g).to(dev).requires_grad_()
    lo = torch.randn(B, (H // 2) * (W // 2), C, generator=g).to(dev).requires_grad_() if up else None
    gy = torch.randn(B, H * W, C, generator=g).to(dev)
    y = ops.GroupNormRowsFn.apply(x, gamma, beta, G, 1e-5, relu, lo, (H, W), (H // 2, W // 2))
    y.backward(gy)
    xd = x.detach().double().view(B, H, W, C).permute(0, 3, 1, 2).requires_grad_()
    gd, bd = gamma.detach().double().requires_grad_(), beta.detach().double().requires_grad_()
    ref = F.group_norm(xd, G, gd, bd, 1e-5)
    if up:
        lod = lo.detach().double().view(B, H // 2, W // 2, C).permute(0, 3, 1, 2).requires_grad_()
        ref = ref + F.interpolate(lod, size=(H, W), mode='bilinear', align_corners=False)
    if relu:
        ref = torch.relu(ref)
    ref.backward(gy.double().view(B, H, W, C).permute(0, 3, 1, 2))
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(B, -1, C)      # noqa: E731
    assert (y.double() - rows(ref)).abs().max().item() <= 1e-5
    for got, want in [(x.grad, rows(xd.grad)), (gamma.grad, gd.grad), (beta.grad, bd.grad)] + ([(lo.grad, rows(lod.grad))] if up else []):
        assert (got.double() - want).abs().max().item() <= 2e-5 * (want.abs().max().item() + 1e-9), (relu, up)


@pytest.mark.gpu
def test_fpn_level_rows_path_vs_float64(dev):
    """PARITY-mode training of the pixel decoder's FPN level on channel-last rows (`runtime.fpn_level_x3_train`: x3 row GEMMs for the
    lateral / mask-feature 1 x 1 convolutions, `_X3FpnLevelFn` for GroupNorm + up-sample + 3 x 3 convolution + GroupNorm + ReLU)
    against the same composite in float64 torch ops (F.conv2d / group_norm / interpolate on NCHW): mask_feature and every gradient
    (the level's parameters, the backbone map, the coarser level) at a 1e-5-scale upstream gradient."""
    from cgg_amd import runtime, synthetic
    from cgg_amd.pixel_decoder import MSDeformAttnPixelDecoder
    cfg = dict(synthetic.model_config(num_things=8, num_stuff=0, num_unknown=0, num_queries=10, enc_layers=1)['panoptic_head']['pixel_decoder'])
    cfg.pop('type')
    torch.manual_seed(5)
    pd = MSDeformAttnPixelDecoder(in_channels=[32, 64, 96, 128], strides=[4, 8, 16, 32], feat_channels=256, out_channels=256, **cfg)
    pd.init_weights()
    pd = pd.to(dev).train()
    B, H, W = 6, 96, 128                           # (B H W >= 65 536: the x3 convolution's size rule)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 32, H, W, generator=g).to(dev).requires_grad_()
    lo = torch.randn(B, (H // 2) * (W // 2), 256, generator=g).to(dev).requires_grad_()
    gm = (torch.randn(B, 256, H, W, generator=g) * 1e-5).to(dev)
    lat, outc, mf = pd.lateral_convs[0], pd.output_convs[0], pd.mask_feature
    params = dict(lat_w=lat.conv.weight, gn1_w=lat.gn.weight, gn1_b=lat.gn.bias, conv_w=outc.conv.weight, gn2_w=outc.gn.weight,
                  gn2_b=outc.gn.bias, mf_w=mf.weight, mf_b=mf.bias)
    with runtime.precision_scope('fp32'):
        assert runtime.x3_fpn_level_ok(pd, x, (H // 2, W // 2))
        got = runtime.fpn_level_x3_train(pd, x, lo, (H // 2, W // 2))
    (got * gm).sum().backward()
    grads = {k: v.grad.detach().double() for k, v in params.items()}
    grads['x'], grads['lo'] = x.grad.double(), lo.grad.double()

    def composite(dt):
        d = {k: v.detach().to(dt).requires_grad_() for k, v in params.items()}
        xd, lod = x.detach().to(dt).requires_grad_(), lo.detach().to(dt).requires_grad_()
        t = F.group_norm(F.conv2d(xd, d['lat_w']), 32, d['gn1_w'], d['gn1_b'], lat.gn.eps)
        t = t + F.interpolate(lod.view(B, H // 2, W // 2, 256).permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=False)
        t = torch.relu(F.group_norm(F.conv2d(t, d['conv_w'], padding=1), 32, d['gn2_w'], d['gn2_b'], outc.gn.eps))
        out = F.conv2d(t, d['mf_w'], d['mf_b'])
        (out * gm.to(dt)).sum().backward()
        gr = {k: v.grad.double() for k, v in d.items()}
        gr['x'], gr['lo'] = xd.grad.double(), lod.grad.double()
        return out.detach().double(), gr

    ref, want = composite(torch.float64)
    _, lib32 = composite(torch.float32)                    # the same composite on torch's f32 library kernels: the yardstick
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    for k in want:
        scale = want[k].abs().max().item()
        err = (grads[k] - want[k]).abs().max().item()
        err32 = (lib32[k] - want[k]).abs().max().item()
        # gradients through two GroupNorms are cancelling sums over 73 728 pixels: f32 library kernels are 1e-4 .. 1e-3 of scale from
        # float64 on the weights; the channel-last path must be no worse than twice that (or 3e-5)
        assert scale > 0 and err <= max(3e-5 * scale, 2.0 * err32), (k, err, err32, scale)


@pytest.mark.gpu
@pytest.mark.parametrize('first_stride,s2_x3', [(2, False), (2, True), (1, False)])
def test_resnet_stage_rows_path_vs_float64(dev, first_stride, s2_x3, monkeypatch):
    """PARITY-mode training of a trainable ResNet stage with frozen BatchNorm on channel-last maps (`runtime.resnet_stage_x3_train`:
    one `_X3ConvBnFn` node per convolution + BN affine + ReLU / residual, the 3 x 3 / stride-2 convolution on the library) against the
    same Bottlenecks in float64 torch modules: the stage output and every gradient (all filters incl. the down-sample path, the
    input map) at a 1e-5-scale upstream gradient, with the f32 library modules as the yardstick."""
    import copy
    from cgg_amd import runtime
    from cgg_amd.backbones import Bottleneck
    # (s2_x3: the 3 x 3 / stride-2 convolution on the x3 node as well -- CGG_X3_RESNET_S2=1, off by default because it is slower)
    monkeypatch.setattr(runtime, '_X3_RESNET_S2', s2_x3)
    torch.manual_seed(11)
    cin, planes = 64, 32
    down = torch.nn.Sequential(torch.nn.Conv2d(cin, planes * 4, 1, stride=first_stride, bias=False), torch.nn.BatchNorm2d(planes * 4))
    stage = torch.nn.Sequential(Bottleneck(cin, planes, first_stride, down), Bottleneck(planes * 4, planes), Bottleneck(planes * 4, planes))
    g = torch.Generator().manual_seed(12)
    for m in stage.modules():
        if isinstance(m, torch.nn.BatchNorm2d):      # a "trained" frozen BatchNorm: non-trivial statistics and affine
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.weight.requires_grad = m.bias.requires_grad = False
    stage = stage.to(dev).eval()                     # norm_eval=True: BatchNorm in eval mode while the filters train
    B, H, W = 8, 64, 64                              # (8 192 rows behind the stride-2 block: the x3 nodes' size rule)
    x = torch.randn(B, H, W, cin, generator=g).to(dev).requires_grad_()
    OH, OW = (H - 1) // first_stride + 1, (W - 1) // first_stride + 1
    gm = (torch.randn(B, OH, OW, planes * 4, generator=g) * 1e-5).to(dev)
    masks = []                                        # the ReLU decisions of the x3 run, NCHW like the modules' maps
    with runtime.precision_scope('fp32'):
        assert runtime.x3_resnet_stage_ok(stage, x.detach())
        got = runtime.resnet_stage_x3_train(stage, x, tap=lambda y: masks.append((y.detach() > 0).permute(0, 3, 1, 2)))
    assert len(masks) == 9
    (got * gm).sum().backward()
    names = [n for n, p in stage.named_parameters() if p.requires_grad]
    grads = {n: p.grad.detach().double() for n, p in stage.named_parameters() if p.requires_grad}
    assert all(n.endswith('weight') and ('conv' in n or 'downsample.0' in n) for n in names) and len(names) == 10
    grads['x'] = x.grad.double()

    class PinnedReLU(torch.nn.Module):
        # an activation that is zero to rounding is a tie between the two arithmetics, and ONE flipped tie moves a weight gradient by
        # a whole row's contribution (~5e-4 of its scale here): the references take the x3 run's decisions
        def __init__(self):
            super().__init__()
            self.it = iter(masks)

        def forward(self, z):
            return z * next(self.it).to(z.dtype)

    def composite(dt):
        st = copy.deepcopy(stage).to(dt)
        relu = PinnedReLU()
        for blk in st:
            blk.relu = relu
        for p in st.parameters():
            p.grad = None
        xd = x.detach().to(dt).requires_grad_()
        out = st(xd.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        (out * gm.to(dt)).sum().backward()
        gr = {n: p.grad.double() for n, p in st.named_parameters() if p.requires_grad}
        gr['x'] = xd.grad.double()
        return out.detach().double(), gr

    ref, want = composite(torch.float64)
    _, lib32 = composite(torch.float32)
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    bad = []
    for k in want:
        scale = want[k].abs().max().item()
        err = (grads[k] - want[k]).abs().max().item()
        err32 = (lib32[k] - want[k]).abs().max().item()
        print(f'{k}: err {err / scale:.2e} of scale, f32 library {err32 / scale:.2e}')
        if not (scale > 0 and err <= max(3e-5 * scale, 2.0 * err32)):
            bad.append((k, err, err32, scale))
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize('handed', ['frozen_nhwc', 'trainable_rows'])
def test_encoder_input_level_rows_path_vs_float64(dev, handed):
    """PARITY-mode training of an encoder input level of the pixel decoder (`runtime.input_level_x3_train`: 1 x 1 convolution + bias
    as an x3 row GEMM on the backbone map's channel-last original, GroupNorm by the channel-last kernels) against the same ConvModule
    in float64 (F.conv2d + F.group_norm on NCHW): rows, filter / bias / GroupNorm gradients and -- for a trainable backbone stage's
    map -- the gradient into the map; a map without a channel-last original takes the module path (None)."""
    from cgg_amd import runtime
    from cgg_amd.pixel_decoder import ConvModule
    torch.manual_seed(21)
    Cin, C, B, H, W = 512, 256, 4, 48, 64
    cm = ConvModule(Cin, C, kernel_size=1, norm_cfg=dict(type="GN", num_groups=32), act_cfg=None, bias=True).to(dev).train()
    with torch.no_grad():
        cm.conv.bias.normal_(0, 0.1)
        cm.gn.weight.uniform_(0.5, 1.5)
        cm.gn.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(22)
    nhwc = torch.randn(B, H, W, Cin, generator=g).to(dev)
    gm = (torch.randn(B, H * W, C, generator=g) * 1e-5).to(dev)
    with runtime.precision_scope('fp32'):
        assert runtime.input_level_x3_train(cm, nhwc.permute(0, 3, 1, 2).contiguous()) is None      # nothing handed over
        if handed == 'frozen_nhwc':
            feat = runtime.hand_nhwc(nhwc.permute(0, 3, 1, 2).contiguous(), nhwc)
            leaf = None
        else:
            leaf = nhwc.clone().requires_grad_()
            feat = runtime.nhwc_to_nchw_train(leaf)
        got = runtime.input_level_x3_train(cm, feat)
    assert got is not None and tuple(got.shape) == (B, H * W, C)
    (got * gm).sum().backward()
    params = dict(w=cm.conv.weight, b=cm.conv.bias, gamma=cm.gn.weight, beta=cm.gn.bias)
    grads = {k: v.grad.detach().double() for k, v in params.items()}
    if leaf is not None:
        grads['x'] = leaf.grad.double()

    def composite(dt):
        d = {k: v.detach().to(dt).requires_grad_() for k, v in params.items()}
        xd = nhwc.detach().to(dt).requires_grad_()
        y = F.group_norm(F.conv2d(xd.permute(0, 3, 1, 2), d['w'], d['b']), 32, d['gamma'], d['beta'], cm.gn.eps)
        y = y.flatten(2).transpose(1, 2)
        (y * gm.to(dt)).sum().backward()
        gr = {k: v.grad.double() for k, v in d.items()}
        gr['x'] = xd.grad.double()
        return y.detach().double(), gr

    ref, want = composite(torch.float64)
    _, lib32 = composite(torch.float32)
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    for k in grads:
        scale = want[k].abs().max().item()
        err = (grads[k] - want[k]).abs().max().item()
        err32 = (lib32[k] - want[k]).abs().max().item()
        assert scale > 0 and err <= max(3e-5 * scale, 2.0 * err32), (k, err, err32, scale)
