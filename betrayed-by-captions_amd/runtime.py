"""Process-wide execution switches of the MI355X path.

precision:
  'fp32' -- parity mode (default). Inference (no autograd): every contraction in f32-class f16 x 3 arithmetic on the matrix
            cores (csrc/x3.h: two f16 pieces per f32 operand, three v_mfma_f32_32x32x16_f16 per product, f32 accumulate) on
            own kernels -- the LDS-DMA GEMM / implicit-GEMM convolution over pre-split "x3a" activation rows
            (csrc/x3s_gemm.hip), the one-launch encoder tail, the x3 query-decoder chains, the x3 mask-logit einsum; f32
            MSDeformAttn values / norms / softmax; f32-MFMA attention. Under autograd (training): f32 library GEMMs /
            convolutions (hipBLASLt / MIOpen) with the HIP kernels for attention, MSDeformAttn, losses.
            CGG_X3=0 restores round 2's f32-library parity path, CGG_X3A=0 round 3's f32-row x3 stream (A/B only).
  'bf16' -- throughput mode named by BASELINE.json's north_star: bf16 MFMA contractions
            (mask logits 1x bf16 MFMA, bf16 values for the MSDeformAttn gather, torch GEMMs/convs
            under bf16 autocast); softmax / normalisation / accumulation stay f32.
"""
import contextlib
import weakref

import torch

_STATE = {'precision': 'fp32'}


def set_precision(p):
    if p not in ('fp32', 'bf16'):
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {p!r}")
    if p == 'fp32' and _STATE['precision'] == 'bf16' and 'cudnn_benchmark' in _STATE:
        torch.backends.cudnn.benchmark = _STATE.pop('cudnn_benchmark')
    if p == 'bf16' and _STATE['precision'] != 'bf16':
        _STATE['cudnn_benchmark'] = torch.backends.cudnn.benchmark
    _STATE['precision'] = p
    if p == 'bf16':
        # throughput mode: let MIOpen benchmark its solvers per convolution shape (first call per shape) instead of
        # the immediate-mode heuristic, which picks split-K implicit GEMMs that are 2-4x slower on the 3x3
        # convolutions of the backbone / FPN (measured on MI355X: 88 -> 21 us for 128->128 @128x128, batch 2)
        torch.backends.cudnn.benchmark = True


def precision():
    return _STATE['precision']


def is_bf16():
    return _STATE['precision'] == 'bf16'


import os as _os

# read once at import (the hot path called os.environ.get per linear)
_X3 = _os.environ.get('CGG_X3', '1') != '0'
_X3A = _os.environ.get('CGG_X3A', '1') != '0'


def effective_cpu_count():
    """CPUs this process may actually burn: min(os.cpu_count(), the scheduler affinity, the cgroup CPU quota). On the GPU boxes
    `nproc` says 256 while the container's cgroup grants 16 CPUs (`cpu.max` = 1600000 100000): 64 busy helper threads are then
    throttled by the CFS for up to 100 ms at a time (measured round 4: the host RLE path ran at 45-70 images/s instead of 330)."""
    n = _os.cpu_count() or 1
    try:
        n = min(n, len(_os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                      # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()[:2]
            if q != 'max':
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:     # cgroup v1
                q = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def x3_enabled():
    """Parity mode on the f32-class f16 x 3 kernels (csrc/x3.h); CGG_X3=0 restores the round-2 parity path (f32 library
    GEMMs, f32-MFMA skinny linears) for A/B measurements."""
    return _STATE['precision'] == 'fp32' and _X3


def x3a_enabled():
    """Round 4: the parity-mode inference stream keeps its GEMM-consumed activations as pre-split x3a rows (csrc/x3.h) and runs
    the LDS-DMA GEMM / implicit-GEMM convolution of csrc/x3s_gemm.hip; CGG_X3A=0 restores round 3's f32 stream (cgg_gemm_x3) for
    A/B measurements."""
    return x3_enabled() and _X3A


@contextlib.contextmanager
def precision_scope(p):
    old = _STATE['precision']
    set_precision(p)
    try:
        yield
    finally:
        set_precision(old)


def autocast():
    """autocast context for the plain library GEMMs / convs (hipBLASLt / MIOpen)."""
    if is_bf16():
        return torch.autocast(device_type='cuda', dtype=torch.bfloat16)
    return contextlib.nullcontext()


_WCACHE = {}


def _wkey(p, dtype):
    return (p.data_ptr(), tuple(p.shape), tuple(p.stride()), dtype)


def cast_cached(p, dtype=torch.bfloat16):
    """bf16 copy of a parameter (or of a VIEW of one, e.g. `in_proj_weight[E:]`), re-made only when the parameter
    changes (inference: made once). Keyed by the slice's address / geometry and validated against a weak
    reference to the owning parameter -- `id()` of a temporary view is recycled by Python and must not be a key."""
    base = p._base if p._base is not None else p
    key = _wkey(p, dtype)
    hit = _WCACHE.get(key)
    if hit is not None and hit[0]() is base and hit[1] == base._version and hit[2].device == p.device:
        return hit[2]
    if len(_WCACHE) > 8192:
        for k in [k for k, v in _WCACHE.items() if v[0]() is None]:
            del _WCACHE[k]
    t = p.detach().to(dtype).contiguous()
    _WCACHE[key] = (weakref.ref(base), base._version, t)
    return t


def cast_cache_replace(p, t, dtype=torch.bfloat16):
    """Swap the cached low-precision copy of `p` for a re-laid-out one (e.g. channels_last conv filters)."""
    base = p._base if p._base is not None else p
    _WCACHE[_wkey(p, dtype)] = (weakref.ref(base), base._version, t)


class _SplitKLinearFn(torch.autograd.Function):
    """Training-time `F.linear` in bf16 for operands with VERY many rows (the encoder stream: B * 21 504 rows). Forward and
    grad-input are the library GEMMs autocast would run; the weight gradient dW = dY^T X has a tiny output (<= 1024 x 256) and a
    reduction over all rows -- the library picks a 64 x 64 tile without split-K and leaves 240 of 256 CUs idle (0.66 ms for
    45 GFLOP at configs[2]). Here the rows are cut into S slabs, ONE batched GEMM produces S partial gradients (S x 16+ tiles)
    and their f32 sum is the gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        import torch.nn.functional as F
        x16 = x.to(torch.bfloat16)
        w16 = weight.to(torch.bfloat16)
        y = F.linear(x16, w16, bias.to(torch.bfloat16) if bias is not None else None)
        ctx.save_for_backward(x16, w16)
        ctx.x_dtype = x.dtype
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x16, w16 = ctx.saved_tensors
        N, K = w16.shape
        g2 = gy.reshape(-1, N).to(torch.bfloat16)
        x2 = x16.reshape(-1, K)
        M = x2.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (g2 @ w16).view(x16.shape).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            S = next((s for s in (32, 16, 8, 4, 2) if M % s == 0 and M // s >= 4096), 1)
            gw = torch.bmm(g2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K)).sum(0, dtype=torch.float32)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0, dtype=torch.float32)
        return gx, gw, gb


class _X3LinearFn(torch.autograd.Function):
    """Training-time `F.linear` in PARITY mode (the reference trains in f32, open_set/apis/train.py:182-189) for the encoder
    stream's row counts: forward y = x W^T + b and grad-input dx = dy W on the f32-class x3 GEMM (`ops.gemm_x3`, three f16 MFMAs
    per product, as accurate as an f32 GEMM -- tests/test_x3_gpu.py); the x3 images of W and W^T are re-packed when the optimiser
    changes the weight (cached against its version). The weight gradient dW = dy^T x reduces over ALL rows into a tiny output:
    `ops.wgrad_x3` (csrc/wgrad_x3.hip: both row-major operands through `ds_read_b64_tr_b16` transpose reads, row ranges split over
    the grid, partial tiles summed in fixed order); CGG_X3_WGRAD=0 = one batched f32 library GEMM over S row slabs (round 3)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import ops
        N, K = weight.shape
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        wk = derived_cached('x3_image', (weight,), lambda: ops.pack_linear_weight_x3(weight))
        y = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)    # (not a view: callers apply relu_ in place)
        ops.gemm_x3(x2, wk, N, bias.detach() if bias is not None else None, out=y.view(-1, N))
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        g2 = gy.reshape(-1, N)
        if g2.stride(1) != 1 or g2.stride(0) % 4 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        M = x2.shape[0]
        gx = gw = gb = None
        # grad_output is not unit scale (|g| ~ 1e-4 .. 1e-8 behind a normalised loss): its f16 pieces are taken after a per-tensor
        # power-of-two pre-scale from max |g| (ONE exact streaming pass, shared by the grad-input GEMM and the weight-gradient kernel;
        # a SAMPLED maximum was tried and is unsafe: encoder gradients are sparse -- most rows ~0, a few boundary rows 10^6 x larger --
        # so every 8th row can miss all of them and the true maximum then overflows f16: NaN weight gradients at configs[2] shapes)
        # instead of the activations' fixed 2^4 (with 2^4 a gradient of 1e-6 kept ~10 of its 22 bits); CGG_X3_GSCALE=0
        # restores the fixed scale for A/B
        amax = ops.absmax(g2) if _X3_GSCALE and N % 4 == 0 else None
        if ctx.needs_input_grad[0]:
            wtk = derived_cached('x3_image_t', (weight,), lambda: ops.pack_linear_weight_x3(weight.detach().t().contiguous()))
            gx = torch.empty(ctx.x_shape, dtype=torch.float32, device=g2.device)
            ops.gemm_x3(g2, wtk, K, out=gx.view(-1, K), amax=amax)
        if ctx.needs_input_grad[1]:
            if _X3_WGRAD and ctx.has_bias and ctx.needs_input_grad[2] and N % 4 == 0:
                gw, gb = ops.wgrad_x3(g2, x2, want_bias=True, amax=amax)      # the bias gradient from the same pass over grad_output
            elif _X3_WGRAD:
                gw = ops.wgrad_x3(g2, x2, amax=amax)   # transpose-read x3 kernel (csrc/wgrad_x3.hip), partial tiles summed in fixed order
            else:
                S = next((s for s in (32, 16, 8, 4, 2) if M % s == 0 and M // s >= 4096), 1)
                gw = torch.bmm(g2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K)).sum(0)
        if gb is None and ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


class _X3LinearTableFn(torch.autograd.Function):
    """y[b] = x[b] W^T + table for x (B, S, K), table (S, N) -- a linear layer whose bias differs per ROW but not per image (the
    query decoder's [K | V] projection of a memory level: `(mem + pos) Wk^T + b_k | mem Wv^T + b_v` with the position part folded into
    the table, query_decoder.project_kv) as one x3 node in PARITY-mode training: the table rides in the GEMM's row-periodic residual
    input instead of a broadcast add over the (B, S, N) result (1 GB per stride-8 level at configs[2]); backward as `_X3LinearFn`
    plus table.grad = sum over the images of grad_output."""

    @staticmethod
    def forward(ctx, x, weight, table):
        from . import ops
        B, S, K = x.shape
        N = weight.shape[0]
        x2 = _rows(x, K)
        wk = derived_cached('x3_image', (weight,), lambda: ops.pack_linear_weight_x3(weight))
        tb = table.detach().contiguous()
        y = torch.empty((B, S, N), dtype=torch.float32, device=x.device)
        ops.gemm_x3_table(x2, wk, N, tb, out=y.view(B * S, N))        # row m adds table[m % S]: one launch for all images
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        g2 = _rows(gy, N)
        gx = gw = gt = None
        amax = ops.absmax(g2) if _X3_GSCALE else None
        if ctx.needs_input_grad[0]:
            wtk = derived_cached('x3_image_t', (weight,), lambda: ops.pack_linear_weight_x3(weight.detach().t().contiguous()))
            gx = torch.empty(ctx.x_shape, dtype=torch.float32, device=g2.device)
            ops.gemm_x3(g2, wtk, K, out=gx.view(-1, K), amax=amax)
        if ctx.needs_input_grad[1]:
            gw = ops.wgrad_x3(g2, x2, amax=amax)
        if ctx.needs_input_grad[2]:
            gt = g2.view(ctx.x_shape[0], -1, N).sum(0)
        return gx, gw, gt


class _X3FfnFn(torch.autograd.Function):
    """Training-time FFN branch y = W2 relu(W1 x + b1) + b2 of an encoder layer ([3P] FFN behind mask2former_head.py:787) in PARITY
    mode as ONE autograd node on the x3 kernels. Compared with two `_X3LinearFn`s around `torch.relu_`: the ReLU is the first GEMM's
    epilogue (no clamp pass over the (rows, F) hidden tensor: 1.4 GB at configs[2]), its backward is the grad-input GEMM's epilogue
    (`ops.gemm_x3_bwd(mask=h)`: no threshold_backward pass), and that epilogue also reports max |grad_hidden| for the next two
    contractions' pre-scale (no absmax pass): -9 ms per step at configs[2]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        from . import ops
        F_, K = w1.shape
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        w1k = derived_cached('x3_image', (w1,), lambda: ops.pack_linear_weight_x3(w1))
        w2k = derived_cached('x3_image', (w2,), lambda: ops.pack_linear_weight_x3(w2))
        h = ops.gemm_x3(x2, w1k, F_, b1.detach(), relu=True)
        y = torch.empty((*x.shape[:-1], w2.shape[0]), dtype=torch.float32, device=x.device)
        ops.gemm_x3(h, w2k, w2.shape[0], b2.detach(), out=y.view(-1, w2.shape[0]))
        ctx.save_for_backward(x2, h, w1, w2)
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x2, h, w1, w2 = ctx.saved_tensors
        F_, K = w1.shape
        N = w2.shape[0]
        g2 = gy.reshape(-1, N)
        if g2.stride(1) != 1 or g2.stride(0) % 4 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        amax = ops.absmax(g2) if _X3_GSCALE else None
        gw2, gb2 = ops.wgrad_x3(g2, h, want_bias=True, amax=amax)
        w2t = derived_cached('x3_image_t', (w2,), lambda: ops.pack_linear_weight_x3(w2.detach().t().contiguous()))
        gh, amax_h = ops.gemm_x3_bwd(g2, w2t, F_, amax=amax, mask=h, want_amax=_X3_GSCALE)       # d / d(pre-activation)
        gw1, gb1 = ops.wgrad_x3(gh, x2, want_bias=True, amax=amax_h)
        gx = None
        if ctx.needs_input_grad[0]:
            w1t = derived_cached('x3_image_t', (w1,), lambda: ops.pack_linear_weight_x3(w1.detach().t().contiguous()))
            gx = torch.empty(ctx.x_shape, dtype=torch.float32, device=g2.device)
            ops.gemm_x3(gh, w1t, K, out=gx.view(-1, K), amax=amax_h)
        return gx, gw1, gb1, gw2, gb2


_X3_LAYER_NODES = _os.environ.get('CGG_X3_LAYER_NODES', '1') != '0'      # A/B: the per-block nodes below vs. linear / FFN / LayerNorm nodes


def _rows(x, K):
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    return x2


def _x3_img(w):
    from . import ops
    return derived_cached('x3_image', (w,), lambda: ops.pack_linear_weight_x3(w))


def _x3_img_t(w):
    from . import ops
    return derived_cached('x3_image_t', (w,), lambda: ops.pack_linear_weight_x3(w.detach().t().contiguous()))


class _X3FfnBlockFn(torch.autograd.Function):
    """y = LayerNorm(x + W2 relu(W1 x + b1) + b2): the FFN half of a post-norm encoder layer ([3P] BaseTransformerLayer behind
    open_set/models/mask2former_head.py:787) as ONE autograd node in PARITY-mode training. Against `_X3FfnFn` + `_AddLayerNormFn`:
    the residual is the second GEMM's epilogue (`res=`), so LayerNorm reads ONE tensor z forward and backward and only z is saved;
    the backward's two gradient paths into x (through the FFN and through the residual) meet in the last grad-input GEMM's epilogue
    instead of an autograd accumulation pass over (rows, 256); the LayerNorm backward kernel reports max |dz| for the pre-scale of the
    contractions behind it (no `absmax` pass). Per layer at configs[2]: one add pass, one absmax pass and 352 MB of LayerNorm
    backward reads less."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps):
        from . import ops
        F_, K = w1.shape
        x2 = _rows(x, K)
        h = ops.gemm_x3(x2, _x3_img(w1), F_, b1.detach(), relu=True)
        z = ops.gemm_x3(h, _x3_img(w2), K, b2.detach(), res=x2)
        y = ops.add_layernorm_stream(z, None, gamma.detach(), beta.detach(), eps, want_f32=True, want_bf16=False)[0]
        ctx.save_for_backward(x2, h, z, w1, w2, gamma)
        ctx.eps = eps
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x2, h, z, w1, w2, gamma = ctx.saved_tensors
        F_, K = w1.shape
        gz, _, dgamma, dbeta, amax = ops.add_layernorm_backward(_rows(gy, K), z, None, gamma, ctx.eps, want_amax=True)
        gw2, gb2 = ops.wgrad_x3(gz, h, want_bias=True, amax=amax)
        gh, amax_h = ops.gemm_x3_bwd(gz, _x3_img_t(w2), F_, amax=amax, mask=h, want_amax=True)      # d / d(pre-activation)
        gw1, gb1 = ops.wgrad_x3(gh, x2, want_bias=True, amax=amax_h)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.gemm_x3(gh, _x3_img_t(w1), K, res=gz, amax=amax_h).view(gy.shape)             # FFN path + residual path
        return gx, gw1, gb1, gw2, gb2, dgamma, dbeta, None


class _X3MsdaBlockFn(torch.autograd.Function):
    """y = LayerNorm(src + output_proj(MSDeformAttn(value_proj(src), [sampling_offsets | attention_weights](src + pos)))): the
    self-attention half of an encoder layer ([3P] MultiScaleDeformableAttention + norm) as ONE autograd node in PARITY-mode training.
    Forward: three x3 GEMMs (the residual in the output projection's epilogue), the fused-prologue sampling kernel, LayerNorm of one
    tensor. Backward, hand-ordered: LayerNorm backward (-> dz, max |dz|), output projection grad-input / grad-weight, the MSDeformAttn
    backward kernels, then the THREE gradient paths into src (residual, value_proj, offsets / logits) are summed by the epilogues of the
    two grad-input GEMMs (`res=`, the second one in place) -- autograd's accumulation passes and the (B, N, 256) gradient of
    `src + pos` never exist; pos' gradient is (sum over the batch of d rows) W, a (N, 288) reduction and a small GEMM."""

    @staticmethod
    def forward(ctx, src, pos, ref, wv, bv, w_off, b_off, w_att, b_att, wo, bo, gamma, beta, eps, num_heads, level_hw, level_start,
                num_points):
        from . import ops
        B, N, C = src.shape
        src2 = _rows(src, C)
        srcp = (src.detach() + pos.detach()[None]).view(-1, C)
        value = ops.gemm_x3(src2, _x3_img(wv), C, bv.detach())
        pk_cat, b_cat, n_cat = packed_cached((w_off, w_att), (b_off, b_att))
        rows = ops.gemm_x3(srcp, pk_cat, n_cat, b_cat)
        geom = (tuple(tuple(int(v) for v in hw) for hw in level_hw), tuple(int(s) for s in level_start), int(num_points))
        core = ops.msda_forward_fused(value.view(B, N, num_heads, C // num_heads), geom[0], geom[1], rows.view(B, N, n_cat), ref,
                                      geom[2])
        z = ops.gemm_x3(core.view(-1, C), _x3_img(wo), C, bo.detach(), res=src2)
        y = ops.add_layernorm_stream(z, None, gamma.detach(), beta.detach(), eps, want_f32=True, want_bf16=False)[0]
        ctx.save_for_backward(src2, srcp, value, rows, core, z, ref, wv, w_off, w_att, wo, gamma)
        ctx.geom, ctx.eps, ctx.heads, ctx.shape = geom, eps, num_heads, (B, N, C)
        return y.view(B, N, C)

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        src2, srcp, value, rows, core, z, ref, wv, w_off, w_att, wo, gamma = ctx.saved_tensors
        B, N, C = ctx.shape
        H = ctx.heads
        level_hw, level_start, P = ctx.geom
        n_off, n_cat = w_off.shape[0], w_off.shape[0] + w_att.shape[0]
        gz, _, dgamma, dbeta, amax = ops.add_layernorm_backward(_rows(gy, C), z, None, gamma, ctx.eps, want_amax=True)
        gwo, gbo = ops.wgrad_x3(gz, core.view(-1, C), want_bias=True, amax=amax)
        gcore = ops.gemm_x3(gz, _x3_img_t(wo), C, amax=amax)
        gv, grows = ops.msda_rows_backward(value.view(B, N, H, C // H), rows.view(B, N, n_cat), ref, level_hw, level_start, P,
                                           gcore.view(B, N, C))
        gv2, gr2 = gv.view(-1, C), grows.view(-1, n_cat)
        amax_v = ops.absmax(gv2)
        gwv, gbv = ops.wgrad_x3(gv2, src2, want_bias=True, amax=amax_v)
        gsrc = ops.gemm_x3(gv2, _x3_img_t(wv), C, res=gz, amax=amax_v)                     # residual + value_proj paths
        amax_r = ops.absmax(gr2)
        gw_cat, gb_cat = ops.wgrad_x3(gr2, srcp, want_bias=True, amax=amax_r)
        w_cat_t = derived_cached('x3_image_t_cat', (w_off, w_att),
                                 lambda: ops.pack_linear_weight_x3(torch.cat([w_off.detach(), w_att.detach()], 0).t().contiguous()))
        ops.gemm_x3(gr2, w_cat_t, C, res=gsrc, out=gsrc, amax=amax_r)                      # + offsets / logits path, in place
        gpos = None
        if ctx.needs_input_grad[1]:
            gpos = grows.view(B, N, n_cat).sum(0) @ torch.cat([w_off.detach(), w_att.detach()], 0)
        return (gsrc.view(B, N, C) if ctx.needs_input_grad[0] else None, gpos, None, gwv, gbv, gw_cat[:n_off], gb_cat[:n_off],
                gw_cat[n_off:], gb_cat[n_off:], gwo, gbo, dgamma, dbeta, None, None, None, None, None)


def x3_layer_nodes_ok(layer, src):
    """PARITY-mode training of a post-norm (self_attn, norm, ffn, norm) encoder layer on the two block nodes above."""
    attn, ffn = layer.attentions[0], layer.ffns[0]
    lin1, lin2 = ffn.layers[0][0], ffn.layers[1]
    C = src.shape[-1]
    w_cat_rows = attn.sampling_offsets.weight.shape[0] + attn.attention_weights.weight.shape[0]
    return (_X3_LAYER_NODES and _X3_GSCALE and _X3_WGRAD and src.dim() == 3 and C == 256
            and x3_train_linear_ok(src, attn.value_proj.weight) and x3_train_linear_ok(src, attn.output_proj.weight)
            and x3_train_ffn_ok(src, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
            and attn.sampling_offsets.weight.shape[1] == C and w_cat_rows % 32 == 0
            and all(m.bias is not None for m in (attn.value_proj, attn.output_proj, attn.sampling_offsets, attn.attention_weights))
            and attn.num_levels * attn.num_points <= 16
            and w_cat_rows == 3 * attn.num_heads * attn.num_levels * attn.num_points
            and src.shape[0] * src.shape[1] * max(w_cat_rows, lin1.weight.shape[0]) * 4 < _X3_MAX_BYTES)


def encoder_layer_x3_train(layer, src, pos, ref, level_hw, level_start):
    attn, ffn = layer.attentions[0], layer.ffns[0]
    lin1, lin2 = ffn.layers[0][0], ffn.layers[1]
    n0, n1 = layer.norms[0], layer.norms[1]
    mid = _X3MsdaBlockFn.apply(src, pos, ref, attn.value_proj.weight, attn.value_proj.bias, attn.sampling_offsets.weight,
                               attn.sampling_offsets.bias, attn.attention_weights.weight, attn.attention_weights.bias,
                               attn.output_proj.weight, attn.output_proj.bias, n0.weight, n0.bias, n0.eps, attn.num_heads, level_hw,
                               level_start, attn.num_points)
    return _X3FfnBlockFn.apply(mid, lin1.weight, lin1.bias, lin2.weight, lin2.bias, n1.weight, n1.bias, n1.eps)


def x3_train_ffn_ok(x, w1, b1, w2, b2):
    return (x3_train_linear_ok(x, w1) and b1 is not None and b2 is not None and w2.shape[1] == w1.shape[0] and w2.shape[0] % 32 == 0
            and w1.shape[0] % 32 == 0 and w1.requires_grad and w2.requires_grad and b1.requires_grad and b2.requires_grad)


def ffn_x3_train(x, w1, b1, w2, b2):
    return _X3FfnFn.apply(x, w1, b1, w2, b2)


class _X3Conv3x3Fn(torch.autograd.Function):
    """Training-time 3x3 / stride 1 / pad 1 convolution (no bias) in PARITY mode on the f32-class x3 kernels -- the FPN output
    convolution of the pixel decoder ([3P] MSDeformAttnPixelDecoder.output_convs, 256 -> 256 at 256^2: 30.8 ms of MIOpen f32
    implicit GEMMs per step at configs[2], 10 % of the step). Forward and grad-input are `ops.conv_x3s_nhwc` on channel-last x3a
    maps (grad-input = the convolution of grad_output with the flipped, transposed filter); the weight gradient is nine
    `ops.wgrad_x3` contractions, one per filter tap, over the ZERO-PADDED channel-last maps: with both maps padded by one pixel a
    tap is a constant row offset between two row-major matrices (border rows of grad_output are zero, so nothing leaks across
    rows or images). Layout changes are paid explicitly: the tiled transposes write the padded channel-last maps directly
    (`ops.nchw_to_nhwc_pad1`), ~2 ms of the ~15 ms this costs."""

    @staticmethod
    def forward(ctx, x, weight):
        from . import ops
        N = weight.shape[0]
        # the input goes channel-last straight into its ZERO-PADDED form (B, H + 2, W + 2, C): the forward convolution then runs
        # with pad 0 (the same zeros the implicit padding supplied) and the backward's weight-gradient taps need no padding copy
        xp = ops.nchw_to_nhwc_pad1(x.detach())
        wk = derived_cached('x3_conv_image', (weight,), lambda: ops.pack_conv_weight_x3(weight))
        y = ops.conv_x3s_nhwc(ops.x3a_encode(xp), wk, N, 3, 1, 0, None, out_split=False)
        ctx.save_for_backward(xp, weight)
        return ops.nhwc_to_nchw(y)                       # contiguous NCHW: what the GroupNorm behind it wants

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        xp, weight = ctx.saved_tensors
        B, Hp, Wp, C = xp.shape
        H, W = Hp - 2, Wp - 2
        N = weight.shape[0]
        gp = ops.nchw_to_nhwc_pad1(gy)                    # (B, H + 2, W + 2, N), zero border
        gx = gw = None
        # per-tensor pre-scale of grad_output (see _X3LinearFn); the zero border does not change the maximum
        amax = ops.absmax(gp.view(-1, N)) if _X3_GSCALE else None
        if ctx.needs_input_grad[0]:
            wt = derived_cached('x3_conv_image_dgrad', (weight,),
                                lambda: ops.pack_conv_weight_x3(weight.detach().flip(2, 3).transpose(0, 1).contiguous()))
            if amax is not None:
                # f32 rows split in the kernel with the per-tensor scale (the x3a form bakes the fixed 2^4 into the stored pieces)
                gx = ops.nhwc_to_nchw(ops.conv_x3_nhwc(gp, wt, C, 3, 1, 0, amax=amax))
            else:
                # (contiguous NCHW: a channel-last-strided gradient sent the producer's backward -- the FPN's bilinear up-sample -- down
                # torch's NHWC kernel, 4.3 ms instead of 1.4)
                gx = ops.nhwc_to_nchw(ops.conv_x3s_nhwc(ops.x3a_encode(gp), wt, C, 3, 1, 0, None, out_split=False))
        if ctx.needs_input_grad[1]:
            xr, gr = xp.view(-1, C), gp.view(-1, N)       # rows of the (B, H + 2, W + 2) grid
            Mp = xr.shape[0]
            lo, hi = W + 3, Mp - (W + 3)                  # rows outside are border rows: grad_output is zero there
            gw = torch.empty((N, 3, 3, C), dtype=torch.float32, device=xp.device)
            for ky in range(3):
                for kx in range(3):
                    off = (ky - 1) * (W + 2) + (kx - 1)
                    gw[:, ky, kx, :] = ops.wgrad_x3(gr[lo:hi], xr[lo + off:hi + off], amax=amax)
            gw = gw.permute(0, 3, 1, 2)
        return gx, gw


_X3_RESNET_TRAIN = _os.environ.get('CGG_X3_RESNET_TRAIN', '1') != '0'      # parity-mode training: trainable ResNet stages channel-last on own kernels (A/B)


class _X3ConvBnFn(torch.autograd.Function):
    """y = act(conv(x, W) * s + t (+ res)) on CHANNEL-LAST f32 maps: one convolution of a trainable ResNet stage with its FROZEN
    BatchNorm (norm_eval=True, requires_grad=False: a per-channel affine s = gamma / sqrt(var + eps), t = beta - mean s), the
    Bottleneck's ReLU and its residual add as ONE autograd node in PARITY-mode training ([3P] mmdet ResNet layer4 under
    frozen_stages=3, configs/_base_ backbones; reference call site open_set/models/mask2former_head.py:787's inputs). 1 x 1 (stride 1 / 2)
    and 3 x 3 / stride 1 / pad 1 filters. Replaces MIOpen's f32 implicit GEMMs (~120 TF/s at these shapes) + batch_norm + relu +
    add launches and their backward kernels: forward = the x3 implicit GEMM with s folded into the packed filter and (t, res, ReLU) in
    its epilogue; backward = one ReLU-mask pass, one max |g| pass (per-tensor pre-scale, see `_X3LinearFn`), grad-input as the x3
    GEMM / convolution with the transposed (flipped) filter, grad-weight as `ops.wgrad_x3` (nine taps over zero-padded maps for
    3 x 3, see `_X3Conv3x3Fn`), scaled by s.
    3 x 3 / STRIDE 2 / pad 1 (even H, W; the first block's convolution): x[iy] reaches out[oy] through tap ky = iy + 1 - 2 oy, so the
    input pixels of one parity class (iy & 1, ix & 1) see a fixed 1 x 1, 1 x 2, 2 x 1 or 2 x 2 sub-filter of grad_output -- grad-input is
    FOUR small stride-1 convolutions over grad_output (zero row / column appended: the 2-tap reach to oy + 1) whose results interleave,
    nine taps' worth of MFMA work like the forward (a zero-dilated grad_output would cost four times that); grad-weight pairs
    grad_output with the four parity-class sub-maps of the zero-padded input, in which a tap is again a constant row offset."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, res, stride, relu):
        from . import ops
        N, C, k, _ = weight.shape
        B, H, W, _ = x.shape
        x = x.detach().contiguous()
        wk = derived_cached('x3_convbn_image', (weight, scale),
                            lambda: ops.pack_conv_weight_x3(weight.detach() * scale.view(-1, 1, 1, 1)))
        r = res.detach().contiguous() if res is not None else None
        if k == 1 and stride == 1:
            y = ops.gemm_x3(x.view(-1, C), wk, N, shift, res=r.view(-1, N) if r is not None else None, relu=relu).view(B, H, W, N)
        else:
            y = ops.conv_x3_nhwc(x, wk, N, k, stride, k // 2, bias=shift, res=r, relu=relu)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.cfg = (int(stride), bool(relu), res is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        import torch.nn.functional as F
        x, weight, scale, y = ctx.saved_tensors
        stride, relu, has_res = ctx.cfg
        N, C, k, _ = weight.shape
        B, H, W, _ = x.shape
        g = gy.contiguous()
        amax = None
        if relu and _X3_GSCALE and g.numel() % 4 == 0:
            g, amax = ops.relu_backward_absmax(g, y)          # ReLU mask and max |g| in one pass
        elif relu:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)
        OH, OW = g.shape[1], g.shape[2]
        g2 = g.view(-1, N)
        if amax is None and _X3_GSCALE:
            amax = ops.absmax(g2)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if k == 1:
                wt = derived_cached('x3_convbn_image_t', (weight, scale), lambda: ops.pack_linear_weight_x3(
                    (weight.detach().flatten(1) * scale.view(-1, 1)).t().contiguous()))
                gs = ops.gemm_x3(g2, wt, C, amax=amax).view(B, OH, OW, C)
                if stride == 1:
                    gx = gs
                else:       # the pixels a stride-s 1 x 1 filter never read get no gradient
                    gx = torch.zeros_like(x)
                    gx[:, ::stride, ::stride] = gs
            elif stride == 1:
                wt = derived_cached('x3_convbn_image_t', (weight, scale), lambda: ops.pack_conv_weight_x3(
                    (weight.detach() * scale.view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1).contiguous()))
                gx = ops.conv_x3_nhwc(g, wt, C, k, 1, k // 2, amax=amax)
            else:
                taps = ([1], [2, 0])                      # filter taps seen by even / odd input coordinates, at offsets 0 (, + 1)

                def sub_images():
                    ws = weight.detach() * scale.view(-1, 1, 1, 1)
                    return [ops.pack_conv_weight_x3(ws[:, :, taps[py]][:, :, :, taps[px]].transpose(0, 1).contiguous())
                            for py in (0, 1) for px in (0, 1)]
                imgs = derived_cached('x3_convbn_image_t_s2', (weight, scale), sub_images)
                gp = F.pad(g, (0, 0, 0, 1, 0, 1))          # (B, OH + 1, OW + 1, N): zero last row / column
                gx = torch.empty_like(x)
                for py in (0, 1):
                    for px in (0, 1):
                        o = ops.conv_x3_nhwc(gp, imgs[2 * py + px], C, (len(taps[py]), len(taps[px])), 1, 0, amax=amax)
                        gx[:, py::2, px::2] = o[:, :OH, :OW]
        if ctx.needs_input_grad[1]:
            if k == 1:
                xs = x if stride == 1 else x[:, ::stride, ::stride].contiguous()
                gw = ops.wgrad_x3(g2, xs.view(-1, C), amax=amax).view(N, C, 1, 1)
            elif stride == 2:
                xp = F.pad(x, (0, 0, 1, 1, 1, 1))
                gr = F.pad(g, (0, 0, 0, 1, 0, 1)).view(-1, N)      # rows of the (B, OH + 1, OW + 1) grid the sub-maps share
                Mp = gr.shape[0]
                ph = [[xp[:, py::2, px::2].contiguous().view(-1, C) for px in (0, 1)] for py in (0, 1)]
                gw = torch.empty((N, k, k, C), dtype=torch.float32, device=x.device)
                for ky in range(3):
                    for kx in range(3):
                        off = (ky >> 1) * (OW + 1) + (kx >> 1)
                        gw[:, ky, kx, :] = ops.wgrad_x3(gr[:Mp - off], ph[ky & 1][kx & 1][off:], amax=amax)
                gw = gw.permute(0, 3, 1, 2)
            else:
                # both maps zero-padded by one pixel: a filter tap is a constant row offset between two row-major matrices
                xr = F.pad(x, (0, 0, 1, 1, 1, 1)).view(-1, C)
                gr = F.pad(g, (0, 0, 1, 1, 1, 1)).view(-1, N)
                Mp = xr.shape[0]
                lo, hi = W + 3, Mp - (W + 3)
                gw = torch.empty((N, k, k, C), dtype=torch.float32, device=x.device)
                for ky in range(3):
                    for kx in range(3):
                        off = (ky - 1) * (W + 2) + (kx - 1)
                        gw[:, ky, kx, :] = ops.wgrad_x3(gr[lo:hi], xr[lo + off:hi + off], amax=amax)
                gw = gw.permute(0, 3, 1, 2)
            gw = gw * scale.view(-1, 1, 1, 1)
        return gx, gw, None, None, (g if has_res else None), None, None


def _bn_affine(bn):
    """(s, t) of a frozen BatchNorm2d: y = x s + t"""
    def make():
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        return s.contiguous(), (bn.bias.detach().float() - bn.running_mean.detach().float() * s).contiguous()
    return derived_cached('bn_affine', (bn.weight, bn.bias, bn.running_mean, bn.running_var), make)


def _x3_convbn_ok(conv, bn):
    k, s = tuple(conv.kernel_size), tuple(conv.stride)
    return (isinstance(conv, torch.nn.Conv2d) and isinstance(bn, torch.nn.BatchNorm2d) and not bn.training and bn.affine
            and bn.track_running_stats and not bn.weight.requires_grad and not bn.bias.requires_grad and conv.bias is None
            and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and getattr(conv, 'padding_mode', 'zeros') == 'zeros'
            and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0 and conv.weight.requires_grad
            and ((k == (1, 1) and s in ((1, 1), (2, 2)) and tuple(conv.padding) == (0, 0))
                 or (k == (3, 3) and s in ((1, 1), (2, 2)) and tuple(conv.padding) == (1, 1))))


# the first block's 3 x 3 / stride-2 convolution on `_X3ConvBnFn` too (four sub-filter convolutions for grad-input, parity-class
# sub-maps for grad-weight). OFF by default: correct (tests/test_x3s_gpu.py) but 2 % SLOWER per configs[2] step than MIOpen's kernels
# on the channel-last views (85.0 vs 86.8 images/s) -- the padding / parity-class copies and 13 small launches cost more than the one
# convolution saves (profiles/r6_resnet_stage_train_ab.txt)
_X3_RESNET_S2 = _os.environ.get('CGG_X3_RESNET_S2', '0') == '1'


def _conv2_s2_on_x3(blk, H, W):
    return _X3_RESNET_S2 and tuple(blk.conv2.stride) == (2, 2) and H % 2 == 0 and W % 2 == 0


def x3_resnet_stage_ok(stage, x):
    """PARITY-mode training of a ResNet stage of Bottlenecks with frozen BatchNorm on channel-last maps and own kernels
    (`_X3ConvBnFn`; a 3 x 3 / stride-2 convolution -- the first block's -- needs even H, W, else it stays a library call on the
    channel-last view -- which is also the default, CGG_X3_RESNET_S2=1 moves it). x: the stage's channel-last f32 input (B, H, W, C)."""
    from .backbones import Bottleneck
    if not (_X3_RESNET_TRAIN and _X3_TRAIN and _X3_WGRAD and x3_enabled() and torch.is_grad_enabled() and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    B, H, W, C = x.shape
    for j, blk in enumerate(stage):
        if not isinstance(blk, Bottleneck) or not isinstance(blk.relu, torch.nn.ReLU):
            return False
        pairs = [(blk.conv1, blk.bn1), (blk.conv3, blk.bn3)]
        if blk.downsample is not None:
            if len(blk.downsample) != 2:
                return False
            pairs.append((blk.downsample[0], blk.downsample[1]))
        s2 = tuple(blk.conv2.stride)
        if s2 == (1, 1) or _conv2_s2_on_x3(blk, H, W):
            pairs.append((blk.conv2, blk.bn2))
        elif not (isinstance(blk.bn2, torch.nn.BatchNorm2d) and not blk.bn2.training):
            return False
        if not all(_x3_convbn_ok(c, b) for c, b in pairs) or tuple(blk.conv1.stride) != (1, 1) or tuple(blk.conv3.stride) != (1, 1):
            return False
        if blk.downsample is not None and tuple(blk.downsample[0].stride) != s2:
            return False
        if blk.downsample is None and (s2 != (1, 1) or blk.conv1.in_channels != blk.conv3.out_channels):
            return False
        H, W = (H - 1) // s2[0] + 1, (W - 1) // s2[1] + 1
    # the smallest map of the stage still has to fill the GEMM grid, the largest operand has to fit a 32-bit buffer descriptor
    return (B * H * W >= X3_TRAIN_ROWS
            and x.numel() * 4 < _X3_MAX_BYTES and B * (H + 2) * (W + 2) * stage[-1].conv3.out_channels * 4 < _X3_MAX_BYTES)


def resnet_stage_x3_train(stage, x, tap=None):
    """channel-last (B, H, W, C) f32 -> the stage's channel-last output, under autograd (see `x3_resnet_stage_ok`). tap: called with
    every post-ReLU map in execution order (tests pin the ReLU decisions of a float64 reference to them: an activation that is zero
    to rounding is a tie, and a flipped tie moves a gradient by one row's contribution -- ~1e-3 of a weight gradient)."""
    tap = tap or (lambda y: None)

    def cb(x, conv, bn, relu, res=None):
        s, t = _bn_affine(bn)
        y = _X3ConvBnFn.apply(x, conv.weight, s, t, res, int(conv.stride[0]), relu)
        if relu:
            tap(y)
        return y

    for blk in stage:
        identity = x if blk.downsample is None else cb(x, blk.downsample[0], blk.downsample[1], False)
        y = cb(x, blk.conv1, blk.bn1, True)
        if tuple(blk.conv2.stride) == (1, 1) or _conv2_s2_on_x3(blk, y.shape[1], y.shape[2]):
            y = cb(y, blk.conv2, blk.bn2, True)
        else:
            # (3 x 3 / stride 2: grad-input would be four sub-filter convolutions; the library's channel-last kernels on the views)
            y = torch.relu(blk.bn2(blk.conv2(y.permute(0, 3, 1, 2)))).permute(0, 2, 3, 1).contiguous()
            tap(y)
        x = cb(y, blk.conv3, blk.bn3, True, identity)
    return x


_X3_FPN_ROWS_BF16 = _os.environ.get('CGG_X3_FPN_ROWS_BF16', '1') != '0'      # ... also in throughput (bf16) mode: 170.4 -> 166.0 ms per configs[2] step
_X3_FPN_ROWS = _os.environ.get('CGG_X3_FPN_ROWS', '1') != '0'      # parity-mode training: the finest FPN level channel-last on own kernels (A/B)


class _NchwToRowsFn(torch.autograd.Function):
    """(B, C, H, W) f32 -> channel-last rows (B, H W, C) by the tiled transpose kernel (and back for the gradient)."""

    @staticmethod
    def forward(ctx, x):
        from . import ops
        B, C, H, W = x.shape
        ctx.hw = (H, W)
        return ops.nchw_to_nhwc(x.detach()).reshape(B, H * W, C)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        B, HW, C = g.shape
        return ops.nhwc_to_nchw(g.contiguous().view(B, ctx.hw[0], ctx.hw[1], C))


class _RowsToNchwFn(torch.autograd.Function):
    """channel-last rows (B, H W, C) -> contiguous (B, C, H, W) by the tiled transpose kernel (and back for the gradient)."""

    @staticmethod
    def forward(ctx, x, hw):
        from . import ops
        B, HW, C = x.shape
        return ops.nhwc_to_nchw(x.detach().contiguous().view(B, hw[0], hw[1], C))

    @staticmethod
    def backward(ctx, g):
        from . import ops
        B, C, H, W = g.shape
        return ops.nchw_to_nhwc(g).reshape(B, H * W, C), None


class _X3FpnLevelFn(torch.autograd.Function):
    """y = relu(GN2(conv3x3(GN1(cur) + up(lo)))) on channel-last f32 rows: the [3P] MSDeformAttnPixelDecoder FPN step behind a lateral
    1 x 1 convolution (lateral GroupNorm, + bilinear up-sample of the coarser level, output ConvModule = 3 x 3 convolution + GroupNorm +
    ReLU; open_set/models/mask2former_head.py:787) as ONE autograd node in PARITY-mode training. The 256^2 level of configs[2] costs
    ~38 ms per step as library modules on NCHW maps (MIOpen 1 x 1 / transposes, torch GroupNorm, add, up-sample, ReLU and their
    backward kernels, layout copies around the x3 convolution). Here: both GroupNorms are the channel-last kernels of csrc/norm.hip
    (forward with the up-sample + add / the ReLU fused, backward `cgg_group_norm_nhwc_f32_backward` incl. the up-sample's transpose);
    GN1 writes straight into the zero-bordered map the x3 convolution reads with pad 0, GN2's backward writes grad_output of the
    convolution straight into its zero-bordered form -- the maps the nine weight-gradient taps pair row by row -- so no layout or
    padding copy exists at all. cur (B, H W, C) = lateral convolution output, lo (B, h w, C) = coarser level (both channel-last rows)."""

    @staticmethod
    def forward(ctx, cur, lo, g1, b1, wc, g2, b2, groups, eps1, eps2, hw, lo_hw):
        from . import ops
        B, HW, C = cur.shape
        H, W = int(hw[0]), int(hw[1])
        N = wc.shape[0]
        cur, lo = cur.contiguous(), lo.contiguous()
        ws = ops.group_norm_nhwc_workspace(B, HW, groups, cur.device)
        y1p = ops.zero_border_map(B, H, W, C, cur.device)
        ops.group_norm_nhwc_padout(cur, g1.detach(), b1.detach(), groups, eps1, ws, (H, W), y1p, lo=lo, lo_hw=lo_hw)
        stats1 = ws[:B * groups * 2].clone()
        wk = derived_cached('x3_conv_image', (wc,), lambda: ops.pack_conv_weight_x3(wc))
        y2 = ops.conv_x3s_nhwc(ops.x3a_encode(y1p), wk, N, 3, 1, 0, None, out_split=False).view(B, HW, N)
        y3 = torch.empty_like(y2)
        ops.group_norm_nhwc(y2, g2.detach(), b2.detach(), groups, eps2, ws, relu=True, W=W, out32=(y3, 0, HW * N))
        stats2 = ws[:B * groups * 2].clone()
        ctx.save_for_backward(cur, stats1, y1p, y2, stats2, y3, g1, wc, g2)
        ctx.cfg = (int(groups), float(eps1), float(eps2), (H, W), tuple(int(v) for v in lo_hw))
        return y3

    @staticmethod
    def backward(ctx, g3):
        from . import ops
        cur, stats1, y1p, y2, stats2, y3, g1, wc, g2 = ctx.saved_tensors
        groups, eps1, eps2, (H, W), lo_hw = ctx.cfg
        B, HW, C = cur.shape
        N = wc.shape[0]
        gp = ops.zero_border_map(B, H, W, N, cur.device)
        _, dg2, db2, _ = ops.group_norm_nhwc_backward(y2, y3, g3.contiguous(), stats2, g2, groups, eps2, True, (H, W), dx_padded=gp)
        amax = ops.absmax(gp.view(-1, N)) if _X3_GSCALE else None
        wt = derived_cached('x3_conv_image_dgrad', (wc,),
                            lambda: ops.pack_conv_weight_x3(wc.detach().flip(2, 3).transpose(0, 1).contiguous()))
        if amax is not None:
            gy1 = ops.conv_x3_nhwc(gp, wt, C, 3, 1, 0, amax=amax)
        else:
            gy1 = ops.conv_x3s_nhwc(ops.x3a_encode(gp), wt, C, 3, 1, 0, None, out_split=False)
        xr, gr = y1p.view(-1, C), gp.view(-1, N)
        Mp = xr.shape[0]
        lo_, hi_ = W + 3, Mp - (W + 3)                    # rows outside are border rows: grad_output is zero there
        gw = torch.empty((N, 3, 3, C), dtype=torch.float32, device=cur.device)
        for ky in range(3):
            for kx in range(3):
                off = (ky - 1) * (W + 2) + (kx - 1)
                gw[:, ky, kx, :] = ops.wgrad_x3(gr[lo_:hi_], xr[lo_ + off:hi_ + off], amax=amax)
        dcur, dg1, db1, dlo = ops.group_norm_nhwc_backward(cur, None, gy1.view(B, HW, C), stats1, g1, groups, eps1, False, (H, W),
                                                           lo_hw=lo_hw)
        return dcur, dlo, dg1, db1, gw.permute(0, 3, 1, 2), dg2, db2, None, None, None, None, None


def nhwc_to_nchw_train(xn):
    """channel-last (B, H, W, C) f32 under autograd -> contiguous (B, C, H, W) (tiled transpose kernel both ways); the channel-last
    original rides along as `_cgg_rows` (B, H W, C) for consumers that read rows under autograd."""
    B, H, W, C = xn.shape
    rows = xn.reshape(B, H * W, C)
    out = _RowsToNchwFn.apply(rows, (H, W))
    out._cgg_rows = rows
    return out


_X3_INPUT_ROWS = _os.environ.get('CGG_X3_INPUT_ROWS', '1') != '0'      # parity-mode training: the encoder's input levels on channel-last rows (A/B)


def input_level_x3_train(cm, feat):
    """One encoder input level of the pixel decoder ([3P] MSDeformAttnPixelDecoder.input_convs: 1 x 1 convolution with bias +
    GroupNorm, no activation; open_set/models/mask2former_head.py:787's inputs) in PARITY-mode training on channel-last rows: the
    backbone map's channel-last original (`hand_nhwc` of a frozen stage / `_cgg_rows` of a trainable x3 stage) x the filter as an x3
    row GEMM (`_X3LinearFn`: forward, grad-input, grad-weight + bias), GroupNorm by the channel-last kernels (`ops.GroupNormRowsFn`).
    Returns (B, H W, C) rows -- what the encoder concatenates -- or None when the level has to take the module path (NCHW library
    convolution + GroupNorm, then flatten + transpose)."""
    import torch.nn as nn
    from . import ops
    conv = cm.conv
    gn = getattr(cm, cm.norm_name, None) if cm.norm_name else None
    if not (_X3_INPUT_ROWS and _X3_TRAIN and _X3_WGRAD and x3_enabled() and torch.is_grad_enabled() and feat.is_cuda
            and feat.dim() == 4 and feat.dtype == torch.float32 and cm.activate is None and isinstance(gn, nn.GroupNorm)
            and gn.num_groups * 8 == conv.out_channels == gn.num_channels and gn.affine
            and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1):
        return None
    B, Cin, H, W = feat.shape
    rows = getattr(feat, '_cgg_rows', None)
    if rows is not None and (tuple(rows.shape) != (B, H * W, Cin) or rows.dtype != torch.float32):
        rows = None
    if rows is None:
        nh = handed_nhwc(feat)
        rows = nh.view(B, H * W, Cin) if nh is not None else None
    if rows is None or not x3_train_linear_ok(rows, conv.weight.flatten(1)):
        return None
    y = _X3LinearFn.apply(rows, conv.weight.flatten(1), conv.bias)
    return ops.GroupNormRowsFn.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, False, None, (H, W), None)


def x3_fpn_level_ok(pd, x, lo_hw):
    """PARITY-mode training of the pixel decoder's single FPN level (lateral 1 x 1 + GN, + up-sample, 3 x 3 + GN + ReLU, mask-feature
    1 x 1) on channel-last rows and own kernels: `_X3FpnLevelFn` + `_X3LinearFn`s."""
    import torch.nn as nn
    bf = is_bf16() and _X3 and _X3_FPN_ROWS_BF16      # throughput mode may take the (more accurate) f32-class level too
    if not (_X3_FPN_ROWS and _X3_TRAIN and _X3A and _X3_WGRAD and (x3_enabled() or bf) and torch.is_grad_enabled() and x.is_cuda
            and x.dtype in ((torch.float32, torch.bfloat16) if bf else (torch.float32,)) and x.dim() == 4 and len(pd.lateral_convs) == 1):
        return False
    lat, outc, mf = pd.lateral_convs[0], pd.output_convs[0], pd.mask_feature
    gn1 = getattr(lat, lat.norm_name, None) if lat.norm_name else None
    gn2 = getattr(outc, outc.norm_name, None) if outc.norm_name else None
    B, Cin, H, W = x.shape
    C = lat.conv.out_channels
    ok = (isinstance(gn1, nn.GroupNorm) and isinstance(gn2, nn.GroupNorm) and gn1.num_groups * 8 == C == gn2.num_channels
          and gn1.num_groups == gn2.num_groups and 256 % gn1.num_groups == 0 and lat.activate is None and isinstance(outc.activate, nn.ReLU)
          and lat.conv.bias is None and outc.conv.bias is None and tuple(lat.conv.kernel_size) == (1, 1) and tuple(mf.kernel_size) == (1, 1)
          and lat.conv.groups == 1 and mf.groups == 1 and tuple(lat.conv.stride) == (1, 1) and tuple(mf.stride) == (1, 1)
          and tuple(lat.conv.padding) == (0, 0) and tuple(mf.padding) == (0, 0)
          and tuple(lat.conv.dilation) == (1, 1) and tuple(mf.dilation) == (1, 1)
          # the GEOMETRY of the output convolution is checked in both precisions (throughput mode skipped it, so a
          # non-standard output conv would have been computed silently as 3x3 / pad 1 / no bias); only the precision test is `bf`'s
          and _conv3x3_geometry_ok(outc.conv)
          and (bf or x3_train_conv3x3_ok(outc.conv, x.new_empty((B, C, H, W))))
          and Cin % 32 == 0 and C % 32 == 0 and mf.out_channels % 32 == 0 and B * H * W >= X3_TRAIN_ROWS
          and B * H * W * max(Cin, C, mf.out_channels) * 4 < _X3_MAX_BYTES and lo_hw[0] > 0 and lo_hw[1] > 0)
    return bool(ok)


def hand_nhwc(nchw, nhwc):
    """Attach the contiguous channel-last f32 tensor `nchw` (B, C, H, W) was transposed from, for consumers that read rows; valid only
    while neither tensor is written in place (`handed_nhwc` checks both version counters)."""
    nchw._cgg_nhwc = nhwc
    nchw._cgg_nhwc_versions = (nchw._version, nhwc._version)
    return nchw


def handed_nhwc(x, allow_grad=False):
    """The channel-last f32 original of the NCHW map x (`hand_nhwc`), or None: absent, x under autograd (unless `allow_grad`: a
    consumer that only READS the values, under no_grad), shapes / dtype that do not match, or either tensor modified in place since
    the hand-over."""
    nh = getattr(x, '_cgg_nhwc', None)
    if nh is None or (x.requires_grad and not allow_grad) or x.dim() != 4 or nh.dtype != torch.float32 or not nh.is_contiguous():
        return None
    B, C, H, W = x.shape
    if tuple(nh.shape) != (B, H, W, C) or getattr(x, '_cgg_nhwc_versions', None) != (x._version, nh._version):
        return None
    return nh


def fpn_level_x3_train(pd, x, lo_rows, lo_hw):
    """mask_feature (B, C_out, H, W) of the pixel decoder from the stride-4 backbone map x (B, Cin, H, W) and the finest encoder memory
    level lo_rows (B, h w, C) -- see `x3_fpn_level_ok`."""
    lat, outc, mf = pd.lateral_convs[0], pd.output_convs[0], pd.mask_feature
    gn1, gn2 = getattr(lat, lat.norm_name), getattr(outc, outc.norm_name)
    B, Cin, H, W = x.shape
    nh = handed_nhwc(x)                           # a frozen backbone stage hands its channel-last original along (backbones.ResNet)
    if nh is not None:
        xr = nh.view(B, H * W, Cin)
    else:
        xr = _NchwToRowsFn.apply(x.float())
    cur = _X3LinearFn.apply(xr, lat.conv.weight.flatten(1), None)
    y = _X3FpnLevelFn.apply(cur, lo_rows.float(), gn1.weight, gn1.bias, outc.conv.weight, gn2.weight, gn2.bias, gn1.num_groups, gn1.eps, gn2.eps,
                            (H, W), tuple(lo_hw))
    m = _X3LinearFn.apply(y, mf.weight.flatten(1), mf.bias)
    out = _RowsToNchwFn.apply(m, (H, W))
    # the channel-last form the map was computed in, for consumers that sample it channel-last (the head's loss points): saves them a
    # 1-GB strided copy back
    return hand_nhwc(out, m.detach().view(B, H, W, m.shape[-1]))


X3_CONV_ROWS = 65536      # output pixels from which a 3 x 3 training convolution takes the x3 node (16 384 -- the trainable ResNet stage's
                          # 512-channel convolutions at 32^2 -- measured the same as MIOpen: 212.4 vs 212.1 ms per configs[2] step)


def _conv3x3_geometry_ok(conv):
    """3x3 / stride 1 / pad 1 / dilation 1 / ungrouped / bias-free with channel counts the x3 kernels tile"""
    return (tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is None and conv.in_channels % 32 == 0
            and conv.out_channels % 32 == 0 and getattr(conv, 'padding_mode', 'zeros') == 'zeros')


def x3_train_conv3x3_ok(conv, x):
    """parity mode under autograd, a 3x3 / s1 / p1 / ungrouped / bias-free convolution whose channel counts the x3 kernels tile,
    large enough to be worth the layout changes."""
    return (_X3_TRAIN and _X3A and x3_enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and _conv3x3_geometry_ok(conv)
            and x.shape[0] * x.shape[2] * x.shape[3] >= X3_CONV_ROWS and x.shape[2] >= 4 and x.shape[3] >= 4
            # the kernels address their operands through 32-bit buffer descriptors: the zero-padded maps of the weight-gradient
            # contraction are the largest operand (from B = 64 at 256^2 x 256 the call raised instead of running on MIOpen)
            and x.shape[0] * (x.shape[2] + 2) * (x.shape[3] + 2) * max(conv.in_channels, conv.out_channels) * 4 < _X3_MAX_BYTES)


def conv3x3_x3_train(x, conv):
    return _X3Conv3x3Fn.apply(x, conv.weight)


_X3_MAX_BYTES = 0xFFFFFF00          # operand span the x3 kernels can address (32-bit buffer descriptors); larger -> library path
X3_TRAIN_ROWS = 8192               # rows from which parity-mode training linears run on the x3 GEMM (CGG_X3_TRAIN=0 disables)
_X3_TRAIN = _os.environ.get('CGG_X3_TRAIN', '1') != '0'
_X3_WGRAD = _os.environ.get('CGG_X3_WGRAD', '1') != '0'
_X3_GSCALE = _os.environ.get('CGG_X3_GSCALE', '1') != '0'          # per-tensor pre-scale of grad_output in the x3 training kernels


def x3_train_linear_ok(x, weight):
    return (_X3_TRAIN and x3_enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
            and weight.shape[1] % 32 == 0 and weight.shape[0] % 32 == 0 and x.shape[-1] == weight.shape[1]
            and x.numel() // x.shape[-1] >= X3_TRAIN_ROWS
            and (x.numel() // x.shape[-1]) * max(weight.shape) * 4 < _X3_MAX_BYTES)


X3_TRAIN_ROWS_MODULE = 1024        # rows from which `ParityLinear` modules (caption transformer) take the x3 node in parity-mode training


class ParityLinear(torch.nn.Linear):
    """`nn.Linear` (same parameters / state_dict keys) whose PARITY-mode training forward + backward run on the f32-class x3 node
    (`_X3LinearFn`: forward, grad-input and weight / bias gradient) once the row count makes the kernels worth it; every other
    case (inference, bf16 mode, CPU, odd shapes) is `F.linear`. Measured on the caption transformer's shapes at configs[2]
    (forward + backward, `scratch/linear_shapes_bench.py`): 1.1-1.6 x the f32 library GEMMs, and closer to float64."""

    def forward(self, x):
        import torch.nn.functional as F
        w = self.weight
        rows = x.numel() // max(x.shape[-1], 1)
        if (_X3_TRAIN and x3_enabled() and not is_bf16() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
                and w.dtype == torch.float32 and w.shape[1] % 32 == 0 and w.shape[0] % 32 == 0 and x.shape[-1] == w.shape[1]
                and rows >= X3_TRAIN_ROWS_MODULE and rows * max(w.shape) * 4 < _X3_MAX_BYTES
                and (x.requires_grad or w.requires_grad)):
            return _X3LinearFn.apply(x, w, self.bias)
        return F.linear(x, w, self.bias)


SPLITK_WGRAD_ROWS = 32768          # rows from which the training linears use `_SplitKLinearFn` (CGG_SPLITK_WGRAD=0 disables)
_SPLITK_WGRAD = _os.environ.get('CGG_SPLITK_WGRAD', '1') != '0'      # read once, like every switch of this module


def linear_bf16_train(x, weight, bias=None):
    """bf16 `F.linear` for the training step (returns bf16): split-K weight gradient for huge row counts, autocast otherwise."""
    import torch.nn.functional as F
    rows = x.numel() // x.shape[-1]
    if rows >= SPLITK_WGRAD_ROWS and weight.requires_grad and _SPLITK_WGRAD:
        return _SplitKLinearFn.apply(x, weight, bias)
    with torch.autocast(device_type='cuda', dtype=torch.bfloat16):
        return F.linear(x, weight, bias)


def linear(x, weight, bias=None):
    """Large-M library GEMM (hipBLASLt): f32 in parity mode, bf16 operands (f32 accumulate) in throughput
    mode with the bf16 weight copy cached. Returns f32."""
    import torch.nn.functional as F
    if is_bf16() and not (torch.is_grad_enabled() and weight.requires_grad):
        y = F.linear(x.to(torch.bfloat16), cast_cached(weight), cast_cached(bias) if bias is not None else None)
        return y.float()
    if is_bf16():
        return linear_bf16_train(x, weight, bias).float()
    if x3_linear_ok(x, weight):
        return linear_x3(x, weight, bias)
    if x3_train_linear_ok(x, weight):
        return _X3LinearFn.apply(x, weight, bias)
    if x3_enabled() and not torch.is_grad_enabled() and x.is_cuda and x.numel() // max(x.shape[-1], 1) >= 512:
        note_fallback('linear', f'K={weight.shape[1]} (K % 32) / dtype {x.dtype}: f32 library GEMM instead of the x3 kernel')
    return F.linear(x, weight, bias)


# Parity mode's inference stream has shape rules (K % 32, C % 32, N % 8, 1 x 1 / 3 x 3 convolutions, GN-32 over 256 channels, ...).
# A module that falls outside them runs on f32 LIBRARY calls instead of the x3 kernels -- correct, slower, and formerly silent:
# every such event is counted here (bench.py reports the total as `library_fallbacks`) and logged once per kind.
FALLBACKS = {}


def note_fallback(kind, detail):
    import logging
    if kind not in FALLBACKS:
        logging.getLogger('cgg_amd').warning('parity mode: %s left the x3 stream -- %s', kind, detail)
    FALLBACKS[kind] = FALLBACKS.get(kind, 0) + 1


def library_fallbacks():
    return sum(FALLBACKS.values())


def x3_linear_ok(x, weight):
    """parity mode, no autograd, ROCm f32 rows whose K the x3 GEMM tiles (K % 32 == 0), enough rows to be worth a tile grid."""
    return (x3_enabled() and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
            and weight.shape[1] % 32 == 0 and x.shape[-1] == weight.shape[1] and x.numel() // x.shape[-1] >= 512)


def linear_x3s(x, weight, bias=None, res=None, res_split=False, res_mod=0, relu=False, out=None, out_split=False):
    """act(x W^T + b (+ res)) on the LDS-DMA x3 GEMM (`ops.gemm_x3s`): x = x3a rows (..., K) (2-D, or a 3-D (B, R, K) view of a
    stack of images), result f32 or x3a (`out_split`), the weight's x3 image cached like `packed_cached`."""
    from . import ops
    N, K = weight.shape
    x2 = x if x.dim() <= 3 else x.reshape(-1, K)
    wk = derived_cached('x3_image', (weight,), lambda: ops.pack_linear_weight_x3(weight))
    y = ops.gemm_x3s(x2, wk, N, bias.detach() if bias is not None else None, res=res, res_split=res_split, res_mod=res_mod,
                     relu=relu, out=out.view(-1, N) if out is not None else None, out_split=out_split)
    return out if out is not None else y.view(*x.shape[:-1], N)


def linear_x3(x, weight, bias=None, res=None, relu=False, out=None):
    """act(x W^T + b (+ res)) on the f32-class x3 GEMM (`ops.gemm_x3`), the weight's x3 image cached like `packed_cached`."""
    from . import ops
    N, K = weight.shape
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    wk = derived_cached('x3_image', (weight,), lambda: ops.pack_linear_weight_x3(weight))
    r2 = res.reshape(-1, N) if res is not None else None
    y = ops.gemm_x3(x2, wk, N, bias.detach() if bias is not None else None, res=r2, relu=relu,
                    out=out.view(-1, N) if out is not None else None)
    return out if out is not None else y.view(*x.shape[:-1], N)


_CONST = {}


def const_tensor(values, like):
    """Device tensor of a python constant (list / tuple / scalar), cached per (values, dtype, device): building it
    with `new_tensor` every call is a pageable host->device copy, i.e. a stream synchronisation."""
    key = (tuple(values) if isinstance(values, (list, tuple)) else values, like.dtype, str(like.device))
    t = _CONST.get(key)
    if t is None:
        t = like.new_tensor(values)
        _CONST[key] = t
    return t


_PCACHE = {}


def packed_cached(weights, biases=None):
    """(packed MFMA-fragment image of cat(weights, 0), cat(biases) | None, N) for `ops.linear_rows_bf16`, cached
    like `cast_cached` (slice address + geometry, validated by weak references / versions of the owning
    parameters). `weights` is a tuple of (N_i, K) tensors (parameters or views of parameters). The image is bf16 in
    throughput mode and an x3 image (f32-class f16 x 3 contraction, `ops.pack_linear_weight_x3`) in parity mode; the
    kernels pick their variant from the image type."""
    from . import ops
    allp = tuple(weights) + tuple(b for b in (biases or ()) if b is not None)
    bases = [t._base if t._base is not None else t for t in allp]
    kind = 'packed' if is_bf16() else 'x3'
    key = tuple(_wkey(t, kind) for t in allp)
    hit = _PCACHE.get(key)
    if hit is not None and all(r() is b for r, b in zip(hit[0], bases)) and \
            hit[1] == tuple(b._version for b in bases) and hit[2].device == weights[0].device:
        return hit[2], hit[3], hit[4]
    if len(_PCACHE) > 4096:
        for k in [k for k, v in _PCACHE.items() if any(r() is None for r in v[0])]:
            del _PCACHE[k]
    w = torch.cat([t.detach().float() for t in weights], 0) if len(weights) > 1 else weights[0].detach().float()
    packed = ops.pack_linear_weight(w.contiguous()) if kind == 'packed' else ops.pack_linear_weight_x3(w.contiguous())
    bias = None
    if biases is not None:
        bias = torch.cat([b.detach().float() for b in biases], 0).contiguous() if len(biases) > 1 \
            else biases[0].detach().float().contiguous()
    _PCACHE[key] = ([weakref.ref(b) for b in bases], tuple(b._version for b in bases), packed, bias, int(w.shape[0]))
    return packed, bias, int(w.shape[0])


_DCACHE = {}


_KEEPALIVE = []        # stack of lists: the tensors the caches hand out while a graph pipeline warms up / is captured


class keepalive_scope:
    """`with keepalive_scope() as refs:` -- every tensor that `derived_cached`, the positional-encoding / reference-point / projection-
    table caches return inside the scope is appended to `refs`. A hipGraph replays raw device addresses: whoever captures graphs
    (pipeline.StagePipeline) keeps `refs` for its lifetime, so that a bounded cache may drop an entry (another image shape arrives)
    without freeing memory a captured graph of an earlier shape still reads."""

    def __enter__(self):
        self.refs = {}                      # id -> object (a cache hit is registered once, however often it is returned)
        _KEEPALIVE.append(self.refs)
        return self.refs

    def __exit__(self, *exc):
        _KEEPALIVE[:] = [r for r in _KEEPALIVE if r is not self.refs]
        return False


def keepalive(val):
    """register a cache's return value with every active `keepalive_scope` (cheap no-op outside one); returns val"""
    if _KEEPALIVE:
        for refs in _KEEPALIVE:
            refs[id(val)] = val
    return val


def derived_cached(tag, tensors, fn):
    """Cache `fn()` (any tensor derived from parameters, e.g. a folded bias) until one of `tensors` changes."""
    bases = [t._base if t._base is not None else t for t in tensors]
    key = (tag,) + tuple(_wkey(t, None) for t in tensors)
    hit = _DCACHE.get(key)
    if hit is not None and all(r() is b for r, b in zip(hit[0], bases)) and hit[1] == tuple(b._version for b in bases):
        return keepalive(hit[2])
    if len(_DCACHE) > 256:
        # entries whose source tensors are gone (e.g. the positional encoding of an image shape that is no longer the current one:
        # a (S, 512) bias table per decoder level = ~100 MB per shape at 1024^2) are dropped as soon as the cache has some size --
        # an evaluation set has hundreds of distinct padded shapes. A captured pipeline of such a shape holds what its graphs read
        # through `keepalive_scope`.
        for k in [k for k, v in _DCACHE.items() if any(r() is None for r in v[0])]:
            del _DCACHE[k]
    with torch.no_grad():
        val = fn()
    _DCACHE[key] = ([weakref.ref(b) for b in bases], tuple(b._version for b in bases), val)
    return keepalive(val)
