"""Torch-facing wrappers of the HIP kernels in libcgg_hip.so.

Every function takes/returns torch tensors on a ROCm device, allocates outputs with PyTorch's
caching allocator, and enqueues on torch's current stream. None of them has a CPU path: a CPU
tensor raises `CggError` (see _lib.dev_ptr). References are to the reference repo / SURVEY.md
kernel ids (K1..K19).
"""
import ctypes
import os
import math

import torch

from . import _lib
from ._lib import CGG_BF16, CGG_F32, CggError, check, dev_ptr, stream_ptr

_WS_CACHE = {}

# bench.py sets this to a dict to collect (start, end) torch.cuda.Event pairs around selected launches
# (events are recorded on torch's current stream -- the stream the kernels are launched on).
KERNEL_EVENTS = None


KERNEL_META = None          # with KERNEL_EVENTS: name -> list of per-launch dicts (flops, bytes, shape) in launch order


class _timed:
    def __init__(self, name, **meta):
        self.name = name
        self.meta = meta
        self.on = KERNEL_EVENTS is not None and not torch.cuda.is_current_stream_capturing()

    def __enter__(self):
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            KERNEL_EVENTS.setdefault(self.name, []).append((self.s, self.e))
            if KERNEL_META is not None and self.meta:
                KERNEL_META.setdefault(self.name, []).append(self.meta)


def _lib_():
    return _lib.load()


def _int_array(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


# ------------------------------------------------------------------------------------------------
# K1/K2  MSDeformAttn core   ([3P] mmcv MultiScaleDeformableAttnFunction)
# ------------------------------------------------------------------------------------------------
def _msda_dims(value, sampling_loc):
    B, Nv, H, D = value.shape
    _, Nq, H2, L, P, two = sampling_loc.shape
    if H2 != H or two != 2:
        raise CggError(f'sampling_locations shape {tuple(sampling_loc.shape)} does not match value '
                       f'{tuple(value.shape)}')
    return B, Nv, H, D, L, Nq, P


def msda_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """value (B,Nv,H,D) f32|bf16; spatial_shapes (L,2) int64; level_start_index (L,) int64;
    sampling_locations (B,Nq,H,L,P,2) f32; attention_weights (B,Nq,H,L,P) f32 -> (B,Nq,H*D) f32."""
    B, Nv, H, D, L, Nq, P = _msda_dims(value, sampling_locations)
    vdt = CGG_BF16 if value.dtype == torch.bfloat16 else CGG_F32
    if value.dtype not in (torch.float32, torch.bfloat16):
        raise CggError(f'msda_forward: value dtype {value.dtype}')
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=value.device)
    rc = _lib_().cgg_msda_forward(
        dev_ptr(value, 'value'), dev_ptr(spatial_shapes, 'spatial_shapes', torch.int64),
        dev_ptr(level_start_index, 'level_start_index', torch.int64),
        dev_ptr(sampling_locations, 'sampling_locations', torch.float32),
        dev_ptr(attention_weights, 'attention_weights', torch.float32), dev_ptr(out), B, Nv, H, D, L,
        Nq, P, vdt, stream_ptr(value.device))
    check(rc, 'cgg_msda_forward')
    return out


def msda_forward_hostlevels(value, level_hw, level_start, sampling_locations, attention_weights):
    """Same op, level table as python lists (no D2H; legal under hipGraph capture)."""
    B, Nv, H, D, L, Nq, P = _msda_dims(value, sampling_locations)
    vdt = CGG_BF16 if value.dtype == torch.bfloat16 else CGG_F32
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=value.device)
    hw = _int_array([v for pair in level_hw for v in pair])
    st = _int_array(level_start)
    rc = _lib_().cgg_msda_forward_hostlevels(
        dev_ptr(value, 'value'), hw, st, dev_ptr(sampling_locations, 'sampling_locations', torch.float32),
        dev_ptr(attention_weights, 'attention_weights', torch.float32), None, 0, dev_ptr(out), B, Nv,
        H, D, L, Nq, P, vdt, 0, stream_ptr(value.device))
    check(rc, 'cgg_msda_forward_hostlevels')
    return out


def msda_forward_fused(value, level_hw, level_start, offs_logits, ref_points, num_points):
    """value (B,Nv,H,D); offs_logits (B,Nq,ld) raw [offsets | logits] of the two linears;
    ref_points (Nq,2) -> (B,Nq,H*D). Softmax over L*P and loc = ref + off/(W,H) run in-kernel."""
    B, Nv, H, D = value.shape
    Bq, Nq, ld = offs_logits.shape
    L = len(level_start)
    P = int(num_points)
    vdt = CGG_BF16 if value.dtype == torch.bfloat16 else CGG_F32
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=value.device)
    hw = _int_array([v for pair in level_hw for v in pair])
    st = _int_array(level_start)
    vld = _value_row_stride(value)
    if vld != H * D:                        # padded value rows (`padded_value_rows`): the strided-value form of the f32 kernel
        with _timed('msda_fused'):
            rc = _lib_().cgg_msda_forward_fused_vld(ctypes.c_void_p(value.data_ptr()), vld, hw, st,
                                                    dev_ptr(offs_logits, 'offs_logits', torch.float32), ld,
                                                    dev_ptr(ref_points, 'ref_points', torch.float32), dev_ptr(out), B, Nv, H, D, L, Nq, P,
                                                    stream_ptr(value.device))
        check(rc, 'cgg_msda_forward_fused_vld')
        return out
    with _timed('msda_fused'):
        rc = _lib_().cgg_msda_forward_hostlevels(
            dev_ptr(value, 'value'), hw, st, dev_ptr(offs_logits, 'offs_logits', torch.float32), None,
            dev_ptr(ref_points, 'ref_points', torch.float32), ld, dev_ptr(out), B, Nv, H, D, L, Nq, P, vdt,
            1, stream_ptr(value.device))
    check(rc, 'cgg_msda_forward_hostlevels(fused)')
    return out


MSDA_VALUE_PAD = 0 if os.environ.get('CGG_MSDA_VALUE_PAD', '1') == '0' else 32


def padded_value_rows(B, Nv, H, D, device):
    """(buffer (B, Nv, H D + pad) f32, value view (B, Nv, H, D) into it): MSDeformAttn value rows with a padded row stride. A stride
    that is a multiple of 512 bytes (H D = 256 floats = 1 KiB) sends the four corner lines of a tap, and neighbouring pixels' lines,
    to the same L2 channels; 288 floats per row instead of 256 made the forward gather 20 % faster at configs[1] / [2] shapes
    (scratch/msda_vld_probe.py: 256 -> 115 us, 288 -> 92, 320 -> 98, 384 -> 110, 512 -> 112, 544 -> 93). CGG_MSDA_VALUE_PAD=0: packed."""
    C = H * D
    buf = torch.empty((B, Nv, C + MSDA_VALUE_PAD), dtype=torch.float32, device=device)
    return buf, buf[..., :C].unflatten(-1, (H, D))


def _value_row_stride(value):
    """floats per pixel of a (B, Nv, H, D) value operand: H D when packed, more for a `padded_value_rows` view"""
    B, Nv, H, D = value.shape
    if value.is_contiguous():
        return H * D
    vld = value.stride(1)
    if value.dtype != torch.float32 or value.stride(3) != 1 or value.stride(2) != D or value.stride(0) != Nv * vld or vld < H * D \
            or vld % 4 or value.data_ptr() % 16:
        raise CggError(f'value: (B, Nv, H, D) f32 rows, packed or with a padded row stride, expected (strides {value.stride()})')
    return vld


def msda_forward_fused_rows(rows_all, level_hw, level_start, ref_points, num_points, num_heads, head_dim):
    """`msda_forward_fused` on the rows of the MERGED projection GEMM: rows_all (B, Nq, H D + 3 H L P) f32 = [value | offsets |
    logits] per token (queries == value pixels) -> (B, Nq, H D). The value operand is read as the first H D columns of the rows
    (`cgg_msda_forward_fused_vld`), the raw offsets / logits as the rest -- nothing is copied apart."""
    B, Nq, ldr = rows_all.shape
    H, D = int(num_heads), int(head_dim)
    L, P = len(level_start), int(num_points)
    if rows_all.dtype != torch.float32 or not rows_all.is_contiguous() or ldr != H * D + 3 * H * L * P or ldr % 32:
        raise CggError(f'msda_forward_fused_rows: rows {tuple(rows_all.shape)} for H={H} D={D} L={L} P={P}')
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=rows_all.device)
    hw = _int_array([v for pair in level_hw for v in pair])
    st = _int_array(level_start)
    base = dev_ptr(rows_all, 'rows_all', torch.float32)
    offs = ctypes.c_void_p(rows_all.data_ptr() + H * D * 4)
    with _timed('msda_fused'):
        rc = _lib_().cgg_msda_forward_fused_vld(base, ldr, hw, st, offs, ldr, dev_ptr(ref_points, 'ref_points', torch.float32),
                                                dev_ptr(out), B, Nq, H, D, L, Nq, P, stream_ptr(rows_all.device))
    check(rc, 'cgg_msda_forward_fused_vld')
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                  grad_output):
    B, Nv, H, D, L, Nq, P = _msda_dims(value, sampling_locations)
    gv = torch.zeros_like(value)
    gl = torch.zeros_like(sampling_locations)
    gw = torch.zeros_like(attention_weights)
    # algorithmic bytes: value + locations + weights + grad_output in, the three gradients out
    nbytes = 4.0 * (2 * value.numel() + 2 * sampling_locations.numel() + 2 * attention_weights.numel() + grad_output.numel())
    with _timed('msda_backward', bytes=nbytes, flops=0.0, shape=(B, Nq, H, D, L, P)):
        rc = _lib_().cgg_msda_backward(
            dev_ptr(value, 'value', torch.float32), dev_ptr(spatial_shapes, 'spatial_shapes', torch.int64),
            dev_ptr(level_start_index, 'level_start_index', torch.int64),
            dev_ptr(sampling_locations, 'sampling_locations', torch.float32),
            dev_ptr(attention_weights, 'attention_weights', torch.float32),
            dev_ptr(grad_output, 'grad_output', torch.float32), dev_ptr(gv), dev_ptr(gl), dev_ptr(gw), B,
            Nv, H, D, L, Nq, P, stream_ptr(value.device))
    check(rc, 'cgg_msda_backward')
    return gv, gl, gw


def msda_backward_hostlevels(value, level_hw, level_start, sampling_locations, attention_weights, grad_output):
    """`msda_backward` with the level table as host integers: no device tensors for the shapes, no read-back / stream
    synchronisation inside the call (csrc/msda.hip `cgg_msda_backward_hostlevels`)."""
    B, Nv, H, D, L, Nq, P = _msda_dims(value, sampling_locations)
    hw = _int_array([v for pair in level_hw for v in pair])
    st = _int_array(level_start)
    # split backward on a tileable pyramid: grad_loc / grad_attn are written by the gather kernel (no zero-fill, no read of old values)
    ow = bool(_lib_().cgg_msda_backward_overwrites(hw, st, B, Nv, H, D, L, Nq, P)) and \
        sampling_locations.data_ptr() % 16 == 0 and attention_weights.data_ptr() % 16 == 0
    vld = _value_row_stride(value)
    if vld != H * D:                        # padded value rows: grad_value comes back in the same padded layout
        if not ow:
            raise CggError('msda_backward_hostlevels: padded value rows need the split backward (tileable pyramid, D = 32, P = 4)')
        gbuf = torch.zeros((B, Nv, vld), dtype=torch.float32, device=value.device)
        gv = gbuf[..., :H * D].unflatten(-1, (H, D))
    else:
        gv = torch.zeros_like(value)
    gl = torch.empty_like(sampling_locations) if ow else torch.zeros_like(sampling_locations)
    gw = torch.empty_like(attention_weights) if ow else torch.zeros_like(attention_weights)
    nbytes = 4.0 * (2 * value.numel() + 2 * sampling_locations.numel() + 2 * attention_weights.numel() + grad_output.numel())
    # split backward: the gather kernel (grad_loc / grad_attn) on a second stream next to the sorted-scatter kernel (grad_value); the
    # C call forks / joins by events, every result is ordered on the caller's stream (CGG_MSDA_BWD_2S=0: one stream)
    side = _msda_side_stream(value.device) if (ow and MSDA_BWD_2S) else None
    # two-pass sorted scatter (corners beyond the first pass' 4-pixel halo are re-sorted on larger tiles instead of costing one
    # 128-byte atomic each): a per-region counter workspace, 0 bytes where the geometry has no such form (CGG_MSDA_BWD_2P=0: A/B)
    wsb = int(_lib_().cgg_msda_backward_workspace_bytes(hw, st, B, Nv, H, D, L, Nq, P)) if MSDA_BWD_2P else 0
    ws = _workspace(wsb, value.device) if wsb > 0 else None
    with _timed('msda_backward', bytes=nbytes, flops=0.0, shape=(B, Nq, H, D, L, P)):
        rc = _lib_().cgg_msda_backward_hostlevels_ws(
            ctypes.c_void_p(value.data_ptr()), vld, hw, st, dev_ptr(sampling_locations, 'sampling_locations', torch.float32),
            dev_ptr(attention_weights, 'attention_weights', torch.float32), dev_ptr(grad_output, 'grad_output', torch.float32),
            ctypes.c_void_p(gv.data_ptr()), dev_ptr(gl), dev_ptr(gw), B, Nv, H, D, L, Nq, P, int(ow),
            dev_ptr(ws) if ws is not None else None, wsb,
            stream_ptr(value.device), ctypes.c_void_p(side.cuda_stream) if side is not None else None)
    check(rc, 'cgg_msda_backward_hostlevels_ws')
    return gv, gl, gw


MSDA_BWD_2S = os.environ.get('CGG_MSDA_BWD_2S', '1') != '0'
MSDA_BWD_2P = os.environ.get('CGG_MSDA_BWD_2P', '1') != '0'
_MSDA_SIDE = {}


def _msda_side_stream(dev):
    s = _MSDA_SIDE.get(dev)
    if s is None:
        s = _MSDA_SIDE[dev] = torch.cuda.Stream(dev)
    return s


def msda_read_levels(spatial_shapes, level_start_index, Nv):
    """mmcv's DEVICE level table -> host ((H_l, W_l), ...), (start_l, ...): `cgg_msda_read_levels`, the one synchronising call of
    the MSDeformAttn family. The result is remembered ON the `spatial_shapes` tensor object (with both tensors' version counters),
    so the six encoder layers of a forward pass and their backwards -- which share one tensor -- pay ONE device->host read, and
    every op call goes through the non-synchronising *_hostlevels entries."""
    key = (spatial_shapes._version, level_start_index.data_ptr(), level_start_index._version, int(Nv))
    hit = getattr(spatial_shapes, '_cgg_levels', None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    L = int(spatial_shapes.shape[0])
    hw, st = (ctypes.c_int32 * (2 * L))(), (ctypes.c_int32 * L)()
    rc = _lib_().cgg_msda_read_levels(dev_ptr(spatial_shapes, 'spatial_shapes', torch.int64),
                                      dev_ptr(level_start_index, 'level_start_index', torch.int64), L, int(Nv), hw, st,
                                      stream_ptr(spatial_shapes.device))
    check(rc, 'cgg_msda_read_levels')
    level_hw = tuple((int(hw[2 * l]), int(hw[2 * l + 1])) for l in range(L))
    level_start = tuple(int(st[l]) for l in range(L))
    try:
        spatial_shapes._cgg_levels = (key, level_hw, level_start)
    except Exception:
        pass
    return level_hw, level_start


class MultiScaleDeformableAttnFunction(torch.autograd.Function):
    """Drop-in for [3P] mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction (same positional signature incl. the
    unused im2col_step). The device level table is read once per `value_spatial_shapes` tensor (`msda_read_levels`); forward and
    backward then run the *_hostlevels entries: no per-call synchronisation, the split (sorted-scatter) backward where the pyramid
    allows it."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step=64):
        ctx.save_for_backward(value, sampling_locations, attention_weights)
        ctx.levels = msda_read_levels(value_spatial_shapes, value_level_start_index, value.shape[1])
        return msda_forward_hostlevels(value.contiguous(), ctx.levels[0], ctx.levels[1], sampling_locations.contiguous(),
                                       attention_weights.contiguous())

    @staticmethod
    def backward(ctx, grad_output):
        value, loc, attw = ctx.saved_tensors
        gv, gl, gw = msda_backward_hostlevels(value.contiguous().float(), ctx.levels[0], ctx.levels[1], loc.contiguous(),
                                              attw.contiguous(), grad_output.contiguous())
        return gv.to(value.dtype), None, None, gl, gw, None


class MSDeformAttnRowsFunction(torch.autograd.Function):
    """Training form of the encoder's MSDeformAttn core on the RAW projection rows: out = msda(value, loc(rows), softmax(rows))
    with loc / softmax computed inside the forward kernel (`msda_forward_fused`) and, for the backward, by one prologue kernel,
    the gather's backward kernel and one kernel mapping (grad_loc, grad_attn) back to the rows -- the ~10 autograd-recorded
    elementwise passes over (B, Nq, H, L, P[, 2]) tensors of the un-fused formulation are gone.
    value (B, Nv, H, D) f32, rows (B, Nq, 3 H L P) f32 = [offsets | logits], ref_points (Nq, 2) f32."""

    @staticmethod
    def forward(ctx, value, rows, ref_points, level_hw, level_start, num_points):
        value, rows = value.contiguous(), rows.contiguous()
        ctx.save_for_backward(value, rows, ref_points)
        ctx.geom = (tuple(tuple(int(v) for v in hw) for hw in level_hw), tuple(int(s) for s in level_start), int(num_points))
        return msda_forward_fused(value, ctx.geom[0], ctx.geom[1], rows, ref_points, ctx.geom[2])

    @staticmethod
    def backward(ctx, grad_out):
        value, rows, ref_points = ctx.saved_tensors
        level_hw, level_start, P = ctx.geom
        gv, grows = msda_rows_backward(value, rows, ref_points, level_hw, level_start, P, grad_out)
        return gv, grows, None, None, None, None


def msda_rows_backward(value, rows, ref_points, level_hw, level_start, P, grad_out):
    """Backward of `msda_forward_fused` on the raw projection rows: (grad_value (B, Nv, H, D), grad_rows (B, Nq, 3 H L P)) for
    grad_out (B, Nq, H D) -- the prologue kernel (loc, softmax), the MSDeformAttn backward kernels, the kernel mapping
    (grad_loc, grad_attn) back to the rows."""
    B, Nv, H, D = value.shape
    _, Nq, ld = rows.shape
    L = len(level_hw)
    hw = _int_array([v for pair in level_hw for v in pair])
    loc = torch.empty((B, Nq, H, L, P, 2), dtype=torch.float32, device=value.device)
    aw = torch.empty((B, Nq, H, L, P), dtype=torch.float32, device=value.device)
    check(_lib_().cgg_msda_prologue(dev_ptr(rows, 'rows', torch.float32), ld, dev_ptr(ref_points, 'ref', torch.float32), hw,
                                    dev_ptr(loc), dev_ptr(aw), B, Nq, H, L, P, stream_ptr(value.device)), 'cgg_msda_prologue')
    gv, gl, gw = msda_backward_hostlevels(value, level_hw, level_start, loc, aw, grad_out.contiguous())
    grows = torch.empty_like(rows)
    check(_lib_().cgg_msda_prologue_backward(dev_ptr(gl), dev_ptr(gw), dev_ptr(rows), ld, hw, dev_ptr(grows), B, Nq, H, L, P,
                                             stream_ptr(value.device)), 'cgg_msda_prologue_backward')
    return gv, grows


# ------------------------------------------------------------------------------------------------
# K3/K4/K5  mask logits  (open_set/models/mask2former_head.py:748-759, :825-826)
# ------------------------------------------------------------------------------------------------
# parity mode: contract the mask logits in exact f32 (f32 MFMA) instead of 3 x bf16 on (hi, lo) pairs; CGG_EXACT_F32_LOGITS=0
# restores the split kernel (A/B)
EXACT_F32_LOGITS = os.environ.get('CGG_EXACT_F32_LOGITS', '1') != '0'
# parity mode's inference cross-attention on the f16 x 3 contraction (CGG_XATTN_X3=0 = the f32-MFMA kernel, A/B)
XATTN_X3 = os.environ.get('CGG_XATTN_X3', '1') != '0'


def _throughput_mode():
    """bf16 (throughput) mode keeps the 3 x bf16 split GEMM for f32 rows: the exact-f32 MFMA form is parity mode's."""
    from . import runtime
    return runtime.is_bf16()


class PackedFeature:
    """mask_feature packed for the MFMA B operand (see include/cgg_hip.h)."""

    def __init__(self, hi, lo, B, C, h, w, f32=None):
        self.hi, self.lo, self.B, self.C, self.h, self.w = hi, lo, B, C, h, w
        self.npix = h * w
        self.words = (self.npix + 31) // 32
        self.f32 = f32        # parity mode: the un-packed f32 map (B, C, h, w) for the exact-f32 contraction


def pack_mask_feature(feat, pool=1, split=True):
    """feat (B,C,H,W) f32 -> PackedFeature of the map bilinear-downsampled by `pool`."""
    B, C, H, W = feat.shape
    if H % pool or W % pool:
        raise CggError(f'pack_mask_feature: {H}x{W} not divisible by pool={pool}')
    h, w = H // pool, W // pool
    T = (h * w + 31) // 32
    hi = torch.empty((B, T, C // 8, 32, 8), dtype=torch.bfloat16, device=feat.device)
    lo = torch.empty_like(hi) if split else None
    rc = _lib_().cgg_pack_mask_feature(dev_ptr(feat, 'mask_feature', torch.float32), dev_ptr(hi),
                                       dev_ptr(lo), B, C, H, W, pool, stream_ptr(feat.device))
    check(rc, 'cgg_pack_mask_feature')
    f32 = None
    from . import runtime
    if split and C == 256 and EXACT_F32_LOGITS and not runtime.x3_enabled():
        # parity mode: keep the f32 map for `cgg_mask_logits_f32` (pool > 1: the same 2x2 mean, same association order
        # as the pack kernel and as torch's bilinear down-sampling with all lambdas 0.5)
        if pool == 1:
            f32 = feat
        else:
            o = pool // 2 - 1
            f32 = (((feat[:, :, o::pool, o::pool] + feat[:, :, o::pool, o + 1::pool])
                    + (feat[:, :, o + 1::pool, o::pool] + feat[:, :, o + 1::pool, o + 1::pool])) * 0.25).contiguous()
    return PackedFeature(hi, lo, B, C, h, w, f32)


def mask_logits(embed, packed, want_logits=True, want_bits=False, out=None):
    """embed (B,Q,C) f32 x PackedFeature -> (logits (B,Q,h,w) f32 | None, bits (B,Q,words) int32 | None). out: a contiguous f32
    tensor of B Q h w elements that receives the logits (packed kernels only)."""
    B, Q, C = embed.shape
    if out is not None and (packed.f32 is not None or not want_logits or out.dtype != torch.float32 or not out.is_contiguous()
                            or out.numel() != B * Q * packed.npix):
        raise CggError('mask_logits: `out` must be a contiguous float32 tensor of B Q h w elements (packed kernels, want_logits)')
    if B != packed.B or C != packed.C:
        raise CggError(f'mask_logits: embed {tuple(embed.shape)} vs packed B={packed.B} C={packed.C}')
    if packed.f32 is not None and Q > 128:
        parts = [mask_logits(embed[:, s:s + 128].contiguous(), packed, want_logits, want_bits) for s in range(0, Q, 128)]
        return (torch.cat([p[0] for p in parts], 1) if want_logits else None,
                torch.cat([p[1] for p in parts], 1) if want_bits else None)
    if packed.f32 is not None:
        out = torch.empty((B, Q, packed.h, packed.w), dtype=torch.float32, device=embed.device) if want_logits else None
        bits = torch.empty((B, Q, packed.words), dtype=torch.int32, device=embed.device) if want_bits else None
        with _timed('mask_logits_full' if want_logits else 'mask_logits_bits'):
            rc = _lib_().cgg_mask_logits_f32(dev_ptr(embed, 'mask_embed', torch.float32),
                                             dev_ptr(packed.f32.contiguous(), 'mask_feature', torch.float32), dev_ptr(out),
                                             dev_ptr(bits), B, Q, C, packed.npix, stream_ptr(embed.device))
        check(rc, 'cgg_mask_logits_f32')
        return out, bits
    if out is not None:
        out = out.view(B, Q, packed.h, packed.w)
    else:
        out = torch.empty((B, Q, packed.h, packed.w), dtype=torch.float32, device=embed.device) if want_logits else None
    bits = torch.empty((B, Q, packed.words), dtype=torch.int32, device=embed.device) \
        if want_bits else None
    tag = 'mask_logits_full' if want_logits else 'mask_logits_bits'
    with _timed(tag):
        rc = _lib_().cgg_mask_logits(dev_ptr(embed, 'mask_embed', torch.float32), dev_ptr(packed.hi),
                                     dev_ptr(packed.lo), dev_ptr(out), dev_ptr(bits), B, Q, C, packed.npix,
                                     stream_ptr(embed.device))
    check(rc, 'cgg_mask_logits')
    return out, bits


def mask_logits_launches(Q, split):
    """kernel launches of one `mask_logits` call on the packed (f16 x 3 / bf16) kernels: 1 -- split mode's row groups of 128 queries
    (LDS holds 4 query tiles of hi + lo fragments) are workgroups of ONE launch since round 6 (it was ceil(Q / 128) launches)."""
    return 1


def mask_logits_bits_astat(embed, packed):
    """embed (B, Q, 256) f32 x bf16 PackedFeature -> attn-mask bits (B, Q, words) int32 of (logit < 0), consumer-fused: the query
    tiles stay in registers, the logits are never stored (csrc/mask_logits_astat.hip). Same bits as `mask_logits(..., want_bits=True)`."""
    B, Q, C = embed.shape
    if B != packed.B or C != packed.C or packed.lo is not None:
        raise CggError('mask_logits_bits_astat: bf16 PackedFeature of the same batch / channels expected')
    bits = torch.empty((B, Q, packed.words), dtype=torch.int32, device=embed.device)
    with _timed('mask_logits_bits_astat'):
        rc = _lib_().cgg_mask_logits_bits_astat(dev_ptr(embed, 'mask_embed', torch.float32), dev_ptr(packed.hi), dev_ptr(bits), B, Q, C,
                                                packed.npix, stream_ptr(embed.device))
    check(rc, 'cgg_mask_logits_bits_astat')
    return bits


def mask_logits_backward_ok(embed, feat):
    """shapes the HIP backward kernels cover (otherwise the caller keeps the torch contractions)."""
    return (embed.is_cuda and embed.dtype == torch.float32 and feat.dtype == torch.float32 and embed.shape[-1] == 256
            and (feat.shape[-2] * feat.shape[-1]) % 8 == 0)


def mask_logits_backward(embed, feat, grad_out, split, need_embed=True, need_feat=True):
    """Gradients of einsum('bqc,bchw->bqhw', embed, feat): embed (B,Q,C) f32, feat (B,C,h,w) f32, grad_out (B,Q,h,w) f32
    -> (grad_embed (B,Q,C) | None, grad_feat (B,C,h,w) | None) on cgg_mask_logits_backward (any Q: the library walks the row
    groups of its kernels -- 128 rows in split mode, 256 otherwise -- on the operands in place and sums their grad_feat parts
    in the kernel's store; round 5: no row-group copies of grad_out, no partial grad_feat maps)."""
    B, Q, C = embed.shape
    h, w = feat.shape[-2:]
    npix = h * w
    feat = feat.contiguous()
    go = grad_out.contiguous()
    embed = embed.contiguous()
    lib = _lib_()
    ge = torch.empty((B, Q, C), dtype=torch.float32, device=embed.device) if need_embed else None
    gf = torch.empty((B, C, h, w), dtype=torch.float32, device=embed.device) if need_feat else None
    ws = _workspace(lib.cgg_mask_logits_backward_workspace_bytes(B, min(Q, 128), C, npix), embed.device)
    rc = lib.cgg_mask_logits_backward(dev_ptr(embed, 'embed', torch.float32), dev_ptr(feat, 'feat', torch.float32),
                                      dev_ptr(go, 'grad_out', torch.float32), dev_ptr(ge), dev_ptr(gf), dev_ptr(ws), B, Q, C, npix,
                                      int(bool(split)), stream_ptr(embed.device))
    check(rc, 'cgg_mask_logits_backward')
    return ge, gf


def attn_mask_fix_full_rows(bits, npix):
    """In place: rows of the bit mask that block all `npix` keys are cleared."""
    rows = bits.numel() // bits.shape[-1]
    rc = _lib_().cgg_attn_mask_fix_full_rows(dev_ptr(bits, 'bits', torch.int32), rows, npix,
                                             stream_ptr(bits.device))
    check(rc, 'cgg_attn_mask_fix_full_rows')
    return bits


def attn_mask_from_logits(logits, size):
    """logits (B,Q,H,W) f32 -> bits (B,Q,ceil(h*w/32)) int32 of (bilinear-resized logit < 0)."""
    B, Q, H, W = logits.shape
    h, w = int(size[0]), int(size[1])
    words = (h * w + 31) // 32
    bits = torch.empty((B, Q, words), dtype=torch.int32, device=logits.device)
    rc = _lib_().cgg_attn_mask_from_logits(dev_ptr(logits, 'logits', torch.float32), dev_ptr(bits),
                                           B * Q, H, W, h, w, stream_ptr(logits.device))
    check(rc, 'cgg_attn_mask_from_logits')
    return bits


def unpack_bits(bits, npix):
    """bits (..., words) int32 -> bool (..., npix)  (test / debugging helper, plain torch)."""
    shifts = torch.arange(32, device=bits.device, dtype=torch.int32)
    b = ((bits.unsqueeze(-1) >> shifts) & 1).bool()
    return b.flatten(-2)[..., :npix]


# ------------------------------------------------------------------------------------------------
# K6  masked cross attention core
# ------------------------------------------------------------------------------------------------
def _workspace(nbytes, device):
    """Per-(device, stream) scratch buffer shared by the kernels that need one (grown on demand, never shrunk)."""
    key = (device.index, stream_ptr(device).value)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS_CACHE[key] = ws
    return ws


def _xattn_ws(lib_fn, q, B, Q, H, D, S):
    return _workspace(lib_fn(B, Q, H, D, S), q.device)


def masked_xattn(q, kv, bits, num_heads, scale=None, return_lse=False):
    """q (B,Q,E) f32 projected queries; kv (B,S,2E) f32 [K|V] projected; bits (B,Q,ceil(S/32)) int32
    (bit set = blocked) or None -> (B,Q,E) f32 = softmax(q k^T * scale + mask) v, per head.
    return_lse: also the (B,H,Q) log-sum-exp rows that `masked_xattn_backward` consumes."""
    B, Q, E = q.shape
    if Q > 128:     # the kernels hold <= 4 query tiles per workgroup: split the (independent) queries
        parts = [masked_xattn(q[:, s:s + 128].contiguous(), kv, None if bits is None else bits[:, s:s + 128].contiguous(),
                              num_heads, scale, return_lse) for s in range(0, Q, 128)]
        if return_lse:
            return torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 2)
        return torch.cat(parts, 1)
    S = kv.shape[1]
    H = int(num_heads)
    D = E // H
    dev_ptr(q, 'q', torch.float32)       # refuses CPU tensors before any torch.cuda call
    if kv.shape[2] != 2 * E:
        raise CggError(f'masked_xattn: kv last dim {kv.shape[2]} != 2*{E}')
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = _lib_()
    ws = _xattn_ws(lib.cgg_masked_xattn_workspace_bytes, q, B, Q, H, D, S)
    out = torch.empty((B, Q, E), dtype=torch.float32, device=q.device)
    from . import runtime
    if XATTN_X3 and not return_lse and runtime.x3_enabled() and kv.is_cuda and kv.dtype == torch.float32 \
            and kv.stride(2) == 1 and kv.stride(1) % 4 == 0 and kv.stride(0) % 4 == 0:
        # parity mode, inference: both products on the f32-class f16 x 3 contraction (csrc/xattn_x3.hip); kv may be a column
        # slice of a merged projection (rows / images at their strides)
        rc = lib.cgg_masked_xattn_forward_x3(dev_ptr(q, 'q', torch.float32), ctypes.c_void_p(kv.data_ptr()), kv.stride(1), kv.stride(0),
                                             dev_ptr(bits, 'bits', torch.int32), dev_ptr(out), dev_ptr(ws), B, Q, H, D, S, float(scale),
                                             stream_ptr(q.device))
        check(rc, 'cgg_masked_xattn_forward_x3')
        return out
    if not kv.is_contiguous() and not return_lse:
        # a column slice of the merged [K | V] projection of the layers that read one level: rows / images at their strides
        if not kv.is_cuda or kv.dtype != torch.float32 or kv.stride(2) != 1:
            raise CggError('masked_xattn: a strided kv must be a float32 ROCm tensor with a contiguous last dim')
        rc = lib.cgg_masked_xattn_forward_strided(dev_ptr(q, 'q', torch.float32), ctypes.c_void_p(kv.data_ptr()), kv.stride(1),
                                                  kv.stride(0), dev_ptr(bits, 'bits', torch.int32), dev_ptr(out), dev_ptr(ws), B, Q,
                                                  H, D, S, float(scale), stream_ptr(q.device))
        check(rc, 'cgg_masked_xattn_forward_strided')
        return out
    dev_ptr(kv, 'kv', torch.float32)
    if return_lse:
        lse = torch.empty((B, H, Q), dtype=torch.float32, device=q.device)
        rc = lib.cgg_masked_xattn_forward_lse(dev_ptr(q, 'q', torch.float32), dev_ptr(kv, 'kv', torch.float32),
                                              dev_ptr(bits, 'bits', torch.int32), dev_ptr(out), dev_ptr(lse), dev_ptr(ws),
                                              B, Q, H, D, S, float(scale), _xattn_train_dtype(forward=True), stream_ptr(q.device))
        check(rc, 'cgg_masked_xattn_forward_lse')
        return out, lse
    rc = lib.cgg_masked_xattn_forward(dev_ptr(q, 'q', torch.float32), dev_ptr(kv, 'kv', torch.float32),
                                      dev_ptr(bits, 'bits', torch.int32), dev_ptr(out), dev_ptr(ws), B,
                                      Q, H, D, S, float(scale), CGG_F32, stream_ptr(q.device))
    check(rc, 'cgg_masked_xattn_forward')
    return out


CGG_F32_BF16MFMA = 2                 # include/cgg_hip.h: f32 rows in memory, bf16 MFMA operands (training kernels of throughput mode)
CGG_F32_X3 = 3                       # cgg_masked_xattn_forward_lse only: the forward on the f32-class f16 x 3 contraction
XATTN_BF16_TRAIN = os.environ.get('CGG_XATTN_BF16_TRAIN', '1') != '0'
XATTN_X3_TRAIN = os.environ.get('CGG_XATTN_X3_TRAIN', '1') != '0'
XATTN_X3_BWD = os.environ.get('CGG_XATTN_X3_BWD', '1') != '0'       # parity-mode training: the cross-attention backward on f16 x 3 (0: f32 MFMA)


def _xattn_train_dtype(forward=False):
    """kv_dtype of the training cross-attention kernels: throughput (bf16) mode multiplies on bf16 MFMA operands (f32 accumulate,
    f32 rows in memory) like its GEMMs do; parity mode runs the forward on the f32-class f16 x 3 contraction (the inference kernel,
    which also writes the log-sum-exp rows) and the backward on the exact f32 MFMA."""
    from . import runtime
    if runtime.is_bf16():
        return CGG_F32_BF16MFMA if XATTN_BF16_TRAIN else CGG_F32
    return CGG_F32_X3 if (forward and XATTN_X3_TRAIN and XATTN_X3 and runtime.x3_enabled()) else CGG_F32


def masked_xattn_backward(q, kv, bits, out, lse, grad_out, num_heads, scale=None):
    """Gradients of `masked_xattn` w.r.t. q (B,Q,E) and kv (B,S,2E) from the saved output / log-sum-exp rows: recomputes
    the probabilities tile by tile from the bit mask, nothing of size Q x S is stored (cgg_masked_xattn_backward)."""
    B, Q, E = q.shape
    S = kv.shape[1]
    H = int(num_heads)
    D = E // H
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    grad_out = grad_out.contiguous()
    if Q > 128:     # independent query groups: grad_q per group, grad_kv summed over the groups
        gq, gkv = [], None
        for s in range(0, Q, 128):
            a, b = masked_xattn_backward(q[:, s:s + 128].contiguous(), kv,
                                         None if bits is None else bits[:, s:s + 128].contiguous(),
                                         out[:, s:s + 128].contiguous(), lse[:, :, s:s + 128].contiguous(),
                                         grad_out[:, s:s + 128].contiguous(), num_heads, scale)
            gq.append(a)
            gkv = b if gkv is None else gkv.add_(b)
        return torch.cat(gq, 1), gkv
    lib = _lib_()
    ws = _xattn_ws(lib.cgg_masked_xattn_backward_workspace_bytes, q, B, Q, H, D, S)
    gq = torch.empty_like(q)
    gkv = torch.empty_like(kv)
    from . import runtime
    if XATTN_X3_BWD and XATTN_X3 and not runtime.is_bf16() and runtime.x3_enabled() and E % 4 == 0:
        # parity mode: the f16 x 3 form of the kernel (f32-class, 2.7 x less matrix-pipe time than the f32 MFMA); the gradient operands
        # take their pre-scale from max |grad_out|
        amax = absmax(grad_out.view(-1, E))
        rc = lib.cgg_masked_xattn_backward_x3(dev_ptr(q, 'q', torch.float32), dev_ptr(kv, 'kv', torch.float32),
                                              dev_ptr(bits, 'bits', torch.int32), dev_ptr(out, 'out', torch.float32),
                                              dev_ptr(lse, 'lse', torch.float32), dev_ptr(grad_out, 'grad_out', torch.float32),
                                              dev_ptr(amax), dev_ptr(gq), dev_ptr(gkv), dev_ptr(ws), B, Q, H, D, S, float(scale),
                                              stream_ptr(q.device))
        check(rc, 'cgg_masked_xattn_backward_x3')
        return gq, gkv
    rc = lib.cgg_masked_xattn_backward(dev_ptr(q, 'q', torch.float32), dev_ptr(kv, 'kv', torch.float32),
                                       dev_ptr(bits, 'bits', torch.int32), dev_ptr(out, 'out', torch.float32),
                                       dev_ptr(lse, 'lse', torch.float32), dev_ptr(grad_out, 'grad_out', torch.float32),
                                       dev_ptr(gq), dev_ptr(gkv), dev_ptr(ws), B, Q, H, D, S, float(scale), _xattn_train_dtype(),
                                       stream_ptr(q.device))
    check(rc, 'cgg_masked_xattn_backward')
    return gq, gkv


# ------------------------------------------------------------------------------------------------
# K16  caption grounding pair costs
# ------------------------------------------------------------------------------------------------
def grounding_supported(pred, cap):
    return (pred.is_cuda and pred.dtype == torch.float32 and cap.dtype == torch.float32 and pred.shape[1] <= 256
            and cap.shape[1] <= 64 and pred.shape[2] % 8 == 0)


def grounding_pair_costs(pred, cap, cap_mask, inv_temperature):
    """pred (Bp,Q,d) f32, cap (Bc,T,d) f32, cap_mask (Bc,T) int32 -> cost (2,Bc,Bp) f32 (l2v, v2l) of
    grounding_loss.py:32-58 for every (caption, image) pair."""
    Bp, Q, d = pred.shape
    Bc, T, _ = cap.shape
    cost = torch.empty((2, Bc, Bp), dtype=torch.float32, device=pred.device)
    rc = _lib_().cgg_grounding_pair_costs(dev_ptr(pred, 'pred', torch.float32), dev_ptr(cap, 'cap', torch.float32),
                                          dev_ptr(cap_mask, 'cap_mask', torch.int32), dev_ptr(cost), Bp, Bc, Q, T, d,
                                          float(inv_temperature), stream_ptr(pred.device))
    check(rc, 'cgg_grounding_pair_costs')
    return cost


def grounding_pair_costs_backward(pred, cap, cap_mask, grad_cost, inv_temperature):
    """-> dsim (Bp, Bc*T, Q): d loss / d (cap[i,t] . pred[j,q]) from grad_cost (2,Bc,Bp)."""
    Bp, Q, d = pred.shape
    Bc, T, _ = cap.shape
    dsim = torch.empty((Bp, Bc * T, Q), dtype=torch.float32, device=pred.device)
    rc = _lib_().cgg_grounding_pair_costs_backward(
        dev_ptr(pred, 'pred', torch.float32), dev_ptr(cap, 'cap', torch.float32),
        dev_ptr(cap_mask, 'cap_mask', torch.int32), dev_ptr(grad_cost.contiguous(), 'grad_cost', torch.float32),
        dev_ptr(dsim), Bp, Bc, Q, T, d, float(inv_temperature), stream_ptr(pred.device))
    check(rc, 'cgg_grounding_pair_costs_backward')
    return dsim


# ------------------------------------------------------------------------------------------------
# K18  generator + cross-entropy row kernels
# ------------------------------------------------------------------------------------------------
def _ce_dtype(logits):
    if logits.dtype == torch.float32:
        return CGG_F32
    if logits.dtype == torch.bfloat16:
        return CGG_BF16
    raise CggError(f'ce_rows: logits must be f32 or bf16 (got {logits.dtype})')


def ce_rows_forward(logits, target, ignore_index):
    """logits (M, N) f32 / bf16 (row stride >= N), target (M,) int64 -> (loss (M,), lse (M,)) f32."""
    M, N = logits.shape
    if logits.stride(1) != 1:
        raise CggError('ce_rows_forward: rows must be contiguous')
    loss = torch.empty(M, dtype=torch.float32, device=logits.device)
    lse = torch.empty(M, dtype=torch.float32, device=logits.device)
    rc = _lib_().cgg_ce_rows_forward(ctypes.c_void_p(logits.data_ptr()), dev_ptr(target, 'target', torch.int64), dev_ptr(loss),
                                     dev_ptr(lse), M, N, int(logits.stride(0)), int(-100 if ignore_index is None else ignore_index),
                                     _ce_dtype(logits), stream_ptr(logits.device))
    check(rc, 'cgg_ce_rows_forward')
    return loss, lse


def match_cost_rows(x, square=False):
    """x (..., P) f32 point-sampled mask logits (contiguous, P % 4 == 0) -> (sigmoid(x) (..., P), sum_p softplus(x) (...),
    sum_p sigmoid(x) [or sigmoid(x)^2 with square] (...)) in one pass (csrc/ce_rows.hip `cgg_match_cost_rows`): the prediction-only
    halves of the Hungarian mask / dice costs."""
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[-1] % 4:
        raise CggError('match_cost_rows: x must be a contiguous float32 ROCm tensor with a last dim that is a multiple of 4')
    P = x.shape[-1]
    rows = x.numel() // P
    sig = torch.empty_like(x)
    sp = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    ss = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    rc = _lib_().cgg_match_cost_rows(dev_ptr(x, 'x', torch.float32), dev_ptr(sig), dev_ptr(sp), dev_ptr(ss), rows, P, int(bool(square)),
                                     stream_ptr(x.device))
    check(rc, 'cgg_match_cost_rows')
    return sig, sp, ss


def ce_rows_backward_(logits, target, lse, grad_rows, ignore_index):
    """logits (M, N) -> d loss / d logits IN PLACE: grad_rows[row] * (softmax - onehot); returns `logits`."""
    M, N = logits.shape
    rc = _lib_().cgg_ce_rows_backward(ctypes.c_void_p(logits.data_ptr()), dev_ptr(target, 'target', torch.int64),
                                      dev_ptr(lse, 'lse', torch.float32), dev_ptr(grad_rows.contiguous(), 'grad_rows', torch.float32),
                                      M, N, int(logits.stride(0)), int(-100 if ignore_index is None else ignore_index),
                                      _ce_dtype(logits), stream_ptr(logits.device))
    check(rc, 'cgg_ce_rows_backward')
    return logits


# ------------------------------------------------------------------------------------------------
# K19  inference tail
# ------------------------------------------------------------------------------------------------
def upsample_bilinear(x, size):
    """x (N,C,H,W) or (N,H,W) f32 -> bilinear (align_corners=False) resize to `size`."""
    shp = x.shape
    H, W = shp[-2:]
    h, w = int(size[0]), int(size[1])
    n = x.numel() // (H * W)
    y = torch.empty(tuple(shp[:-2]) + (h, w), dtype=torch.float32, device=x.device)
    rc = _lib_().cgg_upsample_bilinear(dev_ptr(x, 'x', torch.float32), dev_ptr(y), n, H, W, h, w,
                                       stream_ptr(x.device))
    check(rc, 'cgg_upsample_bilinear')
    return y


def instance_masks(logits, sel, up_size, crop_size, out_size):
    """logits (Q,H,W) f32 low-res; sel (n,) int32 -> masks (n,oh,ow) bool, mask_score (n,), bbox (n,4)."""
    Q, H, W = logits.shape
    n = sel.numel()
    oh, ow = int(out_size[0]), int(out_size[1])
    masks = torch.empty((n, oh, ow), dtype=torch.uint8, device=logits.device)
    score = torch.empty((n,), dtype=torch.float32, device=logits.device)
    bbox = torch.empty((n, 4), dtype=torch.float32, device=logits.device)
    if n == 0:
        return masks.bool(), score, bbox
    ws = torch.empty((n, 8), dtype=torch.int32, device=logits.device)
    rc = _lib_().cgg_instance_masks(dev_ptr(logits, 'logits', torch.float32),
                                    dev_ptr(sel, 'sel', torch.int32), dev_ptr(masks), dev_ptr(score),
                                    dev_ptr(bbox), dev_ptr(ws), Q, H, W, int(up_size[0]),
                                    int(up_size[1]), int(crop_size[0]), int(crop_size[1]), oh, ow, n,
                                    stream_ptr(logits.device))
    check(rc, 'cgg_instance_masks')
    return masks.view(torch.bool), score, bbox


def panoptic_argmax(logits, keep, score, up_size, crop_size, out_size):
    """-> ids (oh,ow) int32 in [0,n), win_half (oh,ow) uint8, counts (n,3) int32."""
    Q, H, W = logits.shape
    n = keep.numel()
    oh, ow = int(out_size[0]), int(out_size[1])
    ids = torch.empty((oh, ow), dtype=torch.int32, device=logits.device)
    half = torch.empty((oh, ow), dtype=torch.uint8, device=logits.device)
    counts = torch.empty((n, 3), dtype=torch.int32, device=logits.device)
    rc = _lib_().cgg_panoptic_argmax(dev_ptr(logits, 'logits', torch.float32),
                                     dev_ptr(keep, 'keep', torch.int32),
                                     dev_ptr(score, 'score', torch.float32), dev_ptr(ids), dev_ptr(half),
                                     dev_ptr(counts), Q, H, W, int(up_size[0]), int(up_size[1]),
                                     int(crop_size[0]), int(crop_size[1]), oh, ow, n,
                                     stream_ptr(logits.device))
    check(rc, 'cgg_panoptic_argmax')
    return ids, half, counts


def panoptic_paint(ids, win_half, lut_val, lut_half, void_label):
    seg = torch.empty_like(ids)
    rc = _lib_().cgg_panoptic_paint(dev_ptr(ids, 'ids', torch.int32), dev_ptr(win_half, 'win_half', torch.uint8),
                                    dev_ptr(lut_val, 'lut_val', torch.int32),
                                    dev_ptr(lut_half, 'lut_half', torch.int32), dev_ptr(seg),
                                    ids.numel(), int(void_label), stream_ptr(ids.device))
    check(rc, 'cgg_panoptic_paint')
    return seg


def rowwise_softmax_argmax(x, want_prob=True):
    """x (rows,n) f32 -> (prob (rows,n) | None, max prob (rows,), argmax (rows,) int64)."""
    rows, n = x.shape
    prob = torch.empty_like(x) if want_prob else None
    maxv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    arg = torch.empty((rows,), dtype=torch.int64, device=x.device)
    rc = _lib_().cgg_rowwise_softmax_argmax(dev_ptr(x, 'x', torch.float32), dev_ptr(prob), dev_ptr(maxv),
                                            dev_ptr(arg), rows, n, stream_ptr(x.device))
    check(rc, 'cgg_rowwise_softmax_argmax')
    return prob, maxv, arg


# ------------------------------------------------------------------------------------------------
# K7/K11  skinny linear / residual LayerNorm (query side of the decoder)
# ------------------------------------------------------------------------------------------------
def linear_rows(x, weight, bias=None, relu=False, res=None, split=True, out=None):
    """x (..., K) f32 (last dim contiguous) @ weight (N,K)^T + bias -> (..., N) f32; optional ReLU, then + res."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is None:
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    else:
        y = out.reshape(-1, out.shape[-1]) if out.dim() != 2 else out
        if y.stride(-1) != 1 or y.shape[0] != M or y.shape[1] != N or y.dtype != torch.float32:
            raise CggError('linear_rows: bad `out` view')
    r2 = None
    if res is not None:
        r2 = res.reshape(-1, N)
        if r2.stride(-1) != 1:
            r2 = r2.contiguous()
    for t, nm in ((x2, 'x'), (weight, 'weight')):
        if not t.is_cuda or t.dtype != torch.float32:
            raise CggError(f'linear_rows: {nm} must be a float32 ROCm tensor')
    if not weight.is_contiguous():
        weight = weight.contiguous()
    rc = _lib_().cgg_linear_rows(ctypes.c_void_p(x2.data_ptr()), x2.stride(0), dev_ptr(weight),
                                 dev_ptr(bias, 'bias', torch.float32),
                                 ctypes.c_void_p(r2.data_ptr()) if r2 is not None else None,
                                 r2.stride(0) if r2 is not None else 0, ctypes.c_void_p(y.data_ptr()),
                                 y.stride(0), M, N, K,
                                 1 if relu else 0, (2 if EXACT_F32_LOGITS and not _throughput_mode() else 1) if split else 0,
                                 stream_ptr(x.device))
    check(rc, 'cgg_linear_rows')
    return y.view(*x.shape[:-1], N) if out is None else out


def add_layernorm(a, b, gamma, beta, eps=1e-5):
    """LayerNorm(a + b) over the last dim (b may be None)."""
    N = a.shape[-1]
    a2 = a.contiguous()
    b2 = b.contiguous() if b is not None else None
    y = torch.empty_like(a2)
    rc = _lib_().cgg_add_layernorm(dev_ptr(a2, 'a', torch.float32), dev_ptr(b2, 'b', torch.float32),
                                   dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(beta, 'beta', torch.float32),
                                   dev_ptr(y), a2.numel() // N, N, float(eps), stream_ptr(a.device))
    check(rc, 'cgg_add_layernorm')
    return y


def group_norm(x, gamma, beta, groups, eps=1e-5, relu=False):
    """x (B,C,H,W) f32 NCHW -> GroupNorm(groups)(x) (* gamma + beta) (+ ReLU)."""
    B, C, H, W = x.shape
    x = x.contiguous()
    lib = _lib_()
    nbytes = lib.cgg_group_norm_workspace_bytes(B, C, H, W, int(groups))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    rc = lib.cgg_group_norm(dev_ptr(x, 'x', torch.float32), dev_ptr(gamma, 'gamma', torch.float32),
                            dev_ptr(beta, 'beta', torch.float32), dev_ptr(y), dev_ptr(ws), B, C, H, W,
                            int(groups), float(eps), 1 if relu else 0, stream_ptr(x.device))
    check(rc, 'cgg_group_norm')
    return y


# ------------------------------------------------------------------------------------------------
# throughput-mode encoder stream (bf16 activations between the library GEMMs, f32 residual stream)
# ------------------------------------------------------------------------------------------------
def msda_forward_fused_bf16(value, level_hw, level_start, offs_logits, ref_points, num_points, head_major=False):
    """value (B,Nv,H,D) bf16 -- or, head_major=True, (B,H,Nv,D) as `encoder_proj(..., value_head_major=True)` writes it --,
    offs_logits (B,Nq,ld) bf16 raw [offsets | logits], ref_points (Nq,2) f32 -> (B,Nq,H*D) bf16."""
    if head_major:
        B, H, Nv, D = value.shape
    else:
        B, Nv, H, D = value.shape
    _, Nq, ld = offs_logits.shape
    out = torch.empty((B, Nq, H * D), dtype=torch.bfloat16, device=value.device)
    hw = _int_array([v for pair in level_hw for v in pair])
    st = _int_array(level_start)
    fn = _lib_().cgg_msda_forward_fused_bf16_hm if head_major else _lib_().cgg_msda_forward_fused_bf16
    with _timed('msda_fused'):
        rc = fn(dev_ptr(value, 'value', torch.bfloat16), hw, st, dev_ptr(offs_logits, 'offs_logits', torch.bfloat16),
                ld, dev_ptr(ref_points, 'ref_points', torch.float32), dev_ptr(out), B, Nv, H, D, len(level_start), Nq,
                int(num_points), stream_ptr(value.device))
    check(rc, 'cgg_msda_forward_fused_bf16')
    return out


def add_layernorm_stream(a, b, gamma, beta, eps=1e-5, pos=None, want_f32=True, want_bf16=True, want_pos=False):
    """LN(a + b) over the last dim (256). a f32, b f32|bf16|None. Returns (y f32 | None, bf16(y) | None,
    bf16(y + pos[row % len(pos)]) | None) from ONE pass."""
    N = a.shape[-1]
    rows = a.numel() // N
    y32 = torch.empty(a.shape, dtype=torch.float32, device=a.device) if want_f32 else None
    y16 = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device) if want_bf16 else None
    yp16 = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device) if want_pos else None
    adt = CGG_BF16 if a.dtype == torch.bfloat16 else CGG_F32
    bdt = CGG_BF16 if (b is not None and b.dtype == torch.bfloat16) else CGG_F32
    rc = _lib_().cgg_add_layernorm_ex(
        dev_ptr(a, 'a'), adt, dev_ptr(b, 'b'), bdt, dev_ptr(gamma, 'gamma', torch.float32),
        dev_ptr(beta, 'beta', torch.float32), dev_ptr(pos, 'pos', torch.float32),
        pos.shape[0] if pos is not None else 0, dev_ptr(y32), dev_ptr(y16), dev_ptr(yp16), rows, N, float(eps),
        stream_ptr(a.device))
    check(rc, 'cgg_add_layernorm_ex')
    return y32, y16, yp16


def pack_bottleneck64(w1, b1, w2, b2, w3, b3):
    """BN-folded weights of a ResNet layer1 identity Bottleneck -> the packed operands of `bottleneck64`:
    w1 (64, 256), w2 (64, 64, 3, 3), w3 (256, 64) (any float dtype), biases (64,), (64,), (256,)."""
    dev = w1.device
    w1p = pack_linear_weight(w1.reshape(64, 256).float())
    w2p = pack_linear_weight(w2.float().permute(0, 2, 3, 1).reshape(64, 576).contiguous())
    T = torch.arange(8, device=dev).view(8, 1)
    jj = torch.arange(32, device=dev).view(1, 32)
    perm = (64 * (T // 2) + 4 * (jj // 2) + 2 * (T % 2) + (jj % 2)).reshape(-1)          # packed row 32 T + j <- output channel
    w3p = pack_linear_weight(w3.reshape(256, 64).float()[perm].contiguous())
    return (w1p, b1.float().contiguous(), w2p, b2.float().contiguous(), w3p, b3.float().contiguous())


def bottleneck64_ok(x):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] == 256 and x.is_contiguous()
            and x.shape[1] % 8 == 0 and x.shape[2] % 16 == 0 and x.numel() < (1 << 31))


def bottleneck64(x, packed):
    """x (B, H, W, 256) bf16 channel-last -> relu(conv3(relu(conv2(relu(conv1 x)))) + x) of a BN-folded ResNet layer1 identity
    Bottleneck in ONE launch; `packed` from `pack_bottleneck64`."""
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    w1p, b1, w2p, b2, w3p, b3 = packed
    with _timed('bottleneck64'):
        rc = _lib_().cgg_bottleneck64_bf16(dev_ptr(x, 'x', torch.bfloat16), dev_ptr(w1p), dev_ptr(b1, 'b1', torch.float32),
                                           dev_ptr(w2p), dev_ptr(b2, 'b2', torch.float32), dev_ptr(w3p),
                                           dev_ptr(b3, 'b3', torch.float32), dev_ptr(y), B, H, W, C, 64, stream_ptr(x.device))
    check(rc, 'cgg_bottleneck64_bf16')
    return y


def pack_decoder_k_weight(weight):
    """weight (n * 256, 256) f32 (stacked key projections) -> packed bf16 operand of `decoder_kv_proj` (uint8 tensor)."""
    N, K = weight.shape
    out = torch.empty((_lib_().cgg_linear_rows_packed_bytes(N, K),), dtype=torch.uint8, device=weight.device)
    w = weight.detach().float().contiguous()
    check(_lib_().cgg_decoder_kv_pack_k(dev_ptr(w, 'weight', torch.float32), dev_ptr(out), N, K, stream_ptr(weight.device)),
          'cgg_decoder_kv_pack_k')
    return out


def decoder_kv_proj(m16, mp16, wkp, bk, wvp):
    """k (B, hw, NK) = mp16 Wk^T + bk and vt (B, NK, hw) = Wv m16^T (bf16) for one memory level in ONE launch; m16 / mp16
    (B, hw, 256) bf16, wkp from `pack_decoder_k_weight`, wvp from `pack_linear_weight`, bk (NK,) f32."""
    B, hw, C = m16.shape
    NK = bk.numel()
    k = torch.empty((B, hw, NK), dtype=torch.bfloat16, device=m16.device)
    vt = torch.empty((B, NK, hw), dtype=torch.bfloat16, device=m16.device)
    with _timed('decoder_kv_proj'):
        rc = _lib_().cgg_decoder_kv_proj_bf16(dev_ptr(m16, 'm16', torch.bfloat16), dev_ptr(mp16, 'mp16', torch.bfloat16),
                                              dev_ptr(wkp), dev_ptr(bk, 'bk', torch.float32), dev_ptr(wvp), dev_ptr(k),
                                              dev_ptr(vt), B, hw, C, NK, stream_ptr(m16.device))
    check(rc, 'cgg_decoder_kv_proj_bf16')
    return k, vt


def pack_encoder_proj_weight(weight):
    """weight (256 .. 384 in steps of 32, 256) f32 -> packed bf16 operand of `encoder_proj` (uint8 tensor)."""
    N, K = weight.shape
    out = torch.empty((_lib_().cgg_linear_rows_packed_bytes(N, K),), dtype=torch.uint8, device=weight.device)
    w = weight.detach().float().contiguous()
    check(_lib_().cgg_encoder_proj_pack(dev_ptr(w, 'weight', torch.float32), dev_ptr(out), N, K, stream_ptr(weight.device)),
          'cgg_encoder_proj_pack')
    return out


def encoder_proj(x16, xp16, wvp, bv, wcp, bc, pos16=None, value_head_major=False):
    """value = x16 Wv^T + bv (..., 256) and offs = xp16 Wc^T + bc (..., NC), both bf16, in ONE launch over the bf16 rows
    (weights from `pack_encoder_proj_weight`, biases f32). xp16=None: xp = bf16(x16 + pos16[row % len(pos16)]) is formed
    inside the kernel from the bf16 table pos16. value_head_major (x16 (B, N, 256)): value comes back (B, 8, N, 32)."""
    C = x16.shape[-1]
    M = x16.numel() // C
    hm_rows = 0
    if value_head_major:
        if x16.dim() != 3 or bv.numel() != 256:
            raise CggError('encoder_proj: value_head_major needs x16 (B, N, 256) and 256 value columns')
        hm_rows = x16.shape[1]
        value = torch.empty((x16.shape[0], 8, hm_rows, 32), dtype=torch.bfloat16, device=x16.device)
    else:
        value = torch.empty(x16.shape[:-1] + (bv.numel(),), dtype=torch.bfloat16, device=x16.device)
    offs = torch.empty(x16.shape[:-1] + (bc.numel(),), dtype=torch.bfloat16, device=x16.device)
    with _timed('encoder_proj'):
        rc = _lib_().cgg_encoder_proj_bf16(
            dev_ptr(x16, 'x16', torch.bfloat16), dev_ptr(xp16, 'xp16', torch.bfloat16), dev_ptr(pos16, 'pos16', torch.bfloat16),
            pos16.shape[0] if pos16 is not None else 0, dev_ptr(wvp), dev_ptr(bv, 'bv', torch.float32), dev_ptr(wcp),
            dev_ptr(bc, 'bc', torch.float32), dev_ptr(value), dev_ptr(offs), M, C, bv.numel(), bc.numel(), hm_rows,
            stream_ptr(x16.device))
    check(rc, 'cgg_encoder_proj_bf16')
    return value, offs


def encoder_ffn_ln(x16, w1p, b1, w2p, b2, gamma, beta, eps=1e-5, pos=None, want_f32=False, want_bf16=True, want_pos=False):
    """LN(x + W2 relu(W1 x + b1) + b2) on (..., 256) bf16 rows in ONE launch (hidden activation stays on chip); w1p / w2p
    from `pack_linear_weight`. Returns (y f32 | None, bf16(y) | None, bf16(y + pos[row % len(pos)]) | None)."""
    C = x16.shape[-1]
    M = x16.numel() // C
    F_ = b1.numel()
    y32 = torch.empty(x16.shape, dtype=torch.float32, device=x16.device) if want_f32 else None
    y16 = torch.empty(x16.shape, dtype=torch.bfloat16, device=x16.device) if want_bf16 else None
    yp16 = torch.empty(x16.shape, dtype=torch.bfloat16, device=x16.device) if want_pos else None
    rc = _lib_().cgg_encoder_ffn_ln_bf16(
        dev_ptr(x16, 'x16', torch.bfloat16), dev_ptr(w1p), dev_ptr(b1, 'b1', torch.float32), dev_ptr(w2p),
        dev_ptr(b2, 'b2', torch.float32), dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(beta, 'beta', torch.float32),
        float(eps), dev_ptr(pos, 'pos', torch.float32), pos.shape[0] if pos is not None else 0, dev_ptr(y16), dev_ptr(yp16),
        dev_ptr(y32), M, C, F_, stream_ptr(x16.device))
    check(rc, 'cgg_encoder_ffn_ln_bf16')
    return y32, y16, yp16


def encoder_ffn_ln_kv(x16, w1p, b1, w2p, b2, gamma, beta, eps, shift, pos, level_start, want_f32=True):
    """Last encoder layer: `encoder_ffn_ln` whose LayerNorm also emits the query decoder's K / V operands like
    `add_layernorm_kv` (x16 (B, S, 256) bf16). Returns (y f32 | None, m16, mp16), the bf16 pair level-major."""
    B, S, C = x16.shape
    y32 = torch.empty(x16.shape, dtype=torch.float32, device=x16.device) if want_f32 else None
    m16 = torch.empty((B * S, C), dtype=torch.bfloat16, device=x16.device)
    mp16 = torch.empty((B * S, C), dtype=torch.bfloat16, device=x16.device)
    rc = _lib_().cgg_encoder_ffn_ln_kv_bf16(
        dev_ptr(x16, 'x16', torch.bfloat16), dev_ptr(w1p), dev_ptr(b1, 'b1', torch.float32), dev_ptr(w2p),
        dev_ptr(b2, 'b2', torch.float32), dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(beta, 'beta', torch.float32),
        float(eps), dev_ptr(shift, 'shift', torch.float32), dev_ptr(pos, 'pos', torch.float32), S, _int_array(level_start),
        len(level_start), dev_ptr(y32), dev_ptr(m16), dev_ptr(mp16), B * S, C, b1.numel(), stream_ptr(x16.device))
    check(rc, 'cgg_encoder_ffn_ln_kv_bf16')
    return y32, m16, mp16


def encoder_layer_tail(a16, x16, wop, bo, norm0, w1p, b1, w2p, b2, norm1, pos=None, kv=None, want_f32=False, want_bf16=True,
                       want_pos=False):
    """Post-attention half of an encoder layer in ONE launch: x1 = LN0(x16 + a16 Wo^T + bo), y = LN1(x1 + FFN(x1)).
    norm0 / norm1 = (gamma, beta, eps); weights from `pack_linear_weight`. kv=None: returns (y f32 | None, bf16(y) | None,
    bf16(y + pos[row % len(pos)]) | None) like `encoder_ffn_ln`; kv=(shift, pos, level_start) (x16 (B, S, 256)): returns
    (y f32 | None, m16, mp16) like `encoder_ffn_ln_kv`."""
    C = x16.shape[-1]
    M = x16.numel() // C
    dev = x16.device
    y32 = torch.empty(x16.shape, dtype=torch.float32, device=dev) if want_f32 else None
    if kv is not None:
        shift, pos, level_start = kv
        S = x16.shape[-2]
        y16 = torch.empty((M, C), dtype=torch.bfloat16, device=dev)
        yp16 = torch.empty((M, C), dtype=torch.bfloat16, device=dev)
        pos_rows, ls, nl = S, _int_array(level_start), len(level_start)
    else:
        shift, ls, nl = None, None, 0
        y16 = torch.empty(x16.shape, dtype=torch.bfloat16, device=dev) if want_bf16 else None
        yp16 = torch.empty(x16.shape, dtype=torch.bfloat16, device=dev) if want_pos else None
        pos_rows = pos.shape[0] if pos is not None else 0
    with _timed('encoder_tail_kv' if kv is not None else 'encoder_tail'):
        rc = _lib_().cgg_encoder_layer_tail_bf16(
            dev_ptr(a16, 'a16', torch.bfloat16), dev_ptr(x16, 'x16', torch.bfloat16), dev_ptr(wop),
            dev_ptr(bo, 'bo', torch.float32), dev_ptr(norm0[0], 'gamma0', torch.float32),
            dev_ptr(norm0[1], 'beta0', torch.float32), float(norm0[2]), dev_ptr(w1p), dev_ptr(b1, 'b1', torch.float32),
            dev_ptr(w2p), dev_ptr(b2, 'b2', torch.float32), dev_ptr(norm1[0], 'gamma1', torch.float32),
            dev_ptr(norm1[1], 'beta1', torch.float32), float(norm1[2]), dev_ptr(pos, 'pos', torch.float32), pos_rows,
            dev_ptr(shift, 'shift', torch.float32), ls, nl, dev_ptr(y16), dev_ptr(yp16), dev_ptr(y32), M, C, b1.numel(),
            stream_ptr(dev))
    check(rc, 'cgg_encoder_layer_tail_bf16')
    return y32, y16, yp16


def add_layernorm_backward(dy, a, b, gamma, eps, want_bf16=False, dy16a=None, dy16b=None, want_amax=False):
    """Backward of LN(a + b) (b None: LN(a)) over the last dim (256) for the upstream gradient dy (f32) + dy16a + dy16b (bf16; any
    subset): returns (dx f32, bf16(dx) | None, dgamma, dbeta); d/da = d/db = dx. want_amax: -> (..., max |dx| as a device scalar),
    the pre-scale of the x3 contractions that consume dx as grad_output (no `absmax` pass)."""
    N = a.shape[-1]
    rows = a.numel() // N
    dx = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    dx16 = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device) if want_bf16 else None
    nb = _lib_().cgg_add_layernorm_backward_partials(rows)
    partial = torch.empty((nb, 2 * N), dtype=torch.float32, device=a.device)
    bdt = CGG_BF16 if (b is not None and b.dtype == torch.bfloat16) else CGG_F32
    if want_amax:
        amax = torch.empty(1, dtype=torch.float32, device=a.device)
        rc = _lib_().cgg_add_layernorm_backward_amax(dev_ptr(dy, 'dy', torch.float32), dev_ptr(dy16a, 'dy16a', torch.bfloat16),
                                                     dev_ptr(dy16b, 'dy16b', torch.bfloat16), dev_ptr(a, 'a', torch.float32),
                                                     dev_ptr(b, 'b'), bdt, dev_ptr(gamma, 'gamma', torch.float32), float(eps),
                                                     dev_ptr(dx), dev_ptr(dx16), dev_ptr(partial), dev_ptr(amax), rows, N,
                                                     stream_ptr(a.device))
        check(rc, 'cgg_add_layernorm_backward_amax')
        sums = partial.sum(0)
        return dx, dx16, sums[:N], sums[N:], amax
    rc = _lib_().cgg_add_layernorm_backward(dev_ptr(dy, 'dy', torch.float32), dev_ptr(dy16a, 'dy16a', torch.bfloat16),
                                            dev_ptr(dy16b, 'dy16b', torch.bfloat16), dev_ptr(a, 'a', torch.float32),
                                            dev_ptr(b, 'b'), bdt, dev_ptr(gamma, 'gamma', torch.float32), float(eps),
                                            dev_ptr(dx), dev_ptr(dx16), dev_ptr(partial), rows, N, stream_ptr(a.device))
    check(rc, 'cgg_add_layernorm_backward')
    sums = partial.sum(0)
    return dx, dx16, sums[:N], sums[N:]


class _AddLayerNormFn(torch.autograd.Function):
    """(y, bf16(y) | None, bf16(y + pos) | None) = LayerNorm(a + b) (a f32, b f32 | bf16, 256 channels) with the one-pass HIP
    forward and backward; the bf16 outputs feed the next bf16 GEMMs directly (no cast pass) and their bf16 gradients are summed
    with dy inside the backward kernel (no accumulation pass)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps, pos, want16, wantp):
        a = a.contiguous()
        b = b.contiguous()
        y, y16, yp16 = add_layernorm_stream(a, b, gamma, beta, eps, pos=pos if wantp else None, want_f32=True,
                                            want_bf16=bool(want16), want_pos=bool(wantp))
        ctx.save_for_backward(a, b, gamma)
        ctx.eps = eps
        ctx.pos_shape = tuple(pos.shape) if (pos is not None and wantp) else None
        return y, y16, yp16

    @staticmethod
    def backward(ctx, gy, gy16, gyp16):
        a, b, gamma = ctx.saved_tensors
        c = lambda t, dt: None if t is None else t.contiguous().to(dt)
        dx, dx16, dgamma, dbeta = add_layernorm_backward(c(gy, torch.float32), a, b, gamma, ctx.eps,
                                                         want_bf16=b.dtype == torch.bfloat16 and ctx.needs_input_grad[1],
                                                         dy16a=c(gy16, torch.bfloat16), dy16b=c(gyp16, torch.bfloat16))
        gb = None
        if ctx.needs_input_grad[1]:
            gb = dx16 if b.dtype == torch.bfloat16 else dx
        gpos = None
        if ctx.pos_shape is not None and gyp16 is not None and ctx.needs_input_grad[5]:
            # yp = y + pos[row % len(pos)]: the table's gradient is the sum of the bf16 gradient over the repeats (the batch)
            gpos = gyp16.reshape((-1,) + ctx.pos_shape).sum(0, dtype=torch.float32)
        return (dx if ctx.needs_input_grad[0] else None), gb, dgamma, dbeta, None, gpos, None, None


def add_layernorm_train_ok(a, b, norm):
    return (a.is_cuda and a.dtype == torch.float32 and a.shape[-1] == 256 and b.shape == a.shape
            and b.dtype in (torch.float32, torch.bfloat16) and tuple(norm.normalized_shape) == (256,) and norm.elementwise_affine
            and norm.bias is not None)


def add_layernorm_train(a, b, norm, pos=None, want_bf16=False, want_pos=False):
    """LayerNorm(a + b) of an `nn.LayerNorm(256)` for the training step (autograd-aware). Returns y, or with want_bf16 / want_pos
    the triple (y, bf16(y) | None, bf16(y + pos[row % len(pos)]) | None)."""
    y, y16, yp16 = _AddLayerNormFn.apply(a, b, norm.weight, norm.bias, norm.eps, pos, want_bf16, want_pos)
    if want_bf16 or want_pos:
        return y, y16, yp16
    return y


def add_layernorm_kv(a, b, gamma, beta, eps, shift, pos, level_start, want_f32=True):
    """Last encoder LayerNorm of the inference stream: y = LN(a + b) (a (B, S, 256) f32, b f32|bf16|None) plus the
    query decoder's K / V operands m16 = bf16(y + shift[s]), mp16 = bf16(y + shift[s] + pos[s]) (shift, pos (S, 256)
    f32), both LEVEL-MAJOR: (B * S, 256) with level l = rows [B * start_l, B * start_{l+1}) laid out (B, hw_l, 256).
    Returns (y f32 | None, m16, mp16)."""
    B, S, N = a.shape
    y32 = torch.empty(a.shape, dtype=torch.float32, device=a.device) if want_f32 else None
    m16 = torch.empty((B * S, N), dtype=torch.bfloat16, device=a.device)
    mp16 = torch.empty((B * S, N), dtype=torch.bfloat16, device=a.device)
    adt = CGG_BF16 if a.dtype == torch.bfloat16 else CGG_F32
    bdt = CGG_BF16 if (b is not None and b.dtype == torch.bfloat16) else CGG_F32
    rc = _lib_().cgg_add_layernorm_kv(
        dev_ptr(a, 'a'), adt, dev_ptr(b, 'b'), bdt, dev_ptr(gamma, 'gamma', torch.float32),
        dev_ptr(beta, 'beta', torch.float32), dev_ptr(shift, 'shift', torch.float32), dev_ptr(pos, 'pos', torch.float32),
        S, _int_array(level_start), len(level_start), dev_ptr(y32), dev_ptr(m16), dev_ptr(mp16), B * S, N, float(eps),
        stream_ptr(a.device))
    check(rc, 'cgg_add_layernorm_kv')
    return y32, m16, mp16


def bias_act_nhwc_(y, bias=None, res=None, relu=True):
    """In place on a channel-last bf16 activation `y` (..., C): y <- act(y + bias[C] + res)."""
    C = y.shape[-1]
    rc = _lib_().cgg_bias_act_nhwc(dev_ptr(y, 'y', torch.bfloat16), dev_ptr(bias, 'bias', torch.bfloat16),
                                   dev_ptr(res, 'res', torch.bfloat16), y.numel() // C, C, int(bool(relu)),
                                   stream_ptr(y.device))
    check(rc, 'cgg_bias_act_nhwc')
    return y


def _ptr_at(t, elem_offset=0):
    """Raw device address `elem_offset` elements into tensor `t` (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CggError(f'tensor must live on a ROCm device (got {t.device})')
    return ctypes.c_void_p(t.data_ptr() + int(elem_offset) * t.element_size())


def group_norm_nhwc_workspace(B, HW, groups, device):
    """f32 scratch for `group_norm_nhwc` (totals + per-block partial sums)."""
    n = _lib_().cgg_group_norm_nhwc_workspace_bytes(int(B), int(HW), int(groups))
    return torch.empty((max(n // 4, 1),), dtype=torch.float32, device=device)


def group_norm_nhwc(x, gamma, beta, groups, eps, ws, relu=False, up=None, W=0, out32=None, out16=None, pos=None,
                    outp16=None):
    """GroupNorm of a channel-last bf16 activation x (B, HW, C), C / groups == 8 (see include/cgg_hip.h).
      up     = (f32 tensor, element offset, batch stride, h, w): low-res NHWC map, bilinearly up-sampled and added
      out32  = (f32 tensor, element offset, batch stride) destination of y
      out16  / outp16 = (bf16 tensor, element offset, batch stride) destinations of bf16(y) / bf16(y + pos)
      pos    = (f32 tensor, element offset) rows [HW, C]
    Destinations are raw (tensor, offset, stride) triples so the three encoder levels can land directly inside the
    (B, N, C) stream tensors."""
    B, HW, C = x.shape
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise CggError(f'group_norm_nhwc: x dtype {x.dtype}')
    fn = _lib_().cgg_group_norm_nhwc if x.dtype == torch.bfloat16 else _lib_().cgg_group_norm_nhwc_f32
    need = _lib_().cgg_group_norm_nhwc_workspace_bytes(B, HW, int(groups))
    if ws is None or ws.numel() * ws.element_size() < need:
        raise CggError(f'group_norm_nhwc: workspace too small ({need} bytes needed; see group_norm_nhwc_workspace)')
    b16 = out16[2] if out16 is not None else (outp16[2] if outp16 is not None else 0)
    if out16 is not None and outp16 is not None and out16[2] != outp16[2]:
        raise CggError('group_norm_nhwc: out16 and outp16 must share the batch stride')
    rc = fn(
        dev_ptr(x, 'x', x.dtype), dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(beta, 'beta', torch.float32),
        dev_ptr(ws, 'ws', torch.float32), B, HW, C, int(groups), float(eps), int(bool(relu)),
        _ptr_at(up[0], up[1]) if up is not None else None, up[3] if up is not None else 0,
        up[4] if up is not None else 0, up[2] if up is not None else 0, int(W),
        _ptr_at(out32[0], out32[1]) if out32 is not None else None, out32[2] if out32 is not None else 0,
        _ptr_at(out16[0], out16[1]) if out16 is not None else None,
        _ptr_at(pos[0], pos[1]) if pos is not None else None,
        _ptr_at(outp16[0], outp16[1]) if outp16 is not None else None, b16, stream_ptr(x.device))
    check(rc, 'cgg_group_norm_nhwc')


def group_norm_nhwc_x3a(x, gamma, beta, groups, eps, ws, out, relu=False, up=None, W=0, pos=None, outp=None):
    """GroupNorm of a channel-last F32 map x (B, HW, C) with the result written as x3a rows (csrc/x3.h):
      out  = (tensor, element offset, batch stride) destination of y (may alias x);
      outp = (tensor, element offset) destination of y + pos (same batch stride), pos = (f32 tensor, element offset) rows [HW, C];
      up   = (x3a tensor, element offset, batch stride, h, w): low-res x3a map, bilinearly up-sampled and added before the ReLU."""
    B, HW, C = x.shape
    if x.dtype != torch.float32:
        raise CggError(f'group_norm_nhwc_x3a: x dtype {x.dtype}')
    need = _lib_().cgg_group_norm_nhwc_workspace_bytes(B, HW, int(groups))
    if ws is None or ws.numel() * ws.element_size() < need:
        raise CggError(f'group_norm_nhwc_x3a: workspace too small ({need} bytes needed; see group_norm_nhwc_workspace)')
    rc = _lib_().cgg_group_norm_nhwc_f32_x3a(
        dev_ptr(x, 'x', torch.float32), dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(beta, 'beta', torch.float32),
        dev_ptr(ws, 'ws', torch.float32), B, HW, C, int(groups), float(eps), int(bool(relu)),
        _ptr_at(up[0], up[1]) if up is not None else None, up[3] if up is not None else 0,
        up[4] if up is not None else 0, up[2] if up is not None else 0, int(W),
        _ptr_at(out[0], out[1]), out[2], _ptr_at(pos[0], pos[1]) if pos is not None else None,
        _ptr_at(outp[0], outp[1]) if outp is not None else None, stream_ptr(x.device))
    check(rc, 'cgg_group_norm_nhwc_f32_x3a')


def zero_border_map(B, H, W, C, device):
    """(B, H + 2, W + 2, C) f32 channel-last map whose one-pixel border is zero and whose interior is uninitialised (four thin fills)."""
    y = torch.empty((B, H + 2, W + 2, C), dtype=torch.float32, device=device)
    y[:, 0].zero_()
    y[:, H + 1].zero_()
    y[:, 1:H + 1, 0].zero_()
    y[:, 1:H + 1, W + 1].zero_()
    return y


def group_norm_nhwc_padout(x, gamma, beta, groups, eps, ws, hw, y_padded, relu=False, lo=None, lo_hw=None):
    """`group_norm_nhwc` of f32 rows x (B, HW, C) (+ bilinear up-sample of lo (B, lo_h lo_w, C)) with y written into the interior of
    `y_padded` (B, H + 2, W + 2, C) (`zero_border_map`): the x3 training convolution's padded input without a padding copy."""
    B, HW, C = x.shape
    H, W = int(hw[0]), int(hw[1])
    if x.dtype != torch.float32 or not x.is_contiguous() or tuple(y_padded.shape) != (B, H + 2, W + 2, C) or not y_padded.is_contiguous():
        raise CggError('group_norm_nhwc_padout: x (B, HW, C) float32 contiguous and y_padded (B, H + 2, W + 2, C) expected')
    if lo is not None and (not lo.is_contiguous() or lo.dtype != torch.float32):
        raise CggError('group_norm_nhwc_padout: lo must be contiguous float32 rows')
    check(_lib_().cgg_group_norm_nhwc_f32_padout(
        dev_ptr(x), dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(beta, 'beta', torch.float32), dev_ptr(ws), B, HW, C, int(groups),
        float(eps), int(bool(relu)), dev_ptr(lo) if lo is not None else None, int(lo_hw[0]) if lo is not None else 0,
        int(lo_hw[1]) if lo is not None else 0, int(lo.shape[1]) * C if lo is not None else 0, W, dev_ptr(y_padded),
        stream_ptr(x.device)), 'cgg_group_norm_nhwc_f32_padout')


def group_norm_nhwc_backward(x, y, dy, stats, gamma, groups, eps, relu, hw, lo_hw=None, dx_padded=None):
    """Backward of the channel-last f32 GroupNorm (`cgg_group_norm_nhwc_f32_backward`): x, dy (and y when relu) (B, HW, C), stats
    (B * groups * 2) = the forward workspace's head. -> (dx, dgamma, dbeta, dlo | None); dx_padded: a `zero_border_map` that receives
    dx in its interior (then returned as dx)."""
    B, HW, C = x.shape
    lib = _lib_()
    ws = torch.empty(max(lib.cgg_group_norm_nhwc_backward_workspace_bytes(B, HW, int(groups)) // 4, 1), dtype=torch.float32, device=x.device)
    dx = dx_padded if dx_padded is not None else torch.empty_like(x)
    tot = torch.empty((B, int(groups), 16), dtype=torch.float32, device=x.device)
    dlo = torch.empty((B, lo_hw[0] * lo_hw[1], C), dtype=torch.float32, device=x.device) if lo_hw is not None else None
    check(lib.cgg_group_norm_nhwc_f32_backward(dev_ptr(x), dev_ptr(y) if relu else None, dev_ptr(dy), dev_ptr(stats),
                                               dev_ptr(gamma, 'gamma', torch.float32), dev_ptr(ws), B, HW, C, int(groups), float(eps),
                                               int(bool(relu)), dev_ptr(dx), dev_ptr(tot), dev_ptr(dlo) if dlo is not None else None,
                                               lo_hw[0] if lo_hw else 0, lo_hw[1] if lo_hw else 0, int(hw[1]),
                                               int(dx_padded is not None), stream_ptr(x.device)), 'cgg_group_norm_nhwc_f32_backward')
    t = tot.sum(0)                                        # (groups, 16)
    return dx, t[:, :8].reshape(C), t[:, 8:].reshape(C), dlo


class GroupNormRowsFn(torch.autograd.Function):
    """y = act(GroupNorm(x) [+ bilinear up-sample of `lo`]) on CHANNEL-LAST f32 rows for the training step (parity mode;
    `cgg_group_norm_nhwc_f32` forward, `cgg_group_norm_nhwc_f32_backward`): x (B, HW, C) with C / groups == 8, lo (B, lo_h * lo_w, C) |
    None (the [3P] lateral ConvModule has no activation: relu only without `lo`, the output ConvModule). Returns y (B, HW, C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu, lo, hw, lo_hw):
        B, HW, C = x.shape
        x = x.contiguous()
        ws = group_norm_nhwc_workspace(B, HW, groups, x.device)
        y = torch.empty_like(x)
        up = None
        if lo is not None:
            lo = lo.contiguous()
            up = (lo, 0, lo.shape[1] * C, int(lo_hw[0]), int(lo_hw[1]))
        group_norm_nhwc(x, gamma.detach(), beta.detach(), groups, eps, ws, relu=relu, up=up, W=int(hw[1]), out32=(y, 0, HW * C))
        stats = ws[:B * groups * 2].clone()
        ctx.save_for_backward(x, y if relu else x.new_empty(0), stats, gamma)
        ctx.cfg = (int(groups), float(eps), bool(relu), tuple(hw), tuple(lo_hw) if lo is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, stats, gamma = ctx.saved_tensors
        groups, eps, relu, hw, lo_hw = ctx.cfg
        dx, dgamma, dbeta, dlo = group_norm_nhwc_backward(x, y, gy.contiguous(), stats, gamma, groups, eps, relu, hw, lo_hw)
        return dx, dgamma, dbeta, None, None, None, dlo, None, None


def pack_mask_feature_nhwc(feat, pool=1):
    """feat (B, H, W, C) bf16 channel-last -> PackedFeature (hi image only; throughput mode)."""
    B, H, W, C = feat.shape
    if H % pool or W % pool:
        raise CggError(f'pack_mask_feature_nhwc: {H}x{W} not divisible by pool={pool}')
    h, w = H // pool, W // pool
    T = (h * w + 31) // 32
    hi = torch.empty((B, T, C // 8, 32, 8), dtype=torch.bfloat16, device=feat.device)
    rc = _lib_().cgg_pack_mask_feature_nhwc(dev_ptr(feat, 'mask_feature', torch.bfloat16), dev_ptr(hi), B, C, H, W,
                                            int(pool), stream_ptr(feat.device))
    check(rc, 'cgg_pack_mask_feature_nhwc')
    return PackedFeature(hi, None, B, C, h, w)


def pack_mask_feature_nhwc_x3(feat, pools):
    """feat (B, H, W, C) F32 channel-last -> [PackedFeature with hi / lo x3 images for each pool in `pools`] (<= 4) from ONE
    launch: parity mode's packed mask feature (split mode of `mask_logits`)."""
    B, H, W, C = feat.shape
    if feat.dtype != torch.float32 or not feat.is_contiguous():
        raise CggError('pack_mask_feature_nhwc_x3: feat must be a contiguous (B, H, W, C) float32 tensor')
    outs, hp, lp = [], [], []
    for pool in pools:
        if H % pool or W % pool:
            raise CggError(f'pack_mask_feature_nhwc_x3: {H}x{W} not divisible by pool={pool}')
        h, w = H // pool, W // pool
        hi = torch.empty((B, (h * w + 31) // 32, C // 8, 32, 8), dtype=torch.bfloat16, device=feat.device)
        lo = torch.empty_like(hi)
        outs.append(PackedFeature(hi, lo, B, C, h, w))
        hp.append(hi.data_ptr())
        lp.append(lo.data_ptr())
    n = len(pools)
    rc = _lib_().cgg_pack_mask_feature_nhwc_f32_x3(dev_ptr(feat, 'mask_feature', torch.float32), (ctypes.c_void_p * n)(*hp),
                                                   (ctypes.c_void_p * n)(*lp), _int_array([int(p) for p in pools]), n,
                                                   B, C, H, W, stream_ptr(feat.device))
    check(rc, 'cgg_pack_mask_feature_nhwc_f32_x3')
    return outs


def pack_mask_feature_nhwc_multi(feat, pools):
    """feat (B, H, W, C) bf16 channel-last -> [PackedFeature for each pool in `pools`] (<= 4) from ONE launch."""
    B, H, W, C = feat.shape
    outs, ptrs = [], []
    for pool in pools:
        if H % pool or W % pool:
            raise CggError(f'pack_mask_feature_nhwc: {H}x{W} not divisible by pool={pool}')
        h, w = H // pool, W // pool
        hi = torch.empty((B, (h * w + 31) // 32, C // 8, 32, 8), dtype=torch.bfloat16, device=feat.device)
        outs.append(PackedFeature(hi, None, B, C, h, w))
        ptrs.append(hi.data_ptr())
    n = len(pools)
    rc = _lib_().cgg_pack_mask_feature_nhwc_multi(dev_ptr(feat, 'mask_feature', torch.bfloat16),
                                                  (ctypes.c_void_p * n)(*ptrs), _int_array([int(p) for p in pools]), n,
                                                  B, C, H, W, stream_ptr(feat.device))
    check(rc, 'cgg_pack_mask_feature_nhwc_multi')
    return outs


# ------------------------------------------------------------------------------------------------
# throughput-mode query-side linear (packed bf16 weights, fused LayerNorm / `+ pos` / split-K)
# ------------------------------------------------------------------------------------------------
class X3Image(torch.Tensor):
    """uint8 buffer holding the x3 image of a weight (csrc/x3.h): the query-side wrappers route to the *_x3 kernels when they
    are handed one instead of a bf16 packed buffer."""

    def __deepcopy__(self, memo):       # cached images sit in module __dict__s that get deep-copied with the module
        return self.as_subclass(torch.Tensor).clone().as_subclass(X3Image)


def is_x3(packed):
    return isinstance(packed, X3Image)


def _x3_check(packed, N, K, who):
    """An x3 image carries no shape: its byte size must be the one `cgg_x3_packed_bytes(N, K)` prescribes (a wrong
    (N, K) made the kernel read past the image)."""
    want = _lib_().cgg_x3_packed_bytes(int(N), int(K))
    if want <= 0 or packed.numel() * packed.element_size() != want:
        raise CggError(f'{who}: the x3 image holds {packed.numel() * packed.element_size()} bytes, an (N={N}, K={K}) weight needs {want}')


def _x3_fn(name, *packed):
    """C entry point `cgg_<name>_bf16` or its f32-class twin `cgg_<name>_x3`; all packed operands must be of one kind."""
    kinds = {is_x3(p) for p in packed if p is not None}
    if len(kinds) > 1:
        raise CggError(f'{name}: bf16 and x3 packed weights mixed in one call')
    return getattr(_lib_(), f'cgg_{name}_x3' if True in kinds else f'cgg_{name}_bf16'), ('x3' if True in kinds else 'bf16')


def pack_linear_weight_x3(weight):
    """weight (N, K) f32 -> x3 image (hi / lo f16 MFMA-B fragments + per-column scales) for the parity-mode kernels."""
    N, K = weight.shape
    nbytes = _lib_().cgg_x3_packed_bytes(N, K)
    if nbytes <= 0:
        raise CggError(f'pack_linear_weight_x3: unsupported shape {tuple(weight.shape)} (K % 16)')
    out = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
    w = weight.detach().float().contiguous()
    rc = _lib_().cgg_x3_pack(dev_ptr(w, 'weight', torch.float32), dev_ptr(out), N, K, stream_ptr(weight.device))
    check(rc, 'cgg_x3_pack')
    return out.as_subclass(X3Image)


def pack_conv_weight_x3(weight):
    """conv filter (N, C, KH, KW) -> x3 image with k = (ky, kx, c), the order `conv_x3_nhwc` walks a channel-last map in."""
    N = weight.shape[0]
    return pack_linear_weight_x3(weight.detach().float().permute(0, 2, 3, 1).reshape(N, -1))


def topk_select(x, k):
    """x (rows, N) float32 ROCm (last dim contiguous) -> (rows, k) int64: the indices of each row's k largest values as a SET
    (no order, ties at the threshold arbitrary) -- csrc/topk_select.hip, a radix select instead of torch.topk's sort."""
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.stride(1) != 1 or not 0 < k <= x.shape[1]:
        raise CggError('topk_select: (rows, N) float32 ROCm matrix with a contiguous last dim and 0 < k <= N expected')
    rows, N = x.shape
    out = torch.empty((rows, k), dtype=torch.int64, device=x.device)
    if rows:
        check(_lib_().cgg_topk_select(ctypes.c_void_p(x.data_ptr()), int(x.stride(0)), rows, N, int(k), dev_ptr(out),
                                      stream_ptr(x.device)), 'cgg_topk_select')
    return out


def absmax(x):
    """max |x| of a float32 ROCm matrix / tensor (last dim contiguous, % 4) -> device scalar (1,) f32: the per-tensor pre-scale
    of the x3 contractions' grad_output operands (`gemm_x3(..., amax=)`, `wgrad_x3(..., amax=)`; csrc/x3.h). Exact: one streaming
    pass over the tensor (a sampled maximum is NOT safe for gradients -- they are sparse, see runtime._X3LinearFn.backward)."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise CggError('absmax: float32 ROCm tensor expected')
    x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x
    if x2.dim() != 2 or x2.stride(1) != 1:
        raise CggError('absmax: a 2-D view with a contiguous last dim expected')
    if x2.is_contiguous() and x2.numel() % 4 == 0:
        x2 = x2.view(1, -1)                       # dense: one long row
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    M, N = x2.shape
    check(_lib_().cgg_absmax_f32(ctypes.c_void_p(x2.data_ptr()), int(x2.stride(0)) if M > 1 else N, M, N, dev_ptr(out),
                                 stream_ptr(x.device)), 'cgg_absmax_f32')
    return out


def relu_backward_absmax(gy, y):
    """(g, amax): g = gy where y > 0 else 0 (the ReLU whose output is y), amax = max |g| as a device scalar -- one pass
    (`cgg_relu_bwd_absmax_f32`) instead of threshold_backward + `absmax`. gy, y contiguous float32 ROCm tensors of one shape, numel % 4 == 0."""
    if gy.dtype != torch.float32 or y.dtype != torch.float32 or not gy.is_cuda or gy.shape != y.shape or not gy.is_contiguous() \
            or not y.is_contiguous() or gy.numel() % 4 or gy.numel() == 0:
        raise CggError('relu_backward_absmax: two contiguous float32 ROCm tensors of one shape with numel % 4 == 0 expected')
    g = torch.empty_like(gy)
    amax = torch.empty(1, dtype=torch.float32, device=gy.device)
    check(_lib_().cgg_relu_bwd_absmax_f32(dev_ptr(gy), dev_ptr(y), dev_ptr(g), gy.numel(), dev_ptr(amax), stream_ptr(gy.device)),
          'cgg_relu_bwd_absmax_f32')
    return g, amax


def gemm_x3(a, packed, N, bias=None, res=None, relu=False, out=None, amax=None):
    """a (M, K) f32 rows (row stride free, last dim contiguous) x x3 image -> act(a W^T + bias (+ res)) (M, N) f32:
    parity mode's large linear (csrc/x3_gemm.hip). amax: device scalar max |a| (`absmax`) -> a is pre-scaled per tensor instead
    of by the fixed 2^4 (operands that are not unit scale: gradients)."""
    if a.dim() != 2 or a.stride(1) != 1 or a.dtype != torch.float32 or not a.is_cuda or not is_x3(packed):
        raise CggError('gemm_x3: a must be a 2-D float32 ROCm tensor with a contiguous last dim, packed an x3 image')
    M, K = a.shape
    _x3_check(packed, N, K, 'gemm_x3')
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    if y.dim() != 2 or y.stride(1) != 1 or y.shape != (M, N) or y.dtype != torch.float32:
        raise CggError('gemm_x3: bad `out` view')
    if res is not None and (res.dim() != 2 or res.stride(1) != 1 or res.shape != (M, N) or res.dtype != torch.float32):
        raise CggError('gemm_x3: bad `res` view')
    with _timed('gemm_x3', flops=2.0 * M * N * K, bytes=4.0 * (M * K + M * N * (2 if res is not None else 1)) + 4.0 * N * K,
                shape=(M, N, K)):
        if amax is not None:
            rc = _lib_().cgg_gemm_x3_scaled(ctypes.c_void_p(a.data_ptr()), a.stride(0), dev_ptr(amax, 'amax', torch.float32),
                                            dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
                                            ctypes.c_void_p(res.data_ptr()) if res is not None else None,
                                            res.stride(0) if res is not None else 0, ctypes.c_void_p(y.data_ptr()), y.stride(0), M, N,
                                            K, int(bool(relu)), stream_ptr(a.device))
        else:
            rc = _lib_().cgg_gemm_x3(ctypes.c_void_p(a.data_ptr()), a.stride(0), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
                                     ctypes.c_void_p(res.data_ptr()) if res is not None else None,
                                     res.stride(0) if res is not None else 0, ctypes.c_void_p(y.data_ptr()), y.stride(0), M, N, K,
                                     int(bool(relu)), stream_ptr(a.device))
    check(rc, 'cgg_gemm_x3')
    return y


def gemm_x3_bwd(g, packed, N, amax=None, mask=None, out=None, want_amax=False):
    """Backward-side x3 GEMM (csrc/x3_gemm.hip `cgg_gemm_x3_bwd`): out (M, N) = g (M, K) W^T with g pre-scaled per tensor (`amax`),
    the result zeroed where `mask` (M, N) <= 0 (ReLU backward of the layer whose output `mask` is) and, with want_amax, max |out|
    from the epilogue as a device scalar -> (out, out_amax | None)."""
    if g.dim() != 2 or g.stride(1) != 1 or g.dtype != torch.float32 or not g.is_cuda or not is_x3(packed):
        raise CggError('gemm_x3_bwd: g must be a 2-D float32 ROCm tensor with a contiguous last dim, packed an x3 image')
    M, K = g.shape
    _x3_check(packed, N, K, 'gemm_x3_bwd')
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=g.device)
    if y.dim() != 2 or y.stride(1) != 1 or y.shape != (M, N) or y.dtype != torch.float32:
        raise CggError('gemm_x3_bwd: bad `out` view')
    if mask is not None and (mask.dim() != 2 or mask.stride(1) != 1 or mask.shape != (M, N) or mask.dtype != torch.float32):
        raise CggError('gemm_x3_bwd: bad `mask` view')
    oa = torch.empty(1, dtype=torch.float32, device=g.device) if want_amax else None
    with _timed('gemm_x3', flops=2.0 * M * N * K, bytes=4.0 * (M * K + M * N * (2 if mask is not None else 1)) + 4.0 * N * K,
                shape=(M, N, K)):
        rc = _lib_().cgg_gemm_x3_bwd(ctypes.c_void_p(g.data_ptr()), g.stride(0), dev_ptr(amax, 'amax', torch.float32), dev_ptr(packed),
                                     ctypes.c_void_p(mask.data_ptr()) if mask is not None else None,
                                     mask.stride(0) if mask is not None else 0, ctypes.c_void_p(y.data_ptr()), y.stride(0),
                                     dev_ptr(oa), M, N, K, stream_ptr(g.device))
    check(rc, 'cgg_gemm_x3_bwd')
    return y, oa


def encoder_layer_tail_x3(a, x, wo, bo, norm0, w1, b1, w2, b2, norm1, pos=None, want_pos=False, x3a=False):
    """Parity mode's encoder layer tail in ONE launch: a (attention rows), x (layer input) (..., 256) f32 ->
    y = LN1(x1 + FFN(x1)), x1 = LN0(x + a Wo^T + bo) (and y + pos[row % len(pos)] when want_pos); wo / w1 / w2 x3 images,
    norm_* = (gamma, beta, eps). x3a=True: x and both outputs are x3a rows (csrc/x3.h), a stays f32."""
    C = a.shape[-1]
    M = a.numel() // C
    for t in (a, x):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != a.shape or not t.is_cuda:
            raise CggError('encoder_layer_tail_x3: a / x must be matching contiguous float32 ROCm tensors')
    if not (is_x3(wo) and is_x3(w1) and is_x3(w2)):
        raise CggError('encoder_layer_tail_x3: weights must be x3 images')
    F = b1.numel()
    y = torch.empty_like(a)
    yp = torch.empty_like(a) if want_pos else None
    with _timed('encoder_tail_x3', flops=2.0 * M * (C * C + 2 * C * F), bytes=4.0 * M * C * (3 + (1 if want_pos else 0)),
                shape=(M, C, F)):
        fn = _lib_().cgg_encoder_layer_tail_x3a if x3a else _lib_().cgg_encoder_layer_tail_x3
        rc = fn(
            dev_ptr(a), dev_ptr(x), dev_ptr(wo), dev_ptr(bo, 'bo', torch.float32), dev_ptr(norm0[0], 'gamma0', torch.float32),
            dev_ptr(norm0[1], 'beta0', torch.float32), float(norm0[2]), dev_ptr(w1), dev_ptr(b1, 'b1', torch.float32), dev_ptr(w2),
            dev_ptr(b2, 'b2', torch.float32), dev_ptr(norm1[0], 'gamma1', torch.float32), dev_ptr(norm1[1], 'beta1', torch.float32),
            float(norm1[2]), dev_ptr(pos, 'pos', torch.float32) if want_pos else None, pos.shape[0] if want_pos else 0, dev_ptr(y),
            dev_ptr(yp), M, C, int(F), stream_ptr(a.device))
    check(rc, 'cgg_encoder_layer_tail_x3')
    return y, yp


def transpose_f32(x):
    """x (B, R, C) contiguous f32 -> (B, C, R) contiguous (csrc/epilogue.hip: tiled through LDS); `nchw_to_nhwc` / `nhwc_to_nchw`
    are this on the (B, C, H W) / (B, H W, C) views."""
    if x.dim() != 3 or not x.is_contiguous() or x.dtype != torch.float32 or not x.is_cuda:
        raise CggError('transpose_f32: contiguous (B, R, C) float32 ROCm tensor expected')
    B, R, C = x.shape
    y = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    check(_lib_().cgg_transpose_f32(dev_ptr(x), dev_ptr(y), B, R, C, stream_ptr(x.device)), 'cgg_transpose_f32')
    return y


def nchw_to_nhwc(x):
    """(B, C, H, W) f32 (any strides) -> (B, H, W, C) contiguous."""
    B, C, H, W = x.shape
    if x.permute(0, 2, 3, 1).is_contiguous():
        return x.permute(0, 2, 3, 1)
    return transpose_f32(x.contiguous().view(B, C, H * W)).view(B, H, W, C)


def nchw_to_nhwc_pad1(x):
    """(B, C, H, W) f32 -> (B, H + 2, W + 2, C) channel-last with a ZERO one-pixel border (interior by the tiled transpose kernel,
    border by four thin fills): what `F.pad(nchw_to_nhwc(x), (0, 0, 1, 1, 1, 1))` returns, without the 1-GB padding copy."""
    if x.dim() != 4 or x.dtype != torch.float32 or not x.is_cuda:
        raise CggError('nchw_to_nhwc_pad1: (B, C, H, W) float32 ROCm tensor expected')
    B, C, H, W = x.shape
    x = x.contiguous()
    y = zero_border_map(B, H, W, C, x.device)
    check(_lib_().cgg_nchw_to_nhwc_pad1_f32(dev_ptr(x), dev_ptr(y), B, C, H, W, stream_ptr(x.device)), 'cgg_nchw_to_nhwc_pad1_f32')
    return y


def nhwc_to_nchw(x):
    """(B, H, W, C) contiguous f32 -> (B, C, H, W) contiguous."""
    B, H, W, C = x.shape
    return transpose_f32(x.view(B, H * W, C)).view(B, C, H, W)


def wgrad_x3(dy, x, want_bias=False, amax=None):
    """dW (N, K) = dy^T x for dy (M, N), x (M, K) f32 rows (contiguous last dim, row strides % 4 == 0; N, K % 4 == 0) on the
    f32-class f16 x 3 contraction with transpose reads (csrc/wgrad_x3.hip): the weight gradient of a linear layer.
    want_bias: -> (dW, db) with db (N) = dy.sum(0) from the same pass over dy. amax: device scalar max |dy| (`absmax`) -> dy is
    pre-scaled per tensor instead of by the fixed 2^4 (gradients are not unit scale, csrc/x3.h)."""
    if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != x.shape[0] or dy.dtype != torch.float32 or x.dtype != torch.float32 \
            or not dy.is_cuda or dy.stride(1) != 1 or x.stride(1) != 1:
        raise CggError('wgrad_x3: dy (M, N) and x (M, K) must be float32 ROCm matrices with contiguous rows')
    M, N = dy.shape
    K = x.shape[1]
    lib = _lib_()
    nbytes = lib.cgg_wgrad_x3_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device)
    splits = ctypes.c_int(0)
    wsb = torch.empty((nbytes // (4 * N * K)) * N, dtype=torch.float32, device=dy.device) if want_bias else None
    with _timed('wgrad_x3', flops=2.0 * M * N * K, bytes=4.0 * (M * N + M * K + N * K), shape=(M, N, K)):
        if amax is not None:
            rc = lib.cgg_wgrad_x3_scaled(ctypes.c_void_p(dy.data_ptr()), dy.stride(0), dev_ptr(amax, 'amax', torch.float32),
                                         ctypes.c_void_p(x.data_ptr()), x.stride(0), dev_ptr(ws), dev_ptr(wsb), ctypes.byref(splits),
                                         M, N, K, stream_ptr(dy.device))
        elif want_bias:
            rc = lib.cgg_wgrad_bias_x3(ctypes.c_void_p(dy.data_ptr()), dy.stride(0), ctypes.c_void_p(x.data_ptr()), x.stride(0),
                                       dev_ptr(ws), dev_ptr(wsb), ctypes.byref(splits), M, N, K, stream_ptr(dy.device))
        else:
            rc = lib.cgg_wgrad_x3(ctypes.c_void_p(dy.data_ptr()), dy.stride(0), ctypes.c_void_p(x.data_ptr()), x.stride(0), dev_ptr(ws),
                                  ctypes.byref(splits), M, N, K, stream_ptr(dy.device))
    check(rc, 'cgg_wgrad_x3')
    gw = ws.view(splits.value, N, K).sum(0) if splits.value > 1 else ws.view(N, K).clone()
    if want_bias:
        return gw, wsb.view(-1, N)[:splits.value].sum(0)
    return gw


def gemm_x3_split(a, packed, N, col2, bias=None, res_table=None):
    """a (M, K) f32 rows x x3 image of an (N, K) weight -> (y1 (M, col2), y2 (M, N - col2)) = column blocks of
    a W^T + bias + res_table[row % len(res_table)] (res_table (R, N) f32 | None), one launch (`cgg_gemm_x3_ex`)."""
    if a.dim() != 2 or a.stride(1) != 1 or a.dtype != torch.float32 or not a.is_cuda or not is_x3(packed):
        raise CggError('gemm_x3_split: a must be a 2-D float32 ROCm tensor with a contiguous last dim, packed an x3 image')
    M, K = a.shape
    _x3_check(packed, N, K, 'gemm_x3_split')
    y1 = torch.empty((M, col2), dtype=torch.float32, device=a.device)
    y2 = torch.empty((M, N - col2), dtype=torch.float32, device=a.device)
    if res_table is not None and (res_table.dim() != 2 or res_table.shape[1] != N or not res_table.is_contiguous()
                                  or res_table.dtype != torch.float32):
        raise CggError('gemm_x3_split: res_table must be a contiguous (R, N) float32 tensor')
    with _timed('gemm_x3', flops=2.0 * M * N * K, bytes=4.0 * (M * K + M * N + N * K), shape=(M, N, K)):
        rc = _lib_().cgg_gemm_x3_ex(ctypes.c_void_p(a.data_ptr()), a.stride(0), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
                                    dev_ptr(res_table), N if res_table is not None else 0,
                                    res_table.shape[0] if res_table is not None else 0, dev_ptr(y1), col2, dev_ptr(y2), N - col2,
                                    int(col2), M, N, K, 0, stream_ptr(a.device))
    check(rc, 'cgg_gemm_x3_ex')
    return y1, y2


def gemm_x3_table(a, packed, N, res_table, bias=None, out=None):
    """a (M, K) f32 rows x x3 image of an (N, K) weight -> a W^T + bias + res_table[row % len(res_table)] (M, N) f32, one launch
    (`cgg_gemm_x3_ex` with a row-periodic residual): a linear layer whose bias is a per-token table shared by the images of a batch."""
    if a.dim() != 2 or a.stride(1) != 1 or a.dtype != torch.float32 or not a.is_cuda or not is_x3(packed):
        raise CggError('gemm_x3_table: a must be a 2-D float32 ROCm tensor with a contiguous last dim, packed an x3 image')
    M, K = a.shape
    _x3_check(packed, N, K, 'gemm_x3_table')
    if res_table.dim() != 2 or res_table.shape[1] != N or not res_table.is_contiguous() or res_table.dtype != torch.float32 \
            or M % res_table.shape[0]:
        raise CggError('gemm_x3_table: res_table must be a contiguous (R, N) float32 tensor with R dividing the row count')
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    if y.shape != (M, N) or not y.is_contiguous() or y.dtype != torch.float32:
        raise CggError('gemm_x3_table: bad `out`')
    with _timed('gemm_x3', flops=2.0 * M * N * K, bytes=4.0 * (M * K + M * N + N * K), shape=(M, N, K)):
        rc = _lib_().cgg_gemm_x3_ex(ctypes.c_void_p(a.data_ptr()), a.stride(0), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
                                    dev_ptr(res_table), N, res_table.shape[0], dev_ptr(y), N, None, 0, 0, M, N, K, 0,
                                    stream_ptr(a.device))
    check(rc, 'cgg_gemm_x3_ex')
    return y


def conv_x3_nhwc(x, packed, N, kernel, stride=1, pad=0, bias=None, res=None, relu=False, amax=None):
    """x (B, H, W, C) f32 channel-last (contiguous) -> act(conv + bias (+ res)) (B, OH, OW, N) f32 as an implicit GEMM on the
    x3 image made by `pack_conv_weight_x3` (C % 32 == 0). amax: device scalar max |x| -> per-tensor pre-scale (gradient maps)."""
    if x.dim() != 4 or not x.is_contiguous() or x.dtype != torch.float32 or not x.is_cuda or not is_x3(packed):
        raise CggError('conv_x3_nhwc: x must be a contiguous (B, H, W, C) float32 ROCm tensor, packed an x3 image')
    B, H, W, C = x.shape
    KH, KW = (kernel, kernel) if isinstance(kernel, int) else kernel
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    _x3_check(packed, N, C * KH * KW, 'conv_x3_nhwc')
    y = torch.empty((B, OH, OW, N), dtype=torch.float32, device=x.device)
    if res is not None and (tuple(res.shape) != (B, OH, OW, N) or not res.is_contiguous() or res.dtype != torch.float32):
        raise CggError('conv_x3_nhwc: res must be a contiguous (B, OH, OW, N) float32 tensor')
    with _timed('gemm_x3', flops=2.0 * B * OH * OW * N * C * KH * KW,
                bytes=4.0 * (B * H * W * C + B * OH * OW * N * (2 if res is not None else 1) + N * C * KH * KW),
                shape=(B * OH * OW, N, C * KH * KW)):
        if amax is not None:
            rc = _lib_().cgg_conv_x3_nhwc_scaled(dev_ptr(x), dev_ptr(amax, 'amax', torch.float32), dev_ptr(packed),
                                                 dev_ptr(bias, 'bias', torch.float32), dev_ptr(res), dev_ptr(y), B, H, W, C, N, KH, KW,
                                                 int(stride), int(pad), int(bool(relu)), stream_ptr(x.device))
        else:
            rc = _lib_().cgg_conv_x3_nhwc(dev_ptr(x), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32), dev_ptr(res), dev_ptr(y),
                                          B, H, W, C, N, KH, KW, int(stride), int(pad), int(bool(relu)), stream_ptr(x.device))
    check(rc, 'cgg_conv_x3_nhwc')
    return y


# ---- round 4: pre-split activation rows ("x3a", csrc/x3.h) and the LDS-DMA GEMM that consumes them (csrc/x3s_gemm.hip) ----
X3A_F32, X3A_SPLIT = 1, 2


class X3ATensor(torch.Tensor):
    """float32-tagged tensor whose storage holds x3a rows (csrc/x3.h: per 8 channels [8 f16 hi | 8 f16 lo] of 16 x): the marker
    the stream modules hand each other (a ResNet's parity-mode outputs, the encoder memories). Views keep the tag; arithmetic on
    it is meaningless -- `x3a_to_f32` gives the values."""


def is_x3a(t):
    return isinstance(t, X3ATensor)


def as_x3a(t):
    return t.as_subclass(X3ATensor)


def x3a_to_f32(t):
    """values of an x3a-tagged tensor whose channel dim is contiguous in memory (e.g. the (B, C, H, W)-shaped permuted view of a
    channel-last map) -> plain float32 tensor of the same shape."""
    base = t.as_subclass(torch.Tensor)
    if base.dim() == 4 and base.stride(1) == 1 and not base.is_contiguous():      # (B, C, H, W) view of (B, H, W, C)
        return x3a_decode(base.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    return x3a_decode(base.contiguous())


def x3a_encode(x):
    """f32 tensor (contiguous, numel % 8 == 0, channel groups of 8 contiguous) -> x3a tensor of the same shape / dtype tag
    (float32 storage holding [8 f16 hi | 8 f16 lo] of 16 x per group)."""
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or x.numel() % 8:
        raise CggError('x3a_encode: contiguous float32 ROCm tensor with numel % 8 == 0 expected')
    y = torch.empty_like(x)
    check(_lib_().cgg_x3a_encode(dev_ptr(x), dev_ptr(y), x.numel(), stream_ptr(x.device)), 'cgg_x3a_encode')
    return y


def x3a_decode(x):
    """x3a tensor -> the f32 values it stands for (22 significant bits)."""
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or x.numel() % 8:
        raise CggError('x3a_decode: contiguous float32-tagged ROCm tensor with numel % 8 == 0 expected')
    y = torch.empty_like(x)
    check(_lib_().cgg_x3a_decode(dev_ptr(x), dev_ptr(y), x.numel(), stream_ptr(x.device)), 'cgg_x3a_decode')
    return y


def x3_overflow_check(device, reset=True):
    """-> True if an x3a producer on `device` has stored a value outside f16's range (|a| >= 4094) since the last reset.
    Synchronises the current stream."""
    v = ctypes.c_int(0)
    with torch.cuda.device(device):
        check(_lib_().cgg_x3_overflow_check(int(bool(reset)), ctypes.byref(v), stream_ptr(device)), 'cgg_x3_overflow_check')
    return v.value != 0


def _x3s_fmt(split):
    return X3A_SPLIT if split else X3A_F32


def gemm_x3s(a, packed, N, bias=None, res=None, res_split=False, res_mod=0, relu=False, out=None, out_split=False, cfg=-1):
    """a (M, K) x3a rows (row stride free, multiples of 8) x x3 image -> act(a W^T + bias (+ res)) (M, N), f32 or x3a
    (`out_split`); res (M, N) | (res_mod, N) in f32 or x3a (`res_split`). csrc/x3s_gemm.hip."""
    if a.dim() not in (2, 3) or a.stride(-1) != 1 or a.dtype != torch.float32 or not a.is_cuda or not is_x3(packed):
        raise CggError('gemm_x3s: a must be a 2-D / 3-D float32-tagged ROCm tensor with a contiguous last dim, packed an x3 image')
    rpb, bstride = 0, 0
    if a.dim() == 3:                    # (B, R, K) view of a stack of images (row stride and image stride free): rows = B * R
        rpb, bstride = int(a.shape[1]), int(a.stride(0))
        M, K, lda = a.shape[0] * a.shape[1], a.shape[2], int(a.stride(1))
    else:
        (M, K), lda = a.shape, int(a.stride(0))
    _x3_check(packed, N, K, 'gemm_x3s')
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    if y.dim() != 2 or y.stride(1) != 1 or y.shape != (M, N) or y.dtype != torch.float32:
        raise CggError('gemm_x3s: bad `out` view')
    if res is not None and (res.dim() != 2 or res.stride(1) != 1 or res.shape != ((res_mod or M), N) or res.dtype != torch.float32):
        raise CggError('gemm_x3s: bad `res` view')
    with _timed('gemm_x3', flops=2.0 * M * N * K, bytes=4.0 * (M * K + M * N * (2 if res is not None and not res_mod else 1)) +
                4.0 * N * K, shape=(M, N, K)):
        rc = _lib_().cgg_gemm_x3s_batched(ctypes.c_void_p(a.data_ptr()), lda, rpb, bstride, dev_ptr(packed),
                                          dev_ptr(bias, 'bias', torch.float32),
                                          ctypes.c_void_p(res.data_ptr()) if res is not None else None,
                                          res.stride(0) if res is not None else 0,
                                          0 if res is None else _x3s_fmt(res_split), int(res_mod), ctypes.c_void_p(y.data_ptr()),
                                          y.stride(0), _x3s_fmt(out_split), M, N, K, int(bool(relu)), stream_ptr(a.device)) \
            if rpb else \
            _lib_().cgg_gemm_x3s_cfg(ctypes.c_void_p(a.data_ptr()), lda, dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
                                     ctypes.c_void_p(res.data_ptr()) if res is not None else None,
                                     res.stride(0) if res is not None else 0,
                                     0 if res is None else _x3s_fmt(res_split), int(res_mod), ctypes.c_void_p(y.data_ptr()),
                                     y.stride(0), _x3s_fmt(out_split), M, N, K, int(bool(relu)), int(cfg), stream_ptr(a.device))
    check(rc, 'cgg_gemm_x3s')
    return y


def conv_x3s_nhwc(x, packed, N, kernel, stride=1, pad=0, bias=None, res=None, res_split=True, relu=False, out_split=True, cfg=-1):
    """x (B, H, W, C) x3a channel-last -> act(conv + bias (+ res)) (B, OH, OW, N), x3a (default) or f32: implicit GEMM on the
    x3 image made by `pack_conv_weight_x3` (C % 32 == 0)."""
    if x.dim() != 4 or not x.is_contiguous() or x.dtype != torch.float32 or not x.is_cuda or not is_x3(packed):
        raise CggError('conv_x3s_nhwc: x must be a contiguous (B, H, W, C) float32-tagged ROCm tensor, packed an x3 image')
    B, H, W, C = x.shape
    KH, KW = (kernel, kernel) if isinstance(kernel, int) else kernel
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    _x3_check(packed, N, C * KH * KW, 'conv_x3s_nhwc')
    y = torch.empty((B, OH, OW, N), dtype=torch.float32, device=x.device)
    if res is not None and (tuple(res.shape) != (B, OH, OW, N) or not res.is_contiguous() or res.dtype != torch.float32):
        raise CggError('conv_x3s_nhwc: res must be a contiguous (B, OH, OW, N) float32-tagged tensor')
    with _timed('gemm_x3', flops=2.0 * B * OH * OW * N * C * KH * KW,
                bytes=4.0 * (B * H * W * C + B * OH * OW * N * (2 if res is not None else 1) + N * C * KH * KW),
                shape=(B * OH * OW, N, C * KH * KW)):
        rc = _lib_().cgg_conv_x3s_nhwc_cfg(dev_ptr(x), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32), dev_ptr(res),
                                           0 if res is None else _x3s_fmt(res_split), dev_ptr(y), _x3s_fmt(out_split), B, H, W, C, N,
                                           KH, KW, int(stride), int(pad), int(bool(relu)), int(cfg), stream_ptr(x.device))
    check(rc, 'cgg_conv_x3s_nhwc')
    return y


def pack_linear_weight(weight):
    """weight (N, K) f32 -> opaque packed bf16 buffer for `linear_rows_bf16` (uint8 tensor)."""
    N, K = weight.shape
    nbytes = _lib_().cgg_linear_rows_packed_bytes(N, K)
    if nbytes <= 0:
        raise CggError(f'pack_linear_weight: unsupported shape {tuple(weight.shape)} (K % 16)')
    out = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
    w = weight.detach().float().contiguous()
    rc = _lib_().cgg_linear_rows_pack(dev_ptr(w, 'weight', torch.float32), dev_ptr(out), N, K,
                                      stream_ptr(weight.device))
    check(rc, 'cgg_linear_rows_pack')
    return out


def linear_rows_bf16(x, packed, N, bias=None, res=None, relu_cols=0, ln=None, pos=None, want_pos=False, ksplit=1,
                     out=None):
    """x (M, K) f32 rows (row stride free, last dim contiguous) @ packed weight -> y (M, N) f32.
    ln = (gamma, beta, eps) fuses a LayerNorm over the N <= 256 outputs; pos (rows, N) with want_pos returns
    (y, y + pos[row % rows]). `out` may be a 2-D view with its own row stride. ksplit > 1 returns the (ksplit, M, N)
    split-K partial planes (bias / res in plane 0) for `layernorm_chain` to add."""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32 or not x.is_cuda:
        raise CggError('linear_rows_bf16: x must be a 2-D float32 ROCm tensor with a contiguous last dim')
    M, K = x.shape
    fn, _ = _x3_fn('linear_rows', packed)
    if ksplit > 1:
        if out is not None or want_pos or ln is not None or relu_cols:
            raise CggError('linear_rows_bf16: split-K returns (ksplit, M, N) partial planes; no out / ln / pos / relu')
        y = torch.empty((int(ksplit), M, N), dtype=torch.float32, device=x.device)
        rc = fn(
            ctypes.c_void_p(x.data_ptr()), x.stride(0), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
            ctypes.c_void_p(res.data_ptr()) if res is not None else None, res.stride(0) if res is not None else 0,
            dev_ptr(y), N, None, None, 0.0, None, 0, None, 0, M, N, K, 0, int(ksplit), None, 0, 0, None, 0, 0,
            stream_ptr(x.device))
        check(rc, 'cgg_linear_rows_bf16(split-K)')
        return y
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=x.device)
    if y.dim() != 2 or y.stride(1) != 1 or y.shape != (M, N) or y.dtype != torch.float32:
        raise CggError('linear_rows_bf16: bad `out` view')
    if res is not None and (res.dim() != 2 or res.stride(1) != 1 or res.shape != (M, N)):
        raise CggError('linear_rows_bf16: bad `res` view')
    yp = torch.empty((M, N), dtype=torch.float32, device=x.device) if want_pos else None
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    rc = fn(
        ctypes.c_void_p(x.data_ptr()), x.stride(0), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32),
        ctypes.c_void_p(res.data_ptr()) if res is not None else None, res.stride(0) if res is not None else 0,
        ctypes.c_void_p(y.data_ptr()), y.stride(0), dev_ptr(g, 'gamma', torch.float32),
        dev_ptr(b, 'beta', torch.float32), float(eps), dev_ptr(pos, 'pos', torch.float32) if want_pos else None,
        pos.shape[0] if want_pos else 0, dev_ptr(yp), N if want_pos else 0, M, N, K, int(relu_cols), int(ksplit),
        None, 0, 0, None, 0, 0, stream_ptr(x.device))
    check(rc, 'cgg_linear_rows_bf16')
    return (y, yp) if want_pos else y


def linear_rows_bf16_qkv(xqk, xv, packed, bias, E):
    """Self-attention projections as ONE launch over the concatenated packed [Wq; Wk; Wv] (3E x E, E % 256 == 0):
    q, k from xqk (= query + query_pos), v from xv (= query); all (M, E) f32 rows -> (q (M, E), kv (M, 2E))."""
    M, K = xqk.shape
    for t in (xqk, xv):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32 or not t.is_cuda or t.shape != (M, K):
            raise CggError('linear_rows_bf16_qkv: inputs must be matching 2-D float32 ROCm tensors')
    if E % 256:
        raise CggError('linear_rows_bf16_qkv: E must be a multiple of 256')
    q = torch.empty((M, E), dtype=torch.float32, device=xqk.device)
    kv = torch.empty((M, 2 * E), dtype=torch.float32, device=xqk.device)
    rc = _x3_fn('linear_rows', packed)[0](
        ctypes.c_void_p(xqk.data_ptr()), xqk.stride(0), dev_ptr(packed), dev_ptr(bias, 'bias', torch.float32), None, 0,
        dev_ptr(q), E, None, None, 0.0, None, 0, None, 0, M, 3 * E, K, 0, 1,
        ctypes.c_void_p(xv.data_ptr()), xv.stride(0), 2 * E, dev_ptr(kv), 2 * E, E, stream_ptr(xqk.device))
    check(rc, 'cgg_linear_rows_bf16(qkv)')
    return q, kv


def layernorm_chain(a, norm_a, pos=None, norm_b=None):
    """a (M, N) f32 -> (y = LN_a(a), yp = y + pos[row % len(pos)] | None, z = LN_b(y) | None); norm_* are
    (gamma, beta, eps) triples."""
    nsum, plane = 1, 0
    if a.dim() == 3:                       # split-K partial planes (nsum, M, N): summed in the kernel, fixed order
        if not a.is_contiguous():
            raise CggError('layernorm_chain: partial planes must be contiguous')
        nsum, plane = a.shape[0], a.shape[1] * a.shape[2]
        a = a[0]
    M, N = a.shape
    if a.stride(1) != 1:
        raise CggError('layernorm_chain: last dim must be contiguous')
    y = torch.empty((M, N), dtype=torch.float32, device=a.device)
    yp = torch.empty_like(y) if pos is not None else None
    z = torch.empty_like(y) if norm_b is not None else None
    gb, bb, eb = norm_b if norm_b is not None else (None, None, 0.0)
    rc = _lib_().cgg_layernorm_chain(
        ctypes.c_void_p(a.data_ptr()), a.stride(0), dev_ptr(norm_a[0], 'gamma', torch.float32),
        dev_ptr(norm_a[1], 'beta', torch.float32), float(norm_a[2]), dev_ptr(pos, 'pos', torch.float32),
        pos.shape[0] if pos is not None else 0, dev_ptr(gb, 'gamma_b', torch.float32),
        dev_ptr(bb, 'beta_b', torch.float32), float(eb), dev_ptr(y), dev_ptr(yp), dev_ptr(z), M, N, int(nsum),
        int(plane), stream_ptr(a.device))
    check(rc, 'cgg_layernorm_chain')
    return y, yp, z


def decoder_mid(core, wo, bo, res, norm, pos=None, qkv=None):
    """x1 = LN(core @ Wo^T + bo + res) on (M, 256) f32 rows; with qkv = (packed [Wq; Wk; Wv], bias (768,)) and pos (Q, 256)
    also q = (x1 + pos) Wq^T + bq (M, 256) and kv = [(x1 + pos) Wk^T + bk | x1 Wv^T + bv] (M, 512). -> (x1, q, kv)."""
    M, C = core.shape
    for t in (core, res):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32 or t.shape != (M, C):
            raise CggError('decoder_mid: core / res must be matching (M, C) float32 rows')
    dev = core.device
    x1 = torch.empty((M, C), dtype=torch.float32, device=dev)
    q = kv = None
    wqkv = bqkv = None
    if qkv is not None:
        wqkv, bqkv = qkv
        q = torch.empty((M, C), dtype=torch.float32, device=dev)
        kv = torch.empty((M, 2 * C), dtype=torch.float32, device=dev)
    rc = _x3_fn('decoder_mid', wo, wqkv)[0](
        ctypes.c_void_p(core.data_ptr()), core.stride(0), dev_ptr(wo), dev_ptr(bo, 'bo', torch.float32),
        ctypes.c_void_p(res.data_ptr()), res.stride(0), dev_ptr(norm[0], 'gamma', torch.float32),
        dev_ptr(norm[1], 'beta', torch.float32), float(norm[2]), dev_ptr(pos, 'pos', torch.float32),
        pos.shape[0] if pos is not None else 0, dev_ptr(wqkv), dev_ptr(bqkv, 'bqkv', torch.float32), dev_ptr(x1),
        dev_ptr(q), dev_ptr(kv), M, C, stream_ptr(dev))
    check(rc, 'cgg_decoder_mid_bf16')
    return x1, q, kv


def decoder_ffn(x, w1, b1, w2, b2, F):
    """x (M, 256) f32 rows -> (F / 256, M, 256) partial planes of x + relu(x W1^T + b1) W2^T + b2 (sum them in plane
    order: `decoder_tail` / `layernorm_chain`). w1 (F x 256), w2 (256 x F) packed by `pack_linear_weight`."""
    M, C = x.shape
    if x.stride(1) != 1 or x.dtype != torch.float32 or F % 256:
        raise CggError('decoder_ffn: x must be (M, C) float32 rows and F a multiple of 256')
    planes = torch.empty((F // 256, M, C), dtype=torch.float32, device=x.device)
    rc = _x3_fn('decoder_ffn', w1, w2)[0](ctypes.c_void_p(x.data_ptr()), x.stride(0), dev_ptr(w1),
                                      dev_ptr(b1, 'b1', torch.float32), dev_ptr(w2), dev_ptr(b2, 'b2', torch.float32),
                                      dev_ptr(planes), M, C, int(F), stream_ptr(x.device))
    check(rc, 'cgg_decoder_ffn_bf16')
    return planes


def self_attn_rows_bf16(q, kv, B, num_heads, scale=None):
    """q (M, E), kv (M, 2E) f32 rows (M = B*Q, Q <= 128, head dim 32) -> softmax(scale q k^T) v, (M, E) f32."""
    M, E = q.shape
    D = E // num_heads
    if q.stride(1) != 1 or kv.stride(1) != 1 or kv.shape != (M, 2 * E) or q.dtype != torch.float32 or kv.dtype != torch.float32:
        raise CggError('self_attn_rows_bf16: q (M, E) / kv (M, 2E) float32 rows expected')
    out = torch.empty((M, E), dtype=torch.float32, device=q.device)
    rc = _lib_().cgg_self_attn_rows_bf16(ctypes.c_void_p(q.data_ptr()), q.stride(0), ctypes.c_void_p(kv.data_ptr()),
                                         kv.stride(0), dev_ptr(out), B, M // B, num_heads, D,
                                         float(scale if scale is not None else D ** -0.5), stream_ptr(q.device))
    check(rc, 'cgg_self_attn_rows_bf16')
    return out


def decoder_tail(planes, norm_a, pos, norm_b, mlp, qproj=None, want_pos=False):
    """planes (nsum, M, 256) f32 (FFN split-K partials, bias + residual in plane 0) -> (y = LN_a(sum), y + pos | None,
    mask_embed = MLP3(LN_b(y)), qn = Wq (y + pos) + bq | None). norm_* = (gamma, beta, eps); mlp = (w1, b1, w2, b2, w3, b3)
    and qproj = (wq, bq) with weights packed by `pack_linear_weight` (256 x 256)."""
    if planes.dim() != 3 or not planes.is_contiguous() or planes.dtype != torch.float32:
        raise CggError('decoder_tail: planes must be a contiguous (nsum, M, C) float32 tensor')
    nsum, M, C = planes.shape
    dev = planes.device
    y = torch.empty((M, C), dtype=torch.float32, device=dev)
    yp = torch.empty_like(y) if want_pos else None
    me = torch.empty_like(y)
    qn = torch.empty_like(y) if qproj is not None else None
    wq, bq = qproj if qproj is not None else (None, None)
    rc = _x3_fn('decoder_tail', mlp[0], mlp[2], mlp[4], wq)[0](
        dev_ptr(planes), nsum, M * C, C, dev_ptr(norm_a[0], 'gamma_a', torch.float32),
        dev_ptr(norm_a[1], 'beta_a', torch.float32), float(norm_a[2]), dev_ptr(pos, 'pos', torch.float32), pos.shape[0],
        dev_ptr(norm_b[0], 'gamma_b', torch.float32), dev_ptr(norm_b[1], 'beta_b', torch.float32), float(norm_b[2]),
        dev_ptr(mlp[0]), dev_ptr(mlp[1], 'b1', torch.float32), dev_ptr(mlp[2]), dev_ptr(mlp[3], 'b2', torch.float32),
        dev_ptr(mlp[4]), dev_ptr(mlp[5], 'b3', torch.float32), dev_ptr(wq), dev_ptr(bq, 'bq', torch.float32),
        dev_ptr(y), dev_ptr(yp), dev_ptr(me), dev_ptr(qn), M, C, stream_ptr(dev))
    check(rc, 'cgg_decoder_tail_bf16')
    return y, yp, me, qn


def masked_xattn_bf16(q, k, vt, bits, num_heads, scale=None, fix_full_rows=False):
    """q (B,Q,E) f32; k (B,S,E) bf16; vt (B,E,S) bf16 (value projection, transposed); bits as `masked_xattn`.
    fix_full_rows: rows whose mask blocks every key attend to all keys (what `attn_mask_fix_full_rows` does to `bits`
    beforehand), decided inside the kernels; `bits` itself is not modified."""
    B, Q, E = q.shape
    if Q > 128:
        return torch.cat([masked_xattn_bf16(q[:, s:s + 128].contiguous(), k, vt,
                                            None if bits is None else bits[:, s:s + 128].contiguous(), num_heads,
                                            scale, fix_full_rows) for s in range(0, Q, 128)], 1)
    S = k.shape[1]
    H = int(num_heads)
    D = E // H
    if tuple(k.shape) != (B, S, E) or tuple(vt.shape) != (B, E, S):
        raise CggError(f'masked_xattn_bf16: k {tuple(k.shape)} / vt {tuple(vt.shape)} do not match q {tuple(q.shape)}')
    # k may be a column slice of a wider (B, S, n*E) projection, vt a row block of a taller (B, n*E, S) one
    if k.stride(2) != 1 or k.stride(0) != S * k.stride(1) or vt.stride(2) != 1 or vt.stride(1) != S or \
            k.dtype != torch.bfloat16 or vt.dtype != torch.bfloat16:
        raise CggError('masked_xattn_bf16: unsupported k / vt strides or dtype')
    ldk, vtb = k.stride(1), vt.stride(0)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = _lib_()
    nbytes = lib.cgg_masked_xattn_workspace_bytes(B, Q, H, D, S)
    key = (q.device.index, stream_ptr(q.device).value)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=q.device)
        _WS_CACHE[key] = ws
    out = torch.empty((B, Q, E), dtype=torch.float32, device=q.device)
    rc = lib.cgg_masked_xattn_forward_bf16(dev_ptr(q, 'q', torch.float32), ctypes.c_void_p(k.data_ptr()),
                                           ctypes.c_void_p(vt.data_ptr()), dev_ptr(bits, 'bits', torch.int32),
                                           dev_ptr(out), dev_ptr(ws), B, Q, H, D, S, float(scale), int(ldk), int(vtb),
                                           int(bool(fix_full_rows)), stream_ptr(q.device))
    check(rc, 'cgg_masked_xattn_forward_bf16')
    return out


def instance_masks_multi(logits, index_lists, up_size, crop_size, out_size):
    """logits (Q,H,W) f32 low-res; index_lists = [query indices (n_t,) long, ...] (one per evaluation type)
    -> ([masks (n_t,oh,ow) bool ...], mask_score (Q,), bbox (Q,4)). Every picked query's mask is interpolated ONCE and
    written to all its detections; score / bbox are per QUERY (index them with the query indices)."""
    Q, H, W = logits.shape
    oh, ow = int(out_size[0]), int(out_size[1])
    dev = logits.device
    sizes = [int(ix.numel()) for ix in index_lists]
    total = sum(sizes)
    score = torch.empty((Q,), dtype=torch.float32, device=dev)
    bbox = torch.empty((Q, 4), dtype=torch.float32, device=dev)
    masks = torch.empty((total, oh, ow), dtype=torch.uint8, device=dev)
    if total == 0:
        return [masks[:0].view(torch.bool) for _ in sizes], score, bbox
    cat = torch.cat([ix.reshape(-1) for ix in index_lists]) if len(index_lists) > 1 else index_lists[0].reshape(-1)
    order = torch.argsort(cat, stable=True).to(torch.int32)
    # (torch.bincount synchronises to size its output -> not capturable in a hipGraph; scatter_add is)
    counts = torch.zeros((Q,), dtype=torch.int32, device=dev).scatter_add_(
        0, cat, torch.ones_like(cat, dtype=torch.int32))
    off = torch.zeros((Q + 1,), dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(counts, 0)
    ws = torch.empty((Q, 8), dtype=torch.int32, device=dev)
    rc = _lib_().cgg_instance_masks_multi(dev_ptr(logits, 'logits', torch.float32), dev_ptr(off), dev_ptr(order),
                                          dev_ptr(masks), dev_ptr(score), dev_ptr(bbox), dev_ptr(ws), Q, H, W,
                                          int(up_size[0]), int(up_size[1]), int(crop_size[0]), int(crop_size[1]), oh,
                                          ow, stream_ptr(dev))
    check(rc, 'cgg_instance_masks_multi')
    out, o = [], 0
    for n in sizes:
        out.append(masks[o:o + n].view(torch.bool))
        o += n
    return out, score, bbox


def class_topk(dots, B, col0, ncols, k):
    """dots (B*Q, ld) f32 class-embedding dot products against concatenated class tables; evaluation type t owns columns
    col0[t] .. col0[t]+ncols[t] (last = background). -> (labels, scores, query indices), each (B, T, k): the k best
    (query, class) pairs of softmax(dots_t)[:, :-1], descending score (ties: ascending flat index)."""
    rows, ld = dots.shape
    Q, T = rows // B, len(col0)
    dev = dots.device
    labels = torch.empty((B, T, k), dtype=torch.int64, device=dev)
    qidx = torch.empty((B, T, k), dtype=torch.int64, device=dev)
    scores = torch.empty((B, T, k), dtype=torch.float32, device=dev)
    rc = _lib_().cgg_class_topk(dev_ptr(dots, 'dots', torch.float32), ld, B, Q, T, _int_array(col0), _int_array(ncols),
                                int(k), dev_ptr(labels), dev_ptr(scores), dev_ptr(qidx), stream_ptr(dev))
    check(rc, 'cgg_class_topk')
    return labels, scores, qidx


def class_topk_supported(Q, ncols, k):
    return (0 < k <= 1024 and 1 <= len(ncols) <= 8 and all(n >= 2 and k <= Q * (n - 1) for n in ncols)
            and 8 * k + 4 * Q * (max(ncols) - 1) + 4 * Q <= 62 * 1024)


def instance_masks_picks(logits, qidx, cls_scores, up_size, crop_size, out_size, bitpack=False):
    """logits (Q,H,W) f32 low-res; qidx (n,) long / cls_scores (n,) f32: the picks of ALL evaluation types of one image
    -> masks (n,oh,ow) bool, bboxes (n,5) f32 = (box of the picked query, class score x mask score)."""
    Q, H, W = logits.shape
    oh, ow = int(out_size[0]), int(out_size[1])
    dev = logits.device
    n = int(qidx.numel())
    if bitpack and ow % 16:
        raise CggError('instance_masks_picks: bit-packed masks need out_w % 16 == 0')
    masks = torch.empty((n, oh, ow // 8 if bitpack else ow), dtype=torch.uint8, device=dev)
    bboxes = torch.empty((n, 5), dtype=torch.float32, device=dev)
    ws = torch.empty((int(_lib_().cgg_instance_masks_picks_workspace_bytes(Q, n)) + 3) // 4, dtype=torch.int32, device=dev)
    rc = _lib_().cgg_instance_masks_picks(dev_ptr(logits, 'logits', torch.float32), dev_ptr(qidx, 'qidx', torch.int64),
                                          dev_ptr(cls_scores, 'cls_scores', torch.float32), n, dev_ptr(masks),
                                          dev_ptr(bboxes), dev_ptr(ws), Q, H, W, int(up_size[0]), int(up_size[1]),
                                          int(crop_size[0]), int(crop_size[1]), oh, ow, int(bool(bitpack)), stream_ptr(dev))
    check(rc, 'cgg_instance_masks_picks')
    return (masks if bitpack else masks.view(torch.bool)), bboxes


def instance_masks_bitpack_ok(logits_hw, up_size, crop_size, out_size):
    """Shapes the bit-packed mask output supports: integer up-scale 2 / 4 / 8, no second resize, width % 16 == 0."""
    h, w = logits_hw
    s = up_size[0] // h if h else 0
    return (s in (2, 4, 8) and s * h == up_size[0] and s * w == up_size[1] and tuple(crop_size) == tuple(out_size)
            and out_size[1] % 16 == 0)


def bias_relu_maxpool_nhwc(x, bias):
    """x (B, H, W, C) channel-last bf16 raw convolution output -> relu(maxpool3x3/s2/p1(x) + bias) (B, Ho, Wo, C)."""
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    rc = _lib_().cgg_bias_relu_maxpool_nhwc(dev_ptr(x, 'x', torch.bfloat16), dev_ptr(bias, 'bias', torch.bfloat16),
                                            dev_ptr(y), B, H, W, C, stream_ptr(x.device))
    check(rc, 'cgg_bias_relu_maxpool_nhwc')
    return y


_BLASLT_READY = False


def _blaslt_init():
    """dlopen the hipBLASLt copy torch itself uses (one library instance per process)."""
    global _BLASLT_READY
    if not _BLASLT_READY:
        import os
        cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libhipblaslt.so')
        path = cand if os.path.exists(cand) else 'libhipblaslt.so'
        check(_lib_().cgg_blaslt_init(path.encode()), 'cgg_blaslt_init')
        check(_lib_().cgg_blaslt_set_tuning(int(os.environ.get('CGG_GEMM_TUNE', '16'))), 'cgg_blaslt_set_tuning')
        _BLASLT_READY = True


def blaslt_last_tuning():
    a, b = ctypes.c_float(0), ctypes.c_float(0)
    _lib_().cgg_blaslt_last_tuning(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def gemm_bias_res_act_bf16(x, w, bias, res=None, relu=True):
    """y = act(x @ w^T + bias + res): x (M, K), w (N, K), bias (N,), res (M, N) bf16 contiguous -> (M, N) bf16, one
    hipBLASLt call (residual through beta = 1, bias + ReLU in the epilogue)."""
    _blaslt_init()
    M, K = x.shape
    N = w.shape[0]
    for t, n in ((x, 'x'), (w, 'w'), (res, 'res'), (bias, 'bias')):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous()):
            raise CggError('gemm_bias_res_act_bf16: %s must be a contiguous bfloat16 tensor' % n)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    rc = _lib_().cgg_gemm_bias_res_act_bf16(dev_ptr(x, 'x', torch.bfloat16), dev_ptr(w, 'w', torch.bfloat16),
                                            dev_ptr(bias), dev_ptr(res), dev_ptr(y), M, N, K,
                                            int(bool(relu)), stream_ptr(x.device))
    check(rc, 'cgg_gemm_bias_res_act_bf16')
    return y


def im2col3x3_nhwc(x, stride=1):
    """x (B, H, W, C) channel-last bf16 -> patch matrix (B * Ho * Wo, 9 * C) of a 3x3 / padding-1 convolution, columns
    ordered (ky, kx, c); returns (matrix, Ho, Wo)."""
    B, H, W, C = x.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        raise CggError('im2col3x3_nhwc: x must be a contiguous (B, H, W, C) bfloat16 tensor')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B * Ho * Wo, 9 * C), dtype=torch.bfloat16, device=x.device)
    check(_lib_().cgg_im2col3x3_nhwc(dev_ptr(x), dev_ptr(y), B, H, W, C, int(stride), stream_ptr(x.device)),
          'cgg_im2col3x3_nhwc')
    return y, Ho, Wo


def pack_stem_weight(w):
    """w (64, 3, 7, 7) (BN already folded) -> bf16 MFMA A fragments for `stem_conv7x7` (uint8 tensor)."""
    if tuple(w.shape) != (64, 3, 7, 7):
        raise CggError(f'pack_stem_weight: expected (64, 3, 7, 7), got {tuple(w.shape)}')
    wk = torch.zeros((64, 7, 8, 3), dtype=torch.float32, device=w.device)      # (cout, ky, kx padded to 8, c)
    wk[:, :, :7, :] = w.detach().float().permute(0, 2, 3, 1)
    wk = torch.cat([wk.reshape(64, 168), torch.zeros((64, 8), dtype=torch.float32, device=w.device)], 1)   # K = 176
    frag = wk.view(2, 32, 11, 2, 8).permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16)    # (mt, ks, hi, j, e)
    out = frag.view(torch.uint8).reshape(-1)
    assert out.numel() == _lib_().cgg_stem_conv7x7_packed_bytes()
    return out


def stem_conv7x7(img, packed):
    """img (B, 3, H, W) f32 NCHW -> raw 7x7 / stride-2 / padding-3 convolution (B, Ho, Wo, 64) bf16 channel-last."""
    B, C, H, W = img.shape
    if C != 3 or img.dtype != torch.float32 or not img.is_contiguous():
        raise CggError('stem_conv7x7: img must be a contiguous (B, 3, H, W) float32 tensor')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, 64), dtype=torch.bfloat16, device=img.device)
    check(_lib_().cgg_stem_conv7x7_nchw(dev_ptr(img), dev_ptr(packed), dev_ptr(y), B, H, W, stream_ptr(img.device)),
          'cgg_stem_conv7x7_nchw')
    return y


def pack_stem_weight_x3(w):
    """w (64, 3, 7, 7) f32 (BN folded) -> (hi | lo f16 MFMA A fragments (uint8), un-scaling factors (64,) f32) for
    `stem_conv7x7_x3`: rows scaled by a power of two so that max |w'| is in [2^10, 2^11) (csrc/x3.h)."""
    if tuple(w.shape) != (64, 3, 7, 7):
        raise CggError(f'pack_stem_weight_x3: expected (64, 3, 7, 7), got {tuple(w.shape)}')
    wk = torch.zeros((64, 7, 8, 3), dtype=torch.float32, device=w.device)      # (cout, ky, kx padded to 8, c)
    wk[:, :, :7, :] = w.detach().float().permute(0, 2, 3, 1)
    wk = torch.cat([wk.reshape(64, 168), torch.zeros((64, 8), dtype=torch.float32, device=w.device)], 1)   # K = 176
    amax = wk.abs().amax(1).clamp_min(1e-30)
    e = torch.floor(torch.log2(amax)) + 1                                      # amax = f 2^e, f in [0.5, 1)
    s = torch.pow(2.0, 11 - e)
    ws = wk * s[:, None]
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    frag = lambda t: t.view(2, 32, 11, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(torch.uint8).reshape(-1)   # (mt, ks, hi, j, e)
    packed = torch.cat([frag(hi), frag(lo)])
    assert packed.numel() == 2 * _lib_().cgg_stem_conv7x7_packed_bytes()
    return packed, (1.0 / (s * 16.0)).float().contiguous()


def stem_conv7x7_x3(img, packed, wscale):
    """img (B, 3, H, W) f32 NCHW -> raw 7x7 / stride-2 / padding-3 convolution (B, Ho, Wo, 64) F32 channel-last, f32-class."""
    B, C, H, W = img.shape
    if C != 3 or img.dtype != torch.float32 or not img.is_contiguous():
        raise CggError('stem_conv7x7_x3: img must be a contiguous (B, 3, H, W) float32 tensor')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, 64), dtype=torch.float32, device=img.device)
    check(_lib_().cgg_stem_conv7x7_x3_nchw(dev_ptr(img), dev_ptr(packed), dev_ptr(wscale, 'wscale', torch.float32), dev_ptr(y), B, H, W,
                                           stream_ptr(img.device)), 'cgg_stem_conv7x7_x3_nchw')
    return y


def bias_relu_maxpool_nhwc_f32(x, bias, x3a=False):
    """x (B, H, W, C) channel-last F32 raw convolution output -> relu(maxpool3x3/s2/p1(x) + bias) (B, Ho, Wo, C) f32, or the
    same map as x3a rows (`x3a=True`: the layout the x3s convolutions consume)."""
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    fn = _lib_().cgg_bias_relu_maxpool_nhwc_f32_x3a if x3a else _lib_().cgg_bias_relu_maxpool_nhwc_f32
    rc = fn(dev_ptr(x, 'x', torch.float32), dev_ptr(bias, 'bias', torch.float32), dev_ptr(y), B, H, W, C, stream_ptr(x.device))
    check(rc, 'cgg_bias_relu_maxpool_nhwc_f32')
    return y


def linear_sum_assignment_batch(costs):
    """costs: list of 2-D float32 CPU tensors -> list of (rows, cols) int64 CPU tensors; every problem of a step in ONE
    call into the C++ solver (same indices as scipy.optimize.linear_sum_assignment, incl. ties)."""
    if not costs:
        return []
    flat = torch.cat([c.detach().to(device='cpu', dtype=torch.float32).contiguous().reshape(-1) for c in costs]) \
        if len(costs) > 1 else costs[0].detach().to(device='cpu', dtype=torch.float32).contiguous().reshape(-1)
    nr = [int(c.shape[0]) for c in costs]
    nc = [int(c.shape[1]) for c in costs]
    sizes = [min(a, b) for a, b in zip(nr, nc)]
    total = sum(sizes)
    rows = torch.empty((max(total, 1),), dtype=torch.int64)
    cols = torch.empty((max(total, 1),), dtype=torch.int64)
    rc = _lib_().cgg_linear_sum_assignment_f32(ctypes.c_void_p(flat.data_ptr()) if flat.numel() else None, len(costs),
                                               _int_array(nr), _int_array(nc), ctypes.c_void_p(rows.data_ptr()),
                                               ctypes.c_void_p(cols.data_ptr()))
    check(rc, 'cgg_linear_sum_assignment_f32')
    out, o = [], 0
    for n in sizes:
        out.append((rows[o:o + n], cols[o:o + n]))
        o += n
    return out


def point_sample_planes(planes, index, points):
    """planes (N, H, W) f32, index (rows,) int32, points (rows, P, 2) f32 in [0, 1] (x, y) -> (rows, P): row j = bilinear
    sample (grid_sample arithmetic, zeros outside) of planes[index[j]] at points[j]."""
    N, H, W = planes.shape
    rows, P, _ = points.shape
    out = torch.empty((rows, P), dtype=torch.float32, device=planes.device)
    if rows == 0:
        return out
    check(_lib_().cgg_point_sample_planes(dev_ptr(planes, 'planes', torch.float32), dev_ptr(index, 'index', torch.int32),
                                          dev_ptr(points.contiguous(), 'points', torch.float32), dev_ptr(out), N, H, W, rows, P,
                                          stream_ptr(planes.device)), 'cgg_point_sample_planes')
    return out


class _PointSampleRowsFn(torch.autograd.Function):
    """out (rows, P) = bilinear samples of planes[j] (rows, H, W) f32 at points[j] (rows, P, 2) (grid_sample arithmetic, zeros
    outside, align_corners=False); differentiable wrt the planes only (the points are constants of the loss)."""

    @staticmethod
    def forward(ctx, planes, points):
        planes = planes.contiguous()
        points = points.contiguous()
        rows, H, W = planes.shape
        idx = torch.arange(rows, dtype=torch.int32, device=planes.device)
        ctx.save_for_backward(points)
        ctx.shape = (rows, H, W)
        return point_sample_planes(planes, idx, points)

    @staticmethod
    def backward(ctx, gout):
        (points,) = ctx.saved_tensors
        rows, H, W = ctx.shape
        if rows and rows <= 65535 and W <= 16384:
            # every row has its own plane: band-stationary LDS accumulation, planes written once (no zero-fill, no global atomics)
            gp = torch.empty((rows, H, W), dtype=torch.float32, device=gout.device)
            check(_lib_().cgg_point_sample_planes_backward_rows(dev_ptr(gout.contiguous().float(), 'grad', torch.float32),
                                                                dev_ptr(points, 'points', torch.float32), dev_ptr(gp), H, W, rows,
                                                                points.shape[1], stream_ptr(gout.device)),
                  'cgg_point_sample_planes_backward_rows')
            return gp, None
        gp = torch.zeros((rows, H, W), dtype=torch.float32, device=gout.device)
        if rows:
            check(_lib_().cgg_point_sample_planes_backward(dev_ptr(gout.contiguous().float(), 'grad', torch.float32), None,
                                                           dev_ptr(points, 'points', torch.float32), dev_ptr(gp), rows, H, W, rows,
                                                           points.shape[1], stream_ptr(gout.device)),
                  'cgg_point_sample_planes_backward')
        return gp, None


def point_sample_rows(planes, points):
    """planes (rows, H, W) f32 (may require grad), points (rows, P, 2) f32 in [0, 1] -> (rows, P); == [3P] mmcv point_sample of
    planes.unsqueeze(1) at the same points (F.grid_sample bilinear / zeros / align_corners=False)."""
    return _PointSampleRowsFn.apply(planes, points)


def point_sample_rows_ok(planes, points):
    return (planes.is_cuda and planes.dtype == torch.float32 and planes.dim() == 3 and points.dim() == 3
            and points.dtype == torch.float32 and points.shape[0] == planes.shape[0] and points.shape[2] == 2)


def point_sample_nhwc(feat, points):
    """feat (B, H, W, C) f32 channel-last, points (B, P, 2) in [0, 1] (x, y) -> (B, P, C): [3P] mmcv point_sample
    (grid_sample bilinear / zeros / align_corners=False) with the layout that makes a point's taps contiguous rows."""
    B, H, W, C = feat.shape
    P = points.shape[1]
    if feat.dtype != torch.float32 or not feat.is_contiguous() or points.dtype != torch.float32 or tuple(points.shape) != (B, P, 2):
        raise CggError('point_sample_nhwc: feat (B, H, W, C) contiguous float32, points (B, P, 2) float32 expected')
    pts = points.contiguous()
    out = torch.empty((B, P, C), dtype=torch.float32, device=feat.device)
    check(_lib_().cgg_point_sample_nhwc(dev_ptr(feat), dev_ptr(pts), dev_ptr(out), B, H, W, C, P, stream_ptr(feat.device)),
          'cgg_point_sample_nhwc')
    return out


def point_sample_nhwc_x3_ok(feat, points, groups):
    """shapes `point_sample_nhwc_x3` covers"""
    return (feat.dim() == 4 and feat.dtype == torch.float32 and feat.is_cuda and feat.is_contiguous() and feat.shape[-1] % 8 == 0
            and feat.shape[-1] <= 1024 and points.dim() == 3 and points.dtype == torch.float32 and groups > 0
            and points.shape[1] % groups == 0 and (points.shape[1] // groups) % 32 == 0)


def point_sample_nhwc_x3(feat, points, groups):
    """feat (B, H, W, C) f32 channel-last, points (B, groups * P, 2) in [0, 1] (x, y), group g's points at [g P, (g + 1) P) ->
    [PackedFeature] * groups: the samples of group g as the x3 images of a (B, C, 1, P) "map" (`cgg_point_sample_nhwc_x3`: sampler and
    pack kernel in one, no f32 sample tensor) -- `mask_logits(embed_g, packed[g])` is then mask_embed_g . sample(feat) at the group's
    points, the training loss' point logits of one decoder layer, on the f32-class MFMA einsum."""
    B, H, W, C = feat.shape
    Pt = points.shape[1]
    if not point_sample_nhwc_x3_ok(feat, points, groups) or tuple(points.shape) != (B, Pt, 2):
        raise CggError('point_sample_nhwc_x3: feat (B, H, W, C) contiguous float32 (C % 8 == 0), points (B, groups * P, 2) float32 '
                       'with P % 32 == 0 expected')
    P = Pt // groups
    pts = points.contiguous()
    hi = torch.empty((groups * B, P // 32, C // 8, 32, 8), dtype=torch.bfloat16, device=feat.device)     # (16-bit pieces: f16 bits)
    lo = torch.empty_like(hi)
    check(_lib_().cgg_point_sample_nhwc_x3(dev_ptr(feat), dev_ptr(pts), dev_ptr(hi), dev_ptr(lo), B, H, W, C, Pt, P,
                                           stream_ptr(feat.device)), 'cgg_point_sample_nhwc_x3')
    return [PackedFeature(hi[g * B:(g + 1) * B], lo[g * B:(g + 1) * B], B, C, 1, P) for g in range(groups)]


def subsample_nhwc(x, stride):
    """x (B, H, W, C) channel-last bf16 -> x[:, ::stride, ::stride, :] as a new contiguous tensor."""
    B, H, W, C = x.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or C % 8:
        return x[:, ::stride, ::stride, :].contiguous()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    check(_lib_().cgg_subsample_nhwc(dev_ptr(x), dev_ptr(y), B, H, W, C, int(stride), stream_ptr(x.device)), 'cgg_subsample_nhwc')
    return y


# ------------------------------------------------------------------------------------------------
# host side of the inference tail: COCO RLE of bit-packed masks (host function of the library, no device work)
# ------------------------------------------------------------------------------------------------
def rle_encode_bitmasks(bits, width, threads=8):
    """bits: HOST uint8 tensor / numpy array (n, H, row_bytes), pixel x in bit (x & 7) of byte (x >> 3) (what
    `instance_masks_picks(bitpack=True)` writes) -> list of n COCO RLE dicts {'size': [H, W], 'counts': bytes}."""
    import numpy as np
    arr = bits.numpy() if torch.is_tensor(bits) else np.asarray(bits)
    if arr.dtype != np.uint8 or arr.ndim != 3 or not arr.flags['C_CONTIGUOUS']:
        raise CggError('rle_encode_bitmasks: (n, H, row_bytes) contiguous uint8 host array expected')
    n, H, rb = arr.shape
    W = int(width)
    if n == 0:
        return []
    lib = _lib_()
    offs = np.empty(n + 1, dtype=np.int64)
    cap = max(1 << 16, n * 4096)          # ~2 KB per mask at 1024^2 with this benchmark's noisy masks: a too-small buffer encodes twice
    while True:
        out = np.empty(cap, dtype=np.uint8)
        total = lib.cgg_rle_encode_bitmasks(ctypes.c_void_p(arr.ctypes.data), n, H, W, H * rb, rb, int(threads),
                                            ctypes.c_void_p(out.ctypes.data), cap, ctypes.c_void_p(offs.ctypes.data))
        if total < 0:
            check(int(-total), 'cgg_rle_encode_bitmasks')
        if total <= cap:
            break
        cap = int(total)
    raw = out[:total].tobytes()
    o = offs.tolist()
    return [dict(size=[H, W], counts=raw[o[i]:o[i + 1]]) for i in range(n)]
