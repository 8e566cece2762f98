// K19: inference tail -- wavefront softmax/argmax and the fused "upsample -> threshold / argmax"
// kernels that keep the (Q, H_img, W_img) f32 logit tensor of
// open_set/models/mask2former_head.py:960-964 from ever being written.
// All HBM-bound streaming kernels; low-res logits (Q x H/4 x W/4) are re-read through L1/L2.
#include "cgg_common.h"

// torch upsample_bilinear2d (align_corners=False) source coordinate for output index `o`
struct BiTap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ BiTap cgg_bitap(int o, float scale, int in_size) {
  float f = scale * ((float)o + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  BiTap t;
  t.i0 = (int)f;
  if (t.i0 > in_size - 1) t.i0 = in_size - 1;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = f - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}
__device__ __forceinline__ float cgg_bilerp(const float* __restrict__ s, int W, const BiTap& ty,
                                            const BiTap& tx) {
  return ty.l0 * (tx.l0 * s[(size_t)ty.i0 * W + tx.i0] + tx.l1 * s[(size_t)ty.i0 * W + tx.i1]) +
         ty.l1 * (tx.l0 * s[(size_t)ty.i1 * W + tx.i0] + tx.l1 * s[(size_t)ty.i1 * W + tx.i1]);
}

// value of the (optionally two-stage) resized logit at final pixel (oy, ox):
//   stage 1: [H, W] -> [up_h, up_w] (batch_input_shape), crop to [crop_h, crop_w] (img_shape)
//   stage 2 (if out != crop): [crop_h, crop_w] -> [out_h, out_w] (ori_shape, `rescale`)
struct ResizeGeom {
  int H, W, up_h, up_w, crop_h, crop_w, out_h, out_w;
  float s1y, s1x, s2y, s2x;
  int two_stage;
};
__device__ __forceinline__ float cgg_resized_logit(const float* __restrict__ s, const ResizeGeom& g,
                                                   int oy, int ox) {
  if (!g.two_stage) {
    const BiTap ty = cgg_bitap(oy, g.s1y, g.H), tx = cgg_bitap(ox, g.s1x, g.W);
    return cgg_bilerp(s, g.W, ty, tx);
  }
  const BiTap uy = cgg_bitap(oy, g.s2y, g.crop_h), ux = cgg_bitap(ox, g.s2x, g.crop_w);
  float v[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const BiTap ty = cgg_bitap(a ? uy.i1 : uy.i0, g.s1y, g.H);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const BiTap tx = cgg_bitap(c ? ux.i1 : ux.i0, g.s1x, g.W);
      v[a][c] = cgg_bilerp(s, g.W, ty, tx);
    }
  }
  return uy.l0 * (ux.l0 * v[0][0] + ux.l1 * v[0][1]) + uy.l1 * (ux.l0 * v[1][0] + ux.l1 * v[1][1]);
}

static ResizeGeom make_geom(int H, int W, int up_h, int up_w, int crop_h, int crop_w, int out_h,
                            int out_w) {
  ResizeGeom g;
  g.H = H; g.W = W; g.up_h = up_h; g.up_w = up_w; g.crop_h = crop_h; g.crop_w = crop_w;
  g.out_h = out_h; g.out_w = out_w;
  g.s1y = (float)H / (float)up_h;
  g.s1x = (float)W / (float)up_w;
  g.two_stage = (out_h != crop_h || out_w != crop_w) ? 1 : 0;
  g.s2y = (float)crop_h / (float)out_h;
  g.s2x = (float)crop_w / (float)out_w;
  return g;
}

__device__ __forceinline__ float cgg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cgg_upsample_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int H, int W, int h,
                                                           int w, float sy, float sx) {
  const int n = blockIdx.y;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)h * w) return;
  const int oy = (int)(p / w), ox = (int)(p - (long long)oy * w);
  const BiTap ty = cgg_bitap(oy, sy, H), tx = cgg_bitap(ox, sx, W);
  y[(size_t)n * h * w + p] = cgg_bilerp(x + (size_t)n * H * W, W, ty, tx);
}

// instance masks: thread = 16 consecutive output pixels (one 16-B mask store) of detection blockIdx.y;
// block-level reduction, then ONE set of atomics per block (4096 pixels).
// ws per detection: [0] f32 sum sigmoid*[m>0], [1] i32 count, [2..5] i32 xmin, ymin, xmax, ymax
#define IM_PPT 16
// Multi-destination mode (dest_off != nullptr): instance i IS query i and its mask is written to every slot
// dest_slot[dest_off[i] .. dest_off[i+1]) of `masks` -- the detections of all evaluation types that picked this query --
// so each query's mask is interpolated once and no gather pass over the (n, H, W) masks follows. Queries nobody
// picked exit immediately.
__global__ __launch_bounds__(256) void cgg_instance_masks_kernel(const float* __restrict__ logits,
                                                                 const int32_t* __restrict__ sel,
                                                                 uint8_t* __restrict__ masks,
                                                                 int32_t* __restrict__ ws,
                                                                 ResizeGeom g,
                                                                 const int32_t* __restrict__ dest_off,
                                                                 const int32_t* __restrict__ dest_slot) {
  const int i = blockIdx.y;
  int d0 = i, d1 = i + 1;
  if (dest_off != nullptr) {
    d0 = dest_off[i];
    d1 = dest_off[i + 1];
    if (d0 == d1) return;                                  // block-uniform
  }
  const long long npix = (long long)g.out_h * g.out_w;
  const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * IM_PPT;
  const float* s = logits + (size_t)(dest_off != nullptr ? i : sel[i]) * g.H * g.W;
  float sig = 0.f;
  int cnt = 0, xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
  uint32_t packed[IM_PPT / 4] = {0u, 0u, 0u, 0u};
  if (p0 < npix) {
    int oy = (int)(p0 / g.out_w), ox = (int)(p0 - (long long)oy * g.out_w);
#pragma unroll
    for (int k = 0; k < IM_PPT; ++k) {
      if (p0 + k < npix) {
        const float v = cgg_resized_logit(s, g, oy, ox);
        if (v > 0.f) {
          packed[k >> 2] |= 1u << (8 * (k & 3));
          sig += cgg_sigmoid(v);
          cnt += 1;
          xmin = min(xmin, ox); xmax = max(xmax, ox);
          ymin = min(ymin, oy); ymax = max(ymax, oy);
        }
      }
      if (++ox == g.out_w) { ox = 0; ++oy; }
    }
    for (int d = d0; d < d1; ++d) {
      const size_t slot = dest_off != nullptr ? (size_t)dest_slot[d] : (size_t)i;
      uint8_t* dst = masks + slot * npix + p0;
      if (p0 + IM_PPT <= npix && (((slot * npix + p0) & 15) == 0)) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
      } else {
        for (int k = 0; k < IM_PPT && p0 + k < npix; ++k) dst[k] = (packed[k >> 2] >> (8 * (k & 3))) & 0xff;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    sig += __shfl_xor(sig, o);
    cnt += __shfl_xor(cnt, o);
    xmin = min(xmin, __shfl_xor(xmin, o));
    ymin = min(ymin, __shfl_xor(ymin, o));
    xmax = max(xmax, __shfl_xor(xmax, o));
    ymax = max(ymax, __shfl_xor(ymax, o));
  }
  __shared__ float s_sig[4];
  __shared__ int s_i[4][5];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_sig[wave] = sig;
    s_i[wave][0] = cnt; s_i[wave][1] = xmin; s_i[wave][2] = ymin; s_i[wave][3] = xmax; s_i[wave][4] = ymax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      sig += s_sig[w];
      cnt += s_i[w][0];
      xmin = min(xmin, s_i[w][1]); ymin = min(ymin, s_i[w][2]);
      xmax = max(xmax, s_i[w][3]); ymax = max(ymax, s_i[w][4]);
    }
    if (cnt > 0) {
      int32_t* w = ws + (size_t)i * 8;
      atomicAdd(reinterpret_cast<float*>(w), sig);
      atomicAdd(w + 1, cnt);
      atomicMin(w + 2, xmin);
      atomicMin(w + 3, ymin);
      atomicMax(w + 4, xmax);
      atomicMax(w + 5, ymax);
    }
  }
}

// Fast path: one-stage resize by an INTEGER factor S (up == S * low-res, out == crop, out_w % 16 == 0).
// src = (o + .5)/S - .5: the tap indices / fractions repeat with period S, so a thread that owns 16
// consecutive output pixels needs only 16/S + 2 source columns of two source rows (12 loads for S = 4
// instead of 64) and compile-time weights. Same association order as torch:
// hy*(hx*a + lx*b) + ly*(hx*c + lx*d).
// BITS: the masks leave bit-packed, [slot][out_h][out_w / 8] bytes with pixel x in bit (x & 7) of byte x >> 3 (numpy
// unpackbits(bitorder='little')): 8x fewer bytes to write and, for host-side consumers, to copy over PCIe.
template <int S, bool BITS>
__global__ __launch_bounds__(256) void cgg_instance_masks_int_kernel(const float* __restrict__ logits,
                                                                     const int32_t* __restrict__ sel,
                                                                     uint8_t* __restrict__ masks,
                                                                     int32_t* __restrict__ ws,
                                                                     ResizeGeom g,
                                                                     const int32_t* __restrict__ dest_off,
                                                                     const int32_t* __restrict__ dest_slot) {
  // thread = a 16-wide x S-tall block of output pixels = source row m (and its neighbours m-1, m+1) x
  // 16/S + 2 source columns: 3 * NC loads feed 16 * S pixels. (The first version gave each thread ONE output
  // row: 2 * NC dependent loads for 16 pixels left the kernel latency-bound at ~0.5 TB/s of mask bytes.)
  constexpr int NC = IM_PPT / S + 2;  // source columns per thread
  const int i = blockIdx.y;
  const int tiles_x = g.out_w / IM_PPT;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // tile index: (m, x-tile)
  const long long ntile = (long long)((g.out_h + S - 1) / S) * tiles_x;
  const long long npix = (long long)g.out_h * g.out_w;
  int d0 = i, d1 = i + 1;
  if (dest_off != nullptr) {
    d0 = dest_off[i];
    d1 = dest_off[i + 1];
    if (d0 == d1) return;                                  // block-uniform: nobody picked this query
  }
  const float* s = logits + (size_t)(dest_off != nullptr ? i : sel[i]) * g.H * g.W;
  float sig = 0.f;
  int cnt = 0, xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
  if (t < ntile) {
    const int m = (int)(t / tiles_x), ox0 = (int)(t - (long long)m * tiles_x) * IM_PPT;
    const int m0 = ox0 / S;
    const int rm = max(m - 1, 0), rp = min(m + 1, g.H - 1);
    float ra[NC], rb[NC], rc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = min(max(m0 - 1 + c, 0), g.W - 1);
      ra[c] = s[(size_t)rm * g.W + col];
      rb[c] = s[(size_t)m * g.W + col];
      rc[c] = s[(size_t)rp * g.W + col];
    }
#pragma unroll
    for (int ry = 0; ry < S; ++ry) {
      constexpr float inv = 1.f / (float)S;
      const int oy = S * m + ry;
      if (oy >= g.out_h) break;
      const bool lowy = (2 * ry + 1) < S;             // src row < m -> taps (m-1, m), else (m, m+1)
      float ly1 = ((float)ry + 0.5f) * inv + (lowy ? 0.5f : -0.5f);
      if (lowy && m == 0) ly1 = 0.f;                  // top border: src clamps to 0 (torch: lambda = 0)
      const float ly0 = 1.f - ly1;
      uint32_t packed[IM_PPT / 4] = {0u, 0u, 0u, 0u};
      uint32_t run = 0u;
#pragma unroll
      for (int k = 0; k < IM_PPT; ++k) {
        const int r = k % S, mrel = k / S;              // ox = S*(m0 + mrel) + r
        const bool lowhalf = (2 * r + 1) < S;           // src < m  -> taps (m-1, m)
        const int c0 = mrel + (lowhalf ? 0 : 1);        // index into the row arrays (column m0-1+c0)
        float l1 = ((float)r + 0.5f) * inv + (lowhalf ? 0.5f : -0.5f);
        if (lowhalf && m0 + mrel == 0) l1 = 0.f;        // left border
        const float l0 = 1.f - l1;
        const float top = lowy ? (l0 * ra[c0] + l1 * ra[c0 + 1]) : (l0 * rb[c0] + l1 * rb[c0 + 1]);
        const float bot = lowy ? (l0 * rb[c0] + l1 * rb[c0 + 1]) : (l0 * rc[c0] + l1 * rc[c0 + 1]);
        const float v = ly0 * top + ly1 * bot;
        const bool on = v > 0.f;
        run |= (on ? 1u : 0u) << k;
        packed[k >> 2] |= (on ? 1u : 0u) << (8 * (k & 3));
        // sigmoid as v_exp_f32 + v_rcp_f32 (1 ulp each): `1.f / x` and __frcp_rn expand to the 10-instruction IEEE
        // division sequence, which made this kernel VALU-bound at 45 instructions per pixel
        sig += on ? __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)) : 0.f;
      }
      if (run) {
        cnt += __popc(run);
        xmin = min(xmin, ox0 + (__ffs(run) - 1));
        xmax = max(xmax, ox0 + (31 - __clz(run)));
        ymin = min(ymin, oy);
        ymax = max(ymax, oy);
      }
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
      const u32x4_t pkv = {packed[0], packed[1], packed[2], packed[3]};
      for (int d = d0; d < d1; ++d) {
        const size_t slot = dest_off != nullptr ? (size_t)dest_slot[d] : (size_t)i;
        if (BITS)
          reinterpret_cast<uint16_t*>(masks)[(slot * npix + (size_t)oy * g.out_w + ox0) >> 4] = (uint16_t)run;
        else
          __builtin_nontemporal_store(pkv, reinterpret_cast<u32x4_t*>(masks + slot * npix + (size_t)oy * g.out_w + ox0));
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    sig += __shfl_xor(sig, o);
    cnt += __shfl_xor(cnt, o);
    xmin = min(xmin, __shfl_xor(xmin, o));
    ymin = min(ymin, __shfl_xor(ymin, o));
    xmax = max(xmax, __shfl_xor(xmax, o));
    ymax = max(ymax, __shfl_xor(ymax, o));
  }
  __shared__ float s_sig[4];
  __shared__ int s_i[4][5];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_sig[wave] = sig;
    s_i[wave][0] = cnt; s_i[wave][1] = xmin; s_i[wave][2] = ymin; s_i[wave][3] = xmax; s_i[wave][4] = ymax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      sig += s_sig[w];
      cnt += s_i[w][0];
      xmin = min(xmin, s_i[w][1]); ymin = min(ymin, s_i[w][2]);
      xmax = max(xmax, s_i[w][3]); ymax = max(ymax, s_i[w][4]);
    }
    if (cnt > 0) {
      int32_t* w = ws + (size_t)i * 8;
      atomicAdd(reinterpret_cast<float*>(w), sig);
      atomicAdd(w + 1, cnt);
      atomicMin(w + 2, xmin);
      atomicMin(w + 3, ymin);
      atomicMax(w + 4, xmax);
      atomicMax(w + 5, ymax);
    }
  }
}

__global__ void cgg_instance_init_kernel(int32_t* __restrict__ ws, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t* w = ws + (size_t)i * 8;
  w[0] = 0; w[1] = 0; w[2] = 0x7fffffff; w[3] = 0x7fffffff; w[4] = -1; w[5] = -1; w[6] = 0; w[7] = 0;
}

__global__ void cgg_instance_final_kernel(const int32_t* __restrict__ ws, float* __restrict__ score,
                                          float* __restrict__ bbox, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t* w = ws + (size_t)i * 8;
  const float sum = __int_as_float(w[0]);
  const int cnt = w[1];
  score[i] = sum / ((float)cnt + 1e-6f);
  if (cnt > 0) {
    bbox[4 * i] = (float)w[2];
    bbox[4 * i + 1] = (float)w[3];
    bbox[4 * i + 2] = (float)(w[4] + 1);
    bbox[4 * i + 3] = (float)(w[5] + 1);
  } else {
    bbox[4 * i] = bbox[4 * i + 1] = bbox[4 * i + 2] = bbox[4 * i + 3] = 0.f;
  }
}

// panoptic: per output pixel argmax_k score[k]*sigmoid(logit_keep[k]); first max wins (torch.argmax)
// counts [n, 3] i32: #(id==k), #(sigmoid_k >= 0.5), #(id==k && sigmoid_k >= 0.5)
__global__ __launch_bounds__(256) void cgg_panoptic_argmax_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ keep, const float* __restrict__ score,
    int32_t* __restrict__ ids, uint8_t* __restrict__ win_half, int32_t* __restrict__ counts,
    ResizeGeom g, int n) {
  extern __shared__ int32_t hist[];  // [n*3]
  for (int i = threadIdx.x; i < n * 3; i += 256) hist[i] = 0;
  __syncthreads();
  const long long npix = (long long)g.out_h * g.out_w;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < npix) {
    const int oy = (int)(p / g.out_w), ox = (int)(p - (long long)oy * g.out_w);
    float best = -INFINITY;
    int bk = 0;
    bool bhalf = false;
    for (int k = 0; k < n; ++k) {
      const float v = cgg_resized_logit(logits + (size_t)keep[k] * g.H * g.W, g, oy, ox);
      const float sg = cgg_sigmoid(v);
      const bool half = sg >= 0.5f;
      if (half) atomicAdd(&hist[k * 3 + 1], 1);
      const float pv = score[k] * sg;
      if (pv > best) {
        best = pv;
        bk = k;
        bhalf = half;
      }
    }
    ids[p] = bk;
    win_half[p] = bhalf ? 1 : 0;
    atomicAdd(&hist[bk * 3], 1);
    if (bhalf) atomicAdd(&hist[bk * 3 + 2], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n * 3; i += 256)
    if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// paint: seg[p] = lut_val[id] if lut_val[id] >= 0 and (!lut_half[id] || win_half[p]) else void
__global__ __launch_bounds__(256) void cgg_panoptic_paint_kernel(const int32_t* __restrict__ ids,
                                                                 const uint8_t* __restrict__ win_half,
                                                                 const int32_t* __restrict__ lut_val,
                                                                 const int32_t* __restrict__ lut_half,
                                                                 int32_t* __restrict__ seg,
                                                                 long long npix, int void_label) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int k = ids[p];
  const int v = lut_val[k];
  const bool ok = v >= 0 && (!lut_half[k] || win_half[p]);
  seg[p] = ok ? v : void_label;
}

// row-wise softmax + (max, argmax); one wavefront per row, first max wins
__global__ __launch_bounds__(256) void cgg_softmax_argmax_kernel(const float* __restrict__ x,
                                                                 float* __restrict__ prob,
                                                                 float* __restrict__ maxv,
                                                                 int64_t* __restrict__ argmax,
                                                                 int rows, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * n;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int i = lane; i < n; i += 64) {
    const float v = xr[i];
    if (v > m || (v == m && i < mi)) { m = v; mi = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o);
    const int oi = __shfl_xor(mi, o);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) sum += expf(xr[i] - m);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  if (prob != nullptr)
    for (int i = lane; i < n; i += 64) prob[(size_t)row * n + i] = expf(xr[i] - m) * inv;
  if (lane == 0) {
    if (maxv) maxv[row] = inv;  // exp(m - m) / sum
    if (argmax) argmax[row] = mi;
  }
}

// -------------------------------------------------------------------------------------------------
extern "C" int cgg_upsample_bilinear(const float* x, float* y, int N, int H, int W, int h, int w,
                                     cgg_stream_t stream) {
  CGG_REQUIRE(x && y, CGG_EINVAL, "cgg_upsample_bilinear: null pointer");
  CGG_REQUIRE(N > 0 && H > 0 && W > 0 && h > 0 && w > 0, CGG_EINVAL, "cgg_upsample_bilinear: bad sizes");
  const long long npix = (long long)h * w;
  hipLaunchKernelGGL(cgg_upsample_kernel, dim3((unsigned)((npix + 255) / 256), N), dim3(256), 0,
                     (hipStream_t)stream, x, y, H, W, h, w, (float)H / (float)h, (float)W / (float)w);
  CGG_CHECK_LAUNCH("cgg_upsample_bilinear");
  return CGG_OK;
}

static int instance_masks_launch(const float* logits, const int32_t* sel, const int32_t* dest_off,
                                 const int32_t* dest_slot, uint8_t* masks, float* mask_score, float* bbox, void* ws,
                                 int Q, int H, int W, int up_h, int up_w, int crop_h, int crop_w, int out_h, int out_w,
                                 int n, cgg_stream_t stream);

extern "C" int cgg_instance_masks(const float* logits, const int32_t* sel, uint8_t* masks,
                                  float* mask_score, float* bbox, void* ws, int Q, int H, int W,
                                  int up_h, int up_w, int crop_h, int crop_w, int out_h, int out_w,
                                  int n, cgg_stream_t stream) {
  CGG_REQUIRE(sel != nullptr, CGG_EINVAL, "cgg_instance_masks: null pointer");
  return instance_masks_launch(logits, sel, nullptr, nullptr, masks, mask_score, bbox, ws, Q, H, W, up_h, up_w, crop_h,
                               crop_w, out_h, out_w, n, stream);
}

extern "C" int cgg_instance_masks_multi(const float* logits, const int32_t* dest_off, const int32_t* dest_slot,
                                        uint8_t* masks, float* mask_score, float* bbox, void* ws, int Q, int H, int W,
                                        int up_h, int up_w, int crop_h, int crop_w, int out_h, int out_w,
                                        cgg_stream_t stream) {
  CGG_REQUIRE(dest_off && dest_slot, CGG_EINVAL, "cgg_instance_masks_multi: null pointer");
  return instance_masks_launch(logits, nullptr, dest_off, dest_slot, masks, mask_score, bbox, ws, Q, H, W, up_h, up_w,
                               crop_h, crop_w, out_h, out_w, Q, stream);
}

// the mask pass itself (between the workspace init and the per-instance finalisation)
static bool instance_masks_dispatch(const float* logits, const int32_t* sel, const int32_t* dest_off,
                                    const int32_t* dest_slot, uint8_t* masks, int32_t* ws, const ResizeGeom& g, int H, int W,
                                    int up_h, int up_w, int out_h, int out_w, int n, hipStream_t s, bool bits = false) {
  const long long npix = (long long)out_h * out_w;
  const long long per_block = 256LL * IM_PPT;
  const dim3 grid((unsigned)((npix + per_block - 1) / per_block), n);
  const int S = up_h / H;
  const bool int_path = !g.two_stage && S * H == up_h && S * W == up_w && (out_w % IM_PPT) == 0 &&
                        (((uintptr_t)masks) & 15) == 0;
  const long long ntile = (long long)((out_h + S - 1) / S) * (out_w / IM_PPT);   // one thread per 16 x S block
  const dim3 tgrid((unsigned)((ntile + 255) / 256), n);
  if (bits) {
    if (!(int_path && (S == 2 || S == 4 || S == 8))) return false;     // bit-packed output: integer-scale path only
    if (S == 4)
      hipLaunchKernelGGL((cgg_instance_masks_int_kernel<4, true>), tgrid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
    else if (S == 2)
      hipLaunchKernelGGL((cgg_instance_masks_int_kernel<2, true>), tgrid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
    else
      hipLaunchKernelGGL((cgg_instance_masks_int_kernel<8, true>), tgrid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
    return true;
  }
  if (int_path && S == 4)
    hipLaunchKernelGGL((cgg_instance_masks_int_kernel<4, false>), tgrid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
  else if (int_path && S == 2)
    hipLaunchKernelGGL((cgg_instance_masks_int_kernel<2, false>), tgrid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
  else if (int_path && S == 8)
    hipLaunchKernelGGL((cgg_instance_masks_int_kernel<8, false>), tgrid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
  else
    hipLaunchKernelGGL(cgg_instance_masks_kernel, grid, dim3(256), 0, s, logits, sel, masks, ws, g, dest_off, dest_slot);
  return true;
}

static int instance_masks_launch(const float* logits, const int32_t* sel, const int32_t* dest_off,
                                 const int32_t* dest_slot, uint8_t* masks, float* mask_score, float* bbox, void* ws,
                                 int Q, int H, int W, int up_h, int up_w, int crop_h, int crop_w, int out_h, int out_w,
                                 int n, cgg_stream_t stream) {
  CGG_REQUIRE(logits && masks && mask_score && bbox && ws, CGG_EINVAL,
              "cgg_instance_masks: null pointer");
  CGG_REQUIRE(Q > 0 && H > 0 && W > 0 && up_h > 0 && up_w > 0 && out_h > 0 && out_w > 0 && n > 0,
              CGG_EINVAL, "cgg_instance_masks: bad sizes");
  CGG_REQUIRE(crop_h > 0 && crop_h <= up_h && crop_w > 0 && crop_w <= up_w, CGG_EINVAL,
              "cgg_instance_masks: crop %dx%d outside %dx%d", crop_h, crop_w, up_h, up_w);
  hipStream_t s = (hipStream_t)stream;
  const ResizeGeom g = make_geom(H, W, up_h, up_w, crop_h, crop_w, out_h, out_w);
  hipLaunchKernelGGL(cgg_instance_init_kernel, dim3((n + 63) / 64), dim3(64), 0, s, (int32_t*)ws, n);
  instance_masks_dispatch(logits, sel, dest_off, dest_slot, masks, (int32_t*)ws, g, H, W, up_h, up_w, out_h, out_w, n, s);
  hipLaunchKernelGGL(cgg_instance_final_kernel, dim3((n + 63) / 64), dim3(64), 0, s,
                     (const int32_t*)ws, mask_score, bbox, n);
  CGG_CHECK_LAUNCH("cgg_instance_masks");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// Open-vocabulary instance tail in four launches per image (maskformer_fusion_head.py:317-363 for every evaluation
// type at once): (query, class) picks straight from the class-embedding dot products, the slot plan of the
// multi-destination mask pass, the mask pass, and the per-detection boxes / scores.

struct ClsTypes { int n; int col0[8]; int ncols[8]; };

// One workgroup (16 wavefronts) per (evaluation type, image). dots [B*Q, ld]: row q of image b holds emb_q . E_c for
// the concatenated class tables; type t owns columns col0[t] .. col0[t]+ncols[t], the last of them the background class.
//   prob = softmax over the type's columns (same arithmetic, lane order and reductions as cgg_softmax_argmax_kernel),
//   background column dropped (:340), then the k best of the Q*(ncols-1) (query, class) pairs (:342-347).
// torch.topk(sorted=False) leaves the order (and the choice among equal scores) unspecified; here both are fixed:
// descending score, ties by ascending flat index q*(ncols-1)+c. Selection = 4-pass byte radix select on the float bits
// (probabilities are >= 0, so the unsigned order is the float order); the k winners are ordered by rank counting.
// Everything that would be a serial single-thread loop over LDS (64+ clocks per dependent access) is a wavefront scan.
#define CT_WAVES 16

// exclusive prefix sum over n ints in LDS by ONE wavefront (in place); returns the total
__device__ __forceinline__ uint32_t cgg_wave_excl_scan(uint32_t* a, int n, int lane) {
  uint32_t carry = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const uint32_t c = i < n ? a[i] : 0u;
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o);
      if (lane >= o) inc += up;
    }
    if (i < n) a[i] = carry + inc - c;
    carry += __shfl(inc, 63);
  }
  return carry;
}

__global__ __launch_bounds__(64 * CT_WAVES) void cgg_class_topk_kernel(const float* __restrict__ dots, int ld, int Q,
                                                                       ClsTypes ty, int k, int64_t* __restrict__ labels,
                                                                       float* __restrict__ scores,
                                                                       int64_t* __restrict__ qidx) {
  extern __shared__ unsigned char smem_raw[];
  const int t = blockIdx.x, b = blockIdx.y, T = ty.n;
  const int nc = ty.ncols[t], n = nc - 1, M = Q * n;
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem_raw);        // [k]
  uint32_t* prob = reinterpret_cast<uint32_t*>(cand + k);                             // [M] float bits, row-major (q, c)
  uint32_t* rowcnt = prob + M;                                                        // [Q]
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sel_prefix, sel_remaining, ncand;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // ---- softmax rows (one wavefront per row; up to 256 columns live in registers) ----
  for (int q = wave; q < Q; q += CT_WAVES) {
    const float* xr = dots + ((size_t)b * Q + q) * ld + ty.col0[t];
    if (nc <= 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = lane + 64 * u < nc ? xr[lane + 64 * u] : -INFINITY;
      float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = expf(v[u] - m);
        if (lane + 64 * u < nc) sum += v[u];
      }
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.f / sum;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < n) prob[q * n + lane + 64 * u] = __float_as_uint(v[u] * inv);
    } else {
      float m = -INFINITY;
      for (int i = lane; i < nc; i += 64) m = fmaxf(m, xr[i]);
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float sum = 0.f;
      for (int i = lane; i < nc; i += 64) sum += expf(xr[i] - m);
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.f / sum;
      for (int i = lane; i < n; i += 64) prob[q * n + i] = __float_as_uint(expf(xr[i] - m) * inv);
    }
  }
  if (tid == 0) { sel_prefix = 0u; sel_remaining = (uint32_t)k; ncand = 0u; }
  __syncthreads();
  // ---- radix select: bits of the k-th largest value, and how many values equal to it belong to the top k ----
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const uint32_t prefix = sel_prefix;
    for (int i = tid; i < M; i += 64 * CT_WAVES) {
      const uint32_t v = prob[i];
      if (pass == 0 || (v >> (shift + 8)) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      // lane owns bins 4*lane .. 4*lane+3; `above` = count in all higher bins
      const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const uint32_t mine = c0 + c1 + c2 + c3;
      uint32_t suf = mine;                                       // inclusive suffix sum over lanes >= lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t dn = __shfl_down(suf, o);
        if (lane + o < 64) suf += dn;
      }
      const uint32_t above = suf - mine, rem = sel_remaining;
      if (above < rem && rem <= suf) {                            // exactly one lane
        uint32_t r = rem - above;
        int bin;
        if (c3 >= r) bin = 3;
        else if (c3 + c2 >= r) { bin = 2; r -= c3; }
        else if (c3 + c2 + c1 >= r) { bin = 1; r -= c3 + c2; }
        else { bin = 0; r -= c3 + c2 + c1; }
        sel_prefix = (prefix << 8) | (uint32_t)(4 * lane + bin);
        sel_remaining = r;
      }
    }
    __syncthreads();
  }
  const uint32_t thr = sel_prefix, need_eq = sel_remaining;
  // ---- winners: everything above the threshold, plus the first need_eq values equal to it in flat-index order ----
  for (int q = wave; q < Q; q += CT_WAVES) {
    uint32_t c = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const bool eq = i0 + lane < n && prob[q * n + i0 + lane] == thr;
      c += (uint32_t)__popcll(__ballot(eq));
    }
    if (lane == 0) rowcnt[q] = c;
  }
  __syncthreads();
  if (wave == 0) cgg_wave_excl_scan(rowcnt, Q, lane);
  __syncthreads();
  for (int q = wave; q < Q; q += CT_WAVES) {
    uint32_t base = rowcnt[q];
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int c = i0 + lane;
      const uint32_t v = c < n ? prob[q * n + c] : 0u;
      const bool eq = c < n && v == thr;
      const unsigned long long mask = __ballot(eq);
      const uint32_t rank = base + (uint32_t)__popcll(mask & lt_mask);
      if (c < n && (v > thr || (eq && rank < need_eq))) {
        const uint32_t pos = atomicAdd(&ncand, 1u);
        if (pos < (uint32_t)k)
          cand[pos] = ((unsigned long long)v << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(q * n + c));
      }
      base += (uint32_t)__popcll(mask);
    }
  }
  __syncthreads();
  // ---- order: rank of a winner = number of winners with a larger (score bits, ~index) key (keys are unique) ----
  const size_t out0 = ((size_t)b * T + t) * k;
  for (int j = tid; j < k; j += 64 * CT_WAVES) {
    const unsigned long long key = cand[j];
    int rank = 0;
#pragma unroll 8
    for (int i = 0; i < k; ++i) rank += cand[i] > key ? 1 : 0;
    const uint32_t idx = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
    scores[out0 + rank] = __uint_as_float((uint32_t)(key >> 32));
    labels[out0 + rank] = (int64_t)(idx % (uint32_t)n);
    qidx[out0 + rank] = (int64_t)(idx / (uint32_t)n);
  }
}

extern "C" int cgg_class_topk(const float* dots, int ld, int B, int Q, int n_types, const int* col0_host,
                              const int* ncols_host, int k, int64_t* labels, float* scores, int64_t* qidx,
                              cgg_stream_t stream) {
  CGG_REQUIRE(dots && col0_host && ncols_host && labels && scores && qidx, CGG_EINVAL, "cgg_class_topk: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && ld > 0 && k > 0, CGG_EINVAL, "cgg_class_topk: bad sizes");
  CGG_REQUIRE(n_types >= 1 && n_types <= 8, CGG_EUNSUPPORTED, "cgg_class_topk: n_types=%d (1..8)", n_types);
  CGG_REQUIRE(k <= 1024, CGG_EUNSUPPORTED, "cgg_class_topk: k=%d > 1024", k);
  ClsTypes ty;
  ty.n = n_types;
  int max_m = 0;
  for (int t = 0; t < n_types; ++t) {
    ty.col0[t] = col0_host[t];
    ty.ncols[t] = ncols_host[t];
    CGG_REQUIRE(ty.ncols[t] >= 2 && ty.col0[t] >= 0 && ty.col0[t] + ty.ncols[t] <= ld, CGG_EINVAL,
                "cgg_class_topk: type %d columns [%d, +%d) outside ld=%d", t, ty.col0[t], ty.ncols[t], ld);
    const long long m = (long long)Q * (ty.ncols[t] - 1);
    CGG_REQUIRE(m >= k, CGG_EINVAL, "cgg_class_topk: type %d has %lld (query, class) pairs < k=%d", t, m, k);
    CGG_REQUIRE(m <= (1 << 20), CGG_EUNSUPPORTED, "cgg_class_topk: type %d has %lld pairs", t, m);
    if ((int)m > max_m) max_m = (int)m;
  }
  const size_t lds = (size_t)k * 8 + (size_t)max_m * 4 + (size_t)Q * 4;
  CGG_REQUIRE(lds <= 62 * 1024, CGG_EUNSUPPORTED, "cgg_class_topk: Q=%d x %d classes, k=%d need %zu B of LDS (> 62 KiB)", Q,
              max_m / Q, k, lds);
  hipLaunchKernelGGL(cgg_class_topk_kernel, dim3(n_types, B), dim3(64 * CT_WAVES), lds, (hipStream_t)stream, dots, ld, Q,
                     ty, k, labels, scores, qidx);
  CGG_CHECK_LAUNCH("cgg_class_topk");
  return CGG_OK;
}

// workspace init + slot plan: dest_off[q] .. dest_off[q+1] list (in pick order) the detections that picked query q.
// ws = [Q*8 stats | Q+1 dest_off | n_picks dest_slot] int32.
__global__ __launch_bounds__(256) void cgg_instance_plan_kernel(const int64_t* __restrict__ qidx, int n_picks, int Q,
                                                                int32_t* __restrict__ ws) {
  extern __shared__ int32_t pq[];                 // [n_picks] query of every pick, then [Q + 1] offsets
  uint32_t* off = reinterpret_cast<uint32_t*>(pq + n_picks);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int j = tid; j < n_picks; j += 256) {
    const long long q = qidx[j];
    pq[j] = (q >= 0 && q < Q) ? (int32_t)q : -1;
  }
  for (int i = tid; i < Q; i += 256) {
    int32_t* w = ws + (size_t)i * 8;
    w[0] = 0; w[1] = 0; w[2] = 0x7fffffff; w[3] = 0x7fffffff; w[4] = -1; w[5] = -1; w[6] = 0; w[7] = 0;
    off[i] = 0u;
  }
  __syncthreads();
  for (int j = tid; j < n_picks; j += 256)
    if (pq[j] >= 0) atomicAdd(&off[pq[j]], 1u);
  __syncthreads();
  if (tid < 64) {
    const uint32_t total = cgg_wave_excl_scan(off, Q, lane);
    if (lane == 0) off[Q] = total;
  }
  __syncthreads();
  int32_t* g_off = ws + (size_t)Q * 8;
  int32_t* g_slot = g_off + Q + 1;
  for (int q = tid; q <= Q; q += 256) g_off[q] = (int32_t)off[q];
  for (int j = tid; j < n_picks; j += 256) {
    const int q = pq[j];
    if (q < 0) continue;
    int r = 0;                                     // picks of the same query before this one
#pragma unroll 8
    for (int i = 0; i < j; ++i) r += pq[i] == q ? 1 : 0;
    g_slot[off[q] + r] = j;
  }
}

// per-query mask score / box (cgg_instance_final_kernel) followed by the per-detection rows (:349-363):
// bboxes[j] = (box of query qidx[j], cls_score[j] * mask_score[qidx[j]]).
__global__ __launch_bounds__(256) void cgg_instance_final_picks_kernel(const int32_t* __restrict__ ws,
                                                                       const int64_t* __restrict__ qidx,
                                                                       const float* __restrict__ cls_scores, int n_picks,
                                                                       int Q, float* __restrict__ bboxes) {
  extern __shared__ float qs[];                    // [Q][5]: x0, y0, x1, y1, mask score
  const int tid = threadIdx.x;
  for (int i = tid; i < Q; i += 256) {
    const int32_t* w = ws + (size_t)i * 8;
    const float sum = __int_as_float(w[0]);
    const int cnt = w[1];
    qs[5 * i + 4] = sum / ((float)cnt + 1e-6f);
    if (cnt > 0) {
      qs[5 * i] = (float)w[2]; qs[5 * i + 1] = (float)w[3]; qs[5 * i + 2] = (float)(w[4] + 1); qs[5 * i + 3] = (float)(w[5] + 1);
    } else {
      qs[5 * i] = qs[5 * i + 1] = qs[5 * i + 2] = qs[5 * i + 3] = 0.f;
    }
  }
  __syncthreads();
  for (int j = tid; j < n_picks; j += 256) {
    const long long q = qidx[j];
    float* o = bboxes + (size_t)j * 5;
    if (q < 0 || q >= Q) { o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; continue; }
    o[0] = qs[5 * q]; o[1] = qs[5 * q + 1]; o[2] = qs[5 * q + 2]; o[3] = qs[5 * q + 3];
    o[4] = cls_scores[j] * qs[5 * q + 4];
  }
}

extern "C" int64_t cgg_instance_masks_picks_workspace_bytes(int Q, int n_picks) {
  return (int64_t)(((size_t)Q * 8 + (size_t)Q + 1 + (size_t)n_picks) * sizeof(int32_t));
}

extern "C" int cgg_instance_masks_picks(const float* logits, const int64_t* qidx, const float* cls_scores, int n_picks,
                                        uint8_t* masks, float* bboxes, void* ws, int Q, int H, int W, int up_h, int up_w,
                                        int crop_h, int crop_w, int out_h, int out_w, int bitpack, cgg_stream_t stream) {
  CGG_REQUIRE(logits && qidx && cls_scores && masks && bboxes && ws, CGG_EINVAL, "cgg_instance_masks_picks: null pointer");
  CGG_REQUIRE(Q > 0 && H > 0 && W > 0 && up_h > 0 && up_w > 0 && out_h > 0 && out_w > 0 && n_picks > 0, CGG_EINVAL,
              "cgg_instance_masks_picks: bad sizes");
  CGG_REQUIRE(crop_h > 0 && crop_h <= up_h && crop_w > 0 && crop_w <= up_w, CGG_EINVAL,
              "cgg_instance_masks_picks: crop %dx%d outside %dx%d", crop_h, crop_w, up_h, up_w);
  CGG_REQUIRE((size_t)(n_picks + Q + 1) * 4 <= 60000 && (size_t)Q * 20 <= 60000, CGG_EUNSUPPORTED,
              "cgg_instance_masks_picks: Q=%d, n_picks=%d do not fit LDS", Q, n_picks);
  hipStream_t s = (hipStream_t)stream;
  const ResizeGeom g = make_geom(H, W, up_h, up_w, crop_h, crop_w, out_h, out_w);
  int32_t* wsi = (int32_t*)ws;
  hipLaunchKernelGGL(cgg_instance_plan_kernel, dim3(1), dim3(256), (size_t)(n_picks + Q + 1) * 4, s, qidx, n_picks, Q, wsi);
  const bool ok = instance_masks_dispatch(logits, nullptr, wsi + (size_t)Q * 8, wsi + (size_t)Q * 9 + 1, masks, wsi, g, H, W,
                                          up_h, up_w, out_h, out_w, Q, s, bitpack != 0);
  CGG_REQUIRE(ok, CGG_EUNSUPPORTED,
              "cgg_instance_masks_picks: bit-packed masks need an integer up-scale of 2 / 4 / 8, no second resize and out_w %% 16 == 0");
  hipLaunchKernelGGL(cgg_instance_final_picks_kernel, dim3(1), dim3(256), (size_t)Q * 20, s, (const int32_t*)wsi, qidx,
                     cls_scores, n_picks, Q, bboxes);
  CGG_CHECK_LAUNCH("cgg_instance_masks_picks");
  return CGG_OK;
}

extern "C" int cgg_panoptic_argmax(const float* logits, const int32_t* keep, const float* score,
                                   int32_t* ids, uint8_t* win_half, int32_t* counts, int Q, int H,
                                   int W, int up_h, int up_w, int crop_h, int crop_w, int out_h,
                                   int out_w, int n, cgg_stream_t stream) {
  CGG_REQUIRE(logits && keep && score && ids && win_half && counts, CGG_EINVAL,
              "cgg_panoptic_argmax: null pointer");
  CGG_REQUIRE(Q > 0 && H > 0 && W > 0 && up_h > 0 && up_w > 0 && out_h > 0 && out_w > 0 && n > 0,
              CGG_EINVAL, "cgg_panoptic_argmax: bad sizes");
  CGG_REQUIRE(n <= 1024, CGG_EUNSUPPORTED, "cgg_panoptic_argmax: n=%d kept queries (max 1024)", n);
  CGG_REQUIRE(crop_h > 0 && crop_h <= up_h && crop_w > 0 && crop_w <= up_w, CGG_EINVAL,
              "cgg_panoptic_argmax: crop outside the upsampled image");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * 3 * n, s);
  if (e != hipSuccess) {
    cgg_set_error("cgg_panoptic_argmax: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  const ResizeGeom g = make_geom(H, W, up_h, up_w, crop_h, crop_w, out_h, out_w);
  const long long npix = (long long)out_h * out_w;
  hipLaunchKernelGGL(cgg_panoptic_argmax_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256),
                     sizeof(int32_t) * 3 * n, s, logits, keep, score, ids, win_half, counts, g, n);
  CGG_CHECK_LAUNCH("cgg_panoptic_argmax");
  return CGG_OK;
}

extern "C" int cgg_panoptic_paint(const int32_t* ids, const uint8_t* win_half, const int32_t* lut_val,
                                  const int32_t* lut_half, int32_t* seg, int64_t npix, int void_label,
                                  cgg_stream_t stream) {
  CGG_REQUIRE(ids && win_half && lut_val && lut_half && seg, CGG_EINVAL, "cgg_panoptic_paint: null pointer");
  CGG_REQUIRE(npix > 0, CGG_EINVAL, "cgg_panoptic_paint: bad size");
  hipLaunchKernelGGL(cgg_panoptic_paint_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, ids, win_half, lut_val, lut_half, seg, (long long)npix,
                     void_label);
  CGG_CHECK_LAUNCH("cgg_panoptic_paint");
  return CGG_OK;
}

extern "C" int cgg_rowwise_softmax_argmax(const float* x, float* prob, float* maxv, int64_t* argmax,
                                          int rows, int n, cgg_stream_t stream) {
  CGG_REQUIRE(x, CGG_EINVAL, "cgg_rowwise_softmax_argmax: null pointer");
  CGG_REQUIRE(rows > 0 && n > 0, CGG_EINVAL, "cgg_rowwise_softmax_argmax: bad sizes");
  hipLaunchKernelGGL(cgg_softmax_argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     x, prob, maxv, argmax, rows, n);
  CGG_CHECK_LAUNCH("cgg_rowwise_softmax_argmax");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// Point sampling of a CHANNEL-LAST f32 map (training: the mask feature at the matching points of all decoder layers;
// sample(E F) = E sample(F), mask2former_head.py:357-366 / [3P] mmcv.ops.point_sample = F.grid_sample(points * 2 - 1,
// bilinear, zeros, align_corners=False)). ATen's grid_sampler_2d walks the C = 256 channel planes of an NCHW tensor per
// point (4 scattered 4-byte loads per channel); here a point's 4 taps are 4 contiguous C * 4-byte rows and C / 4 lanes
// share the geometry. Same coordinate arithmetic and accumulation order as the ATen kernel (unnormalise ((g + 1) * size - 1)
// / 2 on g = 2 p - 1; nw, ne, sw, se).
__global__ __launch_bounds__(256) void cgg_point_sample_nhwc_kernel(const float* __restrict__ feat, const float* __restrict__ pts,
                                                                   float* __restrict__ out, int H, int W, int C4, int P,
                                                                   long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (b, p, c4)
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  const long long bp = i / C4;
  const int b = (int)(bp / P);
  const float px = pts[bp * 2], py = pts[bp * 2 + 1];
  const float gx = __fsub_rn(__fmul_rn(px, 2.0f), 1.0f), gy = __fsub_rn(__fmul_rn(py, 2.0f), 1.0f);
  const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 2.f);
  const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 2.f);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = __fsub_rn(ix, fx), ty = __fsub_rn(iy, fy);             // ix - ix_nw, iy - iy_nw
  const float ux = __fsub_rn(__fadd_rn(fx, 1.f), ix), uy = __fsub_rn(__fadd_rn(fy, 1.f), iy);   // ix_se - ix, iy_se - iy
  const float wnw = __fmul_rn(ux, uy), wne = __fmul_rn(tx, uy), wsw = __fmul_rn(ux, ty), wse = __fmul_rn(tx, ty);
  const f32x4* fb = reinterpret_cast<const f32x4*>(feat) + (size_t)b * H * W * C4 + c4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  f32x4 v;
  if (xin0 && yin0) { v = fb[((size_t)y0 * W + x0) * C4];
    acc[0] = fmaf(v[0], wnw, acc[0]); acc[1] = fmaf(v[1], wnw, acc[1]); acc[2] = fmaf(v[2], wnw, acc[2]); acc[3] = fmaf(v[3], wnw, acc[3]); }
  if (xin1 && yin0) { v = fb[((size_t)y0 * W + x0 + 1) * C4];
    acc[0] = fmaf(v[0], wne, acc[0]); acc[1] = fmaf(v[1], wne, acc[1]); acc[2] = fmaf(v[2], wne, acc[2]); acc[3] = fmaf(v[3], wne, acc[3]); }
  if (xin0 && yin1) { v = fb[((size_t)(y0 + 1) * W + x0) * C4];
    acc[0] = fmaf(v[0], wsw, acc[0]); acc[1] = fmaf(v[1], wsw, acc[1]); acc[2] = fmaf(v[2], wsw, acc[2]); acc[3] = fmaf(v[3], wsw, acc[3]); }
  if (xin1 && yin1) { v = fb[((size_t)(y0 + 1) * W + x0 + 1) * C4];
    acc[0] = fmaf(v[0], wse, acc[0]); acc[1] = fmaf(v[1], wse, acc[1]); acc[2] = fmaf(v[2], wse, acc[2]); acc[3] = fmaf(v[3], wse, acc[3]); }
  reinterpret_cast<f32x4*>(out)[i] = acc;
}

// Single-channel variant with a plane index per row: out[j][p] = bilinear sample of planes[index[j]] (H x W, f32) at
// pts[j][p]. Training losses sample the ASSIGNED ground-truth mask of every matched query at its own 12 544 points
// (mask2former_head.py:609-612): one launch per decoder layer for all images instead of one F.grid_sample per image
// over ALL of the image's masks (G x the work, 160 launches per step). Same arithmetic as the kernel above.
__global__ __launch_bounds__(256) void cgg_point_sample_planes_kernel(const float* __restrict__ planes,
                                                                     const int32_t* __restrict__ index,
                                                                     const float* __restrict__ pts, float* __restrict__ out,
                                                                     int H, int W, int P, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (row j, point p)
  if (i >= total) return;
  const long long j = i / P;
  const float px = pts[i * 2], py = pts[i * 2 + 1];
  const float gx = __fsub_rn(__fmul_rn(px, 2.0f), 1.0f), gy = __fsub_rn(__fmul_rn(py, 2.0f), 1.0f);
  const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 2.f);
  const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 2.f);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = __fsub_rn(ix, fx), ty = __fsub_rn(iy, fy);
  const float ux = __fsub_rn(__fadd_rn(fx, 1.f), ix), uy = __fsub_rn(__fadd_rn(fy, 1.f), iy);
  const float wnw = __fmul_rn(ux, uy), wne = __fmul_rn(tx, uy), wsw = __fmul_rn(ux, ty), wse = __fmul_rn(tx, ty);
  const float* pl = planes + (size_t)index[j] * H * W;
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  float acc = 0.f;
  if (xin0 && yin0) acc = fmaf(pl[(size_t)y0 * W + x0], wnw, acc);
  if (xin1 && yin0) acc = fmaf(pl[(size_t)y0 * W + x0 + 1], wne, acc);
  if (xin0 && yin1) acc = fmaf(pl[(size_t)(y0 + 1) * W + x0], wsw, acc);
  if (xin1 && yin1) acc = fmaf(pl[(size_t)(y0 + 1) * W + x0 + 1], wse, acc);
  out[i] = acc;
}

extern "C" int cgg_point_sample_planes(const float* planes, const int32_t* index, const float* pts, float* out, int N, int H,
                                       int W, int rows, int P, cgg_stream_t stream) {
  CGG_REQUIRE(planes && index && pts && out, CGG_EINVAL, "cgg_point_sample_planes: null pointer");
  CGG_REQUIRE(N > 0 && H > 0 && W > 0 && rows > 0 && P > 0, CGG_EINVAL, "cgg_point_sample_planes: bad sizes");
  const long long total = (long long)rows * P;
  hipLaunchKernelGGL(cgg_point_sample_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     planes, index, pts, out, H, W, P, total);
  CGG_CHECK_LAUNCH("cgg_point_sample_planes");
  return CGG_OK;
}

// Backward of cgg_point_sample_planes wrt the planes (the points are constants of the loss: `get_uncertain_point_coords...` runs
// under no_grad at mask2former_head.py:600-606, so no gradient wrt the grid is needed -- F.grid_sample's backward computes it
// anyway): grad_planes[index[j]] += the four tap weights x grad_out[j][p]; f32 hardware atomics into the caller's zeroed buffer.
__global__ __launch_bounds__(256) void cgg_point_sample_planes_bwd_kernel(const float* __restrict__ gout,
                                                                         const int32_t* __restrict__ index,
                                                                         const float* __restrict__ pts, float* __restrict__ gplanes,
                                                                         int H, int W, int P, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (row j, point p)
  if (i >= total) return;
  const long long j = i / P;
  const float px = pts[i * 2], py = pts[i * 2 + 1];
  const float gx = __fsub_rn(__fmul_rn(px, 2.0f), 1.0f), gy = __fsub_rn(__fmul_rn(py, 2.0f), 1.0f);
  const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 2.f);
  const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 2.f);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = __fsub_rn(ix, fx), ty = __fsub_rn(iy, fy);
  const float ux = __fsub_rn(__fadd_rn(fx, 1.f), ix), uy = __fsub_rn(__fadd_rn(fy, 1.f), iy);
  const float g = gout[i];
  float* pl = gplanes + (size_t)(index ? index[j] : (int)j) * H * W;
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  if (xin0 && yin0) atomicAdd(pl + (size_t)y0 * W + x0, __fmul_rn(__fmul_rn(ux, uy), g));
  if (xin1 && yin0) atomicAdd(pl + (size_t)y0 * W + x0 + 1, __fmul_rn(__fmul_rn(tx, uy), g));
  if (xin0 && yin1) atomicAdd(pl + (size_t)(y0 + 1) * W + x0, __fmul_rn(__fmul_rn(ux, ty), g));
  if (xin1 && yin1) atomicAdd(pl + (size_t)(y0 + 1) * W + x0 + 1, __fmul_rn(__fmul_rn(tx, ty), g));
}

extern "C" int cgg_point_sample_planes_backward(const float* grad_out, const int32_t* index, const float* pts, float* grad_planes,
                                                int N, int H, int W, int rows, int P, cgg_stream_t stream) {
  CGG_REQUIRE(grad_out && pts && grad_planes, CGG_EINVAL, "cgg_point_sample_planes_backward: null pointer");
  CGG_REQUIRE(N > 0 && H > 0 && W > 0 && rows > 0 && P > 0 && (index || rows <= N), CGG_EINVAL,
              "cgg_point_sample_planes_backward: bad sizes");
  const long long total = (long long)rows * P;
  hipLaunchKernelGGL(cgg_point_sample_planes_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out, index, pts, grad_planes, H, W, P, total);
  CGG_CHECK_LAUNCH("cgg_point_sample_planes_backward");
  return CGG_OK;
}

// ... the same gradient WITHOUT global atomics for the case every row has its own plane (index == NULL: the loss' mask planes of the
// positives): a workgroup owns a band of `band` rows of ONE plane in LDS, scans the row's P points, adds the taps that fall into its
// band with LDS float atomics and writes the band out once, coalesced -- every element of grad_planes is written, no zero-fill. The
// atomic form above ran at 21 G atomics/s (436 us for 184 planes x 12 544 points at configs[2], 10 calls per training step).
__global__ __launch_bounds__(256) void cgg_point_sample_planes_bwd_tiled_kernel(const float* __restrict__ gout, const float* __restrict__ pts,
                                                                               float* __restrict__ gplanes, int H, int W, int P, int band) {
  extern __shared__ float tile[];                     // band x W
  const int j = blockIdx.y;
  const int r0 = blockIdx.x * band, r1 = min(H, r0 + band);
  const int n = (r1 - r0) * W;
  for (int i = threadIdx.x; i < n; i += 256) tile[i] = 0.f;
  __syncthreads();
  const float* pj = pts + (size_t)j * P * 2;
  const float* gj = gout + (size_t)j * P;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float2 xy = *reinterpret_cast<const float2*>(pj + 2 * p);
    const float gx = __fsub_rn(__fmul_rn(xy.x, 2.0f), 1.0f), gy = __fsub_rn(__fmul_rn(xy.y, 2.0f), 1.0f);
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 2.f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 2.f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    if (y0 + 1 < r0 || y0 >= r1) continue;            // neither tap row in this band
    const float tx = __fsub_rn(ix, fx), ty = __fsub_rn(iy, fy);
    const float ux = __fsub_rn(__fadd_rn(fx, 1.f), ix), uy = __fsub_rn(__fadd_rn(fy, 1.f), iy);
    const float g = gj[p];
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yin0 = y0 >= r0 && y0 < r1, yin1 = y0 + 1 >= r0 && y0 + 1 < r1;
    float* t0 = tile + (y0 - r0) * W + x0;
    if (xin0 && yin0) atomicAdd(t0, __fmul_rn(__fmul_rn(ux, uy), g));
    if (xin1 && yin0) atomicAdd(t0 + 1, __fmul_rn(__fmul_rn(tx, uy), g));
    if (xin0 && yin1) atomicAdd(t0 + W, __fmul_rn(__fmul_rn(ux, ty), g));
    if (xin1 && yin1) atomicAdd(t0 + W + 1, __fmul_rn(__fmul_rn(tx, ty), g));
  }
  __syncthreads();
  float* o = gplanes + ((size_t)j * H + r0) * W;
  if (n % 4 == 0 && ((uintptr_t)o & 15) == 0) {
    for (int i = threadIdx.x; i < n / 4; i += 256) reinterpret_cast<f32x4*>(o)[i] = reinterpret_cast<const f32x4*>(tile)[i];
  } else {
    for (int i = threadIdx.x; i < n; i += 256) o[i] = tile[i];
  }
}

// grad_planes (rows, H, W) is OVERWRITTEN (no zero-fill needed): row j's gradient goes to plane j.
extern "C" int cgg_point_sample_planes_backward_rows(const float* grad_out, const float* pts, float* grad_planes, int H, int W, int rows,
                                                     int P, cgg_stream_t stream) {
  CGG_REQUIRE(grad_out && pts && grad_planes, CGG_EINVAL, "cgg_point_sample_planes_backward_rows: null pointer");
  CGG_REQUIRE(H > 0 && W > 0 && rows > 0 && P > 0 && W <= 16384 && rows <= 65535, CGG_EINVAL, "cgg_point_sample_planes_backward_rows: bad sizes");
  CGG_REQUIRE(((uintptr_t)pts & 7) == 0, CGG_EALIGN, "cgg_point_sample_planes_backward_rows: points must be 8-byte aligned");
  int band = 16384 / W;                               // 64 KiB of LDS per workgroup (two per CU)
  band = band < 1 ? 1 : (band > H ? H : band);
  const int nbands = (H + band - 1) / band;
  hipLaunchKernelGGL(cgg_point_sample_planes_bwd_tiled_kernel, dim3(nbands, rows), dim3(256), (size_t)band * W * sizeof(float),
                     (hipStream_t)stream, grad_out, pts, grad_planes, H, W, P, band);
  CGG_CHECK_LAUNCH("cgg_point_sample_planes_backward_rows");
  return CGG_OK;
}

extern "C" int cgg_point_sample_nhwc(const float* feat, const float* pts, float* out, int B, int H, int W, int C, int P,
                                     cgg_stream_t stream) {
  CGG_REQUIRE(feat && pts && out, CGG_EINVAL, "cgg_point_sample_nhwc: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && P > 0, CGG_EINVAL, "cgg_point_sample_nhwc: bad sizes");
  CGG_REQUIRE(C % 4 == 0, CGG_EUNSUPPORTED, "cgg_point_sample_nhwc: C %% 4 != 0 (C=%d)", C);
  CGG_REQUIRE(cgg_aligned16(feat) && cgg_aligned16(out), CGG_EALIGN, "cgg_point_sample_nhwc: alignment");
  const long long total = (long long)B * P * (C / 4);
  hipLaunchKernelGGL(cgg_point_sample_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     feat, pts, out, H, W, C / 4, P, total);
  CGG_CHECK_LAUNCH("cgg_point_sample_nhwc");
  return CGG_OK;
}
