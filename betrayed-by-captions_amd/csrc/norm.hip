// GroupNorm (+ optional ReLU) for the pixel decoder's ConvModules (norm_cfg=dict(type='GN', num_groups=32),
// configs/instance/coco_b48n17.py:40; [3P] MSDeformAttnPixelDecoder input/lateral/output convs).
// HBM-bound: at 256x256 a (batch, group) row holds 524288 elements but there are only B*32 = 64 rows, so a
// row-per-block reduction leaves 3/4 of the chip idle. Here every row is split into chunks (pass 1: partial
// sum / sum of squares per chunk, full-chip), and pass 2 folds the partials and normalises.
#include "x3.h"

#define GN_CHUNK 16384  // elements per partial

template <int VEC>
__global__ __launch_bounds__(256) void cgg_gn_partial_kernel(const float* __restrict__ x,
                                                             float* __restrict__ part, long long row_len,
                                                             int chunks) {
  const int row = blockIdx.y, chunk = blockIdx.x;
  const long long beg = (long long)chunk * GN_CHUNK;
  const long long end = min(row_len, beg + GN_CHUNK);
  const float* xr = x + (size_t)row * row_len;
  float s = 0.f, ss = 0.f;
  if (VEC == 4) {  // H*W % 4 == 0 and 16-B aligned rows -> float4 loads
    for (long long i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
      s += (v[0] + v[1]) + (v[2] + v[3]);
      ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  } else {
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
      const float v = xr[i];
      s += v;
      ss += v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  __shared__ float sm[8];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[wave] = s;
    sm[4 + wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)row * chunks + chunk) * 2] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    part[((size_t)row * chunks + chunk) * 2 + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void cgg_gn_apply_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ part,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ y, int C, int groups,
                                                           long long hw, int chunks, float eps, int relu) {
  // blockIdx.y = (b, channel), blockIdx.x = 1024-element span of that plane
  const int bc = blockIdx.y;
  const int c = bc % C, b = bc / C;
  const int cpg = C / groups;
  const int row = b * groups + c / cpg;
  double s = 0.0, ss = 0.0;
  for (int i = 0; i < chunks; ++i) {
    s += (double)part[((size_t)row * chunks + i) * 2];
    ss += (double)part[((size_t)row * chunks + i) * 2 + 1];
  }
  const double n = (double)cpg * (double)hw;
  const double mean = s / n;
  const double var = fmax(ss / n - mean * mean, 0.0);
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = gamma[c] * rstd;
  const float sh = beta[c] - (float)mean * a;
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i < hw) {
    const size_t off = (size_t)bc * hw + i;
    if (VEC == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r = v[e] * a + sh;
        v[e] = relu ? fmaxf(r, 0.f) : r;
      }
      *reinterpret_cast<f32x4*>(y + off) = v;
    } else {
      const float r = x[off] * a + sh;
      y[off] = relu ? fmaxf(r, 0.f) : r;
    }
  }
}

extern "C" int64_t cgg_group_norm_workspace_bytes(int B, int C, int H, int W, int groups) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups) return 0;
  const long long row_len = (long long)(C / groups) * H * W;
  const int chunks = (int)((row_len + GN_CHUNK - 1) / GN_CHUNK);
  return (int64_t)B * groups * chunks * 2 * (int64_t)sizeof(float);
}

extern "C" int cgg_group_norm(const float* x, const float* gamma, const float* beta, float* y, void* ws, int B,
                              int C, int H, int W, int groups, float eps, int relu, cgg_stream_t stream) {
  CGG_REQUIRE(x && gamma && beta && y && ws, CGG_EINVAL, "cgg_group_norm: null pointer");
  CGG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && groups > 0, CGG_EINVAL, "cgg_group_norm: bad sizes");
  CGG_REQUIRE(C % groups == 0, CGG_EINVAL, "cgg_group_norm: C=%d not divisible by groups=%d", C, groups);
  const long long hw = (long long)H * W;
  const bool vec = (hw % 4 == 0) && cgg_aligned16(x) && cgg_aligned16(y);
  const long long row_len = (long long)(C / groups) * hw;
  const int chunks = (int)((row_len + GN_CHUNK - 1) / GN_CHUNK);
  hipStream_t s = (hipStream_t)stream;
  if (vec) {
    hipLaunchKernelGGL(cgg_gn_partial_kernel<4>, dim3(chunks, B * groups), dim3(256), 0, s, x, (float*)ws,
                       row_len, chunks);
    hipLaunchKernelGGL(cgg_gn_apply_kernel<4>, dim3((unsigned)((hw / 4 + 255) / 256), B * C), dim3(256), 0, s,
                       x, (const float*)ws, gamma, beta, y, C, groups, hw, chunks, eps, relu);
  } else {
    hipLaunchKernelGGL(cgg_gn_partial_kernel<1>, dim3(chunks, B * groups), dim3(256), 0, s, x, (float*)ws,
                       row_len, chunks);
    hipLaunchKernelGGL(cgg_gn_apply_kernel<1>, dim3((unsigned)((hw + 255) / 256), B * C), dim3(256), 0, s, x,
                       (const float*)ws, gamma, beta, y, C, groups, hw, chunks, eps, relu);
  }
  CGG_CHECK_LAUNCH("cgg_group_norm");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// Channel-last (NHWC) bf16 GroupNorm for the throughput-mode pixel decoder: x [B, HW, C] is the bf16 output of
// a 1x1-conv GEMM / MIOpen NHWC conv; C / groups == 8, so ONE 16-byte vector is exactly one group of one pixel.
//   stats : per (b, g) sum and sum of squares (f32 partials per thread, LDS tree, one atomic pair per block)
//   apply : y = (x - mean) * rstd * gamma + beta [+ bilinear x2 up-sample of a low-res f32 NHWC map] [ReLU], with
//           up to three outputs from the one pass:
//             y32 f32, rows at b * y32_bstride (lets the three encoder levels land directly in the (B, N, 256)
//                 residual stream `src` -- no flatten / transpose / cat),
//             y16 = bf16(y) dense, yp16 = bf16(y + pos[p]) dense (the first encoder layer's GEMM inputs).
// -------------------------------------------------------------------------------------------------
#define GNH_PIX 64   // pixels per stats block (1024 blocks per 256x256 image; partials are reduced by a second kernel)

// the 8 channels of (pixel-group) vector i: one 16-byte bf16 vector, or two of f32 (XF32: parity mode's f32 channel-last maps)
template <bool XF32>
__device__ __forceinline__ void gnh_load(const uint4* __restrict__ x, size_t i, float* f) {
  if constexpr (XF32) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[k] = a[k];
      f[k + 4] = b[k];
    }
  } else {
    const uint4 v = x[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __uint_as_float(w[k] << 16);
      f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
  }
}

template <bool XF32>
__global__ __launch_bounds__(256) void cgg_gn_nhwc_stats_kernel(const uint4* __restrict__ x, float* __restrict__ ws,
                                                               int HW, int G, int ppb, int B_G2_floats) {
  // thread t: group g = t % G (G <= 256 and 256 % G == 0), pixel lane = t / G
  __shared__ float red[2][256];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int g = tid % G, pl = tid / G, PL = 256 / G;
  const int p0 = blockIdx.x * ppb;
  const int p1 = min(p0 + ppb, HW);
  float s = 0.f, q = 0.f;
  // f32 maps (parity mode): sums of (x - pivot), pivot = the group's first value of the image, the same for every block -- the
  // one-pass E[x^2] - mean^2 of the bf16 path cancels catastrophically when |mean| >> std, which torch's f32 GroupNorm
  // (Welford / two-pass) does not
  const float pv = XF32 ? reinterpret_cast<const float*>(x)[((size_t)b * HW * G + g) * 8] : 0.f;
  for (int p = p0 + pl; p < p1; p += PL) {
    float f[8];
    gnh_load<XF32>(x, ((size_t)b * HW + p) * G + g, f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = f[2 * k] - pv, hi = f[2 * k + 1] - pv;
      s += lo + hi;
      q += lo * lo + hi * hi;
    }
  }
  red[0][tid] = s;
  red[1][tid] = q;
  __syncthreads();
  if (tid < G) {
    for (int k = 1; k < PL; ++k) {
      s += red[0][tid + k * G];
      q += red[1][tid + k * G];
    }
    // per-block partials (no atomics: 512 blocks hammering 64 addresses made this pass 3x slower than the read)
    float* part = ws + (size_t)B_G2_floats + (((size_t)b * gridDim.x + blockIdx.x) * G + g) * 2;
    part[0] = s;
    part[1] = q;
  }
}

// second level: one 64-lane wave per (b, g) sums the per-block partials -> ws[b][g][2]
__global__ __launch_bounds__(64) void cgg_gn_nhwc_reduce_kernel(float* __restrict__ ws, int B_G2_floats, int nblk, int G,
                                                               int B, const float* __restrict__ x32, int HW, float inv_n) {
  const int i = blockIdx.x;                  // (b, g)
  const int b = i / G, g = i - b * G;
  const float* part = ws + B_G2_floats + ((size_t)b * nblk * G + g) * 2;
  float s = 0.f, q = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 64) {
    const float2 v = *reinterpret_cast<const float2*>(part + (size_t)k * G * 2);
    s += v.x;
    q += v.y;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (threadIdx.x == 0) {
    // -> (mean, variance). x32 != NULL: the sums are of (x - pivot) (see the stats kernel); the pivot is read HERE, before the
    // apply pass -- which may run in place -- overwrites it
    const float pv = x32 ? x32[((size_t)b * HW * G + g) * 8] : 0.f;
    const float dm = s * inv_n;
    ws[(size_t)i * 2] = pv + dm;
    ws[(size_t)i * 2 + 1] = fmaxf(q * inv_n - dm * dm, 0.f);
  }
}

__device__ __forceinline__ void gnh_unpack(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 gnh_pack(const float* f) {
  return make_uint4(cgg_pack2(cgg_f2bf(f[0]), cgg_f2bf(f[1])), cgg_pack2(cgg_f2bf(f[2]), cgg_f2bf(f[3])),
                    cgg_pack2(cgg_f2bf(f[4]), cgg_f2bf(f[5])), cgg_pack2(cgg_f2bf(f[6]), cgg_f2bf(f[7])));
}

template <bool UPADD, bool XF32>
__global__ __launch_bounds__(256) void cgg_gn_nhwc_apply_kernel(
    const uint4* __restrict__ x, const float* __restrict__ ws, const float* __restrict__ gamma,
    const float* __restrict__ beta, int HW, int G, float inv_n, float eps, int relu,
    const float* __restrict__ lo, int lo_h, int lo_w, long long lo_bstride, int W,
    float* __restrict__ y32, long long y32_bstride, uint4* __restrict__ y16, const float* __restrict__ pos,
    uint4* __restrict__ yp16, long long y16_bstride, float* __restrict__ yp32, int x3a, int* __restrict__ flag, int out_padw) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // vector index inside the image
  if (i >= (long long)HW * G) return;
  const int p = (int)(i / G), g = (int)(i - (long long)p * G);
  const float s = ws[((size_t)b * G + g) * 2], q = ws[((size_t)b * G + g) * 2 + 1];
  const float mean = s;                                   // the reduce kernel leaves (mean, variance) in ws
  const float rstd = rsqrtf(q + eps);
  float f[8];
  gnh_load<XF32>(x, (size_t)b * HW * G + i, f);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g * 8), gb = *reinterpret_cast<const f32x4*>(gamma + g * 8 + 4);
  const f32x4 ba = *reinterpret_cast<const f32x4*>(beta + g * 8), bb = *reinterpret_cast<const f32x4*>(beta + g * 8 + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[k] = (f[k] - mean) * rstd * ga[k] + ba[k];
    f[k + 4] = (f[k + 4] - mean) * rstd * gb[k] + bb[k];
  }
  if (UPADD) {
    // F.interpolate(lo, (2*lo_h.., ..), 'bilinear', align_corners=False) at (py, px); PyTorch's source index rule
    const int H = HW / W;
    const int py = p / W, px = p - py * W;
    const float sy = fmaxf(((float)py + 0.5f) * ((float)lo_h / (float)H) - 0.5f, 0.f);
    const float sx = fmaxf(((float)px + 0.5f) * ((float)lo_w / (float)W) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, lo_h - 1), x1 = min(x0 + 1, lo_w - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* base = lo + (size_t)b * lo_bstride + g * 8;
    const size_t C = (size_t)G * 8;
    const float* r00 = base + ((size_t)y0 * lo_w + x0) * C;
    const float* r01 = base + ((size_t)y0 * lo_w + x1) * C;
    const float* r10 = base + ((size_t)y1 * lo_w + x0) * C;
    const float* r11 = base + ((size_t)y1 * lo_w + x1) * C;
    if (x3a & 2) {
      // the low-resolution map is stored as x3a rows (csrc/x3.h): a group of 8 channels = [8 hi | 8 lo]
      float a[8], c[8], d[8], e[8];
      const cgg_u32x4* q00 = reinterpret_cast<const cgg_u32x4*>(r00);
      const cgg_u32x4* q01 = reinterpret_cast<const cgg_u32x4*>(r01);
      const cgg_u32x4* q10 = reinterpret_cast<const cgg_u32x4*>(r10);
      const cgg_u32x4* q11 = reinterpret_cast<const cgg_u32x4*>(r11);
      cgg_x3a_decode8(q00[0], q00[1], a);
      cgg_x3a_decode8(q01[0], q01[1], c);
      cgg_x3a_decode8(q10[0], q10[1], d);
      cgg_x3a_decode8(q11[0], q11[1], e);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] += (1.f - ly) * ((1.f - lx) * a[k] + lx * c[k]) + ly * ((1.f - lx) * d[k] + lx * e[k]);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(r00 + 4 * h), c = *reinterpret_cast<const f32x4*>(r01 + 4 * h);
        const f32x4 d = *reinterpret_cast<const f32x4*>(r10 + 4 * h), e = *reinterpret_cast<const f32x4*>(r11 + 4 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          f[4 * h + k] += (1.f - ly) * ((1.f - lx) * a[k] + lx * c[k]) + ly * ((1.f - lx) * d[k] + lx * e[k]);
      }
    }
  }
  if (relu) {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], 0.f);
  }
  if (y32) {
    // out_padw > 0: y32 is the interior of a zero-bordered (H + 2) x (out_padw + 2) channel-last map (the x3 training convolution's input)
    size_t ov = (size_t)i;
    if (out_padw > 0) {
      const int oy = p / out_padw, ox = p - oy * out_padw;
      ov = ((size_t)(oy + 1) * (size_t)(out_padw + 2) + (size_t)(ox + 1)) * (size_t)G + (size_t)g;
    }
    float* o = y32 + (size_t)b * y32_bstride + ov * 8;
    if (x3a & 1) {             // x3a rows: the group's hi and lo pieces
      cgg_u32x4 h, l;
      cgg_x3a_encode8(f, h, l);
      reinterpret_cast<cgg_u32x4*>(o)[0] = h;
      reinterpret_cast<cgg_u32x4*>(o)[1] = l;
      float am = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) am = fmaxf(am, fabsf(f[k]));
      if (flag && !(am * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(flag, 1);
    } else {
      *reinterpret_cast<f32x4*>(o) = f32x4{f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
  }
  if (yp32) {                  // y + pos (per-token table) as x3a rows, same layout as y32
    const f32x4 pa = *reinterpret_cast<const f32x4*>(pos + (size_t)i * 8), pb = *reinterpret_cast<const f32x4*>(pos + (size_t)i * 8 + 4);
    float g2[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g2[k] = f[k] + pa[k];
      g2[k + 4] = f[k + 4] + pb[k];
    }
    cgg_u32x4 h, l;
    cgg_x3a_encode8(g2, h, l);
    cgg_u32x4* o = reinterpret_cast<cgg_u32x4*>(yp32 + (size_t)b * y32_bstride + (size_t)i * 8);
    o[0] = h;
    o[1] = l;
  }
  if (y16) y16[(size_t)b * y16_bstride + i] = gnh_pack(f);
  if (yp16) {
    const f32x4 pa = *reinterpret_cast<const f32x4*>(pos + (size_t)i * 8), pb = *reinterpret_cast<const f32x4*>(pos + (size_t)i * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[k] += pa[k];
      f[k + 4] += pb[k];
    }
    yp16[(size_t)b * y16_bstride + i] = gnh_pack(f);
  }
}

extern "C" int64_t cgg_group_norm_nhwc_workspace_bytes(int B, int HW, int groups) {
  if (B <= 0 || HW <= 0 || groups <= 0) return 0;
  const int64_t nblk = (HW + GNH_PIX - 1) / GNH_PIX;
  return ((int64_t)B * groups * 2 + (int64_t)B * nblk * groups * 2) * (int64_t)sizeof(float);
}

int* cgg_x3_overflow_flag_ptr();       // x3s_gemm.hip

static int gnh_launch(bool xf32, const void* x, const float* gamma, const float* beta, void* ws, int B, int HW,
                      int C, int groups, float eps, int relu, const float* up_src, int up_h, int up_w,
                      int64_t up_bstride, int W, float* y32, int64_t y32_bstride, void* y16,
                      const float* pos, void* yp16, int64_t y16_bstride, cgg_stream_t stream, float* yp32 = nullptr,
                      int x3a = 0, int out_padw = 0) {
  CGG_REQUIRE(x && gamma && beta && ws, CGG_EINVAL, "cgg_group_norm_nhwc: null pointer");
  CGG_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0, CGG_EINVAL, "cgg_group_norm_nhwc: bad sizes");
  CGG_REQUIRE(C == groups * 8 && groups <= 256 && 256 % groups == 0, CGG_EUNSUPPORTED,
              "cgg_group_norm_nhwc: needs C / groups == 8 and groups | 256 (C=%d, groups=%d)", C, groups);
  CGG_REQUIRE(y32 || y16 || yp16, CGG_EINVAL, "cgg_group_norm_nhwc: no output requested");
  CGG_REQUIRE((!yp16 && !yp32) || pos, CGG_EINVAL, "cgg_group_norm_nhwc: yp16 / yp32 need pos");
  CGG_REQUIRE(cgg_aligned16(yp32), CGG_EALIGN, "cgg_group_norm_nhwc: yp32 must be 16-byte aligned");
  CGG_REQUIRE(!up_src || (W > 0 && HW % W == 0 && up_h > 0 && up_w > 0), CGG_EINVAL,
              "cgg_group_norm_nhwc: bad up-sample geometry");
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(gamma) && cgg_aligned16(beta) && cgg_aligned16(up_src) &&
                  cgg_aligned16(y32) && cgg_aligned16(y16) && cgg_aligned16(pos) && cgg_aligned16(yp16) &&
                  y32_bstride % 4 == 0 && up_bstride % 4 == 0 && y16_bstride % 8 == 0,
              CGG_EALIGN, "cgg_group_norm_nhwc: pointers / strides must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (HW + GNH_PIX - 1) / GNH_PIX;
  const int head = B * groups * 2;        // ws = [B][G][2] totals, then [B][nblk][G][2] per-block partials
  if (xf32)
    hipLaunchKernelGGL(cgg_gn_nhwc_stats_kernel<true>, dim3(nblk, B), dim3(256), 0, s, (const uint4*)x, (float*)ws, HW, groups,
                       GNH_PIX, head);
  else
    hipLaunchKernelGGL(cgg_gn_nhwc_stats_kernel<false>, dim3(nblk, B), dim3(256), 0, s, (const uint4*)x, (float*)ws, HW, groups,
                       GNH_PIX, head);
  const float inv_n = 1.f / ((float)HW * 8.f);
  hipLaunchKernelGGL(cgg_gn_nhwc_reduce_kernel, dim3(B * groups), dim3(64), 0, s, (float*)ws, head, nblk, groups, B,
                     xf32 ? (const float*)x : nullptr, HW, inv_n);
  const long long nvec = (long long)HW * groups;
  const dim3 grid((unsigned)((nvec + 255) / 256), B);
#define GNH_APPLY(UP, XF)                                                                                                   \
  hipLaunchKernelGGL((cgg_gn_nhwc_apply_kernel<UP, XF>), grid, dim3(256), 0, s, (const uint4*)x, (const float*)ws, gamma, beta,  \
                     HW, groups, inv_n, eps, relu, up_src, up_h, up_w, (long long)up_bstride, W, y32, (long long)y32_bstride,  \
                     (uint4*)y16, pos, (uint4*)yp16, (long long)(y16_bstride / 8), yp32, x3a,                                \
                     (x3a & 1) ? cgg_x3_overflow_flag_ptr() : nullptr, out_padw)
  if (up_src && xf32) GNH_APPLY(true, true);
  else if (up_src) GNH_APPLY(true, false);
  else if (xf32) GNH_APPLY(false, true);
  else GNH_APPLY(false, false);
#undef GNH_APPLY
  CGG_CHECK_LAUNCH("cgg_group_norm_nhwc");
  return CGG_OK;
}

extern "C" int cgg_group_norm_nhwc(const void* x, const float* gamma, const float* beta, void* ws, int B, int HW,
                                   int C, int groups, float eps, int relu, const float* up_src, int up_h, int up_w,
                                   int64_t up_bstride, int W, float* y32, int64_t y32_bstride, void* y16,
                                   const float* pos, void* yp16, int64_t y16_bstride, cgg_stream_t stream) {
  return gnh_launch(false, x, gamma, beta, ws, B, HW, C, groups, eps, relu, up_src, up_h, up_w, up_bstride, W, y32, y32_bstride,
                    y16, pos, yp16, y16_bstride, stream);
}

extern "C" int cgg_group_norm_nhwc_f32(const float* x, const float* gamma, const float* beta, void* ws, int B, int HW,
                                       int C, int groups, float eps, int relu, const float* up_src, int up_h, int up_w,
                                       int64_t up_bstride, int W, float* y32, int64_t y32_bstride, void* y16,
                                       const float* pos, void* yp16, int64_t y16_bstride, cgg_stream_t stream) {
  return gnh_launch(true, x, gamma, beta, ws, B, HW, C, groups, eps, relu, up_src, up_h, up_w, up_bstride, W, y32, y32_bstride,
                    y16, pos, yp16, y16_bstride, stream);
}

// ... with y written into the INTERIOR of a (B, H + 2, W + 2, C) channel-last map (border untouched: the caller zeroes it): the input
// of the x3 training convolution behind the [3P] FPN lateral GroupNorm, without a padding copy (W = the map's width, HW % W == 0).
extern "C" int cgg_group_norm_nhwc_f32_padout(const float* x, const float* gamma, const float* beta, void* ws, int B, int HW, int C,
                                              int groups, float eps, int relu, const float* up_src, int up_h, int up_w, int64_t up_bstride,
                                              int W, float* y_padded, cgg_stream_t stream) {
  CGG_REQUIRE(y_padded && W > 0 && HW % W == 0, CGG_EINVAL, "cgg_group_norm_nhwc_f32_padout: null output / bad W");
  return gnh_launch(true, x, gamma, beta, ws, B, HW, C, groups, eps, relu, up_src, up_h, up_w, up_bstride, W, y_padded,
                    (int64_t)(HW / W + 2) * (W + 2) * C, nullptr, nullptr, nullptr, 0, stream, nullptr, 0, W);
}

// Parity mode's stream (round 4): GroupNorm over an f32 map with the output(s) written as x3a rows (csrc/x3.h) -- the A operand
// of the next x3 GEMM / the encoder's residual stream -- y = GN(x) (+ up-sample(up_src), ReLU) and, optionally, yp = y + pos (the
// first encoder layer's `query + query_pos` rows). up_src is itself an x3a map (the encoder's finest memory level).
extern "C" int cgg_group_norm_nhwc_f32_x3a(const float* x, const float* gamma, const float* beta, void* ws, int B, int HW, int C,
                                           int groups, float eps, int relu, const void* up_src_x3a, int up_h, int up_w,
                                           int64_t up_bstride, int W, void* y_x3a, int64_t y_bstride, const float* pos,
                                           void* yp_x3a, cgg_stream_t stream) {
  CGG_REQUIRE(y_x3a, CGG_EINVAL, "cgg_group_norm_nhwc_f32_x3a: null output");
  return gnh_launch(true, x, gamma, beta, ws, B, HW, C, groups, eps, relu, (const float*)up_src_x3a, up_h, up_w, up_bstride, W,
                    (float*)y_x3a, y_bstride, nullptr, pos, nullptr, 0, stream, (float*)yp_x3a, up_src_x3a ? 3 : 1);
}

// -------------------------------------------------------------------------------------------------
// Backward of the channel-last f32 GroupNorm above (training, parity mode: the FPN level of the pixel decoder kept channel-last --
// [3P] MSDeformAttnPixelDecoder lateral / output ConvModules behind open_set/models/mask2former_head.py:787; torch's GroupNorm
// breaks a channel-last chain: its channel-last path is 2.2 x slower and returns NCHW):
//     y = act(xhat gamma + beta [+ up(lo)]),  xhat = (x - mean_bg) rstd_bg,  g = dy (where y > 0 if act = ReLU)
//     dgamma_c = sum g xhat,  dbeta_c = sum g,  per (b, group): S1 = sum g gamma, S2 = sum g gamma xhat  (n = HW * 8 elements)
//     dx = rstd (g gamma - S1 / n - xhat S2 / n),   d lo = the bilinear up-sampling's transpose applied to dy
// Pass 1 accumulates, per thread, (g xhat, g) for the 8 channels of its group over its pixels -- S1 / S2 follow from them with gamma --
// and leaves per-block partials; pass 2 sums them per (b, group); pass 3 applies. C / groups == 8 like the forward.
// -------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(256) void cgg_gn_nhwc_bwd_partials_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                      const float* __restrict__ dy, const float* __restrict__ stats,
                                                                      float eps, float* __restrict__ part, int HW, int G, int ppb) {
  __shared__ float red[16][256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int g = tid % G, pl = tid / G, PL = 256 / G;
  const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, HW);
  const float mean = stats[((size_t)b * G + g) * 2], rstd = rsqrtf(stats[((size_t)b * G + g) * 2 + 1] + eps);
  float dg[8], db[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) dg[k] = db[k] = 0.f;
  for (int p = p0 + pl; p < p1; p += PL) {
    const size_t o = (((size_t)b * HW + p) * G + g) * 8;
    const f32x4 xa = *reinterpret_cast<const f32x4*>(x + o), xb = *reinterpret_cast<const f32x4*>(x + o + 4);
    f32x4 ga = *reinterpret_cast<const f32x4*>(dy + o), gb = *reinterpret_cast<const f32x4*>(dy + o + 4);
    if constexpr (RELU) {
      const f32x4 ya = *reinterpret_cast<const f32x4*>(y + o), yb = *reinterpret_cast<const f32x4*>(y + o + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ga[k] = ya[k] > 0.f ? ga[k] : 0.f;
        gb[k] = yb[k] > 0.f ? gb[k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dg[k] = fmaf(ga[k], (xa[k] - mean) * rstd, dg[k]);
      dg[k + 4] = fmaf(gb[k], (xb[k] - mean) * rstd, dg[k + 4]);
      db[k] += ga[k];
      db[k + 4] += gb[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[k][tid] = dg[k];
    red[8 + k][tid] = db[k];
  }
  __syncthreads();
  // 16 values x G groups per block: thread (v = tid / G... ) sums the PL pixel lanes of one (value, group)
  for (int i = tid; i < 16 * G; i += 256) {
    const int v = i / G, gg = i - v * G;
    float s = 0.f;
    for (int k = 0; k < PL; ++k) s += red[v][gg + k * G];
    part[(((size_t)b * gridDim.x + blockIdx.x) * G + gg) * 16 + v] = s;
  }
}

// one wave per (b, group): block partials -> tot[b][g][16] = (dgamma part x 8 | dbeta part x 8) and s12[b][g] = (S1 / n, S2 / n)
__global__ __launch_bounds__(64) void cgg_gn_nhwc_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                                   float* __restrict__ tot, float* __restrict__ s12, int nblk, int G,
                                                                   float inv_n) {
  const int i = blockIdx.x, b = i / G, g = i - b * G, lane = threadIdx.x;
  float acc[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  for (int k = lane; k < nblk; k += 64) {
    const float* p = part + (((size_t)b * nblk + k) * G + g) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * q + e] += t[e];
    }
  }
#pragma unroll
  for (int v = 0; v < 16; ++v)
    for (int o = 32; o > 0; o >>= 1) acc[v] += __shfl_xor(acc[v], o);
  if (lane == 0) {
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float gm = gamma[g * 8 + k];
      S2 = fmaf(acc[k], gm, S2);
      S1 = fmaf(acc[8 + k], gm, S1);
      tot[(size_t)i * 16 + k] = acc[k];
      tot[(size_t)i * 16 + 8 + k] = acc[8 + k];
    }
    s12[(size_t)i * 2] = S1 * inv_n;
    s12[(size_t)i * 2 + 1] = S2 * inv_n;
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void cgg_gn_nhwc_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ dy, const float* __restrict__ stats,
                                                                   const float* __restrict__ s12, const float* __restrict__ gamma,
                                                                   float eps, float* __restrict__ dx, int HW, int G, int out_padw) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // vector index inside the image
  if (i >= (long long)HW * G) return;
  const int g = (int)(i % G);
  // out_padw > 0: dx goes to the interior of a zero-bordered (H + 2) x (out_padw + 2) map (grad_output of the convolution in front)
  size_t od = ((size_t)b * HW * G + (size_t)i) * 8;
  if (out_padw > 0) {
    const int p = (int)(i / G), oy = p / out_padw, ox = p - oy * out_padw;
    od = (((size_t)b * (size_t)(HW / out_padw + 2) + (size_t)(oy + 1)) * (size_t)(out_padw + 2) + (size_t)(ox + 1)) * (size_t)G * 8 + (size_t)g * 8;
  }
  const float mean = stats[((size_t)b * G + g) * 2], rstd = rsqrtf(stats[((size_t)b * G + g) * 2 + 1] + eps);
  const float c1 = s12[((size_t)b * G + g) * 2], c2 = s12[((size_t)b * G + g) * 2 + 1];
  const size_t o = ((size_t)b * HW * G + (size_t)i) * 8;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o + 4 * h);
    f32x4 gv = *reinterpret_cast<const f32x4*>(dy + o + 4 * h);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + g * 8 + 4 * h);
    if constexpr (RELU) {
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    }
    f32x4 d;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = rstd * (gv[k] * gm[k] - c1 - (xv[k] - mean) * rstd * c2);
    *reinterpret_cast<f32x4*>(dx + od + 4 * h) = d;
  }
}

// d lo[b, Y, X, :] = sum over the high-resolution pixels whose bilinear taps (the forward's index rule) include (Y, X)
__global__ __launch_bounds__(256) void cgg_upsample_bilinear_nhwc_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dlo, int H,
                                                                            int W, int lo_h, int lo_w, int G) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)lo_h * lo_w * G) return;
  const int g = (int)(i % G), pix = (int)(i / G);
  const int Y = pix / lo_w, X = pix - Y * lo_w;
  const float sh = (float)H / (float)lo_h, sw = (float)W / (float)lo_w;
  const int py0 = max(0, (int)floorf(((float)Y - 0.5f) * sh - 0.5f)), py1 = min(H - 1, (int)ceilf(((float)Y + 1.5f) * sh - 0.5f));
  const int px0 = max(0, (int)floorf(((float)X - 0.5f) * sw - 0.5f)), px1 = min(W - 1, (int)ceilf(((float)X + 1.5f) * sw - 0.5f));
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int py = py0; py <= py1; ++py) {
    const float sy = fmaxf(((float)py + 0.5f) * ((float)lo_h / (float)H) - 0.5f, 0.f);
    const int y0 = (int)sy, y1 = min(y0 + 1, lo_h - 1);
    const float ly = sy - (float)y0;
    const float wy = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int px = px0; px <= px1; ++px) {
      const float sx = fmaxf(((float)px + 0.5f) * ((float)lo_w / (float)W) - 0.5f, 0.f);
      const int x0 = (int)sx, x1 = min(x0 + 1, lo_w - 1);
      const float lx = sx - (float)x0;
      const float wx = (x0 == X ? 1.f - lx : 0.f) + (x1 == X ? lx : 0.f);
      if (wx == 0.f) continue;
      const float w = wy * wx;
      const float* s = dy + ((((size_t)b * H + py) * W + px) * G + g) * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(s), c = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = fmaf(w, a[k], acc[k]);
        acc[k + 4] = fmaf(w, c[k], acc[k + 4]);
      }
    }
  }
  float* d = dlo + ((size_t)b * lo_h * lo_w * G + (size_t)i) * 8;
  *reinterpret_cast<f32x4*>(d) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  *reinterpret_cast<f32x4*>(d + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

extern "C" int64_t cgg_group_norm_nhwc_backward_workspace_bytes(int B, int HW, int groups) {
  if (B <= 0 || HW <= 0 || groups <= 0) return 0;
  const int64_t nblk = (HW + GNH_PIX - 1) / GNH_PIX;
  return ((int64_t)B * nblk * groups * 16 + (int64_t)B * groups * 2) * (int64_t)sizeof(float);
}

// x, dy (and y, the forward's output, when relu) (B, HW, C) f32 channel-last dense; stats = the forward's workspace head
// (B, groups, 2) = (mean, variance); ws >= cgg_group_norm_nhwc_backward_workspace_bytes; -> dx (B, HW, C), tot (B, groups, 16) =
// per-image (dgamma | dbeta) pieces (the caller sums over the batch: dgamma[8 g + k] = sum_b tot[b][g][k], dbeta = ...[8 + k]);
// dlo (nullable): the gradient of the up-sampled low-resolution map (B, lo_h, lo_w, C) the forward added (no ReLU in that form).
// dx_padded: dx is the interior of a (B, H + 2, W + 2, C) map whose border the caller zeroed (grad_output of the convolution in front).
extern "C" int cgg_group_norm_nhwc_f32_backward(const float* x, const float* y, const float* dy, const float* stats, const float* gamma,
                                                void* ws, int B, int HW, int C, int groups, float eps, int relu, float* dx, float* tot,
                                                float* dlo, int lo_h, int lo_w, int W, int dx_padded, cgg_stream_t stream) {
  CGG_REQUIRE(x && dy && stats && gamma && ws && dx && tot, CGG_EINVAL, "cgg_group_norm_nhwc_f32_backward: null pointer");
  CGG_REQUIRE(!dx_padded || (W > 0 && HW % W == 0), CGG_EINVAL, "cgg_group_norm_nhwc_f32_backward: dx_padded needs the map width W");
  CGG_REQUIRE(!relu || y, CGG_EINVAL, "cgg_group_norm_nhwc_f32_backward: relu needs the forward's output y");
  CGG_REQUIRE(B > 0 && HW > 0 && C == groups * 8 && groups <= 256 && 256 % groups == 0, CGG_EUNSUPPORTED,
              "cgg_group_norm_nhwc_f32_backward: needs C / groups == 8 and groups | 256 (C=%d, groups=%d)", C, groups);
  CGG_REQUIRE(!dlo || (!relu && W > 0 && HW % W == 0 && lo_h > 0 && lo_w > 0), CGG_EINVAL,
              "cgg_group_norm_nhwc_f32_backward: bad up-sample geometry (and the up-sample form has no ReLU)");
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(y) && cgg_aligned16(dy) && cgg_aligned16(gamma) && cgg_aligned16(ws) && cgg_aligned16(dx) &&
                  cgg_aligned16(tot) && cgg_aligned16(dlo), CGG_EALIGN, "cgg_group_norm_nhwc_f32_backward: 16-B alignment");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (HW + GNH_PIX - 1) / GNH_PIX;
  float* part = (float*)ws;
  float* s12 = part + (size_t)B * nblk * groups * 16;
  const float inv_n = 1.f / ((float)HW * 8.f);
  if (relu)
    hipLaunchKernelGGL(cgg_gn_nhwc_bwd_partials_kernel<true>, dim3(nblk, B), dim3(256), 0, s, x, y, dy, stats, eps, part, HW, groups, GNH_PIX);
  else
    hipLaunchKernelGGL(cgg_gn_nhwc_bwd_partials_kernel<false>, dim3(nblk, B), dim3(256), 0, s, x, y, dy, stats, eps, part, HW, groups, GNH_PIX);
  hipLaunchKernelGGL(cgg_gn_nhwc_bwd_reduce_kernel, dim3(B * groups), dim3(64), 0, s, (const float*)part, gamma, tot, s12, nblk, groups, inv_n);
  const long long nvec = (long long)HW * groups;
  const dim3 grid((unsigned)((nvec + 255) / 256), B);
  if (relu)
    hipLaunchKernelGGL(cgg_gn_nhwc_bwd_apply_kernel<true>, grid, dim3(256), 0, s, x, y, dy, stats, (const float*)s12, gamma, eps, dx, HW, groups, dx_padded ? W : 0);
  else
    hipLaunchKernelGGL(cgg_gn_nhwc_bwd_apply_kernel<false>, grid, dim3(256), 0, s, x, y, dy, stats, (const float*)s12, gamma, eps, dx, HW, groups,
                       dx_padded ? W : 0);
  if (dlo) {
    const long long nlo = (long long)lo_h * lo_w * groups;
    hipLaunchKernelGGL(cgg_upsample_bilinear_nhwc_bwd_kernel, dim3((unsigned)((nlo + 255) / 256), B), dim3(256), 0, s, dy, dlo, HW / W, W,
                       lo_h, lo_w, groups);
  }
  CGG_CHECK_LAUNCH("cgg_group_norm_nhwc_f32_backward");
  return CGG_OK;
}
