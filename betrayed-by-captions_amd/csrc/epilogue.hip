// Channel-last bf16 epilogue for the BN-folded backbone convolutions (the [3P] mmdet ResNet
// Bottleneck tail `relu(bn(conv(x)) + identity)`): ONE in-place pass
//     y[r, c] = act(y[r, c] + bias[c] + res[r, c])
// instead of the three library passes (bias add, residual add, clamp). Pure HBM streaming: every lane owns one
// 16-byte vector (8 channels), loads are issued before any use, the bias vector comes from L1/L2.
#include "x3.h"

template <bool BIAS, bool RES, bool RELU>
__global__ __launch_bounds__(256) void cgg_bias_act_kernel(uint4* __restrict__ y, const uint4* __restrict__ bias,
                                                          const uint4* __restrict__ res, long long nvec, int c8) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    uint4 v = y[i];
    uint4 r = RES ? res[i] : make_uint4(0, 0, 0, 0);
    uint4 b = BIAS ? bias[(int)(i % c8)] : make_uint4(0, 0, 0, 0);
    uint32_t vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // mirror the library sequence: each intermediate is rounded to bf16 (bias add, then residual add)
      float lo = cgg_bf2f((uint16_t)(vv[k] & 0xffffu)), hi = cgg_bf2f((uint16_t)(vv[k] >> 16));
      if (BIAS) {
        lo = cgg_bf2f(cgg_f2bf(lo + cgg_bf2f((uint16_t)(bb[k] & 0xffffu))));
        hi = cgg_bf2f(cgg_f2bf(hi + cgg_bf2f((uint16_t)(bb[k] >> 16))));
      }
      if (RES) {
        lo += cgg_bf2f((uint16_t)(rr[k] & 0xffffu));
        hi += cgg_bf2f((uint16_t)(rr[k] >> 16));
      }
      if (RELU) {
        lo = fmaxf(lo, 0.f);
        hi = fmaxf(hi, 0.f);
      }
      vv[k] = cgg_pack2(cgg_f2bf(lo), cgg_f2bf(hi));
    }
    y[i] = make_uint4(vv[0], vv[1], vv[2], vv[3]);
  }
}

extern "C" int cgg_bias_act_nhwc(void* y, const void* bias, const void* res, int64_t rows, int C, int relu,
                                 cgg_stream_t stream) {
  CGG_REQUIRE(y != nullptr && rows >= 0 && C > 0, CGG_EINVAL, "cgg_bias_act_nhwc: bad arguments");
  CGG_REQUIRE(C % 8 == 0, CGG_EUNSUPPORTED, "cgg_bias_act_nhwc: C %% 8 != 0 (C=%d)", C);
  CGG_REQUIRE(cgg_aligned16(y) && cgg_aligned16(bias) && cgg_aligned16(res), CGG_EALIGN,
              "cgg_bias_act_nhwc: pointers must be 16-byte aligned");
  if (rows == 0) return 0;
  const long long nvec = (long long)rows * (C / 8);
  const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  uint4* yy = (uint4*)y;
  const uint4* bb = (const uint4*)bias;
  const uint4* rr = (const uint4*)res;
#define CGG_BA(B_, R_, A_) \
  hipLaunchKernelGGL((cgg_bias_act_kernel<B_, R_, A_>), dim3(blocks), dim3(256), 0, s, yy, bb, rr, nvec, C / 8)
  const int sel = (bias ? 4 : 0) | (res ? 2 : 0) | (relu ? 1 : 0);
  switch (sel) {
    case 0: return 0;
    case 1: CGG_BA(false, false, true); break;
    case 2: CGG_BA(false, true, false); break;
    case 3: CGG_BA(false, true, true); break;
    case 4: CGG_BA(true, false, false); break;
    case 5: CGG_BA(true, false, true); break;
    case 6: CGG_BA(true, true, false); break;
    default: CGG_BA(true, true, true); break;
  }
#undef CGG_BA
  CGG_CHECK_LAUNCH("cgg_bias_act_nhwc");
  return 0;
}

// Stem tail of the BN-folded ResNet: relu(conv + bias) followed by MaxPool2d(3, stride 2, padding 1) in ONE pass over
// the channel-last bf16 convolution output. max and (+ bias, ReLU) commute (both are monotone per channel), so the
// kernel pools the raw convolution output and applies bias + ReLU to the pooled value: the 67-MB stem activation is
// read once and never rewritten (library sequence: bias/ReLU pass 22 us + max_pool 44 us at configs[1]).
__global__ __launch_bounds__(256) void cgg_bias_relu_maxpool_kernel(const uint4* __restrict__ x, const uint4* __restrict__ bias,
                                                                   uint4* __restrict__ y, int H, int W, int Ho, int Wo,
                                                                   int c8, long long nvec) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // output vector: (b, oy, ox, c8)
  if (i >= nvec) return;
  const int c = (int)(i % c8);
  long long p = i / c8;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = -__builtin_inff();
  const int y0 = 2 * oy - 1, x0 = 2 * ox - 1;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = y0 + dy;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = x0 + dx;
      if (ix < 0 || ix >= W) continue;
      const uint4 v = x[(((size_t)b * H + iy) * W + ix) * c8 + c];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        m[2 * k] = fmaxf(m[2 * k], cgg_bf2f((uint16_t)(w[k] & 0xffffu)));
        m[2 * k + 1] = fmaxf(m[2 * k + 1], cgg_bf2f((uint16_t)(w[k] >> 16)));
      }
    }
  }
  const uint4 bv = bias[c];
  const uint32_t bw[4] = {bv.x, bv.y, bv.z, bv.w};
  uint32_t o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // same rounding as the library sequence: bf16(conv + bias), then ReLU
    const float lo = fmaxf(cgg_bf2f(cgg_f2bf(m[2 * k] + cgg_bf2f((uint16_t)(bw[k] & 0xffffu)))), 0.f);
    const float hi = fmaxf(cgg_bf2f(cgg_f2bf(m[2 * k + 1] + cgg_bf2f((uint16_t)(bw[k] >> 16)))), 0.f);
    o[k] = cgg_pack2(cgg_f2bf(lo), cgg_f2bf(hi));
  }
  y[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

// parity mode's twin on f32 maps (the x3 stem's raw f32 output): 4 channels per thread
__global__ __launch_bounds__(256) void cgg_bias_relu_maxpool_f32_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ bias,
                                                                       f32x4* __restrict__ y, int H, int W, int Ho, int Wo,
                                                                       int c4, long long nvec, int x3a, int* __restrict__ flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // output vector: (b, oy, ox, c4)
  if (i >= nvec) return;
  const int c = (int)(i % c4);
  long long p = i / c4;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  const float ninf = -__builtin_inff();
  f32x4 m = {ninf, ninf, ninf, ninf};
  const int y0 = 2 * oy - 1, x0 = 2 * ox - 1;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = y0 + dy;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = x0 + dx;
      if (ix < 0 || ix >= W) continue;
      const f32x4 v = x[(((size_t)b * H + iy) * W + ix) * c4 + c];
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
    }
  }
  const f32x4 bv = bias[c];
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = fmaxf(m[k] + bv[k], 0.f);
  if (!x3a) {
    y[i] = o;
    return;
  }
  // x3a rows (csrc/x3.h): the 4 channels are half of an 8-channel group [8 hi | 8 lo]: 8 bytes of each piece
  uint2 h, l;
  cgg_x3_split4(o, h, l);
  uint2* yo = reinterpret_cast<uint2*>(y) + (i >> 1) * 4 + (i & 1);
  yo[0] = h;
  yo[2] = l;
  if (flag && !(fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3])) * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(flag, 1);
}

int* cgg_x3_overflow_flag_ptr();       // x3s_gemm.hip

static int brm_f32_launch(const float* x, const float* bias, float* y, int B, int H, int W, int C, int x3a, cgg_stream_t stream,
                          const char* who) {
  CGG_REQUIRE(x && bias && y, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(C % (x3a ? 8 : 4) == 0, CGG_EUNSUPPORTED, "%s: C %% %d != 0 (C=%d)", who, x3a ? 8 : 4, C);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(bias) && cgg_aligned16(y), CGG_EALIGN, "%s: pointers must be 16-byte aligned", who);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long nvec = (long long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(cgg_bias_relu_maxpool_f32_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)x, (const f32x4*)bias, (f32x4*)y, H, W, Ho, Wo, C / 4, nvec, x3a,
                     x3a ? cgg_x3_overflow_flag_ptr() : nullptr);
  CGG_CHECK_LAUNCH(who);
  return 0;
}

extern "C" int cgg_bias_relu_maxpool_nhwc_f32(const float* x, const float* bias, float* y, int B, int H, int W, int C,
                                              cgg_stream_t stream) {
  return brm_f32_launch(x, bias, y, B, H, W, C, 0, stream, "cgg_bias_relu_maxpool_nhwc_f32");
}

extern "C" int cgg_bias_relu_maxpool_nhwc_f32_x3a(const float* x, const float* bias, void* y_x3a, int B, int H, int W, int C,
                                                  cgg_stream_t stream) {
  return brm_f32_launch(x, bias, (float*)y_x3a, B, H, W, C, 1, stream, "cgg_bias_relu_maxpool_nhwc_f32_x3a");
}

extern "C" int cgg_bias_relu_maxpool_nhwc(const void* x, const void* bias, void* y, int B, int H, int W, int C,
                                          cgg_stream_t stream) {
  CGG_REQUIRE(x && bias && y, CGG_EINVAL, "cgg_bias_relu_maxpool_nhwc: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, CGG_EINVAL, "cgg_bias_relu_maxpool_nhwc: bad sizes");
  CGG_REQUIRE(C % 8 == 0, CGG_EUNSUPPORTED, "cgg_bias_relu_maxpool_nhwc: C %% 8 != 0 (C=%d)", C);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(bias) && cgg_aligned16(y), CGG_EALIGN,
              "cgg_bias_relu_maxpool_nhwc: pointers must be 16-byte aligned");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long nvec = (long long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(cgg_bias_relu_maxpool_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, (const uint4*)bias, (uint4*)y, H, W, Ho, Wo, C / 8, nvec);
  CGG_CHECK_LAUNCH("cgg_bias_relu_maxpool_nhwc");
  return 0;
}

// im2col for a 3x3 / padding 1 convolution on a channel-last bf16 map: rows = output pixels, columns = (ky, kx, c).
// Deep ResNet stages (64^2 / 32^2 maps, 256-512 channels) have too few output tiles for the implicit-GEMM convolution
// kernels (45-54 us for 9.7 GFLOP at configs[1]); as an explicit [M, 9C] x [9C, Cout] library GEMM with the bias + ReLU
// epilogue they take 18-20 us, and the patch matrix is only 19-38 MB.
__global__ __launch_bounds__(256) void cgg_im2col3x3_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int H, int W,
                                                           int Ho, int Wo, int c8, int stride, long long nvec) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (b, oy, ox, tap, c8)
  if (i >= nvec) return;
  const int c = (int)(i % c8);
  long long p = i / c8;
  const int tap = (int)(p % 9);
  p /= 9;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * H + iy) * W + ix) * c8 + c];
  y[i] = v;
}

extern "C" int cgg_im2col3x3_nhwc(const void* x, void* y, int B, int H, int W, int C, int stride, cgg_stream_t stream) {
  CGG_REQUIRE(x && y, CGG_EINVAL, "cgg_im2col3x3_nhwc: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2), CGG_EINVAL, "cgg_im2col3x3_nhwc: bad sizes");
  CGG_REQUIRE(C % 8 == 0, CGG_EUNSUPPORTED, "cgg_im2col3x3_nhwc: C %% 8 != 0 (C=%d)", C);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(y), CGG_EALIGN, "cgg_im2col3x3_nhwc: pointers must be 16-byte aligned");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long nvec = (long long)B * Ho * Wo * 9 * (C / 8);
  hipLaunchKernelGGL(cgg_im2col3x3_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, (uint4*)y, H, W, Ho, Wo, C / 8, stride, nvec);
  CGG_CHECK_LAUNCH("cgg_im2col3x3_nhwc");
  return 0;
}

// x[:, ::s, ::s, :] of a channel-last bf16 map as a contiguous tensor (input of the stride-s 1x1 downsample convolution
// of a ResNet stage, which then is a plain GEMM): 16-byte vectors, one read of the kept pixels only.
__global__ __launch_bounds__(256) void cgg_subsample_nhwc_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int H, int W,
                                                                int Ho, int Wo, int c8, int stride, long long nvec) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (b, oy, ox, c8)
  if (i >= nvec) return;
  const int c = (int)(i % c8);
  long long p = i / c8;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  y[i] = x[(((size_t)b * H + (size_t)oy * stride) * W + (size_t)ox * stride) * c8 + c];
}

extern "C" int cgg_subsample_nhwc(const void* x, void* y, int B, int H, int W, int C, int stride, cgg_stream_t stream) {
  CGG_REQUIRE(x && y, CGG_EINVAL, "cgg_subsample_nhwc: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && stride >= 1, CGG_EINVAL, "cgg_subsample_nhwc: bad sizes");
  CGG_REQUIRE(C % 8 == 0, CGG_EUNSUPPORTED, "cgg_subsample_nhwc: C %% 8 != 0 (C=%d)", C);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(y), CGG_EALIGN, "cgg_subsample_nhwc: pointers must be 16-byte aligned");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long nvec = (long long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(cgg_subsample_nhwc_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, (uint4*)y, H, W, Ho, Wo, C / 8, stride, nvec);
  CGG_CHECK_LAUNCH("cgg_subsample_nhwc");
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Batched 2-D transpose of f32 matrices, in (B, R, C) -> out (B, C, R): the NCHW <-> NHWC layout changes around the x3 training
// convolution (runtime._X3Conv3x3Fn; torch's strided copy ran them at ~2 TB/s: 1.1 ms per 1-GB map, eight of them per step at
// configs[2]). 64 x 64 tiles through LDS (row stride 65 floats: conflict-free column reads), 256 threads, 16-byte global accesses on
// both sides when R and C are multiples of 4 (scalar edge path otherwise).
// PADW > 0 (the NCHW -> zero-padded NHWC form): the output rows are the pixels of a (H + 2) x (PADW + 2) grid with a one-pixel
// border the kernel does not touch -- input column c = y PADW + x lands in output row (y + 1) (PADW + 2) + x + 1, batch stride
// `out_bstride` floats.
template <bool PAD>
__global__ __launch_bounds__(256) void cgg_transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                                int tiles_c, int tiles_r, int padw, long long out_bstride) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int bid = blockIdx.x;
  const int b = bid / (tiles_c * tiles_r);
  const int rem = bid - b * tiles_c * tiles_r;
  const int tr = rem / tiles_c, tc = rem - tr * tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const float* ib = in + (size_t)b * R * C;
  float* ob = out + (PAD ? (size_t)b * (size_t)out_bstride : (size_t)b * R * C);
  const bool vec = (R % 4 == 0) && (C % 4 == 0);
  // load: thread -> (row t / 16 + 16 i, 4 columns 4 (t % 16))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i, c = 4 * (t & 15);
    const int gr = r0 + r, gc = c0 + c;
    if (vec) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gr < R && gc < C) v = *reinterpret_cast<const f32x4*>(ib + (size_t)gr * C + gc);
      tile[r][c] = v[0];
      tile[r][c + 1] = v[1];
      tile[r][c + 2] = v[2];
      tile[r][c + 3] = v[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[r][c + e] = (gr < R && gc + e < C) ? ib[(size_t)gr * C + gc + e] : 0.f;
    }
  }
  __syncthreads();
  // store: thread -> (output row = input column t / 16 + 16 i, 4 output columns = input rows 4 (t % 16))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (t >> 4) + 16 * i, r = 4 * (t & 15);
    const int gc = c0 + c, gr = r0 + r;
    if (gc >= C) continue;
    size_t orow = (size_t)gc;
    if constexpr (PAD) {
      const int y = gc / padw, x = gc - y * padw;
      orow = (size_t)(y + 1) * (size_t)(padw + 2) + (size_t)(x + 1);
    }
    if (vec) {
      if (gr < R) *reinterpret_cast<f32x4*>(ob + orow * R + gr) = f32x4{tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gr + e < R) ob[orow * R + gr + e] = tile[r + e][c];
    }
  }
}

extern "C" int cgg_transpose_f32(const float* in, float* out, int B, int R, int C, cgg_stream_t stream) {
  CGG_REQUIRE(in && out, CGG_EINVAL, "cgg_transpose_f32: null pointer");
  CGG_REQUIRE(B > 0 && R > 0 && C > 0, CGG_EINVAL, "cgg_transpose_f32: bad sizes");
  CGG_REQUIRE(cgg_aligned16(in) && cgg_aligned16(out), CGG_EALIGN, "cgg_transpose_f32: 16-B alignment");
  const int tiles_c = (C + 63) / 64, tiles_r = (R + 63) / 64;
  const long long nb = (long long)B * tiles_c * tiles_r;
  CGG_REQUIRE(nb < (1ll << 31), CGG_EUNSUPPORTED, "cgg_transpose_f32: too many tiles");
  hipLaunchKernelGGL(cgg_transpose_f32_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, out, R, C, tiles_c,
                     tiles_r, 0, 0ll);
  CGG_CHECK_LAUNCH("cgg_transpose_f32");
  return CGG_OK;
}

// in (B, C, H, W) f32 contiguous -> the INTERIOR of out (B, H + 2, W + 2, C): channel-last with a one-pixel border that this call does
// not write (the caller zeroes it once): the padded channel-last maps of the x3 training convolution (runtime._X3Conv3x3Fn: the
// weight-gradient taps pair rows of the two padded maps at constant offsets) without a separate padding copy of a 1-GB map.
extern "C" int cgg_nchw_to_nhwc_pad1_f32(const float* in, float* out, int B, int C, int H, int W, cgg_stream_t stream) {
  CGG_REQUIRE(in && out, CGG_EINVAL, "cgg_nchw_to_nhwc_pad1_f32: null pointer");
  CGG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 31), CGG_EINVAL, "cgg_nchw_to_nhwc_pad1_f32: bad sizes");
  CGG_REQUIRE(cgg_aligned16(in) && cgg_aligned16(out), CGG_EALIGN, "cgg_nchw_to_nhwc_pad1_f32: 16-B alignment");
  const int R = C, Cc = H * W;
  const int tiles_c = (Cc + 63) / 64, tiles_r = (R + 63) / 64;
  const long long nb = (long long)B * tiles_c * tiles_r;
  CGG_REQUIRE(nb < (1ll << 31), CGG_EUNSUPPORTED, "cgg_nchw_to_nhwc_pad1_f32: too many tiles");
  hipLaunchKernelGGL(cgg_transpose_f32_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, out, R, Cc, tiles_c,
                     tiles_r, W, (long long)(H + 2) * (W + 2) * C);
  CGG_CHECK_LAUNCH("cgg_nchw_to_nhwc_pad1_f32");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// max |x| over an (M, N) f32 matrix (row stride ld) -> *amax (device scalar): the per-tensor pre-scale of the f32-class x3
// contractions' grad_output operands (x3.h "per-tensor pre-scale"; runtime._X3LinearFn.backward / _X3Conv3x3Fn.backward: autograd's
// grad_output behind the F.linear / conv calls of open_set/models/mask2former_head.py:787). One streaming pass, 16-byte loads,
// wave max by DPP-free shuffles, one u32 atomicMax per wave (|x| as a bit pattern is monotone; a NaN compares above inf and
// selects the default scale downstream).
__global__ __launch_bounds__(256) void cgg_absmax_f32_kernel(const float* __restrict__ x, long long ld4, long long n4_per_row,
                                                             long long total4, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  const f32x4* xv = reinterpret_cast<const f32x4*>(x);
  const bool dense = ld4 == n4_per_row;
  const long long stride = (long long)gridDim.x * 256;
  auto at = [&](long long u) -> f32x4 {
    long long off = u;
    if (!dense) {
      const long long r = u / n4_per_row;
      off = r * ld4 + (u - r * n4_per_row);
    }
    return __builtin_nontemporal_load(xv + off);
  };
  auto take = [&](const f32x4& v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = v[k];      // (copied to a scalar first: __builtin_bit_cast on the vector-element lvalue reads element 0, hipcc 7.2)
      const uint32_t b = __builtin_bit_cast(uint32_t, f) & 0x7fffffffu;
      m = b > m ? b : m;
    }
  };
  // four independent 16-B loads in flight per lane (one per trip measured 2.4 TB/s on the 352-MB gradient maps)
  long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; u + 3 * stride < total4; u += 4 * stride) {
    const f32x4 v0 = at(u), v1 = at(u + stride), v2 = at(u + 2 * stride), v3 = at(u + 3 * stride);
    take(v0); take(v1); take(v2); take(v3);
  }
  for (; u < total4; u += stride) take(at(u));
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)m, s, 64);
    m = o > m ? o : m;
  }
  // ONE atomic per workgroup, and only when it can raise the running maximum: thousands of atomics on one address serialise (2 048
  // workgroups x 4 wavefront atomics cost ~60 us of this kernel's 125 us on a 352-MB map; 512 workgroups: 69 us)
  __shared__ uint32_t wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = wm[0] > wm[1] ? wm[0] : wm[1], b = wm[2] > wm[3] ? wm[2] : wm[3];
    const uint32_t bm = a > b ? a : b;
    if (bm > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bm);
  }
}

// ReLU backward + max |.| in ONE pass: g = y > 0 ? gy : 0 (autograd's threshold_backward of a ReLU whose OUTPUT is y) and *amax =
// max |g| -- the two passes in front of the gradient contractions of a conv + frozen-BN + ReLU training node
// (runtime._X3ConvBnFn.backward, the trainable ResNet stage behind open_set/models/mask2former_head.py:787's inputs): the mask
// pass wrote g and cgg_absmax_f32 read it again. Dense tensors of n % 4 == 0 elements; g may alias gy.
__global__ __launch_bounds__(256) void cgg_relu_bwd_absmax_f32_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                      float* __restrict__ g, long long total4,
                                                                      uint32_t* __restrict__ out) {
  uint32_t m = 0;
  const f32x4* gv = reinterpret_cast<const f32x4*>(gy);
  const f32x4* yv = reinterpret_cast<const f32x4*>(y);
  f32x4* ov = reinterpret_cast<f32x4*>(g);
  const long long stride = (long long)gridDim.x * 256;
  auto one = [&](long long u, const f32x4& a, const f32x4& b) {
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = b[k] > 0.f ? a[k] : 0.f;
      r[k] = f;
      const uint32_t bits = __builtin_bit_cast(uint32_t, f) & 0x7fffffffu;
      m = bits > m ? bits : m;
    }
    ov[u] = r;
  };
  long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; u + stride < total4; u += 2 * stride) {              // two independent load pairs in flight per lane
    const f32x4 a0 = __builtin_nontemporal_load(gv + u), b0 = __builtin_nontemporal_load(yv + u);
    const f32x4 a1 = __builtin_nontemporal_load(gv + u + stride), b1 = __builtin_nontemporal_load(yv + u + stride);
    one(u, a0, b0);
    one(u + stride, a1, b1);
  }
  for (; u < total4; u += stride) one(u, __builtin_nontemporal_load(gv + u), __builtin_nontemporal_load(yv + u));
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)m, s, 64);
    m = o > m ? o : m;
  }
  __shared__ uint32_t wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = wm[0] > wm[1] ? wm[0] : wm[1], b = wm[2] > wm[3] ? wm[2] : wm[3];
    const uint32_t bm = a > b ? a : b;
    if (bm > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bm);
  }
}

extern "C" int cgg_relu_bwd_absmax_f32(const float* gy, const float* y, float* g, long long n, float* amax, cgg_stream_t stream) {
  CGG_REQUIRE(gy && y && g && amax, CGG_EINVAL, "cgg_relu_bwd_absmax_f32: null pointer");
  CGG_REQUIRE(n > 0 && n % 4 == 0, CGG_EUNSUPPORTED, "cgg_relu_bwd_absmax_f32: n=%lld must be a positive multiple of 4", n);
  CGG_REQUIRE(cgg_aligned16(gy) && cgg_aligned16(y) && cgg_aligned16(g), CGG_EALIGN, "cgg_relu_bwd_absmax_f32: 16-B alignment");
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(float), (hipStream_t)stream);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_relu_bwd_absmax_f32: memset failed");
  const long long total4 = n / 4;
  long long nb = (total4 + 256 * 4 - 1) / (256 * 4);
  nb = nb > 1024 ? 1024 : (nb < 1 ? 1 : nb);
  hipLaunchKernelGGL(cgg_relu_bwd_absmax_f32_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, gy, y, g, total4,
                     reinterpret_cast<uint32_t*>(amax));
  CGG_CHECK_LAUNCH("cgg_relu_bwd_absmax_f32");
  return CGG_OK;
}

extern "C" int cgg_absmax_f32(const float* x, int ld, int M, int N, float* amax, cgg_stream_t stream) {
  CGG_REQUIRE(x && amax, CGG_EINVAL, "cgg_absmax_f32: null pointer");
  CGG_REQUIRE(M > 0 && N > 0 && ld >= N, CGG_EINVAL, "cgg_absmax_f32: bad sizes");
  CGG_REQUIRE(N % 4 == 0 && ld % 4 == 0 && cgg_aligned16(x), CGG_EUNSUPPORTED, "cgg_absmax_f32: N, ld %% 4 and 16-B alignment");
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(float), (hipStream_t)stream);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_absmax_f32: memset failed");
  const long long total4 = (long long)M * (N / 4);
  // 512 workgroups: 6.5 TB/s on a 352-MB map; 256 / 1 024 / 2 048: 5.1 / 6.0 / 5.5 (measured with the one-atomic-per-workgroup tail)
  long long nb = (total4 + 256 * 8 - 1) / (256 * 8);
  nb = nb > 512 ? 512 : (nb < 1 ? 1 : nb);
  hipLaunchKernelGGL(cgg_absmax_f32_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long long)(ld / 4),
                     (long long)(N / 4), total4, reinterpret_cast<uint32_t*>(amax));
  CGG_CHECK_LAUNCH("cgg_absmax_f32");
  return CGG_OK;
}
