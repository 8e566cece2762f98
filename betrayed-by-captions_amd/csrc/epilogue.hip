// Channel-last bf16 epilogue for the BN-folded backbone convolutions (the [3P] mmdet ResNet
// Bottleneck tail `relu(bn(conv(x)) + identity)`): ONE in-place pass
//     y[r, c] = act(y[r, c] + bias[c] + res[r, c])
// instead of the three library passes (bias add, residual add, clamp). Pure HBM streaming: every lane owns one
// 16-byte vector (8 channels), loads are issued before any use, the bias vector comes from L1/L2.
#include "cgg_common.h"

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
