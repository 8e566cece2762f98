// K3 in exact f32 for parity mode: mask_pred[b,q,p] = sum_c mask_embed[b,q,c] * mask_feature[b,c,p]
// (open_set/models/mask2former_head.py:748) on v_mfma_f32_32x32x2_f32 -- f32 products, f32 accumulation.
//
// Why it exists: the 3 x bf16 (hi, lo) contraction of cgg_mask_logits keeps 16 mantissa bits per operand, i.e. an error of
// ~2e-5 of sum |e_c f_c|: 4e-4 .. 8e-4 on logits of scale 20-30 at the BASELINE configs (tests/test_fullsize_gpu.py) --
// inside the north-star 1e-3, but most of the budget. This kernel reads the UN-packed f32 feature map (B operand: one
// row-coalesced dword load per channel and lane, 256 registers-worth streamed through 128-register halves) against
// mask_embed held in LDS with a bank-skewed stride, and has the same epilogue (row stores + ballot attention-mask bits) as
// the bf16 kernel. HBM-bound like it (the f32 feature is 4 bytes / element): ~55 us at configs[1].
#include "cgg_common.h"

#define MLF_LD 257   // LDS row stride of mask_embed in floats (odd: lanes i = consecutive queries hit distinct banks)

__global__ __launch_bounds__(256) void cgg_mask_logits_f32_kernel(const float* __restrict__ embed,
                                                                  const float* __restrict__ feat, float* __restrict__ out,
                                                                  uint32_t* __restrict__ bits, int Q, int npix, int T,
                                                                  int MT) {
  constexpr int C = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Es = reinterpret_cast<float*>(smem_raw);                 // [MT * 32][MLF_LD]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const float* eb = embed + (size_t)b * Q * C;
  for (int i = tid; i < MT * 32 * C; i += 256) {
    const int q = i >> 8, c = i & 255;
    Es[q * MLF_LD + c] = q < Q ? eb[(size_t)q * C + c] : 0.f;
  }
  __syncthreads();
  const float* fb = feat + (size_t)b * C * npix;
  float* ob = out ? out + (size_t)b * Q * npix : nullptr;
  uint32_t* bb = bits ? bits + (size_t)b * Q * T : nullptr;
  for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
    const int p = 32 * t + j;
    const bool pin = p < npix;
    const unsigned pc = pin ? p : npix - 1;
    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    // two halves of 128 channels: k-step s of a half uses channel c = 128 half + 2 s + hi (A and B agree)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      float bv[64];
#pragma unroll
      for (int s = 0; s < 64; ++s) {
        const float* rowp = fb + (size_t)(128 * half + 2 * s) * npix;            // uniform
        bv[s] = rowp[(unsigned)hi * (unsigned)npix + pc];
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (mt < MT) {                                                           // workgroup-uniform
          const float* ar = Es + (mt * 32 + j) * MLF_LD + 128 * half + hi;
#pragma unroll
          for (int s = 0; s < 64; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * s], bv[s], acc[mt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < MT) {
        const int rows_left = Q - mt * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (ob != nullptr && pin && ql < rows_left) ob[(size_t)(mt * 32 + ql) * npix + p] = acc[mt][r];
          if (bb != nullptr) {
            const unsigned long long m = __ballot(pin && acc[mt][r] < 0.f);
            const uint32_t w = hi ? (uint32_t)(m >> 32) : (uint32_t)m;
            if (j == 0 && ql < rows_left) bb[(size_t)(mt * 32 + ql) * T + t] = w;
          }
        }
      }
    }
  }
}

extern "C" int cgg_mask_logits_f32(const float* embed, const float* feat, float* out, uint32_t* bits, int B, int Q, int C,
                                   int npix, cgg_stream_t stream) {
  CGG_REQUIRE(embed && feat && (out || bits), CGG_EINVAL, "cgg_mask_logits_f32: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && npix > 0, CGG_EINVAL, "cgg_mask_logits_f32: bad sizes");
  CGG_REQUIRE(C == 256, CGG_EUNSUPPORTED, "cgg_mask_logits_f32: C=%d (only 256 is built)", C);
  CGG_REQUIRE(Q <= 128, CGG_EUNSUPPORTED, "cgg_mask_logits_f32: Q=%d > 128 (split the queries)", Q);
  const int MT = (Q + 31) / 32, T = (npix + 31) / 32;
  const size_t lds = (size_t)MT * 32 * MLF_LD * sizeof(float);
  int gx = (T + 3) / 4, cap = (512 + B - 1) / B;
  if (gx > cap) gx = cap;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)cgg_mask_logits_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_mask_logits_f32: cannot raise dynamic LDS to %zu", lds);
  }
  hipLaunchKernelGGL(cgg_mask_logits_f32_kernel, dim3(gx, B), dim3(256), lds, (hipStream_t)stream, embed, feat, out, bits, Q,
                     npix, T, MT);
  CGG_CHECK_LAUNCH("cgg_mask_logits_f32");
  return CGG_OK;
}
