// K16: caption-grounding pair costs (open_set/models/losses/grounding_loss.py:32-58, called for each of the 10 decoder
// outputs at open_set/models/mask2former_head.py:542-548) and their backward.
//
// For every (caption i, image j) pair the reference builds sim = E_i P_j^T / temperature (T x Q; it physically repeats
// both operands B times, :23-30), a softmax over the queries (language -> vision attention, masked by the caption's token
// mask), a softmax over the TOKENS (vision -> language, unmasked -- as the reference), and reduces each attention-weighted
// distance map to one scalar:
//     cost_l2v[i][j] = sum_t m_t sum_q softmax_q(s)[t][q] (-s[t][q]) / max(n_i, 1)
//     cost_v2l[i][j] = sum_q sum_t softmax_t(s)[t][q] (-s[t][q]) / Q
// One workgroup per pair: the 64 x 128 (padded T x Q) score tile comes from v_mfma_f32_32x32x2_f32 (exact f32 products)
// straight from global operand rows, the token softmax is an in-register reduction (a lane owns one query column),
// the query softmax goes through a 33-KB LDS tile; the B^2 x T x Q score / attention tensors of the reference (3.6 MB x 4
// per layer at B = 16) are never stored. The backward recomputes the tile, forms d cost / d sim in place and writes it in
// the layout of the one batched GEMM that follows (grad_pred[j] = dsim[j]^T x captions, a library GEMM on the caller's
// side): dsim [Bp][Bc * T][Q].
#include "cgg_common.h"

// NW wavefronts per workgroup = NW x 32 query columns (4: Q <= 128, 8: Q <= 256, e.g. the 200 queries of configs[3])

__device__ __forceinline__ int gr_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <bool BWD, int NW>
__global__ __launch_bounds__(NW * 64) void cgg_grounding_kernel(const float* __restrict__ pred, const float* __restrict__ cap,
                                                            const int32_t* __restrict__ cmask, float* __restrict__ cost,
                                                            const float* __restrict__ gcost, float* __restrict__ dsim,
                                                            int Bp, int Bc, int Q, int T, int d, float inv_temp) {
  const int i = blockIdx.x, j = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = lane & 31, hi = lane >> 5;
  constexpr int GR_LD = NW * 32 + 4;                    // LDS row stride (floats) of the score tile
  constexpr int QP = NW * 32;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float gr_smem[];
  float* S = gr_smem;                                   // [64][GR_LD]
  float (*cst)[QP] = reinterpret_cast<float (*)[QP]>(S + 64 * GR_LD);     // [3][QP] per query column: max, sum, u
  float (*rst)[64] = reinterpret_cast<float (*)[64]>(S + 64 * GR_LD + 3 * QP);   // [3][64] per token row

  const int half = d >> 1;
  const int t0 = x, t1 = 32 + x, qq = 32 * wave + x;
  const bool ok0 = t0 < T, ok1 = t1 < T, okq = qq < Q;
  const float* ea0 = cap + ((size_t)i * T + (ok0 ? t0 : 0)) * d + hi * half;
  const float* ea1 = cap + ((size_t)i * T + (ok1 ? t1 : 0)) * d + hi * half;
  const float* pb = pred + ((size_t)j * Q + (okq ? qq : 0)) * d + hi * half;
  const float f0 = ok0 ? 1.f : 0.f, f1 = ok1 ? 1.f : 0.f, fq = okq ? 1.f : 0.f;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const bool two = T > 32;              // workgroup-uniform: captions of <= 32 tokens need one token tile
#pragma unroll 2
  for (int c = 0; c < half; c += 4) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(pb + c) * fq;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ea0 + c) * f0;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc0, 0, 0, 0);
    if (two) {
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ea1 + c) * f1;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e], acc1, 0, 0, 0);
    }
  }
  // ---- scaled scores: lane (x, hi) holds s[t = gr_row(r, hi) + 32 tt][q = qq] ----
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] *= inv_temp;
    acc1[r] *= inv_temp;
    const int t = gr_row(r, hi);
    S[t * GR_LD + qq] = acc0[r];
    S[(t + 32) * GR_LD + qq] = acc1[r];
    if (t < T) mx = fmaxf(mx, acc0[r]);
    if (t + 32 < T) mx = fmaxf(mx, acc1[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f, us = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = gr_row(r, hi);
    if (t < T) { const float p = __expf(acc0[r] - mx); sum += p; us += p * acc0[r]; }
    if (t + 32 < T) { const float p = __expf(acc1[r] - mx); sum += p; us += p * acc1[r]; }
  }
  sum += __shfl_xor(sum, 32);
  us += __shfl_xor(us, 32);
  if (hi == 0) {
    cst[0][qq] = mx;
    cst[1][qq] = sum;
    cst[2][qq] = us / sum;
  }
  __syncthreads();
  // ---- query softmax of every token row: one wavefront per row, lanes over the queries ----
  for (int t = wave; t < T; t += NW) {
    float v[QP / 64];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < QP / 64; ++c) {
      v[c] = lane + 64 * c < Q ? S[t * GR_LD + lane + 64 * c] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sm = 0.f, u = 0.f;
#pragma unroll
    for (int c = 0; c < QP / 64; ++c) {
      const float p = lane + 64 * c < Q ? __expf(v[c] - m) : 0.f;
      sm += p;
      u += lane + 64 * c < Q ? p * v[c] : 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); u += __shfl_xor(u, o); }
    if (lane == 0) { rst[0][t] = m; rst[1][t] = sm; rst[2][t] = u / sm; }
  }
  __syncthreads();
  // token count of the caption (every wave computes it: T <= 64 = one value per lane)
  float mt = lane < T ? (cmask[(size_t)i * T + lane] != 0 ? 1.f : 0.f) : 0.f;
  float n = mt;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  const float inv_n = 1.f / fmaxf(n, 1.f);
  if (!BWD) {
    if (wave == 0) {
      float l2v = lane < T ? mt * rst[2][lane] : 0.f;
      float v2l = 0.f;
#pragma unroll
      for (int c = 0; c < QP / 64; ++c) v2l += lane + 64 * c < Q ? cst[2][lane + 64 * c] : 0.f;
      for (int o = 32; o > 0; o >>= 1) { l2v += __shfl_xor(l2v, o); v2l += __shfl_xor(v2l, o); }
      if (lane == 0) {
        cost[(size_t)i * Bp + j] = -l2v * inv_n;
        cost[(size_t)Bc * Bp + (size_t)i * Bp + j] = -v2l / (float)Q;
      }
    }
  } else {
    const float g1 = -gcost[(size_t)i * Bp + j] * inv_n * inv_temp;
    const float g2 = -gcost[(size_t)Bc * Bp + (size_t)i * Bp + j] / (float)Q * inv_temp;
    float* out = dsim + ((size_t)j * Bc + i) * T * (size_t)Q;
    for (int idx = tid; idx < T * Q; idx += NT) {
      const int t = idx / Q, q2 = idx - t * Q;
      const float s = S[t * GR_LD + q2];
      const float m_t = cmask[(size_t)i * T + t] != 0 ? 1.f : 0.f;
      const float a = __expf(s - rst[0][t]) / rst[1][t];
      const float bb = __expf(s - cst[0][q2]) / cst[1][q2];
      out[idx] = g1 * m_t * a * (1.f + s - rst[2][t]) + g2 * bb * (1.f + s - cst[2][q2]);
    }
  }
}

static int grounding_check(const void* pred, const void* cap, const void* cmask, int Bp, int Bc, int Q, int T, int d,
                           const char* who) {
  CGG_REQUIRE(pred && cap && cmask, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(Bp > 0 && Bc > 0 && Q > 0 && T > 0 && d > 0, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(Q <= 256 && T <= 64, CGG_EUNSUPPORTED, "%s: Q=%d (<= 256) / T=%d (<= 64)", who, Q, T);
  CGG_REQUIRE(d % 8 == 0, CGG_EUNSUPPORTED, "%s: embedding width %d must be a multiple of 8", who, d);
  CGG_REQUIRE(cgg_aligned16(pred) && cgg_aligned16(cap), CGG_EALIGN, "%s: pred / cap must be 16-B aligned", who);
  return CGG_OK;
}

template <bool BWD>
static int grounding_launch(const float* pred, const float* cap, const int32_t* cap_mask, float* cost, const float* gcost,
                            float* dsim, int Bp, int Bc, int Q, int T, int d, float inv_t, hipStream_t s) {
  const int nw = Q <= 128 ? 4 : 8;
  const size_t lds = (size_t)(64 * (nw * 32 + 4) + 3 * nw * 32 + 3 * 64) * sizeof(float);
  if (nw == 4) {
    hipLaunchKernelGGL((cgg_grounding_kernel<BWD, 4>), dim3(Bc, Bp), dim3(256), lds, s, pred, cap, cap_mask, cost, gcost, dsim,
                       Bp, Bc, Q, T, d, inv_t);
  } else {
    auto kern = cgg_grounding_kernel<BWD, 8>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_grounding: cannot raise dynamic LDS to %zu", lds);
    hipLaunchKernelGGL(kern, dim3(Bc, Bp), dim3(512), lds, s, pred, cap, cap_mask, cost, gcost, dsim, Bp, Bc, Q, T, d, inv_t);
  }
  return CGG_OK;
}

extern "C" int cgg_grounding_pair_costs(const float* pred, const float* cap, const int32_t* cap_mask, float* cost,
                                        int Bp, int Bc, int Q, int T, int d, float inv_temperature,
                                        cgg_stream_t stream) {
  int rc = grounding_check(pred, cap, cap_mask, Bp, Bc, Q, T, d, "cgg_grounding_pair_costs");
  if (rc != CGG_OK) return rc;
  CGG_REQUIRE(cost, CGG_EINVAL, "cgg_grounding_pair_costs: null cost");
  rc = grounding_launch<false>(pred, cap, cap_mask, cost, nullptr, nullptr, Bp, Bc, Q, T, d, inv_temperature,
                               (hipStream_t)stream);
  if (rc != CGG_OK) return rc;
  CGG_CHECK_LAUNCH("cgg_grounding_pair_costs");
  return CGG_OK;
}

extern "C" int cgg_grounding_pair_costs_backward(const float* pred, const float* cap, const int32_t* cap_mask,
                                                 const float* grad_cost, float* dsim, int Bp, int Bc, int Q, int T, int d,
                                                 float inv_temperature, cgg_stream_t stream) {
  int rc = grounding_check(pred, cap, cap_mask, Bp, Bc, Q, T, d, "cgg_grounding_pair_costs_backward");
  if (rc != CGG_OK) return rc;
  CGG_REQUIRE(grad_cost && dsim, CGG_EINVAL, "cgg_grounding_pair_costs_backward: null pointer");
  rc = grounding_launch<true>(pred, cap, cap_mask, nullptr, grad_cost, dsim, Bp, Bc, Q, T, d, inv_temperature,
                              (hipStream_t)stream);
  if (rc != CGG_OK) return rc;
  CGG_CHECK_LAUNCH("cgg_grounding_pair_costs_backward");
  return CGG_OK;
}
