// K18: row kernels of the caption head's generator + cross-entropy (open_set/models/mask2former_head.py:551-565:
// `caption_generator(...)[1]` -> (B*34, 30522) logits -> CrossEntropyLoss(ignore_index=0)).
//
// The reference materialises the f32 logits, their log-softmax and the gradient of both (3 x 664 MB per step when the 10
// decoder outputs are batched). Here the generator runs in row chunks: a library GEMM writes one chunk of logits
// (bf16 in throughput mode, f32 in parity mode) into a buffer that is immediately consumed --
//   forward : one pass per row -> log-sum-exp and the row loss  lse - x[target]          (nothing else is kept)
//   backward: the chunk is recomputed, one pass turns it IN PLACE into  g_row (exp(x - lse) - onehot(target)),
//             the operand of the two gradient GEMMs (d hidden, d weight).
// One workgroup per row, 16-byte vector loads, online (max, sum) in registers, wavefront shuffles + a 4-slot LDS exchange.
#include "cgg_common.h"

template <typename T> struct CeVec;
template <> struct CeVec<float> {
  static constexpr int N = 4;
  __device__ static void load(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  }
  __device__ static void store(float* p, const float (&v)[8]) {
    const f32x4 a = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = a;
  }
  __device__ static float get(const float* p) { return *p; }
};
template <> struct CeVec<uint16_t> {
  static constexpr int N = 8;
  __device__ static void load(const uint16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static void store(uint16_t* p, const float (&v)[8]) {
    const uint4 u = make_uint4(cgg_pack2(cgg_f2bf(v[0]), cgg_f2bf(v[1])), cgg_pack2(cgg_f2bf(v[2]), cgg_f2bf(v[3])),
                               cgg_pack2(cgg_f2bf(v[4]), cgg_f2bf(v[5])), cgg_pack2(cgg_f2bf(v[6]), cgg_f2bf(v[7])));
    *reinterpret_cast<uint4*>(p) = u;
  }
  __device__ static float get(const uint16_t* p) { return cgg_bf2f(*p); }
};

template <typename T>
__global__ __launch_bounds__(256) void cgg_ce_rows_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                              float* __restrict__ loss, float* __restrict__ lse, int N,
                                                              int64_t ld, int64_t ignore_index) {
  constexpr int V = CeVec<T>::N;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* x = logits + (size_t)row * ld;
  float m = -INFINITY, s = 0.f;
  for (int c = tid * V; c < N; c += 256 * V) {
    float v[8];
    CeVec<T>::load(x + c, v);
    float cm = v[0];
#pragma unroll
    for (int e = 1; e < V; ++e) cm = fmaxf(cm, v[e]);
    const float mn = fmaxf(m, cm);
    float add = 0.f;
#pragma unroll
    for (int e = 0; e < V; ++e) add += __expf(v[e] - mn);
    s = s * __expf(m - mn) + add;
    m = mn;
  }
  // combine (m, s) over the wavefront, then over the 4 wavefronts
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
  }
  __shared__ float sm[4], ss[4];
  if (lane == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) S += sm[w] == -INFINITY ? 0.f : ss[w] * __expf(sm[w] - M);
    const float l = M + logf(S);
    lse[row] = l;
    const int64_t t = target[row];
    loss[row] = (t == ignore_index || t < 0 || t >= N) ? 0.f : l - CeVec<T>::get(x + t);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cgg_ce_rows_bwd_kernel(T* __restrict__ logits, const int64_t* __restrict__ target,
                                                              const float* __restrict__ lse, const float* __restrict__ grow,
                                                              int N, int64_t ld, int64_t ignore_index) {
  constexpr int V = CeVec<T>::N;
  const int row = blockIdx.x, tid = threadIdx.x;
  T* x = logits + (size_t)row * ld;
  const int64_t t = target[row];
  const bool live = !(t == ignore_index || t < 0 || t >= N);
  const float g = live ? grow[row] : 0.f;
  const float l = lse[row];
  for (int c = tid * V; c < N; c += 256 * V) {
    float v[8];
    CeVec<T>::load(x + c, v);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = g * (__expf(v[e] - l) - ((int64_t)(c + e) == t ? 1.f : 0.f));
    CeVec<T>::store(x + c, v);
  }
}

static int ce_check(const void* logits, const void* target, int M, int N, int64_t ld, int dtype, const char* who) {
  CGG_REQUIRE(logits && target, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(M > 0 && N > 0 && ld >= N, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(dtype == CGG_F32 || dtype == CGG_BF16, CGG_EUNSUPPORTED, "%s: dtype %d", who, dtype);
  const int v = dtype == CGG_F32 ? 4 : 8;
  CGG_REQUIRE(N % v == 0 && ld % v == 0 && cgg_aligned16(logits), CGG_EALIGN,
              "%s: N=%d / ld=%lld must be multiples of %d elements and the buffer 16-B aligned (pad the vocabulary)", who, N,
              (long long)ld, v);
  return CGG_OK;
}

extern "C" int cgg_ce_rows_forward(const void* logits, const int64_t* target, float* loss, float* lse, int M, int N,
                                   int64_t ld, int64_t ignore_index, int dtype, cgg_stream_t stream) {
  int rc = ce_check(logits, target, M, N, ld, dtype, "cgg_ce_rows_forward");
  if (rc != CGG_OK) return rc;
  CGG_REQUIRE(loss && lse, CGG_EINVAL, "cgg_ce_rows_forward: null output");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGG_F32)
    hipLaunchKernelGGL(cgg_ce_rows_fwd_kernel<float>, dim3(M), dim3(256), 0, s, (const float*)logits, target, loss, lse, N, ld,
                       ignore_index);
  else
    hipLaunchKernelGGL(cgg_ce_rows_fwd_kernel<uint16_t>, dim3(M), dim3(256), 0, s, (const uint16_t*)logits, target, loss, lse,
                       N, ld, ignore_index);
  CGG_CHECK_LAUNCH("cgg_ce_rows_forward");
  return CGG_OK;
}

extern "C" int cgg_ce_rows_backward(void* logits, const int64_t* target, const float* lse, const float* grad_rows, int M,
                                    int N, int64_t ld, int64_t ignore_index, int dtype, cgg_stream_t stream) {
  int rc = ce_check(logits, target, M, N, ld, dtype, "cgg_ce_rows_backward");
  if (rc != CGG_OK) return rc;
  CGG_REQUIRE(lse && grad_rows, CGG_EINVAL, "cgg_ce_rows_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGG_F32)
    hipLaunchKernelGGL(cgg_ce_rows_bwd_kernel<float>, dim3(M), dim3(256), 0, s, (float*)logits, target, lse, grad_rows, N, ld,
                       ignore_index);
  else
    hipLaunchKernelGGL(cgg_ce_rows_bwd_kernel<uint16_t>, dim3(M), dim3(256), 0, s, (uint16_t*)logits, target, lse, grad_rows,
                       N, ld, ignore_index);
  CGG_CHECK_LAUNCH("cgg_ce_rows_backward");
  return CGG_OK;
}


// ----------------------------------------------------------------------------------------------
// Prediction-only halves of the Hungarian matching costs (open_set/models/mask2former_head.py:320-390 -> mmdet
// CrossEntropyLossCost(use_sigmoid) / DiceCost(pred_act) on the point-sampled mask logits), one pass over the (rows, P) logits:
//   sp_sum[row]  = sum_p softplus(x)                (pos . t + neg . (1 - t) = sp_sum - x . t, since softplus(-x) - softplus(x) = -x)
//   sig[row][p]  = sigmoid(x)                       (the dice numerator's operand: the caller contracts it with the targets)
//   sig_sum[row] = sum_p sigmoid(x)  (square = 0)   |  sum_p sigmoid(x)^2  (square = 1, DiceCost(naive_dice = False))
// One workgroup per row, float4 loads / stores, sums in a fixed order (lane-serial, then the butterfly, then the four waves).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cgg_match_cost_rows_kernel(const float* __restrict__ x, float* __restrict__ sig,
                                                                  float* __restrict__ sp_sum, float* __restrict__ sig_sum, int P,
                                                                  int square) {
  const size_t row = blockIdx.x;
  const float4* xr = reinterpret_cast<const float4*>(x + row * (size_t)P);
  float4* sr = reinterpret_cast<float4*>(sig + row * (size_t)P);
  float a_sp = 0.f, a_sg = 0.f;
  auto one = [&](float v, float& s) {
    const float e = expf(-fabsf(v));                     // in (0, 1]
    a_sp += fmaxf(v, 0.f) + log1pf(e);
    s = (v >= 0.f ? 1.f : e) / (1.f + e);                // sigmoid without overflow on either side
    a_sg += square ? s * s : s;
  };
  for (int i = threadIdx.x; i < P / 4; i += 256) {
    const float4 v = xr[i];
    float4 s;
    one(v.x, s.x);
    one(v.y, s.y);
    one(v.z, s.z);
    one(v.w, s.w);
    sr[i] = s;
  }
  for (int o = 32; o > 0; o >>= 1) {
    a_sp += __shfl_xor(a_sp, o);
    a_sg += __shfl_xor(a_sg, o);
  }
  __shared__ float red[2][4];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a_sp;
    red[1][threadIdx.x >> 6] = a_sg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sp_sum[row] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sig_sum[row] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

extern "C" int cgg_match_cost_rows(const float* x, float* sig, float* sp_sum, float* sig_sum, int rows, int P, int square,
                                   cgg_stream_t stream) {
  CGG_REQUIRE(x && sig && sp_sum && sig_sum, CGG_EINVAL, "cgg_match_cost_rows: null pointer");
  CGG_REQUIRE(rows > 0 && P > 0, CGG_EINVAL, "cgg_match_cost_rows: bad sizes");
  CGG_REQUIRE(P % 4 == 0, CGG_EUNSUPPORTED, "cgg_match_cost_rows: P=%d must be a multiple of 4", P);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(sig), CGG_EALIGN, "cgg_match_cost_rows: x / sig must be 16-B aligned");
  hipLaunchKernelGGL(cgg_match_cost_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, sig, sp_sum, sig_sum, P, square);
  CGG_CHECK_LAUNCH("cgg_match_cost_rows");
  return CGG_OK;
}
