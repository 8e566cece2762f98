// Skinny linear for the query side of the decoder: y[M,N] = act(x[M,K] W[N,K]^T + b) (+ res),  M = B*Q ~ 200.
// Replaces the q/out projections, self-attention projections, FFN and head MLPs of
// DetrTransformerDecoderLayer / forward_head (open_set/models/mask2former_head.py:734-746, 829-840) that a
// BLAS library serves with a single 256x208 macro-tile workgroup (54 us per call measured on MI355X).
//
// Latency-bound, not roofline-bound: the whole problem is < 1 MB. Decomposition = (32-column n-tile,
// 128-row m-block) per workgroup, one 32-row m-tile per wave; W tile lives in LDS as bf16 MFMA B fragments
// (converted on the fly, K chunked by 512), x fragments go global -> registers -> bf16.
// SPLIT = f32-class accuracy via 3 bf16 MFMAs on (hi, lo) operands; otherwise plain bf16.
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define LR_KC 256  // K chunk staged per pass (16 MFMA k-steps)

__device__ __forceinline__ void lr_cvt8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo, bool split) {
  float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  uint16_t h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (split) cgg_split_bf(v[e], h[e], l[e]);
    else { h[e] = cgg_f2bf(v[e]); l[e] = 0; }
  }
  hi = u32x4{cgg_pack2(h[0], h[1]), cgg_pack2(h[2], h[3]), cgg_pack2(h[4], h[5]), cgg_pack2(h[6], h[7])};
  lo = u32x4{cgg_pack2(l[0], l[1]), cgg_pack2(l[2], l[3]), cgg_pack2(l[4], l[5]), cgg_pack2(l[6], l[7])};
}

// MODE 0: bf16 operands; 1: 3 bf16 MFMAs on (hi, lo) pairs (~2e-5 of sum |x w|); 2: EXACT f32 (v_mfma_f32_32x32x2_f32 on
// the un-rounded operands: the W tile sits in LDS as f32, same slot layout, k pairs = (8 hi + e) of each 16-block).
template <int MODE>
__global__ __launch_bounds__(256) void cgg_linear_rows_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ res, int ldr, float* __restrict__ y, int ldy, int M, int N, int K, int relu) {
  constexpr bool SPLIT = MODE == 1;
  constexpr bool EXACT = MODE == 2;
  constexpr int STEPS = LR_KC / 16;
  __shared__ __attribute__((aligned(16))) u32x4 w_hi[EXACT ? 1 : STEPS * 64];
  __shared__ __attribute__((aligned(16))) u32x4 w_lo[SPLIT ? STEPS * 64 : 1];
  __shared__ __attribute__((aligned(16))) f32x4 w_f32[EXACT ? STEPS * 64 * 2 : 1];
  const int n0 = blockIdx.x * 32;
  const int m0 = blockIdx.y * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int row = m0 + wave * 32 + j;            // A-operand row of this lane
  const bool row_ok = row < M;
  const bool wave_live = m0 + wave * 32 < M;     // wave-uniform
  const float* xr = x + (size_t)(row_ok ? row : 0) * ldx + 8 * hi5;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += LR_KC) {
    const int steps = min(LR_KC, K - k0) >> 4;
    // ---- (1) issue ALL x loads of this chunk (latency overlaps the W staging below) ----
    f32x4 xa[STEPS][2];
#pragma unroll
    for (int ks = 0; ks < STEPS; ++ks) {
      if (wave_live && row_ok && ks < steps) {
        xa[ks][0] = *reinterpret_cast<const f32x4*>(xr + k0 + ks * 16);
        xa[ks][1] = *reinterpret_cast<const f32x4*>(xr + k0 + ks * 16 + 4);
      } else {
        xa[ks][0] = zero4;
        xa[ks][1] = zero4;
      }
    }
    // ---- (2) stage W[n0..n0+31][k0..k0+16*steps) as B fragments; loads first, then convert + store ----
    //      slot (ks, lane) = W[n0 + (lane&31)][k0 + ks*16 + 8*(lane>>5) .. +7]
    f32x4 wa[STEPS / 4][2];
#pragma unroll
    for (int it = 0; it < STEPS / 4; ++it) {
      const int slot = tid + 256 * it;
      const int l = slot & 63, ks = slot >> 6;
      const int n = n0 + (l & 31);
      if (ks < steps && n < N) {
        const float* src = w + (size_t)n * K + k0 + ks * 16 + 8 * (l >> 5);
        wa[it][0] = *reinterpret_cast<const f32x4*>(src);
        wa[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
      } else {
        wa[it][0] = zero4;
        wa[it][1] = zero4;
      }
    }
    __syncthreads();  // previous chunk's fragment reads are done
#pragma unroll
    for (int it = 0; it < STEPS / 4; ++it) {
      if constexpr (EXACT) {
        w_f32[(tid + 256 * it) * 2] = wa[it][0];
        w_f32[(tid + 256 * it) * 2 + 1] = wa[it][1];
      } else {
        u32x4 h, lo;
        lr_cvt8(wa[it][0], wa[it][1], h, lo, SPLIT);
        w_hi[tid + 256 * it] = h;
        if (SPLIT) w_lo[tid + 256 * it] = lo;
      }
    }
    __syncthreads();
    if (!wave_live) continue;
    // ---- (3) MFMA ----
#pragma unroll
    for (int ks = 0; ks < STEPS; ++ks) {
      if (ks < steps) {
        if constexpr (EXACT) {
          const f32x4 b0 = w_f32[(ks * 64 + lane) * 2], b1 = w_f32[(ks * 64 + lane) * 2 + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[ks][0][e], b0[e], acc, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[ks][1][e], b1[e], acc, 0, 0, 0);
          continue;
        }
        u32x4 ah, al;
        lr_cvt8(xa[ks][0], xa[ks][1], ah, al, SPLIT);
        const bf16x8 vah = __builtin_bit_cast(bf16x8, ah);
        const bf16x8 vbh = __builtin_bit_cast(bf16x8, w_hi[ks * 64 + lane]);
        if (SPLIT) {
          const bf16x8 val = __builtin_bit_cast(bf16x8, al);
          const bf16x8 vbl = __builtin_bit_cast(bf16x8, w_lo[ks * 64 + lane]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, vbh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, vbl, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, vbh, acc, 0, 0, 0);
      }
    }
  }
  // ---- epilogue: acc[r] = y[m0 + wave*32 + (r&3) + 8*(r>>2) + 4*hi5][n0 + j]
  const int n = n0 + j;
  if (n < N && wave_live) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (m < M) {
        float v = acc[r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (res) v += res[(size_t)m * ldr + n];
        y[(size_t)m * ldy + n] = v;
      }
    }
  }
}

// row-wise LayerNorm of y = a (+ b): one wave per row (N <= 4096), eps as nn.LayerNorm
__global__ __launch_bounds__(256) void cgg_add_layernorm_kernel(const float* __restrict__ a,
                                                                const float* __restrict__ b,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ y, int rows, int N,
                                                                float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* ar = a + (size_t)row * N;
  const float* br = b ? b + (size_t)row * N : nullptr;
  float s = 0.f;
  for (int i = lane; i < N; i += 64) s += ar[i] + (br ? br[i] : 0.f);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)N;
  float v = 0.f;
  for (int i = lane; i < N; i += 64) {
    const float d = ar[i] + (br ? br[i] : 0.f) - mean;
    v += d * d;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = rsqrtf(v / (float)N + eps);
  for (int i = lane; i < N; i += 64) {
    const float d = ar[i] + (br ? br[i] : 0.f) - mean;
    y[(size_t)row * N + i] = d * rstd * gamma[i] + beta[i];
  }
}

extern "C" int cgg_linear_rows(const float* x, int ldx, const float* w, const float* bias, const float* res,
                               int ldr, float* y, int ldy, int M, int N, int K, int relu, int split,
                               cgg_stream_t stream) {
  CGG_REQUIRE(x && w && y, CGG_EINVAL, "cgg_linear_rows: null pointer");
  CGG_REQUIRE(M > 0 && N > 0 && K > 0, CGG_EINVAL, "cgg_linear_rows: bad sizes");
  CGG_REQUIRE(K % 16 == 0 && ldx % 4 == 0, CGG_EUNSUPPORTED, "cgg_linear_rows: K=%d (ldx=%d) must be a multiple of 16 (4)", K, ldx);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(w), CGG_EALIGN, "cgg_linear_rows: x / w must be 16-B aligned");
  dim3 grid((N + 31) / 32, (M + 127) / 128);
  hipStream_t s = (hipStream_t)stream;
  if (split == 2)
    hipLaunchKernelGGL(cgg_linear_rows_kernel<2>, grid, dim3(256), 0, s, x, ldx, w, bias, res, ldr, y, ldy, M, N, K, relu);
  else if (split)
    hipLaunchKernelGGL(cgg_linear_rows_kernel<1>, grid, dim3(256), 0, s, x, ldx, w, bias, res, ldr, y, ldy, M, N, K, relu);
  else
    hipLaunchKernelGGL(cgg_linear_rows_kernel<0>, grid, dim3(256), 0, s, x, ldx, w, bias, res, ldr, y, ldy, M, N, K, relu);
  CGG_CHECK_LAUNCH("cgg_linear_rows");
  return CGG_OK;
}

// Encoder-stream variant (43 008 rows x 256 at configs[1]): y = LN(a + b) with b in bf16 (a library GEMM's
// output) or f32, and up to three outputs written in the same pass so that no separate cast / add pass runs:
//   y32 (f32 residual stream), y16 = bf16(y) (next GEMM's input), yp16 = bf16(y + pos[row % pos_rows])
//   (the "query + query_pos" input of the next layer's sampling-offset GEMM). One wave per row, N == 256:
//   each lane owns 4 consecutive channels (16-B loads / stores).
__device__ __forceinline__ f32x4 cgg_ln_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 cgg_ln_load4(const uint16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}

template <typename AT, typename BT>
__global__ __launch_bounds__(256) void cgg_add_layernorm256_kernel(
    const AT* __restrict__ a, const BT* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ pos, int pos_rows, float* __restrict__ y32,
    uint16_t* __restrict__ y16, uint16_t* __restrict__ yp16, int rows, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t off = (size_t)row * 256 + lane * 4;
  f32x4 v = cgg_ln_load4(a + off);
  if (b != nullptr) v += cgg_ln_load4(b + off);
  float s = (v[0] + v[1]) + (v[2] + v[3]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.f / 256.f);
  f32x4 d = v - mean;
  float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q * (1.f / 256.f) + eps);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + lane * 4);
  f32x4 y = d * rstd * g + be;
  if (y32) *reinterpret_cast<f32x4*>(y32 + off) = y;
  if (y16) {
    *reinterpret_cast<uint2*>(y16 + off) =
        make_uint2(cgg_pack2(cgg_f2bf(y[0]), cgg_f2bf(y[1])), cgg_pack2(cgg_f2bf(y[2]), cgg_f2bf(y[3])));
  }
  if (yp16) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)(row % pos_rows) * 256 + lane * 4);
    const f32x4 z = y + p;
    *reinterpret_cast<uint2*>(yp16 + off) =
        make_uint2(cgg_pack2(cgg_f2bf(z[0]), cgg_f2bf(z[1])), cgg_pack2(cgg_f2bf(z[2]), cgg_f2bf(z[3])));
  }
}

// bf16 + bf16 -> bf16 (the encoder stream with the bf16 residual): HALF a wavefront per row, 8 channels = one 16-byte
// vector per lane, two rows per wave -- twice the bytes in flight per load instruction of the 4-channel kernel above.
__global__ __launch_bounds__(256) void cgg_add_layernorm256_bf16x8_kernel(
    const uint4* __restrict__ a, const uint4* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ pos, int pos_rows, uint4* __restrict__ y16,
    uint4* __restrict__ yp16, int rows, float eps) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  if (row >= rows) return;
  const size_t off = (size_t)row * 32 + l;
  const uint4 ua = a[off], ub = b[off];
  const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
  float v[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __uint_as_float(wa[k] << 16) + __uint_as_float(wb[k] << 16);
    v[2 * k + 1] = __uint_as_float(wa[k] & 0xffff0000u) + __uint_as_float(wb[k] & 0xffff0000u);
  }
  float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.f / 256.f);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] -= mean; q += v[k] * v[k]; }
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q * (1.f / 256.f) + eps);
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + l * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + l * 8 + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + l * 8), b1 = *reinterpret_cast<const f32x4*>(beta + l * 8 + 4);
  float y[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    y[k] = v[k] * rstd * g0[k] + b0[k];
    y[4 + k] = v[4 + k] * rstd * g1[k] + b1[k];
  }
  if (y16)
    y16[off] = make_uint4(cgg_pack2(cgg_f2bf(y[0]), cgg_f2bf(y[1])), cgg_pack2(cgg_f2bf(y[2]), cgg_f2bf(y[3])),
                          cgg_pack2(cgg_f2bf(y[4]), cgg_f2bf(y[5])), cgg_pack2(cgg_f2bf(y[6]), cgg_f2bf(y[7])));
  if (yp16) {
    const float* pr = pos + (size_t)(row % pos_rows) * 256 + l * 8;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(pr), p1 = *reinterpret_cast<const f32x4*>(pr + 4);
    yp16[off] = make_uint4(cgg_pack2(cgg_f2bf(y[0] + p0[0]), cgg_f2bf(y[1] + p0[1])), cgg_pack2(cgg_f2bf(y[2] + p0[2]), cgg_f2bf(y[3] + p0[3])),
                           cgg_pack2(cgg_f2bf(y[4] + p1[0]), cgg_f2bf(y[5] + p1[1])), cgg_pack2(cgg_f2bf(y[6] + p1[2]), cgg_f2bf(y[7] + p1[3])));
  }
}

template <typename AT>
static void add_layernorm_ex_launch(const AT* a, const void* b, int b_dtype, const float* gamma, const float* beta,
                                    const float* pos, int pos_rows, float* y32, void* y16, void* yp16, int rows, float eps,
                                    hipStream_t s) {
  dim3 grid((rows + 3) / 4);
  if (b_dtype == CGG_F32)
    hipLaunchKernelGGL((cgg_add_layernorm256_kernel<AT, float>), grid, dim3(256), 0, s, a, (const float*)b, gamma, beta,
                       pos, pos_rows, y32, (uint16_t*)y16, (uint16_t*)yp16, rows, eps);
  else
    hipLaunchKernelGGL((cgg_add_layernorm256_kernel<AT, uint16_t>), grid, dim3(256), 0, s, a, (const uint16_t*)b, gamma,
                       beta, pos, pos_rows, y32, (uint16_t*)y16, (uint16_t*)yp16, rows, eps);
}

extern "C" int cgg_add_layernorm_ex(const void* a, int a_dtype, const void* b, int b_dtype, const float* gamma,
                                    const float* beta, const float* pos, int pos_rows, float* y32, void* y16,
                                    void* yp16, int rows, int N, float eps, cgg_stream_t stream) {
  CGG_REQUIRE(a && gamma && beta, CGG_EINVAL, "cgg_add_layernorm_ex: null pointer");
  CGG_REQUIRE(y32 || y16 || yp16, CGG_EINVAL, "cgg_add_layernorm_ex: no output requested");
  CGG_REQUIRE(rows > 0, CGG_EINVAL, "cgg_add_layernorm_ex: bad sizes");
  CGG_REQUIRE(N == 256, CGG_EUNSUPPORTED, "cgg_add_layernorm_ex: N=%d (only 256 is built)", N);
  CGG_REQUIRE(!yp16 || (pos && pos_rows > 0), CGG_EINVAL, "cgg_add_layernorm_ex: yp16 needs pos");
  CGG_REQUIRE((a_dtype == CGG_F32 || a_dtype == CGG_BF16) && (b_dtype == CGG_F32 || b_dtype == CGG_BF16),
              CGG_EUNSUPPORTED, "cgg_add_layernorm_ex: a / b dtype %d / %d", a_dtype, b_dtype);
  hipStream_t s = (hipStream_t)stream;
  if (a_dtype == CGG_BF16 && b != nullptr && b_dtype == CGG_BF16 && y32 == nullptr && cgg_aligned16(a) && cgg_aligned16(b) &&
      cgg_aligned16(y16) && cgg_aligned16(yp16) && cgg_aligned16(gamma) && cgg_aligned16(beta) && cgg_aligned16(pos)) {
    hipLaunchKernelGGL(cgg_add_layernorm256_bf16x8_kernel, dim3((rows + 7) / 8), dim3(256), 0, s, (const uint4*)a,
                       (const uint4*)b, gamma, beta, pos, pos_rows, (uint4*)y16, (uint4*)yp16, rows, eps);
    CGG_CHECK_LAUNCH("cgg_add_layernorm_ex");
    return CGG_OK;
  }
  if (a_dtype == CGG_F32)
    add_layernorm_ex_launch((const float*)a, b, b_dtype, gamma, beta, pos, pos_rows, y32, y16, yp16, rows, eps, s);
  else
    add_layernorm_ex_launch((const uint16_t*)a, b, b_dtype, gamma, beta, pos, pos_rows, y32, y16, yp16, rows, eps, s);
  CGG_CHECK_LAUNCH("cgg_add_layernorm_ex");
  return CGG_OK;
}

// Last encoder layer of the inference stream: the query decoder's K / V projections read `memory_l + level_embed_l`
// (V) and `memory_l + level_embed_l + decoder_pos_l` (K) per level (mask2former_head.py:795-812). Both bf16 operands
// are produced here, by the LayerNorm that finishes the memory, written LEVEL-MAJOR ([level][batch][hw_l][256]) so each
// level is one contiguous (B * hw_l, 256) GEMM operand:  m16 = bf16(y + shift[s]),  mp16 = bf16((y + shift[s]) + pos[s]).
struct LnKvLevels { int n; int start[9]; };

template <typename AT, typename BT>
__global__ __launch_bounds__(256) void cgg_add_layernorm256_kv_kernel(
    const AT* __restrict__ a, const BT* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ shift, const float* __restrict__ pos, int S, int B,
    LnKvLevels lv, float* __restrict__ y32, uint16_t* __restrict__ m16, uint16_t* __restrict__ mp16, int rows, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t off = (size_t)row * 256 + lane * 4;
  f32x4 v = cgg_ln_load4(a + off);
  if (b != nullptr) v += cgg_ln_load4(b + off);
  float s = (v[0] + v[1]) + (v[2] + v[3]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.f / 256.f);
  f32x4 d = v - mean;
  float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q * (1.f / 256.f) + eps);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + lane * 4);
  f32x4 y = d * rstd * g + be;
  if (y32) *reinterpret_cast<f32x4*>(y32 + off) = y;
  const int bi = row / S, si = row - bi * S;
  int l = 0;
  while (l + 1 < lv.n && si >= lv.start[l + 1]) ++l;
  const int hw = lv.start[l + 1] - lv.start[l];
  const size_t orow = (size_t)B * lv.start[l] + (size_t)bi * hw + (si - lv.start[l]);
  const size_t ooff = orow * 256 + lane * 4;
  const f32x4 m = y + *reinterpret_cast<const f32x4*>(shift + (size_t)si * 256 + lane * 4);
  *reinterpret_cast<uint2*>(m16 + ooff) =
      make_uint2(cgg_pack2(cgg_f2bf(m[0]), cgg_f2bf(m[1])), cgg_pack2(cgg_f2bf(m[2]), cgg_f2bf(m[3])));
  const f32x4 z = m + *reinterpret_cast<const f32x4*>(pos + (size_t)si * 256 + lane * 4);
  *reinterpret_cast<uint2*>(mp16 + ooff) =
      make_uint2(cgg_pack2(cgg_f2bf(z[0]), cgg_f2bf(z[1])), cgg_pack2(cgg_f2bf(z[2]), cgg_f2bf(z[3])));
}

extern "C" int cgg_add_layernorm_kv(const void* a, int a_dtype, const void* b, int b_dtype, const float* gamma, const float* beta,
                                    const float* shift, const float* pos, int S, const int* level_start_host,
                                    int n_levels, float* y32, void* m16, void* mp16, int rows, int N, float eps,
                                    cgg_stream_t stream) {
  CGG_REQUIRE(a && gamma && beta && shift && pos && m16 && mp16 && level_start_host, CGG_EINVAL,
              "cgg_add_layernorm_kv: null pointer");
  CGG_REQUIRE(rows > 0 && S > 0 && rows % S == 0, CGG_EINVAL, "cgg_add_layernorm_kv: rows=%d not a multiple of S=%d", rows, S);
  CGG_REQUIRE(N == 256, CGG_EUNSUPPORTED, "cgg_add_layernorm_kv: N=%d (only 256 is built)", N);
  CGG_REQUIRE(n_levels >= 1 && n_levels <= 8, CGG_EUNSUPPORTED, "cgg_add_layernorm_kv: n_levels=%d (1..8)", n_levels);
  CGG_REQUIRE(b_dtype == CGG_F32 || b_dtype == CGG_BF16, CGG_EUNSUPPORTED, "cgg_add_layernorm_kv: b dtype %d", b_dtype);
  LnKvLevels lv;
  lv.n = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    lv.start[l] = level_start_host[l];
    CGG_REQUIRE(lv.start[l] >= 0 && lv.start[l] < S && (l == 0 ? lv.start[l] == 0 : lv.start[l] > lv.start[l - 1]),
                CGG_EINVAL, "cgg_add_layernorm_kv: level_start must start at 0 and increase (level %d: %d)", l, lv.start[l]);
  }
  lv.start[n_levels] = S;
  CGG_REQUIRE(a_dtype == CGG_F32 || a_dtype == CGG_BF16, CGG_EUNSUPPORTED, "cgg_add_layernorm_kv: a dtype %d", a_dtype);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4);
#define CGG_LNKV_LAUNCH(AT, BT)                                                                                          \
  hipLaunchKernelGGL((cgg_add_layernorm256_kv_kernel<AT, BT>), grid, dim3(256), 0, s, (const AT*)a, (const BT*)b, gamma, \
                     beta, shift, pos, S, rows / S, lv, y32, (uint16_t*)m16, (uint16_t*)mp16, rows, eps)
  if (a_dtype == CGG_F32 && b_dtype == CGG_F32) CGG_LNKV_LAUNCH(float, float);
  else if (a_dtype == CGG_F32) CGG_LNKV_LAUNCH(float, uint16_t);
  else if (b_dtype == CGG_F32) CGG_LNKV_LAUNCH(uint16_t, float);
  else CGG_LNKV_LAUNCH(uint16_t, uint16_t);
#undef CGG_LNKV_LAUNCH
  CGG_CHECK_LAUNCH("cgg_add_layernorm_kv");
  return CGG_OK;
}

extern "C" int cgg_add_layernorm(const float* a, const float* b, const float* gamma, const float* beta,
                                 float* y, int rows, int N, float eps, cgg_stream_t stream) {
  CGG_REQUIRE(a && gamma && beta && y, CGG_EINVAL, "cgg_add_layernorm: null pointer");
  CGG_REQUIRE(rows > 0 && N > 0, CGG_EINVAL, "cgg_add_layernorm: bad sizes");
  hipLaunchKernelGGL(cgg_add_layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, b,
                     gamma, beta, y, rows, N, eps);
  CGG_CHECK_LAUNCH("cgg_add_layernorm");
  return CGG_OK;
}
