// Tail of one query-decoder layer as ONE launch (M = B*Q ~ 200 rows, C = 256), replacing five:
//   cgg_layernorm_chain            y  = LN_a(sum of the FFN's split-K planes)        (DetrTransformerDecoderLayer's last norm)
//                                  z  = LN_b(y)                                       (decoder post_norm, mask2former_head.py:734)
//   3 x cgg_linear_rows_bf16       me = W3 relu(W2 relu(W1 z + b1) + b2) + b3         (mask_embed MLP, :741-746)
//   1 x cgg_linear_rows_bf16       qn = Wq (y + query_pos) + bq                       (the NEXT layer's cross-attention query, :829)
// Every one of those is latency-bound (7 workgroups, ~6.5 us each, of which ~1 us is arithmetic): a workgroup owns 32
// complete rows, so the whole chain runs out of LDS -- each linear's output is written back as bf16 MFMA A-fragments for
// the next one, the B fragments of the next weight are prefetched while the current one multiplies.
// Arithmetic per stage is that of the kernels it replaces (bf16 operands, f32 accumulate, f32 LayerNorm).
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define DT_C 256
#define DT_STEPS (DT_C / 16)

struct DtLds {
  u32x4 frag[3][DT_STEPS * 64];     // three 32 x 256 bf16 A-fragment images (16 KiB each)
};

// B fragments of n-tile `nt` (32 output columns) of a packed [N/32][16][64] weight
__device__ __forceinline__ void dt_load_b(u32x4 (&bf)[DT_STEPS], const u32x4* __restrict__ wp, int nt, int lane) {
#pragma unroll
  for (int s = 0; s < DT_STEPS; ++s) bf[s] = wp[((size_t)nt * DT_STEPS + s) * 64 + lane];
}

__device__ __forceinline__ void dt_mma(f32x16& acc, const u32x4* __restrict__ a_frag, const u32x4 (&bf)[DT_STEPS], int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < DT_STEPS; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_frag[s * 64 + lane]),
                                                  __builtin_bit_cast(bf16x8, bf[s]), acc, 0, 0, 0);
}

// D layout (lane (j, hi5) of wave w holds column n = 32 w + j, rows (r&3) + 8 (r>>2) + 4 hi5) -> bf16 A-fragment image
__device__ __forceinline__ void dt_store_frag(u32x4* __restrict__ frag, const float (&v)[16], int wave, int j, int hi5) {
  uint16_t* f16 = reinterpret_cast<uint16_t*>(frag);
  const int kstep = 2 * wave + (j >> 4), half = (j >> 3) & 1, e = j & 7;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi5;
    f16[((kstep * 64 + half * 32 + row) << 3) + e] = cgg_f2bf(v[r]);
  }
}

__global__ __launch_bounds__(512) void cgg_decoder_tail_kernel(
    const float* __restrict__ planes, int nsum, long long plane_stride, int ld, const float* __restrict__ ga,
    const float* __restrict__ ba, float eps_a, const float* __restrict__ pos, int pos_rows,
    const float* __restrict__ gb, const float* __restrict__ bb, float eps_b, const u32x4* __restrict__ w1,
    const float* __restrict__ b1, const u32x4* __restrict__ w2, const float* __restrict__ b2,
    const u32x4* __restrict__ w3, const float* __restrict__ b3, const u32x4* __restrict__ wq,
    const float* __restrict__ bq, float* __restrict__ y, float* __restrict__ yp, float* __restrict__ me,
    float* __restrict__ qn, int M) {
  __shared__ DtLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const int n = wave * 32 + j;

  // ---- phase 1, row-major (the arithmetic and reduction order of cgg_ln_chain_kernel): wave w owns rows 4w .. 4w+3,
  // lane l the columns 4l .. 4l+3; all 4 x nsum plane loads of a wave are in flight together
  f32x4 acc4[4];
  {
    f32x4 ld4[4][8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int m = m0 + 4 * wave + rr;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        ld4[rr][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < M && p < nsum)
          ld4[rr][p] = *reinterpret_cast<const f32x4*>(planes + (size_t)p * plane_stride + (size_t)m * ld + 4 * lane);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      acc4[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 8; ++p) acc4[rr] += ld4[rr][p];          // fixed order (adding the zero planes is exact)
    }
  }
  u32x4 bf[DT_STEPS];
  dt_load_b(bf, w1, wave, lane);                       // flies while the rows are normalised
  const float bias1 = b1[n], bias2 = b2[n], bias3 = b3[n], biasq = wq ? bq[n] : 0.f;
  {
    const f32x4 g_a = *reinterpret_cast<const f32x4*>(ga + 4 * lane), b_a = *reinterpret_cast<const f32x4*>(ba + 4 * lane);
    const f32x4 g_b = *reinterpret_cast<const f32x4*>(gb + 4 * lane), b_b = *reinterpret_cast<const f32x4*>(bb + 4 * lane);
    constexpr float inv_n = 1.f / (float)DT_C;
    // A-fragment slot of (row, columns 4l .. 4l+3): k-step l/4, half (l/2)&1, elements 4(l&1) .. +3 -> one 8-byte store
    const int fslot = ((lane >> 2) * 64 + ((lane >> 1) & 1) * 32) * 8 + 4 * (lane & 1);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * wave + rr, m = m0 + row;
      f32x4 v = acc4[rr];
#pragma unroll
      for (int stage = 0; stage < 2; ++stage) {
        float sm = (v[0] + v[1]) + (v[2] + v[3]);
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        const float mean = sm * inv_n;
        const f32x4 d = v - mean;
        float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * inv_n + (stage == 0 ? eps_a : eps_b));
        v = (v - mean) * rstd * (stage == 0 ? g_a : g_b) + (stage == 0 ? b_a : b_b);
        uint16_t* f16 = reinterpret_cast<uint16_t*>(L.frag[stage == 0 ? 0 : 1]) + fslot + row * 8;
        if (stage == 0) {
          f32x4 xp = v;
          if (m < M) {
            xp += *reinterpret_cast<const f32x4*>(pos + (size_t)(m % pos_rows) * DT_C + 4 * lane);
            *reinterpret_cast<f32x4*>(y + (size_t)m * DT_C + 4 * lane) = v;
            if (yp) *reinterpret_cast<f32x4*>(yp + (size_t)m * DT_C + 4 * lane) = xp;
          }
          if (wq)
            *reinterpret_cast<uint2*>(f16) =
                make_uint2(cgg_pack2(cgg_f2bf(xp[0]), cgg_f2bf(xp[1])), cgg_pack2(cgg_f2bf(xp[2]), cgg_f2bf(xp[3])));
        } else {
          *reinterpret_cast<uint2*>(f16) =
              make_uint2(cgg_pack2(cgg_f2bf(v[0]), cgg_f2bf(v[1])), cgg_pack2(cgg_f2bf(v[2]), cgg_f2bf(v[3])));
        }
      }
    }
  }
  __syncthreads();
  float v[16];

  f32x16 acc;
  // mask_embed[0]: relu(z W1^T + b1) -> frag[2]
  dt_mma(acc, L.frag[1], bf, lane);
  dt_load_b(bf, w2, wave, lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r] + bias1, 0.f);
  dt_store_frag(L.frag[2], v, wave, j, hi5);
  __syncthreads();
  // mask_embed[2]: relu(h W2^T + b2) -> frag[1]  (every wave is past its reads of frag[1]: barrier above)
  dt_mma(acc, L.frag[2], bf, lane);
  dt_load_b(bf, w3, wave, lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r] + bias2, 0.f);
  dt_store_frag(L.frag[1], v, wave, j, hi5);
  __syncthreads();
  // mask_embed[4]: h W3^T + b3 -> global
  dt_mma(acc, L.frag[1], bf, lane);
  if (wq) dt_load_b(bf, wq, wave, lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    if (m < M) me[(size_t)m * DT_C + n] = acc[r] + bias3;
  }
  if (wq) {
    dt_mma(acc, L.frag[0], bf, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (m < M) qn[(size_t)m * DT_C + n] = acc[r] + biasq;
    }
  }
}

extern "C" int cgg_decoder_tail_bf16(const float* planes, int nsum, int64_t plane_stride, int ld, const float* gamma_a,
                                     const float* beta_a, float eps_a, const float* pos, int pos_rows,
                                     const float* gamma_b, const float* beta_b, float eps_b, const void* w1,
                                     const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                     const void* wq, const float* bq, float* y, float* yp, float* mask_embed, float* qn,
                                     int M, int C, cgg_stream_t stream) {
  CGG_REQUIRE(planes && gamma_a && beta_a && pos && gamma_b && beta_b && w1 && b1 && w2 && b2 && w3 && b3 && y &&
                  mask_embed,
              CGG_EINVAL, "cgg_decoder_tail_bf16: null pointer");
  CGG_REQUIRE(C == DT_C, CGG_EUNSUPPORTED, "cgg_decoder_tail_bf16: C=%d (only 256 is built)", C);
  CGG_REQUIRE(M > 0 && nsum >= 1 && pos_rows > 0 && ld >= C, CGG_EINVAL, "cgg_decoder_tail_bf16: bad sizes");
  CGG_REQUIRE(!wq || (bq && qn), CGG_EINVAL, "cgg_decoder_tail_bf16: wq needs bq and qn");
  CGG_REQUIRE(cgg_aligned16(w1) && cgg_aligned16(w2) && cgg_aligned16(w3) && cgg_aligned16(wq), CGG_EALIGN,
              "cgg_decoder_tail_bf16: packed weights must be 16-B aligned");
  hipLaunchKernelGGL(cgg_decoder_tail_kernel, dim3((M + 31) / 32), dim3(512), 0, (hipStream_t)stream, planes, nsum,
                     (long long)plane_stride, ld, gamma_a, beta_a, eps_a, pos, pos_rows, gamma_b, beta_b, eps_b,
                     (const u32x4*)w1, b1, (const u32x4*)w2, b2, (const u32x4*)w3, b3, (const u32x4*)wq, bq, y, yp,
                     mask_embed, qn, M);
  CGG_CHECK_LAUNCH("cgg_decoder_tail_bf16");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// Middle of a decoder layer: attention output projection + residual + post-norm LayerNorm, and (cross-attention
// only) the self-attention's fused q | k | v projection of the normalised rows -- one launch instead of
// cgg_linear_rows_bf16(LN epilogue) + cgg_linear_rows_bf16(q|k|v):
//   x1 = LN(core Wo^T + bo + res);   q = (x1 + pos) Wq^T + bq;  k = (x1 + pos) Wk^T + bk;  v = x1 Wv^T + bv.
// The LayerNorm runs row-major (one wavefront per row, shuffle reductions: the arithmetic of cgg_ln_chain_kernel) on an
// f32 LDS tile of the projection, not as the two-pass cross-wave reduction in MFMA layout of cgg_lr2_kernel<true>.
#define DM_TS 260          // f32 tile row stride (floats): rows r and r+4 of one store land in different banks

struct DmLds {
  u32x4 frag0[DT_STEPS * 64];           // core, later bf16(x1 + pos)
  union {
    float tile[32 * DM_TS];             // projection + bias + residual, f32
    u32x4 frag1[DT_STEPS * 64];         // bf16(x1), written after every wave has left the tile
  };
};

__global__ __launch_bounds__(512) void cgg_decoder_mid_kernel(
    const float* __restrict__ core, int ldc, const u32x4* __restrict__ wo, const float* __restrict__ bo,
    const float* __restrict__ res, int ldr, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ pos, int pos_rows, const u32x4* __restrict__ wqkv, const float* __restrict__ bqkv,
    float* __restrict__ x1, float* __restrict__ qo, float* __restrict__ kvo, int M) {
  __shared__ DmLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const int n = wave * 32 + j;
  const int fslot = ((lane >> 2) * 64 + ((lane >> 1) & 1) * 32) * 8 + 4 * (lane & 1);   // see cgg_decoder_tail_kernel

  // ---- A: attention output rows -> bf16 A fragments; residual / bias / Wo in flight meanwhile ----
  f32x4 cr[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int m = m0 + 4 * wave + rr;
    cr[rr] = m < M ? *reinterpret_cast<const f32x4*>(core + (size_t)m * ldc + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 bf[DT_STEPS];
  dt_load_b(bf, wo, wave, lane);
  const float bias_o = bo[n];
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    rv[r] = m < M ? res[(size_t)m * ldr + n] : 0.f;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    uint16_t* f16 = reinterpret_cast<uint16_t*>(L.frag0) + fslot + (4 * wave + rr) * 8;
    *reinterpret_cast<uint2*>(f16) = make_uint2(cgg_pack2(cgg_f2bf(cr[rr][0]), cgg_f2bf(cr[rr][1])),
                                                cgg_pack2(cgg_f2bf(cr[rr][2]), cgg_f2bf(cr[rr][3])));
  }
  __syncthreads();
  // ---- B: projection + bias + residual -> f32 tile ----
  f32x16 acc;
  dt_mma(acc, L.frag0, bf, lane);
  if (wqkv) dt_load_b(bf, wqkv, wave, lane);                  // q tile of this wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi5;
    L.tile[row * DM_TS + n] = acc[r] + bias_o + rv[r];
  }
  __syncthreads();
  // ---- C: row-major LayerNorm; x1 -> global, bf16(x1 + pos) -> frag0, bf16(x1) kept for frag1 ----
  uint2 keep[4];
  {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * lane), be = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
    constexpr float inv_n = 1.f / (float)DT_C;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * wave + rr, m = m0 + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(&L.tile[row * DM_TS + 4 * lane]);
      float sm = (v[0] + v[1]) + (v[2] + v[3]);
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      const float mean = sm * inv_n;
      const f32x4 d = v - mean;
      float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q * inv_n + eps);
      v = d * rstd * g + be;
      keep[rr] = make_uint2(cgg_pack2(cgg_f2bf(v[0]), cgg_f2bf(v[1])), cgg_pack2(cgg_f2bf(v[2]), cgg_f2bf(v[3])));
      if (m < M) *reinterpret_cast<f32x4*>(x1 + (size_t)m * DT_C + 4 * lane) = v;
      if (wqkv) {
        f32x4 xp = v;
        if (m < M) xp += *reinterpret_cast<const f32x4*>(pos + (size_t)(m % pos_rows) * DT_C + 4 * lane);
        uint16_t* f16 = reinterpret_cast<uint16_t*>(L.frag0) + fslot + row * 8;
        *reinterpret_cast<uint2*>(f16) =
            make_uint2(cgg_pack2(cgg_f2bf(xp[0]), cgg_f2bf(xp[1])), cgg_pack2(cgg_f2bf(xp[2]), cgg_f2bf(xp[3])));
      }
    }
  }
  if (!wqkv) return;
  __syncthreads();                                            // every wave has read its tile rows
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(L.frag1) + fslot + (4 * wave + rr) * 8) = keep[rr];
  __syncthreads();
  // ---- D: q | k from x1 + pos, v from x1 (this wave's 32-column tile of each) ----
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    dt_mma(acc, c < 2 ? L.frag0 : L.frag1, bf, lane);
    if (c < 2) dt_load_b(bf, wqkv, 8 * (c + 1) + wave, lane);
    const float bias = bqkv[c * DT_C + n];
    float* dst = c == 0 ? qo : kvo + (c - 1) * DT_C;
    const int ldd = c == 0 ? DT_C : 2 * DT_C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (m < M) dst[(size_t)m * ldd + n] = acc[r] + bias;
    }
  }
}

extern "C" int cgg_decoder_mid_bf16(const float* core, int ldc, const void* wo, const float* bo, const float* res, int ldr,
                                    const float* gamma, const float* beta, float eps, const float* pos, int pos_rows,
                                    const void* wqkv, const float* bqkv, float* x1, float* q, float* kv, int M, int C,
                                    cgg_stream_t stream) {
  CGG_REQUIRE(core && wo && bo && res && gamma && beta && x1, CGG_EINVAL, "cgg_decoder_mid_bf16: null pointer");
  CGG_REQUIRE(C == DT_C, CGG_EUNSUPPORTED, "cgg_decoder_mid_bf16: C=%d (only 256 is built)", C);
  CGG_REQUIRE(M > 0 && ldc >= C && ldr >= C && ldc % 4 == 0, CGG_EINVAL, "cgg_decoder_mid_bf16: bad sizes");
  CGG_REQUIRE(!wqkv || (bqkv && q && kv && pos && pos_rows > 0), CGG_EINVAL,
              "cgg_decoder_mid_bf16: the q|k|v stage needs bqkv, q, kv and pos");
  CGG_REQUIRE(cgg_aligned16(core) && cgg_aligned16(wo) && cgg_aligned16(wqkv) && cgg_aligned16(x1) && cgg_aligned16(pos) &&
                  cgg_aligned16(gamma) && cgg_aligned16(beta),
              CGG_EALIGN, "cgg_decoder_mid_bf16: 16-B alignment");
  hipLaunchKernelGGL(cgg_decoder_mid_kernel, dim3((M + 31) / 32), dim3(512), 0, (hipStream_t)stream, core, ldc,
                     (const u32x4*)wo, bo, res, ldr, gamma, beta, eps, pos, pos_rows, (const u32x4*)wqkv, bqkv, x1, q, kv, M);
  CGG_CHECK_LAUNCH("cgg_decoder_mid_bf16");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// The decoder layer's FFN (256 -> F -> 256, ReLU) as ONE launch: workgroup (cb, rb) computes the 32 x 256 block
// h[rb, 256 cb ..] = relu(x W1^T + b1) and multiplies it straight away with the matching K-slice of W2 -- the
// split-K partition of the second projection IS the column partition of the first, so no workgroup ever needs another
// one's hidden block and h never leaves LDS. Output = F / 256 partial planes [cb][M][256] (b2 and the residual x in
// plane 0), summed in fixed order by cgg_decoder_tail_bf16 / cgg_layernorm_chain: deterministic, no atomics.
__global__ __launch_bounds__(512) void cgg_decoder_ffn_kernel(const float* __restrict__ x, int ldx,
                                                              const u32x4* __restrict__ w1, const float* __restrict__ b1,
                                                              const u32x4* __restrict__ w2, const float* __restrict__ b2,
                                                              float* __restrict__ planes, int M, int F) {
  __shared__ __attribute__((aligned(16))) u32x4 frag[2][DT_STEPS * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int cb = blockIdx.x, m0 = blockIdx.y * 32;
  const int n = wave * 32 + j;
  const int fslot = ((lane >> 2) * 64 + ((lane >> 1) & 1) * 32) * 8 + 4 * (lane & 1);
  f32x4 xr[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int m = m0 + 4 * wave + rr;
    xr[rr] = m < M ? *reinterpret_cast<const f32x4*>(x + (size_t)m * ldx + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 bf[DT_STEPS];
  dt_load_b(bf, w1, cb * 8 + wave, lane);                     // W1 rows 256 cb + 32 wave .. (K = 256: 16 k-steps)
  const float bias1 = b1[cb * DT_C + n];
  const float bias2 = cb == 0 ? b2[n] : 0.f;
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    rv[r] = (cb == 0 && m < M) ? x[(size_t)m * ldx + n] : 0.f;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    uint16_t* f16 = reinterpret_cast<uint16_t*>(frag[0]) + fslot + (4 * wave + rr) * 8;
    *reinterpret_cast<uint2*>(f16) = make_uint2(cgg_pack2(cgg_f2bf(xr[rr][0]), cgg_f2bf(xr[rr][1])),
                                                cgg_pack2(cgg_f2bf(xr[rr][2]), cgg_f2bf(xr[rr][3])));
  }
  __syncthreads();
  f32x16 acc;
  dt_mma(acc, frag[0], bf, lane);
  // W2 [256, F] packed [8 n-tiles][F / 16 k-steps][64]: this workgroup's K-slice = k-steps 16 cb .. 16 cb + 15
  const int KS2 = F >> 4;
#pragma unroll
  for (int s = 0; s < DT_STEPS; ++s) bf[s] = w2[((size_t)wave * KS2 + cb * DT_STEPS + s) * 64 + lane];
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r] + bias1, 0.f);
  dt_store_frag(frag[1], v, wave, j, hi5);
  __syncthreads();
  dt_mma(acc, frag[1], bf, lane);
  float* out = planes + (size_t)cb * M * DT_C;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    if (m < M) out[(size_t)m * DT_C + n] = acc[r] + bias2 + rv[r];
  }
}

extern "C" int cgg_decoder_ffn_bf16(const float* x, int ldx, const void* w1, const float* b1, const void* w2,
                                    const float* b2, float* planes, int M, int C, int F, cgg_stream_t stream) {
  CGG_REQUIRE(x && w1 && b1 && w2 && b2 && planes, CGG_EINVAL, "cgg_decoder_ffn_bf16: null pointer");
  CGG_REQUIRE(C == DT_C, CGG_EUNSUPPORTED, "cgg_decoder_ffn_bf16: C=%d (only 256 is built)", C);
  CGG_REQUIRE(M > 0 && F >= 256 && F % 256 == 0 && ldx >= C && ldx % 4 == 0, CGG_EUNSUPPORTED,
              "cgg_decoder_ffn_bf16: F=%d must be a multiple of 256 (ldx=%d)", F, ldx);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(w1) && cgg_aligned16(w2), CGG_EALIGN, "cgg_decoder_ffn_bf16: alignment");
  hipLaunchKernelGGL(cgg_decoder_ffn_kernel, dim3(F / 256, (M + 31) / 32), dim3(512), 0, (hipStream_t)stream, x, ldx,
                     (const u32x4*)w1, b1, (const u32x4*)w2, b2, planes, M, F);
  CGG_CHECK_LAUNCH("cgg_decoder_ffn_bf16");
  return CGG_OK;
}
