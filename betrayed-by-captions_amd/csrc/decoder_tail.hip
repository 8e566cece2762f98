// Tail of one query-decoder layer as ONE launch (M = B*Q ~ 200 rows, C = 256), replacing five:
//   cgg_layernorm_chain            y  = LN_a(sum of the FFN's split-K planes)        (DetrTransformerDecoderLayer's last norm)
//                                  z  = LN_b(y)                                       (decoder post_norm, mask2former_head.py:734)
//   3 x cgg_linear_rows_bf16       me = W3 relu(W2 relu(W1 z + b1) + b2) + b3         (mask_embed MLP, :741-746)
//   1 x cgg_linear_rows_bf16       qn = Wq (y + query_pos) + bq                       (the NEXT layer's cross-attention query, :829)
// Every one of those is latency-bound (7 workgroups, ~6.5 us each, of which ~1 us is arithmetic): a workgroup owns 32
// complete rows, so the whole chain runs out of LDS -- each linear's output is written back as bf16 MFMA A-fragments for
// the next one, the B fragments of the next weight are prefetched while the current one multiplies.
// Arithmetic per stage is that of the kernels it replaces (bf16 operands, f32 accumulate, f32 LayerNorm).
// X3 = true (parity mode): every linear of the chain is the f32-class f16 x 3 contraction of x3.h -- weights are x3 images,
// each fragment image has a hi and a lo half, outputs are un-scaled by the weight's column scale. Same launches, same
// structure; the f32-MFMA launch chain it replaces ran 123 x 17 us per forward.
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define DT_C 256
#define DT_STEPS (DT_C / 16)

#define DT_FRAG (DT_STEPS * 64)     // u32x4 slots of one 32 x 256 A-fragment image (16 KiB)

// a 32 x 256 A-fragment image: one 16-KiB half (bf16) or a hi and a lo half (x3: f16 pieces)
template <bool X3>
struct DtImg {
  u32x4* hi;
  u32x4* lo;
  __device__ __forceinline__ DtImg(u32x4* base) : hi(base), lo(X3 ? base + DT_FRAG : nullptr) {}
};
#define DT_IMG_SLOTS(X3) ((X3) ? 2 * DT_FRAG : DT_FRAG)

// B fragments of n-tile `nt` (32 output columns) of a packed [N/32][KS][64] weight, k-steps ks0 .. ks0 + 15
template <bool X3>
__device__ __forceinline__ void dt_load_b(u32x4 (&bf)[DT_STEPS], u32x4 (&bl)[X3 ? DT_STEPS : 1], const CggX3W& w, int nt, int lane,
                                          int KS = DT_STEPS, int ks0 = 0) {
#pragma unroll
  for (int s = 0; s < DT_STEPS; ++s) bf[s] = w.hi[((size_t)nt * KS + ks0 + s) * 64 + lane];
  if constexpr (X3) {
#pragma unroll
    for (int s = 0; s < DT_STEPS; ++s) bl[s] = w.lo[((size_t)nt * KS + ks0 + s) * 64 + lane];
  }
}

// acc = A W^T over the 16 k-steps held in (bf, bl), then the B fragments of the NEXT linear (n-tile nnt of `next`, k-steps
// nks0 .. of KS) are requested into the same registers. x3: the products that need only the hi weight pieces (al bh + ah bh)
// run first, so the next hi pieces are in flight under the ah bl pass; the scheduling barriers stop the compiler from
// hoisting the reloads above the MFMAs that still read the registers (it would double the 128 fragment registers and spill).
template <bool X3>
__device__ __forceinline__ void dt_mma(f32x16& acc, const DtImg<X3>& a, u32x4 (&bf)[DT_STEPS], u32x4 (&bl)[X3 ? DT_STEPS : 1],
                                       int lane, const CggX3W& next, bool has_next, int nnt, int KS = DT_STEPS, int nks0 = 0) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (X3) {
#pragma unroll
    for (int s = 0; s < DT_STEPS; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.lo[s * 64 + lane]), __builtin_bit_cast(f16x8, bf[s]),
                                                   acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.hi[s * 64 + lane]), __builtin_bit_cast(f16x8, bf[s]),
                                                   acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
#pragma unroll
      for (int s = 0; s < DT_STEPS; ++s) bf[s] = next.hi[((size_t)nnt * KS + nks0 + s) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < DT_STEPS; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.hi[s * 64 + lane]), __builtin_bit_cast(f16x8, bl[s]),
                                                   acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
#pragma unroll
      for (int s = 0; s < DT_STEPS; ++s) bl[s] = next.lo[((size_t)nnt * KS + nks0 + s) * 64 + lane];
    }
  } else {
#pragma unroll
    for (int s = 0; s < DT_STEPS; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.hi[s * 64 + lane]),
                                                    __builtin_bit_cast(bf16x8, bf[s]), acc, 0, 0, 0);
    if (has_next) dt_load_b<false>(bf, bl, next, nnt, lane, KS, nks0);
  }
}

// D layout (lane (j, hi5) of wave w holds column n = 32 w + j, rows (r&3) + 8 (r>>2) + 4 hi5) -> A-fragment image
template <bool X3>
__device__ __forceinline__ void dt_store_frag(const DtImg<X3>& frag, const float (&v)[16], int wave, int j, int hi5) {
  uint16_t* fh = reinterpret_cast<uint16_t*>(frag.hi);
  uint16_t* fl = reinterpret_cast<uint16_t*>(frag.lo);
  const int kstep = 2 * wave + (j >> 4), half = (j >> 3) & 1, e = j & 7;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi5;
    const int o = ((kstep * 64 + half * 32 + row) << 3) + e;
    if constexpr (X3) {
      uint16_t h, l;
      cgg_x3_split1(v[r], h, l);
      fh[o] = h;
      fl[o] = l;
    } else {
      fh[o] = cgg_f2bf(v[r]);
    }
  }
}

// four consecutive columns of one row -> the 8-byte half slot at 16-bit element offset `o` of the image
template <bool X3>
__device__ __forceinline__ void dt_store4(const DtImg<X3>& frag, int o, const f32x4 v) {
  if constexpr (X3) {
    uint2 h, l;
    cgg_x3_split4(v, h, l);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(frag.hi) + o) = h;
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(frag.lo) + o) = l;
  } else {
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(frag.hi) + o) =
        make_uint2(cgg_pack2(cgg_f2bf(v[0]), cgg_f2bf(v[1])), cgg_pack2(cgg_f2bf(v[2]), cgg_f2bf(v[3])));
  }
}

// the un-scaling of an x3 accumulator; compiled out in bf16 mode
template <bool X3>
__device__ __forceinline__ float dt_us(float a, float cs) { return X3 ? a * cs : a; }

// dynamic LDS above 64 KiB needs the attribute once per kernel
template <typename KernelT>
static int dt_raise_lds(KernelT kern, size_t lds, const char* who) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CGG_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise dynamic LDS to %zu: %s", who, lds, hipGetErrorString(e));
  }
  return CGG_OK;
}

template <bool X3>
__global__ __launch_bounds__(512) void cgg_decoder_tail_kernel(
    const float* __restrict__ planes, int nsum, long long plane_stride, int ld, const float* __restrict__ ga,
    const float* __restrict__ ba, float eps_a, const float* __restrict__ pos, int pos_rows,
    const float* __restrict__ gb, const float* __restrict__ bb, float eps_b, const CggX3W w1,
    const float* __restrict__ b1, const CggX3W w2, const float* __restrict__ b2,
    const CggX3W w3, const float* __restrict__ b3, const CggX3W wq,
    const float* __restrict__ bq, float* __restrict__ y, float* __restrict__ yp, float* __restrict__ me,
    float* __restrict__ qn, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dt_smem[];   // three 32 x 256 A-fragment images
  u32x4* const lds = reinterpret_cast<u32x4*>(dt_smem);
  const DtImg<X3> F0(lds), F1(lds + DT_IMG_SLOTS(X3)), F2(lds + 2 * DT_IMG_SLOTS(X3));
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
        // clamped (row M - 1 / plane nsum - 1: lines this wave reads anyway), zeroed after: predicated loads compile to
        // branch / load / s_waitcnt vmcnt(0) chains, one memory latency per row at the head of the kernel
        const f32x4 pv = *reinterpret_cast<const f32x4*>(planes + (size_t)(p < nsum ? p : nsum - 1) * plane_stride +
                                                         (size_t)(m < M ? m : M - 1) * ld + 4 * lane);
        ld4[rr][p] = (m < M && p < nsum) ? pv : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      acc4[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 8; ++p) acc4[rr] += ld4[rr][p];          // fixed order (adding the zero planes is exact)
    }
  }
  u32x4 bf[DT_STEPS], bl[X3 ? DT_STEPS : 1];
  // flies while the rows are normalised (x3: the hi pieces only -- 128 fragment registers do not fit beside the
  // normalisation; the lo pieces are requested after it and arrive under the first MFMA pass, which does not read them)
  if constexpr (X3) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < DT_STEPS; ++s) bf[s] = w1.hi[((size_t)wave * DT_STEPS + s) * 64 + lane];
  } else {
    dt_load_b<X3>(bf, bl, w1, wave, lane);
  }
  const float bias1 = b1[n], bias2 = b2[n], bias3 = b3[n], biasq = wq.hi ? bq[n] : 0.f;
  const float cs1 = X3 ? w1.scale[n] : 1.f, cs2 = X3 ? w2.scale[n] : 1.f, cs3 = X3 ? w3.scale[n] : 1.f;
  const float csq = (X3 && wq.hi) ? wq.scale[n] : 1.f;
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
        if (stage == 0) {
          f32x4 xp = v;
          if (m < M) {
            xp += *reinterpret_cast<const f32x4*>(pos + (size_t)(m % pos_rows) * DT_C + 4 * lane);
            *reinterpret_cast<f32x4*>(y + (size_t)m * DT_C + 4 * lane) = v;
            if (yp) *reinterpret_cast<f32x4*>(yp + (size_t)m * DT_C + 4 * lane) = xp;
          }
          if (wq.hi) dt_store4<X3>(F0, fslot + row * 8, xp);
        } else {
          dt_store4<X3>(F1, fslot + row * 8, v);
        }
      }
    }
  }
  if constexpr (X3) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < DT_STEPS; ++s) bl[s] = w1.lo[((size_t)wave * DT_STEPS + s) * 64 + lane];
  }
  __syncthreads();
  float v[16];

  f32x16 acc;
  // mask_embed[0]: relu(z W1^T + b1) -> F2
  dt_mma<X3>(acc, F1, bf, bl, lane, w2, true, wave);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(dt_us<X3>(acc[r], cs1) + bias1, 0.f);
  dt_store_frag<X3>(F2, v, wave, j, hi5);
  __syncthreads();
  // mask_embed[2]: relu(h W2^T + b2) -> F1  (every wave is past its reads of F1: barrier above)
  dt_mma<X3>(acc, F2, bf, bl, lane, w3, true, wave);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(dt_us<X3>(acc[r], cs2) + bias2, 0.f);
  dt_store_frag<X3>(F1, v, wave, j, hi5);
  __syncthreads();
  // mask_embed[4]: h W3^T + b3 -> global
  dt_mma<X3>(acc, F1, bf, bl, lane, wq, wq.hi != nullptr, wave);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    if (m < M) me[(size_t)m * DT_C + n] = dt_us<X3>(acc[r], cs3) + bias3;
  }
  if (wq.hi) {
    dt_mma<X3>(acc, F0, bf, bl, lane, wq, false, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (m < M) qn[(size_t)m * DT_C + n] = dt_us<X3>(acc[r], csq) + biasq;
    }
  }
}

// view of a packed 256-column-tile weight: bf16 image or x3 image
static inline CggX3W dt_view(bool x3, const void* p, int N, int K) {
  if (x3) return cgg_x3_view(p, N, K);
  return CggX3W{(const cgg_u32x4*)p, nullptr, nullptr};
}

static int dt_tail_launch(bool x3, const char* who, const float* planes, int nsum, int64_t plane_stride, int ld,
                          const float* gamma_a, const float* beta_a, float eps_a, const float* pos, int pos_rows,
                          const float* gamma_b, const float* beta_b, float eps_b, const void* w1, const float* b1,
                          const void* w2, const float* b2, const void* w3, const float* b3, const void* wq, const float* bq,
                          float* y, float* yp, float* mask_embed, float* qn, int M, int C, cgg_stream_t stream) {
  CGG_REQUIRE(planes && gamma_a && beta_a && pos && gamma_b && beta_b && w1 && b1 && w2 && b2 && w3 && b3 && y &&
                  mask_embed,
              CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(C == DT_C, CGG_EUNSUPPORTED, "%s: C=%d (only 256 is built)", who, C);
  CGG_REQUIRE(M > 0 && nsum >= 1 && pos_rows > 0 && ld >= C, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(!wq || (bq && qn), CGG_EINVAL, "%s: wq needs bq and qn", who);
  CGG_REQUIRE(cgg_aligned16(w1) && cgg_aligned16(w2) && cgg_aligned16(w3) && cgg_aligned16(wq), CGG_EALIGN,
              "%s: packed weights must be 16-B aligned", who);
  const size_t lds = (size_t)3 * DT_IMG_SLOTS(x3) * 16;
#define DT_TAIL(X3)                                                                                                          \
  do {                                                                                                                       \
    int rc = dt_raise_lds(cgg_decoder_tail_kernel<X3>, lds, who);                                                            \
    if (rc != CGG_OK) return rc;                                                                                             \
    hipLaunchKernelGGL(cgg_decoder_tail_kernel<X3>, dim3((M + 31) / 32), dim3(512), lds, (hipStream_t)stream, planes, nsum,   \
                       (long long)plane_stride, ld, gamma_a, beta_a, eps_a, pos, pos_rows, gamma_b, beta_b, eps_b,           \
                       dt_view(X3, w1, DT_C, DT_C), b1, dt_view(X3, w2, DT_C, DT_C), b2, dt_view(X3, w3, DT_C, DT_C), b3,    \
                       dt_view(X3, wq, DT_C, DT_C), bq, y, yp, mask_embed, qn, M);                                           \
  } while (0)
  if (x3) DT_TAIL(true);
  else DT_TAIL(false);
#undef DT_TAIL
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_decoder_tail_bf16(const float* planes, int nsum, int64_t plane_stride, int ld, const float* gamma_a,
                                     const float* beta_a, float eps_a, const float* pos, int pos_rows,
                                     const float* gamma_b, const float* beta_b, float eps_b, const void* w1,
                                     const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                     const void* wq, const float* bq, float* y, float* yp, float* mask_embed, float* qn,
                                     int M, int C, cgg_stream_t stream) {
  return dt_tail_launch(false, "cgg_decoder_tail_bf16", planes, nsum, plane_stride, ld, gamma_a, beta_a, eps_a, pos, pos_rows,
                        gamma_b, beta_b, eps_b, w1, b1, w2, b2, w3, b3, wq, bq, y, yp, mask_embed, qn, M, C, stream);
}

extern "C" int cgg_decoder_tail_x3(const float* planes, int nsum, int64_t plane_stride, int ld, const float* gamma_a,
                                   const float* beta_a, float eps_a, const float* pos, int pos_rows,
                                   const float* gamma_b, const float* beta_b, float eps_b, const void* w1,
                                   const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                   const void* wq, const float* bq, float* y, float* yp, float* mask_embed, float* qn,
                                   int M, int C, cgg_stream_t stream) {
  return dt_tail_launch(true, "cgg_decoder_tail_x3", planes, nsum, plane_stride, ld, gamma_a, beta_a, eps_a, pos, pos_rows,
                        gamma_b, beta_b, eps_b, w1, b1, w2, b2, w3, b3, wq, bq, y, yp, mask_embed, qn, M, C, stream);
}

// -------------------------------------------------------------------------------------------------
// Middle of a decoder layer: attention output projection + residual + post-norm LayerNorm, and (cross-attention
// only) the self-attention's fused q | k | v projection of the normalised rows -- one launch instead of
// cgg_linear_rows_bf16(LN epilogue) + cgg_linear_rows_bf16(q|k|v):
//   x1 = LN(core Wo^T + bo + res);   q = (x1 + pos) Wq^T + bq;  k = (x1 + pos) Wk^T + bk;  v = x1 Wv^T + bv.
// The LayerNorm runs row-major (one wavefront per row, shuffle reductions: the arithmetic of cgg_ln_chain_kernel) on an
// f32 LDS tile of the projection, not as the two-pass cross-wave reduction in MFMA layout of cgg_lr2_kernel<true>.
#define DM_TS 260          // f32 tile row stride (floats): rows r and r+4 of one store land in different banks

// LDS: image 0 (core, later x1 + pos) | union { f32 tile [32][DM_TS] (projection + bias + residual), image 1 (x1, written
// after every wave has left the tile) }
#define DM_TILE_BYTES (32 * DM_TS * 4)
#define DM_LDS_BYTES(X3) ((size_t)DT_IMG_SLOTS(X3) * 16 + ((size_t)DT_IMG_SLOTS(X3) * 16 > DM_TILE_BYTES ? (size_t)DT_IMG_SLOTS(X3) * 16 : DM_TILE_BYTES))

template <bool X3>
__global__ __launch_bounds__(512) void cgg_decoder_mid_kernel(
    const float* __restrict__ core, int ldc, const CggX3W wo, const float* __restrict__ bo,
    const float* __restrict__ res, int ldr, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ pos, int pos_rows, const CggX3W wqkv, const float* __restrict__ bqkv,
    float* __restrict__ x1, float* __restrict__ qo, float* __restrict__ kvo, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dt_smem[];
  u32x4* const lds = reinterpret_cast<u32x4*>(dt_smem);
  const DtImg<X3> F0(lds), F1(lds + DT_IMG_SLOTS(X3));
  float* const tile = reinterpret_cast<float*>(lds + DT_IMG_SLOTS(X3));
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
    // clamped, not predicated: a predicated load compiles to branch / load / s_waitcnt vmcnt(0) per row (four serial latencies)
    const f32x4 cv = *reinterpret_cast<const f32x4*>(core + (size_t)(m < M ? m : M - 1) * ldc + 4 * lane);
    cr[rr] = m < M ? cv : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 bf[DT_STEPS], bl[X3 ? DT_STEPS : 1];
  dt_load_b<X3>(bf, bl, wo, wave, lane);
  const float bias_o = bo[n];
  const float cso = X3 ? wo.scale[n] : 1.f;
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    const float rvv = res[(size_t)(m < M ? m : M - 1) * ldr + n];
    rv[r] = m < M ? rvv : 0.f;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) dt_store4<X3>(F0, fslot + (4 * wave + rr) * 8, cr[rr]);
  __syncthreads();
  // ---- B: projection + bias + residual -> f32 tile ----
  f32x16 acc;
  dt_mma<X3>(acc, F0, bf, bl, lane, wqkv, wqkv.hi != nullptr, wave);       // next: the q tile of this wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi5;
    tile[row * DM_TS + n] = dt_us<X3>(acc[r], cso) + bias_o + rv[r];
  }
  __syncthreads();
  // ---- C: row-major LayerNorm; x1 -> global, (x1 + pos) -> image 0, x1 kept in registers for image 1 ----
  f32x4 keep[4];
  {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * lane), be = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
    constexpr float inv_n = 1.f / (float)DT_C;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * wave + rr, m = m0 + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(&tile[row * DM_TS + 4 * lane]);
      float sm = (v[0] + v[1]) + (v[2] + v[3]);
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      const float mean = sm * inv_n;
      const f32x4 d = v - mean;
      float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q * inv_n + eps);
      v = d * rstd * g + be;
      keep[rr] = v;
      if (m < M) *reinterpret_cast<f32x4*>(x1 + (size_t)m * DT_C + 4 * lane) = v;
      if (wqkv.hi) {
        f32x4 xp = v;
        if (m < M) xp += *reinterpret_cast<const f32x4*>(pos + (size_t)(m % pos_rows) * DT_C + 4 * lane);
        dt_store4<X3>(F0, fslot + row * 8, xp);
      }
    }
  }
  if (!wqkv.hi) return;
  __syncthreads();                                            // every wave has read its tile rows
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) dt_store4<X3>(F1, fslot + (4 * wave + rr) * 8, keep[rr]);
  __syncthreads();
  // ---- D: q | k from x1 + pos, v from x1 (this wave's 32-column tile of each) ----
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    dt_mma<X3>(acc, c < 2 ? F0 : F1, bf, bl, lane, wqkv, c < 2, 8 * (c + 1) + wave);
    const float bias = bqkv[c * DT_C + n];
    const float cs = X3 ? wqkv.scale[c * DT_C + n] : 1.f;
    float* dst = c == 0 ? qo : kvo + (c - 1) * DT_C;
    const int ldd = c == 0 ? DT_C : 2 * DT_C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (m < M) dst[(size_t)m * ldd + n] = dt_us<X3>(acc[r], cs) + bias;
    }
  }
}

static int dt_mid_launch(bool x3, const char* who, const float* core, int ldc, const void* wo, const float* bo,
                         const float* res, int ldr, const float* gamma, const float* beta, float eps, const float* pos,
                         int pos_rows, const void* wqkv, const float* bqkv, float* x1, float* q, float* kv, int M, int C,
                         cgg_stream_t stream) {
  CGG_REQUIRE(core && wo && bo && res && gamma && beta && x1, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(C == DT_C, CGG_EUNSUPPORTED, "%s: C=%d (only 256 is built)", who, C);
  CGG_REQUIRE(M > 0 && ldc >= C && ldr >= C && ldc % 4 == 0, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(!wqkv || (bqkv && q && kv && pos && pos_rows > 0), CGG_EINVAL, "%s: the q|k|v stage needs bqkv, q, kv and pos", who);
  CGG_REQUIRE(cgg_aligned16(core) && cgg_aligned16(wo) && cgg_aligned16(wqkv) && cgg_aligned16(x1) && cgg_aligned16(pos) &&
                  cgg_aligned16(gamma) && cgg_aligned16(beta),
              CGG_EALIGN, "%s: 16-B alignment", who);
#define DT_MID(X3)                                                                                                           \
  do {                                                                                                                       \
    const size_t lds = DM_LDS_BYTES(X3);                                                                                     \
    int rc = dt_raise_lds(cgg_decoder_mid_kernel<X3>, lds, who);                                                             \
    if (rc != CGG_OK) return rc;                                                                                             \
    hipLaunchKernelGGL(cgg_decoder_mid_kernel<X3>, dim3((M + 31) / 32), dim3(512), lds, (hipStream_t)stream, core, ldc,      \
                       dt_view(X3, wo, DT_C, DT_C), bo, res, ldr, gamma, beta, eps, pos, pos_rows,                           \
                       dt_view(X3, wqkv, 3 * DT_C, DT_C), bqkv, x1, q, kv, M);                                               \
  } while (0)
  if (x3) DT_MID(true);
  else DT_MID(false);
#undef DT_MID
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_decoder_mid_bf16(const float* core, int ldc, const void* wo, const float* bo, const float* res, int ldr,
                                    const float* gamma, const float* beta, float eps, const float* pos, int pos_rows,
                                    const void* wqkv, const float* bqkv, float* x1, float* q, float* kv, int M, int C,
                                    cgg_stream_t stream) {
  return dt_mid_launch(false, "cgg_decoder_mid_bf16", core, ldc, wo, bo, res, ldr, gamma, beta, eps, pos, pos_rows, wqkv, bqkv,
                       x1, q, kv, M, C, stream);
}

extern "C" int cgg_decoder_mid_x3(const float* core, int ldc, const void* wo, const float* bo, const float* res, int ldr,
                                  const float* gamma, const float* beta, float eps, const float* pos, int pos_rows,
                                  const void* wqkv, const float* bqkv, float* x1, float* q, float* kv, int M, int C,
                                  cgg_stream_t stream) {
  return dt_mid_launch(true, "cgg_decoder_mid_x3", core, ldc, wo, bo, res, ldr, gamma, beta, eps, pos, pos_rows, wqkv, bqkv,
                       x1, q, kv, M, C, stream);
}

// -------------------------------------------------------------------------------------------------
// The decoder layer's FFN (256 -> F -> 256, ReLU) as ONE launch: workgroup (cb, rb) computes the 32 x 256 block
// h[rb, 256 cb ..] = relu(x W1^T + b1) and multiplies it straight away with the matching K-slice of W2 -- the
// split-K partition of the second projection IS the column partition of the first, so no workgroup ever needs another
// one's hidden block and h never leaves LDS. Output = F / 256 partial planes [cb][M][256] (b2 and the residual x in
// plane 0), summed in fixed order by cgg_decoder_tail_bf16 / cgg_layernorm_chain: deterministic, no atomics.
template <bool X3>
__global__ __launch_bounds__(512) void cgg_decoder_ffn_kernel(const float* __restrict__ x, int ldx, const CggX3W w1,
                                                              const float* __restrict__ b1, const CggX3W w2,
                                                              const float* __restrict__ b2, float* __restrict__ planes, int M,
                                                              int F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dt_smem[];   // two A-fragment images
  u32x4* const lds = reinterpret_cast<u32x4*>(dt_smem);
  const DtImg<X3> F0(lds), F1(lds + DT_IMG_SLOTS(X3));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int cb = blockIdx.x, m0 = blockIdx.y * 32;
  const int n = wave * 32 + j;
  const int fslot = ((lane >> 2) * 64 + ((lane >> 1) & 1) * 32) * 8 + 4 * (lane & 1);
  f32x4 xr[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int m = m0 + 4 * wave + rr;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(m < M ? m : M - 1) * ldx + 4 * lane);   // clamped, see the mid kernel
    xr[rr] = m < M ? xv : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 bf[DT_STEPS], bl[X3 ? DT_STEPS : 1];
  dt_load_b<X3>(bf, bl, w1, cb * 8 + wave, lane);             // W1 rows 256 cb + 32 wave .. (K = 256: 16 k-steps)
  const float bias1 = b1[cb * DT_C + n];
  const float bias2 = cb == 0 ? b2[n] : 0.f;
  const float cs1 = X3 ? w1.scale[cb * DT_C + n] : 1.f, cs2 = X3 ? w2.scale[n] : 1.f;
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    rv[r] = (cb == 0 && m < M) ? x[(size_t)m * ldx + n] : 0.f;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) dt_store4<X3>(F0, fslot + (4 * wave + rr) * 8, xr[rr]);
  __syncthreads();
  f32x16 acc;
  // next: W2 [256, F] packed [8 n-tiles][F / 16 k-steps][64]; this workgroup's K-slice = k-steps 16 cb .. 16 cb + 15
  dt_mma<X3>(acc, F0, bf, bl, lane, w2, true, wave, F >> 4, cb * DT_STEPS);
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(dt_us<X3>(acc[r], cs1) + bias1, 0.f);
  dt_store_frag<X3>(F1, v, wave, j, hi5);
  __syncthreads();
  dt_mma<X3>(acc, F1, bf, bl, lane, w2, false, 0);
  float* out = planes + (size_t)cb * M * DT_C;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    if (m < M) out[(size_t)m * DT_C + n] = dt_us<X3>(acc[r], cs2) + bias2 + rv[r];
  }
}

static int dt_ffn_launch(bool x3, const char* who, const float* x, int ldx, const void* w1, const float* b1, const void* w2,
                         const float* b2, float* planes, int M, int C, int F, cgg_stream_t stream) {
  CGG_REQUIRE(x && w1 && b1 && w2 && b2 && planes, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(C == DT_C, CGG_EUNSUPPORTED, "%s: C=%d (only 256 is built)", who, C);
  CGG_REQUIRE(M > 0 && F >= 256 && F % 256 == 0 && ldx >= C && ldx % 4 == 0, CGG_EUNSUPPORTED,
              "%s: F=%d must be a multiple of 256 (ldx=%d)", who, F, ldx);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(w1) && cgg_aligned16(w2), CGG_EALIGN, "%s: alignment", who);
#define DT_FFN(X3)                                                                                                           \
  do {                                                                                                                       \
    const size_t lds = (size_t)2 * DT_IMG_SLOTS(X3) * 16;                                                                    \
    int rc = dt_raise_lds(cgg_decoder_ffn_kernel<X3>, lds, who);                                                             \
    if (rc != CGG_OK) return rc;                                                                                             \
    hipLaunchKernelGGL(cgg_decoder_ffn_kernel<X3>, dim3(F / 256, (M + 31) / 32), dim3(512), lds, (hipStream_t)stream, x, ldx, \
                       dt_view(X3, w1, F, DT_C), b1, dt_view(X3, w2, DT_C, F), b2, planes, M, F);                            \
  } while (0)
  if (x3) DT_FFN(true);
  else DT_FFN(false);
#undef DT_FFN
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_decoder_ffn_bf16(const float* x, int ldx, const void* w1, const float* b1, const void* w2,
                                    const float* b2, float* planes, int M, int C, int F, cgg_stream_t stream) {
  return dt_ffn_launch(false, "cgg_decoder_ffn_bf16", x, ldx, w1, b1, w2, b2, planes, M, C, F, stream);
}

extern "C" int cgg_decoder_ffn_x3(const float* x, int ldx, const void* w1, const float* b1, const void* w2,
                                  const float* b2, float* planes, int M, int C, int F, cgg_stream_t stream) {
  return dt_ffn_launch(true, "cgg_decoder_ffn_x3", x, ldx, w1, b1, w2, b2, planes, M, C, F, stream);
}
