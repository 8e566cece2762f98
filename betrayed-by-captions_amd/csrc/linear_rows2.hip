// Throughput-mode query-side linear for the decoder (M = B*Q ~ 200 rows): the successor of cgg_linear_rows for
// bf16 mode. What changed, and why (MI355X measurements of the old kernel: 9-10 us per 200x256x256 call, 46 us for
// the K = 2048 FFN projection, 122 calls + 37 LayerNorm + 36 `x + pos` launches per forward):
//   * the weight arrives PRE-PACKED: bf16, MFMA-B-fragment order [N/32][K/16][64 lanes][8] (made once per weight
//     and cached by the host), so a wave's B operand for one k-step is ONE coalesced 1-KiB load -- no f32 read,
//     no in-kernel conversion, no LDS staging of W;
//   * one workgroup = 32 rows x 256 columns (8 waves, one 32x32 MFMA tile each); the 32 x K activation block is
//     converted to bf16 fragments ONCE into LDS and shared by the 8 waves;
//   * because a workgroup owns complete rows when N == 256, the residual add + LayerNorm run in the epilogue
//     (two-pass mean / variance over LDS), and a second output `y + pos` (the next projection's input) is written
//     in the same pass -- the separate LayerNorm and add launches disappear;
//   * K can be split over gridDim.z (one partial plane per split, summed in fixed order by the following
//     LayerNorm-chain kernel: deterministic, no atomics) so the K = 2048 projection uses 56 workgroups instead of 7;
//   * ReLU can be limited to the first `relu_cols` columns, which lets cls_embed / v2l_transform / mask_embed[0]
//     (open_set/models/mask2former_head.py:734-746) run as ONE GEMM over concatenated weights.
//   * X3 = true (parity mode, runtime precision 'fp32'): the same kernel on the f32-class f16 x 3 contraction of x3.h -- the
//     weight is an x3 image (hi / lo fragments + per-column scale), the activation block is split into (hi, lo) f16 fragment
//     images, three MFMAs per k-step into the one accumulator; it replaces the f32-MFMA cgg_linear_rows_kernel<2> of round 2
//     (17 us per call: 128 dependent v_mfma_f32_32x32x2_f32 per tile) at the bf16 kernel's launch time.
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define LR2_KC 256  // K chunk per LDS fill (16 MFMA k-steps)

template <bool LN, bool X3>
__global__ __launch_bounds__(512) void cgg_lr2_kernel(
    const float* __restrict__ x, int ldx, const u32x4* __restrict__ wp, const u32x4* __restrict__ wlo,
    const float* __restrict__ wscale, const float* __restrict__ bias,
    const float* __restrict__ res, int ldr, float* __restrict__ y, int ldy, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, const float* __restrict__ pos, int pos_rows,
    float* __restrict__ yp, int ldyp, int M, int N, int K, int relu_cols, const float* __restrict__ x2, int ldx2,
    int x2_col, float* __restrict__ y2, int ldy2, int y2_col) {
  constexpr int STEPS = LR2_KC / 16;
  // fused projections over concatenated weights (self-attention q | k | v): 256-column blocks from x2_col on read
  // the second input, blocks from y2_col on write the second output (both multiples of 256 -> workgroup-uniform)
  if (x2 != nullptr && (int)blockIdx.x * 256 >= x2_col) { x = x2; ldx = ldx2; }
  int ncol0 = 0;
  if (y2 != nullptr && (int)blockIdx.x * 256 >= y2_col) { y = y2; ldy = ldy2; ncol0 = y2_col; }
  __shared__ __attribute__((aligned(16))) u32x4 a_frag[STEPS * 64];
  __shared__ __attribute__((aligned(16))) u32x4 a_lo[X3 ? STEPS * 64 : 1];
  __shared__ float red[8][32];
  __shared__ float stat[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int m0 = blockIdx.y * 32;
  const int nt = blockIdx.x * 8 + wave;           // this wave's 32-column tile
  const int n = nt * 32 + j;
  const bool tile_live = nt * 32 < N;             // wave-uniform
  const int KS = K >> 4;
  const int ksplit = gridDim.z;
  const int ks_per = (KS + ksplit - 1) / ksplit;
  const int ks_lo = blockIdx.z * ks_per;
  const int ks_hi = min(ks_lo + ks_per, KS);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // epilogue operands are fetched NOW: with 7-56 workgroups in flight nothing hides a dependent L2 round trip,
  // so every load the epilogue needs is issued before the MFMA loop instead of after it
  const bool col_ok = tile_live && n < N;
  const bool first = blockIdx.z == 0;
  const float bv = (col_ok && bias && first) ? bias[n] : 0.f;
  const float cs = (X3 && col_ok) ? wscale[n] : 1.f;
  float resv[16], posv[16];
  float lg = 0.f, lb = 0.f;
  if (LN && col_ok) {
    lg = gamma[n];
    lb = beta[n];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    resv[r] = (res && first && col_ok && m < M) ? res[(size_t)m * ldr + n] : 0.f;
    posv[r] = (yp && col_ok && m < M) ? pos[(size_t)(m % pos_rows) * N + n] : 0.f;
  }

  for (int kc = ks_lo; kc < ks_hi; kc += STEPS) {
    const int steps = min(STEPS, ks_hi - kc);
    // B fragments of this chunk: issued first so the loads fly while the A block is staged
    u32x4 bf[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      bf[s] = (tile_live && s < steps) ? wp[((size_t)nt * KS + kc + s) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    u32x4 bl[X3 ? STEPS : 1];
    if constexpr (X3) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
        bl[s] = (tile_live && s < steps) ? wlo[((size_t)nt * KS + kc + s) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    }
    // A: rows m0..m0+31, k = 16*kc .. : thread -> (row, 8-float group), coalesced along k
    __syncthreads();   // previous chunk's fragment reads are done
    const int groups = steps * 2;               // 8-float groups per row in this chunk
    for (int idx = tid; idx < 32 * groups; idx += 512) {
      const int row = idx / groups, kg = idx - row * groups;
      const int m = m0 + row;
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (m < M) {
        const float* src = x + (size_t)m * ldx + (size_t)kc * 16 + kg * 8;
        v0 = *reinterpret_cast<const f32x4*>(src);
        v1 = *reinterpret_cast<const f32x4*>(src + 4);
      }
      if constexpr (X3) {
        u32x4 ph, pl;
        cgg_x3_split8(v0, v1, ph, pl);
        a_frag[(kg >> 1) * 64 + (kg & 1) * 32 + row] = ph;
        a_lo[(kg >> 1) * 64 + (kg & 1) * 32 + row] = pl;
      } else {
        const u32x4 p = {cgg_pack2(cgg_f2bf(v0[0]), cgg_f2bf(v0[1])), cgg_pack2(cgg_f2bf(v0[2]), cgg_f2bf(v0[3])),
                         cgg_pack2(cgg_f2bf(v1[0]), cgg_f2bf(v1[1])), cgg_pack2(cgg_f2bf(v1[2]), cgg_f2bf(v1[3]))};
        a_frag[(kg >> 1) * 64 + (kg & 1) * 32 + row] = p;
      }
    }
    __syncthreads();
    if (tile_live) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (s < steps) {
          if constexpr (X3) {
            cgg_x3_mfma(acc, a_frag[s * 64 + lane], a_lo[s * 64 + lane], bf[s], bl[s]);
          } else {
            const bf16x8 va = __builtin_bit_cast(bf16x8, a_frag[s * 64 + lane]);
            const bf16x8 vb = __builtin_bit_cast(bf16x8, bf[s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: acc[r] = out[m0 + (r&3) + 8*(r>>2) + 4*hi5][n] ----
  float v[16];
  {
    const bool do_relu = n < relu_cols;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = (X3 ? acc[r] * cs : acc[r]) + bv;
      if (do_relu) t = fmaxf(t, 0.f);
      t += resv[r];
      v[r] = col_ok ? t : 0.f;
    }
  }
  if (ksplit > 1) {
    // split-K: every split writes its own partial plane y[z][M][ldy] with plain stores; the consumer
    // (cgg_layernorm_chain, nsum = ksplit) adds the planes in a fixed order -> deterministic, no atomics, no memset
    if (col_ok) {
      float* yz = y + (size_t)blockIdx.z * M * ldy;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
        if (m < M) yz[(size_t)m * ldy + n - ncol0] = v[r];
      }
    }
    return;
  }
  if (LN) {
    // complete rows live in this workgroup (N <= 256): two-pass LayerNorm over the 8 waves
    const float inv_n = 1.f / (float)N;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float part[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float t = v[r];
        if (pass == 1) {
          const float d = col_ok ? v[r] - stat[0][(r & 3) + 8 * (r >> 2) + 4 * hi5] : 0.f;
          t = d * d;
        }
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o);     // over the 32 columns of this half-wave
        part[r] = t;
      }
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * hi5] = part[r];
      }
      __syncthreads();
      if (tid < 32) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w][tid];
        stat[pass][tid] = pass == 0 ? t * inv_n : rsqrtf(t * inv_n + eps);
      }
      __syncthreads();
    }
    if (col_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi5;
        v[r] = (v[r] - stat[0][row]) * stat[1][row] * lg + lb;
      }
    }
  }
  if (col_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (m < M) {
        y[(size_t)m * ldy + n - ncol0] = v[r];
        if (yp) yp[(size_t)m * ldyp + n] = v[r] + posv[r];
      }
    }
  }
}

// Pack W [N, K] f32 -> bf16 B fragments [ceil(N/32)][K/16][64][8]; rows >= N are zero.
__global__ __launch_bounds__(256) void cgg_lr2_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int N,
                                                          int K, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long long t = i >> 6;
  const int KS = K >> 4;
  const int ks = (int)(t % KS), nt = (int)(t / KS);
  const int n = nt * 32 + (lane & 31);
  const int k = ks * 16 + 8 * (lane >> 5);
  u32x4 p = {0u, 0u, 0u, 0u};
  if (n < N) {
    const float* s = w + (size_t)n * K + k;
    p = u32x4{cgg_pack2(cgg_f2bf(s[0]), cgg_f2bf(s[1])), cgg_pack2(cgg_f2bf(s[2]), cgg_f2bf(s[3])),
              cgg_pack2(cgg_f2bf(s[4]), cgg_f2bf(s[5])), cgg_pack2(cgg_f2bf(s[6]), cgg_f2bf(s[7]))};
  }
  out[i] = p;
}

// Row LayerNorm chain after the split-K FFN projection: y = LN_a(a); yp = y + pos[row % pos_rows];
// z = LN_b(y) (the decoder's post_norm, mask2former_head.py:734). One wave per row, N <= 1024, N % 4 == 0.
__global__ __launch_bounds__(256) void cgg_ln_chain_kernel(const float* __restrict__ a, int lda,
                                                          const float* __restrict__ ga, const float* __restrict__ ba,
                                                          float eps_a, const float* __restrict__ pos, int pos_rows,
                                                          const float* __restrict__ gb, const float* __restrict__ bb,
                                                          float eps_b, float* __restrict__ y, float* __restrict__ yp,
                                                          float* __restrict__ z, int rows, int N, int nsum,
                                                          long long plane) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int nv = N >> 2;                       // float4 per row, <= 256 -> <= 4 per lane
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < nv) {
      for (int p = 0; p < nsum; ++p)             // split-K planes, fixed order
        v[k] += *reinterpret_cast<const f32x4*>(a + (size_t)p * plane + (size_t)row * lda + 4 * i);
    }
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float inv_n = 1.f / (float)N;
#pragma unroll
  for (int stage = 0; stage < 2; ++stage) {
    if (stage == 1) {
      if (z == nullptr) break;
      s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (lane + 64 * k < nv) {
        const f32x4 d = v[k] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_n + (stage == 0 ? eps_a : eps_b));
    const float* g = stage == 0 ? ga : gb;
    const float* bt = stage == 0 ? ba : bb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = lane + 64 * k;
      if (i < nv) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i), bv = *reinterpret_cast<const f32x4*>(bt + 4 * i);
        v[k] = (v[k] - mean) * rstd * gg + bv;
        float* dst = (stage == 0 ? y : z) + (size_t)row * N + 4 * i;
        *reinterpret_cast<f32x4*>(dst) = v[k];
        if (stage == 0 && yp) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + (size_t)(row % pos_rows) * N + 4 * i);
          *reinterpret_cast<f32x4*>(yp + (size_t)row * N + 4 * i) = v[k] + pv;
        }
      }
    }
  }
}

extern "C" int cgg_layernorm_chain(const float* a, int lda, const float* gamma_a, const float* beta_a, float eps_a,
                                   const float* pos, int pos_rows, const float* gamma_b, const float* beta_b,
                                   float eps_b, float* y, float* yp, float* z, int rows, int N, int nsum,
                                   int64_t plane, cgg_stream_t stream) {
  CGG_REQUIRE(a && gamma_a && beta_a && y, CGG_EINVAL, "cgg_layernorm_chain: null pointer");
  CGG_REQUIRE(rows > 0 && N > 0 && N % 4 == 0 && N <= 1024 && lda % 4 == 0, CGG_EUNSUPPORTED,
              "cgg_layernorm_chain: N=%d lda=%d", N, lda);
  CGG_REQUIRE(!yp || (pos && pos_rows > 0), CGG_EINVAL, "cgg_layernorm_chain: yp needs pos");
  CGG_REQUIRE(!z || (gamma_b && beta_b), CGG_EINVAL, "cgg_layernorm_chain: z needs the second LayerNorm");
  CGG_REQUIRE(cgg_aligned16(a) && cgg_aligned16(y) && cgg_aligned16(yp) && cgg_aligned16(z) && cgg_aligned16(pos) &&
                  cgg_aligned16(gamma_a) && cgg_aligned16(beta_a) && cgg_aligned16(gamma_b) && cgg_aligned16(beta_b),
              CGG_EALIGN, "cgg_layernorm_chain: alignment");
  hipLaunchKernelGGL(cgg_ln_chain_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, lda, gamma_a,
                     beta_a, eps_a, pos, pos_rows, gamma_b, beta_b, eps_b, y, yp, z, rows, N, nsum < 1 ? 1 : nsum,
                     (long long)plane);
  CGG_CHECK_LAUNCH("cgg_layernorm_chain");
  return CGG_OK;
}

extern "C" int64_t cgg_linear_rows_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 16) return 0;
  return (int64_t)((N + 31) / 32) * (K / 16) * 64 * 16;
}

extern "C" int cgg_linear_rows_pack(const float* w, void* packed, int N, int K, cgg_stream_t stream) {
  CGG_REQUIRE(w && packed, CGG_EINVAL, "cgg_linear_rows_pack: null pointer");
  CGG_REQUIRE(N > 0 && K > 0 && K % 16 == 0, CGG_EUNSUPPORTED, "cgg_linear_rows_pack: N=%d K=%d (K %% 16)", N, K);
  const long long total = (long long)((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(cgg_lr2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (u32x4*)packed, N, K, total);
  CGG_CHECK_LAUNCH("cgg_linear_rows_pack");
  return CGG_OK;
}

static int lr2_launch(bool x3, const char* who, const float* x, int ldx, const void* w_packed, const float* bias,
                      const float* res, int ldr, float* y, int ldy, const float* ln_gamma,
                      const float* ln_beta, float ln_eps, const float* pos, int pos_rows, float* yp,
                      int ldyp, int M, int N, int K, int relu_cols, int ksplit, const float* x2, int ldx2,
                      int x2_col, float* y2, int ldy2, int y2_col, cgg_stream_t stream) {
  CGG_REQUIRE(x && w_packed && y, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(!x2 || (x2_col > 0 && x2_col % 256 == 0 && ldx2 % 4 == 0 && cgg_aligned16(x2)), CGG_EUNSUPPORTED,
              "%s: x2_col=%d must be a positive multiple of 256 (ldx2=%d)", who, x2_col, ldx2);
  CGG_REQUIRE(!y2 || (y2_col > 0 && y2_col % 256 == 0), CGG_EUNSUPPORTED,
              "%s: y2_col=%d must be a positive multiple of 256", who, y2_col);
  CGG_REQUIRE(M > 0 && N > 0 && K > 0, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(K % 16 == 0 && ldx % 4 == 0, CGG_EUNSUPPORTED, "%s: K=%d ldx=%d", who, K, ldx);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(w_packed), CGG_EALIGN, "%s: alignment", who);
  if (ksplit < 1) ksplit = 1;
  const bool ln = ln_gamma != nullptr;
  CGG_REQUIRE(!ln || (ln_beta && N <= 256 && ksplit == 1), CGG_EUNSUPPORTED,
              "%s: the LayerNorm epilogue needs N <= 256 and no K split", who);
  CGG_REQUIRE(!yp || (pos && pos_rows > 0 && ksplit == 1), CGG_EINVAL, "%s: yp needs pos, no split", who);
  CGG_REQUIRE(ksplit == 1 || relu_cols == 0, CGG_EUNSUPPORTED, "%s: no ReLU with a K split", who);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((N + 255) / 256, (M + 31) / 32, ksplit);
  const CggX3W w = x3 ? cgg_x3_view(w_packed, N, K) : CggX3W{(const cgg_u32x4*)w_packed, nullptr, nullptr};
#define LR2_LAUNCH(LN, X3)                                                                                                  \
  hipLaunchKernelGGL((cgg_lr2_kernel<LN, X3>), grid, dim3(512), 0, s, x, ldx, (const u32x4*)w.hi, (const u32x4*)w.lo, w.scale, \
                     bias, res, ldr, y, ldy, ln_gamma, ln_beta, ln_eps, pos, pos_rows, yp, ldyp, M, N, K, relu_cols, x2, ldx2,  \
                     x2_col, y2, ldy2, y2_col)
  if (ln && x3) LR2_LAUNCH(true, true);
  else if (ln) LR2_LAUNCH(true, false);
  else if (x3) LR2_LAUNCH(false, true);
  else LR2_LAUNCH(false, false);
#undef LR2_LAUNCH
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_linear_rows_bf16(const float* x, int ldx, const void* w_packed, const float* bias,
                                    const float* res, int ldr, float* y, int ldy, const float* ln_gamma,
                                    const float* ln_beta, float ln_eps, const float* pos, int pos_rows, float* yp,
                                    int ldyp, int M, int N, int K, int relu_cols, int ksplit, const float* x2, int ldx2,
                                    int x2_col, float* y2, int ldy2, int y2_col, cgg_stream_t stream) {
  return lr2_launch(false, "cgg_linear_rows_bf16", x, ldx, w_packed, bias, res, ldr, y, ldy, ln_gamma, ln_beta, ln_eps, pos,
                    pos_rows, yp, ldyp, M, N, K, relu_cols, ksplit, x2, ldx2, x2_col, y2, ldy2, y2_col, stream);
}

extern "C" int cgg_linear_rows_x3(const float* x, int ldx, const void* w_x3, const float* bias,
                                  const float* res, int ldr, float* y, int ldy, const float* ln_gamma,
                                  const float* ln_beta, float ln_eps, const float* pos, int pos_rows, float* yp,
                                  int ldyp, int M, int N, int K, int relu_cols, int ksplit, const float* x2, int ldx2,
                                  int x2_col, float* y2, int ldy2, int y2_col, cgg_stream_t stream) {
  return lr2_launch(true, "cgg_linear_rows_x3", x, ldx, w_x3, bias, res, ldr, y, ldy, ln_gamma, ln_beta, ln_eps, pos,
                    pos_rows, yp, ldyp, M, N, K, relu_cols, ksplit, x2, ldx2, x2_col, y2, ldy2, y2_col, stream);
}

// ---- x3 image of a weight (x3.h): per-row power-of-two scale, then hi / lo f16 B fragments ----
__global__ __launch_bounds__(256) void cgg_x3_rowscale_kernel(const float* __restrict__ w, float* __restrict__ colscale, int N,
                                                             int K, int NP) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= NP) return;
  float m = 0.f;
  if (n < N)
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(w[(size_t)n * K + k]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) {
    float cs = 0.f;
    if (n < N) {
      int e = 0;
      if (m > 0.f && m < 3.0e38f) frexpf(m, &e);          // m = f 2^e, f in [0.5, 1): w' = w 2^(11 - e) has max in [2^10, 2^11)
      cs = ldexpf(CGG_X3_INV_ASCALE, e - 11);              // un-scales the accumulator: 2^(e - 11) / ASCALE
    }
    colscale[n] = cs;
  }
}

__global__ __launch_bounds__(256) void cgg_x3_pack_kernel(const float* __restrict__ w, const float* __restrict__ colscale,
                                                         u32x4* __restrict__ hi, u32x4* __restrict__ lo, int N, int K,
                                                         long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long long t = i >> 6;
  const int KS = K >> 4;
  const int ks = (int)(t % KS), nt = (int)(t / KS);
  const int n = nt * 32 + (lane & 31);
  const int k = ks * 16 + 8 * (lane >> 5);
  u32x4 ph = {0u, 0u, 0u, 0u}, pl = ph;
  if (n < N) {
    const float ws = CGG_X3_INV_ASCALE / colscale[n];      // = 2^(11 - e), exact
    const float* s = w + (size_t)n * K + k;
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) cgg_x3_split2(s[2 * e] * ws, s[2 * e + 1] * ws, h[e], l[e]);
    ph = u32x4{h[0], h[1], h[2], h[3]};
    pl = u32x4{l[0], l[1], l[2], l[3]};
  }
  hi[i] = ph;
  lo[i] = pl;
}

extern "C" int64_t cgg_x3_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 16) return 0;
  const int64_t nt = (N + 31) / 32;
  return 2 * nt * (K / 16) * 64 * 16 + nt * 32 * 4;
}

extern "C" int cgg_x3_pack(const float* w, void* packed, int N, int K, cgg_stream_t stream) {
  CGG_REQUIRE(w && packed, CGG_EINVAL, "cgg_x3_pack: null pointer");
  CGG_REQUIRE(N > 0 && K > 0 && K % 16 == 0, CGG_EUNSUPPORTED, "cgg_x3_pack: N=%d K=%d (K %% 16)", N, K);
  CGG_REQUIRE(cgg_aligned16(packed), CGG_EALIGN, "cgg_x3_pack: alignment");
  const int NP = (N + 31) / 32 * 32;
  const long long total = (long long)(NP / 32) * (K / 16) * 64;
  u32x4* hi = (u32x4*)packed;
  u32x4* lo = hi + total;
  float* cs = (float*)(hi + 2 * total);
  hipLaunchKernelGGL(cgg_x3_rowscale_kernel, dim3((NP + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, cs, N, K, NP);
  hipLaunchKernelGGL(cgg_x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (const float*)cs, hi, lo, N, K, total);
  CGG_CHECK_LAUNCH("cgg_x3_pack");
  return CGG_OK;
}
