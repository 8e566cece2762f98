// Round 4: the post-attention half of an MSDeformAttn encoder layer (see encoder_tail_x3.hip for the operator and its reference,
// [3P] BaseTransformerLayer ('self_attn','norm','ffn','norm') built at open_set/models/mask2former_head.py:112-117) as a REGISTER-
// CHAINED kernel on x3a rows:
//
//     x1 = LayerNorm0( x + a Wo^T + bo )                      a = attention rows (f32), x = layer input rows (x3a)
//     y  = LayerNorm1( x1 + W2 relu(W1 x1 + b1) + b2 )        y, yp = y + pos[row % pos_rows] as x3a rows
//
// The first kernel (encoder_tail_x3.hip) keeps 64 rows per workgroup in LDS as fragment images and streams every weight fragment
// L2 -> registers per wave: 2.36 MB of L2 reads per 64 rows, and between the GEMMs the block goes registers -> LDS -> registers
// (row staging, two LayerNorm tiles, the hidden block, 10 barriers) with one wave per SIMD, so nothing overlaps: 26 us of MFMA work
// in a 67-us workgroup. Here every GEMM is computed TRANSPOSED, D = W A^T, so that the accumulator layout of one GEMM (lane = row,
// registers = output channel) IS the B-operand layout of the next one (lane = row, registers = k):
//
//   * a wavefront owns 32 rows for the whole chain; the activations never leave its registers: a Wo^T (8 accumulator tiles =
//     256 channels x 32 rows) -> LayerNorm0 in registers (per-lane sums + one cross-half exchange) -> x1 as 16 hi / lo B fragments
//     -> per 32 hidden units: W1 tile (one accumulator) -> relu -> two B fragments -> W2 into the 8 output accumulator tiles,
//     which were initialised with (x1 + b2) / colscale (exact: colscale is a power of two) -> LayerNorm1 in registers -> stores.
//     The k index inside a 32-block follows the accumulator's register order, so W1 and W2 are packed with their K columns
//     permuted (t2_perm32, ops.py `_tail_v2_weights`); Wo's B operand comes from memory in natural order.
//   * the weights are the only LDS traffic: ONE stream of 144 stages x 16 KiB (8 hi / lo fragment pairs = 24 MFMAs per wave) goes
//     L2 -> LDS by `buffer_load ... lds` into an 8-slot ring shared by the workgroup's 4 waves (2.36 MB per 128 rows: a quarter of
//     the L2 traffic per row), one counted `s_waitcnt vmcnt` + one barrier per stage.
//   * no row staging, no LayerNorm tile, no hidden image: LDS = ring (128 KiB) + the per-channel tables (17 KiB).
//
// v_mfma_f32_32x32x16_f16, x3 arithmetic (x3.h), f32 accumulation. build-flags: (none: accumulators may live in AGPRs)
//
// MEASURED (MI355X, configs[1]: 43 008 rows, F = 1024; profiles/r4_tail_v2_ablation.txt): 274 us against 198 us for the LDS-image
// kernel -- correct (tests/test_x3s_gpu.py) but NOT the default (CGG_TAIL_V2=1 selects it). With one 512-register wave per SIMD
// nothing overlaps the MFMAs: by elimination builds the 2 x 137 us (1344 row units on 1024 SIMDs = two rounds) are ~80 us of MFMAs,
// ~45 us of LDS-DMA issue (576 pieces per wave at ~94 cycles each), ~57 us of the lane-per-row loads / stores (each instruction
// touches 32 rows), ~14 us of barriers and ~80 us of VALU (LayerNorms, splits, AGPR moves), table fill and prologue latency; the
// fragment reads themselves are free (ds_read_b128 at 256 B / clk).
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t t2_u32x4;
typedef __attribute__((address_space(3))) void* t2_lds_t;

#define T2_C 256
#define T2_NS 8                 // ring slots
#define T2_SLOT 16384           // bytes per slot: 8 (hi, lo) fragment pairs
#define T2_NT 256               // 4 waves x 32 rows
#define T2_P1 16                // stages of the output projection (one per k-step)

struct T2Args {
  const float* a32;
  const void* x;
  CggX3W wo, w1, w2;            // w1 / w2: K-permuted images
  uint32_t wo_bytes, w1_bytes, w2_bytes;
  const float *bo, *g0, *be0, *b1, *b2, *g1, *be1, *pos;
  float eps0, eps1;
  int pos_rows;
  void *y, *yp;
  int M, F;
  int* flag;
};

template <int N>
__device__ __forceinline__ void t2_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void t2_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ float t2_xhalf(float v) {      // the value of lane ^ 32
  return __shfl_xor(v, 32);
}

#define T2_MF(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

__global__ __launch_bounds__(T2_NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void cgg_encoder_tail_x3v2_kernel(const T2Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char t2_smem[];
  float* tab = reinterpret_cast<float*>(t2_smem + T2_NS * T2_SLOT);
  // tables (floats): cso | bo | g0 | be0 | 1/cs2 | b2 | cs2 | g1 | be1 | 16 cs1 [F] | 16 b1 [F]
  float* t_cso = tab;
  float* t_bo = tab + 256;
  float* t_g0 = tab + 512;
  float* t_be0 = tab + 768;
  float* t_ics2 = tab + 1024;
  float* t_b2 = tab + 1280;
  float* t_cs2 = tab + 1536;
  float* t_g1 = tab + 1792;
  float* t_be1 = tab + 2048;
  float* t_cs1 = tab + 2304;
  float* t_b1 = tab + 2304 + p.F;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, u = lane >> 5;
  const int m = (blockIdx.x * 4 + wave) * 32 + j;
  const int mc = m < p.M ? m : p.M - 1;
  const int F = p.F;
  const int KS2 = F >> 4;
  const int nstage = T2_P1 + (F >> 5) * 4;

  // ---- tables -> LDS (before anything is in flight: the compiler's own waits for these loads must not drain the DMA ring) ----
  {
    const int c = tid;       // 256 threads = 256 channels
    const float cs2 = p.w2.scale[c];
    t_cso[c] = p.wo.scale[c];
    t_bo[c] = p.bo[c];
    t_g0[c] = p.g0[c];
    t_be0[c] = p.be0[c];
    t_ics2[c] = 1.0f / cs2;
    t_b2[c] = p.b2[c];
    t_cs2[c] = cs2;
    t_g1[c] = p.g1[c];
    t_be1[c] = p.be1[c];
    for (int h = tid; h < F; h += T2_NT) {
      t_cs1[h] = p.w1.scale[h] * CGG_X3_ASCALE;
      t_b1[h] = p.b1[h] * CGG_X3_ASCALE;
    }
  }
  __syncthreads();

  // ---- the attention rows of this wave: B operand of the output projection, k-step ks = channels 16 ks + 8 u .. + 7 of row j ----
  f32x4 av[T2_P1][2];
  {
    const float* src = p.a32 + (size_t)mc * T2_C + 8 * u;
#pragma unroll
    for (int ks = 0; ks < T2_P1; ++ks) {
      av[ks][0] = *reinterpret_cast<const f32x4*>(src + 16 * ks);
      av[ks][1] = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4);
    }
  }

  // ---- the weight stream. Stage g: g < 16: Wo k-step g, pairs = output tiles 0..7; then per 32 hidden units T four stages:
  //      W1 tile T k-steps 0..7, k-steps 8..15 (pairs = k-steps), W2 k-step 2T, 2T + 1 (pairs = output tiles). Wave w fetches pairs
  //      2w, 2w + 1 (hi and lo: four 1-KiB pieces per stage). Stages past the end re-read the last one (uniform vmcnt arithmetic).
  const __amdgpu_buffer_rsrc_t r_wo = __builtin_amdgcn_make_buffer_rsrc(const_cast<t2_u32x4*>(p.wo.hi), 0, p.wo_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<t2_u32x4*>(p.w1.hi), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<t2_u32x4*>(p.w2.hi), 0, p.w2_bytes, 0x00020000);
  const uint32_t lo_wo = (uint32_t)((const unsigned char*)p.wo.lo - (const unsigned char*)p.wo.hi);
  const uint32_t lo_w1 = (uint32_t)((const unsigned char*)p.w1.lo - (const unsigned char*)p.w1.hi);
  const uint32_t lo_w2 = (uint32_t)((const unsigned char*)p.w2.lo - (const unsigned char*)p.w2.hi);
  const int voff = lane * 16;
  auto issue_at = [&](int g, int slot) {
    const int gc = g < nstage ? g : nstage - 1;
    unsigned char* sbase = t2_smem + slot * T2_SLOT;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pr = 2 * wave + q;            // pair index inside the stage
      t2_lds_t dh = (t2_lds_t)(sbase + (2 * pr) * 1024);
      t2_lds_t dl = (t2_lds_t)(sbase + (2 * pr + 1) * 1024);
      if (gc < T2_P1) {
        const uint32_t so = (uint32_t)((pr * 16 + gc) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wo, dh, 16, voff, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wo, dl, 16, voff, so + lo_wo, 0, 0);
      } else {
        const int gg = gc - T2_P1, T = gg >> 2, k = gg & 3;
        if (k < 2) {
          const uint32_t so = (uint32_t)((T * 16 + 8 * k + pr) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w1, dh, 16, voff, so, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w1, dl, 16, voff, so + lo_w1, 0, 0);
        } else {
          const uint32_t so = (uint32_t)((pr * KS2 + 2 * T + (k - 2)) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w2, dh, 16, voff, so, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w2, dl, 16, voff, so + lo_w2, 0, 0);
        }
      }
    }
  };
  // stage top: this wave's pieces of stage g have landed (the 6 younger stages stay in flight), barrier = everyone's pieces landed
  // and everyone is done with stage g - 1, whose slot the DMA of stage g + 7 refills
  auto stage_top = [&](int g) {
    t2_wait_vmcnt<(T2_NS - 2) * 4>();
    t2_barrier();
    issue_at(g + T2_NS - 1, (g + T2_NS - 1) & (T2_NS - 1));
  };
#pragma unroll
  for (int g = 0; g < T2_NS - 1; ++g) issue_at(g, g);

  const unsigned char* fbase = t2_smem + lane * 16;
  auto frag = [&](int slot, int piece) -> t2_u32x4 {
    return *reinterpret_cast<const t2_u32x4*>(fbase + slot * T2_SLOT + piece * 1024);
  };

  // ---- output projection: acc1[t] (32 channels x 32 rows) += Wo tile t . a^T over 16 k-steps ----
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // the layer-input rows (LayerNorm0's residual), x3a: channels 32 t + 8 q + 4 u .. + 3 = half u of the group [8 hi | 8 lo]
  uint2 xh[32], xl[32];
#pragma unroll
  for (int g = 0; g < T2_P1; ++g) {
    stage_top(g);
    if (g == 8 || g == 12) {
      const uint2* xs = reinterpret_cast<const uint2*>((const float*)p.x + (size_t)mc * T2_C) + u;
      const int o = g == 8 ? 0 : 16;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        xh[o + q] = xs[4 * (o + q)];
        xl[o + q] = xs[4 * (o + q) + 2];
      }
    }
    t2_u32x4 bh, bl;
    cgg_x3_split8(av[g][0], av[g][1], bh, bl);
    const int slot = g & (T2_NS - 1);
    t2_u32x4 ah[2], al[2];
    ah[0] = frag(slot, 0);
    al[0] = frag(slot, 1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t + 1 < 8) {
        ah[(t + 1) & 1] = frag(slot, 2 * t + 2);
        al[(t + 1) & 1] = frag(slot, 2 * t + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      T2_MF(al[t & 1], bh, acc[t]);
      T2_MF(ah[t & 1], bl, acc[t]);
      T2_MF(ah[t & 1], bh, acc[t]);
    }
  }

  // ---- LayerNorm0 in registers: v = acc cs + bo + x; lane (j, u) holds channels 32 t + 8 q + 4 u + c of row j ----
  t2_u32x4 x1h[16], x1l[16];          // x1 as B fragments: k-step 2 t + s = registers 8 s .. 8 s + 7 of tile t
  {
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * t + 8 * q + 4 * u;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(t_cso + ch), bo = *reinterpret_cast<const f32x4*>(t_bo + ch);
        f32x4 xr;
        cgg_x3a_decode4(xh[4 * t + q], xl[4 * t + q], xr);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = acc[t][4 * q + c] * cs[c] + bo[c] + xr[c];
          acc[t][4 * q + c] = v;
          sm += v;
        }
      }
    sm += t2_xhalf(sm);
    const float mean = sm * (1.f / T2_C);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[t][r] - mean;
        acc[t][r] = d;
        sq += d * d;
      }
    sq += t2_xhalf(sq);
    const float rstd = rsqrtf(sq * (1.f / T2_C) + p.eps0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float s16[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * t + 8 * q + 4 * u;
        const f32x4 g = *reinterpret_cast<const f32x4*>(t_g0 + ch), be = *reinterpret_cast<const f32x4*>(t_be0 + ch);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(t_b2 + ch), ic = *reinterpret_cast<const f32x4*>(t_ics2 + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float y = acc[t][4 * q + c] * rstd * g[c] + be[c];
          s16[4 * q + c] = y * CGG_X3_ASCALE;
          acc[t][4 * q + c] = (y + b2[c]) * ic[c];       // the FFN's accumulator starts at (x1 + b2) / colscale (exact scaling)
        }
      }
      cgg_x3a_split8_prescaled(s16, x1h[2 * t], x1l[2 * t]);
      cgg_x3a_split8_prescaled(s16 + 8, x1h[2 * t + 1], x1l[2 * t + 1]);
    }
  }

  // ---- FFN: per 32 hidden units T: h^T = W1[T] x1^T (two accumulators over even / odd k-steps), relu, two B fragments,
  //      y^T += W2[:, T] h^T into the 8 output tiles ----
  float hmax = 0.f;
  const int NTL = F >> 5;
  for (int T = 0; T < NTL; ++T) {
    const int g0 = T2_P1 + 4 * T;
    f32x16 h0, h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h0[r] = h1[r] = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      stage_top(g0 + k);
      const int slot = (g0 + k) & (T2_NS - 1);
      t2_u32x4 ah[2], al[2];
      ah[0] = frag(slot, 0);
      al[0] = frag(slot, 1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q + 1 < 8) {
          ah[(q + 1) & 1] = frag(slot, 2 * q + 2);
          al[(q + 1) & 1] = frag(slot, 2 * q + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int ks = 8 * k + q;
        if (q & 1) {
          T2_MF(al[q & 1], x1h[ks], h1);
          T2_MF(ah[q & 1], x1l[ks], h1);
          T2_MF(ah[q & 1], x1h[ks], h1);
        } else {
          T2_MF(al[q & 1], x1h[ks], h0);
          T2_MF(ah[q & 1], x1l[ks], h0);
          T2_MF(ah[q & 1], x1h[ks], h0);
        }
      }
    }
    t2_u32x4 hh[2], hl[2];
    {
      float s16[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hid = 32 * T + 8 * q + 4 * u;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(t_cs1 + hid), b1 = *reinterpret_cast<const f32x4*>(t_b1 + hid);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = fmaxf((h0[4 * q + c] + h1[4 * q + c]) * cs[c] + b1[c], 0.f);      // 16 relu(.): tables are pre-scaled
          s16[4 * q + c] = v;
          hmax = fmaxf(hmax, v);
        }
      }
      cgg_x3a_split8_prescaled(s16, hh[0], hl[0]);
      cgg_x3a_split8_prescaled(s16 + 8, hh[1], hl[1]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      stage_top(g0 + 2 + s);
      const int slot = (g0 + 2 + s) & (T2_NS - 1);
      t2_u32x4 ah[2], al[2];
      ah[0] = frag(slot, 0);
      al[0] = frag(slot, 1);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t + 1 < 8) {
          ah[(t + 1) & 1] = frag(slot, 2 * t + 2);
          al[(t + 1) & 1] = frag(slot, 2 * t + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        T2_MF(al[t & 1], hh[s], acc[t]);
        T2_MF(ah[t & 1], hl[s], acc[t]);
        T2_MF(ah[t & 1], hh[s], acc[t]);
      }
    }
  }
  t2_wait_vmcnt<0>();                  // the run-ahead pieces must land before the workgroup's LDS is released
  if (p.flag && !(hmax <= CGG_X3A_MAX)) atomicOr(p.flag, 1);

  // ---- LayerNorm1 in registers, outputs as x3a rows ----
  {
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(t_cs2 + 32 * t + 8 * q + 4 * u);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = acc[t][4 * q + c] * cs[c];
          acc[t][4 * q + c] = v;
          sm += v;
        }
      }
    sm += t2_xhalf(sm);
    const float mean = sm * (1.f / T2_C);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[t][r] - mean;
        acc[t][r] = d;
        sq += d * d;
      }
    sq += t2_xhalf(sq);
    const float rstd = rsqrtf(sq * (1.f / T2_C) + p.eps1);
    const bool live = m < p.M;
    const float* prow = p.yp ? p.pos + (size_t)(mc % p.pos_rows) * T2_C + 4 * u : nullptr;
    uint2* yo = reinterpret_cast<uint2*>((float*)p.y + (size_t)mc * T2_C) + u;
    uint2* po = p.yp ? reinterpret_cast<uint2*>((float*)p.yp + (size_t)mc * T2_C) + u : nullptr;
    float am = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * t + 8 * q + 4 * u;
        const f32x4 g = *reinterpret_cast<const f32x4*>(t_g1 + ch), be = *reinterpret_cast<const f32x4*>(t_be1 + ch);
        f32x4 y;
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = acc[t][4 * q + c] * rstd * g[c] + be[c];
        uint2 h, l;
        cgg_x3_split4(y, h, l);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
        const int o = 4 * (4 * t + q);                 // uint2 index of the group's hi half (minus u)
        if (live) {
          yo[o] = h;
          yo[o + 2] = l;
        }
        if (po) {
          const f32x4 yp = y + *reinterpret_cast<const f32x4*>(prow + 32 * t + 8 * q);
          cgg_x3_split4(yp, h, l);
          am = fmaxf(am, fmaxf(fmaxf(fabsf(yp[0]), fabsf(yp[1])), fmaxf(fabsf(yp[2]), fabsf(yp[3]))));
          if (live) {
            po[o] = h;
            po[o + 2] = l;
          }
        }
      }
    if (p.flag && !(am * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(p.flag, 1);
  }
}

int* cgg_x3_overflow_flag_ptr();       // x3s_gemm.hip

// Within every 32-block of K the packed position p = 16 s + 8 u + i holds the original column (2 s + i / 4) * 8 + 4 u + i % 4: the
// order in which an accumulator tile's registers (r = 8 s + i of lane half u) enumerate its 32 rows.
extern "C" int cgg_encoder_tail_v2_perm32(int32_t* perm32) {
  CGG_REQUIRE(perm32, CGG_EINVAL, "cgg_encoder_tail_v2_perm32: null pointer");
  for (int pp = 0; pp < 32; ++pp) {
    const int s = pp >> 4, u = (pp >> 3) & 1, i = pp & 7;
    perm32[pp] = (2 * s + (i >> 2)) * 8 + 4 * u + (i & 3);
  }
  return CGG_OK;
}

// a32 f32 attention rows; x, y, yp x3a rows; wo_x3 = x3 image of Wo; w1p_x3 / w2p_x3 = x3 images of W1[:, perm] / W2[:, perm]
// (perm = cgg_encoder_tail_v2_perm32 applied inside every 32-block of the K axis)
extern "C" int cgg_encoder_layer_tail_x3a_v2(const float* a32, const void* x_x3a, const void* wo_x3, const float* bo,
                                             const float* gamma0, const float* beta0, float eps0, const void* w1p_x3,
                                             const float* b1, const void* w2p_x3, const float* b2, const float* gamma1,
                                             const float* beta1, float eps1, const float* pos, int pos_rows, void* y_x3a,
                                             void* yp_x3a, int M, int C, int F, cgg_stream_t stream) {
  const char* who = "cgg_encoder_layer_tail_x3a_v2";
  CGG_REQUIRE(a32 && x_x3a && wo_x3 && bo && gamma0 && beta0 && w1p_x3 && b1 && w2p_x3 && b2 && gamma1 && beta1 && y_x3a, CGG_EINVAL,
              "%s: null pointer", who);
  CGG_REQUIRE(C == T2_C, CGG_EUNSUPPORTED, "%s: C=%d (only 256 is built)", who, C);
  CGG_REQUIRE(M > 0 && F > 0 && F % 32 == 0 && F <= 2048, CGG_EUNSUPPORTED, "%s: F=%d must be a multiple of 32, <= 2048", who, F);
  CGG_REQUIRE(!yp_x3a || (pos && pos_rows > 0), CGG_EINVAL, "%s: yp needs pos", who);
  CGG_REQUIRE(cgg_aligned16(a32) && cgg_aligned16(x_x3a) && cgg_aligned16(wo_x3) && cgg_aligned16(w1p_x3) && cgg_aligned16(w2p_x3) &&
                  (!pos || cgg_aligned16(pos)) && cgg_aligned16(y_x3a) && (!yp_x3a || cgg_aligned16(yp_x3a)),
              CGG_EALIGN, "%s: 16-B alignment", who);
  T2Args p;
  p.a32 = a32;
  p.x = x_x3a;
  p.wo = cgg_x3_view(wo_x3, T2_C, T2_C);
  p.w1 = cgg_x3_view(w1p_x3, F, T2_C);
  p.w2 = cgg_x3_view(w2p_x3, T2_C, F);
  p.wo_bytes = (uint32_t)((size_t)2 * (T2_C / 32) * (T2_C / 16) * 1024);
  p.w1_bytes = (uint32_t)((size_t)2 * (F / 32) * (T2_C / 16) * 1024);
  p.w2_bytes = (uint32_t)((size_t)2 * (T2_C / 32) * (F / 16) * 1024);
  p.bo = bo; p.g0 = gamma0; p.be0 = beta0; p.b1 = b1; p.b2 = b2; p.g1 = gamma1; p.be1 = beta1; p.pos = pos;
  p.eps0 = eps0; p.eps1 = eps1; p.pos_rows = pos_rows; p.y = y_x3a; p.yp = yp_x3a; p.M = M; p.F = F;
  p.flag = cgg_x3_overflow_flag_ptr();
  const size_t lds = (size_t)T2_NS * T2_SLOT + (size_t)(2304 + 2 * F) * sizeof(float);
  const size_t lds_max = (size_t)T2_NS * T2_SLOT + (size_t)(2304 + 2 * 2048) * sizeof(float);
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)cgg_encoder_tail_x3v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    CGG_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise dynamic LDS to %zu", who, lds_max);
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(cgg_encoder_tail_x3v2_kernel, dim3((M + 127) / 128), dim3(T2_NT), lds, (hipStream_t)stream, p);
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}
