// K6 (+K5 contract): masked multi-head cross-attention core, flash-style with split over keys.
// Replaces the baddbmm + masked softmax + bmm inside nn.MultiheadAttention reached from
// open_set/models/mask2former_head.py:829-840 (attn_masks=[attn_mask, None]).
//
// Shape regime (Q=100 queries, D=32, S = 1024 / 4096 / 16384 keys, B*H = 16): K/V are streamed
// exactly once (HBM/L2-bound), so the work is split over (batch, head, key chunk) to fill 256 CUs
// and recombined by a tiny second kernel. The boolean mask is a bit-packed [B, Q, S/32] image shared
// by all heads (never the x8 head repeat of mask2former_head.py:756-757).
//
// Wave mapping ("swapped" QK^T): S^T = K Q^T with keys on MFMA rows and queries on MFMA columns,
// so a lane owns ONE query and 16 keys of the tile -> the row max / row sum of the online softmax
// are in-register reductions + one cross-half shuffle; P^T feeds the PV MFMA as B operand with no
// data movement at all.
//   f32 path:  v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate) -- parity mode.
//   bf16 path: v_mfma_f32_32x32x16_bf16 on bf16 K / V^T (see cgg_masked_xattn_forward_bf16).
#include "cgg_common.h"

#define XA_TK 64  // keys per LDS tile

__device__ __forceinline__ int xa_kswz(int key, int slot) { return slot ^ ((key >> 1) & 7); }

// -------------------------------------------------------------------------------------------------
// partial pass: one workgroup = (key chunk, head, batch); wave w owns queries [32w, 32w+32).
// ws_o  [B, H, nchunks, Q, D]  un-normalised O;  ws_ml [B, H, nchunks, Q, 2]  (m, l)
// -------------------------------------------------------------------------------------------------
// BF (throughput-mode TRAINING): the same kernel with bf16 MFMA operands -- every group of four v_mfma_f32_32x32x2_f32 steps (a lane's
// four consecutive k values) becomes ONE v_mfma_f32_32x32x8_bf16 on the converted 4-vectors; accumulators, layouts, softmax and
// the f32 K / V rows in memory are unchanged. 1/8 of the MFMA time: the f32 pipe was this kernel's bound (0.4 of its 157-TF peak).
typedef __attribute__((ext_vector_type(4))) short xa_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 xa_bf2;
typedef __attribute__((ext_vector_type(2))) float xa_f2;
__device__ __forceinline__ xa_s16x4 xa_cvt4(float a, float b, float c, float d) {
  const xa_f2 lo = {a, b}, hi = {c, d};
  const uint2 u = {__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, xa_bf2)),
                   __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, xa_bf2))};
  return __builtin_bit_cast(xa_s16x4, u);
}

template <bool BF>
__global__ __launch_bounds__(256) void cgg_xattn_partial_f32(
    const float* __restrict__ q, const float* __restrict__ kv, const uint32_t* __restrict__ bits,
    float* __restrict__ ws_o, float* __restrict__ ws_ml, int Q, int H, int S, int words, int KC,
    int nchunks, float scale, float* __restrict__ out_direct, float* __restrict__ lse, int ldkv, long long kv_bstride) {
  constexpr int D = 32;
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int HD = H * D;
  const int s_begin = chunk * KC;
  const int s_end = min(S, s_begin + KC);
  const int cw = KC / 32;       // mask words per row in this chunk
  const int cws = cw + 1;       // padded LDS stride
  const int nmt = (Q + 31) / 32;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Ks = reinterpret_cast<float*>(smem_raw);             // [64][32] swizzled 16-B slots
  float* Vs = Ks + XA_TK * D;                                 // [64][32]
  uint32_t* Ms = reinterpret_cast<uint32_t*>(Vs + XA_TK * D); // [nmt*32][cws]

  // ---- mask words of this chunk -> LDS ----
  for (int i = tid; i < nmt * 32 * cw; i += 256) {
    const int qq = i / cw, w = i - qq * cw;
    const int gw = s_begin / 32 + w;
    uint32_t m = 0u;
    if (bits != nullptr && qq < Q && gw < words) m = bits[((size_t)b * Q + qq) * words + gw];
    Ms[qq * cws + w] = m;
  }

  // ---- this wave's queries (pre-scaled), B operand of S^T: lane (j, hi) holds d = 16*hi + s ----
  const int qi = wave * 32 + j;
  const bool wave_live = wave < nmt;
  float qf[16];
  {
    const bool ok = qi < Q;
    const float* qp = q + ((size_t)b * Q + (ok ? qi : 0)) * HD + h * D + 16 * hi;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = ok ? v[e] * scale : 0.f;
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;

  const float* kvb = kv + (size_t)b * kv_bstride + h * D;      // rows [K | V] at stride ldkv (a column slice of a merged projection)
  // K / V tile: 64 keys x 128 B each; thread = 16-B chunk (2 per operand). The loads of tile t + 1 are issued BEFORE the MFMAs of
  // tile t and held in registers (branch-free: keys past the chunk re-read its last key and are zeroed): the memory latency of a
  // tile -- paid in full, twice, by the predicated load -> LDS loop this replaces -- hides under ~4 000 cycles of f32 MFMAs
  f32x4 kx[2], vx[2];
  auto load_tile = [&](int s0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 256 * it;
      const int key = c >> 3, slot = c & 7;
      const int s = min(s0 + key, s_end - 1);
      const float* row = kvb + (size_t)s * ldkv + slot * 4;
      kx[it] = *reinterpret_cast<const f32x4*>(row);
      vx[it] = *reinterpret_cast<const f32x4*>(row + HD);
    }
  };
  load_tile(s_begin);
  for (int s0 = s_begin; s0 < s_end; s0 += XA_TK) {
    __syncthreads();  // previous tile fully consumed (also orders the Ms fill on first trip)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 256 * it;
      const int key = c >> 3, slot = c & 7;
      const bool live = s0 + key < s_end;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(Ks + key * D + xa_kswz(key, slot) * 4) = live ? kx[it] : z;
      *reinterpret_cast<f32x4*>(Vs + key * D + slot * 4) = live ? vx[it] : z;
    }
    __syncthreads();
    if (s0 + XA_TK < s_end) load_tile(s0 + XA_TK);       // workgroup-uniform
    if (!wave_live) continue;

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kt = sub * 32;  // key offset in tile
      // ---- S^T[key][query] : 16 x v_mfma_f32_32x32x2_f32 ----
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      const int krow = kt + j;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 ka =
            *reinterpret_cast<const f32x4*>(Ks + krow * D + xa_kswz(krow, hi * 4 + s4) * 4);
        if constexpr (BF) {
          sc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(xa_cvt4(ka[0], ka[1], ka[2], ka[3]),
                                                        xa_cvt4(qf[4 * s4], qf[4 * s4 + 1], qf[4 * s4 + 2], qf[4 * s4 + 3]), sc, 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[4 * s4 + e], sc, 0, 0, 0);
        }
      }
      // ---- mask + online softmax (lane = query j; 16 keys in-register, partner lane^32 has the rest)
      const uint32_t mw = Ms[qi * cws + ((s0 - s_begin + kt) >> 5)];
      float rmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ki = (r & 3) + 8 * (r >> 2) + 4 * hi;  // key inside the 32-key sub tile
        const bool blocked = ((mw >> ki) & 1u) || (s0 + kt + ki >= s_end);
        sc[r] = blocked ? -INFINITY : sc[r];
        rmax = fmaxf(rmax, sc[r]);
      }
      rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
      const float m_new = fmaxf(m_run, rmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = expf(m_run - m_use);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[r] = expf(sc[r] - m_use);
        psum += sc[r];
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
      // ---- O^T[d][query] += V^T[d][key] P^T[key][query] : 16 x v_mfma_f32_32x32x2_f32 ----
      if constexpr (BF) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k0 = kt + 8 * g + 4 * hi;            // the lane's four keys of k-step g: the rows its accumulators 4 g .. 4 g + 3 hold
          o = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(xa_cvt4(Vs[k0 * D + j], Vs[(k0 + 1) * D + j], Vs[(k0 + 2) * D + j], Vs[(k0 + 3) * D + j]),
                                                       xa_cvt4(sc[4 * g], sc[4 * g + 1], sc[4 * g + 2], sc[4 * g + 3]), o, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ki = kt + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float va = Vs[ki * D + j];
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(va, sc[r], o, 0, 0, 0);
        }
      }
    }
  }

  // ---- partials out: lane (j, hi) holds O[q = qi][d = (r&3) + 8*(r>>2) + 4*hi] ----
  if (wave_live && qi < Q && out_direct != nullptr) {
    // single chunk: the combine step degenerates to o / l (l == 0 -> NaN, as the reference's all-masked row)
    float* op = out_direct + ((size_t)b * Q + qi) * HD + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {o[4 * g] / l_run, o[4 * g + 1] / l_run, o[4 * g + 2] / l_run, o[4 * g + 3] / l_run};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * hi) = v;
    }
    if (lse != nullptr && hi == 0) lse[((size_t)b * H + h) * Q + qi] = m_run + logf(l_run);
  } else if (wave_live && qi < Q) {
    const size_t base = (((size_t)b * H + h) * nchunks + chunk) * Q + qi;
    float* op = ws_o + base * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * hi) = v;
    }
    if (hi == 0) {
      ws_ml[base * 2] = m_run;
      ws_ml[base * 2 + 1] = l_run;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// bf16 throughput path. K is [B, S, H*D] bf16 (key-major) and V arrives TRANSPOSED, vt [B, H*D, S] bf16
// (the value projection is computed as Wv x mem^T, so no kernel ever transposes anything):
//   S^T = K Q^T : A = K  -> lane (key j, 8 head-dims) is one 16-byte load straight from global,
//                 B = Q^T (bf16 of the pre-scaled f32 query, held in registers for the whole chunk)
//   O^T = V^T P^T: A = V^T -> lane (dim j, keys {0..3, 8..11} + 4*hi of the 16-key step) = two 8-byte loads
//                 of its vt row; B = P^T = the S^T accumulators rounded to bf16, no data movement.
// No LDS for K / V at all (only the mask words); the 4 waves of a workgroup (4 query tiles of one head) re-read
// the same 4 KiB of K / V per 32 keys from L1. Same partial / combine protocol as the f32 kernel.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void cgg_xattn_partial_bf16(
    const float* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ vt,
    const uint32_t* __restrict__ bits, float* __restrict__ ws_o, float* __restrict__ ws_ml, int Q, int H, int S,
    int words, int KC, int nchunks, float scale, float* __restrict__ out_direct, int ldk, long long vt_bstride,
    int auto_unmask) {
  constexpr int D = 32;
  // 8 waves = 2 ADJACENT heads x 4 query tiles: a 128-byte line of K holds the 64-byte slices of two heads, so
  // pairing them in one workgroup makes every fetched line fully useful (one head per workgroup re-fetched each
  // line from L2 for its neighbour: 43 -> 3x us at S = 16 384)
  const int chunk = blockIdx.x, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
  const int wave = wave8 & 3;
  const int h = blockIdx.y * 2 + (wave8 >> 2);
  const int j = lane & 31, hi = lane >> 5;
  const int HD = H * D;
  const int s_begin = chunk * KC;
  const int s_end = min(S, s_begin + KC);
  const int cw = KC / 32, cws = cw + 1;
  const int nmt = (Q + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* Ms = reinterpret_cast<uint32_t*>(smem_raw);   // [nmt*32][cws]
  for (int i = tid; i < nmt * 32 * cw; i += 512) {
    const int qq = i / cw, w = i - qq * cw;
    const int gw = s_begin / 32 + w;
    uint32_t m = 0u;
    if (bits != nullptr && qq < Q && gw < words) m = bits[((size_t)b * Q + qq) * words + gw];
    Ms[qq * cws + w] = m;
  }
  __syncthreads();
  if (wave >= nmt || h >= H) return;

  const int qi = wave * 32 + j;
  // auto_unmask: a query whose mask blocks EVERY key attends to all of them (mask2former_head.py:825-826). Whether
  // that holds is only known across chunks, so a chunk that finds its slice of the row fully blocked processes it
  // UNMASKED and marks its partial as provisional (negative sum); the combine kernel keeps provisional partials only if
  // all chunks of the row are provisional, and drops them otherwise -- exact, and the separate
  // cgg_attn_mask_fix_full_rows launch before every layer disappears.
  bool row_open = false;
  if (auto_unmask && bits != nullptr && qi < Q) {
    uint32_t all = ~0u;
    for (int w = 0; w < cw; ++w) {
      const int gw = s_begin / 32 + w;
      uint32_t m = Ms[qi * cws + w];
      if (gw >= words) m = ~0u;                                           // past the last key
      else if (gw == words - 1 && (S & 31)) m |= ~0u << (S & 31);         // invalid tail bits of the last word
      all &= m;
    }
    row_open = all == ~0u;
  }
  // VALU diet (the kernel is issue-bound, not memory-bound): scores live in the log2 domain (log2 e folded into the
  // query scale -> one v_exp_f32 per probability, no multiply), the (max, sum) partials are written in that domain and
  // cgg_xattn_combine_wave(log2 = 1) rescales with exp2; the chunk tail is folded into the mask word once per step
  const float scale2 = scale * 1.44269504088896341f;
  bf16x8 qb[2];
  {
    const bool ok = qi < Q;
    const float* qp = q + ((size_t)b * Q + (ok ? qi : 0)) * HD + h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * ks), c = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
      uint16_t t[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[e] = cgg_f2bf(ok ? a[e] * scale2 : 0.f);
        t[4 + e] = cgg_f2bf(ok ? c[e] * scale2 : 0.f);
      }
      const uint4 u = make_uint4(cgg_pack2(t[0], t[1]), cgg_pack2(t[2], t[3]), cgg_pack2(t[4], t[5]), cgg_pack2(t[6], t[7]));
      qb[ks] = __builtin_bit_cast(bf16x8, u);
    }
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;

  // K rows may be a column slice of a wider projection (row stride ldk), vt one row block of a taller one (batch
  // stride vt_bstride): the K / V projections of all decoder layers that read one level are a single GEMM each
  const uint16_t* kb = k + (size_t)b * S * ldk + h * D + 8 * hi;         // + key * ldk + 16 * ks
  const uint16_t* vb = vt + (size_t)b * vt_bstride + (size_t)(h * D + j) * S + 4 * hi;   // + key0 + 16 * t (+ 8)
  const int s_cap = S - 4;                                                // last legal 4-key group start
  for (int s0 = s_begin; s0 < s_end; s0 += 32) {
    const int key = min(s0 + j, S - 1);
    const uint4 k0 = *reinterpret_cast<const uint4*>(kb + (size_t)key * ldk);
    const uint4 k1 = *reinterpret_cast<const uint4*>(kb + (size_t)key * ldk + 16);
    uint2 v[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      v[2 * t] = *reinterpret_cast<const uint2*>(vb + min(s0 + 16 * t, s_cap - 4 * hi));
      v[2 * t + 1] = *reinterpret_cast<const uint2*>(vb + min(s0 + 16 * t + 8, s_cap - 4 * hi));
    }
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k0), qb[0], sc, 0, 0, 0);
    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k1), qb[1], sc, 0, 0, 0);
    uint32_t mw = row_open ? 0u : Ms[qi * cws + ((s0 - s_begin) >> 5)];
    if (s0 + 32 > s_end) mw |= ~0u << (s_end - s0);            // keys past the chunk end (last step only)
    mw >>= 4 * hi;                                               // this lane's keys: bits (r&3) + 8 (r>>2)
    float rmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = (mw & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : sc[r];
      rmax = fmaxf(rmax, sc[r]);
    }
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
    const float m_new = fmaxf(m_run, rmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float psum = 0.f;
    uint16_t pb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(sc[r] - m_use);
      psum += p;
      pb[r] = cgg_f2bf(p);
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // a clamped V address only ever pairs with p == 0 (blocked tail keys); V is finite, so 0 * V == 0
      const uint4 va = make_uint4(v[2 * t].x, v[2 * t].y, v[2 * t + 1].x, v[2 * t + 1].y);
      const uint4 pp = make_uint4(cgg_pack2(pb[8 * t], pb[8 * t + 1]), cgg_pack2(pb[8 * t + 2], pb[8 * t + 3]),
                                  cgg_pack2(pb[8 * t + 4], pb[8 * t + 5]), cgg_pack2(pb[8 * t + 6], pb[8 * t + 7]));
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pp), o, 0, 0, 0);
    }
  }
  if (qi < Q && out_direct != nullptr) {
    float* op = out_direct + ((size_t)b * Q + qi) * (H * D) + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 vv = {o[4 * g] / l_run, o[4 * g + 1] / l_run, o[4 * g + 2] / l_run, o[4 * g + 3] / l_run};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * hi) = vv;
    }
  } else if (qi < Q) {
    const size_t base = (((size_t)b * H + h) * nchunks + chunk) * Q + qi;
    float* op = ws_o + base * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 vv = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * hi) = vv;
    }
    if (hi == 0) {
      ws_ml[base * 2] = m_run;
      ws_ml[base * 2 + 1] = row_open ? -l_run : l_run;         // negative sum = provisional (unmasked) partial
    }
  }
}

// Self-attention of the query decoder in throughput mode (mask2former_head.py:832-836): S = Q <= 128 keys, no mask,
// q [M, ldq] and kv = [k | v] [M, ldkv] f32 rows straight from the fused q|k|v projection. One workgroup per (head,
// image), one wavefront per 32-query tile, no LDS: the whole K / V of a head is 2 x 12.5 KiB, every load of a wave is
// issued before the first MFMA. Same tile algebra as cgg_xattn_partial_bf16 (S^T = K Q^T so a lane owns one query's
// softmax; the P operand's key permutation is matched by the V gather); bf16 operands, f32 accumulation / softmax.
__global__ __launch_bounds__(256) void cgg_self_attn_small_kernel(const float* __restrict__ q, int ldq,
                                                                  const float* __restrict__ kv, int ldkv,
                                                                  float* __restrict__ out, int Q, int H, float scale) {
  constexpr int D = 32;
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int E = H * D, S = Q;
  if (wave * 32 >= Q) return;
  const int qi = wave * 32 + j;
  const bool q_ok = qi < Q;
  const float* qp = q + ((size_t)b * Q + (q_ok ? qi : 0)) * ldq + h * D + 8 * hi;
  const float* kbase = kv + (size_t)b * Q * ldkv + h * D;
  const float* vbase = kbase + E;
  f32x4 qf[4], kf[4][4];
  float vf[4][16];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[2 * ks] = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
    qf[2 * ks + 1] = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int s0 = 32 * st;
    if (s0 < S) {                                           // wave-uniform
      const float* kp = kbase + (size_t)min(s0 + j, S - 1) * ldkv + 8 * hi;
      kf[st][0] = *reinterpret_cast<const f32x4*>(kp);
      kf[st][1] = *reinterpret_cast<const f32x4*>(kp + 4);
      kf[st][2] = *reinterpret_cast<const f32x4*>(kp + 16);
      kf[st][3] = *reinterpret_cast<const f32x4*>(kp + 20);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = s0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        vf[st][r] = vbase[(size_t)min(key, S - 1) * ldkv + j];
      }
    }
  }
  bf16x8 qb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const f32x4 a = qf[2 * ks], c = qf[2 * ks + 1];
    const uint4 u = make_uint4(cgg_pack2(cgg_f2bf(q_ok ? a[0] * scale : 0.f), cgg_f2bf(q_ok ? a[1] * scale : 0.f)),
                               cgg_pack2(cgg_f2bf(q_ok ? a[2] * scale : 0.f), cgg_f2bf(q_ok ? a[3] * scale : 0.f)),
                               cgg_pack2(cgg_f2bf(q_ok ? c[0] * scale : 0.f), cgg_f2bf(q_ok ? c[1] * scale : 0.f)),
                               cgg_pack2(cgg_f2bf(q_ok ? c[2] * scale : 0.f), cgg_f2bf(q_ok ? c[3] * scale : 0.f)));
    qb[ks] = __builtin_bit_cast(bf16x8, u);
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int s0 = 32 * st;
    if (s0 < S) {
      bf16x8 kb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f32x4 a = kf[st][2 * ks], c = kf[st][2 * ks + 1];
        const uint4 u = make_uint4(cgg_pack2(cgg_f2bf(a[0]), cgg_f2bf(a[1])), cgg_pack2(cgg_f2bf(a[2]), cgg_f2bf(a[3])),
                                   cgg_pack2(cgg_f2bf(c[0]), cgg_f2bf(c[1])), cgg_pack2(cgg_f2bf(c[2]), cgg_f2bf(c[3])));
        kb[ks] = __builtin_bit_cast(bf16x8, u);
      }
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[0], qb[0], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[1], qb[1], sc, 0, 0, 0);
      float rmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ki = (r & 3) + 8 * (r >> 2) + 4 * hi;
        sc[r] = (s0 + ki >= S) ? -INFINITY : sc[r];
        rmax = fmaxf(rmax, sc[r]);
      }
      rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
      const float m_new = fmaxf(m_run, rmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = expf(m_run - m_use);
      float psum = 0.f;
      uint16_t pb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(sc[r] - m_use);
        psum += p;
        pb[r] = cgg_f2bf(p);
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint4 va = make_uint4(cgg_pack2(cgg_f2bf(vf[st][8 * t]), cgg_f2bf(vf[st][8 * t + 1])),
                                    cgg_pack2(cgg_f2bf(vf[st][8 * t + 2]), cgg_f2bf(vf[st][8 * t + 3])),
                                    cgg_pack2(cgg_f2bf(vf[st][8 * t + 4]), cgg_f2bf(vf[st][8 * t + 5])),
                                    cgg_pack2(cgg_f2bf(vf[st][8 * t + 6]), cgg_f2bf(vf[st][8 * t + 7])));
        const uint4 pp = make_uint4(cgg_pack2(pb[8 * t], pb[8 * t + 1]), cgg_pack2(pb[8 * t + 2], pb[8 * t + 3]),
                                    cgg_pack2(pb[8 * t + 4], pb[8 * t + 5]), cgg_pack2(pb[8 * t + 6], pb[8 * t + 7]));
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pp), o, 0, 0, 0);
      }
    }
  }
  if (q_ok) {
    float* op = out + ((size_t)b * Q + qi) * E + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 vv = {o[4 * g] / l_run, o[4 * g + 1] / l_run, o[4 * g + 2] / l_run, o[4 * g + 3] / l_run};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * hi) = vv;
    }
  }
}

extern "C" int cgg_self_attn_rows_bf16(const float* q, int ldq, const float* kv, int ldkv, float* out, int B, int Q,
                                       int H, int D, float scale, cgg_stream_t stream) {
  CGG_REQUIRE(q && kv && out, CGG_EINVAL, "cgg_self_attn_rows_bf16: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && H > 0, CGG_EINVAL, "cgg_self_attn_rows_bf16: bad sizes");
  CGG_REQUIRE(D == 32, CGG_EUNSUPPORTED, "cgg_self_attn_rows_bf16: head dim %d (only 32 is built)", D);
  CGG_REQUIRE(Q <= 128, CGG_EUNSUPPORTED, "cgg_self_attn_rows_bf16: Q=%d > 128", Q);
  CGG_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldq >= H * D && ldkv >= 2 * H * D && cgg_aligned16(q) && cgg_aligned16(kv) &&
                  cgg_aligned16(out),
              CGG_EALIGN, "cgg_self_attn_rows_bf16: 16-B aligned rows required (ldq=%d ldkv=%d)", ldq, ldkv);
  hipLaunchKernelGGL(cgg_self_attn_small_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, q, ldq, kv, ldkv, out, Q,
                     H, scale);
  CGG_CHECK_LAUNCH("cgg_self_attn_rows_bf16");
  return CGG_OK;
}

// combine the per-chunk partials: thread = (b, q, h, d)
__global__ __launch_bounds__(256) void cgg_xattn_combine(const float* __restrict__ ws_o,
                                                         const float* __restrict__ ws_ml,
                                                         float* __restrict__ out, int B, int Q, int H,
                                                         int D, int nchunks, float* __restrict__ lse) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Q * H * D;
  if (gid >= total) return;
  const int d = (int)(gid % D);
  const int h = (int)((gid / D) % H);
  const int qq = (int)((gid / ((long long)D * H)) % Q);
  const int b = (int)(gid / ((long long)D * H * Q));
  const size_t base = ((size_t)b * H + h) * nchunks;
  float M = -INFINITY;
  for (int c = 0; c < nchunks; ++c) M = fmaxf(M, ws_ml[((base + c) * Q + qq) * 2]);
  float num = 0.f, den = 0.f;
  for (int c = 0; c < nchunks; ++c) {
    const size_t r = (base + c) * Q + qq;
    const float mc = ws_ml[r * 2];
    const float f = (mc == -INFINITY) ? 0.f : expf(mc - M);
    num += f * ws_o[r * D + d];
    den += f * ws_ml[r * 2 + 1];
  }
  out[((size_t)b * Q + qq) * (H * D) + h * D + d] = num / den;  // den == 0 -> NaN, as the reference
  if (lse != nullptr && d == 0) lse[((size_t)b * H + h) * Q + qq] = M + logf(den);
}

// Same result, one wavefront per (b, h, q) row, for nchunks <= 64: lane c first owns chunk c's (max, sum) -- the
// softmax rescale factors come from two wavefront reductions instead of two serial loops of dependent loads -- then
// lane (d, half) accumulates O over the chunks of its parity, 4 independent loads in flight per lane.
__global__ __launch_bounds__(256) void cgg_xattn_combine_wave(const float* __restrict__ ws_o,
                                                              const float* __restrict__ ws_ml,
                                                              float* __restrict__ out, int B, int Q, int H,
                                                              int nchunks, int log2_domain, float* __restrict__ lse) {
  constexpr int D = 32;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // (b, h, q) flattened as ((b*H + h)*Q + q)
  const int lane = threadIdx.x & 63;
  if (row >= B * H * Q) return;
  const int qq = row % Q, bh = row / Q;
  const size_t r0 = (size_t)bh * nchunks * Q + qq;               // chunk c -> row r0 + c*Q of the workspace
  float mc = -INFINITY, lc = 0.f;
  if (lane < nchunks) {
    const float2 ml = *reinterpret_cast<const float2*>(ws_ml + (r0 + (size_t)lane * Q) * 2);
    mc = ml.x;
    lc = ml.y;
  }
  // provisional partials (negative sum, see cgg_xattn_partial_bf16): valid only if every chunk of the row is one
  const bool prov = lc < 0.f;
  const bool any_firm = __ballot(lane < nchunks && !prov) != 0ull;
  if (prov) {
    if (any_firm) { mc = -INFINITY; lc = 0.f; }
    else lc = -lc;
  }
  float M = mc;
  for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
  const float f = (mc == -INFINITY) ? 0.f : (log2_domain ? __builtin_amdgcn_exp2f(mc - M) : expf(mc - M));
  float den = f * lc;
  for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
  const int d = lane & 31, half = lane >> 5;
  float num = 0.f;
  int c = half;
  for (; c + 6 < nchunks; c += 8) {
    const float v0 = ws_o[(r0 + (size_t)c * Q) * D + d], v1 = ws_o[(r0 + (size_t)(c + 2) * Q) * D + d];
    const float v2 = ws_o[(r0 + (size_t)(c + 4) * Q) * D + d], v3 = ws_o[(r0 + (size_t)(c + 6) * Q) * D + d];
    num += __shfl(f, c) * v0;
    num += __shfl(f, c + 2) * v1;
    num += __shfl(f, c + 4) * v2;
    num += __shfl(f, c + 6) * v3;
  }
  for (; c < nchunks; c += 2) num += __shfl(f, c) * ws_o[(r0 + (size_t)c * Q) * D + d];
  num += __shfl_xor(num, 32);
  if (half == 0) {
    const int b = bh / H, h = bh - b * H;
    out[((size_t)b * Q + qq) * (H * D) + h * D + d] = num / den;   // den == 0 -> NaN, as the reference
    if (lse != nullptr && d == 0)       // natural-log rows either way (the x3 partial pass keeps its maxima in log2 units)
      lse[(size_t)bh * Q + qq] = log2_domain ? (M + log2f(den)) * 0.6931471805599453f : M + logf(den);
  }
}

static int xattn_combine_launch(const float* ws_o, const float* ws_ml, float* out, int B, int Q, int H, int D, int nch,
                                hipStream_t s, int log2_domain, float* lse = nullptr) {
  if (nch <= 64 && D == 32) {
    const int rows = B * H * Q;
    hipLaunchKernelGGL(cgg_xattn_combine_wave, dim3((rows + 3) / 4), dim3(256), 0, s, ws_o, ws_ml, out, B, Q, H, nch,
                       log2_domain, lse);
  } else if (log2_domain) {
    return -1;                           // callers keep nch <= 64 for the log2-domain kernel
  } else {
    const long long total = (long long)B * Q * H * D;
    hipLaunchKernelGGL(cgg_xattn_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws_o, ws_ml, out, B,
                       Q, H, D, nch, lse);
  }
  return 0;
}

// -------------------------------------------------------------------------------------------------
static void xattn_plan(int B, int H, int S, int* KC, int* nchunks) {
  // aim at ~2 workgroups per CU; chunk = multiple of XA_TK keys, at most 1024 (mask LDS budget)
  int want = (512 + B * H - 1) / (B * H);
  int tiles = (S + XA_TK - 1) / XA_TK;
  if (tiles <= 4) want = 1;             // self-attention over ~100 queries: one chunk, no combine launch
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  int tpc = (tiles + want - 1) / want;  // tiles per chunk
  if (tpc > 1024 / XA_TK) tpc = 1024 / XA_TK;
  *KC = tpc * XA_TK;
  *nchunks = (S + *KC - 1) / *KC;
}

extern "C" int64_t cgg_masked_xattn_workspace_bytes(int B, int Q, int H, int D, int S) {
  if (B <= 0 || Q <= 0 || H <= 0 || D <= 0 || S <= 0) return 0;
  int KC, nch;
  xattn_plan(B, H, S, &KC, &nch);
  return (int64_t)B * H * nch * Q * (D + 2) * (int64_t)sizeof(float);
}

void cgg_xattn_partial_x3_launch(int nch, int H, int B, size_t mask_lds, hipStream_t s, const float* q, const float* kv,
                                 const uint32_t* bits, float* ws_o, float* ws_ml, int Q, int S, int words, int KC, float scale,
                                 float* out_direct, float* lse, int ldkv, long long kv_bstride);      // xattn_x3.hip

// x3 != 0: the partial pass on the f32-class f16 x 3 contraction (xattn_x3.hip) instead of the f32 MFMA
static int xattn_forward_f32(const float* q, const void* kv, const uint32_t* bits, float* out, float* lse, void* ws, int B,
                             int Q, int H, int D, int S, float scale, int kv_dtype, cgg_stream_t stream, int ldkv = 0,
                             int64_t kv_bstride = 0, int x3 = 0) {
  if (ldkv <= 0) ldkv = 2 * H * D;
  if (kv_bstride <= 0) kv_bstride = (int64_t)S * ldkv;
  CGG_REQUIRE(ldkv >= 2 * H * D && ldkv % 4 == 0 && kv_bstride % 4 == 0, CGG_EINVAL,
              "cgg_masked_xattn_forward: ldkv=%d / kv_bstride=%lld", ldkv, (long long)kv_bstride);
  CGG_REQUIRE(q && kv && out && ws, CGG_EINVAL, "cgg_masked_xattn_forward: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && H > 0 && S > 0, CGG_EINVAL, "cgg_masked_xattn_forward: bad sizes");
  CGG_REQUIRE(D == 32, CGG_EUNSUPPORTED, "cgg_masked_xattn_forward: head dim %d (only 32 is built)", D);
  CGG_REQUIRE(Q <= 128, CGG_EUNSUPPORTED, "cgg_masked_xattn_forward: Q=%d > 128", Q);
  CGG_REQUIRE(kv_dtype == CGG_F32 || kv_dtype == CGG_F32_BF16MFMA, CGG_EUNSUPPORTED,
              "cgg_masked_xattn_forward: kv dtype %d (f32 here; bf16 via cgg_masked_xattn_forward_bf16)",
              kv_dtype);
  const bool bfm = kv_dtype == CGG_F32_BF16MFMA && !x3;      // f32 rows in memory, bf16 MFMA operands (throughput-mode training)
  CGG_REQUIRE(cgg_aligned16(q) && cgg_aligned16(kv) && cgg_aligned16(ws), CGG_EALIGN,
              "cgg_masked_xattn_forward: q / kv / ws must be 16-B aligned");
  int KC, nch;
  xattn_plan(B, H, S, &KC, &nch);
  const int words = (S + 31) / 32;
  const int nmt = (Q + 31) / 32;
  float* ws_o = (float*)ws;
  float* ws_ml = ws_o + (size_t)B * H * nch * Q * D;
  const size_t lds = (size_t)2 * XA_TK * D * sizeof(float) + (size_t)nmt * 32 * (KC / 32 + 1) * 4;
  hipStream_t s = (hipStream_t)stream;
  if (x3)
    cgg_xattn_partial_x3_launch(nch, H, B, (size_t)nmt * 32 * (KC / 32 + 1) * 4, s, q, (const float*)kv, bits, ws_o, ws_ml, Q, S, words, KC,
                                scale, nch == 1 ? out : nullptr, lse, ldkv, (long long)kv_bstride);
  else if (bfm)
    hipLaunchKernelGGL(cgg_xattn_partial_f32<true>, dim3(nch, H, B), dim3(256), lds, s, q, (const float*)kv,
                       bits, ws_o, ws_ml, Q, H, S, words, KC, nch, scale, nch == 1 ? out : nullptr, lse, ldkv, (long long)kv_bstride);
  else
    hipLaunchKernelGGL(cgg_xattn_partial_f32<false>, dim3(nch, H, B), dim3(256), lds, s, q, (const float*)kv,
                       bits, ws_o, ws_ml, Q, H, S, words, KC, nch, scale, nch == 1 ? out : nullptr, lse, ldkv, (long long)kv_bstride);
  CGG_CHECK_LAUNCH("cgg_masked_xattn_forward(partial)");
  if (nch > 1) xattn_combine_launch(ws_o, ws_ml, out, B, Q, H, D, nch, s, x3 ? 1 : 0, lse);
  CGG_CHECK_LAUNCH("cgg_masked_xattn_forward(combine)");
  return CGG_OK;
}

extern "C" int cgg_masked_xattn_forward(const float* q, const void* kv, const uint32_t* bits,
                                        float* out, void* ws, int B, int Q, int H, int D, int S,
                                        float scale, int kv_dtype, cgg_stream_t stream) {
  return xattn_forward_f32(q, kv, bits, out, nullptr, ws, B, Q, H, D, S, scale, kv_dtype, stream);
}

// kv rows at stride ldkv elements, images at kv_bstride: layer j's [K | V] as a 2 E-column slice of the merged projection of the
// decoder layers that read one level (round 4: one x3 GEMM per level instead of one per layer and image)
extern "C" int cgg_masked_xattn_forward_strided(const float* q, const float* kv, int ldkv, int64_t kv_bstride, const uint32_t* bits,
                                                float* out, void* ws, int B, int Q, int H, int D, int S, float scale,
                                                cgg_stream_t stream) {
  return xattn_forward_f32(q, kv, bits, out, nullptr, ws, B, Q, H, D, S, scale, CGG_F32, stream, ldkv, kv_bstride);
}

// The same operator with both products on the f32-class f16 x 3 contraction (csrc/x3.h; |q scale|, |k|, |v| < 4094): parity mode's
// inference path. ldkv / kv_bstride as in cgg_masked_xattn_forward_strided (0 = contiguous (B, S, 2 E) rows).
extern "C" int cgg_masked_xattn_forward_x3(const float* q, const float* kv, int ldkv, int64_t kv_bstride, const uint32_t* bits,
                                           float* out, void* ws, int B, int Q, int H, int D, int S, float scale, cgg_stream_t stream) {
  return xattn_forward_f32(q, kv, bits, out, nullptr, ws, B, Q, H, D, S, scale, CGG_F32, stream, ldkv, kv_bstride, 1);
}

extern "C" int cgg_masked_xattn_forward_lse(const float* q, const void* kv, const uint32_t* bits, float* out, float* lse,
                                            void* ws, int B, int Q, int H, int D, int S, float scale, int kv_dtype,
                                            cgg_stream_t stream) {
  CGG_REQUIRE(lse, CGG_EINVAL, "cgg_masked_xattn_forward_lse: null lse");
  if (kv_dtype == CGG_F32_X3)          // parity-mode training: the forward on the f16 x 3 contraction, the same saved rows
    return xattn_forward_f32(q, kv, bits, out, lse, ws, B, Q, H, D, S, scale, CGG_F32, stream, 0, 0, 1);
  return xattn_forward_f32(q, kv, bits, out, lse, ws, B, Q, H, D, S, scale, kv_dtype, stream);
}

extern "C" int cgg_masked_xattn_forward_bf16(const float* q, const void* k, const void* vt, const uint32_t* bits,
                                             float* out, void* ws, int B, int Q, int H, int D, int S, float scale,
                                             int ldk, int64_t vt_bstride, int auto_unmask, cgg_stream_t stream) {
  if (ldk <= 0) ldk = H * D;
  if (vt_bstride <= 0) vt_bstride = (int64_t)H * D * S;
  CGG_REQUIRE(ldk >= H * D && ldk % 8 == 0 && vt_bstride >= (int64_t)H * D * S && vt_bstride % 4 == 0, CGG_EINVAL,
              "cgg_masked_xattn_forward_bf16: ldk=%d / vt_bstride=%lld", ldk, (long long)vt_bstride);
  CGG_REQUIRE(q && k && vt && out && ws, CGG_EINVAL, "cgg_masked_xattn_forward_bf16: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && H > 0 && S > 0, CGG_EINVAL, "cgg_masked_xattn_forward_bf16: bad sizes");
  CGG_REQUIRE(D == 32, CGG_EUNSUPPORTED, "cgg_masked_xattn_forward_bf16: head dim %d (only 32 is built)", D);
  CGG_REQUIRE(Q <= 128, CGG_EUNSUPPORTED, "cgg_masked_xattn_forward_bf16: Q=%d > 128", Q);
  CGG_REQUIRE(S % 4 == 0 && S >= 8, CGG_EUNSUPPORTED, "cgg_masked_xattn_forward_bf16: S=%d must be a multiple of 4", S);
  CGG_REQUIRE(cgg_aligned16(q) && cgg_aligned16(k) && cgg_aligned16(vt) && cgg_aligned16(ws), CGG_EALIGN,
              "cgg_masked_xattn_forward_bf16: q / k / vt / ws must be 16-B aligned");
  int KC, nch;
  xattn_plan(B, H, S, &KC, &nch);
  const int words = (S + 31) / 32;
  const int nmt = (Q + 31) / 32;
  float* ws_o = (float*)ws;
  float* ws_ml = ws_o + (size_t)B * H * nch * Q * D;
  const size_t lds = (size_t)nmt * 32 * (KC / 32 + 1) * 4;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cgg_xattn_partial_bf16, dim3(nch, (H + 1) / 2, B), dim3(512), lds, s, q, (const uint16_t*)k,
                     (const uint16_t*)vt, bits, ws_o, ws_ml, Q, H, S, words, KC, nch, scale, nch == 1 ? out : nullptr, ldk,
                     (long long)vt_bstride, auto_unmask);
  CGG_CHECK_LAUNCH("cgg_masked_xattn_forward_bf16(partial)");
  if (nch > 1) {
    CGG_REQUIRE(xattn_combine_launch(ws_o, ws_ml, out, B, Q, H, D, nch, s, 1) == 0, CGG_EUNSUPPORTED,
                "cgg_masked_xattn_forward_bf16: %d key chunks (> 64)", nch);
  }
  CGG_CHECK_LAUNCH("cgg_masked_xattn_forward_bf16(combine)");
  return CGG_OK;
}
