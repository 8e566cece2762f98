// Parity mode's cross-attention core on the f32-class f16 x 3 contraction (round 4): cgg_xattn_partial_f32 (xattn.hip: same
// workgroup / wave mapping, same partial / combine protocol, replaces the same reference lines,
// open_set/models/mask2former_head.py:829-840) with the two products S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_f16:
// 12 MFMAs of 32 cycles per 32 keys instead of 32 v_mfma_f32_32x32x2_f32 of 64 cycles. K / V tiles are split into f16 pairs while
// they are staged into LDS (K as swizzled A-fragment rows, V row-major); the P^T operand is the S^T accumulator itself (lane =
// query, registers = keys in accumulator order), so V^T comes from `ds_read_b64_tr_b16` transpose reads that fetch the keys in that
// order. Values: |q scale|, |k|, |v| < 4094 (x3.h); a larger K / V value raises the x3 overflow flag.
#include "x3.h"

#define XA_TK 64  // keys per LDS tile
typedef __attribute__((ext_vector_type(4))) uint32_t x3_u32x4;
typedef __attribute__((ext_vector_type(4))) short xa_s16x4;
typedef __attribute__((address_space(3))) xa_s16x4 xa_lds_s16x4;

__global__ __launch_bounds__(256) void cgg_xattn_partial_x3(
    const float* __restrict__ q, const float* __restrict__ kv, const uint32_t* __restrict__ bits,
    float* __restrict__ ws_o, float* __restrict__ ws_ml, int Q, int H, int S, int words, int KC,
    int nchunks, float scale, float* __restrict__ out_direct, float* __restrict__ lse, int ldkv, long long kv_bstride,
    int* __restrict__ flag) {
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
  unsigned char* Kx = smem_raw;                               // [64 keys][8 slots x 16 B]: slots 0-3 hi (8 dims each), 4-7 lo; slot ^ ((key >> 1) & 7)
  unsigned char* Vh = Kx + XA_TK * 128;                       // [64 keys][32 dims] f16 hi, row-major (transpose-read as the PV A operand)
  unsigned char* Vl = Vh + XA_TK * 64;                        // ... lo
  uint32_t* Ms = reinterpret_cast<uint32_t*>(Vl + XA_TK * 64); // [nmt*32][cws]

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
  x3_u32x4 qh[2], ql[2];      // k-step s: dims 16 s + 8 hi .. + 7 of query qi, pre-scaled by `scale` (and by 16: x3.h)
  {
    const bool ok = qi < Q;
    const float* qp = q + ((size_t)b * Q + (ok ? qi : 0)) * HD + h * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * s), v1 = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
      const float sc_ = ok ? scale * 1.4426950408889634f : 0.f;       // log2 domain
      cgg_x3_split8(v0 * sc_, v1 * sc_, qh[s], ql[s]);
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
  float vmax = 0.f;
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
      const f32x4 kk = live ? kx[it] : z, vv = live ? vx[it] : z;
      vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(kk[0]), fabsf(kk[1])), fmaxf(fabsf(kk[2]), fabsf(kk[3]))));
      vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(vv[0]), fabsf(vv[1])), fmaxf(fabsf(vv[2]), fabsf(vv[3]))));
      uint2 kh2, kl2, vh2, vl2;
      cgg_x3_split4(kk, kh2, kl2);
      cgg_x3_split4(vv, vh2, vl2);
      const int sw = (key >> 1) & 7;
      *reinterpret_cast<uint2*>(Kx + key * 128 + 16 * ((slot >> 1) ^ sw) + 8 * (slot & 1)) = kh2;
      *reinterpret_cast<uint2*>(Kx + key * 128 + 16 * ((4 + (slot >> 1)) ^ sw) + 8 * (slot & 1)) = kl2;
      *reinterpret_cast<uint2*>(Vh + key * 64 + 8 * slot) = vh2;
      *reinterpret_cast<uint2*>(Vl + key * 64 + 8 * slot) = vl2;
    }
    __syncthreads();
    if (s0 + XA_TK < s_end) load_tile(s0 + XA_TK);       // workgroup-uniform
    if (!wave_live) continue;

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kt = sub * 32;  // key offset in tile
      // ---- S^T[key][query] = K Q^T : 2 k-steps x 3 v_mfma_f32_32x32x16_f16 (x3 arithmetic) ----
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      {
        const int krow = kt + j, sw = (krow >> 1) & 7;
        const unsigned char* kr = Kx + krow * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const x3_u32x4 kh_ = *reinterpret_cast<const x3_u32x4*>(kr + 16 * ((2 * s + hi) ^ sw));
          const x3_u32x4 kl_ = *reinterpret_cast<const x3_u32x4*>(kr + 16 * ((4 + 2 * s + hi) ^ sw));
          cgg_x3_mfma(sc, kh_, kl_, qh[s], ql[s]);
        }
      }
      // ---- mask + online softmax in the log2 domain (lane = query j; 16 keys in-register, partner lane ^ 32 has the rest). The
      //      accumulator holds 256 log2(e) logit (the query carries scale x log2 e); e = exp2(acc / 256 - m + 4) = 16 p is the
      //      pre-scaled probability the x3 split wants, so the scaling costs nothing: one fma + one v_exp_f32 per key ----
      uint32_t mw = Ms[qi * cws + ((s0 - s_begin + kt) >> 5)];
      const int left = s_end - (s0 + kt);                    // keys of this sub tile inside the chunk (workgroup-uniform)
      if (left < 32) mw |= left <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << left);
      const uint32_t mws = mw >> (4 * hi);                   // register r <-> bit (r & 3) + 8 (r >> 2)
      float rmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[r] = (mws & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : sc[r];
        rmax = fmaxf(rmax, sc[r]);
      }
      rmax = fmaxf(rmax, __shfl_xor(rmax, 32)) * (1.f / 256.f);
      const float m_new = fmaxf(m_run, rmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      const float off = 4.f - m_use;
      float psum = 0.f;
      float p16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p16[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], 1.f / 256.f, off));
        psum += p16[r];
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;                          // 16 x the row sum
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
      // ---- O^T[d][query] += V^T[d][key] P^T[key][query]: P^T = the S^T accumulators as B fragments (k-step t = registers 8 t .. 8 t + 7:
      //      keys 8 (2 t + i / 4) + 4 hi + i % 4), V^T by transpose reads of the row-major V tile in the same key order ----
      {
        const int g = lane >> 4, u = g >> 1, d0 = 16 * (g & 1), i16 = lane & 15;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          x3_u32x4 ph, pl;
          cgg_x3a_split8_prescaled(p16 + 8 * t, ph, pl);
          // chunk i16 of the transpose read = 4 dims d0 + 4 (i16 & 3) .. of key row (i16 >> 2)
          const int k0 = kt + 8 * (2 * t) + 4 * u + (i16 >> 2), k1 = k0 + 8;
          const int co = 2 * (d0 + 4 * (i16 & 3));
          const xa_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xa_lds_s16x4*)(Vh + k0 * 64 + co));
          const xa_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xa_lds_s16x4*)(Vh + k1 * 64 + co));
          const xa_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xa_lds_s16x4*)(Vl + k0 * 64 + co));
          const xa_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xa_lds_s16x4*)(Vl + k1 * 64 + co));
          const uint2 h0u = __builtin_bit_cast(uint2, h0), h1u = __builtin_bit_cast(uint2, h1);
          const uint2 l0u = __builtin_bit_cast(uint2, l0), l1u = __builtin_bit_cast(uint2, l1);
          const x3_u32x4 vh_ = {h0u.x, h0u.y, h1u.x, h1u.y}, vl_ = {l0u.x, l0u.y, l1u.x, l1u.y};
          cgg_x3_mfma(o, vh_, vl_, ph, pl);
        }
      }
    }
  }
  if (flag && !(vmax * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(flag, 1);
  // o accumulated 256 x (p v), l_run 16 x the row sum, m_run is in the log2 domain (the combine pass runs with log2_domain = 1)
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] *= (1.f / 256.f);
  l_run *= (1.f / 16.f);

  // ---- partials out: lane (j, hi) holds O[q = qi][d = (r&3) + 8*(r>>2) + 4*hi] ----
  if (wave_live && qi < Q && out_direct != nullptr) {
    // single chunk: the combine step degenerates to o / l (l == 0 -> NaN, as the reference's all-masked row)
    float* op = out_direct + ((size_t)b * Q + qi) * HD + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {o[4 * g] / l_run, o[4 * g + 1] / l_run, o[4 * g + 2] / l_run, o[4 * g + 3] / l_run};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * hi) = v;
    }
    if (lse != nullptr && hi == 0) lse[((size_t)b * H + h) * Q + qi] = (m_run + log2f(l_run)) * 0.6931471805599453f;
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


int* cgg_x3_overflow_flag_ptr();       // x3s_gemm.hip

// called by xattn.hip's forward (same plan / workspace / combine as the f32 kernel); LDS = the tiles (16 KiB) + the mask words
void cgg_xattn_partial_x3_launch(int nch, int H, int B, size_t mask_lds, hipStream_t s, const float* q, const float* kv,
                                 const uint32_t* bits, float* ws_o, float* ws_ml, int Q, int S, int words, int KC, float scale,
                                 float* out_direct, float* lse, int ldkv, long long kv_bstride) {
  const size_t lds = (size_t)XA_TK * 128 + 2 * (size_t)XA_TK * 64 + mask_lds;
  hipLaunchKernelGGL(cgg_xattn_partial_x3, dim3(nch, H, B), dim3(256), lds, s, q, kv, bits, ws_o, ws_ml, Q, H, S, words, KC, nch, scale,
                     out_direct, lse, ldkv, kv_bstride, cgg_x3_overflow_flag_ptr());
}
