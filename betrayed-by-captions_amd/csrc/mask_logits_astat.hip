// K3/K4/K5, round 5: the mask-logit einsum with the QUERY operand stationary in registers and the consumer in the epilogue.
//   mask_pred[b, q, p] = sum_c mask_embed[b, q, c] * mask_feature[b, c, p]      (open_set/models/mask2former_head.py:748)
//   attn_mask bit      = mask_pred < 0                                          (:749-759, sigmoid(x) < 0.5 <=> x < 0)
//
// Why a second kernel. `cgg_mask_logits_kernel` (mask_logits.hip) keeps a 32-pixel tile's B fragments in registers and reads the A
// fragment (32 queries x 16 channels, 1 KiB) of EVERY MFMA from LDS: 1 KiB of LDS traffic per 32-cycle MFMA and SIMD = the CU's
// whole 128 B/clk LDS bandwidth at 100 % MFMA rate, so it sits at 10-15 % of the bf16 MFMA peak whether or not the logits are
// stored (round-5 measurement: Q = 200 bits-only 37.0 us vs 38.4 us with f32 logits out -- the store is not the bound).
// Here a wavefront owns up to FOUR query tiles (128 queries) for the whole launch: their 64 A fragments live in 256 VGPRs (one wave
// per SIMD, the 512-register budget), the pixel tiles stream through as 16 coalesced 1-KiB loads each (the packed image of
// cgg_pack_mask_feature*: every B fragment is one global_load_dwordx4 per lane, no LDS), two register sets deep (the next tile in
// flight behind the one being multiplied). Per MFMA the CU moves 256 B instead of 1 KiB, all of it through the vector memory path; LDS is only used once, to
// turn mask_embed into fragment order. Q > 128: two wave groups hold query tiles [0, 4) and [4, MT), the two waves of a pair walk
// the same pixel tiles (the second read of a tile hits L2).
//
// Consumer fused: `bits` (B, Q, T) -- bit j of word (q, t) = logit(q, 32 t + j) < 0 -- straight from the accumulators by wave
// ballots, staged per wave in LDS and written as contiguous pieces of the rows; the logits are never stored (the forward's last
// layer, which needs them, keeps cgg_mask_logits). The kernel's HBM traffic is the packed feature (67 MB at 1024^2, batch 2): arithmetic intensity 97 (Q = 100) / 190 (Q = 200) FLOP/B instead of 56 / 78.
#include "x3.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define MLA_KS 16        // C = 256

__device__ __forceinline__ void mla_load_tile(u32x4 (&dst)[MLA_KS], const u32x4* __restrict__ base, int t, bool stream) {
  const u32x4* p = base + (size_t)t * (MLA_KS * 64);
#pragma unroll
  for (int ks = 0; ks < MLA_KS; ++ks) dst[ks] = stream ? __builtin_nontemporal_load(p + ks * 64) : p[ks * 64];
}

#define MLA_RUN 16       // tiles whose mask words a wave stages in LDS before it writes them out as rows

template <int MTW>
__global__ __launch_bounds__(256, 1) void cgg_mask_logits_astat_kernel(const float* __restrict__ embed, const u32x4* __restrict__ fhi,
                                                                       uint32_t* __restrict__ bits, int Q, int npix, int T, int MT) {
  constexpr int C = MLA_KS * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* a_lds = reinterpret_cast<u32x4*>(smem_raw);          // [MT][KS][64] bf16 A fragments (prologue), then the words' staging
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi5 = lane >> 5, col = lane & 31;
  // wave -> (query-tile group, pixel stream): one group when MT <= MTW (4 streams per workgroup), two otherwise (2 streams)
  const int groups = MT > MTW ? 2 : 1;
  const int streams = 4 / groups;
  const int grp = wave / streams, stream = wave - grp * streams;
  const int mt0 = grp * MTW;
  // this stream's contiguous run of pixel tiles
  const int nstreams = gridDim.x * streams;
  const int sid = blockIdx.x * streams + stream;
  const int per = (T + nstreams - 1) / nstreams;
  const int tbeg = sid * per, tend = min(T, tbeg + per);
  const u32x4* __restrict__ fb = fhi + (size_t)b * T * (MLA_KS * 64) + lane;
  const bool nt = groups == 1;                                 // a tile read by two groups should stay cacheable

  // ---- (0) the first two tiles are requested before the prologue: their HBM latency hides under it ----
  u32x4 b0[MLA_KS], b1[MLA_KS];
  if (tbeg < tend) mla_load_tile(b0, fb, tbeg, nt);
  if (tbeg + 1 < tend) mla_load_tile(b1, fb, tbeg + 1, nt);      // (MLA_RUN >= 2: inside the first run)

  // ---- (1) prologue: mask_embed[b] -> bf16 A fragments in LDS (float4 (q, c4) -> slot (q / 32, c4 / 4, q % 32 + 32 ((c4 / 2) & 1)),
  //      half c4 & 1), rows >= Q are zero; then this wave's MTW x 16 fragments into registers ----
  {
    const f32x4* __restrict__ eb4 = reinterpret_cast<const f32x4*>(embed + (size_t)b * Q * C);
    uint2* a2 = reinterpret_cast<uint2*>(a_lds);
    const int nf = MT * 32 * (C / 4), nvalid = Q * (C / 4);
    // ALL of this thread's loads first (<= 64 float4: the A / accumulator registers are not live yet), then the conversions: one
    // L2 round trip instead of one per group of four (with one workgroup per CU nothing else hides them: the looped form cost ~10 us)
    constexpr int NF = 2 * MTW * 8;                            // float4 per thread for MT = 2 MTW query tiles
    f32x4 ev[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + 256 * u;
      ev[u] = (f < nvalid) ? eb4[f] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + 256 * u;
      if (f < nf) {
        const int q = f >> 6, c4 = f & 63;
        const int slot = ((q >> 5) * MLA_KS + (c4 >> 2)) * 64 + (q & 31) + 32 * ((c4 >> 1) & 1);
        a2[slot * 2 + (c4 & 1)] = make_uint2(cgg_pack2(cgg_f2bf(ev[u][0]), cgg_f2bf(ev[u][1])),
                                             cgg_pack2(cgg_f2bf(ev[u][2]), cgg_f2bf(ev[u][3])));
      }
    }
  }
  __syncthreads();
  u32x4 A[MTW][MLA_KS];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int ks = 0; ks < MLA_KS; ++ks)
      A[i][ks] = (mt0 + i < MT) ? a_lds[((mt0 + i) * MLA_KS + ks) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
  __syncthreads();                                             // the fragment image is dead: its LDS becomes the staging area

  // mask words of this wave: stage[(query row of the group) * MLA_RUN + tile of the run] -- compile-time row strides (the direct
  // global store `bits[(q0 + row) * T + t]` made the compiler keep 16 x MTW row addresses live across the whole stream: 250 spilled
  // registers); a run of <= MLA_RUN tiles leaves as one contiguous piece per query row
  uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw) + wave * (MTW * 32 * MLA_RUN);
  uint32_t* __restrict__ bb = bits + ((size_t)b * Q + mt0 * 32) * T;
  const int nrows = min(MTW * 32, Q - mt0 * 32);               // uniform

  // one pixel tile: MTW x 16 MFMAs against the register-resident query tiles. The threshold consumer of query tile i is interleaved,
  // step by step, with the MFMAs of query tile i + 1 (two accumulators): a wave issues in order, and with one wave per SIMD the
  // 16 ballots + 32 v_writelane of a tile's epilogue would otherwise run while the matrix pipe idles (measured before the
  // interleave: 0.5 us per (tile, query tile) unit against 0.21 us of MFMA time).
  // Threshold consumer: 16 ballots -> the 32 rows' words gathered into ONE register (lane = row) by v_writelane, one LDS store per
  // query tile (a predicated store per ballot = 16 exec-mask branches per query tile cost 0.9 us per unit).
  auto epi_step = [&](const f32x16& acc, int r, bool pin, uint32_t& wv) {
    const unsigned long long m = __ballot(pin && acc[r] < 0.f);
    const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);     // lanes 0-31 voted for row ql, lanes 32-63 for row ql + 4
    // (inline asm is opaque to the hazard recognizer: the s_nop covers "VALU writes an SGPR -> v_writelane reads it" -- without it
    // ONE of the 32 words of a query tile came out as all-ones, the one whose compare the scheduler had put right in front)
    switch (r) {
#define MLA_WL(R)                                                                                         \
  case R:                                                                                                 \
    asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"                 \
                 : "+v"(wv)                                                                               \
                 : "s"(mlo), "s"(mhi), "n"(((R) & 3) + 8 * ((R) >> 2)), "n"(((R) & 3) + 8 * ((R) >> 2) + 4)); \
    break;
      MLA_WL(0) MLA_WL(1) MLA_WL(2) MLA_WL(3) MLA_WL(4) MLA_WL(5) MLA_WL(6) MLA_WL(7)
      MLA_WL(8) MLA_WL(9) MLA_WL(10) MLA_WL(11) MLA_WL(12) MLA_WL(13) MLA_WL(14) MLA_WL(15)
#undef MLA_WL
    }
  };
  auto do_tile = [&](auto nmt_c, const u32x4 (&cur)[MLA_KS], int tt, int slot) {
    constexpr int NMT = decltype(nmt_c)::value;                // query tiles of this wave's group (compile time: straight-line code)
    const bool pin = tt * 32 + col < npix;
    f32x16 accC, accN;
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < MLA_KS; ++ks)
      accC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0][ks]), __builtin_bit_cast(bf16x8, cur[ks]), accC, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NMT; ++i) {
      uint32_t wv = 0u;
      if (i + 1 < NMT) {                                     // (compile time after unrolling)
#pragma unroll
        for (int r = 0; r < 16; ++r) accN[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < MLA_KS; ++ks) {
          accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i + 1 < MTW ? i + 1 : i][ks]),
                                                         __builtin_bit_cast(bf16x8, cur[ks]), accN, 0, 0, 0);
          epi_step(accC, ks, pin, wv);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) epi_step(accC, r, pin, wv);
      }
      if (lane < 32) stage[(i * 32 + lane) * MLA_RUN + slot] = wv;
      accC = accN;
    }
  };
  auto flush = [&](int t0, int n) {                            // tiles [t0, t0 + n) of the run -> bits rows
    for (int idx = lane; idx < nrows * MLA_RUN; idx += 64) {
      const int row = idx / MLA_RUN, k = idx - row * MLA_RUN;
      if (k < n) bb[(size_t)row * T + t0 + k] = stage[idx];
    }
  };

  // ---- (2) stream: two register sets (the next tile in flight behind the one being multiplied; a third set measured the same:
  //      35.8 vs 36.0 us -- the stream was never the bound) + two accumulators: 256 A + 2 x 64 B + 32 accumulator registers ----
  // Runs of <= MLA_RUN tiles: the tile loop of a run contains NO stores (the words wait in LDS), the run's flush follows it. With the
  // flush inside the tile loop the compiler's s_waitcnt pass saw stores of unknown count outstanding at the loop header and fell
  // back to vmcnt(0) in front of the first MFMA of every pair: the prefetched tile was waited for as well, i.e. half of the loads
  // had no compute to hide behind (stream and MFMA time added up: 35 us at Q = 200 instead of max(stream, compute) + prologue).
  auto stream_tiles = [&](auto nmt_c) {
    for (int run0 = tbeg; run0 < tend; run0 += MLA_RUN) {
      const int rend = min(tend, run0 + MLA_RUN);
      if (run0 != tbeg) {                                      // (the first run's tiles were requested before the prologue)
        mla_load_tile(b0, fb, run0, nt);
        if (run0 + 1 < rend) mla_load_tile(b1, fb, run0 + 1, nt);
      }
      for (int t = run0; t < rend; t += 2) {
        do_tile(nmt_c, b0, t, t - run0);
        if (t + 2 < rend) mla_load_tile(b0, fb, t + 2, nt);
        if (t + 1 < rend) {
          do_tile(nmt_c, b1, t + 1, t + 1 - run0);
          if (t + 3 < rend) mla_load_tile(b1, fb, t + 3, nt);
        }
      }
      flush(run0, rend - run0);
    }
  };
  const int nmt = min(MTW, MT - mt0);                          // wave-uniform
  if (nmt == 4) stream_tiles(std::integral_constant<int, 4>{});
  else if (nmt == 3) stream_tiles(std::integral_constant<int, 3>{});
  else if (nmt == 2) stream_tiles(std::integral_constant<int, 2>{});
  else if (nmt == 1) stream_tiles(std::integral_constant<int, 1>{});
}

// Consumer-fused bf16 form of cgg_mask_logits with the query operand stationary: embed (B, Q, 256) f32, hi = packed bf16 feature
// (cgg_pack_mask_feature*), bits (B, Q, ceil(npix / 32)) u32 = (logit < 0); the logits themselves are never stored. Q <= 256.
extern "C" int cgg_mask_logits_bits_astat(const float* embed, const void* hi, uint32_t* bits, int B, int Q, int C, int npix,
                                          cgg_stream_t stream) {
  CGG_REQUIRE(embed && hi && bits, CGG_EINVAL, "cgg_mask_logits_bits_astat: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && npix > 0, CGG_EINVAL, "cgg_mask_logits_bits_astat: bad sizes");
  CGG_REQUIRE(C == 256, CGG_EUNSUPPORTED, "cgg_mask_logits_bits_astat: C=%d (only 256 is built)", C);
  CGG_REQUIRE(Q <= 256, CGG_EUNSUPPORTED, "cgg_mask_logits_bits_astat: Q <= 256 (Q=%d)", Q);
  CGG_REQUIRE(cgg_aligned16(embed) && cgg_aligned16(hi), CGG_EALIGN, "cgg_mask_logits_bits_astat: 16-B alignment");
  const int MT = (Q + 31) / 32, T = (npix + 31) / 32;
  size_t lds = (size_t)MT * MLA_KS * 64 * 16;                  // fragment image; the staging area (4 waves x 128 rows x 16 words) reuses it
  const size_t stage = (size_t)4 * 4 * 32 * MLA_RUN * 4;
  if (lds < stage) lds = stage;
  const int streams = MT > 4 ? 2 : 4;
  // one workgroup (4 waves, one per SIMD) per CU where the image has enough tiles; every stream gets >= 2 tiles where possible
  int gx = (256 + B - 1) / B;
  const int want = (T + 2 * streams - 1) / (2 * streams);
  if (gx > want) gx = want;
  if (gx < 1) gx = 1;
  auto kern = cgg_mask_logits_astat_kernel<4>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_mask_logits_bits_astat: cannot raise dynamic LDS to %zu", lds);
  hipLaunchKernelGGL(kern, dim3(gx, B), dim3(256), lds, (hipStream_t)stream, embed, (const u32x4*)hi, bits, Q, npix, T, MT);
  CGG_CHECK_LAUNCH("cgg_mask_logits_bits_astat");
  return CGG_OK;
}
