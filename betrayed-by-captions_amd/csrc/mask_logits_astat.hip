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

// (by value: __builtin_bit_cast applied directly to an ext-vector ELEMENT expression reads element 0 whatever the index -- hipcc 7.2)
__device__ __forceinline__ uint32_t mla_f2u(float x) { return __builtin_bit_cast(uint32_t, x); }

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
  // (unconditional, clamped into the image: a stream without tiles requests tile T - 1 and never uses it)
  mla_load_tile(b0, fb, min(tbeg, T - 1), nt);
  mla_load_tile(b1, fb, min(tbeg + 1, min(tbeg + MLA_RUN, tend) - 1 < tbeg ? T - 1 : min(tbeg + MLA_RUN, tend) - 1), nt);

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

  // one pixel tile: MTW x 16 MFMAs against the register-resident query tiles, TRANSPOSED (round 6): the pixel fragments are the
  // MFMA's A operand and the query fragments its B operand (the two share one register format), so the accumulator of lane l holds,
  // for query column l & 31, the 16 pixels (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the tile. The threshold consumer is then lane-local:
  // logit < 0 <=> sign bit (the accumulators start at +0 and the operands are finite: no -0, no NaN), shifted into a 16-bit pattern
  // by ONE v_alignbit_b32 per accumulator register, spread to the lane's pixel positions with two shift-or-and steps, and joined with
  // the other half-wave's 16 pixels by one v_permlane32_swap: ~24 VALU operations per (tile, query tile) unit instead of the round-5
  // form's 16 ballots + 32 v_writelane + 16 s_nop (~110 operations, as long as the unit's 16 MFMAs themselves).
  // The consumer of query tile i is interleaved, step by step, with the MFMAs of query tile i + 1 (two accumulators).
  const uint32_t sh = 4u * (uint32_t)hi5;
  auto finish = [&](uint32_t w16, uint32_t tmask) -> uint32_t {
    uint32_t x = (w16 | (w16 << 8)) & 0x00FF00FFu;             // nibbles g of the 16-bit pattern -> bits 8 g .. 8 g + 3
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x <<= sh;                                                  // the upper half-wave owns pixels 8 g + 4 .. 8 g + 7
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (r[0] | r[1]) & tmask;                              // both half-waves now hold query l & 31's 32-pixel word
  };
  auto do_tile = [&](auto nmt_c, const u32x4 (&cur)[MLA_KS], int tt, int slot) {
    constexpr int NMT = decltype(nmt_c)::value;                // query tiles of this wave's group (compile time: straight-line code)
    const int valid = npix - tt * 32;                          // (uniform) < 32 only in the image's last tile
    const uint32_t tmask = valid >= 32 ? 0xffffffffu : ((1u << valid) - 1u);
    f32x16 accC, accN;
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < MLA_KS; ++ks)
      accC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[ks]), __builtin_bit_cast(bf16x8, A[0][ks]), accC, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NMT; ++i) {
      uint32_t w = 0u;
      if (i + 1 < NMT) {                                     // (compile time after unrolling)
#pragma unroll
        for (int r = 0; r < 16; ++r) accN[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < MLA_KS; ++ks) {
          accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[ks]),
                                                         __builtin_bit_cast(bf16x8, A[i + 1 < MTW ? i + 1 : i][ks]), accN, 0, 0, 0);
          w = __builtin_amdgcn_alignbit(w, mla_f2u(accC[15 - ks]), 31);     // w = (w << 1) | sign
        }
      } else {
#pragma unroll
        for (int r = 15; r >= 0; --r) w = __builtin_amdgcn_alignbit(w, mla_f2u(accC[r]), 31);
      }
      stage[(i * 32 + col) * MLA_RUN + slot] = finish(w, tmask);   // (lanes l and l + 32 store the same word to the same slot)
      accC = accN;
    }
  };
  auto flush = [&](int t0, int n) {                            // tiles [t0, t0 + n) of the run -> bits rows
    for (int idx = lane; idx < nrows * MLA_RUN; idx += 64) {
      const int row = idx / MLA_RUN, k = idx - row * MLA_RUN;
      if (k < n) bb[(size_t)row * T + t0 + k] = stage[idx];
    }
  };

  // ---- (2) stream: two register sets (the next tile in flight behind the one being multiplied) + two accumulators: 256 A +
  //      2 x 64 B + 32 accumulator registers. Runs of <= MLA_RUN tiles: the tile loop of a run contains NO stores (the words wait in
  //      LDS), the run's flush follows it.
  // Round 6: the tile loop is BRANCH-FREE -- every prefetch is issued unconditionally (the tile index is clamped to the run's last
  // tile: a redundant L2-hit load at the end of a run; an odd run repeats its last tile, writing the same words again). With the
  // round-5 form's `if (t + 2 < rend)` around the prefetches the compiler's s_waitcnt pass merged "16 loads pending" with "none" at
  // the loop header and waited with vmcnt(15 .. 0) in front of the first query tile's MFMAs -- i.e. for the NEXT tile's loads too, so
  // that no load ever had a tile's worth of MFMAs to hide behind (profiles/r6_einsum_astat.txt: 2.9 us per tile and stream against
  // 0.85 us of MFMA time). Now the loop header waits with vmcnt(31 .. 16): only for the tile it multiplies. ----
  auto stream_tiles = [&](auto nmt_c) {
    for (int run0 = tbeg; run0 < tend; run0 += MLA_RUN) {
      const int rend = min(tend, run0 + MLA_RUN);
      const int last = rend - 1;
      if (run0 != tbeg) {                                      // (the first run's tiles were requested before the prologue)
        mla_load_tile(b0, fb, run0, nt);
        mla_load_tile(b1, fb, min(run0 + 1, last), nt);
      }
      for (int t = run0; t < rend; t += 2) {
        do_tile(nmt_c, b0, t, t - run0);
        mla_load_tile(b0, fb, min(t + 2, last), nt);
        const int t1 = min(t + 1, last);
        do_tile(nmt_c, b1, t1, t1 - run0);
        mla_load_tile(b1, fb, min(t + 3, last), nt);
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

// -------------------------------------------------------------------------------------------------------------------------------
// Round 6, Q > 128: the same contraction with the pixel tiles streamed through an LDS RING by LDS-DMA and TWO wavefronts per SIMD.
// What the register-stream kernel above runs into at Q = 200 (profiles/r6_einsum_astat.txt): (a) one 16-KB tile in flight per
// wavefront, both waves of a group pair fetching the same tile -- after the vmcnt fix 1.98 us per tile against 0.85 us of MFMA time,
// the memory system's loaded latency; (b) with the 256 query-fragment registers of 4 query tiles a wave needs 464 VGPRs = ONE wave
// per SIMD, and a wave issues in order: its consumer VALU work, LDS reads and DMA issue all run with the matrix pipe idle (a first
// ring version with 4 such waves: 1.64 us per tile). Here:
//   * 8 waves per workgroup = 2 per SIMD, each holding TWO query tiles (128 fragment registers; <= 256 VGPRs): while one wave of a
//     SIMD is in its consumer / LDS / DMA code the other one's MFMAs keep the matrix pipe busy;
//   * a stream QUAD (the four waves that multiply the same pixel tiles against query tiles [0,2) [2,4) [4,6) [6,8)) shares a 4-slot
//     ring of 16-KB tiles in LDS, two quads per workgroup; `global_load_lds_dwordx4` moves a tile HBM -> LDS in 16 wave-instructions
//     (4 per wave of the quad), no registers involved; up to 3 tiles (48 KB) per quad = 96 KB per CU in flight, each fetched ONCE;
//   * a wave copies a tile's 16 B fragments LDS -> registers (ds_read_b128, conflict-free: the packed image is in fragment order)
//     INSIDE the previous tile's last MFMA chain -- fragment ks is overwritten right after the last MFMA that reads it -- so one
//     64-register set suffices and the LDS latency hides under MFMAs;
//   * per tile step: [s_waitcnt vmcnt(8): my 4 pieces of tile t + 1 have landed, counted by hand -- the DMAs are inline asm, so the
//     compiler neither waits for them nor drains them at the barrier] -> s_barrier (everybody's pieces have; everybody has copied
//     tile t out of its slot) -> issue my 4 pieces of tile t + 4 into tile t's slot -> MFMAs of tile t (+ the copy of tile t + 1).
// LDS: 2 quads x 4 slots x 16 KB = 128 KB ring + 32 KB word staging = the CU's 160 KB, one workgroup per CU.
// The prologue's A-fragment image (MT x 16 KB <= 128 KB from byte 0, dead after the prologue) overlays staging + the low slots; ring
// slots 3 of both quads ([128 KB, 160 KB)) are outside it, and tile 0 of each quad is requested into them BEFORE the prologue.
#define MLR_D 4                                                // ring depth (tiles per stream quad)
#define MLR_STAGE (8 * 2 * 32 * MLA_RUN * 4)                   // 32 KB: 8 waves x 64 query rows x MLA_RUN words
#define MLR_RING(quad, j) (MLR_STAGE + (((j) * 2 + (quad)) * (MLA_KS * 1024)))   // byte offset of slot j of a quad (interleaved)

// one 1-KB piece (a wave-instruction): lane l's 16 bytes at gsrc -> LDS byte address lds_dst + 16 l. M0 is written in the same
// statement that reads it and restored (it is compiler-reserved).
__device__ __forceinline__ void mlr_dma16(const u32x4* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

__global__ __launch_bounds__(512) void cgg_mask_logits_astat_ring_kernel(const float* __restrict__ embed, const u32x4* __restrict__ fhi,
                                                                         uint32_t* __restrict__ bits, int Q, int npix, int T, int MT,
                                                                         int per) {
  constexpr int MTW = 2;
  constexpr int C = MLA_KS * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* a_lds = reinterpret_cast<u32x4*>(smem_raw);          // prologue: [MT][KS][64] bf16 A fragments from byte 0
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi5 = lane >> 5, col = lane & 31;
  const int grp = wave >> 1, quad = wave & 1;                  // waves {0, 2, 4, 6} = quad 0 (query-tile groups 0..3), odd waves = quad 1
  const int mt0 = grp * MTW;
  const int sid = blockIdx.x * 2 + quad;                       // stream quad -> contiguous run of `per` pixel tiles
  const int tbeg = sid * per;
  const int nvalid = max(0, min(T, tbeg + per) - tbeg);        // tiles of this quad that exist (the image's last quads may run short)
  const u32x4* __restrict__ fb = fhi + (size_t)b * T * (MLA_KS * 64) + lane;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;         // LDS byte address of the dynamic segment (no static LDS in this kernel)
  auto tile_of = [&](int i) { return min(tbeg + i, T - 1); };  // (clamped: a short quad re-reads the image's last tile, never stores it)
  // this wave's 4 pieces of step i's tile -> ring slot (i + 3) % 4 of its quad
  auto dma_tile = [&](int i) {
    const u32x4* src = fb + (size_t)tile_of(i) * (MLA_KS * 64) + grp * (4 * 64);
    const uint32_t dst = lds0 + MLR_RING(quad, (i + 3) & 3) + grp * (4 * 1024);
#pragma unroll
    for (int k = 0; k < 4; ++k) mlr_dma16(src + k * 64, dst + k * 1024);
  };
  dma_tile(0);                                                 // (slot 3: outside the A image)

  // ---- (1) prologue: mask_embed[b] -> bf16 A fragments in LDS -> this wave's 2 x 16 fragments in registers ----
  {
    const f32x4* __restrict__ eb4 = reinterpret_cast<const f32x4*>(embed + (size_t)b * Q * C);
    uint2* a2 = reinterpret_cast<uint2*>(a_lds);
    const int nf = MT * 32 * (C / 4), nvalidf = Q * (C / 4);
    constexpr int NF = 32;                                     // float4 per thread for MT = 8 query tiles (512 threads)
    f32x4 ev[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + 512 * u;
      ev[u] = (f < nvalidf) ? eb4[f] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + 512 * u;
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
      A[i][ks] = a_lds[(min(mt0 + i, MT - 1) * MLA_KS + ks) * 64 + lane];    // (a tile >= MT is never multiplied: `nmt` below)
  __syncthreads();                                             // the A image is dead: its LDS is staging + ring from here on
  dma_tile(1);
  dma_tile(2);

  uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw) + wave * (MTW * 32 * MLA_RUN);
  uint32_t* __restrict__ bb = bits + ((size_t)b * Q + mt0 * 32) * T;
  const int nrows = max(0, min(MTW * 32, Q - mt0 * 32));
  const int nmt = max(0, min(MTW, MT - mt0));                  // (wave-uniform) 2, 1, or 0 query tiles: a wave without any still
                                                               // carries its DMA share and the barriers
  const uint32_t sh = 4u * (uint32_t)hi5;
  auto finish = [&](uint32_t w16, uint32_t tmask) -> uint32_t {
    uint32_t x = (w16 | (w16 << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x <<= sh;
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (r[0] | r[1]) & tmask;
  };
  // all of this wave's DMA pieces except the newest `n` have landed; its own LDS accesses are complete
#define MLR_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")

  // step i multiplies tile i straight out of its ring slot: fragment ks feeds the two query tiles' MFMAs (two independent
  // accumulator chains), a few fragments ahead in registers. When step i starts this wave's outstanding DMA groups are tiles
  // i, i + 1, i + 2 (4 pieces each, in that order).
  auto step = [&](auto nmt_c, int i) {
    constexpr int NMT = decltype(nmt_c)::value;
    __builtin_amdgcn_sched_barrier(0);                         // (nothing of the previous step may sink below the wait / barrier)
    MLR_WAIT(8);                                               // my pieces of tile i have landed
    __builtin_amdgcn_s_barrier();                              // ... everybody's have; everybody is done with tile i - 1's slot
    dma_tile(i + 3);                                           // -> slot (i + 6) % 4 = tile i - 1's
    if constexpr (NMT > 0) {
      const u32x4* p = reinterpret_cast<const u32x4*>(smem_raw + MLR_RING(quad, (i + 3) & 3)) + lane;
      const int tt = tile_of(i), slot = i & (MLA_RUN - 1);
      const int valid = npix - tt * 32;
      const uint32_t tmask = valid >= 32 ? 0xffffffffu : ((1u << valid) - 1u);
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < MLA_KS; ++ks) {
        const bf16x8 bf = __builtin_bit_cast(bf16x8, p[ks * 64]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, __builtin_bit_cast(bf16x8, A[0][ks]), acc0, 0, 0, 0);
        if constexpr (NMT == 2)
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, __builtin_bit_cast(bf16x8, A[1][ks]), acc1, 0, 0, 0);
      }
      uint32_t w = 0u;
#pragma unroll
      for (int r = 15; r >= 0; --r) w = __builtin_amdgcn_alignbit(w, mla_f2u(acc0[r]), 31);
      stage[(0 * 32 + col) * MLA_RUN + slot] = finish(w, tmask);
      if constexpr (NMT == 2) {
        w = 0u;
#pragma unroll
        for (int r = 15; r >= 0; --r) w = __builtin_amdgcn_alignbit(w, mla_f2u(acc1[r]), 31);
        stage[(1 * 32 + col) * MLA_RUN + slot] = finish(w, tmask);
      }
    }
  };
  auto flush = [&](int t0, int n) {
    for (int idx = lane; idx < nrows * MLA_RUN; idx += 64) {
      const int row = idx / MLA_RUN, k = idx - row * MLA_RUN;
      if (k < n) bb[(size_t)row * T + t0 + k] = stage[idx];
    }
  };
  auto stream_tiles = [&](auto nmt_c) {
    for (int i = 0; i < per; ++i) {
      step(nmt_c, i);
      if (((i + 1) & (MLA_RUN - 1)) == 0 || i + 1 >= per) {    // (uniform) a run of MLA_RUN tiles is staged: write its rows
        const int t0 = tbeg + (i & ~(MLA_RUN - 1));
        flush(t0, min(MLA_RUN, tbeg + nvalid - t0));
      }
    }
  };
  if (nmt == 2) stream_tiles(std::integral_constant<int, 2>{});
  else if (nmt == 1) stream_tiles(std::integral_constant<int, 1>{});
  else stream_tiles(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the tail's redundant prefetches must not outlive the workgroup's LDS
#undef MLR_WAIT
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
  if (MT > 4) {
    // more than four query tiles: the LDS-ring kernel (8 waves, two stream quads per workgroup, `per` tiles per quad)
    int gr = (256 + B - 1) / B;
    const int wantr = (T + 3) / 4;                             // >= 2 tiles per quad where possible
    if (gr > wantr) gr = wantr;
    if (gr < 1) gr = 1;
    int per = (T + 2 * gr - 1) / (2 * gr);
    gr = (T + 2 * per - 1) / (2 * per);
    const size_t ldsr = (size_t)MLR_STAGE + 2 * MLR_D * (MLA_KS * 1024);
    auto kr = cgg_mask_logits_astat_ring_kernel;
    hipError_t er = hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr);
    CGG_REQUIRE(er == hipSuccess, (int)er, "cgg_mask_logits_bits_astat: cannot raise dynamic LDS to %zu", ldsr);
    hipLaunchKernelGGL(kr, dim3(gr, B), dim3(512), ldsr, (hipStream_t)stream, embed, (const u32x4*)hi, bits, Q, npix, T, MT, per);
    CGG_CHECK_LAUNCH("cgg_mask_logits_bits_astat(ring)");
    return CGG_OK;
  }
  auto kern = cgg_mask_logits_astat_kernel<4>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_mask_logits_bits_astat: cannot raise dynamic LDS to %zu", lds);
  hipLaunchKernelGGL(kern, dim3(gx, B), dim3(256), lds, (hipStream_t)stream, embed, (const u32x4*)hi, bits, Q, npix, T, MT);
  CGG_CHECK_LAUNCH("cgg_mask_logits_bits_astat");
  return CGG_OK;
}
