// Parity mode's post-attention half of an MSDeformAttn encoder layer ([3P] BaseTransformerLayer ('self_attn','norm','ffn','norm'),
// built at open_set/models/mask2former_head.py:112-117; 6 layers x 43 008 rows at configs[1]) as ONE launch on the f32-class
// f16 x 3 contraction of x3.h -- the f32 twin of cgg_encoder_layer_tail_bf16 (encoder_ffn.hip):
//
//     x1 = LayerNorm0( x + a Wo^T + bo )                      a = attention rows (f32), x = layer input rows (f32)
//     y  = LayerNorm1( x1 + W2 relu(W1 x1 + b1) + b2 )        y32 = y, yp32 = y + pos[row % pos_rows] (optional)
//
// replacing three x3 GEMM launches (256 -> 256, 256 -> 1024 + ReLU, 1024 -> 256) and two residual-LayerNorm passes whose
// f32 intermediates (projection output, x1 twice, the rows x 1024 hidden activation twice: 0.6 GB per layer at configs[1])
// never leave the chip here: HBM traffic per layer 88 MB in, 44-88 MB out.
//
// A workgroup (4 wavefronts, ONE per CU: 130 KiB of LDS) owns 64 complete rows, held in LDS as hi / lo f16 MFMA A-fragment
// images (bank-swizzled by k-step). The hidden dimension runs in chunks of 256. Wave wn computes the 64-row x 64-column block of
// every GEMM as 2 x 2 MFMA tiles: per k-step four A-fragment reads (hi, lo of two m-tiles) from LDS and four B-fragment loads
// (hi, lo of two n-tiles of the x3 image, straight from L2 into registers, EF3_PF k-steps ahead, the queue carried across
// blocks) feed TWELVE MFMAs (a bf16 kernel gets four out of the same operand traffic). The chunk's relu(. + b1) block goes back
// to LDS as split A fragments and is consumed by the second GEMM, whose accumulators persist over the chunks. x1 stays in
// registers (f32) as LayerNorm 1's residual. Epilogues: f32 tile in LDS (it overlays the hidden images) -> row-major LayerNorm,
// 16 lanes per row, DPP row reductions. v_mfma_f32_32x32x16_f16, f32 accumulation; accuracy of an f32 GEMM chain
// (tests/test_x3_gpu.py).
//
// build-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t e3_u32x4;

#define E3_C 256
#define E3_STEPS 16
#define E3_RB 64               // rows per workgroup
#define E3_NT 256              // threads per workgroup: one wavefront per 64 output columns x 64 rows
#define E3_TS 260              // f32 LayerNorm tile row stride
#define E3_PF 4                // B-fragment prefetch distance (k-steps); divides E3_STEPS (the queue rotates across blocks)
#define E3_IMG (2 * E3_STEPS * 64)          // u32x4 slots of one 64 x 256 image piece (32 KiB)

// sum over the 16 lanes of a DPP row, result in every lane: quad xor 1, quad xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float e3_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
  return v;
}

struct E3Q {                   // rotating B-fragment queue of one n-tile: hi and lo pieces
  e3_u32x4 h[E3_PF], l[E3_PF];
};

// One 64 x (32 TN) block of C += A B^T over 16 k-steps, x3 arithmetic. A-fragment images (hi at a, lo at a + lo_off) of the two
// 32-row m-tiles in LDS; B fragments of the TN 32-column n-tiles come through the rotating queues q[], which on entry hold this
// block's first E3_PF k-steps and on exit the NEXT block's (nh / nl), so the weight stream never drains at a block boundary.
// XS = true: row image, swizzled by the whole k-step (slot ^ s); XS = false: hidden image, swizzled by k-step parity (see
// the store below): a0 / a1 = even k-step lane pointers, a0o / a1o odd ones.
// TN = 2: four wavefronts per workgroup (one per SIMD), 64 columns each; TN = 1 (round 6): EIGHT wavefronts = two per SIMD, 32
// columns each -- a wave issues in order, so with one wave per SIMD its LDS reads, weight loads and epilogue VALU work all ran with
// the matrix pipe idle; with two, one wave's MFMAs cover the other's non-MFMA code (the lesson of this round's einsum kernel).
template <bool XS, int TN>
__device__ __forceinline__ void e3_block(f32x16 (&acc)[2][TN], const e3_u32x4* __restrict__ a0, const e3_u32x4* __restrict__ a1,
                                         const e3_u32x4* __restrict__ a0o, const e3_u32x4* __restrict__ a1o, int lo_off, int xl,
                                         E3Q (&q)[TN], const e3_u32x4* const (&bh)[TN], const e3_u32x4* const (&bl)[TN],
                                         const e3_u32x4* const (&nh)[TN], const e3_u32x4* const (&nl)[TN]) {
  if constexpr (XS) asm volatile("" : "+v"(xl));
  auto load_a = [&](int s, e3_u32x4& h0, e3_u32x4& l0, e3_u32x4& h1, e3_u32x4& l1) {
    if constexpr (XS) {
      const int xo = xl ^ s;
      h0 = a0[s * 64 + xo];
      l0 = a0[lo_off + s * 64 + xo];
      h1 = a0[(E3_STEPS + s) * 64 + xo];
      l1 = a0[lo_off + (E3_STEPS + s) * 64 + xo];
    } else {
      const e3_u32x4* p0 = (s & 1) ? a0o : a0;
      const e3_u32x4* p1 = (s & 1) ? a1o : a1;
      h0 = p0[s * 64];
      l0 = p0[lo_off + s * 64];
      h1 = p1[s * 64];
      l1 = p1[lo_off + s * 64];
    }
  };
  e3_u32x4 ah0, al0, ah1, al1;
  load_a(0, ah0, al0, ah1, al1);
#pragma unroll
  for (int s = 0; s < E3_STEPS; ++s) {
    e3_u32x4 bhv[TN], blv[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      bhv[t] = q[t].h[s % E3_PF];
      blv[t] = q[t].l[s % E3_PF];
    }
    const e3_u32x4 vah0 = ah0, val0 = al0, vah1 = ah1, val1 = al1;
    if (s + 1 < E3_STEPS) load_a(s + 1, ah0, al0, ah1, al1);
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      if (s + E3_PF < E3_STEPS) {
        q[t].h[s % E3_PF] = bh[t][(s + E3_PF) * 64];
        q[t].l[s % E3_PF] = bl[t][(s + E3_PF) * 64];
      } else {
        q[t].h[s % E3_PF] = nh[t][(s + E3_PF - E3_STEPS) * 64];
        q[t].l[s % E3_PF] = nl[t][(s + E3_PF - E3_STEPS) * 64];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                 // loads of the later steps issue BEFORE this step's MFMAs
#define E3_MF(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      E3_MF(val0, bhv[t], acc[0][t]);
      E3_MF(val1, bhv[t], acc[1][t]);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      E3_MF(vah0, blv[t], acc[0][t]);
      E3_MF(vah1, blv[t], acc[1][t]);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      E3_MF(vah0, bhv[t], acc[0][t]);
      E3_MF(vah1, bhv[t], acc[1][t]);
    }
#undef E3_MF
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void cgg_encoder_tail_x3_kernel(
    const float* __restrict__ a32, const float* __restrict__ x32, const CggX3W wo, const float* __restrict__ bo,
    const float* __restrict__ gamma0, const float* __restrict__ beta0, float eps0, const CggX3W w1, const float* __restrict__ b1,
    const CggX3W w2, const float* __restrict__ b2, const float* __restrict__ gamma1, const float* __restrict__ beta1, float eps1,
    const float* __restrict__ pos, int pos_rows, float* __restrict__ y32, float* __restrict__ yp32, int M, int F, int x3a,
    int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char e3_smem[];
  e3_u32x4* xfrag = reinterpret_cast<e3_u32x4*>(e3_smem);                   // row image: hi [2 m-tiles][16][64] | lo    64 KiB
  e3_u32x4* hfrag = xfrag + 2 * E3_IMG;                                      // hidden chunk: hi | lo                     64 KiB
  float* tile = reinterpret_cast<float*>(hfrag);                             // [64][E3_TS] f32 LayerNorm tile, overlays the hidden images (65 KiB)
  constexpr int NT = 64 * NWV;                                               // threads per workgroup
  constexpr int TN = 8 / NWV;                                                // 32-column n-tiles per wave (2: four waves, 1: eight)
  constexpr int RW = E3_RB / NWV;                                            // rows per wave in the row-major (LayerNorm) phases
  constexpr int NIT = RW / 4;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int m0 = blockIdx.x * E3_RB;
  const int nchunk = F >> 8;
  const int KS2 = F >> 4;                                                    // k-steps of W2

  // weight stream: the first E3_PF k-steps of the output projection are in flight while the rows are staged
  E3Q q[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int s = 0; s < E3_PF; ++s) {
      q[t].h[s] = wo.hi[((size_t)(TN * wn + t) * E3_STEPS + s) * 64 + lane];
      q[t].l[s] = wo.lo[((size_t)(TN * wn + t) * E3_STEPS + s) * 64 + lane];
    }
  // ---- attention rows -> split A-fragment images: 32-byte piece (row, k8) = 8 consecutive channels -> slot (mt, k-step = k8 / 2,
  //      (row % 32 + 32 (k8 & 1)) ^ k-step): the XOR spreads a row's pieces over all bank groups ----
  //      All 16 loads of a thread are issued before the first split: the rolled loop the compiler made of the one-piece-at-a-time
  //      form waited for every pair of loads (8 serial memory latencies at the head of every workgroup).
  {
    constexpr int NP = E3_RB * 32 / NT;
    f32x4 v0[NP], v1[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 5, k8 = p & 31;
      const int mc = m0 + row < M ? m0 + row : M - 1;
      const float* src = a32 + (size_t)mc * E3_C + 8 * k8;
      v0[i] = *reinterpret_cast<const f32x4*>(src);
      v1[i] = *reinterpret_cast<const f32x4*>(src + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 5, k8 = p & 31;
      e3_u32x4 h, l;
      cgg_x3_split8(v0[i], v1[i], h, l);
      const int slot = ((row >> 5) * E3_STEPS + (k8 >> 1)) * 64 + (((row & 31) + 32 * (k8 & 1)) ^ (k8 >> 1));
      xfrag[slot] = h;
      xfrag[E3_IMG + slot] = l;
    }
  }
  // the layer-input rows LayerNorm 0 adds (its residual) are requested now: they arrive behind the output projection's MFMAs
  const int sub = lane & 15, rsub = lane >> 4;
  f32x4 xr[NIT][4];                                     // residual rows (later: x1, LayerNorm 1's residual), f32
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = RW * wn + 4 * it + rsub;
    const int mc = m0 + row < M ? m0 + row : M - 1;
    if (x3a) {
      // x3a rows (csrc/x3.h): channels c0 = 4 sub + 64 k .. + 3 are half (sub & 1) of the group c0 / 8 = [8 hi | 8 lo]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint2* g = reinterpret_cast<const uint2*>(x32 + (size_t)mc * E3_C + 8 * (sub >> 1) + 64 * k) + (sub & 1);
        cgg_x3a_decode4(g[0], g[2], xr[it][k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) xr[it][k] = *reinterpret_cast<const f32x4*>(x32 + (size_t)mc * E3_C + 4 * sub + 64 * k);
    }
  }
  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  __syncthreads();
  // ---- output projection: a Wo^T (next block in the weight stream: W1's first chunk) ----
  {
    const e3_u32x4 *bh[TN], *bl[TN], *nh[TN], *nl[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const size_t o = ((size_t)(TN * wn + t) * E3_STEPS) * 64 + lane;
      bh[t] = wo.hi + o;
      bl[t] = wo.lo + o;
      nh[t] = w1.hi + o;
      nl[t] = w1.lo + o;
    }
    e3_block<true, TN>(acc, xfrag, xfrag, xfrag, xfrag, E3_IMG, lane, q, bh, bl, nh, nl);
  }
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int col = 32 * TN * wn + 32 * nt + j;
    const float cs = wo.scale[col], bias = bo[col];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi5) * E3_TS + col] = acc[mt][nt][r] * cs + bias;
  }
  __syncthreads();                                     // the tile is complete, every wave is done with the attention-row image
  // ---- LayerNorm 0 over x + attn_out, 16 lanes per row; x1 stays in registers (f32) and goes to the row image as split pieces ----
  {
    uint2* xh2 = reinterpret_cast<uint2*>(xfrag);
    uint2* xl2 = reinterpret_cast<uint2*>(xfrag + E3_IMG);
    constexpr float inv_n = 1.f / (float)E3_C;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = RW * wn + 4 * it + rsub;
      f32x4 v[4];
      float sm = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = *reinterpret_cast<const f32x4*>(&tile[row * E3_TS + 4 * sub + 64 * k]) + xr[it][k];
        sm += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
      }
      sm = e3_row16_sum(sm);
      const float mean = sm * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = v[k] - mean;
        q += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
      }
      q = e3_row16_sum(q);
      const float rstd = rsqrtf(q * inv_n + eps0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma0 + 4 * sub + 64 * k);
        const f32x4 be = *reinterpret_cast<const f32x4*>(beta0 + 4 * sub + 64 * k);
        const f32x4 y = v[k] * rstd * g + be;
        xr[it][k] = y;
        // columns c0 = 4 sub + 64 k .. + 3 of the row are half of the 16-byte slot (k-step c0 / 16, half (c0 / 8) & 1)
        const int ks = (sub >> 2) + 4 * k;
        const int slot = ((row >> 5) * E3_STEPS + ks) * 64 + (((row & 31) + 32 * ((sub >> 1) & 1)) ^ ks);
        uint2 h, l;
        cgg_x3_split4(y, h, l);
        xh2[2 * slot + (sub & 1)] = h;
        xl2[2 * slot + (sub & 1)] = l;
      }
    }
  }
  f32x16 acc2[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
  __syncthreads();                                     // x1 image complete; the tile (hidden region) is free again

  // hidden image, bank-swizzled: slot (k-step, half, row) sits at k-step * 64 + half * 32 + (row ^ 2 (k-step & 1) ^ 8 half), so the
  // stores of one instruction fall into distinct banks (layout of encoder_ffn.hip); hi and lo images share the addressing
  const int hoff_e = hi5 * 32 + (j ^ (8 * hi5)), hoff_o = hi5 * 32 + (j ^ 2 ^ (8 * hi5));
  const e3_u32x4* ha0 = hfrag + hoff_e;
  const e3_u32x4* ha1 = hfrag + E3_STEPS * 64 + hoff_e;
  const e3_u32x4* ha0o = hfrag + hoff_o;
  const e3_u32x4* ha1o = hfrag + E3_STEPS * 64 + hoff_o;
  uint32_t* h32 = reinterpret_cast<uint32_t*>(hfrag);
  const int odd = j & 1, k1 = (j >> 4) & 1, half = (j >> 3) & 1;
  int hb[2][2];                                        // word index of this lane's store for (bit 1, bit 3) of the register row
#pragma unroll
  for (int X = 0; X < 2; ++X)
#pragma unroll
    for (int Y = 0; Y < 2; ++Y)
      hb[X][Y] = (((2 * TN * wn + k1) * 64 + half * 32 + 2 * (X ^ k1) + 8 * (Y ^ half) + 4 * hi5 + odd) << 2) + ((j & 7) >> 1);
  const uint32_t rot = 16u * (uint32_t)odd;
  float hmax = 0.f;                                    // largest hidden activation: beyond the f16 x 3 range it raises the overflow flag
  for (int c = 0; c < nchunk; ++c) {
    // ---- GEMM 1: hidden columns 256 c + 32 TN wn .. (n-tiles 8 c + TN wn .. of W1); next in the stream: this chunk's W2 slice ----
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const e3_u32x4 *w1h[TN], *w1l[TN], *w2h[TN], *w2l[TN], *n1h[TN], *n1l[TN];
    const int cn = c + 1 < nchunk ? c + 1 : 0;         // last chunk: the queue refills with chunk 0 again (unused)
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const size_t g2 = ((size_t)(TN * wn + t) * KS2 + 16 * c) * 64 + lane;
      const size_t g1 = ((size_t)(8 * c + TN * wn + t) * E3_STEPS) * 64 + lane;
      const size_t gn = ((size_t)(8 * cn + TN * wn + t) * E3_STEPS) * 64 + lane;
      w1h[t] = w1.hi + g1;
      w1l[t] = w1.lo + g1;
      w2h[t] = w2.hi + g2;
      w2l[t] = w2.lo + g2;
      n1h[t] = w1.hi + gn;
      n1l[t] = w1.lo + gn;
    }
    e3_block<true, TN>(acc, xfrag, xfrag, xfrag, xfrag, E3_IMG, lane, q, w1h, w1l, w2h, w2l);
    // relu(. + b1) -> split A-fragment images of the chunk; column (64 wn + 32 nt + j) of the chunk = k index of GEMM 2. Lanes j,
    // j ^ 1 hold neighbouring columns: per register pair they swap one value, the even lane then owns row(2 rp), the odd lane
    // row(2 rp + 1), and each stores one 32-bit word per image
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int hc = 256 * c + 32 * TN * wn + 32 * nt + j;
      const float cs1 = w1.scale[hc], bias1 = b1[hc];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
          const float v0 = fmaxf(acc[mt][nt][2 * rp] * cs1 + bias1, 0.f), v1 = fmaxf(acc[mt][nt][2 * rp + 1] * cs1 + bias1, 0.f);
          hmax = fmaxf(hmax, fmaxf(v0, v1));
          const float kept = odd ? v1 : v0, sent = odd ? v0 : v1;
          const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sent), 0xB1, 0xf, 0xf, true));  // lane ^ 1
          uint32_t ph, pl;
          cgg_x3_split2(kept * CGG_X3_ASCALE, recv * CGG_X3_ASCALE, ph, pl);
          const int o = hb[rp & 1][(rp >> 1) & 1] + (((mt * E3_STEPS + 2 * nt) * 64 + 16 * (rp >> 2)) << 2);
          h32[o] = __builtin_amdgcn_alignbit(ph, ph, rot);              // odd lanes: (recv, kept)
          h32[4 * E3_IMG + o] = __builtin_amdgcn_alignbit(pl, pl, rot);
        }
      }
    }
    __syncthreads();                                   // the chunk's hidden block is complete
    // ---- GEMM 2: output columns 32 TN wn .. over the chunk's 256 hidden units (k-steps 16 c .. of W2); next: W1's next chunk ----
    e3_block<false, TN>(acc2, ha0, ha1, ha0o, ha1o, E3_IMG, 0, q, w2h, w2l, n1h, n1l);
    __syncthreads();                                   // hfrag is rewritten by the next chunk (and by the tile below)
  }

  if (flag && !(hmax * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(flag, 1);
  // ---- f32 block (* colscale + b2) -> LDS tile; the hidden images are dead ----
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int col = 32 * TN * wn + 32 * nt + j;
    const float cs2 = w2.scale[col], bias2 = b2[col];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi5) * E3_TS + col] = acc2[mt][nt][r] * cs2 + bias2;
  }
  // the pos rows of this wave's 16 output rows are requested before the barrier (unpredicated: rows past M use row M - 1) and arrive
  // under the LayerNorm reductions; loaded inside the store loop they were two loads + s_waitcnt vmcnt(0) per 4 stores
  f32x4 pp[NIT][4];
  if (yp32) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + RW * wn + 4 * it + rsub;
      const float* prow = pos + (size_t)((m < M ? m : M - 1) % pos_rows) * E3_C;
#pragma unroll
      for (int k = 0; k < 4; ++k) pp[it][k] = *reinterpret_cast<const f32x4*>(prow + 4 * sub + 64 * k);
    }
  }
  __syncthreads();
  // ---- row-major LayerNorm of x1 + ffn(x1): wave w owns rows 16 w .. 16 w + 15, four at a time; 16 lanes share a row ----
  f32x4 g4[4], be4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    g4[k] = *reinterpret_cast<const f32x4*>(gamma1 + 4 * sub + 64 * k);
    be4[k] = *reinterpret_cast<const f32x4*>(beta1 + 4 * sub + 64 * k);
  }
  constexpr float inv_n = 1.f / (float)E3_C;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = RW * wn + 4 * it + rsub, m = m0 + row;
    f32x4 v[4];
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = *reinterpret_cast<const f32x4*>(&tile[row * E3_TS + 4 * sub + 64 * k]) + xr[it][k];
      sm += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    sm = e3_row16_sum(sm);
    const float mean = sm * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = v[k] - mean;
      q += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
    }
    q = e3_row16_sum(q);
    const float rstd = rsqrtf(q * inv_n + eps1);
    if (m >= M) continue;
    if (x3a) {
      // outputs as x3a rows: the next layer's GEMM operands / residual stream in the form they are consumed
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 y = v[k] * rstd * g4[k] + be4[k];
        const size_t o = (size_t)m * E3_C + 8 * (sub >> 1) + 64 * k;
        uint2 h, l;
        cgg_x3_split4(y, h, l);
        uint2* yo = reinterpret_cast<uint2*>(y32 + o) + (sub & 1);
        yo[0] = h;
        yo[2] = l;
        float am = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
        if (yp32) {
          const f32x4 yp = y + pp[it][k];
          cgg_x3_split4(yp, h, l);
          uint2* po = reinterpret_cast<uint2*>(yp32 + o) + (sub & 1);
          po[0] = h;
          po[2] = l;
          am = fmaxf(am, fmaxf(fmaxf(fabsf(yp[0]), fabsf(yp[1])), fmaxf(fabsf(yp[2]), fabsf(yp[3]))));
        }
        if (flag && !(am * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(flag, 1);
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 y = v[k] * rstd * g4[k] + be4[k];
      const size_t o = (size_t)m * E3_C + 4 * sub + 64 * k;
      *reinterpret_cast<f32x4*>(y32 + o) = y;
      if (yp32) *reinterpret_cast<f32x4*>(yp32 + o) = y + pp[it][k];
    }
  }
}

int* cgg_x3_overflow_flag_ptr();       // x3s_gemm.hip

// CGG_TAIL_WAVES=4: round 3-5's four-wavefront form (A/B); read once, when the library is loaded
static const bool e3_waves8 = !(getenv("CGG_TAIL_WAVES") && atoi(getenv("CGG_TAIL_WAVES")) == 4);

static int e3_launch(const float* a32, const float* x32, const void* wo_x3, const float* bo, const float* gamma0,
                     const float* beta0, float eps0, const void* w1_x3, const float* b1, const void* w2_x3,
                     const float* b2, const float* gamma1, const float* beta1, float eps1, const float* pos,
                     int pos_rows, float* y32, float* yp32, int M, int C, int F, int x3a, cgg_stream_t stream, const char* who) {
  CGG_REQUIRE(a32 && x32 && wo_x3 && bo && gamma0 && beta0 && w1_x3 && b1 && w2_x3 && b2 && gamma1 && beta1 && y32, CGG_EINVAL,
              "%s: null pointer", who);
  CGG_REQUIRE(C == E3_C, CGG_EUNSUPPORTED, "%s: C=%d (only 256 is built)", who, C);
  CGG_REQUIRE(M > 0 && F > 0 && F % 256 == 0, CGG_EUNSUPPORTED, "%s: F=%d must be a multiple of 256", who, F);
  CGG_REQUIRE(!yp32 || (pos && pos_rows > 0), CGG_EINVAL, "%s: yp32 needs pos", who);
  CGG_REQUIRE(cgg_aligned16(a32) && cgg_aligned16(x32) && cgg_aligned16(wo_x3) && cgg_aligned16(w1_x3) && cgg_aligned16(w2_x3) &&
                  cgg_aligned16(gamma0) && cgg_aligned16(beta0) && cgg_aligned16(gamma1) && cgg_aligned16(beta1) &&
                  (!pos || cgg_aligned16(pos)) && cgg_aligned16(y32) && (!yp32 || cgg_aligned16(yp32)),
              CGG_EALIGN, "%s: 16-B alignment", who);
  const size_t lds = (size_t)2 * E3_IMG * 16 + (size_t)E3_RB * E3_TS * sizeof(float);      // row images + tile (>= hidden images)
  // the attribute is per device (a process-wide flag broke launches on a second device)
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)cgg_encoder_tail_x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)cgg_encoder_tail_x3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CGG_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise dynamic LDS to %zu", who, lds);
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
#define E3_GO(NWV)                                                                                                               \
  hipLaunchKernelGGL(cgg_encoder_tail_x3_kernel<NWV>, dim3((M + E3_RB - 1) / E3_RB), dim3(64 * NWV), lds, (hipStream_t)stream, a32, x32, \
                     cgg_x3_view(wo_x3, E3_C, E3_C), bo, gamma0, beta0, eps0, cgg_x3_view(w1_x3, F, E3_C), b1,                  \
                     cgg_x3_view(w2_x3, E3_C, F), b2, gamma1, beta1, eps1, pos, pos_rows, y32, yp32, M, F, x3a,                 \
                     cgg_x3_overflow_flag_ptr())
  if (e3_waves8) E3_GO(8);
  else E3_GO(4);
#undef E3_GO
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_encoder_layer_tail_x3(const float* a32, const float* x32, const void* wo_x3, const float* bo, const float* gamma0,
                                         const float* beta0, float eps0, const void* w1_x3, const float* b1, const void* w2_x3,
                                         const float* b2, const float* gamma1, const float* beta1, float eps1, const float* pos,
                                         int pos_rows, float* y32, float* yp32, int M, int C, int F, cgg_stream_t stream) {
  return e3_launch(a32, x32, wo_x3, bo, gamma0, beta0, eps0, w1_x3, b1, w2_x3, b2, gamma1, beta1, eps1, pos, pos_rows, y32, yp32, M,
                   C, F, 0, stream, "cgg_encoder_layer_tail_x3");
}

// round 4: the layer input x and the outputs y / y + pos are x3a rows (csrc/x3.h); the attention rows a32 stay f32
extern "C" int cgg_encoder_layer_tail_x3a(const float* a32, const void* x_x3a, const void* wo_x3, const float* bo,
                                          const float* gamma0, const float* beta0, float eps0, const void* w1_x3, const float* b1,
                                          const void* w2_x3, const float* b2, const float* gamma1, const float* beta1, float eps1,
                                          const float* pos, int pos_rows, void* y_x3a, void* yp_x3a, int M, int C, int F,
                                          cgg_stream_t stream) {
  return e3_launch(a32, (const float*)x_x3a, wo_x3, bo, gamma0, beta0, eps0, w1_x3, b1, w2_x3, b2, gamma1, beta1, eps1, pos, pos_rows,
                   (float*)y_x3a, (float*)yp_x3a, M, C, F, 1, stream, "cgg_encoder_layer_tail_x3a");
}
