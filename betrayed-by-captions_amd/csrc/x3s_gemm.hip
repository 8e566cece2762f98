// Parity mode's large contractions on PRE-SPLIT activation rows (round 4): C[M, N] = act(A W^T * colscale + bias (+ residual)),
// A in the "x3a" storage of x3.h (per 8 channels: 8 f16 hi | 8 f16 lo of 16 a; same bytes and addressing as f32), W an x3 image,
// f32-class f16 x 3 arithmetic (three v_mfma_f32_32x32x16_f16 per fragment pair into one f32 accumulator). Serves
//   * the linears of the pixel decoder / K-V projections ([3P] MSDeformAttnPixelDecoder, nn.MultiheadAttention in_proj; built at
//     open_set/models/mask2former_head.py:112-118, called :787, :829-840): A = (M, K) rows, and
//   * every convolution of the f32 path after the stem as an IMPLICIT GEMM over a channel-last x3a map (1 x 1 / 3 x 3, stride
//     1 / 2: BN-folded ResNet, the pixel decoder's input / lateral / output / mask-feature convolutions).
// It replaces cgg_gemm_x3_kernel (x3_gemm.hip: f32 rows, split in the loop) on the inference stream. Round 3's kernel was
// issue-bound -- ~230 instructions per 24 MFMAs, 48 of them the f32 -> (hi, lo) conversion repeated by every column tile, plus
// register staging and ds_writes of both operands. Here the loop has NO conversion and NO staging registers:
//
//   * both operands go HBM / L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction). An A row's
//     32-channel chunk is one 128-byte line [h0 l0 h1 l1 h2 l2 h3 l3] (16-byte units); 8 lanes fetch one line, a wave instruction
//     8 rows. The LDS image is row-major with the unit index XOR (row >> 1) & 7 -- applied on the SOURCE side (a lane fetches unit
//     (lane & 7) ^ swz(row); LDS-DMA writes lane-linear) and on the fragment read, so the four 16-lane groups of a ds_read_b128
//     hit 16 different 16-byte bank slots. W is the x3 image (fragment order), copied linearly.
//   * S LDS stages (2-4 by tile shape), DMA S - 1 chunks ahead, ONE raw s_barrier per 32-deep chunk behind a counted
//     `s_waitcnt vmcnt(N)` that leaves the younger chunks in flight across the barrier.
//   * workgroup tiles from 256 x 256 (8 wavefronts as 2 x 4, each 4 x 2 MFMA tiles: 12 fragment reads feed 24 MFMAs per k-step)
//     down to 64 x 64, picked by problem shape.
//   * conv: a 32-channel chunk lies inside one filter tap; per lane and row a tap-validity bit mask, out-of-map taps are sent to an
//     out-of-range buffer offset, which the LDS-DMA answers with zeros (checked on MI355X, tests/test_x3s_gpu.py).
//   * epilogue: accumulator tile -> per-wave LDS scratch (XOR-swizzled, conflict-free) -> each lane owns 8 consecutive columns of a
//     row: residual read (f32 or x3a), ReLU, then 32 contiguous bytes out -- 8 floats or the x3a group [8 hi | 8 lo] -- so the
//     next GEMM's A operand is produced in the form it is consumed. |16 v| > 65504 (f16 overflow of a stored value) raises
//     the device-side overflow flag instead of silently storing inf.
#include <mutex>
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void* xs_lds_t;

struct XsConv {
  int H, W, C, OH, OW, KW, stride, pad, taps;      // channel-last input [B][H][W][C]; K = taps * C
};

struct XsArgs {
  const void* A;                  // x3a rows
  int lda;                        // row stride in 4-byte elements (GEMM form)
  int a_rpb;                      // GEMM form, > 0: row m lives in image m / a_rpb at element offset (m / a_rpb) a_bstride + (m % a_rpb) lda
  uint32_t a_bstride;
  CggX3W w;
  const float* bias;
  const void* res;                // residual [M or res_mod rows][N], format res_fmt
  int ldr, res_fmt, res_mod;      // res_fmt: 0 none, 1 f32, 2 x3a
  void* out;
  int ldc, out_fmt;               // 1 f32, 2 x3a
  int M, N, K, relu, tiles_n, n_tiles32;
  XsConv cv;
  uint32_t a_bytes, w_bytes;
  int* flag;                      // overflow flag word (x3a output)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xs_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int N>
__device__ __forceinline__ void xs_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void xs_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#define XS_OOB 0xFFFFFF00u

struct XsTap {                    // position of a DMA stream: chunk, channel offset and filter tap (file scope: a struct local to the
  int c, c0, tap, ky, kx;         // __global__ template made hipcc 7.2 drop the kernel's host stubs)
};

// ---- epilogue of one wave. Everything in the PRE-SCALED domain (16 v): cs = 16 colscale, bs = 16 bias; an x3a residual is
//      already pre-scaled, an f32 one is multiplied. The accumulator tiles of one n-tile column go to the wave's LDS scratch
//      (unrolled: register arrays), then ONE rolled loop finishes them tile by tile -- residual, ReLU, split, 32-byte stores -- with
//      runtime (wave-uniform) format branches: the kernel stays a few KB of code. (The first version instantiated the whole
//      epilogue per format and tile, 70-150 KB per kernel: inside the step every launch then started with instruction-cache
//      misses to HBM, +10 us per launch over the stand-alone timing.) No predicates: rows >= M fall outside the buffer
//      descriptors (stores dropped, loads 0), columns >= N are sent there explicitly. ----
struct XsEpi {
  __amdgpu_buffer_rsrc_t orsrc, rrsrc;
  float lo_clamp;
  int mr0;              // residual row of (tile row lane >> 3, pass 0, m-tile 0), reduced mod res_mod
};

// mask ? b : a with mask = 0 or ~0 (v_bfi_b32)
__device__ __forceinline__ uint32_t xs_sel(uint32_t a, uint32_t b, uint32_t mask) { return (b & mask) | (a & ~mask); }

// 8 bytes of lane ^ 1 (the other half of the lane pair that shares an 8-column group)
__device__ __forceinline__ uint2 xs_pair_swap(const uint2 v) {
  uint2 r;
  r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0xB1, 0xf, 0xf, true);      // quad_perm [1, 0, 3, 2]
  r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0xB1, 0xf, 0xf, true);
  return r;
}

// Finish the TM scratch tiles of one 32-column n-tile (rows mrow0 .. mrow0 + 32 TM - 1). Lane -> (row r8 + 8 pass, 16-byte piece q of
// the row's 128 output bytes): 8 lanes cover one full line, a wave instruction 8 full lines -- every residual load and every store
// is ONE 16-byte access per lane. A lane OWNS the 4 columns 4 q .. 4 q + 3 (half of the 8-column group q >> 1; its partner lane ^ 1
// the other half): it finishes only those 4 values. In x3a form the group's memory image is [8 hi | 8 lo]: the even lane stores the
// hi piece, the odd lane the lo piece, so the pair swaps 8 bytes by DPP -- each lane sends the half-piece it does not store and
// receives the one it does (same swap, other direction, for an x3a residual). Specialised on the formats (the wave-uniform format
// tests of the first version sat between the loads and serialised them); the tile loop is rolled, the next tile's residual pieces
// are requested before the current tile is finished.
// the lane's 4 residual pieces of one 32 x 32 tile: rows mres + 8 pass (mres already reduced mod res_mod; advanced to the next tile)
__device__ __forceinline__ void xs_load_res(u32x4 (&rv)[4], const XsArgs& p, const __amdgpu_buffer_rsrc_t rrsrc, int& mres, int ncol0,
                                            int lane) {
  const int q = lane & 7;
  const bool cok = ncol0 + 4 * q < p.N;
  const uint32_t coff = (uint32_t)(ncol0 * 4 + 16 * q);
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    int mr = mres + 8 * ps;
    if (p.res_mod) {
      mr = mr >= p.res_mod ? mr - p.res_mod : mr;
      if (p.res_mod < 32) mr %= p.res_mod;                     // tiny periods only (tests)
    }
    rv[ps] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, cok ? (uint32_t)(mr * p.ldr) * 4u + coff : XS_OOB, 0, 0);
  }
  mres += 32;
  if (p.res_mod) mres = p.res_mod >= 32 ? (mres >= p.res_mod ? mres - p.res_mod : mres) : mres % p.res_mod;
}

template <int RES, int OUT>
__device__ __forceinline__ float xs_finish_column(const float* scratch, int TM_, const XsArgs& p, const XsEpi& e, int mrow0, int ncol0,
                                                  int lane, float amax, const u32x4 (&rv0)[4], bool have_rv0) {
  const int r8 = lane >> 3, q = lane & 7;
  const uint32_t om = 0u - (uint32_t)(q & 1);                  // all ones in the odd lane of a pair
  const bool cok = ncol0 + 4 * q < p.N;                        // N % 8 == 0: both lanes of a pair agree
  const uint32_t coff = (uint32_t)(ncol0 * 4 + 16 * q);        // byte offset of the lane's piece inside a row
  int mres = e.mr0;                                            // residual row of (tile 0, pass 0), already reduced mod res_mod
  auto load_res = [&](u32x4 (&rv)[4]) { xs_load_res(rv, p, e.rrsrc, mres, ncol0, lane); };
  u32x4 rv[4], rn[4];
  if (RES) {
    if (have_rv0) {                                            // tile 0 of the first column was requested before the main loop
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) rv[ps] = rv0[ps];
      mres += 32;
      if (p.res_mod) mres = p.res_mod >= 32 ? (mres >= p.res_mod ? mres - p.res_mod : mres) : mres % p.res_mod;
    } else {
      load_res(rv);
    }
  }
#pragma unroll 1
  for (int mt = 0; mt < TM_; ++mt) {
    const float* sc = scratch + mt * 1024;
    if (RES && mt + 1 < TM_) load_res(rn);
    f32x4 v[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rr = r8 + 8 * ps;
      v[ps] = *reinterpret_cast<const f32x4*>(sc + rr * 32 + (q ^ ((rr >> 1) & 1)) * 4);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rr = r8 + 8 * ps;
      f32x4 w = v[ps];
      if (RES == 1) {
        w += __builtin_bit_cast(f32x4, rv[ps]) * CGG_X3_ASCALE;
      } else if (RES == 2) {
        // the lane holds hi[0..7] (even) or lo[0..7] (odd) of the group; its own 4 values need the other piece's matching half
        // (bit-field selects on scalars: `odd ? rv[ps][2] : rv[ps][0]` was compiled into a dynamic register index -- a chain of
        // 15 compare / select pairs per element, 20 % of the kernel's time on the ResNet's residual convolutions)
        const uint32_t r0 = rv[ps][0], r1 = rv[ps][1], r2 = rv[ps][2], r3 = rv[ps][3];
        const uint2 keep = {xs_sel(r0, r2, om), xs_sel(r1, r3, om)};                             // my columns of my piece
        const uint2 send = {xs_sel(r2, r0, om), xs_sel(r3, r1, om)};                             // the partner's columns of my piece
        const uint2 got = xs_pair_swap(send);                                                    // my columns of the partner's piece
        const uint2 hh = {xs_sel(keep.x, got.x, om), xs_sel(keep.y, got.y, om)};
        const uint2 ll = {xs_sel(got.x, keep.x, om), xs_sel(got.y, keep.y, om)};
        const cgg_f32x2 h0 = __builtin_convertvector(__builtin_bit_cast(f16x2, hh.x), cgg_f32x2);
        const cgg_f32x2 h1 = __builtin_convertvector(__builtin_bit_cast(f16x2, hh.y), cgg_f32x2);
        const cgg_f32x2 l0 = __builtin_convertvector(__builtin_bit_cast(f16x2, ll.x), cgg_f32x2);
        const cgg_f32x2 l1 = __builtin_convertvector(__builtin_bit_cast(f16x2, ll.y), cgg_f32x2);
        w += f32x4{h0[0] + l0[0], h0[1] + l0[1], h1[0] + l1[0], h1[1] + l1[1]};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = fmaxf(w[k], e.lo_clamp);
      const uint32_t oo = cok ? (uint32_t)((mrow0 + 32 * mt + rr) * p.ldc) * 4u + coff : XS_OOB;
      u32x4 o;
      if (OUT == 2) {
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fmaxf(fabsf(w[0]), fabsf(w[1])), __builtin_fmaxf(fabsf(w[2]), fabsf(w[3]))));
        uint2 h, l;
        cgg_x3_split2(w[0], w[1], h.x, l.x);
        cgg_x3_split2(w[2], w[3], h.y, l.y);
        const uint2 got = xs_pair_swap(uint2{xs_sel(l.x, h.x, om), xs_sel(l.y, h.y, om)});      // even sends lo, odd sends hi
        o = u32x4{xs_sel(h.x, got.x, om), xs_sel(h.y, got.y, om), xs_sel(got.x, l.x, om), xs_sel(got.y, l.y, om)};
      } else {
        o = __builtin_bit_cast(u32x4, w * CGG_X3_INV_ASCALE);
      }
      __builtin_amdgcn_raw_buffer_store_b128(o, e.orsrc, oo, 0, 0);
    }
    if (RES) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) rv[ps] = rn[ps];
    }
  }
  return amax;
}

__device__ __forceinline__ XsEpi xs_epi_setup(const XsArgs& p, int mrow0, int lane) {
  XsEpi e;
  e.orsrc = xs_rsrc(p.out, (uint32_t)(((size_t)(p.M - 1) * p.ldc + p.N) * 4));
  const int res_rows = p.res_mod ? p.res_mod : p.M;
  e.rrsrc = xs_rsrc(p.res_fmt ? p.res : p.out, p.res_fmt ? (uint32_t)(((size_t)(res_rows - 1) * p.ldr + p.N) * 4) : 0u);
  e.lo_clamp = p.relu ? 0.f : -__builtin_inff();
  e.mr0 = mrow0 + (lane >> 3);
  if (p.res_fmt && p.res_mod) e.mr0 %= p.res_mod;
  return e;
}

template <int TM, int TN>
__device__ __forceinline__ float xs_epilogue(const f32x16 (&acc)[TM][TN], float* scratch, const XsArgs& p, const XsEpi& e, int mrow0,
                                             int ntile0, int lane, const u32x4 (&rv0)[4]) {
  const int j = lane & 31, hi5 = lane >> 5;
  const int fmt = p.res_fmt * 2 + (p.out_fmt - 1);             // wave-uniform
  float amax = 0.f;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int ncol = (ntile0 + nt) * 32 + j;
    const bool nok = ncol < p.N;
    const float cs = nok ? p.w.scale[ncol] * CGG_X3_ASCALE : 0.f;
    const float bs = (nok && p.bias) ? p.bias[ncol] * CGG_X3_ASCALE : 0.f;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi5;
        scratch[mt * 1024 + rr * 32 + (((j >> 2) ^ ((rr >> 1) & 1)) << 2) + (j & 3)] = acc[mt][nt][r] * cs + bs;
      }
    }
    const int ncol0 = (ntile0 + nt) * 32;
    switch (fmt) {
      case 0: amax = xs_finish_column<0, 1>(scratch, TM, p, e, mrow0, ncol0, lane, amax, rv0, false); break;
      case 1: amax = xs_finish_column<0, 2>(scratch, TM, p, e, mrow0, ncol0, lane, amax, rv0, false); break;
      case 2: amax = xs_finish_column<1, 1>(scratch, TM, p, e, mrow0, ncol0, lane, amax, rv0, nt == 0); break;
      case 3: amax = xs_finish_column<1, 2>(scratch, TM, p, e, mrow0, ncol0, lane, amax, rv0, nt == 0); break;
      case 4: amax = xs_finish_column<2, 1>(scratch, TM, p, e, mrow0, ncol0, lane, amax, rv0, nt == 0); break;
      default: amax = xs_finish_column<2, 2>(scratch, TM, p, e, mrow0, ncol0, lane, amax, rv0, nt == 0); break;
    }
  }
  return amax;
}

// Workgroup = KG k-groups x (WM x WN) wavefronts; tile (32 TM WM) x (32 TN WN); every loop iteration consumes KG chunks of 32
// channels (k-group g computes chunk KG it + g into its own accumulators; they are summed through LDS before the epilogue: the
// intra-workgroup split-K that gives the deep, small-M shapes two waves per SIMD). SA / SB = LDS ring slots of the A / B operand
// (a slot = the KG chunks of one iteration); the DMA runs SA - 1 / SB - 1 iterations ahead.
template <bool CONV, int TM, int TN, int WM, int WN, int KG, int SA, int SB>
__global__ __launch_bounds__(64 * WM * WN * KG) void cgg_gemm_x3s_kernel(const XsArgs p) {
  constexpr int NWG = WM * WN, NW = NWG * KG, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int A_SLOT = KG * BM * 128, B_SLOT = KG * BN * 128;
  constexpr int NA = BM / 8 / NWG, NB = BN / 8 / NWG;        // LDS-DMA instructions per wave and iteration
  constexpr int DA = SA - 1, DB = SB - 1;
  constexpr int LDS = SA * A_SLOT + SB * B_SLOT;
  static_assert(BM % (8 * NWG) == 0 && BN % (8 * NWG) == 0, "tile rows / columns must split evenly over the waves' DMA pieces");
  static_assert(SA >= SB && SB >= 2, "A runs at least as far ahead as B");
  static_assert(LDS >= NWG * TM * 4096 + (KG - 1) * NWG * TM * TN * 4096, "LDS: the rings must cover the reduce buffer + epilogue scratch");
  static_assert(KG == 1 || KG == 2, "one or two k-groups");
  // allowed outstanding DMA instructions at the top of an iteration (issue order per iteration: B pieces, then A pieces)
  constexpr int WAIT = (DA > DB ? NA : 0) + (DB - 1) * (NA + NB);
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hi5 = lane >> 5;
  const int kg = wave / NWG, wg = wave - kg * NWG;
  const int wm = wg / WN, wn = wg - wm * WN;
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, nt0 = tile_n * (BN / 32);
  const int KS = p.K >> 4, nchunk = p.K >> 5;
  const int niter = (nchunk + KG - 1) / KG;

  // ---- A pieces of this lane: piece q = wave + NW i of the iteration's [k-group][BM / 8] x 1 KiB; lane -> (row, 16-byte unit) ----
  const __amdgpu_buffer_rsrc_t arsrc = xs_rsrc(p.A, p.a_bytes);
  uint32_t arow[NA];             // byte offset of the lane's unit at chunk 0 (CONV: at filter tap (0, 0); may wrap below 0)
  uint32_t amask[NA];            // CONV: bit t <=> tap t of the row's output pixel lies inside the map
  int ag[NA];                    // k-group of the piece (wave-uniform)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = wave + NW * i;
    ag[i] = KG == 1 ? 0 : q / (BM / 8);
    const int r = (q - ag[i] * (BM / 8)) * 8 + (lane >> 3);
    const int u = (lane & 7) ^ ((r >> 1) & 7);
    const int m = m0 + r;
    const int mc = m < p.M ? m : p.M - 1;
    if constexpr (CONV) {
      const int ohw = p.cv.OH * p.cv.OW;
      const int b = mc / ohw, rr = mc - b * ohw;
      const int oy = rr / p.cv.OW, ox = rr - oy * p.cv.OW;
      const int iy0 = oy * p.cv.stride - p.cv.pad, ix0 = ox * p.cv.stride - p.cv.pad;
      arow[i] = (uint32_t)(((b * p.cv.H + iy0) * p.cv.W + ix0) * p.cv.C) * 4u + 16u * u;
      uint32_t mk = 0;
      const int KH = p.cv.taps / p.cv.KW;
      for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < p.cv.KW; ++kx) {
          const int iy = iy0 + ky, ix = ix0 + kx;
          if (iy >= 0 && iy < p.cv.H && ix >= 0 && ix < p.cv.W) mk |= 1u << (ky * p.cv.KW + kx);
        }
      amask[i] = mk;
    } else {
      if (p.a_rpb > 0) {
        const int b = mc / p.a_rpb;
        arow[i] = ((uint32_t)b * p.a_bstride + (uint32_t)(mc - b * p.a_rpb) * (uint32_t)p.lda) * 4u + 16u * u;
      } else {
        arow[i] = (uint32_t)mc * (uint32_t)p.lda * 4u + 16u * u;
      }
      amask[i] = 0;
    }
  }
  // ---- B pieces: piece q = wave + NW i of the iteration's [k-group][n-tile][hi | lo][k-step] x 1 KiB; n-tiles past the weight
  //      re-read its last tile (columns >= N are never stored) ----
  const __amdgpu_buffer_rsrc_t brsrc = xs_rsrc(p.w.hi, p.w_bytes);
  uint32_t boff[NB];
  int bg[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int q = wave + NW * i;
    bg[i] = KG == 1 ? 0 : q / (BN / 8);
    const int b = q - bg[i] * (BN / 8);
    const int nt = b >> 2, piece = (b >> 1) & 1, ks = b & 1;
    const int ntc = nt0 + nt < p.n_tiles32 ? nt0 + nt : p.n_tiles32 - 1;
    boff[i] = (uint32_t)(((piece * p.n_tiles32 + ntc) * KS + ks) * 64 + lane) * 16u;
  }

  // The two DMA streams walk the iterations in order, DA / DB ahead of the MFMAs; positions are scalar state. Past the last chunk
  // they re-read it (into slots nobody reads again), which keeps the vmcnt arithmetic uniform.
  XsTap at[KG];                    // A stream: position of each k-group's next chunk
#pragma unroll
  for (int g = 0; g < KG; ++g) at[g] = {0, 0, 0, 0, 0};
  auto tap_step = [&](XsTap& t) {
    if (t.c + 1 < nchunk) {
      ++t.c;
      if constexpr (CONV) {
        t.c0 += 32;
        if (t.c0 >= p.cv.C) {
          t.c0 = 0;
          ++t.tap;
          if (++t.kx >= p.cv.KW) {
            t.kx = 0;
            ++t.ky;
          }
        }
      }
    }
  };
  if constexpr (KG == 2) tap_step(at[KG - 1]);
  int a_ld = 0, b_ld = 0, b_c = 0;      // ring slots the next issue fills; B stream's base chunk
  auto issue_a = [&]() {
    unsigned char* sbase = xs_smem + a_ld * A_SLOT;
    uint32_t tapoff[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) tapoff[g] = CONV ? (uint32_t)(((at[g].ky * p.cv.W + at[g].kx) * p.cv.C + at[g].c0) * 4) : 0u;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int g = ag[i];
      const XsTap& t = (KG == 2 && g) ? at[KG - 1] : at[0];
      uint32_t off;
      if constexpr (CONV) off = ((amask[i] >> t.tap) & 1u) ? arow[i] + ((KG == 2 && g) ? tapoff[KG - 1] : tapoff[0]) : XS_OOB;
      else off = arow[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (xs_lds_t)(sbase + (wave + NW * i) * 1024), 16, (int)off,
                                               CONV ? 0 : t.c * 128, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
      for (int k = 0; k < KG; ++k) tap_step(at[g]);
    a_ld = a_ld + 1 == SA ? 0 : a_ld + 1;
  };
  auto issue_b = [&]() {
    unsigned char* sbase = xs_smem + SA * A_SLOT + b_ld * B_SLOT;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int c = b_c + bg[i];
      c = c < nchunk ? c : nchunk - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (xs_lds_t)(sbase + (wave + NW * i) * 1024), 16, (int)boff[i], c * 2048, 0, 0);
    }
    b_c = b_c + KG < nchunk ? b_c + KG : b_c;
    b_ld = b_ld + 1 == SB ? 0 : b_ld + 1;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read addresses: A unit (k-step ks, half hi5, piece pc) = 4 ks + 2 hi5 + pc, XOR the row swizzle; B = lane-linear
  const int sw = (j >> 1) & 7;
  const unsigned char* abase = xs_smem + kg * (BM * 128) + (wm * TM * 32 + j) * 128;
  const unsigned char* bbase = xs_smem + SA * A_SLOT + kg * (BN * 128) + wn * TN * 4096 + lane * 16;
  int aoffs[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) aoffs[ks][pc] = ((4 * ks + 2 * hi5 + pc) ^ sw) * 16;

  // one chunk = 2 k-steps. Software-pipelined in source: the B fragments of k-step 1 and the A fragments of the next m-tile are
  // requested while the current m-tile's MFMAs run (one LDS latency per chunk is exposed -- behind the barrier -- instead of four)
  auto compute = [&](const unsigned char* ab, const unsigned char* bb) {
    u32x4 bh[2][TN], bl[2][TN], ah[2], al[2];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      bh[0][nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096);
      bl[0][nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096 + 2048);
    }
    ah[0] = *reinterpret_cast<const u32x4*>(ab + aoffs[0][0]);
    al[0] = *reinterpret_cast<const u32x4*>(ab + aoffs[0][1]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        const int cur = (ks * TM + mt) & 1, nxt = cur ^ 1;
        if (mt + 1 < TM) {
          ah[nxt] = *reinterpret_cast<const u32x4*>(ab + (mt + 1) * 4096 + aoffs[ks][0]);
          al[nxt] = *reinterpret_cast<const u32x4*>(ab + (mt + 1) * 4096 + aoffs[ks][1]);
        } else if (ks == 0) {
          ah[nxt] = *reinterpret_cast<const u32x4*>(ab + aoffs[1][0]);
          al[nxt] = *reinterpret_cast<const u32x4*>(ab + aoffs[1][1]);
        }
        if (ks == 0 && mt == (TM > 1 ? TM - 2 : 0)) {
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) {
            bh[1][nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096 + 1024);
            bl[1][nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096 + 2048 + 1024);
          }
        }
        __builtin_amdgcn_sched_barrier(0);             // the prefetches issue BEFORE this m-tile's MFMAs (the scheduler sinks them to their uses otherwise)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[cur]), __builtin_bit_cast(f16x8, bh[ks][nt]),
                                                               acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur]), __builtin_bit_cast(f16x8, bl[ks][nt]),
                                                               acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur]), __builtin_bit_cast(f16x8, bh[ks][nt]),
                                                               acc[mt][nt], 0, 0, 0);
      }
    }
  };

  // ---- main loop. Virtual iterations -DA .. -1 fill the rings; iteration it: wait until this wave's pieces of iteration it have
  //      landed (younger ones stay in flight), barrier (everyone's pieces landed AND everyone is done reading the slots of it - 1),
  //      refill those slots (B of it + DB, A of it + DA), compute ----
#pragma unroll
  for (int t = -DA; t < 0; ++t) {
    if (t + DB >= 0) issue_b();
    issue_a();
  }
  // the residual pieces of this wave's first output tile are requested NOW: for the shallow, memory-bound shapes (K = 64 .. 256) the
  // operand fetch and the residual fetch are two HBM latencies that would otherwise be paid one after the other
  const XsEpi epi = xs_epi_setup(p, m0 + wm * TM * 32, lane);
  u32x4 rv0[4];
  if (p.res_fmt && (KG == 1 || kg == 0)) {
    int mres = epi.mr0;
    xs_load_res(rv0, p, epi.rrsrc, mres, (nt0 + wn * TN) * 32, lane);
  }
  int a_rd = 0, b_rd = 0;
  for (int it = 0; it < niter; ++it) {
    xs_wait_vmcnt<WAIT>();
    xs_barrier();
    issue_b();
    issue_a();
    if (KG == 1 || KG * it + kg < nchunk) compute(abase + a_rd * A_SLOT, bbase + b_rd * B_SLOT);
    a_rd = a_rd + 1 == SA ? 0 : a_rd + 1;
    b_rd = b_rd + 1 == SB ? 0 : b_rd + 1;
  }
  xs_wait_vmcnt<0>();              // the run-ahead pieces must land before the rings become reduce buffer / epilogue scratch
  xs_barrier();

  // ---- k-groups: group 1's accumulators -> LDS -> added by group 0 (fixed order: deterministic) ----
  if constexpr (KG == 2) {
    u32x4* red = reinterpret_cast<u32x4*>(xs_smem) + (size_t)wg * TM * TN * 256 + lane;
    if (kg == 1) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            red[((mt * TN + nt) * 4 + q) * 64] = __builtin_bit_cast(u32x4, f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1],
                                                                                 acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]});
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = __builtin_bit_cast(f32x4, red[((mt * TN + nt) * 4 + q) * 64]);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[mt][nt][4 * q + k] += v[k];
        }
  }

  // ---- epilogue (per wave, no further workgroup synchronisation): TM scratch tiles per wave behind the reduce buffer ----
  float* scratch = reinterpret_cast<float*>(xs_smem + (KG - 1) * NWG * TM * TN * 4096 + wg * (TM * 4096));
  const float amax = xs_epilogue<TM, TN>(acc, scratch, p, epi, m0 + wm * TM * 32, nt0 + wn * TN, lane, rv0);
  if (p.out_fmt == 2 && p.flag && !(amax <= CGG_X3A_MAX)) atomicOr(p.flag, 1);
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// (round 5: the tile configuration of a call is an ARGUMENT of the *_cfg entry points -- no process-global override, no
// measurement-only branches in the production loop; the library keeps no mutable state besides the per-device overflow flag word)

// the device-side overflow flag of the x3a producers (one word per process and device, zeroed at creation; read by
// cgg_x3_overflow_check)
static int* g_xs_flag[16] = {nullptr};
static std::once_flag g_xs_once[16];
int* cgg_x3_overflow_flag_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  // created once per device, race-free (two autograd / DDP threads may make the first call together); cgg_init(device) makes the
  // creation explicit -- it must have happened before a graph capture, hipMalloc is not capturable
  std::call_once(g_xs_once[dev], [dev] {
    int* q = nullptr;
    if (hipMalloc(&q, 64) != hipSuccess) return;
    if (hipMemset(q, 0, 64) != hipSuccess) { (void)hipFree(q); return; }
    g_xs_flag[dev] = q;
  });
  return g_xs_flag[dev];
}

extern "C" int cgg_init(int device) {
  int prev = 0;
  hipError_t e = hipGetDevice(&prev);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_init: %s", hipGetErrorString(e));
  CGG_REQUIRE(device >= 0 && device < 16, CGG_EINVAL, "cgg_init: device %d (0..15)", device);
  e = hipSetDevice(device);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
  int* f = cgg_x3_overflow_flag_ptr();
  (void)hipSetDevice(prev);
  CGG_REQUIRE(f, CGG_EINVAL, "cgg_init: could not create the overflow flag word on device %d", device);
  return CGG_OK;
}

extern "C" int cgg_x3_overflow_check(int reset, int* value_host, cgg_stream_t stream) {
  CGG_REQUIRE(value_host, CGG_EINVAL, "cgg_x3_overflow_check: null pointer");
  int* f = cgg_x3_overflow_flag_ptr();
  CGG_REQUIRE(f, CGG_EINVAL, "cgg_x3_overflow_check: no flag word on this device");
  hipError_t e = hipMemcpyAsync(value_host, f, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess && reset) e = hipMemsetAsync(f, 0, sizeof(int), (hipStream_t)stream);
  CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_x3_overflow_check: %s", hipGetErrorString(e));
  return CGG_OK;
}

template <bool CONV, int TM, int TN, int WM, int WN, int KG, int SA, int SB>
static int xs_go(const XsArgs& a, hipStream_t stream, const char* who) {
  constexpr int BN = 32 * TN * WN, BM = 32 * TM * WM;
  constexpr int LDS = KG * (SA * BM + SB * BN) * 128;
  static bool attr[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (LDS > 65536 && dev >= 0 && dev < 16 && !attr[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)cgg_gemm_x3s_kernel<CONV, TM, TN, WM, WN, KG, SA, SB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    CGG_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise dynamic LDS to %d", who, LDS);
    attr[dev] = true;
  }
  XsArgs b = a;
  b.tiles_n = (a.N + BN - 1) / BN;
  const dim3 grid((unsigned)(((a.M + BM - 1) / BM) * b.tiles_n)), block(64 * WM * WN * KG);
  hipLaunchKernelGGL((cgg_gemm_x3s_kernel<CONV, TM, TN, WM, WN, KG, SA, SB>), grid, block, LDS, stream, b);
  return CGG_OK;
}

// tile configurations X(id, TM, TN, WM, WN, KG, SA, SB): workgroup tile (32 TM WM) x (32 TN WN), KG k-groups, SA / SB ring slots
//    0: 256 x 256, 8 waves, A 3 / B 2 slots (160 KiB)     5: 64 x 128, 4 waves, 4 slots (96 KiB)          10: 128 x 128, 2 k-groups (8 waves), 2 slots (128 KiB)
//    1: 256 x 128, 8 waves, 3 slots (144 KiB)             6: 128 x 64, 4 waves, 4 slots (96 KiB)          11: 64 x 128, 2 k-groups, 3 slots (144 KiB)
//    2: 128 x 256, 8 waves, 3 slots (144 KiB)             7: 64 x 64, 4 waves, 4 slots (64 KiB)           12: 64 x 64, 2 k-groups, 4 slots (128 KiB)
//    3: 128 x 128, 4 waves, 4 slots (128 KiB)             8: 256 x 64, 4 waves, 3 slots (120 KiB)         13: 64 x 64, 4 waves, 2 slots (32 KiB)
//    4: 128 x 128, 4 waves, 2 slots (64 KiB)              9: 256 x 256, 8 waves, 2 slots (128 KiB)        14: 128 x 64, 4 waves, 2 slots (48 KiB)
//   16: 64 x 64, 2 k-groups, 2 slots (64 KiB)            17: 64 x 128, 2 k-groups, 2 slots (96 KiB)      15: 64 x 128, 4 waves, 2 slots (48 KiB)
#define XS_CONFIGS(X)                                                                                                              \
  X(0, 4, 2, 2, 4, 1, 3, 2) X(1, 2, 2, 4, 2, 1, 3, 3) X(2, 2, 2, 2, 4, 1, 3, 3) X(3, 2, 2, 2, 2, 1, 4, 4) X(4, 2, 2, 2, 2, 1, 2, 2)   \
  X(5, 1, 2, 2, 2, 1, 4, 4) X(6, 2, 1, 2, 2, 1, 4, 4) X(7, 1, 1, 2, 2, 1, 4, 4) X(8, 4, 1, 2, 2, 1, 3, 3) X(9, 4, 2, 2, 4, 1, 2, 2)   \
  X(10, 2, 2, 2, 2, 2, 2, 2) X(11, 1, 2, 2, 2, 2, 3, 3) X(12, 1, 1, 2, 2, 2, 4, 4) X(13, 1, 1, 2, 2, 1, 2, 2)                      \
  X(14, 2, 1, 2, 2, 1, 2, 2) X(15, 1, 2, 2, 2, 1, 2, 2) X(16, 1, 1, 2, 2, 2, 2, 2) X(17, 1, 2, 2, 2, 2, 2, 2)
#define XS_NCFG 18
static const int xs_bm[XS_NCFG] = {256, 256, 128, 128, 128, 64, 128, 64, 256, 256, 128, 64, 64, 64, 128, 64, 64, 64};
static const int xs_bn[XS_NCFG] = {256, 128, 256, 128, 128, 128, 64, 64, 64, 256, 128, 128, 64, 64, 64, 128, 64, 128};

// Tile configuration by shape, from the per-shape sweep on MI355X (scratch/x3s_bench.py, profiles/r4_x3s_gemm_bench.txt). t = the
// number of 128 x 128 output tiles. Few tiles and a deep K -> k-group configurations (two waves per SIMD on a small tile); many
// tiles and a shallow K -> small tiles with 2 ring slots (memory-bound: several workgroups per CU overlap load / store phases);
// the 256 x 256 tile only where the contraction is MFMA-bound (K >= 1024 with >= 1024 tiles: the FPN's 3 x 3 convolution).
static int xs_pick(int M, int N, int K, bool has_res) {
  const long long t = (long long)((M + 127) / 128) * ((N + 127) / 128);
  // column tile by padding waste (N = 288: 5 x 64 wastes 11 %, 3 x 128 33 %); the wider tile on ties
  const int w64 = (N + 63) / 64 * 64, w128 = (N + 127) / 128 * 128;
  const bool narrow = N <= 64 || w64 < w128;
  const bool deep = K >= 512;
  if (narrow) {
    if (deep && (long long)((M + 63) / 64) * ((N + 63) / 64) < 512) return 12;
    return (M >= 16384 && K < 256) || N > 64 ? 14 : 13;
  }
  if (t >= 1024) return (K >= 1024 && N % 256 == 0) ? 9 : ((K <= 128 && !has_res) ? 14 : 4);
  if (t >= 512) return (deep && N % 256 == 0) ? 2 : (N >= 512 ? 4 : 14);
  if (t >= 256) return deep ? 10 : 13;
  // few tiles, deep K: two k-groups on a small tile, deep rings (inside the step the operands come from HBM / the Infinity Cache,
  // not from L2 as in a repeated stand-alone launch: 2-slot rings lose 10-15 % there)
  if (t >= 128) return deep ? 11 : 13;
  return deep ? 12 : 13;
}

static int xs_launch(bool conv, const char* who, const void* a, int lda, const void* w_x3, const float* bias, const void* res, int ldr,
                     int res_fmt, int res_mod, void* out, int ldc, int out_fmt, int M, int N, int K, int relu, const XsConv& cv,
                     cgg_stream_t stream, int a_rpb = 0, int64_t a_bstride = 0, int force_cfg = -1) {
  CGG_REQUIRE(a && w_x3 && out, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(a_rpb >= 0 && (a_rpb == 0 || (!conv && a_bstride >= (int64_t)(a_rpb - 1) * lda + K && a_bstride % 8 == 0)), CGG_EINVAL,
              "%s: bad batched A (rows per image %d, image stride %lld)", who, a_rpb, (long long)a_bstride);
  CGG_REQUIRE(M > 0 && N > 0 && K > 0, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(K % 32 == 0, CGG_EUNSUPPORTED, "%s: K=%d must be a multiple of 32", who, K);
  CGG_REQUIRE(N % 8 == 0, CGG_EUNSUPPORTED, "%s: N=%d must be a multiple of 8", who, N);
  CGG_REQUIRE(out_fmt == 1 || out_fmt == 2, CGG_EINVAL, "%s: out_fmt must be 1 (f32) or 2 (x3a)", who);
  CGG_REQUIRE(res_fmt >= 0 && res_fmt <= 2 && (res_fmt == 0) == (res == nullptr), CGG_EINVAL, "%s: res / res_fmt mismatch", who);
  CGG_REQUIRE(cgg_aligned16(a) && cgg_aligned16(w_x3) && cgg_aligned16(out) && cgg_aligned16(res) && (conv || lda % 8 == 0) &&
                  ldc % 8 == 0 && (!res || ldr % 8 == 0),
              CGG_EALIGN, "%s: pointers must be 16-byte aligned, row strides multiples of 8 (lda=%d ldc=%d ldr=%d)", who, lda, ldc, ldr);
  CGG_REQUIRE(!res || ldr >= N, CGG_EINVAL, "%s: ldr=%d < N", who, ldr);
  CGG_REQUIRE(ldc >= N, CGG_EINVAL, "%s: ldc=%d < N", who, ldc);
  CGG_REQUIRE(res_mod >= 0 && (!res_mod || res), CGG_EINVAL, "%s: res_mod without res", who);
  CGG_REQUIRE(!conv || (cv.taps >= 1 && cv.taps <= 32), CGG_EUNSUPPORTED, "%s: at most 32 filter taps", who);
  // every operand is addressed through 32-bit buffer descriptors
  const uint64_t a_bytes = conv    ? (uint64_t)(M / (cv.OH * cv.OW)) * cv.H * cv.W * cv.C * 4
                           : a_rpb ? ((uint64_t)((M - 1) / a_rpb) * a_bstride + (uint64_t)((M - 1) % a_rpb) * lda + K) * 4
                                   : ((uint64_t)(M - 1) * lda + K) * 4;
  const uint64_t o_bytes = ((uint64_t)(M - 1) * ldc + N) * 4, r_bytes = res ? ((uint64_t)((res_mod ? res_mod : M) - 1) * ldr + N) * 4 : 0;
  CGG_REQUIRE(a_bytes < 0xFFFFFF00ull && o_bytes < 0xFFFFFF00ull && r_bytes < 0xFFFFFF00ull, CGG_EUNSUPPORTED,
              "%s: an operand spans more than 4 GiB (a %llu, out %llu, res %llu bytes)", who, (unsigned long long)a_bytes,
              (unsigned long long)o_bytes, (unsigned long long)r_bytes);
  XsArgs x;
  x.A = a;
  x.lda = lda;
  x.a_rpb = a_rpb;
  x.a_bstride = (uint32_t)a_bstride;
  x.w = cgg_x3_view(w_x3, N, K);
  x.bias = bias;
  x.res = res;
  x.ldr = ldr;
  x.res_fmt = res_fmt;
  x.res_mod = res_mod;
  x.out = out;
  x.ldc = ldc;
  x.out_fmt = out_fmt;
  x.M = M;
  x.N = N;
  x.K = K;
  x.relu = relu;
  x.tiles_n = 0;
  x.n_tiles32 = (N + 31) / 32;
  x.cv = cv;
  x.a_bytes = (uint32_t)a_bytes;
  x.w_bytes = (uint32_t)(2ull * ((N + 31) / 32) * (K / 16) * 64 * 16);
  x.flag = out_fmt == 2 ? cgg_x3_overflow_flag_ptr() : nullptr;
  CGG_REQUIRE(force_cfg < XS_NCFG, CGG_EINVAL, "%s: tile configuration %d (0 .. %d, or -1 = by shape)", who, force_cfg, XS_NCFG - 1);
  const int cfg = force_cfg >= 0 ? force_cfg : xs_pick(M, N, K, res != nullptr);
  int rc = CGG_OK;
#define XS_CASE(ID, TM, TN, WM, WN, KG, SA, SB)                                                 \
  case ID:                                                                                      \
    rc = conv ? xs_go<true, TM, TN, WM, WN, KG, SA, SB>(x, (hipStream_t)stream, who)            \
              : xs_go<false, TM, TN, WM, WN, KG, SA, SB>(x, (hipStream_t)stream, who);          \
    break;
  switch (cfg) { XS_CONFIGS(XS_CASE) }
#undef XS_CASE
  if (rc != CGG_OK) return rc;
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_gemm_x3s(const void* a_x3a, int lda, const void* w_x3, const float* bias, const void* res, int ldr, int res_fmt,
                            int res_mod, void* out, int ldc, int out_fmt, int M, int N, int K, int relu, cgg_stream_t stream) {
  const XsConv cv = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K, CGG_EINVAL, "cgg_gemm_x3s: lda=%d < K", lda);
  return xs_launch(false, "cgg_gemm_x3s", a_x3a, lda, w_x3, bias, res, ldr, res_fmt, res_mod, out, ldc, out_fmt, M, N, K, relu, cv,
                   stream);
}

// ... with the tile configuration chosen by the caller (0 .. 17; -1 = by shape): tests sweep every instantiation, benches compare
// them -- an argument, not process state
extern "C" int cgg_gemm_x3s_cfg(const void* a_x3a, int lda, const void* w_x3, const float* bias, const void* res, int ldr, int res_fmt,
                                int res_mod, void* out, int ldc, int out_fmt, int M, int N, int K, int relu, int cfg,
                                cgg_stream_t stream) {
  const XsConv cv = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K, CGG_EINVAL, "cgg_gemm_x3s_cfg: lda=%d < K", lda);
  return xs_launch(false, "cgg_gemm_x3s_cfg", a_x3a, lda, w_x3, bias, res, ldr, res_fmt, res_mod, out, ldc, out_fmt, M, N, K, relu, cv,
                   stream, 0, 0, cfg);
}

// A = a stack of images: row m is row m % rows_per_image of image m / rows_per_image (image stride a_bstride elements) -- the
// (B, N, C) encoder memory read one level at a time without a copy, the row-periodic residual (res_mod = rows_per_image) being the
// level's per-token table
extern "C" int cgg_gemm_x3s_batched(const void* a_x3a, int lda, int rows_per_image, int64_t a_bstride, const void* w_x3,
                                    const float* bias, const void* res, int ldr, int res_fmt, int res_mod, void* out, int ldc,
                                    int out_fmt, int M, int N, int K, int relu, cgg_stream_t stream) {
  const XsConv cv = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K && rows_per_image > 0, CGG_EINVAL, "cgg_gemm_x3s_batched: lda=%d < K or rows_per_image=%d", lda, rows_per_image);
  return xs_launch(false, "cgg_gemm_x3s_batched", a_x3a, lda, w_x3, bias, res, ldr, res_fmt, res_mod, out, ldc, out_fmt, M, N, K, relu,
                   cv, stream, rows_per_image, a_bstride);
}

static int xs_conv(const void* x_x3a, const void* w_x3, const float* bias, const void* res, int res_fmt, void* out, int out_fmt, int B,
                   int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu, int cfg, cgg_stream_t stream) {
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, CGG_EINVAL,
              "cgg_conv_x3s_nhwc: bad sizes");
  CGG_REQUIRE(C % 32 == 0, CGG_EUNSUPPORTED, "cgg_conv_x3s_nhwc: C=%d must be a multiple of 32", C);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  CGG_REQUIRE(OH > 0 && OW > 0, CGG_EINVAL, "cgg_conv_x3s_nhwc: empty output");
  if (KH == 1 && KW == 1 && stride == 1 && pad == 0) {       // a 1 x 1 convolution IS the row GEMM (no per-row pixel arithmetic)
    const XsConv cv0 = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    return xs_launch(false, "cgg_conv_x3s_nhwc", x_x3a, C, w_x3, bias, res, N, res_fmt, 0, out, N, out_fmt, B * H * W, N, C, relu, cv0,
                     stream, 0, 0, cfg);
  }
  const XsConv cv = {H, W, C, OH, OW, KW, stride, pad, KH * KW};
  return xs_launch(true, "cgg_conv_x3s_nhwc", x_x3a, 0, w_x3, bias, res, N, res_fmt, 0, out, N, out_fmt, B * OH * OW, N,
                   KH * KW * C, relu, cv, stream, 0, 0, cfg);
}

extern "C" int cgg_conv_x3s_nhwc(const void* x_x3a, const void* w_x3, const float* bias, const void* res, int res_fmt, void* out,
                                 int out_fmt, int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu,
                                 cgg_stream_t stream) {
  return xs_conv(x_x3a, w_x3, bias, res, res_fmt, out, out_fmt, B, H, W, C, N, KH, KW, stride, pad, relu, -1, stream);
}
extern "C" int cgg_conv_x3s_nhwc_cfg(const void* x_x3a, const void* w_x3, const float* bias, const void* res, int res_fmt, void* out,
                                     int out_fmt, int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu,
                                     int cfg, cgg_stream_t stream) {
  return xs_conv(x_x3a, w_x3, bias, res, res_fmt, out, out_fmt, B, H, W, C, N, KH, KW, stride, pad, relu, cfg, stream);
}

// ---- f32 <-> x3a rows (API edges and tests; inside the stream every producer writes x3a itself) ----
__global__ __launch_bounds__(256) void cgg_x3a_convert_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long ngroups,
                                                             int decode, int* __restrict__ flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ngroups) return;
  const u32x4 a = in[2 * i], b = in[2 * i + 1];
  if (decode) {
    float f[8];
    cgg_x3a_decode8(a, b, f);
    out[2 * i] = __builtin_bit_cast(u32x4, f32x4{f[0], f[1], f[2], f[3]});
    out[2 * i + 1] = __builtin_bit_cast(u32x4, f32x4{f[4], f[5], f[6], f[7]});
  } else {
    const f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
    const float f[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
    u32x4 h, l;
    cgg_x3a_encode8(f, h, l);
    out[2 * i] = h;
    out[2 * i + 1] = l;
    if (flag && !(amax * CGG_X3_ASCALE <= CGG_X3A_MAX)) atomicOr(flag, 1);
  }
}

static int xs_convert(const void* in, void* out, int64_t n, int decode, cgg_stream_t stream, const char* who) {
  CGG_REQUIRE(in && out, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(n > 0 && n % 8 == 0, CGG_EINVAL, "%s: element count %lld must be a positive multiple of 8", who, (long long)n);
  CGG_REQUIRE(cgg_aligned16(in) && cgg_aligned16(out), CGG_EALIGN, "%s: 16-B alignment", who);
  const long long ng = n / 8;
  hipLaunchKernelGGL(cgg_x3a_convert_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)in,
                     (u32x4*)out, ng, decode, decode ? nullptr : cgg_x3_overflow_flag_ptr());
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_x3a_encode(const float* x, void* out_x3a, int64_t n, cgg_stream_t stream) {
  return xs_convert(x, out_x3a, n, 0, stream, "cgg_x3a_encode");
}
extern "C" int cgg_x3a_decode(const void* x_x3a, float* out, int64_t n, cgg_stream_t stream) {
  return xs_convert(x_x3a, out, n, 1, stream, "cgg_x3a_decode");
}
