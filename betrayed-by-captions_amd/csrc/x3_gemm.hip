// Parity mode's large contractions: C[M, N] = act(A[M, K] W[N, K]^T * colscale + bias (+ residual)) with f32 operands and f32
// results on the f32-class f16 x 3 arithmetic of x3.h. One kernel serves
//   * the encoder / K-V linears of the pixel decoder and query decoder ([3P] MSDeformAttnPixelDecoder / nn.MultiheadAttention
//     in_proj; built at open_set/models/mask2former_head.py:112-118, called :787, :829-840): A row-major f32, and
//   * every convolution of the f32 path as an IMPLICIT GEMM over a channel-last f32 map (1 x 1 and 3 x 3, stride 1 / 2: the
//     pixel decoder's input / lateral / output / mask-feature convolutions and the BN-folded ResNet): row m = output pixel
//     (b, oy, ox), k = (ky, kx, c); a BK = 32 chunk lies inside one filter tap (C % 32 == 0), so the A loader only changes
//     its base address per chunk and zero-fills taps that fall outside the map -- no im2col matrix is ever written.
// Replaces hipBLASLt f32 GEMMs (~100 TF/s) and MIOpen's f32 Winograd / GEMM convolutions of round 2's parity mode.
//
// Tiling (MI355X): workgroup = 128 x 128 outputs (64 x 128 / 128 x 64 / 64 x 64 when the problem has few tiles), 4 wavefronts
// in a 2 x 2 grid, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32;
// K in chunks of 32 (two k-steps). A chunk is read as f32 (two 16-byte loads per thread and row: one 128-byte line per row),
// split ONCE per workgroup into hi / lo f16 A-fragment images in LDS (slot XOR-swizzle: the 8 lanes of a ds_write_b128 group
// hit 8 different 16-byte bank groups, a fragment read stays a permutation inside its 1-KiB block); the weight is the x3
// image (hi / lo B fragments packed once, cgg_x3_pack) copied linearly into LDS. Two LDS stages: the global loads of chunk
// c + 1 are issued before the MFMAs of chunk c and written to the other stage after them -- one barrier per chunk. Per k-step
// a wave reads 8 fragments (8 KiB) for 12 MFMAs (x3 makes three MFMAs out of every fragment pair: operand traffic per MFMA
// is 2/3 of a bf16 kernel's). Blocks are remapped so that the column tiles of one row tile run on one XCD (shared A rows in
// that XCD's L2).
//
// Where the time goes at the training shapes (M = 344 064 rows, round 5, scratch builds with parts removed; K = 256, N = 256: 212 us
// = 0.25 of the f16 MFMA peak for the 3 x products): without the epilogue 159 us, with A served from cache 179 us, both 144 us,
// without the fragment reads + MFMAs 182 us, with none of the three 71 us (split + LDS writes + barriers alone); N = 1024: 846 /
// 610 / 736 / 578 / 747 / 264 us. The phases ADD instead of overlapping: the store epilogue (0.35 / 1.4 GB) costs its full HBM
// time, the split / LDS-write phase and the MFMA phase of a chunk run back to back in each of the 2 workgroups a CU holds. Cache-
// policy bits on the stores (sc0 / nt / sc1) changed nothing (+-2 %); 128 x 64 / 64 x 128 / 64 x 64 tiles (3-4 workgroups per CU) are 15-40 %
// slower at these shapes, and so are 256 x 128 / 128 x 256 / 256 x 256 tiles (TM / TN = 4: 128 x 128 per wave, 256 accumulators in AGPRs,
// one workgroup of 4 waves per CU, compiler-scheduled): 258 / 269 / 259 us at K = 256, N = 256 and 950 / 902 / 840 us at N = 1024 --
// with one wave per SIMD every LDS / memory wait is exposed; that tile needs hand-placed waits and prefetch, not a template argument.
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define XG_BK 32
#define XG_NT 256

struct XgConv {
  int H, W, C, OH, OW, KW, stride, pad;      // channel-last input [B][H][W][C]; K = KH * KW * C
};

// raw buffer descriptor over `bytes` bytes at p: loads at offsets >= bytes return 0 (the conv loader's zero padding)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xg_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Workgroup tile = (64 TM) x (64 TN): 2 x 2 wavefronts, each TM x TN MFMA tiles of 32 x 32. (2, 2) is the throughput shape;
// the smaller ones keep every CU busy when the problem has few tiles (deep ResNet stages: 2048 rows) or N = 64.
template <bool CONV, int TM, int TN>
__global__ __launch_bounds__(XG_NT, (TM + TN == 4) ? 2 : (TM + TN == 3) ? 3 : 4) void cgg_gemm_x3_kernel(
    const float* __restrict__ A, int lda, const CggX3W w, const float* __restrict__ bias, const float* __restrict__ res, int ldr,
    float* __restrict__ out, int ldc, int M, int N, int K, int relu, int tiles_n, int n_tiles32, XgConv cv, uint32_t a_bytes,
    uint32_t w_bytes, int res_mod, float* __restrict__ out2, int ldc2, int col2, const float* __restrict__ a_amax, int res_mask,
    float* __restrict__ out_amax) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  // A pre-scale: 2^4 (activations) or, when the caller hands the tensor's max |value| (device scalar), the power of two that puts
  // it near 2^9 (grad_output operands: x3.h "per-tensor pre-scale"); the epilogue un-scales by 16 / sa on top of colscale
  const float sa = a_amax ? cgg_x3_scale_from_amax(*a_amax) : CGG_X3_ASCALE;
  const float unsa = CGG_X3_ASCALE / sa;
  constexpr int A_SLOTS = 2 * TM * 2 * 64;                 // one piece of the A stage: [2 TM m-tiles][2 k-steps][64 lanes] x 16 B
  constexpr int B_SLOTS = 2 * TN * 2 * 64;
  constexpr int STAGE = 2 * (A_SLOTS + B_SLOTS);           // A hi | A lo | B hi | B lo
  constexpr int NB = 2 * TN;                               // 16-byte B units per thread and chunk
  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, nt0 = tile_n * (BN / 32);
  const int KS = K >> 4, nchunk = K / XG_BK;

  // ---- A loader: thread -> rows ar + 64 i (i < TM), 8 floats k8 of the chunk. Every load is branch-free: rows past M
  //      read row M - 1 (their results are never stored); the conv form sends taps outside the map to an out-of-range
  //      offset of the buffer descriptor, which the hardware answers with zeros ----
  const int ar = tid >> 2, k8 = tid & 3;
  const __amdgpu_buffer_rsrc_t arsrc = xg_rsrc(A, a_bytes);
  uint32_t aoff[TM];                                       // byte offset of the row (GEMM) / of the image (CONV) + 32 k8
  int aiy[TM], aix[TM];                                    // CONV: top-left input coordinate of the row's receptive field
  int aslot[TM];                                           // LDS slot: m-tile (row >> 5), k-step k8 >> 1, half k8 & 1, swizzled row
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = ar + 64 * i, m = m0 + r;
    const int mc = m < M ? m : M - 1;
    if constexpr (CONV) {
      const int ohw = cv.OH * cv.OW;
      const int b = mc / ohw, rr = mc - b * ohw;
      const int oy = rr / cv.OW, ox = rr - oy * cv.OW;
      aiy[i] = oy * cv.stride - cv.pad;
      aix[i] = ox * cv.stride - cv.pad;
      aoff[i] = (uint32_t)b * (uint32_t)(cv.H * cv.W * cv.C) * 4u + 32u * k8;
    } else {
      aoff[i] = (uint32_t)mc * (uint32_t)lda * 4u + 32u * k8;
      aiy[i] = aix[i] = 0;
    }
    aslot[i] = ((r >> 5) * 2 + (k8 >> 1)) * 64 + (((r & 31) ^ ((k8 << 1) & 7)) + 32 * (k8 & 1));
  }
  // ---- B loader: unit u = q 256 + tid -> (piece, n-tile, k-step, lane) of the x3 image; n-tiles past the weight read its last
  //      tile (columns >= N are never stored) ----
  const __amdgpu_buffer_rsrc_t brsrc = xg_rsrc(w.hi, w_bytes);
  uint32_t boff[NB];                                       // byte offset inside the x3 image at chunk 0; + 2 KiB per chunk
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int u = q * 256 + tid;
    const int piece = u / B_SLOTS, within = u - piece * B_SLOTS;
    const int nt = within >> 7, ks = (within >> 6) & 1;
    const int ntc = nt0 + nt < n_tiles32 ? nt0 + nt : n_tiles32 - 1;
    boff[q] = (uint32_t)(((piece * n_tiles32 + ntc) * KS + ks) * 64 + lane) * 16u;
  }

  // the loader walks the chunks in order (two ahead of the MFMAs); its position is scalar state: channel offset and filter tap
  int ld_c = 0, ld_c0 = 0, ld_ky = 0, ld_kx = 0;
  auto load_chunk = [&](u32x4 (&av)[TM][2], u32x4 (&bv)[NB]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (CONV) {
        const int iy = aiy[i] + ld_ky, ix = aix[i] + ld_kx;
        const bool ok = iy >= 0 && iy < cv.H && ix >= 0 && ix < cv.W;
        const uint32_t off = ok ? aoff[i] + (uint32_t)((iy * cv.W + ix) * cv.C) * 4u : 0xFFFFFF00u;
        av[i][0] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, ld_c0 * 4, 0);
        av[i][1] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, off + 16u, ld_c0 * 4, 0);
      } else {
        av[i][0] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i], ld_c * (XG_BK * 4), 0);
        av[i][1] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i] + 16u, ld_c * (XG_BK * 4), 0);
      }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) bv[q] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, boff[q], ld_c * 2048, 0);
    // advance; the run-ahead past the last chunk re-reads it (nothing consumes those registers)
    if (ld_c + 1 < nchunk) {
      ++ld_c;
      if constexpr (CONV) {
        ld_c0 += XG_BK;
        if (ld_c0 >= cv.C) {
          ld_c0 = 0;
          if (++ld_kx >= cv.KW) {
            ld_kx = 0;
            ++ld_ky;
          }
        }
      }
    }
  };
  auto store_chunk = [&](int stage, const u32x4 (&av)[TM][2], const u32x4 (&bv)[NB]) {
    u32x4* s = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      u32x4 h, l;
      cgg_x3_split8_s(__builtin_bit_cast(f32x4, av[i][0]), __builtin_bit_cast(f32x4, av[i][1]), sa, h, l);
      s[aslot[i]] = h;
      s[A_SLOTS + aslot[i]] = l;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int u = q * 256 + tid;
      const int piece = u / B_SLOTS, within = u - piece * B_SLOTS;
      s[2 * A_SLOTS + piece * B_SLOTS + within] = bv[q];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read offsets of this lane: A slot (row j, half hi5) swizzled by the k-step parity; B slot = lane
  const int aread0 = ((j ^ ((hi5 << 1) & 7)) + 32 * hi5);            // k-step 0 of the chunk: k8 = hi5
  const int aread1 = ((j ^ (((2 + hi5) << 1) & 7)) + 32 * hi5);      // k-step 1: k8 = 2 + hi5

  auto compute = [&](int stage) {
    const u32x4* s = lds + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        const int o = ((TM * wm + mt) * 2 + ks) * 64 + (ks ? aread1 : aread0);
        ah[mt] = s[o];
        al[mt] = s[A_SLOTS + o];
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int o = 2 * A_SLOTS + ((TN * wn + nt) * 2 + ks) * 64 + lane;
        bh[nt] = s[o];
        bl[nt] = s[B_SLOTS + o];
      }
      // the three products of a tile go to its accumulator in three rounds over the tiles: neighbouring MFMAs are independent
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[mt]), __builtin_bit_cast(f16x8, bh[nt]),
                                                               acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bl[nt]),
                                                               acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bh[nt]),
                                                               acc[mt][nt], 0, 0, 0);
    }
  };

  // ---- pipeline: the global loads run TWO chunks ahead of the MFMAs (register sets R0 / R1), the LDS stage one chunk ----
  u32x4 a0[TM][2], b0[NB], a1[TM][2], b1[NB];
  load_chunk(a0, b0);
  load_chunk(a1, b1);
  store_chunk(0, a0, b0);
  __syncthreads();
  for (int c = 0; c + 1 < nchunk; c += 2) {
    // even chunk c: compute from stage 0; chunk c + 1 (in R1) -> stage 1; chunk c + 2 -> R0
    load_chunk(a0, b0);
    __builtin_amdgcn_sched_barrier(0);                     // the loads go out BEFORE the MFMAs (the scheduler sinks them otherwise)
    compute(0);
    store_chunk(1, a1, b1);
    __syncthreads();
    // odd chunk c + 1: compute from stage 1; chunk c + 2 (in R0) -> stage 0; chunk c + 3 -> R1
    load_chunk(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    store_chunk(0, a0, b0);
    __syncthreads();
  }
  if (nchunk & 1) compute(0);                              // odd chunk count: the last chunk sits in stage 0

  // ---- epilogue: acc[mt][nt][r] = C[m0 + 32 (TM wm + mt) + (r & 3) + 8 (r >> 2) + 4 hi5][32 (nt0 + TN wn + nt) + j] ----
  const bool full = m0 + BM <= M && (nt0 + 2 * TN) * 32 <= N;        // workgroup-uniform: interior tiles skip the bounds checks
  float lmax = 0.f;                                                  // max |stored value| of this lane (out_amax)
  // Interior tiles of outputs below 4 GiB (every training-size call): buffer stores whose row term is a SCALAR offset. The generic
  // path below spends ~13 VALU / scalar instructions per stored element (64-bit address arithmetic, per-element flag branches);
  // with K = 256 that was as many VALU instructions as the whole main loop (counters: 6.7 VALU per MFMA at K = 256, N = 1024).
  // Here: one fma, one max (ReLU as a floor of 0 / -inf), one max for out_amax, one buffer store.
  const size_t o_span = (size_t)M * (size_t)ldc * 4u, o2_span = out2 ? (size_t)M * (size_t)ldc2 * 4u : 0;
  const size_t r_span = res ? (size_t)M * (size_t)ldr * 4u : 0;
  if (full && res_mod == 0 && o_span < 0xFFFFFF00ull && o2_span < 0xFFFFFF00ull && r_span < 0xFFFFFF00ull) {
    const float floor_v = relu ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t rres = xg_rsrc(res ? res : out, 0xFFFFFF00u);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int ncol0 = __builtin_amdgcn_readfirstlane((nt0 + TN * wn + nt) * 32);      // first column of the tile (wave-uniform)
      const int n = ncol0 + j;
      const float cs = w.scale[n] * unsa;
      const float bs = bias ? bias[n] : 0.f;
      const bool second = out2 != nullptr && ncol0 >= col2;
      const int ldo = second ? ldc2 : ldc;
      const __amdgpu_buffer_rsrc_t rout = xg_rsrc(second ? out2 - col2 : out, 0xFFFFFF00u);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        const int mrow0 = __builtin_amdgcn_readfirstlane(m0 + 32 * (TM * wm + mt));      // wave-uniform
        const uint32_t voff = (uint32_t)((4 * hi5) * ldo + n) * 4u;
        const uint32_t rvoff = (uint32_t)((4 * hi5) * ldr + n) * 4u;
        float rv[16];
        if (res) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rres, rvoff, (uint32_t)(mrow0 + (r & 3) + 8 * (r >> 2)) * (uint32_t)ldr * 4u, 0));
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaf(acc[mt][nt][r], cs, bs);
        if (res && res_mask) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = rv[r] > 0.f ? v[r] : 0.f;      // ReLU backward of the layer behind (res = its output)
        } else if (res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += rv[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = __builtin_amdgcn_fmed3f(v[r], floor_v, __builtin_inff());      // one instruction (fmaxf adds a canonicalising max)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[r]), rout, voff,
                                                (uint32_t)(mrow0 + (r & 3) + 8 * (r >> 2)) * (uint32_t)ldo * 4u, 0);
          lmax = fmaxf(lmax, fabsf(v[r]));
        }
      }
    }
  } else {
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int n = (nt0 + TN * wn + nt) * 32 + j;
    const bool nok = full || n < N;
    const float cs = nok ? w.scale[n] * unsa : 0.f;
    const float bs = (nok && bias) ? bias[n] : 0.f;
    // columns from col2 on (a multiple of 32: uniform per n-tile) go to the second output
    const bool second = out2 != nullptr && n >= col2;
    float* const obase = second ? out2 + (n - col2) : out + n;
    const int ldo = second ? ldc2 : ldc;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int mrow = m0 + 32 * (TM * wm + mt) + 4 * hi5;
      float* orow = obase + (size_t)mrow * ldo;
      if (full && (res_mod == 0 || res_mod >= 32)) {
        // interior tile (workgroup-uniform): no per-element predicate, so the 16 residual loads of the tile are all in flight
        // before the first use -- under the predicate below the compiler emits load / s_waitcnt vmcnt(0) / store per element,
        // 16 serial memory latencies per tile (the ResNet output convolutions ran at 2.3 TB/s because of it)
        float rv[16];
        if (res) {
          // res_mod: the residual rows repeat (per-token table); one modulo per tile, the 28 rows after it wrap by subtraction
          const int mr0 = res_mod > 0 ? mrow % res_mod : mrow;
          const int wrap = res_mod > 0 ? res_mod : 0x7fffffff;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int mr = mr0 + (r & 3) + 8 * (r >> 2);
            mr = mr >= wrap ? mr - wrap : mr;
            rv[r] = res[(size_t)mr * ldr + n];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[mt][nt][r] * cs + bs;
          if (res) v = res_mask ? (rv[r] > 0.f ? v : 0.f) : v + rv[r];      // res_mask: the ReLU backward of the layer behind (res = its output)
          if (relu) v = fmaxf(v, 0.f);
          orow[(size_t)((r & 3) + 8 * (r >> 2)) * ldo] = v;
          lmax = fmaxf(lmax, fabsf(v));
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (full || (nok && mrow + dr < M)) {
          float v = acc[mt][nt][r] * cs + bs;
          if (res) {
            const int mr = res_mod > 0 ? (mrow + dr) % res_mod : mrow + dr;      // res_mod: residual rows repeat (per-token table)
            const float rvv = res[(size_t)mr * ldr + n];
            v = res_mask ? (rvv > 0.f ? v : 0.f) : v + rvv;
          }
          if (relu) v = fmaxf(v, 0.f);
          orow[(size_t)dr * ldo] = v;
          lmax = fmaxf(lmax, fabsf(v));
        }
      }
    }
  }
  }
  // max |out| for the NEXT contraction's per-tensor pre-scale (x3.h): one atomic per wave; |v| as a bit pattern is monotone
  if (out_amax) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, sft, 64));
    // (only when it can raise the running maximum: 21 504 wavefront atomics on one address serialise)
    if (lane == 0 && __float_as_uint(lmax) > __hip_atomic_load(reinterpret_cast<unsigned int*>(out_amax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned int*>(out_amax), __float_as_uint(lmax));
  }
}

static int xg_launch(bool conv, const char* who, const float* a, int lda, const void* w_x3, const float* bias, const float* res,
                     int ldr, float* out, int ldc, int M, int N, int K, int relu, const XgConv& cv, cgg_stream_t stream,
                     int res_mod = 0, float* out2 = nullptr, int ldc2 = 0, int col2 = 0, const float* a_amax = nullptr, int res_mask = 0,
                     float* out_amax = nullptr) {
  CGG_REQUIRE(a && w_x3 && out, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(M > 0 && N > 0 && K > 0, CGG_EINVAL, "%s: bad sizes", who);
  CGG_REQUIRE(K % XG_BK == 0, CGG_EUNSUPPORTED, "%s: K=%d must be a multiple of %d", who, K, XG_BK);
  CGG_REQUIRE(cgg_aligned16(a) && cgg_aligned16(w_x3) && (conv || lda % 4 == 0), CGG_EALIGN, "%s: alignment (lda=%d)", who, lda);
  CGG_REQUIRE(!res || ldr >= N, CGG_EINVAL, "%s: ldr=%d < N", who, ldr);
  CGG_REQUIRE(ldc >= (out2 ? col2 : N), CGG_EINVAL, "%s: ldc=%d too small", who, ldc);
  CGG_REQUIRE(!out2 || (col2 > 0 && col2 % 32 == 0 && col2 < N && ldc2 >= N - col2), CGG_EINVAL,
              "%s: second output needs 0 < col2 < N, col2 %% 32 == 0, ldc2 >= N - col2 (col2=%d ldc2=%d)", who, col2, ldc2);
  CGG_REQUIRE(res_mod >= 0 && (!res_mod || res), CGG_EINVAL, "%s: res_mod without res", who);
  // tile shape: the largest whose grid still has >= 192 workgroups. Measured in the 3-stage pipelined step (A/B pairs on one box,
  // round-4 A/B script, `git log -- scratch/`): the larger tiles are the more efficient ones per FLOP, and the CUs a small grid leaves idle are taken by
  // the other streams' kernels, so the step prefers FEWER, BIGGER tiles than a stand-alone launch does -- threshold 384 + smaller
  // tiles for K <= 512 (the stand-alone optimum of the first version): 349 images/s; 192, no K rule: 358 (eager GEMM time equal);
  // 128: 363 with +7 % eager GEMM time and +3 % latency (not taken).
  const int mintiles = 192;
  auto tiles = [&](int tm, int tn) { return (long long)((M + 64 * tm - 1) / (64 * tm)) * ((N + 64 * tn - 1) / (64 * tn)); };
  int tm = 2, tn = N <= 64 ? 1 : 2;
  if (tiles(tm, tn) < mintiles) tm = 1;
  if (tiles(tm, tn) < mintiles && tn == 2) tn = 1;
  const int tiles_n = (N + 64 * tn - 1) / (64 * tn);
  const CggX3W w = cgg_x3_view(w_x3, N, K);
  const dim3 grid((unsigned)tiles(tm, tn)), block(XG_NT);
  // the A operand is addressed through a 32-bit buffer descriptor
  const uint64_t a_bytes = conv ? (uint64_t)(M / (cv.OH * cv.OW)) * cv.H * cv.W * cv.C * 4 : ((uint64_t)(M - 1) * lda + K) * 4;
  CGG_REQUIRE(a_bytes < 0xFFFFFF00ull, CGG_EUNSUPPORTED, "%s: the A operand spans %llu bytes (limit 4 GiB)", who,
              (unsigned long long)a_bytes);
  const uint32_t w_bytes = (uint32_t)(2ull * ((N + 31) / 32) * (K / 16) * 64 * 16);
#define XG_GO(CONV, TM, TN)                                                                                                   \
  hipLaunchKernelGGL((cgg_gemm_x3_kernel<CONV, TM, TN>), grid, block, 0, (hipStream_t)stream, a, lda, w, bias, res, ldr, out, ldc, \
                     M, N, K, relu, tiles_n, (N + 31) / 32, cv, (uint32_t)a_bytes, w_bytes, res_mod, out2, ldc2, col2, a_amax, res_mask, out_amax)
#define XG_PICK(CONV)                    \
  do {                                   \
    if (tm == 2 && tn == 2) XG_GO(CONV, 2, 2); \
    else if (tm == 2) XG_GO(CONV, 2, 1); \
    else if (tn == 2) XG_GO(CONV, 1, 2); \
    else XG_GO(CONV, 1, 1);              \
  } while (0)
  if (conv) XG_PICK(true);
  else XG_PICK(false);
#undef XG_PICK
#undef XG_GO
  CGG_CHECK_LAUNCH(who);
  return CGG_OK;
}

extern "C" int cgg_gemm_x3(const float* a, int lda, const void* w_x3, const float* bias, const float* res, int ldr, float* out,
                           int ldc, int M, int N, int K, int relu, cgg_stream_t stream) {
  const XgConv cv = {0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K, CGG_EINVAL, "cgg_gemm_x3: lda=%d < K", lda);
  return xg_launch(false, "cgg_gemm_x3", a, lda, w_x3, bias, res, ldr, out, ldc, M, N, K, relu, cv, stream);
}

extern "C" int cgg_gemm_x3_ex(const float* a, int lda, const void* w_x3, const float* bias, const float* res, int ldr, int res_mod,
                              float* out, int ldc, float* out2, int ldc2, int col2, int M, int N, int K, int relu,
                              cgg_stream_t stream) {
  const XgConv cv = {0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K, CGG_EINVAL, "cgg_gemm_x3_ex: lda=%d < K", lda);
  return xg_launch(false, "cgg_gemm_x3_ex", a, lda, w_x3, bias, res, ldr, out, ldc, M, N, K, relu, cv, stream, res_mod, out2, ldc2,
                   col2);
}

extern "C" int cgg_conv_x3_nhwc(const float* x, const void* w_x3, const float* bias, const float* res, float* out, int B, int H,
                                int W, int C, int N, int KH, int KW, int stride, int pad, int relu, cgg_stream_t stream) {
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, CGG_EINVAL,
              "cgg_conv_x3_nhwc: bad sizes");
  CGG_REQUIRE(C % XG_BK == 0, CGG_EUNSUPPORTED, "cgg_conv_x3_nhwc: C=%d must be a multiple of %d", C, XG_BK);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  CGG_REQUIRE(OH > 0 && OW > 0, CGG_EINVAL, "cgg_conv_x3_nhwc: empty output");
  const XgConv cv = {H, W, C, OH, OW, KW, stride, pad};
  return xg_launch(true, "cgg_conv_x3_nhwc", x, 0, w_x3, bias, res, N, out, N, B * OH * OW, N, KH * KW * C, relu, cv, stream);
}

// ... with the A operand pre-scaled per TENSOR: a_amax = device scalar holding max |a| (cgg_absmax_f32), nullable = the fixed 2^4.
// For operands that are not unit scale -- grad_output of the training linears / convolutions (x3.h "per-tensor pre-scale").
extern "C" int cgg_gemm_x3_scaled(const float* a, int lda, const float* a_amax, const void* w_x3, const float* bias, const float* res,
                                  int ldr, float* out, int ldc, int M, int N, int K, int relu, cgg_stream_t stream) {
  const XgConv cv = {0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K, CGG_EINVAL, "cgg_gemm_x3_scaled: lda=%d < K", lda);
  return xg_launch(false, "cgg_gemm_x3_scaled", a, lda, w_x3, bias, res, ldr, out, ldc, M, N, K, relu, cv, stream, 0, nullptr, 0, 0,
                   a_amax);
}

extern "C" int cgg_conv_x3_nhwc_scaled(const float* x, const float* x_amax, const void* w_x3, const float* bias, const float* res,
                                       float* out, int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu,
                                       cgg_stream_t stream) {
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, CGG_EINVAL,
              "cgg_conv_x3_nhwc_scaled: bad sizes");
  CGG_REQUIRE(C % XG_BK == 0, CGG_EUNSUPPORTED, "cgg_conv_x3_nhwc_scaled: C=%d must be a multiple of %d", C, XG_BK);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  CGG_REQUIRE(OH > 0 && OW > 0, CGG_EINVAL, "cgg_conv_x3_nhwc_scaled: empty output");
  const XgConv cv = {H, W, C, OH, OW, KW, stride, pad};
  return xg_launch(true, "cgg_conv_x3_nhwc_scaled", x, 0, w_x3, bias, res, N, out, N, B * OH * OW, N, KH * KW * C, relu, cv, stream, 0,
                   nullptr, 0, 0, x_amax);
}

// Backward-side GEMM of a fused training layer: cgg_gemm_x3_scaled with (a) `mask` (M, N, row stride ldm; nullable): the result is
// zeroed where mask <= 0 -- the ReLU backward of the layer whose OUTPUT `mask` is (autograd's threshold_backward behind the FFN's
// first linear, [3P] FFN in the MSDeformAttn encoder layers, mask2former_head.py:787) -- and (b) out_amax (device scalar,
// nullable): max |out| written by the epilogue (zeroed here first) so that the next contraction's per-tensor pre-scale costs no
// extra pass over the tensor.
extern "C" int cgg_gemm_x3_bwd(const float* a, int lda, const float* a_amax, const void* w_x3, const float* mask, int ldm, float* out,
                               int ldc, float* out_amax, int M, int N, int K, cgg_stream_t stream) {
  const XgConv cv = {0, 0, 0, 0, 0, 0, 0, 0};
  CGG_REQUIRE(lda >= K, CGG_EINVAL, "cgg_gemm_x3_bwd: lda=%d < K", lda);
  if (out_amax) {
    hipError_t e = hipMemsetAsync(out_amax, 0, sizeof(float), (hipStream_t)stream);
    CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_gemm_x3_bwd: memset failed");
  }
  return xg_launch(false, "cgg_gemm_x3_bwd", a, lda, w_x3, nullptr, mask, ldm, out, ldc, M, N, K, 0, cv, stream, 0, nullptr, 0, 0, a_amax,
                   mask ? 1 : 0, out_amax);
}
