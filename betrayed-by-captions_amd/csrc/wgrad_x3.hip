// Round 4, training in parity mode: the weight gradient of a linear layer, dW[n][k] = sum_m dY[m][n] X[m][k], on the f32-class
// f16 x 3 contraction (x3.h) -- what autograd computes as `grad_output.t() @ input` for the F.linear calls under
// open_set/models/mask2former_head.py:787 ([3P] MSDeformAttn encoder layers: value_proj / sampling_offsets / attention_weights /
// output_proj / FFN, 344 064 rows at configs[2]). hipBLASLt's f32 GEMM runs this at the f32 MFMA peak (144 TF measured).
//
// Both operands are row-major with the CONTRACTION index m as the slow one, the layout an MFMA operand cannot take directly (a lane
// needs 8 consecutive m of one column). The 32-row x 128-column tiles of dY and X are split into f16 pairs on their way into LDS,
// stored row-major, and read back as MFMA fragments by `ds_read_b64_tr_b16` transpose reads (within a 16-lane group, lane c
// receives element c % 4 of the 8-byte chunks addressed by lanes 4 j + c / 4: probed on the device, scratch/tr/tr_probe.hip):
// a 16-lane group fetches 4 consecutive rows m x 16 columns and each lane ends up with its column's 4 m-values.
//
// Workgroup = 4 waves (2 x 2), tile 128 (n) x 128 (k), wave 64 x 64 = 2 x 2 MFMA tiles; grid.y = SPLIT contiguous row ranges whose
// partial tiles go to ws[split][N][K] (plain stores: deterministic; the caller sums over the splits). Global loads of chunk c + 1
// are in flight (registers) while chunk c is multiplied; one LDS buffer of 40 KiB (row stride 320 B: four consecutive rows fall
// into disjoint bank groups for the transpose reads).
#include "x3.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) uint32_t wg_u32x4;
typedef __attribute__((ext_vector_type(4))) short wg_s16x4;
typedef __attribute__((address_space(3))) wg_s16x4 wg_lds_s16x4;

template <int RS>
__device__ __forceinline__ wg_u32x4 wg_tr_frag(const unsigned char* plane, int row0, int colbyte) {
  // rows row0 .. row0 + 3 (first half) and row0 + 4 .. row0 + 7 (second half) of this lane's column group
  const wg_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(plane + row0 * RS + colbyte));
  const wg_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(plane + (row0 + 4) * RS + colbyte));
  const uint2 au = __builtin_bit_cast(uint2, a), bu = __builtin_bit_cast(uint2, b);
  return wg_u32x4{au.x, au.y, bu.x, bu.y};
}

// Tile BT x BT (128: 2 x 2 waves of 2 x 2 MFMA tiles; 256: 2 x 4 waves of 4 x 2 MFMA tiles -- each operand is then read once per
// 256 output rows / columns instead of once per 128: the 128-wide tile moved 1.4 GB for a 256 x 256 weight, 5.6 TB/s)
template <int BT>
__global__ __launch_bounds__(BT == 128 ? 256 : 512) void cgg_wgrad_x3_kernel(const float* __restrict__ dy, int ldy,
                                                                             const float* __restrict__ x, int ldx,
                                                                             float* __restrict__ ws, float* __restrict__ ws_bias,
                                                                             int M, int N, int K, int tiles_k, int rows_per_split,
                                                                             const float* __restrict__ dy_amax) {
  // dy's pre-scale: per tensor from its max |value| when the caller provides it (x3.h "per-tensor pre-scale"), else 2^4
  const float sy = dy_amax ? cgg_x3_scale_from_amax(*dy_amax) : CGG_X3_ASCALE;
  const float unscale = 1.f / (sy * CGG_X3_ASCALE);
  constexpr int NT = BT == 128 ? 256 : 512;           // threads
  constexpr int WK = BT == 128 ? 2 : 4;               // waves along k (2 along n)
  constexpr int TA = BT == 128 ? 2 : 4, TB = 2;       // MFMA tiles per wave along n / k
  constexpr int RS = 2 * BT + 64;                     // LDS row stride in bytes (BT f16 + 64 B pad: four consecutive rows fall into
                                                      // disjoint bank groups for the transpose reads)
  constexpr int WG_PLANE = 32 * RS;
  constexpr int CG = BT / 4;                          // 16-byte column groups per row
  constexpr int RPI = NT / CG;                        // rows staged per pass (8)
  // LDS stages of Yh | Yl | Xh | Xl. The 256-wide tile is ONE 8-wave workgroup per CU: with a single stage its waves alternate in lock
  // step between the split / LDS-write phase and the transpose-read / MFMA phase (two barriers per chunk, the matrix pipe idle in
  // between); with two stages (144 of 160 KiB) chunk c + 1 is staged while chunk c is multiplied, one barrier per chunk. The 128-wide
  // tile keeps one stage: four workgroups share a CU and fill each other's phases.
  constexpr int NBUF = BT == 256 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Yh = lds;
  unsigned char* Yl = lds + WG_PLANE;
  unsigned char* Xh = lds + 2 * WG_PLANE;
  unsigned char* Xl = lds + 3 * WG_PLANE;
  constexpr int STAGE_B = 4 * WG_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave - wn * WK;
  // the tiles of one row range run on ONE XCD (they re-read the range's dy / x columns: 4 x for a 1024 x 256 weight -- from that XCD's
  // L2 instead of HBM): the hardware deals workgroups x-fastest round-robin over the 8 XCDs, the remap hands every XCD a contiguous
  // run of (split, tile) pairs, all of them resident at once (one grid round)
  const int lin = cgg_xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
  const int bx = lin % (int)gridDim.x, by = lin / (int)gridDim.x;
  const int tile_n = bx / tiles_k, tile_k = bx - tile_n * tiles_k;
  const int n0 = tile_n * BT, k0 = tile_k * BT;
  const int m_begin = by * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);

  // staging: thread = (row r8 + 8 i, 16-byte column group c4) of both tiles
  const int r8 = tid / CG, c4 = tid - r8 * CG;
  static_assert(RPI == 8, "eight rows per staging pass");
  const int ny = n0 + 4 * c4, kx = k0 + 4 * c4;
  const bool yok = ny < N, xok = kx < K;                  // N, K % 4 == 0: a group is inside or outside
  const float* yp = dy + (yok ? ny : 0);
  const float* xp = x + (xok ? kx : 0);
  f32x4 yv[4], xv[4];
  auto load = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + r8 + 8 * i;
      const int mc = m < m_end ? m : m_end - 1;
      yv[i] = *reinterpret_cast<const f32x4*>(yp + (size_t)mc * ldy);
      xv[i] = *reinterpret_cast<const f32x4*>(xp + (size_t)mc * ldx);
    }
  };
  // bias gradient = the column sums of dy: the workgroups of the first k-tile column add up what they stage anyway (f32)
  const bool want_bias = ws_bias != nullptr && tile_k == 0;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  // interior tiles (every column group inside N and K) and chunks with all 32 rows live skip the zero-fill selects: 48 of the
  // ~200 VALU instructions a chunk cost next to its 48 MFMAs (workgroup-uniform conditions)
  const bool colfull = n0 + BT <= N && k0 + BT <= K;
  auto stage_t = [&](int m0, int sb, auto fullc_t) {
    constexpr bool FULLC = decltype(fullc_t)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 y = yv[i], xx = xv[i];
      if constexpr (!FULLC) {
        const bool live = m0 + r8 + 8 * i < m_end;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        y = (live && yok) ? y : z;
        xx = (live && xok) ? xx : z;
      }
      uint2 h, l;
      if (want_bias) bsum += y;
      cgg_x3_split4_s(y, sy, h, l);
      const int o = sb + (r8 + 8 * i) * RS + 8 * c4;
      *reinterpret_cast<uint2*>(Yh + o) = h;
      *reinterpret_cast<uint2*>(Yl + o) = l;
      cgg_x3_split4(xx, h, l);
      *reinterpret_cast<uint2*>(Xh + o) = h;
      *reinterpret_cast<uint2*>(Xl + o) = l;
    }
  };
  auto stage = [&](int m0, int sb) {
    if (colfull && m0 + 32 <= m_end) stage_t(m0, sb, std::true_type{});
    else stage_t(m0, sb, std::false_type{});
  };

  f32x16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // transpose-read geometry: 16-lane group g: column block 16 (g & 1), row half u = g >> 1; lane's chunk = row (i16 >> 2), columns
  // 4 (i16 & 3) .. + 3 of the block
  const int g = lane >> 4, u = g >> 1, i16 = lane & 15;
  const int colb = 2 * (16 * (g & 1) + 4 * (i16 & 3));      // byte offset inside a 32-column MFMA tile
  const int rowl = 8 * u + (i16 >> 2);

  auto compute = [&](int sb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      wg_u32x4 ah[TA], al[TA], bh[TB], bl[TB];
#pragma unroll
      for (int t = 0; t < TA; ++t) {
        const int cy = 2 * (wn * 32 * TA + t * 32) + colb;
        ah[t] = wg_tr_frag<RS>(Yh + sb, 16 * s + rowl, cy);
        al[t] = wg_tr_frag<RS>(Yl + sb, 16 * s + rowl, cy);
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int cx = 2 * (wk * 32 * TB + t * 32) + colb;
        bh[t] = wg_tr_frag<RS>(Xh + sb, 16 * s + rowl, cx);
        bl[t] = wg_tr_frag<RS>(Xl + sb, 16 * s + rowl, cx);
      }
      // the three products of a tile in three rounds over the tiles (small terms first): neighbouring MFMAs are independent
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[a]), __builtin_bit_cast(f16x8, bh[b]), acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[a]), __builtin_bit_cast(f16x8, bl[b]), acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[a]), __builtin_bit_cast(f16x8, bh[b]), acc[a][b], 0, 0, 0);
    }
  };
  if (m_begin < m_end) load(m_begin);
  if constexpr (NBUF == 2) {
    if (m_begin < m_end) {
      stage(m_begin, 0);
      if (m_begin + 32 < m_end) load(m_begin + 32);
    }
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 32, cur ^= STAGE_B) {
      // chunk m0 is staged by every wave, and every wave is done with the other stage (its MFMAs of the previous iteration)
      __syncthreads();
      if (m0 + 32 < m_end) {               // workgroup-uniform
        stage(m0 + 32, cur ^ STAGE_B);
        if (m0 + 64 < m_end) load(m0 + 64);
      }
      compute(cur);
    }
  } else {
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
      __syncthreads();                       // the previous chunk's fragments are read
      stage(m0, 0);
      __syncthreads();
      if (m0 + 32 < m_end) load(m0 + 32);    // workgroup-uniform
      compute(0);
    }
  }

  if (want_bias) {
    // the 8 row groups of a column group (threads c4, c4 + 32, ...) -> LDS (the tile planes are dead) -> one sum per column
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    red[tid] = bsum;
    __syncthreads();
    if (tid < CG) {
      f32x4 t = red[tid];
#pragma unroll
      for (int r = 1; r < 8; ++r) t += red[tid + CG * r];
      const int n = n0 + 4 * tid;
      if (n < N) *reinterpret_cast<f32x4*>(ws_bias + (size_t)by * N + n) = t;
    }
  }
  // partial tile -> ws[split][n][k]; lane (k column j, half hi5), register r <-> n row 8 (r >> 2) + 4 hi5 + (r & 3)
  const int j = lane & 31, hi5 = lane >> 5;
  float* wsp = ws + (size_t)by * N * K;
  if (colfull && (size_t)N * K * 4u < 0xFFFFFF00ull) {
    // interior tile: buffer stores, the row term of the address in a scalar register (one multiply + one store per element; the
    // generic form below computes a 64-bit address and two bounds tests per element)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(wsp, 0, (uint32_t)((size_t)N * K * 4u), 0x00020000);
    const int wn_u = __builtin_amdgcn_readfirstlane(wn), wk_u = __builtin_amdgcn_readfirstlane(wk);
    const uint32_t voff = (uint32_t)(4 * hi5 * K + j) * 4u;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int kc0 = k0 + wk_u * 32 * TB + b * 32, nr0 = n0 + wn_u * 32 * TA + a * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc[a][b][r] * unscale), rw, voff,
                                                (uint32_t)((nr0 + 8 * (r >> 2) + (r & 3)) * K + kc0) * 4u, 0);
      }
    return;
  }
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int kc = k0 + wk * 32 * TB + b * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 * TA + a * 32 + 8 * (r >> 2) + 4 * hi5 + (r & 3);
        if (n < N && kc < K) wsp[(size_t)n * K + kc] = acc[a][b][r] * unscale;
      }
    }
}

// tile edge: 256 when both extents are multiples of it (every operand byte is then read by half as many workgroups)
static int wgrad_tile(int N, int K) { return (N % 256 == 0 && K % 256 == 0) ? 256 : 128; }

// rows per split and number of splits for (M, N, K): ~768 workgroups of 4 waves / ~256 of 8, row ranges multiples of 32
static void wgrad_plan(int M, int N, int K, int* splits, int* rows_per_split) {
  const int bt = wgrad_tile(N, K);
  const int tiles = ((N + bt - 1) / bt) * ((K + bt - 1) / bt);
  // workgroups aimed at: 256 for the 256-wide tile (one 8-wave workgroup per CU: ONE round; 512 -- two rounds and twice the partial
  // planes for the caller's sum -- measured 236 vs 203 us at 344 064 rows, 128: 291), 768 for the 128-wide tile (1 024: 324 vs 306 us)
  int sp = ((bt == 128 ? 768 : 256) + tiles - 1) / tiles;
  const int max_sp = (M + 255) / 256;                  // at least 256 rows per split
  if (sp > max_sp) sp = max_sp;
  if (sp < 1) sp = 1;
  int rps = ((M + sp - 1) / sp + 31) / 32 * 32;
  *rows_per_split = rps;
  *splits = (M + rps - 1) / rps;
}

extern "C" int64_t cgg_wgrad_x3_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int sp, rps;
  wgrad_plan(M, N, K, &sp, &rps);
  return (int64_t)sp * N * K * (int64_t)sizeof(float);
}

// ws (cgg_wgrad_x3_workspace_bytes) receives `*splits_out` partial (N, K) f32 matrices; dW = their sum (fixed order: the caller's
// reduction). dy (M, N) rows at stride ldy, x (M, K) rows at stride ldx, f32, |values| < 4094; N, K, ldy, ldx multiples of 4.
static int wgrad_launch(const float* dy, int ldy, const float* x, int ldx, float* ws, float* ws_bias, int* splits_out, int M, int N,
                        int K, cgg_stream_t stream, const float* dy_amax = nullptr) {
  CGG_REQUIRE(dy && x && ws && splits_out, CGG_EINVAL, "cgg_wgrad_x3: null pointer");
  CGG_REQUIRE(M > 0 && N > 0 && K > 0 && ldy >= N && ldx >= K, CGG_EINVAL, "cgg_wgrad_x3: bad sizes");
  CGG_REQUIRE(N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0, CGG_EUNSUPPORTED,
              "cgg_wgrad_x3: N=%d, K=%d and the row strides must be multiples of 4", N, K);
  CGG_REQUIRE(cgg_aligned16(dy) && cgg_aligned16(x) && cgg_aligned16(ws), CGG_EALIGN, "cgg_wgrad_x3: 16-B alignment");
  int sp, rps;
  wgrad_plan(M, N, K, &sp, &rps);
  *splits_out = sp;
  const int bt = wgrad_tile(N, K);
  const int tiles_k = (K + bt - 1) / bt, tiles_n = (N + bt - 1) / bt;
  if (bt == 128) {
    hipLaunchKernelGGL(cgg_wgrad_x3_kernel<128>, dim3(tiles_n * tiles_k, sp), dim3(256), 4 * 32 * (2 * 128 + 64), (hipStream_t)stream, dy,
                       ldy, x, ldx, ws, ws_bias, M, N, K, tiles_k, rps, dy_amax);
  } else {
    const int lds = 2 * 4 * 32 * (2 * 256 + 64);       // two stages of 72 KiB: above the default dynamic-LDS limit
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)cgg_wgrad_x3_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_wgrad_x3: cannot raise dynamic LDS to %d", lds);
      if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(cgg_wgrad_x3_kernel<256>, dim3(tiles_n * tiles_k, sp), dim3(512), lds, (hipStream_t)stream, dy, ldy, x, ldx, ws,
                       ws_bias, M, N, K, tiles_k, rps, dy_amax);
  }
  CGG_CHECK_LAUNCH("cgg_wgrad_x3");
  return CGG_OK;
}

extern "C" int cgg_wgrad_x3(const float* dy, int ldy, const float* x, int ldx, float* ws, int* splits_out, int M, int N, int K,
                            cgg_stream_t stream) {
  return wgrad_launch(dy, ldy, x, ldx, ws, nullptr, splits_out, M, N, K, stream);
}

// ... and the bias gradient db[n] = sum_m dy[m][n] from the same pass: ws_bias receives *splits_out partial (N) f32 rows
// (ws_bias >= splits x N floats, 16-B aligned; the splits are those of cgg_wgrad_x3_workspace_bytes / (N K 4)).
extern "C" int cgg_wgrad_bias_x3(const float* dy, int ldy, const float* x, int ldx, float* ws, float* ws_bias, int* splits_out, int M,
                                 int N, int K, cgg_stream_t stream) {
  CGG_REQUIRE(ws_bias && cgg_aligned16(ws_bias), CGG_EINVAL, "cgg_wgrad_bias_x3: ws_bias must be a 16-B aligned buffer");
  return wgrad_launch(dy, ldy, x, ldx, ws, ws_bias, splits_out, M, N, K, stream);
}

// ... with dy pre-scaled per TENSOR: dy_amax = device scalar holding max |dy| (cgg_absmax_f32; nullable = the fixed 2^4), ws_bias
// nullable. grad_output is not unit scale: with the fixed 2^4 a gradient of magnitude 1e-6 keeps ~10 of its 22 bits (x3.h).
extern "C" int cgg_wgrad_x3_scaled(const float* dy, int ldy, const float* dy_amax, const float* x, int ldx, float* ws, float* ws_bias,
                                   int* splits_out, int M, int N, int K, cgg_stream_t stream) {
  CGG_REQUIRE(!ws_bias || cgg_aligned16(ws_bias), CGG_EINVAL, "cgg_wgrad_x3_scaled: ws_bias must be 16-B aligned");
  return wgrad_launch(dy, ldy, x, ldx, ws, ws_bias, splits_out, M, N, K, stream, dy_amax);
}
