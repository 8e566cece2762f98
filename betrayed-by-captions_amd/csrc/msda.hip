// K1/K2: multi-scale deformable attention sampling + aggregation for gfx950.
// Replaces [3P] mmcv ms_deform_attn_forward/backward (MultiScaleDeformableAttnFunction), reached
// from MSDeformAttnPixelDecoder.forward built at open_set/models/mask2former_head.py:112-117.
//
// Roofline: a gather -- HBM/L2-bound, no MFMA. Mapping: one lane owns 4 consecutive channels of one
// (query, head): with H=8, D=32 a wavefront is exactly one query, so
//   * every corner fetch is a 16-B load per lane = one full 128-B line per head (coalesced),
//   * the query's sampling row (offsets/weights) is read as broadcast loads,
//   * the 1-KiB output row leaves as one contiguous store per wave.
// Blocks are remapped so that each XCD walks ONE contiguous range of queries (row-major within a
// level): its private 4-MiB L2 then only has to hold a spatial band of every value level.
#include "msda_common.h"
#include <type_traits>

__device__ __forceinline__ float msda_x(float v) { return v; }
__device__ __forceinline__ float msda_x(uint16_t v) { return cgg_bf2f(v); }

// per-lane channel slice: 16 bytes of the value row -> CPL = 4 (f32) or 8 (bf16) channels
template <typename VT> struct MsdaVec;
template <> struct MsdaVec<float> {
  static constexpr int CPL = 4;
  __device__ static __forceinline__ void fma(float (&acc)[4], float w, const float* p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = fmaf(w, v[c], acc[c]);
  }
};
template <> struct MsdaVec<uint16_t> {
  static constexpr int CPL = 8;
  __device__ static __forceinline__ void fma(float (&acc)[8], float w, const uint16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t q[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[2 * c] = fmaf(w, __uint_as_float(q[c] << 16), acc[2 * c]);
      acc[2 * c + 1] = fmaf(w, __uint_as_float(q[c] & 0xffff0000u), acc[2 * c + 1]);
    }
  }
};

// FUSED: offs_logits row = [H*L*P*2 raw offsets | H*L*P raw logits]; loc = ref + off/(W_l,H_l),
// weights = softmax over L*P. Otherwise loc/attw are given (mmcv boundary).
// P_ > 0: compile-time points per level: the P_*4 corner loads of one level are issued together (16 x 16 B
// in flight per lane); the level loop stays rolled so the register footprint (~100 VGPR) keeps >= 4
// waves per SIMD resident -- a fully unrolled 48-load body needs 256 VGPRs and halves the throughput.
template <typename VT, int L_, int P_, bool FUSED, typename OT = float, typename XT = float>
__global__ __launch_bounds__(256) void cgg_msda_fwd_kernel(
    const VT* __restrict__ value, MsdaLevels lv, const XT* __restrict__ loc,
    const XT* __restrict__ attw, const float* __restrict__ ref, int ld,
    OT* __restrict__ out, int Nv, int H, int D, int Lrt, int Nq, int Prt, long long total) {
  constexpr int CPL = MsdaVec<VT>::CPL;
  const int L = L_ > 0 ? L_ : Lrt;
  const int P = P_ > 0 ? P_ : Prt;
  const int DQ = D / CPL;
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const long long gid = (long long)bid * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cq = (int)(gid % DQ);
  const int h = (int)((gid / DQ) % H);
  const long long bq = gid / ((long long)DQ * H);
  const int b = (int)(bq / Nq);
  const int q = (int)(bq - (long long)b * Nq);
  const size_t rowstride = (size_t)H * D;
  const VT* vb = value + (size_t)b * Nv * rowstride + (size_t)h * D + cq * CPL;
  const int LP = L * P;

  const XT* lp;
  const XT* wp;
  float rx = 0.f, ry = 0.f, smax = 0.f, sinv = 1.f;
  if (FUSED) {
    const XT* row = loc + (size_t)bq * ld;
    lp = row + (size_t)h * LP * 2;
    wp = row + (size_t)H * LP * 2 + (size_t)h * LP;
    rx = ref[2 * q];
    ry = ref[2 * q + 1];
    smax = msda_x(wp[0]);
    for (int i = 1; i < LP; ++i) smax = fmaxf(smax, msda_x(wp[i]));
    float ssum = 0.f;
    for (int i = 0; i < LP; ++i) ssum += __expf(msda_x(wp[i]) - smax);
    sinv = 1.f / ssum;
  } else {
    lp = loc + ((size_t)bq * H + h) * LP * 2;
    wp = attw + ((size_t)bq * H + h) * LP;
  }

  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;

#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const VT* vl = vb + (size_t)lv.start[l] * rowstride;
#pragma unroll
    for (int p = 0; p < (P_ > 0 ? P_ : 1); ++p) {
      for (int pp = (P_ > 0 ? p : 0); pp < (P_ > 0 ? p + 1 : P); ++pp) {
        const int i = l * P + pp;
        float x = msda_x(lp[2 * i]), y = msda_x(lp[2 * i + 1]), w = msda_x(wp[i]);
        if (FUSED) {
          x = rx + x / (float)Wl;
          y = ry + y / (float)Hl;
          w = __expf(w - smax) * sinv;      // v_exp_f32 path: the libm expf costs ~15 VALU, 24 of them per lane
        }
        const MsdaTap t = cgg_msda_tap(x, y, Hl, Wl);
        float s[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) s[c] = 0.f;
        MsdaVec<VT>::fma(s, t.w00, vl + (size_t)t.o00 * rowstride);
        MsdaVec<VT>::fma(s, t.w01, vl + (size_t)t.o01 * rowstride);
        MsdaVec<VT>::fma(s, t.w10, vl + (size_t)t.o10 * rowstride);
        MsdaVec<VT>::fma(s, t.w11, vl + (size_t)t.o11 * rowstride);
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, s[c], acc[c]);
      }
    }
  }
  OT* op = out + (size_t)bq * rowstride + (size_t)h * D + cq * CPL;
  if (sizeof(OT) == 4) {
#pragma unroll
    for (int c = 0; c < CPL; c += 4)
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(op) + c) = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
  } else {
#pragma unroll
    for (int c = 0; c < CPL; c += 4)
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(op) + c) =
          make_uint2(cgg_pack2(cgg_f2bf(acc[c]), cgg_f2bf(acc[c + 1])), cgg_pack2(cgg_f2bf(acc[c + 2]), cgg_f2bf(acc[c + 3])));
  }
}

// Second encoder-stream specialisation. PMC (profiles/r2_pmc_sq_*): the kernel above is INSTRUCTION-ISSUE bound -- the
// SIMDs issue ~100 % of the kernel's cycles (a wave64 VALU instruction occupies a 16-lane SIMD for 4 cycles), 20 % of the
// wave cycles wait on memory -- so this version removes instructions instead of bytes:
//   * the 4 lanes of a (query, head) each compute the tap geometry of ONE of the level's 4 points (corner offsets, bilinear
//     weights already multiplied by the point's attention probability) and broadcast it to their quad with DPP quad_perm
//     moves: 1 geometry evaluation + 32 moves per lane and level instead of 4 evaluations (~200 instructions);
//   * the bilinear / attention weights are folded before the channel loop and the accumulation runs on v_pk_fma_f32
//     (two channels per instruction): 16 packed FMAs per point instead of 40 scalar ones.
// Same inputs / outputs; the summation order differs from the kernel above (w_p c_k folded first), so results agree to
// rounding of the f32 accumulation (<= 1 bf16 ulp on the output), not bit for bit.
typedef float msda_v2f __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float msda_quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int msda_quad_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float msda_quad_bcast_ctrl(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// sum over the 8 lanes of an aligned lane octet by DPP (no LDS): every lane ends up with the octet's sum
__device__ __forceinline__ void msda_reduce8(float& v) {
  v += msda_quad_bcast_ctrl<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v += msda_quad_bcast_ctrl<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v += msda_quad_bcast_ctrl<0x141>(v);      // row_half_mirror
}

template <int PSEL>
__device__ __forceinline__ void msda_point_gather(msda_v2f (&acc)[4], const uint16_t* __restrict__ vl, const int (&my_o)[4],
                                                  const float (&my_w)[4]) {
  constexpr int CTRL = PSEL | (PSEL << 2) | (PSEL << 4) | (PSEL << 6);      // quad_perm: every lane reads lane PSEL of its quad
  uint4 u[4];
  float w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = msda_quad_bcast<CTRL>(my_o[k]);
    w[k] = msda_quad_bcast<CTRL>(my_w[k]);
    u[k] = *reinterpret_cast<const uint4*>(vl + o);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t q[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
    const msda_v2f ww = {w[k], w[k]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const msda_v2f vv = {__uint_as_float(q[c] << 16), __uint_as_float(q[c] & 0xffff0000u)};
      acc[c] = __builtin_elementwise_fma(vv, ww, acc[c]);
    }
  }
}

// HM = true: `value` is HEAD-MAJOR, (B, H, Nv, 32): the two x-neighbours of a bilinear tap are then adjacent 64-byte pieces (one
// 128-byte line when x0 is even) and a line holds only this head's pixels -- in the row layout (B, Nv, H, 32) every 64-byte tap
// drags the other half of its line (another head's channels, which that head samples elsewhere) through L2 and L1.
template <bool HM>
__global__ __launch_bounds__(256) void cgg_msda_fwd_stream2_kernel(
    const uint16_t* __restrict__ value, MsdaLevels lv, const uint16_t* __restrict__ rows, const float* __restrict__ ref,
    int ld, uint16_t* __restrict__ out, int Nv, int Nq, unsigned total) {
  constexpr int D = 32, CPL = 8, H = 8, L = 3, LP = 12;
  // 32-bit index arithmetic with H = 8 and 4 lanes per (query, head) as compile-time shifts: the generic kernel's
  // 64-bit `gid / (DQ * H)` and `bq / Nq` expand to software division loops (~25 % of its instructions)
  unsigned gid = (unsigned)cgg_xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  int cq, h;
  unsigned bq;
  if constexpr (HM) {
    // a wavefront = 16 consecutive queries of ONE head (wave-uniform plane): the 16 tap pieces of a load instruction lie in the
    // same plane, near each other, instead of in 8 planes; a block = 4 heads of a 16-query group, two blocks per group
    const unsigned blk = gid >> 8, t = threadIdx.x;
    cq = (int)(t & 3u);
    h = (int)((blk & 1u) * 4u + (t >> 6));
    bq = (blk >> 1) * 16u + ((t >> 2) & 15u);
    if (bq >= total / 32u) return;        // whole quads
  } else {
    if (gid >= total) return;             // total is a multiple of 4: quads are never split
    cq = (int)(gid & 3u);
    h = (int)((gid >> 2) & 7u);
    bq = gid >> 5;
  }
  const unsigned b = bq / (unsigned)Nq;
  const int q = (int)(bq - b * (unsigned)Nq);
  constexpr int rowstride = HM ? D : H * D;              // elements between two pixels of this head
  const uint16_t* vb = HM ? value + ((size_t)b * H + h) * Nv * D + cq * CPL
                          : value + (size_t)b * Nv * (H * D) + (size_t)h * D + cq * CPL;
  const uint16_t* row = rows + (size_t)bq * ld;
  const uint16_t* lp = row + (size_t)h * LP * 2 + 2 * cq;              // this lane's point: (x, y) of point cq, + 8 per level
  const uint16_t* wp = row + (size_t)H * LP * 2 + (size_t)h * LP;
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  float e[LP];
  {
    const uint2 a = *reinterpret_cast<const uint2*>(wp), c = *reinterpret_cast<const uint2*>(wp + 4),
                d = *reinterpret_cast<const uint2*>(wp + 8);
    const uint32_t u[6] = {a.x, a.y, c.x, c.y, d.x, d.y};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      e[2 * i] = __uint_as_float(u[i] << 16);
      e[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
  }
  float smax = e[0];
#pragma unroll
  for (int i = 1; i < LP; ++i) smax = fmaxf(smax, e[i]);
  float ssum = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    e[i] = __expf(e[i] - smax);
    ssum += e[i];
  }
  const float sinv = 1.f / ssum;
  // attention probability of THIS lane's point on each level
  float pw[L];
#pragma unroll
  for (int l = 0; l < L; ++l)
    pw[l] = (cq == 0 ? e[4 * l] : (cq == 1 ? e[4 * l + 1] : (cq == 2 ? e[4 * l + 2] : e[4 * l + 3]))) * sinv;

  msda_v2f acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = msda_v2f{0.f, 0.f};
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const uint16_t* vl = vb + (size_t)lv.start[l] * rowstride;
    const uint32_t o = *reinterpret_cast<const uint32_t*>(lp + 8 * l);
    const float x = rx + __uint_as_float(o << 16) / (float)Wl;
    const float y = ry + __uint_as_float(o & 0xffff0000u) / (float)Hl;
    const MsdaTap t = cgg_msda_tap(x, y, Hl, Wl);
    const float wl = l == 0 ? pw[0] : (l == 1 ? pw[1] : pw[2]);
    const int my_o[4] = {t.o00 * rowstride, t.o01 * rowstride, t.o10 * rowstride, t.o11 * rowstride};
    const float my_w[4] = {t.w00 * wl, t.w01 * wl, t.w10 * wl, t.w11 * wl};
    msda_point_gather<0>(acc, vl, my_o, my_w);
    msda_point_gather<1>(acc, vl, my_o, my_w);
    msda_point_gather<2>(acc, vl, my_o, my_w);
    msda_point_gather<3>(acc, vl, my_o, my_w);
  }
  uint16_t* op = out + (size_t)bq * (H * D) + (size_t)h * D + cq * CPL;
  *reinterpret_cast<uint4*>(op) =
      make_uint4(cgg_pack2(cgg_f2bf(acc[0][0]), cgg_f2bf(acc[0][1])), cgg_pack2(cgg_f2bf(acc[1][0]), cgg_f2bf(acc[1][1])),
                 cgg_pack2(cgg_f2bf(acc[2][0]), cgg_f2bf(acc[2][1])), cgg_pack2(cgg_f2bf(acc[3][0]), cgg_f2bf(acc[3][1])));
}

// Parity mode's twin of the kernel above: F32 values (B, Nv, 8, 32) -- in f32 a head's 32 channels of a pixel ARE one 128-byte
// line, so the row layout already has the property the head-major bf16 layout was introduced for --, f32 offset / logit rows,
// f32 output. Same lane geometry (4 lanes per (query, head), each evaluates one point's taps and broadcasts them with DPP,
// 8 channels per lane = two 16-byte loads per tap), same arithmetic order as the bf16 kernel with exact f32 operands; replaces
// the generic cgg_msda_fwd_kernel<float> (one lane per 4 channels, software 64-bit divisions) in the x3 encoder stream.
template <int PSEL>
__device__ __forceinline__ void msda_point_gather_f32(msda_v2f (&acc)[4], const float* __restrict__ vl, const int (&my_o)[4],
                                                      const float (&my_w)[4]) {
  constexpr int CTRL = PSEL | (PSEL << 2) | (PSEL << 4) | (PSEL << 6);      // quad_perm: every lane reads lane PSEL of its quad
  f32x4 u[4][2];
  float w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = msda_quad_bcast<CTRL>(my_o[k]);
    w[k] = msda_quad_bcast<CTRL>(my_w[k]);
    u[k][0] = *reinterpret_cast<const f32x4*>(vl + o);
    u[k][1] = *reinterpret_cast<const f32x4*>(vl + o + 16);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const msda_v2f ww = {w[k], w[k]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const msda_v2f vv = {u[k][c >> 1][2 * (c & 1)], u[k][c >> 1][2 * (c & 1) + 1]};
      acc[c] = __builtin_elementwise_fma(vv, ww, acc[c]);
    }
  }
}

// ... with the level base in scalar registers and the taps as 32-bit BYTE offsets (T2D == 2: image and head are block-uniform): the
// loads take the `saddr + voffset` form, no 64-bit pointer arithmetic per tap (28 + 17 of the ~200 VALU instructions per level)
template <int PSEL>
__device__ __forceinline__ void msda_point_gather_f32_b(msda_v2f (&acc)[4], const char* __restrict__ vlb, const uint32_t (&my_o)[4],
                                                        const float (&my_w)[4], uint32_t lane_b) {
  constexpr int CTRL = PSEL | (PSEL << 2) | (PSEL << 4) | (PSEL << 6);
  f32x4 u[4][2];
  float w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t o = (uint32_t)msda_quad_bcast<CTRL>((int)my_o[k]) + lane_b;      // (the lane's own channel offset: AFTER the broadcast)
    w[k] = msda_quad_bcast<CTRL>(my_w[k]);
    u[k][0] = *reinterpret_cast<const f32x4*>(vlb + o);
    u[k][1] = *reinterpret_cast<const f32x4*>(vlb + o + 64u);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const msda_v2f ww = {w[k], w[k]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const msda_v2f vv = {u[k][c >> 1][2 * (c & 1)], u[k][c >> 1][2 * (c & 1) + 1]};
      acc[c] = __builtin_elementwise_fma(vv, ww, acc[c]);
    }
  }
}

// T2D: 0 = a wave is 16 consecutive queries, the block's 4 waves are 4 heads; 1 = a wave is a 4 x 4 pixel tile of its level, the
// block's 4 waves are 4 heads of it; 2 = a wave is a 4 x 4 tile, the block's 4 waves are the four tiles of an 8 x 8 pixel block
// of ONE head (their taps share a (8 + 2 r)^2 window of lines), blockIdx enumerates (8 x 8 block, head)
// VS: `value` rows have a run-time stride of vld floats (the value columns of a wider row: round 6's merged projection GEMM writes
// [value | offsets | logits] rows of 544 floats; a head's 32 channels stay one aligned 128-byte line as long as vld % 32 == 0)
template <int T2D, bool VS = false>
__global__ __launch_bounds__(256) void cgg_msda_fwd_stream2_f32_kernel(
    const float* __restrict__ value, MsdaLevels lv, const float* __restrict__ rows, const float* __restrict__ ref,
    int ld, float* __restrict__ out, int Nv, int Nq, unsigned total, int vld) {
  constexpr int D = 32, CPL = 8, H = 8, L = 3, LP = 12;
  // a wavefront = 16 consecutive queries of ONE head: in f32 a head's slice of a pixel is one 128-byte line, so the x-neighbour
  // taps of neighbouring queries are the SAME lines (the (x + 1) corner of query i is the x corner of query i + 1) and meet in
  // L1; a block = 4 heads of a 16-query group, two blocks per group (the geometry of the head-major bf16 kernel): 127 -> 121 us
  const unsigned blk = (unsigned)cgg_xcd_remap(blockIdx.x, gridDim.x), t = threadIdx.x;
  const int cq = (int)(t & 3u);
  int h = (int)((blk & 1u) * 4u + (t >> 6));
  unsigned bq = (blk >> 1) * 16u + ((t >> 2) & 15u);
  if constexpr (T2D == 2) {
    // blk = (image, 8 x 8 block of a level, head); wave w = tile (w >> 1, w & 1) of the block
    h = (int)(blk & 7u);
    const unsigned blocks_img = (unsigned)Nq >> 6;
    const unsigned bb = blk >> 3, bimg = bb / blocks_img, bl = bb - bimg * blocks_img;     // block index inside the image
    const int l = (int)(bl << 6) >= lv.start[2] ? 2 : ((int)(bl << 6) >= lv.start[1] ? 1 : 0);
    const int g8 = (int)bl - (lv.start[l] >> 6), bx_n = lv.w[l] >> 3;
    const int by = g8 / bx_n, bx = g8 - by * bx_n;
    const int w = (int)(t >> 6), i = (int)((t >> 2) & 15u);
    const int py = 8 * by + 4 * (w >> 1) + (i >> 2), px = 8 * bx + 4 * (w & 1) + (i & 3);
    bq = bimg * (unsigned)Nq + (unsigned)(lv.start[l] + py * lv.w[l] + px);
  }
  if (bq >= total / 32u) return;          // whole quads
  const unsigned b = bq / (unsigned)Nq;
  int q = (int)(bq - b * (unsigned)Nq);
  if constexpr (T2D == 1) {
    // the wave's 16 queries are a 4 x 4 pixel tile of their level instead of a 16 x 1 strip (every level: W, H % 4 == 0,
    // start % 16 == 0; queries == pixels): the taps of a tile fall into a (4 + 2 r)^2 window instead of (16 + 2 r) x (1 + 2 r)
    const int g = q >> 4, i = q & 15;
    const int l = (g << 4) >= lv.start[2] ? 2 : ((g << 4) >= lv.start[1] ? 1 : 0);
    const int gl = g - (lv.start[l] >> 4), tx_n = lv.w[l] >> 2;
    const int ty = gl / tx_n, tx = gl - ty * tx_n;
    q = lv.start[l] + (4 * ty + (i >> 2)) * lv.w[l] + 4 * tx + (i & 3);
    bq = b * (unsigned)Nq + (unsigned)q;
  }
  const int rowstride = VS ? vld : H * D;                    // (a compile-time constant unless VS)
  // lane cq owns channels 4 cq .. + 3 and 16 + 4 cq .. + 3: each of a tap's two loads covers a CONTIGUOUS 64-byte half line per quad
  const float* vb = value + (size_t)b * Nv * rowstride + (size_t)h * D + cq * 4;
  // T2D == 2: image and head are block-uniform -> the level base lives in scalar registers, the lane's channel offset in the taps
  const float* vbu = value + (size_t)__builtin_amdgcn_readfirstlane((int)b) * Nv * rowstride +
                     (size_t)__builtin_amdgcn_readfirstlane(h) * D;
  const uint32_t lane_b = (uint32_t)cq * 16u;
  const float* row = rows + (size_t)bq * ld;
  const float* lp = row + (size_t)h * LP * 2 + 2 * cq;                 // this lane's point: (x, y) of point cq, + 8 per level
  const float* wp = row + (size_t)H * LP * 2 + (size_t)h * LP;
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  float e[LP];
  {
    const f32x4 a = *reinterpret_cast<const f32x4*>(wp), c = *reinterpret_cast<const f32x4*>(wp + 4),
                d = *reinterpret_cast<const f32x4*>(wp + 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      e[i] = a[i];
      e[4 + i] = c[i];
      e[8 + i] = d[i];
    }
  }
  float smax = e[0];
#pragma unroll
  for (int i = 1; i < LP; ++i) smax = fmaxf(smax, e[i]);
  float ssum = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    e[i] = __expf(e[i] - smax);
    ssum += e[i];
  }
  const float sinv = 1.f / ssum;
  float pw[L];                              // attention probability of THIS lane's point on each level
#pragma unroll
  for (int l = 0; l < L; ++l)
    pw[l] = (cq == 0 ? e[4 * l] : (cq == 1 ? e[4 * l + 1] : (cq == 2 ? e[4 * l + 2] : e[4 * l + 3]))) * sinv;

  msda_v2f acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = msda_v2f{0.f, 0.f};
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const float* vl = vb + (size_t)lv.start[l] * rowstride;
    const float2 o = *reinterpret_cast<const float2*>(lp + 8 * l);
    const float x = rx + o.x / (float)Wl;
    const float y = ry + o.y / (float)Hl;
    const MsdaTap t = cgg_msda_tap(x, y, Hl, Wl);
    const float wl = l == 0 ? pw[0] : (l == 1 ? pw[1] : pw[2]);
    const float my_w[4] = {t.w00 * wl, t.w01 * wl, t.w10 * wl, t.w11 * wl};
    if constexpr (T2D == 2) {
      const char* vlb = reinterpret_cast<const char*>(vbu + (size_t)lv.start[l] * rowstride);
      const uint32_t rs4 = (uint32_t)rowstride * 4u;
      const uint32_t my_ob[4] = {(uint32_t)t.o00 * rs4, (uint32_t)t.o01 * rs4, (uint32_t)t.o10 * rs4, (uint32_t)t.o11 * rs4};
      msda_point_gather_f32_b<0>(acc, vlb, my_ob, my_w, lane_b);
      msda_point_gather_f32_b<1>(acc, vlb, my_ob, my_w, lane_b);
      msda_point_gather_f32_b<2>(acc, vlb, my_ob, my_w, lane_b);
      msda_point_gather_f32_b<3>(acc, vlb, my_ob, my_w, lane_b);
    } else {
      const int my_o[4] = {t.o00 * rowstride, t.o01 * rowstride, t.o10 * rowstride, t.o11 * rowstride};
      msda_point_gather_f32<0>(acc, vl, my_o, my_w);
      msda_point_gather_f32<1>(acc, vl, my_o, my_w);
      msda_point_gather_f32<2>(acc, vl, my_o, my_w);
      msda_point_gather_f32<3>(acc, vl, my_o, my_w);
    }
  }
  float* op = out + (size_t)bq * (H * D) + (size_t)h * D + cq * 4;
  *reinterpret_cast<f32x4*>(op) = f32x4{acc[0][0], acc[0][1], acc[1][0], acc[1][1]};
  *reinterpret_cast<f32x4*>(op + 16) = f32x4{acc[2][0], acc[2][1], acc[3][0], acc[3][1]};
}

// Backward (f32). Lane group of DQ lanes = one (query, head): the channel reductions for
// grad_loc / grad_attn are wave shuffles; grad_value is scattered with hardware f32 atomics.
// GV = false: grad_loc / grad_attn only (the gather half of the split backward; grad_value comes from the tiled scatter kernel)
template <int P_, bool GV = true>
__global__ __launch_bounds__(256) void cgg_msda_bwd_kernel(
    const float* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
    const float* __restrict__ attw, const float* __restrict__ gout, float* __restrict__ gvalue,
    float* __restrict__ gloc, float* __restrict__ gattw, int Nv, int H, int D, int L, int Nq,
    int Prt, long long total) {
  const int P = P_ > 0 ? P_ : Prt;
  const int DQ = D >> 2;
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const long long gid = (long long)bid * 256 + threadIdx.x;
  const bool live = gid < total;
  const long long g2 = live ? gid : total - 1;  // keep every lane in the shuffles
  const int cq = (int)(g2 % DQ);
  const int h = (int)((g2 / DQ) % H);
  const long long bq = g2 / ((long long)DQ * H);
  const int b = (int)(bq / Nq);
  const size_t rowstride = (size_t)H * D;
  const size_t coff = (size_t)h * D + cq * 4;
  const float* vb = value + (size_t)b * Nv * rowstride + coff;
  float* gvb = gvalue + (size_t)b * Nv * rowstride + coff;
  const int LP = L * P;
  const float* lp = loc + ((size_t)bq * H + h) * LP * 2;
  const float* wp = attw + ((size_t)bq * H + h) * LP;
  float* glp = gloc + ((size_t)bq * H + h) * LP * 2;
  float* gwp = gattw + ((size_t)bq * H + h) * LP;
  f32x4 g = cgg_ld4(gout + (size_t)bq * rowstride + coff);
  if (!live) g = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const float* vl = vb + (size_t)lv.start[l] * rowstride;
    float* gvl = gvb + (size_t)lv.start[l] * rowstride;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      const float x = lp[2 * i], y = lp[2 * i + 1], w = wp[i];
      const MsdaTap t = cgg_msda_tap(x, y, Hl, Wl);
      const f32x4 v00 = cgg_ld4(vl + (size_t)t.o00 * rowstride);
      const f32x4 v01 = cgg_ld4(vl + (size_t)t.o01 * rowstride);
      const f32x4 v10 = cgg_ld4(vl + (size_t)t.o10 * rowstride);
      const f32x4 v11 = cgg_ld4(vl + (size_t)t.o11 * rowstride);
      const float hh = 1.f - t.lh, hw = 1.f - t.lw;
      // validity flags recovered from the tap (weights can be legitimately 0 on integer coords,
      // so recompute instead of testing w != 0)
      const float him = y * (float)Hl - 0.5f, wim = x * (float)Wl - 0.5f;
      const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
      const bool vh0 = t.in && h0 >= 0, vh1 = t.in && (h0 + 1) <= Hl - 1;
      const bool vw0 = w0 >= 0, vw1 = (w0 + 1) <= Wl - 1;
      const float f00 = (vh0 && vw0) ? 1.f : 0.f, f01 = (vh0 && vw1) ? 1.f : 0.f;
      const float f10 = (vh1 && vw0) ? 1.f : 0.f, f11 = (vh1 && vw1) ? 1.f : 0.f;
      // d(val)/d(w_im), d(val)/d(h_im) per channel, dotted with grad_out
      float dotv = 0.f, dotx = 0.f, doty = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float a00 = f00 * v00[c], a01 = f01 * v01[c], a10 = f10 * v10[c], a11 = f11 * v11[c];
        const float val = hh * hw * a00 + hh * t.lw * a01 + t.lh * hw * a10 + t.lh * t.lw * a11;
        const float dw = hh * (a01 - a00) + t.lh * (a11 - a10);
        const float dh = hw * (a10 - a00) + t.lw * (a11 - a01);
        dotv += val * g[c];
        dotx += dw * g[c];
        doty += dh * g[c];
      }
      for (int o = 1; o < DQ; o <<= 1) {
        dotv += __shfl_xor(dotv, o);
        dotx += __shfl_xor(dotx, o);
        doty += __shfl_xor(doty, o);
      }
      if (live && cq == 0) {
        gwp[i] += dotv;
        glp[2 * i] += (float)Wl * w * dotx;
        glp[2 * i + 1] += (float)Hl * w * doty;
      }
      if (GV && live) {
        const f32x4 wg = w * g;
        if (t.w00 != 0.f) {
          float* d = gvl + (size_t)t.o00 * rowstride;
#pragma unroll
          for (int c = 0; c < 4; ++c) atomicAdd(d + c, t.w00 * wg[c]);
        }
        if (t.w01 != 0.f) {
          float* d = gvl + (size_t)t.o01 * rowstride;
#pragma unroll
          for (int c = 0; c < 4; ++c) atomicAdd(d + c, t.w01 * wg[c]);
        }
        if (t.w10 != 0.f) {
          float* d = gvl + (size_t)t.o10 * rowstride;
#pragma unroll
          for (int c = 0; c < 4; ++c) atomicAdd(d + c, t.w10 * wg[c]);
        }
        if (t.w11 != 0.f) {
          float* d = gvl + (size_t)t.o11 * rowstride;
#pragma unroll
          for (int c = 0; c < 4; ++c) atomicAdd(d + c, t.w11 * wg[c]);
        }
      }
    }
  }
}

// grad_loc / grad_attn for P = 4 (the gather half of the split backward): per level the (query, head)'s 8 location floats and 4
// weights arrive as three 16-byte loads, the 16 corner loads of its 4 points are all in flight before the first use, and the
// results leave as three 16-byte read-modify-writes whose reads were requested with the corner loads (the generic kernel's per-point
// scalar loads and `+=` chains ran this half at 2.3 ms per layer at configs[2]; the forward gathers the same taps in 0.16 ms x 6).
// ACC = false (round 5, the autograd path): grad_loc / grad_attn are WRITTEN, not accumulated -- no read of the old values, no
// zero-fill before the call (the accumulate contract of the mmcv entry point costs 0.4 GB of reads + 0.4 GB of fills per call at
// configs[2] shapes). PATCH = true (H == 8, D == 32, queries == pixels of a pyramid whose levels tile [0, Nq), w % 8, h % 4,
// start % 32 == 0): a block = ONE head of an 8 x 4 pixel patch of a level, the 8 heads of a patch in adjacent blocks -- the
// corner lines of neighbouring queries' taps are shared in L1 / L2 as in the forward's 2-D mapping, and the 8 blocks of a patch read
// the same loc / weight / grad_out rows.
#ifndef MSDA_GATHER_WAVES
#define MSDA_GATHER_WAVES 4  // minimum waves per SIMD asked of the register allocator (the kernel needs 85 registers: 5 waves; budgets 5 / 6 measured the same / 2 % slower)
#endif
template <bool ACC, bool PATCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MSDA_GATHER_WAVES))) void cgg_msda_bwd_gather4_kernel(
    const float* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc, const float* __restrict__ attw,
    const float* __restrict__ gout, float* __restrict__ gloc, float* __restrict__ gattw, int Nv, int H, int D, int L, int Nq,
    long long total, int vld) {
  const int DQ = D >> 2;
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const long long gid = (long long)bid * 256 + threadIdx.x;
  bool live = gid < total;
  const long long g2 = live ? gid : total - 1;  // keep every lane in the shuffles
  int cq = (int)(g2 % DQ);
  int h = (int)((g2 / DQ) % H);
  long long bq = g2 / ((long long)DQ * H);
  if constexpr (PATCH) {
    const int t = threadIdx.x;
    cq = t & 7;
    h = bid & 7;
    const int bb = bid >> 3, per_img = Nq >> 5;
    const int bimg = bb / per_img, bl = bb - bimg * per_img;
    int l = 0;
    for (int k = 1; k < L; ++k)
      if ((bl << 5) >= lv.start[k]) l = k;
    const int g = bl - (lv.start[l] >> 5), px_n = lv.w[l] >> 3;
    const int py = g / px_n, px = g - py * px_n;
    const int pair = t >> 3;
    bq = (long long)bimg * Nq + lv.start[l] + (4 * py + (pair >> 3)) * lv.w[l] + 8 * px + (pair & 7);
    live = true;
  }
  // (PATCH: a block is one head of one 8 x 4 patch -- image and head are block-uniform, the level base stays in scalar registers)
  const int b = PATCH ? __builtin_amdgcn_readfirstlane((int)(bq / Nq)) : (int)(bq / Nq);
  // value rows may be padded (vld floats per pixel, 0 = H D): a row stride that is a multiple of 512 bytes sends the corner lines of
  // neighbouring pixels to the same L2 channels -- 288 floats per row instead of 256 made the forward gather 20 % faster (round 6)
  const size_t rowstride = vld ? (size_t)vld : (size_t)H * D;
  const size_t coff = (size_t)h * D + cq * 4;
  const float* vb = value + (size_t)b * Nv * rowstride + (PATCH ? (size_t)h * D : coff);
  const uint32_t lane_b = PATCH ? (uint32_t)cq * 16u : 0u;      // PATCH: the lane's channel offset goes into the 32-bit corner offsets
  const int LP = L * 4;
  const float* lp = loc + ((size_t)bq * H + h) * LP * 2;
  const float* wp = attw + ((size_t)bq * H + h) * LP;
  float* glp = gloc + ((size_t)bq * H + h) * LP * 2;
  float* gwp = gattw + ((size_t)bq * H + h) * LP;
  f32x4 g = cgg_ld4(gout + (size_t)bq * ((size_t)H * D) + coff);
  if (!live) g = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool owner = live && cq == 0;
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const float* vl = vb + (size_t)lv.start[l] * rowstride;
    const f32x4 xy0 = cgg_ld4(lp + 8 * l), xy1 = cgg_ld4(lp + 8 * l + 4), w4 = cgg_ld4(wp + 4 * l);
    f32x4 ow = {0.f, 0.f, 0.f, 0.f}, ol0 = ow, ol1 = ow;
    if (ACC && owner) {
      ow = cgg_ld4(gwp + 4 * l);
      ol0 = cgg_ld4(glp + 8 * l);
      ol1 = cgg_ld4(glp + 8 * l + 4);
    }
    const float xs[4] = {xy0[0], xy0[2], xy1[0], xy1[2]}, ys[4] = {xy0[1], xy0[3], xy1[1], xy1[3]};
    const msda_v2f glo = {g[0], g[1]}, ghi = {g[2], g[3]};
    const uint32_t rs_b = (uint32_t)rowstride * 4u, row_b = (uint32_t)Wl * rs_b;      // bytes per pixel / per pixel row of the level
    // the level's four points in two PAIRS (8 corner loads in flight each): all 16 at once needed 130+ registers = 3 waves per SIMD
    auto pair = [&](auto p0c) {
      constexpr int P0 = decltype(p0c)::value;
      f32x4 v[2][4];
      float lh[2], lw[2];
      bool k[2][4];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int p = P0 + pp;
        const float him = ys[p] * (float)Hl - 0.5f, wim = xs[p] * (float)Wl - 0.5f;
        const bool in = (him > -1.f) && (wim > -1.f) && (him < (float)Hl) && (wim < (float)Wl);
        const float hf = floorf(him), wf = floorf(wim);
        const int h0 = (int)hf, w0 = (int)wf;
        lh[pp] = him - hf;
        lw[pp] = wim - wf;
        const bool vh0 = in && h0 >= 0, vh1 = in && (h0 + 1) <= Hl - 1;
        const bool vw0 = w0 >= 0, vw1 = (w0 + 1) <= Wl - 1;
        k[pp][0] = vh0 && vw0; k[pp][1] = vh0 && vw1; k[pp][2] = vh1 && vw0; k[pp][3] = vh1 && vw1;
        // corner addresses as 32-bit byte offsets from the level base (a level of (B, Nv, H, 32) f32 values is far below 4 GiB):
        // one multiply for the clamped top-left corner, the other three differ by 0 / one pixel / one row (the clamps only ever
        // collapse a step to 0) -- the four size_t products per point were ~80 of the ~500 VALU instructions per level
        const int ch0 = min(max(h0, 0), Hl - 1), ch1 = min(max(h0 + 1, 0), Hl - 1);
        const int cw0 = min(max(w0, 0), Wl - 1), cw1 = min(max(w0 + 1, 0), Wl - 1);
        const uint32_t o00 = (uint32_t)(ch0 * Wl + cw0) * rs_b + lane_b;
        const uint32_t dxo = cw1 != cw0 ? rs_b : 0u, dyo = ch1 != ch0 ? row_b : 0u;
        const char* vlb = reinterpret_cast<const char*>(vl);
        v[pp][0] = cgg_ld4(reinterpret_cast<const float*>(vlb + o00));
        v[pp][1] = cgg_ld4(reinterpret_cast<const float*>(vlb + (o00 + dxo)));
        v[pp][2] = cgg_ld4(reinterpret_cast<const float*>(vlb + (o00 + dyo)));
        v[pp][3] = cgg_ld4(reinterpret_cast<const float*>(vlb + (o00 + dyo + dxo)));
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        constexpr int PB = P0;
        const int p = PB + pp;
        const float hh = 1.f - lh[pp], hw = 1.f - lw[pp];
        // (a corner outside the map was loaded from its clamped in-range address: its DOT is dropped below, four selects per point
        // instead of sixteen on the loaded channels)
        const f32x4 v00 = v[pp][0], v01 = v[pp][1], v10 = v[pp][2], v11 = v[pp][3];
        // the three gradients are combinations of FOUR corner dots d_k = sum_c v_k[c] g[c] (bilinear interpolation is linear in the
        // corner values): 16 FMAs + 4 reductions per point instead of forming val / d val / dx / d val / dy per channel (~64 VALU)
        // (packed: the natural register pairs of a loaded vector times (g0, g1) / (g2, g3), one horizontal add per corner -- the
        // per-channel form made the compiler pair channels of DIFFERENT corners and shuffle registers for it)
        auto dot4 = [&](const f32x4& vv) {
          msda_v2f t = msda_v2f{vv[0], vv[1]} * glo;
          t = __builtin_elementwise_fma(msda_v2f{vv[2], vv[3]}, ghi, t);
          return t[0] + t[1];
        };
        float d00 = dot4(v00), d01 = dot4(v01), d10 = dot4(v10), d11 = dot4(v11);
        d00 = k[pp][0] ? d00 : 0.f;
        d01 = k[pp][1] ? d01 : 0.f;
        d10 = k[pp][2] ? d10 : 0.f;
        d11 = k[pp][3] ? d11 : 0.f;
        if constexpr (PATCH) {
          // D == 32: the 8 lanes of a (query, head) reduce by three DPP moves (quad xor 1, quad xor 2, row_half_mirror: after the two
          // quad steps every lane holds its quad's sum and lane i of 8 reads lane 7 - i = the other quad) -- `__shfl_xor` is a
          // ds_bpermute on this target: 144 LDS-pipe instructions per wave in the first version of this kernel
          msda_reduce8(d00);
          msda_reduce8(d01);
          msda_reduce8(d10);
          msda_reduce8(d11);
        } else {
          for (int o = 1; o < DQ; o <<= 1) {
            d00 += __shfl_xor(d00, o);
            d01 += __shfl_xor(d01, o);
            d10 += __shfl_xor(d10, o);
            d11 += __shfl_xor(d11, o);
          }
        }
        const float dotv = hh * (hw * d00 + lw[pp] * d01) + lh[pp] * (hw * d10 + lw[pp] * d11);
        const float dotx = hh * (d01 - d00) + lh[pp] * (d11 - d10);
        const float doty = hw * (d10 - d00) + lw[pp] * (d11 - d01);
        ow[p] += dotv;
        if (PB == 0) {
          ol0[2 * pp] += (float)Wl * w4[p] * dotx;
          ol0[2 * pp + 1] += (float)Hl * w4[p] * doty;
        } else {
          ol1[2 * pp] += (float)Wl * w4[p] * dotx;
          ol1[2 * pp + 1] += (float)Hl * w4[p] * doty;
        }
      }
    };
    pair(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    pair(std::integral_constant<int, 2>{});
    if (owner) {
      *reinterpret_cast<f32x4*>(gwp + 4 * l) = ow;
      *reinterpret_cast<f32x4*>(glp + 8 * l) = ol0;
      *reinterpret_cast<f32x4*>(glp + 8 * l + 4) = ol1;
    }
  }
}

// CGG_MSDA_GENERIC=1 forces the generic kernels (the fallback of every other shape; tests run them on the stream's shapes too).
// Read ONCE, when the library is loaded -- no getenv in a launch path.
static const bool generic_only = getenv("CGG_MSDA_GENERIC") != nullptr;

// -------------------------------------------------------------------------------------------------
static int msda_read_levels(const int64_t* spatial_shapes, const int64_t* level_start, int L,
                            int Nv, hipStream_t s, MsdaLevels* lv, const char* who) {
  // The level table is kernel-argument data (<= 8 levels): copy it to the host once per call.
  // It is device-resident in the mmcv contract, hence the (tiny, stream-ordered) D2H.
  int64_t hs[16], st[8];
  hipError_t e = hipMemcpyAsync(hs, spatial_shapes, sizeof(int64_t) * 2 * L, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st, level_start, sizeof(int64_t) * L, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    cgg_set_error("%s: reading level table failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  long long tot = 0;
  for (int l = 0; l < L; ++l) {
    lv->h[l] = (int)hs[2 * l];
    lv->w[l] = (int)hs[2 * l + 1];
    lv->start[l] = (int)st[l];
    if (lv->h[l] <= 0 || lv->w[l] <= 0 || st[l] < 0 || st[l] + hs[2 * l] * hs[2 * l + 1] > Nv) {
      cgg_set_error("%s: level %d (%lldx%lld @%lld) does not fit Nv=%d", who, l, (long long)hs[2 * l],
                    (long long)hs[2 * l + 1], (long long)st[l], Nv);
      return CGG_EINVAL;
    }
    tot += hs[2 * l] * hs[2 * l + 1];
  }
  (void)tot;
  return CGG_OK;
}

// The ONE synchronising helper of the MSDeformAttn family: the mmcv-style device level table -> host int32 arrays for the
// *_hostlevels entry points. A binding calls it once per `spatial_shapes` tensor (once per forward), not once per op call.
extern "C" int cgg_msda_read_levels(const int64_t* spatial_shapes, const int64_t* level_start, int L, int Nv,
                                    int32_t* level_hw_host, int32_t* level_start_host, cgg_stream_t stream) {
  CGG_REQUIRE(spatial_shapes && level_start && level_hw_host && level_start_host, CGG_EINVAL, "cgg_msda_read_levels: null pointer");
  CGG_REQUIRE(L >= 1 && L <= 8 && Nv > 0, CGG_EINVAL, "cgg_msda_read_levels: bad sizes (L=%d, Nv=%d)", L, Nv);
  MsdaLevels lv;
  const int rc = msda_read_levels(spatial_shapes, level_start, L, Nv, (hipStream_t)stream, &lv, "cgg_msda_read_levels");
  if (rc) return rc;
  for (int l = 0; l < L; ++l) {
    level_hw_host[2 * l] = lv.h[l];
    level_hw_host[2 * l + 1] = lv.w[l];
    level_start_host[l] = lv.start[l];
  }
  return CGG_OK;
}

// Host-side level table variant: identical kernels, no D2H (used by the graph-captured forward).
extern "C" int cgg_msda_forward_hostlevels(const void* value, const int32_t* level_hw,
                                           const int32_t* level_start, const float* sampling_loc,
                                           const float* attn_weight, const float* ref_points, int ld,
                                           float* out, int B, int Nv, int H, int D, int L, int Nq,
                                           int P, int value_dtype, int fused, cgg_stream_t stream);

static int msda_fwd_launch(const void* value, const MsdaLevels& lv, const float* loc,
                           const float* attw, const float* ref, int ld, float* out, int B, int Nv,
                           int H, int D, int L, int Nq, int P, int dtype, bool fused, hipStream_t s, int vld = 0) {
  if (vld == H * D) vld = 0;                                // 0 = the packed (B, Nv, H, D) layout
  const int cpl = dtype == CGG_F32 ? 4 : 8;
  const long long total = (long long)B * Nq * H * (D / cpl);
  const int nblk = (int)((total + 255) / 256);
#define CGG_MSDA_LAUNCH(VT, LT, PT, FU)                                                          \
  hipLaunchKernelGGL((cgg_msda_fwd_kernel<VT, LT, PT, FU>), dim3(nblk), dim3(256), 0, s,         \
                     (const VT*)value, lv, loc, attw, ref, ld, out, Nv, H, D, L, Nq, P, total)
  const bool st = (L == 3 && P == 4);  // the shipped configs: num_levels=3, num_points=4
  // the encoder stream's shape (8 heads x 32 channels, 3 levels x 4 points, rows = [offsets | logits]): quad-shared taps
  const long long total8 = (long long)B * Nq * H * (D / 8);
  if (dtype == CGG_F32 && fused && st && H == 8 && D == 32 && ld % 4 == 0 && cgg_aligned16(loc) && !generic_only &&
      (long long)Nv * (vld ? vld : H * D) < (1ll << 31) && total8 < (1ll << 31) && vld % 32 == 0) {
    // Query -> lane mapping when the queries are the pixels of a pyramid whose levels allow it (the encoder), all bit-identical
    // (round 4; scratch/msda_f32_bench.py, init offsets / +-1 px of noise on them):
    //   16 x 1 strips, a block = 4 heads of a strip (round 3)                                   118-119 us / 132 us
    //   4 x 4 tiles per wave, a block = 4 heads of a tile                                        118 us     / 122 us
    //   4 x 4 tiles per wave, a block = the four tiles of an 8 x 8 pixel block of ONE head       114 us     / 116 us
    //   ... with the occupancy capped at 4 blocks per CU by a 36-KB dynamic LDS request          109 us     / 111 us
    // (caps of 26 / 32 / 40 KB: 112 / 110 / 110 us; 44-53 KB = 3 blocks: 114 us; 64 KB = 2 blocks: 134 us). The block's taps then
    // share a (8 + 2 r)^2 window of 128-B lines that a 16-wave CU keeps in its L1; the kernel stays L2-request bound (every tap
    // of the first toucher is a 128-B request), <= 80 us is out of reach for f32 values.
    bool t2d = Nq == Nv && Nq % 16 == 0;
    // the 2-D mappings rebuild q as start[l] + tile arithmetic: the levels must tile [0, Nq) in ascending order without gaps
    // (msda_read_levels only checks st + h w <= Nv; a gapped / reordered table would compute some queries twice and skip others)
    long long expect = 0;
    for (int l = 0; l < L && t2d; ++l) {
      t2d = lv.start[l] == expect;
      expect += (long long)lv.h[l] * lv.w[l];
    }
    t2d = t2d && expect == Nq;
    for (int l = 0; l < L && t2d; ++l) t2d = lv.w[l] % 4 == 0 && lv.h[l] % 4 == 0 && lv.start[l] % 16 == 0;
    bool t8 = t2d && Nq % 64 == 0;
    for (int l = 0; l < L && t8; ++l) t8 = lv.w[l] % 8 == 0 && lv.h[l] % 8 == 0 && lv.start[l] % 64 == 0;
    const unsigned nb = 2 * (unsigned)(((long long)B * Nq + 15) / 16);
#define CGG_S2(T, LDS)                                                                                                          \
  do {                                                                                                                          \
    if (vld)                                                                                                                    \
      hipLaunchKernelGGL((cgg_msda_fwd_stream2_f32_kernel<T, true>), dim3(nb), dim3(256), LDS, s, (const float*)value, lv, loc, ref, ld, \
                         out, Nv, Nq, (unsigned)total8, vld);                                                                  \
    else                                                                                                                        \
      hipLaunchKernelGGL((cgg_msda_fwd_stream2_f32_kernel<T, false>), dim3(nb), dim3(256), LDS, s, (const float*)value, lv, loc, ref, ld, \
                         out, Nv, Nq, (unsigned)total8, 0);                                                                    \
  } while (0)
    if (t8) CGG_S2(2, 36000);
    else if (t2d) CGG_S2(1, 0);
    else CGG_S2(0, 0);
#undef CGG_S2
    CGG_CHECK_LAUNCH("cgg_msda_forward");
    return CGG_OK;
  }
  CGG_REQUIRE(vld == 0, CGG_EUNSUPPORTED, "cgg_msda_forward: strided value rows (vld=%d) need the f32 fused stream kernel (H=8, D=32, L=3, P=4)", vld);
  if (dtype == CGG_F32) {
    if (fused) { if (st) CGG_MSDA_LAUNCH(float, 3, 4, true); else CGG_MSDA_LAUNCH(float, 0, 0, true); }
    else       { if (st) CGG_MSDA_LAUNCH(float, 3, 4, false); else CGG_MSDA_LAUNCH(float, 0, 0, false); }
  } else {
    if (fused) { if (st) CGG_MSDA_LAUNCH(uint16_t, 3, 4, true); else CGG_MSDA_LAUNCH(uint16_t, 0, 0, true); }
    else       { if (st) CGG_MSDA_LAUNCH(uint16_t, 3, 4, false); else CGG_MSDA_LAUNCH(uint16_t, 0, 0, false); }
  }
#undef CGG_MSDA_LAUNCH
  CGG_CHECK_LAUNCH("cgg_msda_forward");
  return CGG_OK;
}

static int msda_check(const char* who, const void* value, const void* a, const void* b_, const void* o,
                      int B, int Nv, int H, int D, int L, int Nq, int P, int dtype) {
  CGG_REQUIRE(value && a && b_ && o, CGG_EINVAL, "%s: null pointer", who);
  CGG_REQUIRE(B > 0 && Nv > 0 && H > 0 && D > 0 && L > 0 && Nq > 0 && P > 0, CGG_EINVAL,
              "%s: bad sizes", who);
  CGG_REQUIRE(D % (dtype == CGG_F32 ? 4 : 8) == 0, CGG_EUNSUPPORTED,
              "%s: head dim D=%d must be a multiple of 4 (f32) / 8 (bf16)", who, D);
  CGG_REQUIRE(L <= 8, CGG_EUNSUPPORTED, "%s: L=%d levels (max 8)", who, L);
  CGG_REQUIRE(dtype == CGG_F32 || dtype == CGG_BF16, CGG_EUNSUPPORTED, "%s: dtype %d", who, dtype);
  CGG_REQUIRE(cgg_aligned16(value) && cgg_aligned16(o), CGG_EALIGN, "%s: value/out must be 16-B aligned", who);
  return CGG_OK;
}

extern "C" int cgg_msda_forward(const void* value, const int64_t* spatial_shapes,
                                const int64_t* level_start, const float* sampling_loc,
                                const float* attn_weight, float* out, int B, int Nv, int H, int D,
                                int L, int Nq, int P, int value_dtype, cgg_stream_t stream) {
  int rc = msda_check("cgg_msda_forward", value, sampling_loc, attn_weight, out, B, Nv, H, D, L, Nq, P,
                      value_dtype);
  if (rc) return rc;
  CGG_REQUIRE(spatial_shapes && level_start, CGG_EINVAL, "cgg_msda_forward: null level table");
  MsdaLevels lv;
  hipStream_t s = (hipStream_t)stream;
  rc = msda_read_levels(spatial_shapes, level_start, L, Nv, s, &lv, "cgg_msda_forward");
  if (rc) return rc;
  return msda_fwd_launch(value, lv, sampling_loc, attn_weight, nullptr, 0, out, B, Nv, H, D, L, Nq, P,
                         value_dtype, false, s);
}

extern "C" int cgg_msda_forward_fused(const void* value, const int64_t* spatial_shapes,
                                      const int64_t* level_start, const float* offs_logits, int ld,
                                      const float* ref_points, float* out, int B, int Nv, int H,
                                      int D, int L, int Nq, int P, int value_dtype,
                                      cgg_stream_t stream) {
  int rc = msda_check("cgg_msda_forward_fused", value, offs_logits, ref_points, out, B, Nv, H, D, L, Nq,
                      P, value_dtype);
  if (rc) return rc;
  CGG_REQUIRE(spatial_shapes && level_start, CGG_EINVAL, "cgg_msda_forward_fused: null level table");
  CGG_REQUIRE(ld >= H * L * P * 3, CGG_EINVAL, "cgg_msda_forward_fused: ld=%d < H*L*P*3", ld);
  MsdaLevels lv;
  hipStream_t s = (hipStream_t)stream;
  rc = msda_read_levels(spatial_shapes, level_start, L, Nv, s, &lv, "cgg_msda_forward_fused");
  if (rc) return rc;
  return msda_fwd_launch(value, lv, offs_logits, nullptr, ref_points, ld, out, B, Nv, H, D, L, Nq, P,
                         value_dtype, true, s);
}

extern "C" int cgg_msda_forward_hostlevels(const void* value, const int32_t* level_hw,
                                           const int32_t* level_start, const float* sampling_loc,
                                           const float* attn_weight, const float* ref_points, int ld,
                                           float* out, int B, int Nv, int H, int D, int L, int Nq,
                                           int P, int value_dtype, int fused, cgg_stream_t stream) {
  int rc = msda_check("cgg_msda_forward_hostlevels", value, sampling_loc,
                      fused ? (const void*)ref_points : (const void*)attn_weight, out, B, Nv, H, D, L,
                      Nq, P, value_dtype);
  if (rc) return rc;
  CGG_REQUIRE(level_hw && level_start, CGG_EINVAL, "cgg_msda_forward_hostlevels: null level table");
  if (fused) CGG_REQUIRE(ld >= H * L * P * 3, CGG_EINVAL, "cgg_msda_forward_hostlevels: ld too small");
  MsdaLevels lv;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = level_start[l];
    CGG_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0 && lv.start[l] >= 0 &&
                    (long long)lv.start[l] + (long long)lv.h[l] * lv.w[l] <= Nv,
                CGG_EINVAL, "cgg_msda_forward_hostlevels: level %d does not fit Nv=%d", l, Nv);
  }
  return msda_fwd_launch(value, lv, sampling_loc, attn_weight, ref_points, ld, out, B, Nv, H, D, L, Nq,
                         P, value_dtype, fused != 0, (hipStream_t)stream);
}

// Fused f32 form with the value operand as COLUMNS of wider rows: value[b, n, h, :] = value_rows[(b Nv + n) vld + h D ...] (vld % 32
// == 0, 16-byte aligned base). For the merged projection GEMM of the x3a encoder stream, whose rows are [value | offsets | logits].
extern "C" int cgg_msda_forward_fused_vld(const float* value_rows, int vld, const int32_t* level_hw, const int32_t* level_start,
                                          const float* offs_logits, int ld, const float* ref_points, float* out, int B, int Nv, int H,
                                          int D, int L, int Nq, int P, cgg_stream_t stream) {
  int rc = msda_check("cgg_msda_forward_fused_vld", value_rows, offs_logits, ref_points, out, B, Nv, H, D, L, Nq, P, CGG_F32);
  if (rc) return rc;
  CGG_REQUIRE(level_hw && level_start, CGG_EINVAL, "cgg_msda_forward_fused_vld: null level table");
  CGG_REQUIRE(vld >= H * D && vld % 32 == 0 && ld >= H * L * P * 3, CGG_EINVAL, "cgg_msda_forward_fused_vld: vld=%d ld=%d", vld, ld);
  MsdaLevels lv;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = level_start[l];
    CGG_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0 && lv.start[l] >= 0 && (long long)lv.start[l] + (long long)lv.h[l] * lv.w[l] <= Nv,
                CGG_EINVAL, "cgg_msda_forward_fused_vld: level %d does not fit Nv=%d", l, Nv);
  }
  return msda_fwd_launch(value_rows, lv, offs_logits, nullptr, ref_points, ld, out, B, Nv, H, D, L, Nq, P, CGG_F32, true,
                         (hipStream_t)stream, vld);
}

static int msda_bwd_launch(const float* value, const MsdaLevels& lv, const float* sampling_loc, const float* attn_weight,
                           const float* grad_out, float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D,
                           int L, int Nq, int P, hipStream_t s, bool overwrite = false, hipStream_t side = nullptr, void* ws = nullptr,
                           long long ws_bytes = 0, int vld = 0);

extern "C" int cgg_msda_backward(const float* value, const int64_t* spatial_shapes,
                                 const int64_t* level_start, const float* sampling_loc,
                                 const float* attn_weight, const float* grad_out, float* grad_value,
                                 float* grad_loc, float* grad_attn, int B, int Nv, int H, int D,
                                 int L, int Nq, int P, cgg_stream_t stream) {
  int rc = msda_check("cgg_msda_backward", value, sampling_loc, attn_weight, grad_out, B, Nv, H, D, L,
                      Nq, P, CGG_F32);
  if (rc) return rc;
  CGG_REQUIRE(spatial_shapes && level_start && grad_value && grad_loc && grad_attn, CGG_EINVAL,
              "cgg_msda_backward: null pointer");
  const int DQ = D / 4;
  CGG_REQUIRE((DQ & (DQ - 1)) == 0 && DQ <= 64, CGG_EUNSUPPORTED,
              "cgg_msda_backward: D/4=%d must be a power of two <= 64", DQ);
  MsdaLevels lv;
  hipStream_t s = (hipStream_t)stream;
  rc = msda_read_levels(spatial_shapes, level_start, L, Nv, s, &lv, "cgg_msda_backward");
  if (rc) return rc;
  return msda_bwd_launch(value, lv, sampling_loc, attn_weight, grad_out, grad_value, grad_loc, grad_attn, B, Nv, H, D, L, Nq, P, s);
}

// Split backward: grad_value by the sorted-scatter kernel (msda_bwd.hip) when the pyramid is tileable (the encoder's
// self-attention: queries == pixels of the value pyramid), grad_loc / grad_attn by the gather kernel at full occupancy; any other
// geometry: the generic one-kernel form with global f32 atomics for grad_value.
static int msda_bwd_launch(const float* value, const MsdaLevels& lv, const float* sampling_loc, const float* attn_weight,
                           const float* grad_out, float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D,
                           int L, int Nq, int P, hipStream_t s, bool overwrite, hipStream_t side, void* ws, long long ws_bytes, int vld) {
  const int DQ = D / 4;
  if (vld == H * D) vld = 0;
  // side != null: the gather kernel (grad_loc / grad_attn) runs on `side` next to the sorted-scatter kernel (grad_value) on `s` --
  // they share inputs only; one is bound by the LDS pipe, the other by VALU issue and L1 gathers. Fork / join by events: `side`
  // starts after everything enqueued on `s` so far, `s` continues after the gather.
  hipEvent_t fork = nullptr, join = nullptr;
  if (side && !generic_only && msda_bwd_sorted_ok(lv, B, Nv, H, D, L, Nq, P)) {
    if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(fork, s) != hipSuccess || hipStreamWaitEvent(side, fork, 0) != hipSuccess) {
      if (fork) (void)hipEventDestroy(fork);
      if (join) (void)hipEventDestroy(join);
      fork = join = nullptr;
      side = nullptr;
    }
  } else {
    side = nullptr;
  }
  // (a workspace selects the two-pass sorted scatter -- robust to offsets of several pixels; without one: the single pass)
  int rc = generic_only ? CGG_EUNSUPPORTED
           : ws ? msda_bwd_sorted_launch_two_pass(lv, sampling_loc, attn_weight, grad_out, grad_value, B, Nv, H, D, L, Nq, P, ws, ws_bytes, s,
                                                  vld)
                : msda_bwd_sorted_launch(lv, sampling_loc, attn_weight, grad_out, grad_value, B, Nv, H, D, L, Nq, P, s, vld);
  CGG_REQUIRE(rc == CGG_OK || vld == 0, CGG_EUNSUPPORTED,
              "cgg_msda_backward: padded value rows (vld=%d) need the split backward (tileable pyramid, D == 32, P == 4)", vld);
  if (rc == CGG_OK) {
    hipStream_t s_main = s;
    if (side) s = side;
    const long long total = (long long)B * Nq * H * DQ;
    const int nb = (int)((total + 255) / 256);
    if (P == 4 && cgg_aligned16(sampling_loc) && cgg_aligned16(attn_weight) && cgg_aligned16(grad_loc) && cgg_aligned16(grad_attn)) {
      // 8 x 4 pixel patches of one head per block when the pyramid allows it (see the kernel)
      bool patch = H == 8 && D == 32 && Nq == Nv && Nq % 32 == 0;
      long long expect = 0;
      for (int l = 0; l < L && patch; ++l) {
        patch = lv.start[l] == expect && lv.w[l] % 8 == 0 && lv.h[l] % 4 == 0 && lv.start[l] % 32 == 0;
        expect += (long long)lv.h[l] * lv.w[l];
      }
      patch = patch && expect == Nq;
#define CGG_G4(ACC, PATCH)                                                                                                      \
  hipLaunchKernelGGL((cgg_msda_bwd_gather4_kernel<ACC, PATCH>), dim3(nb), dim3(256), 0, s, value, lv, sampling_loc, attn_weight, \
                     grad_out, grad_loc, grad_attn, Nv, H, D, L, Nq, total, vld)
      if (overwrite) { if (patch) CGG_G4(false, true); else CGG_G4(false, false); }
      else           { if (patch) CGG_G4(true, true); else CGG_G4(true, false); }
#undef CGG_G4
    } else if (vld) {
      cgg_set_error("cgg_msda_backward: padded value rows (vld=%d) need P == 4 and 16-byte aligned operands", vld);
      return CGG_EUNSUPPORTED;
    } else if (P == 4)
      hipLaunchKernelGGL((cgg_msda_bwd_kernel<4, false>), dim3(nb), dim3(256), 0, s, value, lv, sampling_loc, attn_weight, grad_out,
                         grad_value, grad_loc, grad_attn, Nv, H, D, L, Nq, P, total);
    else
      hipLaunchKernelGGL((cgg_msda_bwd_kernel<0, false>), dim3(nb), dim3(256), 0, s, value, lv, sampling_loc, attn_weight, grad_out,
                         grad_value, grad_loc, grad_attn, Nv, H, D, L, Nq, P, total);
    if (side) {
      const bool ok = hipEventRecord(join, side) == hipSuccess && hipStreamWaitEvent(s_main, join, 0) == hipSuccess;
      (void)hipEventDestroy(fork);
      (void)hipEventDestroy(join);
      CGG_REQUIRE(ok, CGG_EINVAL, "cgg_msda_backward: joining the side stream failed");
    }
    CGG_CHECK_LAUNCH("cgg_msda_backward(gather)");
    return CGG_OK;
  }
  if (fork) {                                    // (the sorted kernel refused after all: nothing ran on `side`)
    (void)hipEventDestroy(fork);
    (void)hipEventDestroy(join);
  }
  if (rc != CGG_EUNSUPPORTED) return rc;
  const long long total = (long long)B * Nq * H * DQ;
  const int nblk = (int)((total + 255) / 256);
  if (P == 4)
    hipLaunchKernelGGL(cgg_msda_bwd_kernel<4>, dim3(nblk), dim3(256), 0, s, value, lv, sampling_loc,
                       attn_weight, grad_out, grad_value, grad_loc, grad_attn, Nv, H, D, L, Nq, P, total);
  else
    hipLaunchKernelGGL(cgg_msda_bwd_kernel<0>, dim3(nblk), dim3(256), 0, s, value, lv, sampling_loc,
                       attn_weight, grad_out, grad_value, grad_loc, grad_attn, Nv, H, D, L, Nq, P, total);
  CGG_CHECK_LAUNCH("cgg_msda_backward");
  return CGG_OK;
}

// cgg_msda_backward with the level table from the HOST (level_hw = [h0, w0, h1, w1, ...], level_start): no device->host copy and
// no stream synchronisation per call (the mmcv-contract entry point above reads its int64 device tensors back, once per call --
// six stalls per training step in the encoder's backward), graph-capturable.
static int msda_bwd_hostlevels(const float* value, const int32_t* level_hw, const int32_t* level_start, const float* sampling_loc,
                               const float* attn_weight, const float* grad_out, float* grad_value, float* grad_loc, float* grad_attn,
                               int B, int Nv, int H, int D, int L, int Nq, int P, int overwrite_loc_attn, cgg_stream_t stream,
                               cgg_stream_t side_stream, void* ws = nullptr, long long ws_bytes = 0, int vld = 0) {
  int rc = msda_check("cgg_msda_backward_hostlevels", value, sampling_loc, attn_weight, grad_out, B, Nv, H, D, L, Nq, P, CGG_F32);
  if (rc) return rc;
  CGG_REQUIRE(level_hw && level_start && grad_value && grad_loc && grad_attn, CGG_EINVAL, "cgg_msda_backward_hostlevels: null pointer");
  const int DQ = D / 4;
  CGG_REQUIRE((DQ & (DQ - 1)) == 0 && DQ <= 64, CGG_EUNSUPPORTED, "cgg_msda_backward_hostlevels: D/4=%d must be a power of two <= 64", DQ);
  MsdaLevels lv;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = level_start[l];
    CGG_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0 && lv.start[l] >= 0 && (long long)lv.start[l] + (long long)lv.h[l] * lv.w[l] <= Nv,
                CGG_EINVAL, "cgg_msda_backward_hostlevels: level %d does not fit Nv=%d", l, Nv);
  }
  // overwrite_loc_attn: grad_loc / grad_attn are written, not accumulated (no zero-fill needed) -- honoured when the split
  // backward with the P == 4 gather kernel runs; *overwrite_loc_attn is only a PERMISSION, so callers that pass it must not rely
  // on accumulation and must still zero the two tensors unless cgg_msda_backward_overwrites(...) says the fast path applies
  const bool tileable = msda_bwd_sorted_ok(lv, B, Nv, H, D, L, Nq, P);
  const bool ow = overwrite_loc_attn && tileable && P == 4 && cgg_aligned16(sampling_loc) && cgg_aligned16(attn_weight) &&
                  cgg_aligned16(grad_loc) && cgg_aligned16(grad_attn) && !generic_only;
  CGG_REQUIRE(!overwrite_loc_attn || ow, CGG_EUNSUPPORTED,
              "cgg_msda_backward_hostlevels: overwrite_loc_attn needs the split backward (tileable pyramid, D == 32, P == 4, aligned)");
  CGG_REQUIRE(vld == 0 || (vld >= H * D && vld % 4 == 0), CGG_EINVAL, "cgg_msda_backward_hostlevels: vld=%d", vld);
  return msda_bwd_launch(value, lv, sampling_loc, attn_weight, grad_out, grad_value, grad_loc, grad_attn, B, Nv, H, D, L, Nq, P,
                         (hipStream_t)stream, ow, (hipStream_t)side_stream, ws, ws_bytes, vld);
}

extern "C" int cgg_msda_backward_hostlevels(const float* value, const int32_t* level_hw, const int32_t* level_start,
                                            const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                            float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D, int L,
                                            int Nq, int P, int overwrite_loc_attn, cgg_stream_t stream) {
  return msda_bwd_hostlevels(value, level_hw, level_start, sampling_loc, attn_weight, grad_out, grad_value, grad_loc, grad_attn, B, Nv,
                             H, D, L, Nq, P, overwrite_loc_attn, stream, nullptr);
}

// ... with the two kernels of the split backward on TWO streams: grad_value (sorted scatter) on `stream`, grad_loc / grad_attn
// (gather) on `side_stream`, forked after the work enqueued on `stream` so far and joined back into it before the call returns (no
// host synchronisation; every result is ordered on `stream`). side_stream == stream or null = cgg_msda_backward_hostlevels.
extern "C" int cgg_msda_backward_hostlevels_2s(const float* value, const int32_t* level_hw, const int32_t* level_start,
                                               const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                               float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D, int L,
                                               int Nq, int P, int overwrite_loc_attn, cgg_stream_t stream, cgg_stream_t side_stream) {
  return msda_bwd_hostlevels(value, level_hw, level_start, sampling_loc, attn_weight, grad_out, grad_value, grad_loc, grad_attn, B, Nv,
                             H, D, L, Nq, P, overwrite_loc_attn, stream, side_stream == stream ? nullptr : side_stream);
}

// ... with a workspace for the TWO-PASS sorted scatter of grad_value (csrc/msda_bwd.hip): corners that leave the 4-pixel halo of
// the first pass are re-sorted on larger tiles with a 12-pixel halo instead of costing one 128-byte atomic each. ws_bytes >=
// cgg_msda_backward_workspace_bytes(...) (0 there = the geometry has no two-pass form; ws may then be null = the _2s entry).
extern "C" int cgg_msda_backward_hostlevels_ws(const float* value, int vld, const int32_t* level_hw, const int32_t* level_start,
                                               const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                               float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D, int L,
                                               int Nq, int P, int overwrite_loc_attn, void* ws, long long ws_bytes, cgg_stream_t stream,
                                               cgg_stream_t side_stream) {
  CGG_REQUIRE(!ws || cgg_aligned16(ws), CGG_EALIGN, "cgg_msda_backward_hostlevels_ws: workspace must be 16-B aligned");
  return msda_bwd_hostlevels(value, level_hw, level_start, sampling_loc, attn_weight, grad_out, grad_value, grad_loc, grad_attn, B, Nv,
                             H, D, L, Nq, P, overwrite_loc_attn, stream, side_stream == stream ? nullptr : side_stream, ws, ws_bytes, vld);
}

extern "C" long long cgg_msda_backward_workspace_bytes(const int32_t* level_hw, const int32_t* level_start, int B, int Nv, int H, int D,
                                                       int L, int Nq, int P) {
  if (!level_hw || !level_start || L < 1 || L > 8 || generic_only) return 0;
  MsdaLevels lv;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = level_start[l];
  }
  return msda_bwd_two_pass_workspace_bytes(lv, B, Nv, H, D, L, Nq, P);
}

// 1 when cgg_msda_backward_hostlevels(..., overwrite_loc_attn = 1) is valid for this geometry (pointer alignment aside)
extern "C" int cgg_msda_backward_overwrites(const int32_t* level_hw, const int32_t* level_start, int B, int Nv, int H, int D, int L,
                                            int Nq, int P) {
  if (!level_hw || !level_start || L < 1 || L > 8) return 0;
  MsdaLevels lv;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = level_start[l];
  }
  return (P == 4 && !generic_only && msda_bwd_sorted_ok(lv, B, Nv, H, D, L, Nq, P)) ? 1 : 0;
}

// Throughput-mode encoder stream: bf16 value, bf16 raw [offsets | logits] rows (a bf16 GEMM's output), bf16 output
// (the next GEMM's input). Level table from the host (graph-capturable). L == 3, P == 4 fast path or generic.
static int msda_fused_bf16_impl(bool head_major, const void* value, const int32_t* level_hw, const int32_t* level_start,
                                const void* offs_logits, int ld, const float* ref_points, void* out, int B, int Nv, int H,
                                int D, int L, int Nq, int P, cgg_stream_t stream) {
  int rc = msda_check("cgg_msda_forward_fused_bf16", value, offs_logits, ref_points, out, B, Nv, H, D, L, Nq, P,
                      CGG_BF16);
  if (rc) return rc;
  CGG_REQUIRE(level_hw && level_start, CGG_EINVAL, "cgg_msda_forward_fused_bf16: null level table");
  CGG_REQUIRE(ld >= H * L * P * 3, CGG_EINVAL, "cgg_msda_forward_fused_bf16: ld too small");
  MsdaLevels lv;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = level_start[l];
    CGG_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0 && lv.start[l] >= 0 &&
                    (long long)lv.start[l] + (long long)lv.h[l] * lv.w[l] <= Nv,
                CGG_EINVAL, "cgg_msda_forward_fused_bf16: level %d does not fit Nv=%d", l, Nv);
  }
  const long long total = (long long)B * Nq * H * (D / 8);
  const int nblk = (int)((total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  // CGG_MSDA_GENERIC=1 forces the generic kernel (the fallback of every other shape; tests run it on the stream's shapes too)
  const bool fast_ok = L == 3 && P == 4 && D == 32 && ld % 8 == 0 && cgg_aligned16(offs_logits) && !generic_only;
  const bool quad_ok = fast_ok && H == 8 && (long long)Nv * H * D < (1ll << 31) && total < (1ll << 31);
  CGG_REQUIRE(!head_major || quad_ok, CGG_EUNSUPPORTED,
              "cgg_msda_forward_fused_bf16_hm: head-major values need H=8, D=32, L=3, P=4 (H=%d D=%d L=%d P=%d)", H, D, L, P);
  if (quad_ok && head_major)
    hipLaunchKernelGGL(cgg_msda_fwd_stream2_kernel<true>, dim3(2 * (unsigned)(((long long)B * Nq + 15) / 16)), dim3(256), 0, s, (const uint16_t*)value, lv,
                       (const uint16_t*)offs_logits, ref_points, ld, (uint16_t*)out, Nv, Nq, (unsigned)total);
  else if (quad_ok)
    hipLaunchKernelGGL(cgg_msda_fwd_stream2_kernel<false>, dim3(nblk), dim3(256), 0, s, (const uint16_t*)value, lv,
                       (const uint16_t*)offs_logits, ref_points, ld, (uint16_t*)out, Nv, Nq, (unsigned)total);
  else if (L == 3 && P == 4)
    hipLaunchKernelGGL((cgg_msda_fwd_kernel<uint16_t, 3, 4, true, uint16_t, uint16_t>), dim3(nblk), dim3(256), 0, s,
                       (const uint16_t*)value, lv, (const uint16_t*)offs_logits, (const uint16_t*)nullptr, ref_points,
                       ld, (uint16_t*)out, Nv, H, D, L, Nq, P, total);
  else
    hipLaunchKernelGGL((cgg_msda_fwd_kernel<uint16_t, 0, 0, true, uint16_t, uint16_t>), dim3(nblk), dim3(256), 0, s,
                       (const uint16_t*)value, lv, (const uint16_t*)offs_logits, (const uint16_t*)nullptr, ref_points,
                       ld, (uint16_t*)out, Nv, H, D, L, Nq, P, total);
  CGG_CHECK_LAUNCH("cgg_msda_forward_fused_bf16");
  return CGG_OK;
}

extern "C" int cgg_msda_forward_fused_bf16(const void* value, const int32_t* level_hw, const int32_t* level_start,
                                           const void* offs_logits, int ld, const float* ref_points, void* out,
                                           int B, int Nv, int H, int D, int L, int Nq, int P,
                                           cgg_stream_t stream) {
  return msda_fused_bf16_impl(false, value, level_hw, level_start, offs_logits, ld, ref_points, out, B, Nv, H, D, L, Nq, P, stream);
}

// value HEAD-MAJOR (B, H, Nv, D) -- what cgg_encoder_proj_bf16 writes with value_head_major != 0
extern "C" int cgg_msda_forward_fused_bf16_hm(const void* value, const int32_t* level_hw, const int32_t* level_start,
                                              const void* offs_logits, int ld, const float* ref_points, void* out,
                                              int B, int Nv, int H, int D, int L, int Nq, int P, cgg_stream_t stream) {
  return msda_fused_bf16_impl(true, value, level_hw, level_start, offs_logits, ld, ref_points, out, B, Nv, H, D, L, Nq, P, stream);
}

// -------------------------------------------------------------------------------------------------
// Training: the prologue of MSDeformAttn ([3P] MultiScaleDeformableAttention.forward: `sampling_offsets(query)` ->
// `reference_points + offsets / (W_l, H_l)`, `attention_weights(query).softmax(-1)`) and its backward as TWO elementwise kernels
// around the gather's backward, instead of ~10 autograd-recorded torch passes over (B, Nq, H, L, P[, 2]) tensors per layer:
//   prologue:  rows (B*Nq, ld) f32 = [offsets H*L*P*2 | logits H*L*P]  ->  loc (B*Nq, H, L, P, 2), aw (B*Nq, H, L, P)
//   backward:  (grad_loc, grad_aw, rows)  ->  grad_rows (B*Nq, ld) f32:  d offs = d loc / (W_l, H_l);
//              d logit = aw * (d aw - sum_j aw_j d aw_j), aw recomputed from the logits
// One thread per (row, head); L * P <= 16.
__global__ __launch_bounds__(256) void cgg_msda_prologue_kernel(const float* __restrict__ rows, int ld, const float* __restrict__ ref,
                                                               MsdaLevels lv, float* __restrict__ loc, float* __restrict__ aw,
                                                               long long total, int Nq, int H, int L, int P) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int h = (int)(gid % H);
  const long long bq = gid / H;
  const int q = (int)(bq % Nq);
  const int LP = L * P;
  const float* ro = rows + (size_t)bq * ld + (size_t)h * LP * 2;
  const float* rl = rows + (size_t)bq * ld + (size_t)H * LP * 2 + (size_t)h * LP;
  float* lo = loc + ((size_t)bq * H + h) * LP * 2;
  float* ao = aw + ((size_t)bq * H + h) * LP;
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  float e[16];
  float m = -3.4e38f;
  for (int i = 0; i < LP; ++i) {
    e[i] = rl[i];
    m = fmaxf(m, e[i]);
  }
  float sum = 0.f;
  for (int i = 0; i < LP; ++i) {
    e[i] = __expf(e[i] - m);
    sum += e[i];
  }
  const float inv = 1.f / sum;
  for (int l = 0; l < L; ++l) {
    const float iw = 1.f / (float)lv.w[l], ih = 1.f / (float)lv.h[l];
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      lo[2 * i] = rx + ro[2 * i] * iw;
      lo[2 * i + 1] = ry + ro[2 * i + 1] * ih;
      ao[i] = e[i] * inv;
    }
  }
}

__global__ __launch_bounds__(256) void cgg_msda_prologue_bwd_kernel(const float* __restrict__ gloc, const float* __restrict__ gaw,
                                                                   const float* __restrict__ rows, int ld, MsdaLevels lv,
                                                                   float* __restrict__ grows, long long total, int H, int L, int P) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int h = (int)(gid % H);
  const long long bq = gid / H;
  const int LP = L * P;
  const float* rl = rows + (size_t)bq * ld + (size_t)H * LP * 2 + (size_t)h * LP;
  const float* gl = gloc + ((size_t)bq * H + h) * LP * 2;
  const float* ga = gaw + ((size_t)bq * H + h) * LP;
  float* go = grows + (size_t)bq * ld + (size_t)h * LP * 2;
  float* gg = grows + (size_t)bq * ld + (size_t)H * LP * 2 + (size_t)h * LP;
  float e[16];
  float m = -3.4e38f;
  for (int i = 0; i < LP; ++i) {
    e[i] = rl[i];
    m = fmaxf(m, e[i]);
  }
  float sum = 0.f;
  for (int i = 0; i < LP; ++i) {
    e[i] = __expf(e[i] - m);
    sum += e[i];
  }
  const float inv = 1.f / sum;
  float dot = 0.f;
  for (int i = 0; i < LP; ++i) {
    e[i] *= inv;
    dot += e[i] * ga[i];
  }
  for (int l = 0; l < L; ++l) {
    const float iw = 1.f / (float)lv.w[l], ih = 1.f / (float)lv.h[l];
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      go[2 * i] = gl[2 * i] * iw;
      go[2 * i + 1] = gl[2 * i + 1] * ih;
      gg[i] = e[i] * (ga[i] - dot);
    }
  }
}

// L = 3, P = 4 (12 points per head): the same two kernels with 16-byte accesses -- a thread's 24 offsets and 12 logits are 6 + 3
// aligned float4 (the scalar versions issue 36 four-byte loads at a 96-byte lane stride and run at a quarter of the bandwidth)
__global__ __launch_bounds__(256) void cgg_msda_prologue12_kernel(const float* __restrict__ rows, int ld, const float* __restrict__ ref,
                                                                 MsdaLevels lv, float* __restrict__ loc, float* __restrict__ aw,
                                                                 long long total, int Nq, int H) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int h = (int)(gid % H);
  const long long bq = gid / H;
  const int q = (int)(bq % Nq);
  const f32x4* ro = reinterpret_cast<const f32x4*>(rows + (size_t)bq * ld + (size_t)h * 24);
  const f32x4* rl = reinterpret_cast<const f32x4*>(rows + (size_t)bq * ld + (size_t)H * 24 + (size_t)h * 12);
  f32x4* lo = reinterpret_cast<f32x4*>(loc + ((size_t)bq * H + h) * 24);
  f32x4* ao = reinterpret_cast<f32x4*>(aw + ((size_t)bq * H + h) * 12);
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  f32x4 e[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) e[i] = rl[i];
  float m = e[0][0];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) m = fmaxf(m, e[i][c]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      e[i][c] = __expf(e[i][c] - m);
      sum += e[i][c];
    }
  const float inv = 1.f / sum;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float iw = 1.f / (float)lv.w[l], ih = 1.f / (float)lv.h[l];
    ao[l] = e[l] * inv;
#pragma unroll
    for (int k = 0; k < 2; ++k) {                        // two float4 = 4 points' (x, y) ... of level l: (x0 y0 x1 y1), (x2 y2 x3 y3)
      const f32x4 o = ro[2 * l + k];
      lo[2 * l + k] = f32x4{rx + o[0] * iw, ry + o[1] * ih, rx + o[2] * iw, ry + o[3] * ih};
    }
  }
}

__global__ __launch_bounds__(256) void cgg_msda_prologue12_bwd_kernel(const float* __restrict__ gloc, const float* __restrict__ gaw,
                                                                     const float* __restrict__ rows, int ld, MsdaLevels lv,
                                                                     float* __restrict__ grows, long long total, int H) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int h = (int)(gid % H);
  const long long bq = gid / H;
  const f32x4* rl = reinterpret_cast<const f32x4*>(rows + (size_t)bq * ld + (size_t)H * 24 + (size_t)h * 12);
  const f32x4* gl = reinterpret_cast<const f32x4*>(gloc + ((size_t)bq * H + h) * 24);
  const f32x4* ga = reinterpret_cast<const f32x4*>(gaw + ((size_t)bq * H + h) * 12);
  f32x4* go = reinterpret_cast<f32x4*>(grows + (size_t)bq * ld + (size_t)h * 24);
  f32x4* gg = reinterpret_cast<f32x4*>(grows + (size_t)bq * ld + (size_t)H * 24 + (size_t)h * 12);
  f32x4 e[3], g[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    e[i] = rl[i];
    g[i] = ga[i];
  }
  float m = e[0][0];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) m = fmaxf(m, e[i][c]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      e[i][c] = __expf(e[i][c] - m);
      sum += e[i][c];
    }
  const float inv = 1.f / sum;
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      e[i][c] *= inv;
      dot += e[i][c] * g[i][c];
    }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float iw = 1.f / (float)lv.w[l], ih = 1.f / (float)lv.h[l];
    gg[l] = e[l] * (g[l] - dot);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x4 o = gl[2 * l + k];
      go[2 * l + k] = f32x4{o[0] * iw, o[1] * ih, o[2] * iw, o[3] * ih};
    }
  }
}

static int msda_prologue_levels(MsdaLevels& lv, const int32_t* level_hw, int L, const char* who) {
  CGG_REQUIRE(level_hw && L >= 1 && L <= 8, CGG_EINVAL, "%s: bad level table (L=%d)", who, L);
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = 0;
    CGG_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0, CGG_EINVAL, "%s: level %d is empty", who, l);
  }
  return CGG_OK;
}

extern "C" int cgg_msda_prologue(const float* rows, int ld, const float* ref_points, const int32_t* level_hw, float* loc,
                                 float* attn, int B, int Nq, int H, int L, int P, cgg_stream_t stream) {
  CGG_REQUIRE(rows && ref_points && loc && attn, CGG_EINVAL, "cgg_msda_prologue: null pointer");
  CGG_REQUIRE(B > 0 && Nq > 0 && H > 0 && P > 0 && L * P <= 16 && ld >= H * L * P * 3, CGG_EUNSUPPORTED,
              "cgg_msda_prologue: H=%d L=%d P=%d ld=%d (L * P <= 16, ld >= 3 H L P)", H, L, P, ld);
  MsdaLevels lv;
  int rc = msda_prologue_levels(lv, level_hw, L, "cgg_msda_prologue");
  if (rc) return rc;
  const long long total = (long long)B * Nq * H;
  if (L == 3 && P == 4 && ld % 4 == 0 && cgg_aligned16(rows) && cgg_aligned16(loc) && cgg_aligned16(attn))
    hipLaunchKernelGGL(cgg_msda_prologue12_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, ld,
                       ref_points, lv, loc, attn, total, Nq, H);
  else
    hipLaunchKernelGGL(cgg_msda_prologue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, ld,
                       ref_points, lv, loc, attn, total, Nq, H, L, P);
  CGG_CHECK_LAUNCH("cgg_msda_prologue");
  return CGG_OK;
}

extern "C" int cgg_msda_prologue_backward(const float* grad_loc, const float* grad_attn, const float* rows, int ld,
                                          const int32_t* level_hw, float* grad_rows, int B, int Nq, int H, int L, int P,
                                          cgg_stream_t stream) {
  CGG_REQUIRE(grad_loc && grad_attn && rows && grad_rows, CGG_EINVAL, "cgg_msda_prologue_backward: null pointer");
  CGG_REQUIRE(B > 0 && Nq > 0 && H > 0 && P > 0 && L * P <= 16 && ld == H * L * P * 3, CGG_EUNSUPPORTED,
              "cgg_msda_prologue_backward: H=%d L=%d P=%d ld=%d (L * P <= 16, ld == 3 H L P: every column gets a gradient)", H, L,
              P, ld);
  MsdaLevels lv;
  int rc = msda_prologue_levels(lv, level_hw, L, "cgg_msda_prologue_backward");
  if (rc) return rc;
  const long long total = (long long)B * Nq * H;
  if (L == 3 && P == 4 && cgg_aligned16(rows) && cgg_aligned16(grad_loc) && cgg_aligned16(grad_attn) && cgg_aligned16(grad_rows))
    hipLaunchKernelGGL(cgg_msda_prologue12_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_loc, grad_attn, rows, ld, lv, grad_rows, total, H);
  else
    hipLaunchKernelGGL(cgg_msda_prologue_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_loc,
                       grad_attn, rows, ld, lv, grad_rows, total, H, L, P);
  CGG_CHECK_LAUNCH("cgg_msda_prologue_backward");
  return CGG_OK;
}
