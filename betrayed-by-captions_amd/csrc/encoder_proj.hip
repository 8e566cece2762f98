// Input projections of one MSDeformAttn encoder layer ([3P] MultiScaleDeformableAttention.forward: `value_proj(value)`,
// `sampling_offsets(query)`, `attention_weights(query)`; the layers are built at open_set/models/mask2former_head.py:112-117)
// as ONE launch over the (B * N) x 256 bf16 stream:
//
//     value = x  Wv^T + bv            (M x 256, bf16)         x  = the layer input rows
//     offs  = xp Wc^T + bc            (M x NC, bf16)          xp = x + pos rows (given, or formed here from a bf16 pos
//                                                             table: xp16 == NULL),  Wc = [W_offsets; W_attention_weights],
//                                                             NC = 3 * heads * levels * points: 288 (3 levels) or 384 (4)
//
// Both are K = 256 GEMMs whose cost is reading and writing the rows (99 MB per layer at configs[1]); the two library
// GEMMs ran at ~2 TB/s (22 + 28 us). Here a workgroup (4 wavefronts) owns 64 rows: x and xp are staged once as MFMA
// A-fragment images in LDS, every wave computes a 64-row x 64-column block of `value` and a 64 x 64 or 64 x 96 block of `offs`
// (NC / 32 = 9 .. 12 n-tiles dealt 2 or 3 per wave, the larger blocks last) as 2 x 2 / 2 x 3 MFMA tiles with the packed weights streamed L2 -> registers (4 k-steps ahead, pinned by scheduling
// barriers as in encoder_ffn.hip), and writes bf16 straight from the accumulators. The weights are packed by
// cgg_encoder_proj_pack with the output columns of a wave's NT n-tiles interleaved in pairs (tile t, lane column j <-> output
// column 2 NT (j / 2) + 2 t + (j & 1) of the wave's block): lanes j, j ^ 1 swap one value per register pair, after which a lane
// holds 2 NT ADJACENT output columns of one row and stores them as one 8- / 12-byte word -- the 16 lanes of a row write one
// contiguous 128- / 192-byte run. (Plain column order gave 4-byte stores in 64-byte runs and cost 10 of 34 us.)
// v_mfma_f32_32x32x16_bf16, f32 accumulation, bias added in f32.
//
// build-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t ep_u32x4;
typedef __attribute__((ext_vector_type(2))) float ep_f2;
typedef __attribute__((ext_vector_type(2))) __bf16 ep_bf2;

#define EP_C 256
#define EP_STEPS 16
#define EP_RB 64
#define EP_PF 4

__device__ __forceinline__ uint32_t ep_pk(float a, float b) {            // v_cvt_pk_bf16_f32
  const ep_f2 pr = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, ep_bf2));
}

// acc[2][NT] += A (two 32-row m-tiles, fragment images in LDS at a0 / a1) x B (NT consecutive n-tiles of the packed weight,
// b = lane pointer to k-step 0 of the first; n-tiles are EP_STEPS * 64 fragments apart)
// TR = true: the operand roles are swapped (weight fragment as A, row image as B): acc[mt][t] is the TRANSPOSED block, rows =
// the 32 output channels of n-tile t, columns = the 32 rows of m-tile mt (fragment layouts of A and B are the same).
template <int NT, bool TR = false>
__device__ __forceinline__ void ep_block(f32x16 (&acc)[2][NT], const ep_u32x4* __restrict__ a0, const ep_u32x4* __restrict__ a1,
                                         int lane, const ep_u32x4* __restrict__ b) {
  ep_u32x4 q[NT][EP_PF];
#pragma unroll
  for (int s = 0; s < EP_PF; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) q[t][s] = b[(t * EP_STEPS + s) * 64];
  ep_u32x4 ua0 = a0[lane], ua1 = a1[lane];                      // k-step 0: lane ^ 0
#pragma unroll
  for (int s = 0; s < EP_STEPS; ++s) {
    bf16x8 vb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) vb[t] = __builtin_bit_cast(bf16x8, q[t][s % EP_PF]);
    const bf16x8 va0 = __builtin_bit_cast(bf16x8, ua0), va1 = __builtin_bit_cast(bf16x8, ua1);
    if (s + 1 < EP_STEPS) {
      ua0 = a0[(s + 1) * 64 + (lane ^ (s + 1))];               // bank swizzle of the image, see the staging loop
      ua1 = a1[(s + 1) * 64 + (lane ^ (s + 1))];
    }
    if (s + EP_PF < EP_STEPS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) q[t][s % EP_PF] = b[(t * EP_STEPS + s + EP_PF) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (TR) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[t], va0, acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[t], va1, acc[1][t], 0, 0, 0);
      } else {
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, vb[t], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, vb[t], acc[1][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct ep_u32x3 { uint32_t a, b, c; };

// output column (within the wave's 32 NT-column block) that tile t, lane column j computes
__host__ __device__ __forceinline__ int ep_col(int NT, int t, int j) { return 2 * NT * (j >> 1) + 2 * t + (j & 1); }

// bf16(acc + bias) -> out rows m0 .. m0 + 63, the wave's columns col0 .. col0 + 32 NT - 1 (ldo = row stride in elements)
template <int NT>
__device__ __forceinline__ void ep_store(const f32x16 (&acc)[2][NT], const float* __restrict__ bias, uint16_t* __restrict__ out,
                                         int ldo, int col0, int m0, int M, int lane) {
  const int j = lane & 31, hi5 = lane >> 5, odd = j & 1;
  const uint32_t rot = 16u * (uint32_t)odd;
  float bs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bs[t] = bias[col0 + ep_col(NT, t, j)];
  uint16_t* ocol = out + col0 + 2 * NT * (j >> 1);                 // both lanes of a pair: the same 2 NT columns, rows r / r + 1
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
      uint32_t pk[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float v0 = acc[mt][t][2 * rp] + bs[t], v1 = acc[mt][t][2 * rp + 1] + bs[t];
        const float kept = odd ? v1 : v0, sent = odd ? v0 : v1;
        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sent), 0xB1, 0xf, 0xf, true));  // lane ^ 1
        // even lane: row(2 rp), columns (even, odd) = (kept, recv); odd lane: row(2 rp + 1), (recv, kept)
        const uint32_t w = ep_pk(kept, recv);
        pk[t] = __builtin_amdgcn_alignbit(w, w, rot);
      }
      const int row = 32 * mt + 2 * (rp & 1) + 8 * (rp >> 1) + 4 * hi5 + odd;
      if (m0 + row < M) {
        uint16_t* o = ocol + (size_t)(m0 + row) * ldo;
        if constexpr (NT == 2) *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
        else *reinterpret_cast<ep_u32x3*>(o) = ep_u32x3{pk[0], pk[1], pk[2]};
      }
    }
  }
}

// `value` block (NT = 2, the wave's 64 columns = heads 2 w, 2 w + 1) stored HEAD-MAJOR: value[(b * 8 + head) * N + n][32]
// for row m = b * N + n -- the layout cgg_msda_forward_fused_bf16_hm gathers from (see msda.hip)
__device__ __forceinline__ void ep_store_value_hm(const f32x16 (&acc)[2][2], const float* __restrict__ bias,
                                                  uint16_t* __restrict__ out, int col0, int m0, int M, int N, int lane) {
  const int j = lane & 31, hi5 = lane >> 5, odd = j & 1;
  const uint32_t rot = 16u * (uint32_t)odd;
  float bs[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) bs[t] = bias[col0 + ep_col(2, t, j)];
  const int col = col0 + 4 * (j >> 1);                              // first of this lane's 4 consecutive columns
  const int head = col >> 5, ch = col & 31;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
      uint32_t pk[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v0 = acc[mt][t][2 * rp] + bs[t], v1 = acc[mt][t][2 * rp + 1] + bs[t];
        const float kept = odd ? v1 : v0, sent = odd ? v0 : v1;
        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sent), 0xB1, 0xf, 0xf, true));  // lane ^ 1
        const uint32_t w = ep_pk(kept, recv);
        pk[t] = __builtin_amdgcn_alignbit(w, w, rot);
      }
      const int m = m0 + 32 * mt + 2 * (rp & 1) + 8 * (rp >> 1) + 4 * hi5 + odd;
      if (m < M) {
        const int b = m / N, n = m - b * N;
        *reinterpret_cast<uint2*>(out + (((size_t)b * 8 + head) * N + n) * 32 + ch) = make_uint2(pk[0], pk[1]);
      }
    }
  }
}

// N / 32 n-tiles dealt to the 4 waves: `base` each, the last `extra` waves one more (N = 256: 2 2 2 2; 288: 2 2 2 3; 384: 3 3 3 3)
__host__ __device__ __forceinline__ int ep_wave_nt(int ntiles, int w) { return ntiles / 4 + (w >= 4 - ntiles % 4 ? 1 : 0); }
__host__ __device__ __forceinline__ int ep_wave_tile0(int ntiles, int w) {
  const int first_big = 4 - ntiles % 4;
  return w * (ntiles / 4) + (w > first_big ? w - first_big : 0);
}

// weight (N x 256) f32 -> bf16 MFMA-B fragments with the column interleave above inside every wave's block
__global__ __launch_bounds__(256) void cgg_encoder_proj_pack_kernel(const float* __restrict__ w, ep_u32x4* __restrict__ out, int ntiles,
                                                                   int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63, t_all = i >> 6;
  const int ks = t_all % EP_STEPS, tile = t_all / EP_STEPS;
  int wave = 3;
  while (wave > 0 && ep_wave_tile0(ntiles, wave) > tile) --wave;
  const int NT = ep_wave_nt(ntiles, wave), tile0 = ep_wave_tile0(ntiles, wave);
  const int n = 32 * tile0 + ep_col(NT, tile - tile0, lane & 31);
  const float* s = w + (size_t)n * EP_C + ks * 16 + 8 * (lane >> 5);
  out[i] = ep_u32x4{ep_pk(s[0], s[1]), ep_pk(s[2], s[3]), ep_pk(s[4], s[5]), ep_pk(s[6], s[7])};
}

__global__ __launch_bounds__(256) void cgg_encoder_proj_kernel(
    const uint16_t* __restrict__ x16, const uint16_t* __restrict__ xp16, const ep_u32x4* __restrict__ wv,
    const float* __restrict__ bv, const ep_u32x4* __restrict__ wc, const float* __restrict__ bc, uint16_t* __restrict__ value,
    uint16_t* __restrict__ offs, int M, int NC, const uint16_t* __restrict__ pos16, int pos_rows, int hm_rows) {
  __shared__ __attribute__((aligned(16))) ep_u32x4 frag[2][2 * EP_STEPS * 64];      // x and xp images: 2 x 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * EP_RB;
  // rows -> A-fragment images: 16-byte piece (row, k8) = 8 channels -> slot (row / 32, k-step = k8 / 2, (row % 32 + 32 (k8 & 1)) ^
  // k-step). The XOR spreads the 32 pieces of a row (one coalesced 512-byte read) over all 16 four-bank groups; unswizzled
  // they share one group and every 128-bit LDS store is a 32-way bank conflict. All 16 loads of a thread are in flight at once.
  // (Requesting the xp rows late, to overlap them with the value block, does not help: vmcnt retires in order, so the first
  // wait on a weight fragment also waits for the older row loads.)
  {
    ep_u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = tid + 256 * i, which = p >> 11, row = (p >> 5) & 63, k8 = p & 31;
      v[i] = ep_u32x4{0u, 0u, 0u, 0u};
      if (m0 + row < M) {
        if (which && xp16 == nullptr) {
          // xp = bf16(x + pos) from the bf16 rows and the bf16 pos table: the (M, 256) `x + pos` rows are never stored
          const ep_u32x4 a = *reinterpret_cast<const ep_u32x4*>(x16 + (size_t)(m0 + row) * EP_C + 8 * k8);
          const ep_u32x4 b = *reinterpret_cast<const ep_u32x4*>(pos16 + (size_t)((m0 + row) % pos_rows) * EP_C + 8 * k8);
#pragma unroll
          for (int d = 0; d < 4; ++d)
            v[i][d] = ep_pk(__uint_as_float(a[d] << 16) + __uint_as_float(b[d] << 16),
                            __uint_as_float(a[d] & 0xffff0000u) + __uint_as_float(b[d] & 0xffff0000u));
        } else {
          v[i] = *reinterpret_cast<const ep_u32x4*>((which ? xp16 : x16) + (size_t)(m0 + row) * EP_C + 8 * k8);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = tid + 256 * i, which = p >> 11, row = (p >> 5) & 63, k8 = p & 31;
      frag[which][((row >> 5) * EP_STEPS + (k8 >> 1)) * 64 + (((row & 31) + 32 * (k8 & 1)) ^ (k8 >> 1))] = v[i];
    }
  }
  __syncthreads();
  {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    ep_block<2>(acc, frag[0], frag[0] + EP_STEPS * 64, lane, wv + (size_t)(2 * wave) * EP_STEPS * 64 + lane);
    if (hm_rows > 0) ep_store_value_hm(acc, bv, value, 64 * wave, m0, M, hm_rows, lane);     // block-uniform
    else ep_store<2>(acc, bv, value, 256, 64 * wave, m0, M, lane);
  }
  const int ntiles = NC >> 5, tile0 = ep_wave_tile0(ntiles, wave);
  if (ep_wave_nt(ntiles, wave) == 3) {                 // wave-uniform
    f32x16 acc[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    ep_block<3>(acc, frag[1], frag[1] + EP_STEPS * 64, lane, wc + (size_t)tile0 * EP_STEPS * 64 + lane);
    ep_store<3>(acc, bc, offs, NC, 32 * tile0, m0, M, lane);
  } else {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    ep_block<2>(acc, frag[1], frag[1] + EP_STEPS * 64, lane, wc + (size_t)tile0 * EP_STEPS * 64 + lane);
    ep_store<2>(acc, bc, offs, NC, 32 * tile0, m0, M, lane);
  }
}

extern "C" int cgg_encoder_proj_pack(const float* w, void* packed, int N, int K, cgg_stream_t stream) {
  CGG_REQUIRE(w && packed, CGG_EINVAL, "cgg_encoder_proj_pack: null pointer");
  CGG_REQUIRE(K == EP_C && N % 32 == 0 && N >= 256 && N <= 384, CGG_EUNSUPPORTED,
              "cgg_encoder_proj_pack: N=%d K=%d (N = 256 .. 384 in steps of 32, K = 256)", N, K);
  const int total = (N / 32) * EP_STEPS * 64;
  hipLaunchKernelGGL(cgg_encoder_proj_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                     (ep_u32x4*)packed, N / 32, total);
  CGG_CHECK_LAUNCH("cgg_encoder_proj_pack");
  return CGG_OK;
}

extern "C" int cgg_encoder_proj_bf16(const void* x16, const void* xp16, const void* pos16, int pos_rows, const void* wv_packed,
                                     const float* bv, const void* wc_packed, const float* bc, void* value, void* offs, int M,
                                     int C, int NV, int NC, int value_head_major_rows, cgg_stream_t stream) {
  CGG_REQUIRE(value_head_major_rows >= 0 && (value_head_major_rows == 0 || M % value_head_major_rows == 0), CGG_EINVAL,
              "cgg_encoder_proj_bf16: value_head_major_rows=%d must divide M=%d", value_head_major_rows, M);
  CGG_REQUIRE(x16 && wv_packed && bv && wc_packed && bc && value && offs, CGG_EINVAL, "cgg_encoder_proj_bf16: null pointer");
  CGG_REQUIRE(xp16 || (pos16 && pos_rows > 0), CGG_EINVAL, "cgg_encoder_proj_bf16: either xp16 or a pos16 table is required");
  CGG_REQUIRE(C == EP_C && NV == 256 && NC % 32 == 0 && NC >= 256 && NC <= 384, CGG_EUNSUPPORTED,
              "cgg_encoder_proj_bf16: C=%d NV=%d NC=%d (built for 256 -> 256 + 256 .. 384 in steps of 32)", C, NV, NC);
  CGG_REQUIRE(M > 0, CGG_EINVAL, "cgg_encoder_proj_bf16: M=%d", M);
  CGG_REQUIRE(cgg_aligned16(x16) && (!xp16 || cgg_aligned16(xp16)) && (!pos16 || cgg_aligned16(pos16)) && cgg_aligned16(wv_packed) && cgg_aligned16(wc_packed) &&
                  cgg_aligned16(value) && cgg_aligned16(offs),
              CGG_EALIGN, "cgg_encoder_proj_bf16: 16-B alignment");
  hipLaunchKernelGGL(cgg_encoder_proj_kernel, dim3((M + EP_RB - 1) / EP_RB), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x16, (const uint16_t*)xp16, (const ep_u32x4*)wv_packed, bv, (const ep_u32x4*)wc_packed, bc,
                     (uint16_t*)value, (uint16_t*)offs, M, NC, (const uint16_t*)pos16, pos_rows, value_head_major_rows);
  CGG_CHECK_LAUNCH("cgg_encoder_proj_bf16");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// K / V projections of the query decoder for ONE memory level (open_set/models/mask2former_head.py:795-812 + the
// [3P] nn.MultiheadAttention in_proj of the cross-attention): the decoder layers that read this level (l, l + 3, l + 6) share
// their input, so their projections are stacked into NK = NVT = n * 256 outputs:
//     k  (M, NK)           = mp16 Wk^T + bk            rows = (batch, pixel), what the attention kernel reads key-major
//     vt (B, NVT, hw)      = Wv m16^T                  the value projection TRANSPOSED (channel-major), no bias (added after P V)
// One launch instead of a library GEMM pair per level (6 per forward, ~2 TB/s on these K = 256 shapes). Same structure as the
// kernel above: 64-row blocks, both operand images in LDS, weights streamed L2 -> registers, 2 x 2 MFMA tiles. The k block
// uses the column-interleaved packing (8-byte stores, 128-byte runs). The vt block swaps the MFMA operand roles, which yields
// the transposed tile directly; the m16 image is staged with its ROWS interleaved between the two m-tiles (image row (mt, j)
// = block row 4 (j / 2) + 2 mt + (j & 1)), so that after the same lane-pair swap a lane holds 4 consecutive pixels of one
// channel: 8-byte stores, 128-byte runs along the pixel axis.
// RAGGED: only pixels < `valid` of the block exist and a channel row (ldc = hw elements) is 4-byte aligned at best (hw even: 4-byte
// stores of pixel pairs; hw odd: 2-byte stores)
template <int NT, bool RAGGED = false>
__device__ __forceinline__ void ep_store_tr(const f32x16 (&acc)[2][NT], uint16_t* __restrict__ vt, int tile0, long long ldc,
                                            int pix0, int lane, int valid = 64) {
  const int j = lane & 31, hi5 = lane >> 5, odd = j & 1;
  const uint32_t rot = 16u * (uint32_t)odd;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
      uint32_t pk[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float v0 = acc[mt][t][2 * rp], v1 = acc[mt][t][2 * rp + 1];
        const float kept = odd ? v1 : v0, sent = odd ? v0 : v1;
        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sent), 0xB1, 0xf, 0xf, true));  // lane ^ 1
        const uint32_t w = ep_pk(kept, recv);          // even lane: channel row(2 rp), pixels (j, j + 1) of m-tile mt; odd: row + 1
        pk[mt] = __builtin_amdgcn_alignbit(w, w, rot);
      }
      const int ch = 32 * (tile0 + t) + 2 * (rp & 1) + 8 * (rp >> 1) + 4 * hi5 + odd;
      uint16_t* o = vt + (size_t)ch * ldc + pix0 + 4 * (j >> 1);
      if constexpr (!RAGGED) {
        *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
      } else {
        const int p = 4 * (j >> 1);
        if (!(ldc & 1)) {                              // uniform: valid is even too (hw and the block origin are)
          if (p < valid) *reinterpret_cast<uint32_t*>(o) = pk[0];
          if (p + 2 < valid) *reinterpret_cast<uint32_t*>(o + 2) = pk[1];
        } else {
          if (p < valid) o[0] = (uint16_t)pk[0];
          if (p + 1 < valid) o[1] = (uint16_t)(pk[0] >> 16);
          if (p + 2 < valid) o[2] = (uint16_t)pk[1];
          if (p + 3 < valid) o[3] = (uint16_t)(pk[1] >> 16);
        }
      }
    }
  }
}

// RAGGED (hw % 64 != 0: 1050 / 4200 / 16800 keys of the 800 x 1333 geometry): blocks are cut PER IMAGE (ceil(hw / 64) each), the last
// block of an image re-reads its last row for the missing ones (finite operands) and masks their stores
template <bool RAGGED>
__global__ __launch_bounds__(256) void cgg_decoder_kv_proj_kernel(
    const uint16_t* __restrict__ m16, const uint16_t* __restrict__ mp16, const ep_u32x4* __restrict__ wk,
    const float* __restrict__ bk, const ep_u32x4* __restrict__ wv, uint16_t* __restrict__ k, uint16_t* __restrict__ vt, int hw,
    int NK, int blocks_per_image) {
  __shared__ __attribute__((aligned(16))) ep_u32x4 frag[2][2 * EP_STEPS * 64];      // m16 (row-interleaved) and mp16 images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.x / blocks_per_image, pix0 = (blockIdx.x - img * blocks_per_image) * EP_RB;
  const int m0 = img * hw + pix0;                        // a block never straddles two images
  const int valid = RAGGED ? min(EP_RB, hw - pix0) : EP_RB;
  {
    ep_u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = tid + 256 * i, which = p >> 11, row = (p >> 5) & 63, k8 = p & 31;
      v[i] = *reinterpret_cast<const ep_u32x4*>((which ? mp16 : m16) + (size_t)(m0 + (RAGGED ? min(row, valid - 1) : row)) * EP_C + 8 * k8);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = tid + 256 * i, which = p >> 11, row = (p >> 5) & 63, k8 = p & 31;
      // image row of block row `row`: the m16 image interleaves the two m-tiles in pixel pairs, the mp16 image is in order
      const int ir = which ? row : (((row >> 1) & 1) * 32 + 2 * (row >> 2) + (row & 1));
      frag[which][((ir >> 5) * EP_STEPS + (k8 >> 1)) * 64 + (((ir & 31) + 32 * (k8 & 1)) ^ (k8 >> 1))] = v[i];
    }
  }
  __syncthreads();
  const int npass = NK >> 8;                             // 64 columns / channels per wave and pass
  for (int ps = 0; ps < npass; ++ps) {
    const int tile0 = 2 * (4 * ps + wave);
    {
      f32x16 acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      ep_block<2>(acc, frag[1], frag[1] + EP_STEPS * 64, lane, wk + (size_t)tile0 * EP_STEPS * 64 + lane);
      ep_store<2>(acc, bk, k, NK, 32 * tile0, m0, m0 + valid, lane);
    }
    {
      f32x16 acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      ep_block<2, true>(acc, frag[0], frag[0] + EP_STEPS * 64, lane, wv + (size_t)tile0 * EP_STEPS * 64 + lane);
      ep_store_tr<2, RAGGED>(acc, vt + (size_t)img * NK * hw, tile0, hw, pix0, lane, valid);
    }
  }
}

// weight (N x 256) f32, N % 64 == 0 -> bf16 B fragments for the k block: 64-column groups, the two n-tiles of a group interleaved
__global__ __launch_bounds__(256) void cgg_decoder_kv_pack_k_kernel(const float* __restrict__ w, ep_u32x4* __restrict__ out, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63, t_all = i >> 6;
  const int ks = t_all % EP_STEPS, tile = t_all / EP_STEPS;
  const int n = 64 * (tile >> 1) + ep_col(2, tile & 1, lane & 31);
  const float* s = w + (size_t)n * EP_C + ks * 16 + 8 * (lane >> 5);
  out[i] = ep_u32x4{ep_pk(s[0], s[1]), ep_pk(s[2], s[3]), ep_pk(s[4], s[5]), ep_pk(s[6], s[7])};
}

extern "C" int cgg_decoder_kv_pack_k(const float* w, void* packed, int N, int K, cgg_stream_t stream) {
  CGG_REQUIRE(w && packed, CGG_EINVAL, "cgg_decoder_kv_pack_k: null pointer");
  CGG_REQUIRE(K == EP_C && N > 0 && N % 256 == 0, CGG_EUNSUPPORTED, "cgg_decoder_kv_pack_k: N=%d K=%d (N %% 256, K = 256)", N, K);
  const int total = (N / 32) * EP_STEPS * 64;
  hipLaunchKernelGGL(cgg_decoder_kv_pack_k_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                     (ep_u32x4*)packed, total);
  CGG_CHECK_LAUNCH("cgg_decoder_kv_pack_k");
  return CGG_OK;
}

extern "C" int cgg_decoder_kv_proj_bf16(const void* m16, const void* mp16, const void* wk_packed, const float* bk,
                                        const void* wv_packed, void* k, void* vt, int B, int hw, int C, int NK,
                                        cgg_stream_t stream) {
  CGG_REQUIRE(m16 && mp16 && wk_packed && bk && wv_packed && k && vt, CGG_EINVAL, "cgg_decoder_kv_proj_bf16: null pointer");
  CGG_REQUIRE(C == EP_C && NK > 0 && NK % 256 == 0, CGG_EUNSUPPORTED, "cgg_decoder_kv_proj_bf16: C=%d NK=%d (C = 256, NK %% 256)", C,
              NK);
  CGG_REQUIRE(B > 0 && hw > 0, CGG_EINVAL, "cgg_decoder_kv_proj_bf16: B=%d hw=%d", B, hw);
  CGG_REQUIRE((long long)B * hw * NK < (1ll << 31), CGG_EUNSUPPORTED, "cgg_decoder_kv_proj_bf16: output too large for 32-bit offsets");
  CGG_REQUIRE(cgg_aligned16(m16) && cgg_aligned16(mp16) && cgg_aligned16(wk_packed) && cgg_aligned16(wv_packed) && cgg_aligned16(k) &&
                  cgg_aligned16(vt),
              CGG_EALIGN, "cgg_decoder_kv_proj_bf16: 16-B alignment");
  const int bpi = (hw + EP_RB - 1) / EP_RB;
  if (hw % EP_RB == 0)
    hipLaunchKernelGGL(cgg_decoder_kv_proj_kernel<false>, dim3(B * bpi), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)m16,
                       (const uint16_t*)mp16, (const ep_u32x4*)wk_packed, bk, (const ep_u32x4*)wv_packed, (uint16_t*)k, (uint16_t*)vt,
                       hw, NK, bpi);
  else
    hipLaunchKernelGGL(cgg_decoder_kv_proj_kernel<true>, dim3(B * bpi), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)m16,
                       (const uint16_t*)mp16, (const ep_u32x4*)wk_packed, bk, (const ep_u32x4*)wv_packed, (uint16_t*)k, (uint16_t*)vt,
                       hw, NK, bpi);
  CGG_CHECK_LAUNCH("cgg_decoder_kv_proj_bf16");
  return CGG_OK;
}
