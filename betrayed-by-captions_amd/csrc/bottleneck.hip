// ResNet-50 identity Bottleneck of layer1 (mmdet ResNet `Bottleneck.forward`, selected at configs/instance/coco_b48n17.py:17-26;
// BN folded into the convolutions, channel-last bf16 activations) as ONE launch:
//
//     t1 = relu(conv1x1(x, W1) + b1)          256 -> 64
//     t2 = relu(conv3x3(t1, W2, pad 1) + b2)   64 -> 64
//     y  = relu(conv1x1(t2, W3) + b3 + x)      64 -> 256
//
// The three library calls (+ a bias / ReLU pass) move 303 MB per block at configs[1] (131 072 pixels): x read twice, the 64-channel
// intermediates written and read back. Here a workgroup (4 wavefronts) owns an 8 x 16 tile of output pixels; t1 is computed on the
// tile's 10 x 18 halo (1.4x recompute of the cheap 1x1 convolution) and lives, like t2, only in LDS (27 + 18 KB): x is read once
// (+ the residual re-read of the tile centre, which is still in L2), y written once.
//   phase 1: C[192 halo px, 64] = X[192, 256] W1^T        A fragments straight from global (a pixel's 8 channels per lane and
//            k-step: the two halves of a lane pair read adjacent 16-byte pieces, a 128-byte line is consumed by 4 consecutive k-steps;
//            giving each lane a contiguous 256-byte walk instead doubles the lines per load instruction: 69 -> 80 us),
//            W1 fragments from the packed image; bias, ReLU, ZERO outside the image (the padding of conv2 is a padding
//            of t1, not of x) -> t1 [pixel][64 ch] in LDS (row stride 144 B: conflict-free 16-byte fragment reads)
//   phase 2: C[128 px, 64] = im2col(t1)[128, 9 * 64] W2^T  implicit GEMM: the A fragment of k-step (tap, 16-channel block) is the
//            t1 row of the shifted pixel; wave w owns tile rows 2 w, 2 w + 1; -> t2 in the wave's private LDS block
//   phase 3: C[128 px, 256] = t2[128, 64] W3^T            + b3 + x, ReLU, bf16; W3 packed with its columns interleaved in pairs so
//            that a lane holds 4 consecutive channels of a pixel (8-byte residual loads / stores, 128-byte runs)
// v_mfma_f32_32x32x16_bf16, f32 accumulation, the bf16 rounding points of the three-call path (t1, t2, y) kept.
//
// build-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t bn_u32x4;
typedef __attribute__((ext_vector_type(2))) float bn_f2;
typedef __attribute__((ext_vector_type(2))) __bf16 bn_bf2;

#define BN_TH 8                 // tile height (output pixels)
#define BN_TW 16                // tile width
#define BN_HW (BN_TW + 2)       // halo width 18
#define BN_HALO ((BN_TH + 2) * BN_HW)   // 180 halo pixels (padded to 6 m-tiles = 192)
#define BN_TS 72                // LDS row stride of t1 / t2 in bf16 elements (144 B)
#define BN_PF 4                 // global prefetch distance (k-steps)

__device__ __forceinline__ uint32_t bn_pk(float a, float b) {
  const bn_f2 pr = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bn_bf2));
}

// lanes j, j ^ 1 hold neighbouring columns: they swap one value per register pair; afterwards the even lane owns row(2 rp), the odd
// lane row(2 rp + 1), each with the bf16 pair of columns (j & ~1, (j & ~1) + 1)
__device__ __forceinline__ uint32_t bn_pair(float v0, float v1, int odd, uint32_t rot) {
  const float kept = odd ? v1 : v0, sent = odd ? v0 : v1;
  const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sent), 0xB1, 0xf, 0xf, true));
  const uint32_t w = bn_pk(kept, recv);
  return __builtin_amdgcn_alignbit(w, w, rot);
}

__global__ __launch_bounds__(256) void cgg_bottleneck64_kernel(const uint16_t* __restrict__ x, const bn_u32x4* __restrict__ w1,
                                                              const float* __restrict__ b1, const bn_u32x4* __restrict__ w2,
                                                              const float* __restrict__ b2, const bn_u32x4* __restrict__ w3,
                                                              const float* __restrict__ b3, uint16_t* __restrict__ y, int H, int W) {
  __shared__ __attribute__((aligned(16))) uint16_t t1[192 * BN_TS];          // 27 KiB
  __shared__ __attribute__((aligned(16))) uint16_t t2[4][32 * BN_TS];        // 18 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi5 = lane >> 5, odd = j & 1;
  const uint32_t rot = 16u * (uint32_t)odd;
  const int tiles_x = W / BN_TW, tiles_y = H / BN_TH;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int y0 = ty * BN_TH, x0 = tx * BN_TW;
  const uint16_t* xb = x + (size_t)b * H * W * 256;

  // ---------------- phase 1: t1 on the halo ----------------
  {
    const int nt = wave & 1;
    const bn_u32x4* wb = w1 + (size_t)nt * 16 * 64 + lane;
    const uint16_t* arow[3];
    bool aval[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int hp = 32 * ((wave >> 1) + 2 * m) + j;                       // this lane's halo pixel in m-tile (wave >> 1) + 2 m
      const int hy = hp / BN_HW, hx = hp - hy * BN_HW;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      aval[m] = hp < BN_HALO && iy >= 0 && iy < H && ix >= 0 && ix < W;
      arow[m] = xb + ((size_t)(aval[m] ? iy : 0) * W + (aval[m] ? ix : 0)) * 256 + 8 * hi5;
    }
    f32x16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    bn_u32x4 qa[3][BN_PF], qb[BN_PF];
#pragma unroll
    for (int s = 0; s < BN_PF; ++s) {
      qb[s] = wb[s * 64];
#pragma unroll
      for (int m = 0; m < 3; ++m) qa[m][s] = *reinterpret_cast<const bn_u32x4*>(arow[m] + 16 * s);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bf16x8 vb = __builtin_bit_cast(bf16x8, qb[s % BN_PF]);
      bf16x8 va[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) va[m] = __builtin_bit_cast(bf16x8, qa[m][s % BN_PF]);
      if (s + BN_PF < 16) {
        qb[s % BN_PF] = wb[(s + BN_PF) * 64];
#pragma unroll
        for (int m = 0; m < 3; ++m) qa[m][s % BN_PF] = *reinterpret_cast<const bn_u32x4*>(arow[m] + 16 * (s + BN_PF));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[m], vb, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // bias + ReLU, zero outside the image, -> t1[halo pixel][channel]
    const float bias = b1[32 * nt + j];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const uint32_t pk = bn_pair(fmaxf(acc[m][2 * rp] + bias, 0.f), fmaxf(acc[m][2 * rp + 1] + bias, 0.f), odd, rot);
        const int hp = 32 * ((wave >> 1) + 2 * m) + 2 * (rp & 1) + 8 * (rp >> 1) + 4 * hi5 + odd;
        const int hy = hp / BN_HW, hx = hp - hy * BN_HW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool in = hp < BN_HALO && iy >= 0 && iy < H && ix >= 0 && ix < W;
        *reinterpret_cast<uint32_t*>(&t1[hp * BN_TS + 32 * nt + (j & ~1)]) = in ? pk : 0u;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2: 3 x 3 convolution on the tile, wave w = tile rows 2 w, 2 w + 1 ----------------
  {
    const int r = j >> 4, c = j & 15;                                      // pixel of the m-tile this lane feeds as an A row
    const uint16_t* abase = t1 + ((2 * wave + r) * BN_HW + c) * BN_TS + 8 * hi5;
    const bn_u32x4* wb0 = w2 + lane;
    const bn_u32x4* wb1 = w2 + (size_t)36 * 64 + lane;
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
    bn_u32x4 q0[BN_PF], q1[BN_PF];
#pragma unroll
    for (int s = 0; s < BN_PF; ++s) {
      q0[s] = wb0[s * 64];
      q1[s] = wb1[s * 64];
    }
    bn_u32x4 ua = *reinterpret_cast<const bn_u32x4*>(abase);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const bf16x8 va = __builtin_bit_cast(bf16x8, ua);
      const bf16x8 vb0 = __builtin_bit_cast(bf16x8, q0[s % BN_PF]), vb1 = __builtin_bit_cast(bf16x8, q1[s % BN_PF]);
      if (s + 1 < 36) {
        const int tap = (s + 1) >> 2, cb = (s + 1) & 3;
        ua = *reinterpret_cast<const bn_u32x4*>(abase + ((tap / 3) * BN_HW + tap % 3) * BN_TS + 16 * cb);
      }
      if (s + BN_PF < 36) {
        q0[s % BN_PF] = wb0[(s + BN_PF) * 64];
        q1[s % BN_PF] = wb1[(s + BN_PF) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb1, acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float bias = b2[32 * n + j];
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const uint32_t pk = bn_pair(fmaxf(acc[n][2 * rp] + bias, 0.f), fmaxf(acc[n][2 * rp + 1] + bias, 0.f), odd, rot);
        const int p = 2 * (rp & 1) + 8 * (rp >> 1) + 4 * hi5 + odd;       // pixel of the m-tile
        *reinterpret_cast<uint32_t*>(&t2[wave][p * BN_TS + 32 * n + (j & ~1)]) = pk;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 3: 64 -> 256, + b3 + x, ReLU ----------------
  {
    const uint16_t* abase = &t2[wave][j * BN_TS + 8 * hi5];
    bn_u32x4 ua[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ua[s] = *reinterpret_cast<const bn_u32x4*>(abase + 16 * s);
#pragma unroll 1
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x16 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
      bn_u32x4 qw[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) qw[t][s] = w3[((size_t)(4 * h2 + t) * 4 + s) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[s]), __builtin_bit_cast(bf16x8, qw[t][s]),
                                                          acc[t], 0, 0, 0);
      // two 64-column groups per half; inside a group tile t' (0 / 1), lane column j <-> channel 64 g + 4 (j / 2) + 2 t' + (j & 1)
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const int g = 2 * h2 + gq;
        const int ch = 64 * g + 4 * (j >> 1);
        const float bs0 = b3[ch + (j & 1)], bs1 = b3[ch + 2 + (j & 1)];
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
          const int p = 2 * (rp & 1) + 8 * (rp >> 1) + 4 * hi5 + odd;      // pixel of the wave's m-tile: row p >> 4, column p & 15
          const size_t off = ((size_t)(y0 + 2 * wave + (p >> 4)) * W + x0 + (p & 15)) * 256 + ch;
          const uint2 res = *reinterpret_cast<const uint2*>(xb + off);
          const float r00 = __uint_as_float(res.x << 16), r01 = __uint_as_float(res.x & 0xffff0000u);
          const float r10 = __uint_as_float(res.y << 16), r11 = __uint_as_float(res.y & 0xffff0000u);
          // the swap exchanges values of different ROWS, so the residual (a property of the final row) is added after it; the bias
          // (a property of the column, which a value keeps) before it
          const float a0 = acc[2 * gq][2 * rp] + bs0, a1 = acc[2 * gq][2 * rp + 1] + bs0;
          const float c0 = acc[2 * gq + 1][2 * rp] + bs1, c1 = acc[2 * gq + 1][2 * rp + 1] + bs1;
          // after the swap: even lane = row(2 rp): columns (ch, ch + 1) from tile 0 and (ch + 2, ch + 3) from tile 1
          const float k0 = odd ? a1 : a0, s0 = odd ? a0 : a1;
          const float k1 = odd ? c1 : c0, s1 = odd ? c0 : c1;
          const float v0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xf, 0xf, true));
          const float v1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xf, 0xf, true));
          // even lane: (k0, v0) = columns (ch, ch + 1); odd lane: (v0, k0)
          const float e0 = odd ? v0 : k0, e1 = odd ? k0 : v0, e2 = odd ? v1 : k1, e3 = odd ? k1 : v1;
          const uint32_t o0 = bn_pk(fmaxf(e0 + r00, 0.f), fmaxf(e1 + r01, 0.f));
          const uint32_t o1 = bn_pk(fmaxf(e2 + r10, 0.f), fmaxf(e3 + r11, 0.f));
          *reinterpret_cast<uint2*>(y + (size_t)b * H * W * 256 + off) = make_uint2(o0, o1);
        }
      }
    }
  }
}

extern "C" int cgg_bottleneck64_bf16(const void* x, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                     const void* w3_packed, const float* b3, void* y, int B, int H, int W, int C, int CMID,
                                     cgg_stream_t stream) {
  CGG_REQUIRE(x && w1_packed && b1 && w2_packed && b2 && w3_packed && b3 && y, CGG_EINVAL, "cgg_bottleneck64_bf16: null pointer");
  CGG_REQUIRE(C == 256 && CMID == 64, CGG_EUNSUPPORTED, "cgg_bottleneck64_bf16: C=%d CMID=%d (256 / 64 is built)", C, CMID);
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && H % BN_TH == 0 && W % BN_TW == 0, CGG_EUNSUPPORTED,
              "cgg_bottleneck64_bf16: H=%d W=%d must be multiples of %d / %d", H, W, BN_TH, BN_TW);
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(y) && cgg_aligned16(w1_packed) && cgg_aligned16(w2_packed) && cgg_aligned16(w3_packed),
              CGG_EALIGN, "cgg_bottleneck64_bf16: 16-B alignment");
  CGG_REQUIRE((long long)B * H * W * 256 < (1ll << 31), CGG_EUNSUPPORTED, "cgg_bottleneck64_bf16: activation too large");
  const int nblk = B * (H / BN_TH) * (W / BN_TW);
  hipLaunchKernelGGL(cgg_bottleneck64_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (const bn_u32x4*)w1_packed, b1, (const bn_u32x4*)w2_packed, b2, (const bn_u32x4*)w3_packed, b3, (uint16_t*)y, H, W);
  CGG_CHECK_LAUNCH("cgg_bottleneck64_bf16");
  return CGG_OK;
}
