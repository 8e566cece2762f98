// ResNet stem convolution (7x7, stride 2, padding 3, 3 -> 64 channels; [3P] mmdet ResNet.conv1 with BN folded) as a
// hand-written MFMA kernel: f32 NCHW image in, RAW bf16 channel-last convolution out (bias + ReLU + max-pool follow in
// cgg_bias_relu_maxpool_nhwc). Replaces: image cast/permute pass (16 us) + MIOpen's split-K implicit GEMM with its
// zero-fill pass (17 + 62 us at configs[1]); the library has no good solver for C_in = 3.
//
// GEMM view: out[cout, pixel] = sum_k W[cout, k] * patch[k, pixel], k = (ky, kx, c) with every ky-row padded from 21 to
// 24 entries and K from 168 to 176 = 11 MFMA k-steps. The input tile lives in LDS channel-interleaved ([row][col][3]
// bf16), so the 24 k-entries of one ky for one output pixel are 24 CONSECUTIVE LDS elements starting at pixel
// (2 oy + ky, 2 ox): a B fragment (8 k-values of one pixel) is one 16-byte LDS read at a 4-byte-aligned address; the
// 8th pixel of each ky-row (kx = 7) and the K tail meet zero weights. A = the packed weights (2 m-tiles x 11 k-steps,
// 88 VGPRs per wave, loaded once). Workgroup = 8 x 32 output pixels, 4 waves x 2 rows; output staged through LDS for
// 16-byte coalesced channel-last stores.
// X3 = true (parity mode): the same kernel on the f32-class f16 x 3 contraction (x3.h): weight and tile are split into hi / lo
// f16 pieces (weights pre-scaled per output channel on the host, tile by 2^4), three MFMAs per fragment pair, RAW f32
// channel-last convolution out (un-scaled in the epilogue); replaces MIOpen's f32 solver (213 us at configs[1]).
#include "x3.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define ST_TH 8                        // output rows per workgroup
#define ST_TW 32                       // output columns per workgroup
#define ST_IH (2 * ST_TH + 5)          // 21 input rows
#define ST_IW (2 * ST_TW + 5)          // 69 input columns
#define ST_IWP 72                      // padded row length (pixels): reads run up to 2 pixels past column 68
#define ST_KS 11                       // k-steps of 16

template <bool X3>
__global__ __launch_bounds__(256) void cgg_stem_conv7x7_kernel(const float* __restrict__ img, const u32x4* __restrict__ wp,
                                                              void* __restrict__ out_, const float* __restrict__ wscale, int H,
                                                              int W, int Ho, int Wo) {
  // LDS: [weights 22 KiB | input tile 9.3 KiB] (x3: both twice, hi and lo pieces) during the contraction, re-used as the
  // [pixel][64] output staging buffer afterwards (32 KiB bf16 / 64 KiB f32)
  constexpr int NP = X3 ? 2 : 1;                                      // pieces
  constexpr int WFRAG = 2 * ST_KS * 64;                               // u32x4 entries per piece
  constexpr int TILE_EL = (ST_IH + 1) * ST_IWP * 3;
  constexpr int OSZ = X3 ? 4 : 2;
  constexpr int LDS_BYTES = NP * (WFRAG * 16 + TILE_EL * 2) > (ST_TH * ST_TW * 64 * OSZ) ? NP * (WFRAG * 16 + TILE_EL * 2)
                                                                                           : (ST_TH * ST_TW * 64 * OSZ);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  u32x4* wl = reinterpret_cast<u32x4*>(lds_raw);                                          // hi fragments | lo fragments
  uint16_t* tile = reinterpret_cast<uint16_t*>(lds_raw + NP * WFRAG * 16);               // hi tile | lo tile
  uint16_t* obuf = reinterpret_cast<uint16_t*>(lds_raw);
  uint16_t* out = reinterpret_cast<uint16_t*>(out_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, oy0 = blockIdx.y * ST_TH, ox0 = blockIdx.x * ST_TW;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

  // weights -> LDS (fragment order, conflict-free ds_read_b128 later)
  u32x4 wv[(NP * WFRAG + 255) / 256];
#pragma unroll
  for (int it = 0; it < (NP * WFRAG + 255) / 256; ++it) {
    const int i = tid + 256 * it;
    wv[it] = i < NP * WFRAG ? wp[i] : u32x4{0u, 0u, 0u, 0u};
  }

  // input tile: NCHW f32 -> [row][col][c] bf16, zero outside the image (padding 3) and in the pad columns / row
  const float* ib = img + (size_t)b * 3 * H * W;
  constexpr int NEL = (ST_IH + 1) * ST_IWP * 3;
  constexpr int NIT = (NEL + 255) / 256;
  float tv[NIT];
  // all global loads of the tile are issued before the first LDS store (a rolled load -> convert -> store loop
  // serialised one L2 round trip per element: 57 us for the kernel)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + 256 * it;
    const int c = i / ((ST_IH + 1) * ST_IWP);                 // channel-major sweep: coalesced global rows
    const int rem = i - c * ((ST_IH + 1) * ST_IWP);
    const int r = rem / ST_IWP, col = rem - r * ST_IWP;
    const int iy = iy0 + r, ix = ix0 + col;
    tv[it] = 0.f;
    if (i < NEL && r < ST_IH && col < ST_IW && iy >= 0 && iy < H && ix >= 0 && ix < W) tv[it] = ib[((size_t)c * H + iy) * W + ix];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + 256 * it;
    const int c = i / ((ST_IH + 1) * ST_IWP);
    const int rem = i - c * ((ST_IH + 1) * ST_IWP);
    const int r = rem / ST_IWP, col = rem - r * ST_IWP;
    if (i < NEL) {
      if constexpr (X3) {
        uint16_t h, l;
        cgg_x3_split1(tv[it], h, l);
        tile[(r * ST_IWP + col) * 3 + c] = h;
        tile[TILE_EL + (r * ST_IWP + col) * 3 + c] = l;
      } else {
        tile[(r * ST_IWP + col) * 3 + c] = cgg_f2bf(tv[it]);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < (NP * WFRAG + 255) / 256; ++it) {
    const int i = tid + 256 * it;
    if (i < NP * WFRAG) wl[i] = wv[it];
  }
  __syncthreads();

  // each wave: output rows 2 wave, 2 wave + 1 (n-tiles of 32 pixels), both m-tiles (64 output channels)
  f32x16 acc[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < ST_KS; ++ks) {
    // this lane's 8 k-values: group g = 2 ks + hi -> ky = g / 3, entries 8 (g % 3) .. + 7 of that ky-row
    const int g = 2 * ks + hi;
    const int ky = g / 3, part = g - 3 * ky;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int oyl = 2 * wave + nt;
      const uint16_t* src = tile + ((2 * oyl + ky) * ST_IWP + 2 * j) * 3 + 8 * part;
      const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);             // 4-byte aligned (even element index)
      const u32x4 bv = {s32[0], s32[1], s32[2], s32[3]};
      if constexpr (X3) {
        const uint32_t* l32 = reinterpret_cast<const uint32_t*>(src + TILE_EL);
        const u32x4 bl = {l32[0], l32[1], l32[2], l32[3]};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)      // A = weights (hi, lo), B = patches (hi, lo)
          cgg_x3_mfma(acc[nt][mt], wl[(mt * ST_KS + ks) * 64 + lane], wl[WFRAG + (mt * ST_KS + ks) * 64 + lane], bv, bl);
      } else {
        const bf16x8 vb = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[(mt * ST_KS + ks) * 64 + lane]), vb,
                                                                acc[nt][mt], 0, 0, 0);
      }
    }
  }
  __syncthreads();                       // weights / tile are dead: the region becomes the output staging buffer
  if constexpr (X3) {
    // f32 staging [pixel][64]; un-scale: weight row scale (per output channel) x 2^-4 of the tile
    float* ob32 = reinterpret_cast<float*>(lds_raw);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pix = (2 * wave + nt) * ST_TW + j;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ch = 32 * mt + 8 * gq + 4 * hi;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(wscale + ch);
          *reinterpret_cast<f32x4*>(ob32 + pix * 64 + ch) =
              f32x4{acc[nt][mt][4 * gq] * sc[0], acc[nt][mt][4 * gq + 1] * sc[1], acc[nt][mt][4 * gq + 2] * sc[2],
                    acc[nt][mt][4 * gq + 3] * sc[3]};
        }
    }
    __syncthreads();
    float* o32 = reinterpret_cast<float*>(out_);
    for (int i = tid; i < ST_TH * ST_TW * 16; i += 256) {
      const int pix = i >> 4, v4 = i & 15;
      const int oy = oy0 + pix / ST_TW, ox = ox0 + pix % ST_TW;
      if (oy < Ho && ox < Wo)
        *reinterpret_cast<f32x4*>(o32 + (((size_t)b * Ho + oy) * Wo + ox) * 64 + v4 * 4) =
            *reinterpret_cast<const f32x4*>(ob32 + pix * 64 + v4 * 4);
    }
    return;
  }
  // D: lane (pixel j of row oyl, hi) holds channels 32 mt + (r&3) + 8 (r>>2) + 4 hi -> LDS [pixel][64]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int pix = (2 * wave + nt) * ST_TW + j;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ch = 32 * mt + 8 * gq + 4 * hi;
        *reinterpret_cast<uint2*>(obuf + pix * 64 + ch) =
            make_uint2(cgg_pack2(cgg_f2bf(acc[nt][mt][4 * gq]), cgg_f2bf(acc[nt][mt][4 * gq + 1])),
                       cgg_pack2(cgg_f2bf(acc[nt][mt][4 * gq + 2]), cgg_f2bf(acc[nt][mt][4 * gq + 3])));
      }
  }
  __syncthreads();
  // coalesced channel-last stores: 8 x 16-byte vectors per pixel
  for (int i = tid; i < ST_TH * ST_TW * 8; i += 256) {
    const int pix = i >> 3, v8 = i & 7;
    const int oy = oy0 + pix / ST_TW, ox = ox0 + pix % ST_TW;
    if (oy < Ho && ox < Wo)
      *reinterpret_cast<uint4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * 64 + v8 * 8) =
          *reinterpret_cast<const uint4*>(obuf + pix * 64 + v8 * 8);
  }
}

extern "C" int64_t cgg_stem_conv7x7_packed_bytes(void) { return (int64_t)2 * ST_KS * 64 * 16; }

extern "C" int cgg_stem_conv7x7_nchw(const float* img, const void* w_packed, void* out, int B, int H, int W,
                                     cgg_stream_t stream) {
  CGG_REQUIRE(img && w_packed && out, CGG_EINVAL, "cgg_stem_conv7x7_nchw: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0, CGG_EINVAL, "cgg_stem_conv7x7_nchw: bad sizes");
  CGG_REQUIRE(cgg_aligned16(w_packed) && cgg_aligned16(out), CGG_EALIGN, "cgg_stem_conv7x7_nchw: alignment");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(cgg_stem_conv7x7_kernel<false>, dim3((Wo + ST_TW - 1) / ST_TW, (Ho + ST_TH - 1) / ST_TH, B), dim3(256), 0,
                     (hipStream_t)stream, img, (const u32x4*)w_packed, out, (const float*)nullptr, H, W, Ho, Wo);
  CGG_CHECK_LAUNCH("cgg_stem_conv7x7_nchw");
  return CGG_OK;
}

// parity mode: w_packed = hi fragments | lo fragments (2 x cgg_stem_conv7x7_packed_bytes() bytes, f16 pieces of the per-channel
// pre-scaled filter), wscale[64] = the factors that un-scale the accumulators (incl. the 2^-4 of the tile); out f32 channel-last
extern "C" int cgg_stem_conv7x7_x3_nchw(const float* img, const void* w_packed, const float* wscale, float* out, int B, int H, int W,
                                        cgg_stream_t stream) {
  CGG_REQUIRE(img && w_packed && wscale && out, CGG_EINVAL, "cgg_stem_conv7x7_x3_nchw: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0, CGG_EINVAL, "cgg_stem_conv7x7_x3_nchw: bad sizes");
  CGG_REQUIRE(cgg_aligned16(w_packed) && cgg_aligned16(out) && cgg_aligned16(wscale), CGG_EALIGN, "cgg_stem_conv7x7_x3_nchw: alignment");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(cgg_stem_conv7x7_kernel<true>, dim3((Wo + ST_TW - 1) / ST_TW, (Ho + ST_TH - 1) / ST_TH, B), dim3(256), 0,
                     (hipStream_t)stream, img, (const u32x4*)w_packed, (void*)out, wscale, H, W, Ho, Wo);
  CGG_CHECK_LAUNCH("cgg_stem_conv7x7_x3_nchw");
  return CGG_OK;
}
