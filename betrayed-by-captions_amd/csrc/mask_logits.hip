// K3/K4/K5: mask logits  mask_pred[b,q,p] = sum_c mask_embed[b,q,c] * mask_feature[b,c,p]
// (open_set/models/mask2former_head.py:748) as v_mfma_f32_32x32x16_bf16 contractions, with the
// attention-mask rule of :749-759 (sigmoid(interp(logit)) < 0.5  <=>  interp(logit) < 0) as a
// ballot epilogue and the all-masked-row fix-up of :825-826.
//
// Roofline: HBM-bound at Q=100 (AI 56..100 FLOP/B vs ridge 312). The kernel is therefore built as a
// STREAM over the packed feature: every B operand is one coalesced 1-KiB global_load_dwordx4 that
// goes straight into MFMA registers (no LDS round trip for the streamed operand); the small,
// 100%-reused A operand (mask_embed, <= 128x256) sits in LDS in fragment order (conflict-free
// ds_read_b128); outputs leave as full 128-B lines.
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// -------------------------------------------------------------------------------------------------
// pack: [B, C, H, W] f32  ->  [B, T, C/8, 32, 8] bf16 (hi [+ lo residual]),  T = ceil(npix/32)
// thread = (pixel p, channel octet kc): 8 coalesced dword loads (lanes run along p), one 16-B store.
// pool = s > 1 packs the 2x2 mean that bilinear(align_corners=False) down-sampling by s reads.
// -------------------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(256) void cgg_pack_kernel(const float* __restrict__ feat,
                                                       u32x4* __restrict__ hi,
                                                       u32x4* __restrict__ lo, int C, int H, int W,
                                                       int pool, int Wp, int npix, int T) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= T * 32) return;
  const int kc = blockIdx.y;
  const int b = blockIdx.z;
  const int KC = C >> 3;
  float v[8];
  if (p < npix) {
    const size_t plane = (size_t)H * W;
    const float* src = feat + ((size_t)b * C + (size_t)kc * 8) * plane;
    if (pool == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[e * plane + p];
    } else {
      const int i = p / Wp, j = p - i * Wp;
      const int r0 = pool * i + (pool >> 1) - 1;
      const int c0 = pool * j + (pool >> 1) - 1;
      const size_t o00 = (size_t)r0 * W + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* s = src + e * plane + o00;
        // same association order as torch's upsample_bilinear2d with all lambdas == 0.5
        v[e] = ((s[0] + s[1]) + (s[W] + s[W + 1])) * 0.25f;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  }
  uint16_t h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (SPLIT) cgg_split_bf(v[e], h[e], l[e]);
    else h[e] = cgg_f2bf(v[e]);
  }
  const size_t slot = (((size_t)b * T + (p >> 5)) * KC + kc) * 32 + (p & 31);
  u32x4 ph = {cgg_pack2(h[0], h[1]), cgg_pack2(h[2], h[3]), cgg_pack2(h[4], h[5]),
              cgg_pack2(h[6], h[7])};
  hi[slot] = ph;
  if (SPLIT) {
    u32x4 pl = {cgg_pack2(l[0], l[1]), cgg_pack2(l[2], l[3]), cgg_pack2(l[4], l[5]),
                cgg_pack2(l[6], l[7])};
    lo[slot] = pl;
  }
}

// -------------------------------------------------------------------------------------------------
// mask logits. Workgroup = 8 waves sharing one image's mask_embed in LDS; each wave streams its own
// 32-pixel tiles: 16 k-steps x MT m-tiles of v_mfma_f32_32x32x16_bf16 (x3 in SPLIT mode).
//   A (LDS, fragment order): slot (mt, ks, lane) holds E[q = mt*32 + (lane&31)][k = ks*16 + 8*(lane>>5) ..+7]
//   B (global, packed):      slot (t, ks, lane)  holds F[k = ks*16 + 8*(lane>>5) ..+7][p = t*32 + (lane&31)]
//   D: acc[mt][r] = logit[q = mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)][p = t*32 + (lane&31)]
// -------------------------------------------------------------------------------------------------
template <int MT, bool SPLIT>
__global__ __launch_bounds__(512, 2) void cgg_mask_logits_kernel(
    const float* __restrict__ embed, const u32x4* __restrict__ fhi, const u32x4* __restrict__ flo,
    float* __restrict__ out, uint32_t* __restrict__ bits, int Q, int npix, int T) {
  constexpr int KS = 16;  // C = 256
  constexpr int C = KS * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* a_hi = reinterpret_cast<u32x4*>(smem_raw);
  u32x4* a_lo = a_hi + MT * KS * 64;

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- prologue: mask_embed[b] -> bf16 fragments in LDS ----
  // Coalesced: thread t owns float4 number t, t+512, ... of the contiguous [Q, 256] block; ALL loads
  // are issued before the first conversion (the old per-slot gather was a chain of dependent loads).
  // float4 (q, c4) lands in slot (mt = q/32, ks = c4/4, lane = q%32 + 32*((c4/2)&1)), half (c4 & 1).
  {
    const f32x4* eb4 = reinterpret_cast<const f32x4*>(embed + (size_t)b * Q * C);
    constexpr int NV = (MT * 32 * (C / 4) + 511) / 512;  // float4 per thread (rows padded to MT*32)
    f32x4 ev[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 512 * i;
      const int q = f / (C / 4);
      ev[i] = (q < Q) ? eb4[f] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    uint2* a_hi2 = reinterpret_cast<uint2*>(a_hi);
    uint2* a_lo2 = reinterpret_cast<uint2*>(a_lo);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 512 * i;
      const int q = f / (C / 4), c4 = f % (C / 4);
      if (q < MT * 32) {
        const int slot = ((q >> 5) * KS + (c4 >> 2)) * 64 + (q & 31) + 32 * ((c4 >> 1) & 1);
        uint16_t h[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (SPLIT) cgg_split_bf(ev[i][e], h[e], lw[e]);
          else h[e] = cgg_f2bf(ev[i][e]);
        }
        a_hi2[slot * 2 + (c4 & 1)] = make_uint2(cgg_pack2(h[0], h[1]), cgg_pack2(h[2], h[3]));
        if (SPLIT) a_lo2[slot * 2 + (c4 & 1)] = make_uint2(cgg_pack2(lw[0], lw[1]), cgg_pack2(lw[2], lw[3]));
      }
    }
  }
  __syncthreads();

  const int hi5 = lane >> 5;
  const int col = lane & 31;
  const int tstride = gridDim.x * 8;
  for (int t = blockIdx.x * 8 + wave; t < T; t += tstride) {
    const size_t tbase = ((size_t)b * T + t) * (KS * 64) + lane;
    u32x4 bh[KS];
    u32x4 bl[SPLIT ? KS : 1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bh[ks] = __builtin_nontemporal_load(fhi + tbase + ks * 64);
    if (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bl[ks] = __builtin_nontemporal_load(flo + tbase + ks * 64);
    }
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 vbh = __builtin_bit_cast(bf16x8, bh[ks]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 vah = __builtin_bit_cast(bf16x8, a_hi[(mt * KS + ks) * 64 + lane]);
        if (SPLIT) {
          const bf16x8 val = __builtin_bit_cast(bf16x8, a_lo[(mt * KS + ks) * 64 + lane]);
          const bf16x8 vbl = __builtin_bit_cast(bf16x8, bl[ks]);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, vbh, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, vbl, acc[mt], 0, 0, 0);
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, vbh, acc[mt], 0, 0, 0);
      }
    }

    // ---- epilogue ----
    const int p = t * 32 + col;
    if (out != nullptr) {
      const bool pin = p < npix;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi5;
          if (q < Q && pin) out[((size_t)b * Q + q) * npix + p] = acc[mt][r];
        }
      }
    }
    if (bits != nullptr) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned long long m = __ballot(acc[mt][r] < 0.f);
          const int q0 = mt * 32 + (r & 3) + 8 * (r >> 2);
          // lanes 0-31 carry row q0, lanes 32-63 row q0 + 4
          if (lane == 0 && q0 < Q) bits[((size_t)b * Q + q0) * T + t] = (uint32_t)m;
          if (lane == 32 && q0 + 4 < Q) bits[((size_t)b * Q + q0 + 4) * T + t] = (uint32_t)(m >> 32);
        }
      }
    }
  }
}

// rows whose every valid bit is set are cleared (mask2former_head.py:825-826); one wave per row
__global__ __launch_bounds__(256) void cgg_fix_full_rows_kernel(uint32_t* __restrict__ bits,
                                                                int rows, int npix, int words) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  uint32_t* rb = bits + (size_t)row * words;
  bool full = true;
  for (int w = lane; w < words; w += 64) {
    const int valid = min(32, npix - w * 32);
    const uint32_t want = valid >= 32 ? 0xffffffffu : ((1u << valid) - 1u);
    full = full && ((rb[w] & want) == want);
  }
  if (__all(full)) {
    for (int w = lane; w < words; w += 64) rb[w] = 0u;
  }
}

// generic: bilinear (align_corners=False, torch semantics) resize of stored logits, then (x<0) bits
__global__ __launch_bounds__(256) void cgg_mask_from_logits_kernel(const float* __restrict__ logits,
                                                                   uint32_t* __restrict__ bits,
                                                                   int H, int W, int h, int w,
                                                                   float sh, float sw, int words) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int npix = h * w;
  bool neg = false;
  if (p < npix) {
    const int oy = p / w, ox = p - oy * w;
    float fy = sh * (oy + 0.5f) - 0.5f;
    float fx = sw * (ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* s = logits + (size_t)n * H * W;
    const float v = hy * (hx * s[(size_t)y0 * W + x0] + lx * s[(size_t)y0 * W + x1]) +
                    ly * (hx * s[(size_t)y1 * W + x0] + lx * s[(size_t)y1 * W + x1]);
    neg = v < 0.f;
  }
  const unsigned long long m = __ballot(neg);
  const int lane = threadIdx.x & 63;
  const int wbase = (blockIdx.x * 256 + (threadIdx.x & ~63)) >> 5;
  if (lane == 0 && wbase < words) bits[(size_t)n * words + wbase] = (uint32_t)m;
  if (lane == 32 && wbase + 1 < words) bits[(size_t)n * words + wbase + 1] = (uint32_t)(m >> 32);
}

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
extern "C" int cgg_pack_mask_feature(const float* feat, void* hi, void* lo, int B, int C, int H,
                                     int W, int pool, cgg_stream_t stream) {
  CGG_REQUIRE(feat && hi, CGG_EINVAL, "cgg_pack_mask_feature: null pointer");
  CGG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, CGG_EINVAL, "cgg_pack_mask_feature: bad sizes");
  CGG_REQUIRE(C % 16 == 0, CGG_EUNSUPPORTED, "cgg_pack_mask_feature: C=%d not a multiple of 16", C);
  CGG_REQUIRE(pool == 1 || pool == 2 || pool == 4 || pool == 8, CGG_EUNSUPPORTED,
              "cgg_pack_mask_feature: pool=%d (want 1,2,4,8)", pool);
  CGG_REQUIRE(H % pool == 0 && W % pool == 0, CGG_EUNSUPPORTED,
              "cgg_pack_mask_feature: %dx%d not divisible by pool=%d", H, W, pool);
  CGG_REQUIRE(cgg_aligned16(hi) && (!lo || cgg_aligned16(lo)), CGG_EALIGN,
              "cgg_pack_mask_feature: packed buffers must be 16-B aligned");
  const int Hp = H / pool, Wp = W / pool;
  const int npix = Hp * Wp;
  const int T = (npix + 31) / 32;
  dim3 grid((T * 32 + 255) / 256, C / 8, B);
  hipStream_t s = (hipStream_t)stream;
  if (lo)
    hipLaunchKernelGGL(cgg_pack_kernel<true>, grid, dim3(256), 0, s, feat, (u32x4*)hi, (u32x4*)lo,
                       C, H, W, pool, Wp, npix, T);
  else
    hipLaunchKernelGGL(cgg_pack_kernel<false>, grid, dim3(256), 0, s, feat, (u32x4*)hi,
                       (u32x4*)nullptr, C, H, W, pool, Wp, npix, T);
  CGG_CHECK_LAUNCH("cgg_pack_mask_feature");
  return CGG_OK;
}

template <int MT, bool SPLIT>
static int launch_mask_logits(const float* embed, const void* hi, const void* lo, float* out,
                              uint32_t* bits, int B, int Q, int npix, hipStream_t s) {
  const int T = (npix + 31) / 32;
  const size_t lds = (size_t)MT * 16 * 64 * 16 * (SPLIT ? 2 : 1);
  // one workgroup per CU-slot; every wave gets >= 1 tile where possible
  int gx = (T + 7) / 8;
  int cap = (512 + B - 1) / B;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  auto kern = cgg_mask_logits_kernel<MT, SPLIT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      cgg_set_error("cgg_mask_logits: cannot raise dynamic LDS to %zu: %s", lds,
                    hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3(gx, B), dim3(512), lds, s, embed, (const u32x4*)hi,
                     (const u32x4*)lo, out, bits, Q, npix, T);
  CGG_CHECK_LAUNCH("cgg_mask_logits");
  return CGG_OK;
}

extern "C" int cgg_mask_logits(const float* embed, const void* hi, const void* lo, float* out,
                               uint32_t* bits, int B, int Q, int C, int npix, cgg_stream_t stream) {
  CGG_REQUIRE(embed && hi, CGG_EINVAL, "cgg_mask_logits: null pointer");
  CGG_REQUIRE(out || bits, CGG_EINVAL, "cgg_mask_logits: neither out nor bits requested");
  CGG_REQUIRE(B > 0 && Q > 0 && npix > 0, CGG_EINVAL, "cgg_mask_logits: bad sizes");
  CGG_REQUIRE(C == 256, CGG_EUNSUPPORTED, "cgg_mask_logits: C=%d (only 256 is built)", C);
  CGG_REQUIRE(cgg_aligned16(embed) && cgg_aligned16(hi) && (!lo || cgg_aligned16(lo)), CGG_EALIGN,
              "cgg_mask_logits: embed / packed buffers must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  const int mt = (Q + 31) / 32;
  if (lo) {
    CGG_REQUIRE(mt <= 4, CGG_EUNSUPPORTED, "cgg_mask_logits: split mode supports Q <= 128 (Q=%d)", Q);
    switch (mt) {
      case 1: return launch_mask_logits<1, true>(embed, hi, lo, out, bits, B, Q, npix, s);
      case 2: return launch_mask_logits<2, true>(embed, hi, lo, out, bits, B, Q, npix, s);
      case 3: return launch_mask_logits<3, true>(embed, hi, lo, out, bits, B, Q, npix, s);
      default: return launch_mask_logits<4, true>(embed, hi, lo, out, bits, B, Q, npix, s);
    }
  }
  CGG_REQUIRE(mt <= 8, CGG_EUNSUPPORTED, "cgg_mask_logits: Q <= 256 (Q=%d)", Q);
  switch (mt) {
    case 1: return launch_mask_logits<1, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    case 2: return launch_mask_logits<2, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    case 3: return launch_mask_logits<3, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    case 4: return launch_mask_logits<4, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    case 5: return launch_mask_logits<5, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    case 6: return launch_mask_logits<6, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    case 7: return launch_mask_logits<7, false>(embed, hi, lo, out, bits, B, Q, npix, s);
    default: return launch_mask_logits<8, false>(embed, hi, lo, out, bits, B, Q, npix, s);
  }
}

extern "C" int cgg_attn_mask_fix_full_rows(uint32_t* bits, int rows, int npix, cgg_stream_t stream) {
  CGG_REQUIRE(bits, CGG_EINVAL, "cgg_attn_mask_fix_full_rows: null pointer");
  CGG_REQUIRE(rows > 0 && npix > 0, CGG_EINVAL, "cgg_attn_mask_fix_full_rows: bad sizes");
  const int words = (npix + 31) / 32;
  hipLaunchKernelGGL(cgg_fix_full_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, bits, rows, npix, words);
  CGG_CHECK_LAUNCH("cgg_attn_mask_fix_full_rows");
  return CGG_OK;
}

extern "C" int cgg_attn_mask_from_logits(const float* logits, uint32_t* bits, int N, int H, int W,
                                         int h, int w, cgg_stream_t stream) {
  CGG_REQUIRE(logits && bits, CGG_EINVAL, "cgg_attn_mask_from_logits: null pointer");
  CGG_REQUIRE(N > 0 && H > 0 && W > 0 && h > 0 && w > 0, CGG_EINVAL,
              "cgg_attn_mask_from_logits: bad sizes");
  const int npix = h * w;
  const int words = (npix + 31) / 32;
  dim3 grid((words * 32 + 255) / 256, N);
  hipLaunchKernelGGL(cgg_mask_from_logits_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits,
                     bits, H, W, h, w, (float)H / (float)h, (float)W / (float)w, words);
  CGG_CHECK_LAUNCH("cgg_attn_mask_from_logits");
  return CGG_OK;
}
