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
//
// SPLIT = true is parity mode's f32-class form (x3.h): both operands are split into two f16 pieces (pre-scaled by 2^4 each,
// un-scaled by 2^-8 in the epilogue), three v_mfma_f32_32x32x16_f16 per k-step into the one accumulator -- as accurate as
// an f32 GEMM (rounds 1-2 used a 3 x bf16 split, 8-10 x less accurate, and an exact-f32 MFMA kernel at 172 us).
#include "x3.h"

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
    if (SPLIT) cgg_x3_split1(v[e], h[e], l[e]);
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

// pack from a channel-last bf16 map (the throughput-mode pixel decoder's mask_feature GEMM output):
// [B, H, W, C] bf16 -> [B, T, C/8, 32, 8] bf16. A pure permutation of 16-byte chunks (plus the 2x2 mean when
// pool > 1): one block = one 32-pixel tile; chunks are read pixel-major (512 contiguous bytes per pixel) and
// written octet-major through LDS so that both sides move full lines.
struct PackJobs { int n; int pool[4]; int Wp[4]; int npix[4]; int T[4]; int t0[4]; u32x4* hi[4]; };

// up to 4 packed images (full resolution + the pooled ones the decoder levels need) from ONE launch: a block's tile
// index selects the job
__global__ __launch_bounds__(256) void cgg_pack_nhwc_kernel(const uint4* __restrict__ feat, PackJobs jobs, int KC, int H,
                                                            int W) {
  extern __shared__ __attribute__((aligned(16))) u32x4 tile[];   // [KC][32]
  int job = 0;
  while (job + 1 < jobs.n && (int)blockIdx.x >= jobs.t0[job + 1]) ++job;
  const int t = blockIdx.x - jobs.t0[job], b = blockIdx.y;
  const int pool = jobs.pool[job], Wp = jobs.Wp[job], npix = jobs.npix[job], T = jobs.T[job];
  u32x4* __restrict__ hi = jobs.hi[job];
  const int n = KC * 32;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    const int pl = idx / KC, kc = idx - pl * KC;
    const int p = t * 32 + pl;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (p < npix) {
      if (pool == 1) {
        const uint4 v = feat[((size_t)b * H * W + p) * KC + kc];
        o = u32x4{v.x, v.y, v.z, v.w};
      } else {
        const int i = p / Wp, j = p - i * Wp;
        const int r0 = pool * i + (pool >> 1) - 1, c0 = pool * j + (pool >> 1) - 1;
        const size_t base = ((size_t)b * H * W + (size_t)r0 * W + c0) * KC + kc;
        const uint4 v00 = feat[base], v01 = feat[base + KC];
        const uint4 v10 = feat[base + (size_t)W * KC], v11 = feat[base + (size_t)W * KC + KC];
        const uint32_t a[4] = {v00.x, v00.y, v00.z, v00.w}, c[4] = {v01.x, v01.y, v01.z, v01.w};
        const uint32_t d[4] = {v10.x, v10.y, v10.z, v10.w}, e[4] = {v11.x, v11.y, v11.z, v11.w};
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // same association order as the NCHW pack kernel / torch's bilinear with all lambdas == 0.5
          const float lo = ((__uint_as_float(a[k] << 16) + __uint_as_float(c[k] << 16)) +
                            (__uint_as_float(d[k] << 16) + __uint_as_float(e[k] << 16))) * 0.25f;
          const float hh = ((__uint_as_float(a[k] & 0xffff0000u) + __uint_as_float(c[k] & 0xffff0000u)) +
                            (__uint_as_float(d[k] & 0xffff0000u) + __uint_as_float(e[k] & 0xffff0000u))) * 0.25f;
          r[k] = cgg_pack2(cgg_f2bf(lo), cgg_f2bf(hh));
        }
        o = u32x4{r[0], r[1], r[2], r[3]};
      }
    }
    tile[kc * 32 + pl] = o;
  }
  __syncthreads();
  u32x4* dst = hi + ((size_t)b * T + t) * n;
  for (int idx = threadIdx.x; idx < n; idx += 256) dst[idx] = tile[idx];
}

// parity mode's pack: channel-last F32 map [B, H, W, C] -> the x3 images (hi, lo: [B, T, C/8, 32, 8] f16 pieces, x3.h) of up to 4
// jobs (full resolution + the 2x2-mean images of the decoder levels) from one launch. One block = one 32-pixel tile: 32-byte
// channel octets are read pixel-major (1 KiB contiguous per pixel), split, and written octet-major through LDS.
struct PackJobsX3 { int n; int pool[4]; int Wp[4]; int npix[4]; int T[4]; int t0[4]; u32x4* hi[4]; u32x4* lo[4]; };

__global__ __launch_bounds__(256) void cgg_pack_nhwc_f32_x3_kernel(const float* __restrict__ feat, PackJobsX3 jobs, int KC, int H,
                                                                   int W) {
  extern __shared__ __attribute__((aligned(16))) u32x4 tile[];   // hi [KC][32] | lo [KC][32]
  int job = 0;
  while (job + 1 < jobs.n && (int)blockIdx.x >= jobs.t0[job + 1]) ++job;
  const int t = blockIdx.x - jobs.t0[job], b = blockIdx.y;
  const int pool = jobs.pool[job], Wp = jobs.Wp[job], npix = jobs.npix[job], T = jobs.T[job];
  const int n = KC * 32;
  const size_t C = (size_t)KC * 8;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    const int pl = idx / KC, kc = idx - pl * KC;
    const int p = t * 32 + pl;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (p < npix) {
      if (pool == 1) {
        const float* s0 = feat + ((size_t)b * H * W + p) * C + kc * 8;
        v0 = *reinterpret_cast<const f32x4*>(s0);
        v1 = *reinterpret_cast<const f32x4*>(s0 + 4);
      } else {
        const int i = p / Wp, j = p - i * Wp;
        const int r0 = pool * i + (pool >> 1) - 1, c0 = pool * j + (pool >> 1) - 1;
        const float* s00 = feat + ((size_t)b * H * W + (size_t)r0 * W + c0) * C + kc * 8;
        const float* s10 = s00 + (size_t)W * C;
        // same association order as the NCHW pack kernel / torch's bilinear with all lambdas == 0.5
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(s00 + 4 * h), c = *reinterpret_cast<const f32x4*>(s00 + C + 4 * h);
          const f32x4 d = *reinterpret_cast<const f32x4*>(s10 + 4 * h), e = *reinterpret_cast<const f32x4*>(s10 + C + 4 * h);
          const f32x4 m = ((a + c) + (d + e)) * 0.25f;
          if (h == 0) v0 = m; else v1 = m;
        }
      }
    }
    u32x4 ph, plo;
    cgg_x3_split8(v0, v1, ph, plo);
    tile[kc * 32 + pl] = ph;
    tile[n + kc * 32 + pl] = plo;
  }
  __syncthreads();
  u32x4* dh = jobs.hi[job] + ((size_t)b * T + t) * n;
  u32x4* dl = jobs.lo[job] + ((size_t)b * T + t) * n;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    dh[idx] = tile[idx];
    dl[idx] = tile[n + idx];
  }
}

// Point sampling of the channel-last f32 mask feature STRAIGHT into x3 images (training, parity mode: the matching costs need
// mask_embed . sample(mask_feature) at the P random points of every decoder layer, open_set/models/mask2former_head.py:899-921 with
// sample(E F) = E sample(F)): round 5 wrote the samples as (B, n P, C) f32 rows (2 GB at configs[2]) and ran one f32 library bmm
// per layer over them (10 x 234 us at 44 TF/s). Here a workgroup samples one 32-point tile -- the arithmetic of
// cgg_point_sample_nhwc_kernel (ATen's grid_sampler_2d order), a point's taps read as 1-KiB channel rows by 32 lanes -- splits the
// values into their f16 pieces and writes the tile octet-major through LDS like the pack kernel above: the result IS the packed B
// operand of cgg_mask_logits' split mode, one image per (layer, batch image) in layer-major order, so the per-layer einsum runs
// on the f32-class MFMA kernel with no pack pass and no f32 sample tensor.
__global__ __launch_bounds__(256) void cgg_point_sample_nhwc_x3_kernel(const float* __restrict__ feat, const float* __restrict__ pts,
                                                                       u32x4* __restrict__ hi, u32x4* __restrict__ lo, int KC, int H,
                                                                       int W, int P_total, int P_group, int B) {
  extern __shared__ __attribute__((aligned(16))) u32x4 tile[];   // hi [KC][32] | lo [KC][32]
  const int t_all = blockIdx.x, b = blockIdx.y;
  const int n = KC * 32;
  const size_t C = (size_t)KC * 8;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    const int pl = idx / KC, kc = idx - pl * KC;
    const long long bp = (long long)b * P_total + (long long)t_all * 32 + pl;
    const float px = pts[bp * 2], py = pts[bp * 2 + 1];
    const float gx = __fsub_rn(__fmul_rn(px, 2.0f), 1.0f), gy = __fsub_rn(__fmul_rn(py, 2.0f), 1.0f);
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), 1.f), 2.f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), 1.f), 2.f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = __fsub_rn(ix, fx), ty = __fsub_rn(iy, fy);
    const float ux = __fsub_rn(__fadd_rn(fx, 1.f), ix), uy = __fsub_rn(__fadd_rn(fy, 1.f), iy);
    const float wnw = __fmul_rn(ux, uy), wne = __fmul_rn(tx, uy), wsw = __fmul_rn(ux, ty), wse = __fmul_rn(tx, ty);
    const float* fb = feat + (size_t)b * H * W * C + kc * 8;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
    auto tap = [&](int yy, int xx, float w) {
      const float* s0 = fb + ((size_t)yy * W + xx) * C;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(s0), v1 = *reinterpret_cast<const f32x4*>(s0 + 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a0[c] = fmaf(v0[c], w, a0[c]);
        a1[c] = fmaf(v1[c], w, a1[c]);
      }
    };
    if (xin0 && yin0) tap(y0, x0, wnw);
    if (xin1 && yin0) tap(y0, x0 + 1, wne);
    if (xin0 && yin1) tap(y0 + 1, x0, wsw);
    if (xin1 && yin1) tap(y0 + 1, x0 + 1, wse);
    u32x4 ph, plo;
    cgg_x3_split8(a0, a1, ph, plo);
    tile[kc * 32 + pl] = ph;
    tile[n + kc * 32 + pl] = plo;
  }
  __syncthreads();
  const int Tg = P_group >> 5;                                   // tiles per (layer, image)
  const int grp = t_all / Tg, t = t_all - grp * Tg;
  const size_t img = (size_t)grp * B + b;                        // layer-major: the B images of a layer are one PackedFeature
  u32x4* dh = hi + (img * Tg + t) * n;
  u32x4* dl = lo + (img * Tg + t) * n;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    dh[idx] = tile[idx];
    dl[idx] = tile[n + idx];
  }
}

extern "C" int cgg_point_sample_nhwc_x3(const float* feat, const float* pts, void* hi, void* lo, int B, int H, int W, int C,
                                        int P_total, int P_group, cgg_stream_t stream) {
  CGG_REQUIRE(feat && pts && hi && lo, CGG_EINVAL, "cgg_point_sample_nhwc_x3: null pointer");
  CGG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && P_total > 0 && P_group > 0, CGG_EINVAL, "cgg_point_sample_nhwc_x3: bad sizes");
  CGG_REQUIRE(C % 8 == 0 && C <= 1024, CGG_EUNSUPPORTED, "cgg_point_sample_nhwc_x3: C=%d", C);
  CGG_REQUIRE(P_group % 32 == 0 && P_total % P_group == 0, CGG_EUNSUPPORTED,
              "cgg_point_sample_nhwc_x3: P_group=%d must be a multiple of 32 dividing P_total=%d", P_group, P_total);
  CGG_REQUIRE(cgg_aligned16(feat) && cgg_aligned16(hi) && cgg_aligned16(lo), CGG_EALIGN, "cgg_point_sample_nhwc_x3: alignment");
  const int KC = C / 8;
  hipLaunchKernelGGL(cgg_point_sample_nhwc_x3_kernel, dim3(P_total / 32, B), dim3(256), (size_t)KC * 32 * 16 * 2, (hipStream_t)stream,
                     feat, pts, (u32x4*)hi, (u32x4*)lo, KC, H, W, P_total, P_group, B);
  CGG_CHECK_LAUNCH("cgg_point_sample_nhwc_x3");
  return CGG_OK;
}

// -------------------------------------------------------------------------------------------------
// mask logits. Workgroup = 8 waves sharing one image's mask_embed in LDS; each wave streams its own
// 32-pixel tiles: 16 k-steps x ceil(Q/32) m-tiles of v_mfma_f32_32x32x16_bf16 (x3 in SPLIT mode).
//   A (LDS, fragment order): slot (mt, ks, lane) holds E[q = mt*32 + (lane&31)][k = ks*16 + 8*(lane>>5) ..+7]
//   B (global, packed):      slot (t, ks, lane)  holds F[k = ks*16 + 8*(lane>>5) ..+7][p = t*32 + (lane&31)]
//   D: acc[r] = logit[q = mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)][p = t*32 + (lane&31)]
// Instruction diet: the first version unrolled all m-tiles (4 x 16 accumulators, 64 predicated stores with
// 64-bit address math) = ~5.5k instructions and 40 KB of code per wave; at 2 waves/SIMD it was ISSUE-bound at
// 2.5 TB/s whatever the memory system did. Here the B fragments of a tile stay in registers and the m-tile
// loop is ROLLED (one 16-register accumulator, 16 MFMA + 16 ds_read_b128 + 16 stores per trip), every global
// access is "uniform base + 32-bit lane offset", and row validity (q < Q) is a wave-uniform scalar compare.
// -------------------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(512, 2) void cgg_mask_logits_kernel(
    const float* __restrict__ embed, const u32x4* __restrict__ fhi, const u32x4* __restrict__ flo,
    float* __restrict__ out, uint32_t* __restrict__ bits, int Q, int npix, int T, int MT, int q_total, int G) {
  // G row groups of <= 128 queries in ONE launch (split mode keeps <= 4 query tiles = 128 KiB of hi + lo fragments in LDS): workgroup
  // x = tile_slot * G + group, so the G workgroups that stream the same feature tiles are dispatched together and the second read of a
  // tile is an Infinity-Cache hit; every group writes its rows q0 .. q0 + Q - 1 of the q_total-row outputs in place.
  const int grp = G > 1 ? (int)(blockIdx.x % (unsigned)G) : 0;
  const int bx = G > 1 ? (int)(blockIdx.x / (unsigned)G) : (int)blockIdx.x;
  const int q0 = grp * 128;
  if (G > 1) {
    Q = q_total - q0 < 128 ? q_total - q0 : 128;
    MT = (Q + 31) / 32;
  }
  constexpr int KS = 16;  // C = 256
  constexpr int C = KS * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* a_hi = reinterpret_cast<u32x4*>(smem_raw);
  u32x4* a_lo = a_hi + MT * KS * 64;      // this group's own MT (the launch sized the LDS for the largest group)

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const int hi5 = lane >> 5;
  const int col = lane & 31;
  const int tstride = (int)(gridDim.x / (unsigned)G) * 8;
  float* __restrict__ ob = out ? out + ((size_t)b * q_total + q0) * npix : nullptr;       // uniform
  uint32_t* __restrict__ bb = bits ? bits + ((size_t)b * q_total + q0) * T : nullptr;     // uniform
  const u32x4* __restrict__ fhb = fhi + (size_t)b * T * (KS * 64) + lane;
  const u32x4* __restrict__ flb = SPLIT ? flo + (size_t)b * T * (KS * 64) + lane : nullptr;
  int t = bx * 8 + wave;

  // ---- (0) request the first tile's B fragments BEFORE the prologue: their HBM latency hides under it ----
  u32x4 bh[KS];
  u32x4 bl[SPLIT ? KS : 1];
  u32x4 bn[SPLIT ? 1 : KS];   // bf16 mode: second register set, next tile in flight while this one computes
  if (t < T) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bh[ks] = __builtin_nontemporal_load(fhb + (size_t)t * (KS * 64) + ks * 64);
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bl[ks] = __builtin_nontemporal_load(flb + (size_t)t * (KS * 64) + ks * 64);
    }
  }

  // ---- (1) prologue: mask_embed[b] -> bf16 fragments in LDS ----
  // thread owns float4 number tid, tid+512, ... of the contiguous [Q, 256] block (coalesced); float4 (q, c4)
  // lands in slot (mt = q/32, ks = c4/4, lane = q%32 + 32*((c4/2)&1)), half (c4 & 1). Rows >= Q are zero.
  {
    const f32x4* __restrict__ eb4 = reinterpret_cast<const f32x4*>(embed + ((size_t)b * q_total + q0) * C);
    uint2* a_hi2 = reinterpret_cast<uint2*>(a_hi);
    uint2* a_lo2 = reinterpret_cast<uint2*>(a_lo);
    const int nf = MT * 32 * (C / 4);
    const int nvalid = Q * (C / 4);
    for (int f0 = tid; f0 < nf; f0 += 512 * 4) {
      f32x4 ev[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + 512 * u;
        ev[u] = (f < nvalid) ? eb4[f] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + 512 * u;
        if (f < nf) {
          const int q = f >> 6, c4 = f & 63;
          const int slot = ((q >> 5) * KS + (c4 >> 2)) * 64 + (q & 31) + 32 * ((c4 >> 1) & 1);
          uint16_t h[4], lw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (SPLIT) cgg_x3_split1(ev[u][e], h[e], lw[e]);
            else h[e] = cgg_f2bf(ev[u][e]);
          }
          a_hi2[slot * 2 + (c4 & 1)] = make_uint2(cgg_pack2(h[0], h[1]), cgg_pack2(h[2], h[3]));
          if (SPLIT) a_lo2[slot * 2 + (c4 & 1)] = make_uint2(cgg_pack2(lw[0], lw[1]), cgg_pack2(lw[2], lw[3]));
        }
      }
    }
  }
  __syncthreads();

  // ---- (2) stream the tiles ----
  // one tile: MT x (16 MFMA + 16 ds_read_b128 + 16 row stores); `cur` holds the tile's B fragments
  auto do_tile = [&](const u32x4 (&cur)[KS], int tt) {
    const bool ragged = (tt * 32 + 32 > npix);                               // wave-uniform, last tile only
    const bool pin = tt * 32 + col < npix;
    const unsigned lane_off = (unsigned)(4 * hi5) * (unsigned)npix + (unsigned)(tt * 32 + col);
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const u32x4* __restrict__ ah = a_hi + mt * (KS * 64) + lane;
      const u32x4* __restrict__ al = a_lo + mt * (KS * 64) + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if constexpr (SPLIT) {
          cgg_x3_mfma(acc, ah[ks * 64], al[ks * 64], cur[ks], bl[ks]);
        } else {
          const bf16x8 vbh = __builtin_bit_cast(bf16x8, cur[ks]);
          const bf16x8 vah = __builtin_bit_cast(bf16x8, ah[ks * 64]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, vbh, acc, 0, 0, 0);
        }
      }
      if constexpr (SPLIT) acc = acc * (CGG_X3_INV_ASCALE * CGG_X3_INV_ASCALE);      // both operands carry the 2^4 pre-scale
      const int rows_left = Q - mt * 32;                                     // uniform; >= 32 for full m-tiles
      if (ob != nullptr) {
        float* __restrict__ tile_row = ob + (size_t)(mt * 32) * npix;        // uniform
        if (!ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ql = (r & 3) + 8 * (r >> 2);                           // compile-time local row (low half)
            float* __restrict__ row = tile_row + (size_t)ql * npix;
            // streamed output: nontemporal stores (+10 % on MI355X: 27.0 -> 24.5 us at configs[1])
            if (ql + 4 < rows_left) __builtin_nontemporal_store(acc[r], row + lane_off);
            else if (ql < rows_left) { if (hi5 == 0) __builtin_nontemporal_store(acc[r], row + lane_off); }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ql = (r & 3) + 8 * (r >> 2) + 4 * hi5;
            if (ql < rows_left && pin) tile_row[(size_t)ql * npix + tt * 32 + col] = acc[r];
          }
        }
      }
      if (bb != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned long long m = __ballot(acc[r] < 0.f);
          const int ql = (r & 3) + 8 * (r >> 2) + 4 * hi5;                   // lanes 0-31: row ql, 32-63: ql + 4
          const uint32_t w = hi5 ? (uint32_t)(m >> 32) : (uint32_t)m;
          if (col == 0 && ql < rows_left) bb[(size_t)(mt * 32 + ql) * T + tt] = w;
        }
      }
    }
  };

  if constexpr (SPLIT) {
    while (t < T) {
      do_tile(bh, t);
      t += tstride;
      if (t < T) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bh[ks] = __builtin_nontemporal_load(fhb + (size_t)t * (KS * 64) + ks * 64);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bl[ks] = __builtin_nontemporal_load(flb + (size_t)t * (KS * 64) + ks * 64);
      }
    }
  } else {
    while (t < T) {
      const int tn = t + tstride;
      if (tn < T) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bn[ks] = __builtin_nontemporal_load(fhb + (size_t)tn * (KS * 64) + ks * 64);
      }
      do_tile(bh, t);
      if (tn >= T) break;
      const int tnn = tn + tstride;
      if (tnn < T) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bh[ks] = __builtin_nontemporal_load(fhb + (size_t)tnn * (KS * 64) + ks * 64);
      }
      do_tile(bn, tn);
      t = tnn;
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

extern "C" int cgg_pack_mask_feature_nhwc_multi(const void* feat, void* const* hi_host, const int* pools_host, int n,
                                                int B, int C, int H, int W, cgg_stream_t stream) {
  CGG_REQUIRE(feat && hi_host && pools_host, CGG_EINVAL, "cgg_pack_mask_feature_nhwc: null pointer");
  CGG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && n >= 1 && n <= 4, CGG_EINVAL, "cgg_pack_mask_feature_nhwc: bad sizes");
  CGG_REQUIRE(C % 8 == 0 && C <= 1024, CGG_EUNSUPPORTED, "cgg_pack_mask_feature_nhwc: C=%d", C);
  CGG_REQUIRE(cgg_aligned16(feat), CGG_EALIGN, "cgg_pack_mask_feature_nhwc: alignment");
  PackJobs jobs;
  jobs.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const int pool = pools_host[i];
    CGG_REQUIRE(pool == 1 || (pool >= 2 && pool % 2 == 0 && H % pool == 0 && W % pool == 0), CGG_EUNSUPPORTED,
                "cgg_pack_mask_feature_nhwc: pool=%d must be 1 or an even divisor of %dx%d", pool, H, W);
    CGG_REQUIRE(hi_host[i] && cgg_aligned16(hi_host[i]), CGG_EALIGN, "cgg_pack_mask_feature_nhwc: output %d", i);
    const int Hp = H / pool, Wp = W / pool;
    jobs.pool[i] = pool;
    jobs.Wp[i] = Wp;
    jobs.npix[i] = Hp * Wp;
    jobs.T[i] = (Hp * Wp + 31) / 32;
    jobs.t0[i] = total;
    jobs.hi[i] = (u32x4*)hi_host[i];
    total += jobs.T[i];
  }
  const int KC = C / 8;
  hipLaunchKernelGGL(cgg_pack_nhwc_kernel, dim3(total, B), dim3(256), (size_t)KC * 32 * 16, (hipStream_t)stream,
                     (const uint4*)feat, jobs, KC, H, W);
  CGG_CHECK_LAUNCH("cgg_pack_mask_feature_nhwc");
  return CGG_OK;
}

extern "C" int cgg_pack_mask_feature_nhwc_f32_x3(const float* feat, void* const* hi_host, void* const* lo_host,
                                                  const int* pools_host, int n, int B, int C, int H, int W, cgg_stream_t stream) {
  CGG_REQUIRE(feat && hi_host && lo_host && pools_host, CGG_EINVAL, "cgg_pack_mask_feature_nhwc_f32_x3: null pointer");
  CGG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && n >= 1 && n <= 4, CGG_EINVAL, "cgg_pack_mask_feature_nhwc_f32_x3: bad sizes");
  CGG_REQUIRE(C % 8 == 0 && C <= 1024, CGG_EUNSUPPORTED, "cgg_pack_mask_feature_nhwc_f32_x3: C=%d", C);
  CGG_REQUIRE(cgg_aligned16(feat), CGG_EALIGN, "cgg_pack_mask_feature_nhwc_f32_x3: alignment");
  PackJobsX3 jobs;
  jobs.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const int pool = pools_host[i];
    CGG_REQUIRE(pool == 1 || (pool >= 2 && pool % 2 == 0 && H % pool == 0 && W % pool == 0), CGG_EUNSUPPORTED,
                "cgg_pack_mask_feature_nhwc_f32_x3: pool=%d must be 1 or an even divisor of %dx%d", pool, H, W);
    CGG_REQUIRE(hi_host[i] && lo_host[i] && cgg_aligned16(hi_host[i]) && cgg_aligned16(lo_host[i]), CGG_EALIGN,
                "cgg_pack_mask_feature_nhwc_f32_x3: output %d", i);
    const int Hp = H / pool, Wp = W / pool;
    jobs.pool[i] = pool;
    jobs.Wp[i] = Wp;
    jobs.npix[i] = Hp * Wp;
    jobs.T[i] = (Hp * Wp + 31) / 32;
    jobs.t0[i] = total;
    jobs.hi[i] = (u32x4*)hi_host[i];
    jobs.lo[i] = (u32x4*)lo_host[i];
    total += jobs.T[i];
  }
  const int KC = C / 8;
  hipLaunchKernelGGL(cgg_pack_nhwc_f32_x3_kernel, dim3(total, B), dim3(256), (size_t)KC * 32 * 16 * 2, (hipStream_t)stream, feat,
                     jobs, KC, H, W);
  CGG_CHECK_LAUNCH("cgg_pack_mask_feature_nhwc_f32_x3");
  return CGG_OK;
}

extern "C" int cgg_pack_mask_feature_nhwc(const void* feat, void* hi, int B, int C, int H, int W, int pool,
                                          cgg_stream_t stream) {
  CGG_REQUIRE(pool >= 1, CGG_EINVAL, "cgg_pack_mask_feature_nhwc: bad sizes");
  void* his[1] = {hi};
  return cgg_pack_mask_feature_nhwc_multi(feat, his, &pool, 1, B, C, H, W, stream);
}

template <bool SPLIT>
static int launch_mask_logits(const float* embed, const void* hi, const void* lo, float* out,
                              uint32_t* bits, int B, int Q, int npix, hipStream_t s, int q_total, int G) {
  // G > 1: Q is the largest group's row count (128)
  const int MT = (Q + 31) / 32;
  const int T = (npix + 31) / 32;
  const size_t lds = (size_t)MT * 16 * 64 * 16 * (SPLIT ? 2 : 1);
  // ~one 8-wave workgroup per CU (two tiles per wave at 1024x1024); every wave gets >= 1 tile where possible
  int gx = (T + 7) / 8;
  int cap = (256 + B - 1) / B;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  auto kern = cgg_mask_logits_kernel<SPLIT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      cgg_set_error("cgg_mask_logits: cannot raise dynamic LDS to %zu: %s", lds,
                    hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3(gx * G, B), dim3(512), lds, s, embed, (const u32x4*)hi,
                     (const u32x4*)lo, out, bits, Q, npix, T, MT, q_total, G);
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
    // split mode keeps <= 4 query tiles (hi + lo fragment images, 128 KiB) in LDS: larger query sets run as row groups of 128
    // inside ONE launch (workgroup x = tile slot * G + group), every group writing its rows of the outputs in place
    const int G = (Q + 127) / 128;
    return launch_mask_logits<true>(embed, hi, lo, out, bits, B, G > 1 ? 128 : Q, npix, s, Q, G);
  }
  CGG_REQUIRE(mt <= 8, CGG_EUNSUPPORTED, "cgg_mask_logits: Q <= 256 (Q=%d)", Q);
  return launch_mask_logits<false>(embed, hi, lo, out, bits, B, Q, npix, s, Q, 1);
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
