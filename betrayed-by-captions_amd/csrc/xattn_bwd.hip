// K6 backward: gradients of the masked multi-head cross-attention core (cgg_masked_xattn_forward_lse) with respect to
// the projected queries and the projected [K | V] rows. This is the backward of the `baddbmm + masked softmax + bmm` inside
// nn.MultiheadAttention on the training path open_set/models/mask2former_head.py:829-840 (called from forward_train,
// :851-921). The reference materialises (B*8, Q, S) scores / probabilities / their gradients (26 MB x 3 per layer and
// image at S = 16 384); here nothing of size Q x S ever leaves the registers:
//
//   P  = exp(scale q k^T + mask - LSE)            (LSE saved by the forward: one float per (b, h, query))
//   dV = P^T dO        dP = dO V^T        dS = P o (dP - delta),   delta = rowsum(dO o O)
//   dQ = scale dS K    dK = scale dS^T Q
//
// Work split: one workgroup = (key chunk, head, image); its 4 wavefronts each own 32 KEYS of every 128-key tile and loop
// over the <= 4 query tiles. Keys never leave their wavefront, so dK / dV accumulate in registers and are stored once
// per key (no atomics, deterministic); dQ is accumulated per wavefront over its keys and written as one partial plane
// per (chunk, wave) that a small second kernel sums in a fixed order.
//
// MFMA orientation: S[query][key] with QUERIES on the MFMA rows and KEYS on the lanes (the transpose of the forward
// kernel): P and dS are then directly the B operands of dV^T = dO^T P and dK^T = Q^T dS (contraction over queries), K / V
// rows are loaded straight from global memory into B-operand registers (64 contiguous bytes per lane), and only dQ
// (contraction over keys) needs dS transposed -- through a per-wave 32 x 32 LDS tile. v_mfma_f32_32x32x2_f32 throughout:
// exact f32 products, f32 accumulation (training / parity precision).
#include "x3.h"

#define XB_LD 36   // LDS row stride in floats: 16-byte aligned rows, rows skewed by 4 banks

__device__ __forceinline__ int xb_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// BF (throughput-mode training, kv_dtype CGG_F32_BF16MFMA): every group of four 32x32x2 f32 steps -- a lane's four consecutive k
// values, which is also how the accumulator rows are grouped (xb_row(4 g + e, hi) = 8 g + 4 hi + e) -- becomes one
// v_mfma_f32_32x32x8_bf16 on the converted 4-vectors; 80 -> 20 MFMAs per (32 queries x 32 keys) at 8 x the rate, layouts unchanged.
typedef __attribute__((ext_vector_type(4))) short xb_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 xb_bf2;
typedef __attribute__((ext_vector_type(2))) float xb_f2;
__device__ __forceinline__ xb_s16x4 xb_cvt4(float a, float b, float c, float d) {
  const xb_f2 lo = {a, b}, hi = {c, d};
  const uint2 u = {__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, xb_bf2)),
                   __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, xb_bf2))};
  return __builtin_bit_cast(xb_s16x4, u);
}

// MODE 2 (round 6, parity-mode training): the same grouping on the f32-class f16 x 3 contraction of x3.h -- every 4-vector operand is
// split into two f16 pieces of (power-of-two scale) x value and the group becomes THREE v_mfma_f32_32x32x8_f16 (al bh + ah bl + ah bh):
// 60 MFMAs of 32 cycles per (32 queries x 32 keys) instead of 80 f32 MFMAs of 64. Pre-scales: 2^4 for scale q, K, V (unit-scale
// rows, like every x3 activation operand); s_g = 2^(9 - floor(log2 max |dO|)) for dO (a gradient: per-tensor scale from a device
// scalar, x3.h "per-tensor pre-scale"); 2^14 for P (<= 1); s_g / 64 for dS = P (dP - delta) (|dS| <= 4000 max |dO| keeps it inside
// f16). All are powers of two: the accumulators are un-scaled exactly (scores before the exponential, gradients when stored).
typedef __attribute__((ext_vector_type(4))) _Float16 xb_h4;
struct XbX3 {
  xb_h4 h, l;
};
// (plain conversions, not x3.h's inline-asm v_fma_mix form: these pieces feed MFMAs straight from registers, and the hazard recogniser
// does not see a VALU write inside an asm block -- the asm form gave stale operands in some lanes; in the GEMM kernels the pieces go
// through LDS first)
__device__ __forceinline__ XbX3 xb_split4(float a, float b, float c, float d, float s) {
  const f32x4 v = f32x4{a, b, c, d} * s;
  XbX3 r;
  r.h = __builtin_convertvector(v, xb_h4);                                  // RNE
  r.l = __builtin_convertvector(v - __builtin_convertvector(r.h, f32x4), xb_h4);   // the residual is exact in f32
  return r;
}
__device__ __forceinline__ f32x16 xb_mfma3(const XbX3& a, const XbX3& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a.h, b.l, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x8f16(a.h, b.h, acc, 0, 0, 0);
}

// MODE 0: f32 MFMA, 1: bf16 MFMA operands (BF), 2: f16 x 3
template <int MODE>
__global__ __launch_bounds__(256) void cgg_xattn_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ kv, const uint32_t* __restrict__ bits,
    const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ gout,
    float* __restrict__ gkv, float* __restrict__ ws_dq, int Q, int H, int S, int words, int KC, int nchunks,
    float scale, const float* __restrict__ gout_amax) {
  constexpr int D = 32;
  constexpr bool BF = MODE == 1, X3 = MODE == 2;
  // x3 pre-scales (see above); sg from the device scalar max |dO|
  const float sg = X3 ? cgg_x3_scale_from_amax(*gout_amax) : 1.f;
  constexpr float SA = CGG_X3_ASCALE, SP = 16384.f;
  const float ss = sg * (1.f / 64.f);
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int HD = H * D;
  const int nmt = (Q + 31) / 32;
  const int s_begin = chunk * KC, s_end = min(S, s_begin + KC);

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Qs = reinterpret_cast<float*>(smem_raw);      // [128][XB_LD]  scale * q
  float* Gs = Qs + 128 * XB_LD;                        // [128][XB_LD]  dO
  float* Ls = Gs + 128 * XB_LD;                        // [128] LSE
  float* Ds = Ls + 128;                                // [128] delta
  float* Kt = Ds + 128 + wave * (32 * XB_LD);          // per wave [32 keys][XB_LD]
  float* Tt = Ds + 128 + 4 * (32 * XB_LD) + wave * (32 * XB_LD);   // per wave dS tile [32 queries][XB_LD]

  // ---- queries, output gradients, LSE and delta of this (image, head) ----
  for (int i = tid; i < 128 * 8; i += 256) {
    const int qq = i >> 3, c4 = (i & 7) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
    if (qq < Q) {
      const size_t off = ((size_t)b * Q + qq) * HD + h * D + c4;
      a = *reinterpret_cast<const f32x4*>(q + off);
      g = *reinterpret_cast<const f32x4*>(gout + off);
      a = a * scale;
    }
    *reinterpret_cast<f32x4*>(Qs + qq * XB_LD + c4) = a;
    *reinterpret_cast<f32x4*>(Gs + qq * XB_LD + c4) = g;
  }
  if (tid < 128) {
    float l = 0.f, dl = 0.f;
    if (tid < Q) {
      l = lse[((size_t)b * H + h) * Q + tid];
      const size_t off = ((size_t)b * Q + tid) * HD + h * D;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(out + off + c);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gout + off + c);
        dl += o4[0] * g4[0] + o4[1] * g4[1] + o4[2] * g4[2] + o4[3] * g4[3];
      }
    }
    Ls[tid] = l;
    Ds[tid] = dl;
  }
  __syncthreads();

  f32x16 dq[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  const float* kvb = kv + (size_t)b * S * (2 * HD) + h * D;
  for (int s0 = s_begin; s0 < s_end; s0 += 128) {
    const int key = s0 + 32 * wave + j;
    if (s0 + 32 * wave >= s_end) break;                          // wave-uniform: this wave's 32 keys are past the chunk
    const bool kvalid = key < s_end;
    // ---- this lane's key: K / V [16 hi .. 16 hi + 15] as B operands of S = Q K^T and dP = dO V^T ----
    float kf[16], vf[16];
    {
      const float* row = kvb + (size_t)(kvalid ? key : s_begin) * (2 * HD) + 16 * hi;
      // all eight loads first: interleaved with the LDS stores below the compiler waited for every load separately
      f32x4 kxa[4], vxa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        kxa[t] = *reinterpret_cast<const f32x4*>(row + 4 * t);
        vxa[t] = *reinterpret_cast<const f32x4*>(row + HD + 4 * t);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 kx = kxa[t], vx = vxa[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kf[4 * t + e] = kvalid ? kx[e] : 0.f;
          vf[4 * t + e] = kvalid ? vx[e] : 0.f;
        }
        f32x4 kz = {kf[4 * t], kf[4 * t + 1], kf[4 * t + 2], kf[4 * t + 3]};
        *reinterpret_cast<f32x4*>(Kt + j * XB_LD + 16 * hi + 4 * t) = kz;     // K tile for the dQ contraction
      }
    }
    // x3: the key tile's operands split once for the <= 4 query tiles (K / V rows as B operands, K columns for the dQ contraction)
    XbX3 kx3[4], vx3[4], kc3[4];
    if constexpr (X3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        kx3[t] = xb_split4(kf[4 * t], kf[4 * t + 1], kf[4 * t + 2], kf[4 * t + 3], SA);
        vx3[t] = xb_split4(vf[4 * t], vf[4 * t + 1], vf[4 * t + 2], vf[4 * t + 3], SA);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k0 = 16 * hi + 4 * t;
        kc3[t] = xb_split4(Kt[k0 * XB_LD + j], Kt[(k0 + 1) * XB_LD + j], Kt[(k0 + 2) * XB_LD + j], Kt[(k0 + 3) * XB_LD + j], SA);
      }
    }
    const int wq = key >> 5, bq = key & 31;                      // mask word / bit of this lane's key
    f32x16 dkt, dvt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }

#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      if (qt >= nmt) break;                                       // wave-uniform
      const float* qrow = Qs + (qt * 32 + j) * XB_LD + 16 * hi;   // A operand rows: lane i = query
      const float* grow = Gs + (qt * 32 + j) * XB_LD + 16 * hi;
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
      // the 16 mask words this lane tests (one per query row of the tile; the 32 keys of a wave share a word) are requested
      // here, unpredicated (rows past Q re-read row Q - 1), and arrive under the 64 MFMAs below -- as `if (!blocked) blocked =
      // bits[...]` they compiled to 16 x (branch, load, s_waitcnt vmcnt(0)): 64 serial memory latencies per 128-key tile
      uint32_t mw[16];
      if (bits != nullptr) {
        const uint32_t* mrow = bits + (size_t)b * Q * words + wq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qt * 32 + xb_row(r, hi);
          mw[r] = mrow[(size_t)(qq < Q ? qq : Q - 1) * words];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(qrow + 4 * t);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(grow + 4 * t);
        if constexpr (X3) {
          sc = xb_mfma3(xb_split4(qa[0], qa[1], qa[2], qa[3], SA), kx3[t], sc);
          dp = xb_mfma3(xb_split4(ga[0], ga[1], ga[2], ga[3], sg), vx3[t], dp);
        } else if constexpr (BF) {
          sc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(xb_cvt4(qa[0], qa[1], qa[2], qa[3]),
                                                        xb_cvt4(kf[4 * t], kf[4 * t + 1], kf[4 * t + 2], kf[4 * t + 3]), sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(xb_cvt4(ga[0], ga[1], ga[2], ga[3]),
                                                        xb_cvt4(vf[4 * t], vf[4 * t + 1], vf[4 * t + 2], vf[4 * t + 3]), dp, 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[4 * t + e], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[e], vf[4 * t + e], dp, 0, 0, 0);
          }
        }
      }
      // ---- P and dS for (query xb_row(r, hi), this lane's key) ----
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qt * 32 + xb_row(r, hi);
        bool blocked = !kvalid || qq >= Q;
        if (bits != nullptr) blocked = blocked || ((mw[r] >> bq) & 1u);
        const float sv = X3 ? sc[r] * (1.f / (SA * SA)) : sc[r];                 // x3: exact un-scaling of the score ...
        const float dv = X3 ? dp[r] * (1.f / SA) / sg : dp[r];                   // ... and of dP (sg is a power of two)
        const float p = blocked ? 0.f : __expf(sv - Ls[qq]);
        sc[r] = p;                                   // P
        dp[r] = p * (dv - Ds[qq]);                   // dS
      }
      // ---- dV^T[d][key] += dO^T[d][query] P[query][key],  dK^T[d][key] += (scale Q)^T[d][query] dS[query][key] ----
      if constexpr (X3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q0 = qt * 32 + 8 * g + 4 * hi;       // the lane's four queries of k-step g = xb_row(4 g + e, hi)
          dvt = xb_mfma3(xb_split4(Gs[q0 * XB_LD + j], Gs[(q0 + 1) * XB_LD + j], Gs[(q0 + 2) * XB_LD + j], Gs[(q0 + 3) * XB_LD + j], sg),
                         xb_split4(sc[4 * g], sc[4 * g + 1], sc[4 * g + 2], sc[4 * g + 3], SP), dvt);
          dkt = xb_mfma3(xb_split4(Qs[q0 * XB_LD + j], Qs[(q0 + 1) * XB_LD + j], Qs[(q0 + 2) * XB_LD + j], Qs[(q0 + 3) * XB_LD + j], SA),
                         xb_split4(dp[4 * g], dp[4 * g + 1], dp[4 * g + 2], dp[4 * g + 3], ss), dkt);
        }
      } else if constexpr (BF) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q0 = qt * 32 + 8 * g + 4 * hi;       // the lane's four queries of k-step g = xb_row(4 g + e, hi)
          dvt = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(
              xb_cvt4(Gs[q0 * XB_LD + j], Gs[(q0 + 1) * XB_LD + j], Gs[(q0 + 2) * XB_LD + j], Gs[(q0 + 3) * XB_LD + j]),
              xb_cvt4(sc[4 * g], sc[4 * g + 1], sc[4 * g + 2], sc[4 * g + 3]), dvt, 0, 0, 0);
          dkt = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(
              xb_cvt4(Qs[q0 * XB_LD + j], Qs[(q0 + 1) * XB_LD + j], Qs[(q0 + 2) * XB_LD + j], Qs[(q0 + 3) * XB_LD + j]),
              xb_cvt4(dp[4 * g], dp[4 * g + 1], dp[4 * g + 2], dp[4 * g + 3]), dkt, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qt * 32 + xb_row(r, hi);
          dvt = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[qq * XB_LD + j], sc[r], dvt, 0, 0, 0);
          dkt = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qq * XB_LD + j], dp[r], dkt, 0, 0, 0);
        }
      }
      // ---- dQ[query][d] += dS[query][key] K[key][d]: dS through the per-wave LDS tile as A operand ----
#pragma unroll
      for (int r = 0; r < 16; ++r) Tt[xb_row(r, hi) * XB_LD + j] = dp[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      f32x16 acc = dq[qt];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 da = *reinterpret_cast<const f32x4*>(Tt + j * XB_LD + 16 * hi + 4 * t);
        if constexpr (X3) {
          acc = xb_mfma3(xb_split4(da[0], da[1], da[2], da[3], ss), kc3[t], acc);
        } else if constexpr (BF) {
          const int k0 = 16 * hi + 4 * t;
          acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(
              xb_cvt4(da[0], da[1], da[2], da[3]),
              xb_cvt4(Kt[k0 * XB_LD + j], Kt[(k0 + 1) * XB_LD + j], Kt[(k0 + 2) * XB_LD + j], Kt[(k0 + 3) * XB_LD + j]), acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], Kt[(16 * hi + 4 * t + e) * XB_LD + j], acc, 0, 0, 0);
        }
      }
      dq[qt] = acc;
      __builtin_amdgcn_wave_barrier();               // Tt is rewritten by the next query tile
    }
    // ---- dK / dV of this lane's key: regs r <-> d = xb_row(r, hi) (4 consecutive d per group) ----
    if (kvalid) {
      float* gk = gkv + ((size_t)b * S + key) * (2 * HD) + h * D + 4 * hi;
      const float uk = X3 ? 1.f / (SA * ss) : 1.f, uv = X3 ? 1.f / (sg * SP) : 1.f;          // exact: powers of two
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = f32x4{dkt[4 * g], dkt[4 * g + 1], dkt[4 * g + 2], dkt[4 * g + 3]} * uk;
        const f32x4 c = f32x4{dvt[4 * g], dvt[4 * g + 1], dvt[4 * g + 2], dvt[4 * g + 3]} * uv;
        *reinterpret_cast<f32x4*>(gk + 8 * g) = a;
        *reinterpret_cast<f32x4*>(gk + HD + 8 * g) = c;
      }
    }
  }
  // ---- dQ partial plane of this (chunk, wave): ws_dq[b][h][chunk * 4 + wave][q][d], lane j = d ----
  float* wp = ws_dq + ((((size_t)b * H + h) * nchunks + chunk) * 4 + wave) * (size_t)Q * D;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < nmt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = t * 32 + xb_row(r, hi);
        if (qq < Q) wp[(size_t)qq * D + j] = X3 ? dq[t][r] * (1.f / (SA * ss)) : dq[t][r];
      }
    }
  }
}

// grad_q[b][q][h*D + d] = scale * sum over the (chunk, wave) planes, fixed order
__global__ __launch_bounds__(256) void cgg_xattn_bwd_reduce_dq(const float* __restrict__ ws_dq, float* __restrict__ gq,
                                                               int B, int Q, int H, int planes, float scale) {
  constexpr int D = 32;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * Q * H * D) return;
  const int d = (int)(gid % D), h = (int)((gid / D) % H);
  const int qq = (int)((gid / ((long long)D * H)) % Q), b = (int)(gid / ((long long)D * H * Q));
  const float* p = ws_dq + ((size_t)b * H + h) * planes * (size_t)Q * D + (size_t)qq * D + d;
  float s = 0.f;
  for (int c = 0; c < planes; ++c) s += p[(size_t)c * Q * D];
  gq[((size_t)b * Q + qq) * (H * D) + h * D + d] = s * scale;
}

static void xattn_bwd_plan(int B, int H, int S, int* KC, int* nchunks) {
  // ~2 workgroups per CU; a chunk is a multiple of 128 keys (4 wavefronts x 32 keys)
  int want = (512 + B * H - 1) / (B * H);
  const int tiles = (S + 127) / 128;
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  const int tpc = (tiles + want - 1) / want;
  *KC = tpc * 128;
  *nchunks = (S + *KC - 1) / *KC;
}

extern "C" int64_t cgg_masked_xattn_backward_workspace_bytes(int B, int Q, int H, int D, int S) {
  if (B <= 0 || Q <= 0 || H <= 0 || D <= 0 || S <= 0) return 0;
  int KC, nch;
  xattn_bwd_plan(B, H, S, &KC, &nch);
  return (int64_t)B * H * nch * 4 * Q * D * (int64_t)sizeof(float);
}

static int xattn_bwd_launch(const float* q, const void* kv, const uint32_t* bits, const float* out, const float* lse,
                            const float* grad_out, const float* gout_amax, float* grad_q, void* grad_kv, void* ws, int B, int Q, int H,
                            int D, int S, float scale, int kv_dtype, cgg_stream_t stream) {
  CGG_REQUIRE(q && kv && out && lse && grad_out && grad_q && grad_kv && ws, CGG_EINVAL,
              "cgg_masked_xattn_backward: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && H > 0 && S > 0, CGG_EINVAL, "cgg_masked_xattn_backward: bad sizes");
  CGG_REQUIRE(D == 32, CGG_EUNSUPPORTED, "cgg_masked_xattn_backward: head dim %d (only 32 is built)", D);
  CGG_REQUIRE(Q <= 128, CGG_EUNSUPPORTED, "cgg_masked_xattn_backward: Q=%d > 128", Q);
  CGG_REQUIRE(kv_dtype == CGG_F32 || kv_dtype == CGG_F32_BF16MFMA || kv_dtype == CGG_F32_X3, CGG_EUNSUPPORTED,
              "cgg_masked_xattn_backward: kv dtype %d (f32 rows only)", kv_dtype);
  CGG_REQUIRE(kv_dtype != CGG_F32_X3 || gout_amax, CGG_EINVAL, "cgg_masked_xattn_backward_x3: null grad_out_amax");
  CGG_REQUIRE(cgg_aligned16(q) && cgg_aligned16(kv) && cgg_aligned16(out) && cgg_aligned16(grad_out) &&
                  cgg_aligned16(grad_q) && cgg_aligned16(grad_kv) && cgg_aligned16(ws),
              CGG_EALIGN, "cgg_masked_xattn_backward: all tensors must be 16-B aligned");
  int KC, nch;
  xattn_bwd_plan(B, H, S, &KC, &nch);
  const int words = (S + 31) / 32;
  const size_t lds = (size_t)(2 * 128 * XB_LD + 256 + 8 * 32 * XB_LD) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cgg_xattn_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cgg_xattn_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cgg_xattn_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  if (kv_dtype == CGG_F32_X3)
    hipLaunchKernelGGL(cgg_xattn_bwd_kernel<2>, dim3(nch, H, B), dim3(256), lds, s, q, (const float*)kv, bits, out, lse, grad_out,
                       (float*)grad_kv, (float*)ws, Q, H, S, words, KC, nch, scale, gout_amax);
  else if (kv_dtype == CGG_F32_BF16MFMA)
    hipLaunchKernelGGL(cgg_xattn_bwd_kernel<1>, dim3(nch, H, B), dim3(256), lds, s, q, (const float*)kv, bits, out, lse, grad_out,
                       (float*)grad_kv, (float*)ws, Q, H, S, words, KC, nch, scale, gout_amax);
  else
    hipLaunchKernelGGL(cgg_xattn_bwd_kernel<0>, dim3(nch, H, B), dim3(256), lds, s, q, (const float*)kv, bits, out, lse, grad_out,
                       (float*)grad_kv, (float*)ws, Q, H, S, words, KC, nch, scale, gout_amax);
  CGG_CHECK_LAUNCH("cgg_masked_xattn_backward(main)");
  const long long total = (long long)B * Q * H * D;
  hipLaunchKernelGGL(cgg_xattn_bwd_reduce_dq, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)ws,
                     grad_q, B, Q, H, nch * 4, scale);
  CGG_CHECK_LAUNCH("cgg_masked_xattn_backward(reduce)");
  return CGG_OK;
}

extern "C" int cgg_masked_xattn_backward(const float* q, const void* kv, const uint32_t* bits, const float* out,
                                         const float* lse, const float* grad_out, float* grad_q, void* grad_kv, void* ws,
                                         int B, int Q, int H, int D, int S, float scale, int kv_dtype,
                                         cgg_stream_t stream) {
  CGG_REQUIRE(kv_dtype != CGG_F32_X3, CGG_EUNSUPPORTED, "cgg_masked_xattn_backward: the f16 x 3 form needs max |grad_out| -- "
              "cgg_masked_xattn_backward_x3");
  return xattn_bwd_launch(q, kv, bits, out, lse, grad_out, nullptr, grad_q, grad_kv, ws, B, Q, H, D, S, scale, kv_dtype, stream);
}

// ... on the f32-class f16 x 3 contraction (kernel MODE 2 above): grad_out_amax = device scalar max |grad_out| (cgg_absmax_f32)
extern "C" int cgg_masked_xattn_backward_x3(const float* q, const void* kv, const uint32_t* bits, const float* out, const float* lse,
                                            const float* grad_out, const float* grad_out_amax, float* grad_q, void* grad_kv, void* ws,
                                            int B, int Q, int H, int D, int S, float scale, cgg_stream_t stream) {
  return xattn_bwd_launch(q, kv, bits, out, lse, grad_out, grad_out_amax, grad_q, grad_kv, ws, B, Q, H, D, S, scale, CGG_F32_X3, stream);
}
