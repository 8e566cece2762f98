// Backward of K3, mask_pred = einsum('bqc,bchw->bqhw', mask_embed, mask_feature) (open_set/models/mask2former_head.py:748,
// differentiated by autograd on the training path :851-921): the two transposed contractions on v_mfma_f32_32x32x16_bf16,
//
//   grad_feat [b][c][p] = sum_q embed[b][q][c] * grad_out[b][q][p]        (C x npix output, K = Q: a STREAM over pixels)
//   grad_embed[b][q][c] = sum_p grad_out[b][q][p] * feat[b][c][p]         (Q x C output, K = npix: split over pixel chunks)
//
// In this build the training loss evaluates the einsum only for the MATCHED queries of all decoder layers (LazyMasks:
// Q = sum of positives, <= 256 rows per image), so these are the kernels behind `loss_mask` / `loss_dice`.
// SPLIT = true keeps f32-class accuracy with 3 MFMAs on (hi, lo) bf16 pairs (parity mode, Q <= 128 per launch).
//
// grad_feat: the structure of the forward kernel with the roles swapped -- embed^T sits in LDS as A fragments (bf16), a wave
//   owns a 32-pixel tile, loads its grad_out columns once as B fragments (8 row-coalesced dword loads per k-step), and
//   emits the 8 channel tiles as full 128-byte rows.
// grad_embed: a wave owns 32 channels and ALL query tiles (accumulators in registers); per 16-pixel k-step it reads 32
//   bytes of its feature row and of each query row straight from global (the 8 waves of a workgroup share the grad_out
//   lines through L1); per-chunk partial planes are summed in a fixed order by a second kernel (no atomics).
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t mlb_u32x4;

__device__ __forceinline__ int mlb_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <bool SPLIT>
__device__ __forceinline__ void mlb_pack8(const float (&v)[8], mlb_u32x4& h, mlb_u32x4& l) {
  uint16_t a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (SPLIT) cgg_split_bf(v[e], a[e], b[e]);
    else a[e] = cgg_f2bf(v[e]);
  }
  h = mlb_u32x4{cgg_pack2(a[0], a[1]), cgg_pack2(a[2], a[3]), cgg_pack2(a[4], a[5]), cgg_pack2(a[6], a[7])};
  if (SPLIT) l = mlb_u32x4{cgg_pack2(b[0], b[1]), cgg_pack2(b[2], b[3]), cgg_pack2(b[4], b[5]), cgg_pack2(b[6], b[7])};
}

// ---------------------------------------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(512) void cgg_mask_logits_grad_feat_kernel(const float* __restrict__ embed,
                                                                        const float* __restrict__ gout,
                                                                        float* __restrict__ gfeat, int Q, int npix, int T,
                                                                        int KS, int q0, int Qtot, int accumulate) {
  constexpr int C = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  mlb_u32x4* a_hi = reinterpret_cast<mlb_u32x4*>(smem_raw);            // [8 ct][KS][64]
  mlb_u32x4* a_lo = a_hi + 8 * KS * 64;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  // ---- prologue: embed^T[b] -> A fragments: slot (ct, ks, lane) = E[q = 16 ks + 8 hi + e][c = 32 ct + j], e = 0..7 ----
  // rows [q0, q0 + Q) of the (B, Qtot, .) operands; accumulate: a later row group adds into the first group's result
  const float* eb = embed + ((size_t)b * Qtot + q0) * C;
  for (int s = tid; s < 8 * KS * 64; s += 512) {
    const int sl = s & 63, ks = (s >> 6) % KS, ct = (s >> 6) / KS;
    const int c = 32 * ct + (sl & 31), q0 = 16 * ks + 8 * (sl >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (q0 + e < Q) ? eb[(size_t)(q0 + e) * C + c] : 0.f;
    mlb_u32x4 h, l;
    mlb_pack8<SPLIT>(v, h, l);
    a_hi[s] = h;
    if (SPLIT) a_lo[s] = l;
  }
  __syncthreads();
  const float* gb = gout + ((size_t)b * Qtot + q0) * npix;
  float* ob = gfeat + (size_t)b * C * npix;
  for (int t = blockIdx.x * 8 + wave; t < T; t += gridDim.x * 8) {
    const int p = 32 * t + j;
    const bool pin = p < npix;
    const int pc = pin ? p : npix - 1;
    mlb_u32x4 bh[16], bl[SPLIT ? 8 : 1];
    // the loads of 8 k-steps (64 values per lane) are issued back to back (clamped addresses, no predication: a predicated
    // load per element serialises on its own wait); out-of-range rows / pixels are zeroed by selects afterwards
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (8 * half < KS) {                                                   // workgroup-uniform
        float v[8][8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            // address = UNIFORM row base (scalar registers) + one 32-bit lane offset: 64 loads share 2 offset registers
            const int qlo = 16 * (8 * half + k8) + e;                          // row of the hi = 0 lanes (uniform)
            const float* rowp = gb + (size_t)min(qlo, Q - 1) * npix;
            const unsigned voff = (unsigned)pc + ((qlo + 8 < Q) ? (unsigned)hi * (unsigned)(8 * npix) : 0u);
            v[k8][e] = __builtin_nontemporal_load(rowp + voff);
          }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const int ks = 8 * half + k8;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[k8][e] = (pin && 16 * ks + 8 * hi + e < Q) ? v[k8][e] : 0.f;
          mlb_u32x4 l;
          mlb_pack8<SPLIT>(v[k8], bh[ks], l);
          if (SPLIT) bl[ks & 7] = l;
        }
      }
    }
#pragma unroll 1
    for (int ct = 0; ct < 8; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const mlb_u32x4* ah = a_hi + (ct * KS) * 64 + lane;
      const mlb_u32x4* al = a_lo + (ct * KS) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks < KS) {
          const bf16x8 va = __builtin_bit_cast(bf16x8, ah[ks * 64]);
          const bf16x8 vb = __builtin_bit_cast(bf16x8, bh[ks]);
          if constexpr (SPLIT) {
            const bf16x8 val = __builtin_bit_cast(bf16x8, al[ks * 64]);
            const bf16x8 vbl = __builtin_bit_cast(bf16x8, bl[ks & 7]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, vb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vbl, acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
        }
      }
      if (pin) {
        float* crow = ob + (size_t)(32 * ct) * npix;                           // uniform
        const unsigned soff = (unsigned)(4 * hi) * (unsigned)npix + (unsigned)p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* dst = crow + (size_t)((r & 3) + 8 * (r >> 2)) * npix + soff;
          __builtin_nontemporal_store(accumulate ? *dst + acc[r] : acc[r], dst);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <bool SPLIT, int MT>
__global__ __launch_bounds__(512) void cgg_mask_logits_grad_embed_kernel(const float* __restrict__ gout,
                                                                         const float* __restrict__ feat,
                                                                         float* __restrict__ ws, int Q, int q0, int Qtot,
                                                                         int npix, int steps_per_chunk, int nchunks) {
  constexpr int C = 256;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int nsteps = (npix + 15) / 16;
  const int s_begin = chunk * steps_per_chunk, s_end = min(nsteps, s_begin + steps_per_chunk);
  const float* frow = feat + ((size_t)b * C + 32 * wave + j) * npix + 8 * hi;     // B operand: channel 32 wave + j
  const float* grow = gout + ((size_t)b * Qtot + q0) * npix + 8 * hi;              // A operand rows: + (32 mt + j) * npix
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  // rows of this lane (clamped: out-of-range rows are zeroed after the load, never predicated)
  const float* grows[MT];
  bool qok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int q = 32 * mt + j;
    qok[mt] = q < Q;
    grows[mt] = grow + (size_t)min(q, Q - 1) * npix;
  }
  const int plast = npix - 8;                              // last legal 8-pixel vector start (npix % 8 == 0)
  auto load_step = [&](int s, f32x4 (&fx)[2], f32x4 (&gx)[MT][2]) {
    const int po = max(0, min(16 * s, plast - 8 * hi));   // frow / grows already carry + 8 hi: keep po + 8 hi <= plast
    fx[0] = *reinterpret_cast<const f32x4*>(frow + po);
    fx[1] = *reinterpret_cast<const f32x4*>(frow + po + 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      gx[mt][0] = *reinterpret_cast<const f32x4*>(grows[mt] + po);
      gx[mt][1] = *reinterpret_cast<const f32x4*>(grows[mt] + po + 4);
    }
  };
  auto mma_step = [&](int s, const f32x4 (&fx)[2], const f32x4 (&gx)[MT][2]) {
    const bool pin = 16 * s + 8 * hi < npix;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = pin ? fx[0][e] : 0.f; v[4 + e] = pin ? fx[1][e] : 0.f; }
    mlb_u32x4 fh, fl;
    mlb_pack8<SPLIT>(v, fh, fl);
    const bf16x8 vfh = __builtin_bit_cast(bf16x8, fh);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = pin && qok[mt];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = ok ? gx[mt][0][e] : 0.f; v[4 + e] = ok ? gx[mt][1][e] : 0.f; }
      mlb_u32x4 gh, gl;
      mlb_pack8<SPLIT>(v, gh, gl);
      const bf16x8 vgh = __builtin_bit_cast(bf16x8, gh);
      if constexpr (SPLIT) {
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, gl), vfh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vgh, __builtin_bit_cast(bf16x8, fl), acc[mt], 0, 0, 0);
      }
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vgh, vfh, acc[mt], 0, 0, 0);
    }
  };
  // two-deep software pipeline: the loads of step s + 1 are in flight while step s is converted and multiplied
  f32x4 fa[2], fb[2], ga[MT][2], gb2[MT][2];
  int s = s_begin;
  if (s < s_end) load_step(s, fa, ga);
  while (s < s_end) {
    if (s + 1 < s_end) load_step(s + 1, fb, gb2);
    mma_step(s, fa, ga);
    ++s;
    if (s >= s_end) break;
    if (s + 1 < s_end) load_step(s + 1, fa, ga);
    mma_step(s, fb, gb2);
    ++s;
  }
  // partial plane ws[b][chunk][q][c]: lane j = channel, regs = queries
  float* wp = ws + (((size_t)b * nchunks + chunk) * Q) * C + 32 * wave + j;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = 32 * mt + mlb_row(r, hi);
      if (q < Q) wp[(size_t)q * C] = acc[mt][r];
    }
}

__global__ __launch_bounds__(256) void cgg_mask_logits_grad_embed_reduce(const float* __restrict__ ws,
                                                                         float* __restrict__ gembed, int B, int Q, int q0,
                                                                         int Qtot, int nchunks) {
  constexpr int C = 256;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * Q * C) return;
  const int b = (int)(gid / ((long long)Q * C));
  const long long rem = gid - (long long)b * Q * C;
  const float* p = ws + (size_t)b * nchunks * Q * C + rem;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += p[(size_t)c * Q * C];
  gembed[((size_t)b * Qtot + q0) * C + rem] = s;
}

static void mlb_plan(int B, int npix, int* steps_per_chunk, int* nchunks) {
  const int nsteps = (npix + 15) / 16;
  int want = (512 + B - 1) / B;                 // ~2 workgroups per CU over the batch
  int spc = (nsteps + want - 1) / want;
  if (spc < 64) spc = nsteps < 64 ? nsteps : 64;
  *steps_per_chunk = spc;
  *nchunks = (nsteps + spc - 1) / spc;
}

extern "C" int64_t cgg_mask_logits_backward_workspace_bytes(int B, int Q, int C, int npix) {
  if (B <= 0 || Q <= 0 || C != 256 || npix <= 0) return 0;
  int spc, nch;
  mlb_plan(B, npix, &spc, &nch);
  return (int64_t)B * nch * Q * C * (int64_t)sizeof(float);
}

template <bool SPLIT>
static int mlb_launch_embed(const float* gout, const float* feat, float* ws, int B, int Q, int q0, int Qtot, int npix,
                            int spc, int nch, hipStream_t s) {
  const int MT = (Q + 31) / 32;
#define CGG_MLB_GE(M)                                                                                                   \
  hipLaunchKernelGGL((cgg_mask_logits_grad_embed_kernel<SPLIT, M>), dim3(nch, B), dim3(512), 0, s, gout, feat, ws, Q, q0,   \
                     Qtot, npix, spc, nch)
  switch (MT) {
    case 1: CGG_MLB_GE(1); break;
    case 2: CGG_MLB_GE(2); break;
    case 3: CGG_MLB_GE(3); break;
    default: CGG_MLB_GE(4); break;
  }
#undef CGG_MLB_GE
  return 0;
}

extern "C" int cgg_mask_logits_backward(const float* embed, const float* feat, const float* grad_out, float* grad_embed,
                                        float* grad_feat, void* ws, int B, int Q, int C, int npix, int split,
                                        cgg_stream_t stream) {
  CGG_REQUIRE(embed && feat && grad_out && ws && (grad_embed || grad_feat), CGG_EINVAL, "cgg_mask_logits_backward: null pointer");
  CGG_REQUIRE(B > 0 && Q > 0 && npix > 0, CGG_EINVAL, "cgg_mask_logits_backward: bad sizes");
  CGG_REQUIRE(C == 256, CGG_EUNSUPPORTED, "cgg_mask_logits_backward: C=%d (only 256 is built)", C);
  CGG_REQUIRE(npix % 8 == 0, CGG_EUNSUPPORTED, "cgg_mask_logits_backward: npix=%d must be a multiple of 8", npix);
  CGG_REQUIRE(cgg_aligned16(embed) && cgg_aligned16(feat) && cgg_aligned16(grad_out) && cgg_aligned16(ws) &&
                  (!grad_embed || cgg_aligned16(grad_embed)) && (!grad_feat || cgg_aligned16(grad_feat)),
              CGG_EALIGN, "cgg_mask_logits_backward: buffers must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  if (grad_feat) {
    // query rows in groups of <= 128 (split: the lo fragments of 8 k-steps live in registers) / 256; a group after the first adds
    // into the result of the ones before it (same lane, same element: no atomics, fixed order)
    const int lim = split ? 128 : 256;
    const int T = (npix + 31) / 32;
    int gx = (T + 7) / 8, cap = (512 + B - 1) / B;
    if (gx > cap) gx = cap;
    auto kern = split ? cgg_mask_logits_grad_feat_kernel<true> : cgg_mask_logits_grad_feat_kernel<false>;
    for (int q0 = 0; q0 < Q; q0 += lim) {
      const int qc = Q - q0 < lim ? Q - q0 : lim;
      const int KS = (qc + 15) / 16;
      const size_t lds = (size_t)8 * KS * 64 * 16 * (split ? 2 : 1);
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_mask_logits_backward: cannot raise dynamic LDS to %zu", lds);
      }
      hipLaunchKernelGGL(kern, dim3(gx, B), dim3(512), lds, s, embed, grad_out, grad_feat, qc, npix, T, KS, q0, Q, q0 > 0 ? 1 : 0);
      CGG_CHECK_LAUNCH("cgg_mask_logits_backward(grad_feat)");
    }
  }
  if (grad_embed) {
    int spc, nch;
    mlb_plan(B, npix, &spc, &nch);
    // query rows in groups of <= 128 (4 row tiles of accumulators per wave; the feature chunk is re-read per group)
    for (int q0 = 0; q0 < Q; q0 += 128) {
      const int qc = Q - q0 < 128 ? Q - q0 : 128;
      if (split) mlb_launch_embed<true>(grad_out, feat, (float*)ws, B, qc, q0, Q, npix, spc, nch, s);
      else mlb_launch_embed<false>(grad_out, feat, (float*)ws, B, qc, q0, Q, npix, spc, nch, s);
      CGG_CHECK_LAUNCH("cgg_mask_logits_backward(grad_embed)");
      const long long total = (long long)B * qc * C;
      hipLaunchKernelGGL(cgg_mask_logits_grad_embed_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                         (const float*)ws, grad_embed, B, qc, q0, Q, nch);
      CGG_CHECK_LAUNCH("cgg_mask_logits_backward(reduce)");
    }
  }
  return CGG_OK;
}
