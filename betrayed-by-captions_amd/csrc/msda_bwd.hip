// K2 (training): grad_value of MSDeformAttn for the encoder's self-attention case -- queries == the pixels of the value pyramid,
// integer scale between levels -- WITHOUT scatter atomics in LDS. Replaces round 2-4's `cgg_msda_bwd_tiled_kernel` (64-bit
// fixed-point LDS windows: 129 K `ds_add_u64` per tile at ~4 lanes / clk / CU, 3.4 ms per layer at configs[2] shapes, 6.2 GB of
// fabric writes for a 352-MB gradient). Reference semantics: [3P] mmcv ms_deform_attn_backward, reached from
// MSDeformAttnPixelDecoder (open_set/models/mask2former_head.py:112-117, :787); oracle/ops.py:9-28.
//
//   grad_value[b, pix, h, :] = sum over (query q, level l, point p, corner k) landing on pix of  attw[b,q,h,l,p] * cw_k * grad_out[b,q,h,:]
//
// One workgroup = (image-space tile, image b, head h, DESTINATION level l). The tile is c x c pixels of the coarsest level and the
// co-located (c s_l)^2 pixels of every finer level: all its queries sample around the same image region, so almost every corner
// lands in the tile's window of level l (tile footprint + R pixels of halo). Instead of adding 32 channels per corner into an LDS
// window, the kernel SORTS the tile's corner records by destination pixel (counting sort in LDS: one 32-bit `ds_add_rtn` per corner
// for the histogram, one for the slot -- 32 x fewer LDS atomics, none of them floating point) and then walks the sorted list
// destination-stationary: a half-wavefront owns one destination pixel at a time, its 32 lanes are the 32 channels of the head,
// the sum lives in a register, grad_out rows of the tile's queries are staged once in LDS (conflict-free 128-byte row reads).
// Each window pixel leaves as ONE 128-byte global f32 atomic (windows of neighbouring tiles overlap in the halo); with c = 4 the
// windows hold 2.9 x the tile's pixels (c = 2: 6 x). Corners outside the window (large learned offsets) go straight to global
// atomics: the result does not depend on the locality assumption, only the speed does.
#include "msda_common.h"

struct MsdaSortPlan {
  int c, R, tx, ty, ntile;
  int s[8];          // W_l / W_coarse
  int maxslots;      // queries of a full tile, all levels
  int maxpix;        // largest window, pixels
};

template <int NT>
__global__ __launch_bounds__(NT) void cgg_msda_bwd_sorted_kernel(MsdaLevels lv, MsdaSortPlan pl, const float* __restrict__ loc,
                                                                const float* __restrict__ attw, const float* __restrict__ gout,
                                                                float* __restrict__ gvalue, int Nv, int H, int L, int Nq, int P) {
  constexpr int D = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;                                                       // [maxslots][32] grad_out rows of the tile's queries
  uint2* rec = reinterpret_cast<uint2*>(gs + (size_t)pl.maxslots * D);    // [maxslots * P * 4] (slot, coefficient bits), sorted by pixel
  uint32_t* cnt = reinterpret_cast<uint32_t*>(rec + (size_t)pl.maxslots * P * 4);
  uint32_t* beg = cnt + pl.maxpix;
  uint32_t* cur = beg + pl.maxpix;
  int* qn = reinterpret_cast<int*>(cur + pl.maxpix);                      // [maxslots] query index of a slot
  __shared__ int nsl[8];
  __shared__ uint32_t wsum[NT / 64];
  const int tid = threadIdx.x;
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const int ld = bid % L;                                                 // destination level
  const int tile = (bid / L) % pl.ntile;
  const int h = (bid / (L * pl.ntile)) % H;
  const int b = bid / (L * pl.ntile * H);
  const int tyi = tile / pl.tx, txi = tile % pl.tx;
  const size_t rowstride = (size_t)H * D;

  // ---- slots: the tile's queries, all query levels flattened ----
  int nslots = 0;
  for (int l = 0; l < L; ++l) {
    const int e = pl.c * pl.s[l];
    const int xx = txi * e, yy = tyi * e;
    const int tww = min(e, lv.w[l] - xx), thh = min(e, lv.h[l] - yy);
    const int ns = tww * thh;
    if (tid == 0) nsl[l] = ns;
    for (int i = tid; i < ns; i += NT) qn[nslots + i] = lv.start[l] + (yy + i / tww) * lv.w[l] + xx + i % tww;
    nslots += ns;
  }
  // window of the destination level
  const int Hd = lv.h[ld], Wd = lv.w[ld];
  const int ed = pl.c * pl.s[ld];
  const int ox = txi * ed - pl.R, oy = tyi * ed - pl.R;
  const int ww = ed + 2 * pl.R;
  const int npix = ww * ww;
  for (int i = tid; i < npix; i += NT) cnt[i] = 0u;
  __syncthreads();
  // ---- stage grad_out[b, q, h, 0:32] of every slot (16-byte loads, 128-byte rows) ----
  for (int i = tid; i < nslots * 8; i += NT) {
    const int slot = i >> 3, cq = i & 7;
    const f32x4 g = cgg_ld4(gout + ((size_t)b * Nq + qn[slot]) * rowstride + (size_t)h * D + cq * 4);
    *reinterpret_cast<f32x4*>(gs + slot * D + cq * 4) = g;
  }
  __syncthreads();

  float* gvl = gvalue + ((size_t)b * Nv + lv.start[ld]) * rowstride + (size_t)h * D;
  const int nitems = nslots * P;
  // one tap's corners: window pixel (or -1 = outside the window -> global path; -2 = not a valid corner) and coefficient
  auto corners = [&](int item, int& slot, int (&pix)[4], float (&cf)[4], int (&ro)[4]) {
    slot = item / P;
    const int p = item - slot * P;
    const size_t idx = ((((size_t)b * Nq + qn[slot]) * H + h) * L + ld) * P + p;
    const float x = loc[2 * idx], y = loc[2 * idx + 1], w = attw[idx];
    const float him = y * (float)Hd - 0.5f, wim = x * (float)Wd - 0.5f;
    const bool in = (him > -1.f) && (wim > -1.f) && (him < (float)Hd) && (wim < (float)Wd);
    const float hf = floorf(him), wf = floorf(wim);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool vh0 = in && h0 >= 0, vh1 = in && (h0 + 1) <= Hd - 1;
    const bool vw0 = w0 >= 0, vw1 = (w0 + 1) <= Wd - 1;
    const bool k[4] = {vh0 && vw0, vh0 && vw1, vh1 && vw0, vh1 && vw1};
    cf[0] = w * hh * hw;
    cf[1] = w * hh * lw;
    cf[2] = w * lh * hw;
    cf[3] = w * lh * lw;
    const int wy0 = h0 - oy, wx0 = w0 - ox;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wy = wy0 + (q >> 1), wx = wx0 + (q & 1);
      const bool inw = (unsigned)wy < (unsigned)ww && (unsigned)wx < (unsigned)ww;
      pix[q] = !k[q] ? -2 : (inw ? wy * ww + wx : -1);
      ro[q] = (h0 + (q >> 1)) * Wd + w0 + (q & 1);
    }
  };

  // ---- pass 1: histogram over the window pixels; corners outside the window take the global path now ----
  for (int item = tid; item < nitems; item += NT) {
    int slot, pix[4], ro[4];
    float cf[4];
    corners(item, slot, pix, cf, ro);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (pix[q] >= 0) {
        atomicAdd(&cnt[pix[q]], 1u);
      } else if (pix[q] == -1) {
        float* d = gvl + (size_t)ro[q] * rowstride;
        for (int c = 0; c < D; ++c) atomicAdd(d + c, cf[q] * gs[slot * D + c]);
      }
    }
  }
  __syncthreads();
  // ---- exclusive scan of the histogram (block-wide: per-thread serial chunks, wave shuffles, wave totals through LDS) ----
  {
    const int per = (npix + NT - 1) / NT;
    const int i0 = tid * per;
    uint32_t sum = 0;
    for (int i = 0; i < per; ++i)
      if (i0 + i < npix) sum += cnt[i0 + i];
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
      if ((tid & 63) >= o) inc += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    uint32_t run = base + inc - sum;
    for (int i = 0; i < per; ++i)
      if (i0 + i < npix) {
        beg[i0 + i] = run;
        cur[i0 + i] = run;
        run += cnt[i0 + i];
      }
  }
  __syncthreads();
  // ---- pass 2: the records into pixel order ----
  for (int item = tid; item < nitems; item += NT) {
    int slot, pix[4], ro[4];
    float cf[4];
    corners(item, slot, pix, cf, ro);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (pix[q] >= 0) {
        const uint32_t pos = atomicAdd(&cur[pix[q]], 1u);
        rec[pos] = make_uint2((uint32_t)slot, __float_as_uint(cf[q]));
      }
  }
  __syncthreads();
  // ---- destination-stationary sums: half-wave = one window pixel, lane = channel ----
  const int lane = tid & 31, hwid = tid >> 5;
  constexpr int NHW = NT / 32;
  for (int pix = hwid; pix < npix; pix += NHW) {
    const int n = (int)cnt[pix];
    if (n == 0) continue;
    const uint2* r = rec + beg[pix];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + 3 < n; i += 4) {
      const uint2 r0 = r[i], r1 = r[i + 1], r2 = r[i + 2], r3 = r[i + 3];
      a0 = fmaf(__uint_as_float(r0.y), gs[r0.x * D + lane], a0);
      a1 = fmaf(__uint_as_float(r1.y), gs[r1.x * D + lane], a1);
      a2 = fmaf(__uint_as_float(r2.y), gs[r2.x * D + lane], a2);
      a3 = fmaf(__uint_as_float(r3.y), gs[r3.x * D + lane], a3);
    }
    for (; i < n; ++i) {
      const uint2 r0 = r[i];
      a0 = fmaf(__uint_as_float(r0.y), gs[r0.x * D + lane], a0);
    }
    const int iy = oy + pix / ww, ix = ox + pix % ww;          // inside the image: only valid corners were counted
    atomicAdd(gvl + (size_t)(iy * Wd + ix) * rowstride + lane, (a0 + a1) + (a2 + a3));
  }
}

int msda_bwd_sorted_launch(const MsdaLevels& lv, const float* loc, const float* attw, const float* gout, float* gvalue, int B, int Nv,
                           int H, int D, int L, int Nq, int P, hipStream_t s) {
  if (D != 32 || Nq != Nv || L < 1 || L > 8 || P < 1 || P > 16) return CGG_EUNSUPPORTED;
  MsdaSortPlan pl;
  int lc = 0;
  long long tot = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.w[l] < lv.w[lc]) lc = l;
    tot += (long long)lv.h[l] * lv.w[l];
  }
  if (tot != Nv) return CGG_EUNSUPPORTED;
  for (int l = 0; l < L; ++l) {
    if (lv.w[l] % lv.w[lc] || lv.h[l] % lv.h[lc] || lv.w[l] / lv.w[lc] != lv.h[l] / lv.h[lc]) return CGG_EUNSUPPORTED;
    pl.s[l] = lv.w[l] / lv.w[lc];
  }
  for (int l = L; l < 8; ++l) pl.s[l] = 0;
  // tile edge c (coarsest-level pixels) and halo R: the largest tile whose buffers fit the LDS budget (bigger tiles = fewer window
  // pixels per tile pixel = fewer flush atomics); CGG_MSDA_BWD_C / _R override (measurement)
  static const int force_c = getenv("CGG_MSDA_BWD_C") ? atoi(getenv("CGG_MSDA_BWD_C")) : 0;
  static const int force_r = getenv("CGG_MSDA_BWD_R") ? atoi(getenv("CGG_MSDA_BWD_R")) : 0;
  const int cands[3] = {4, 2, 1};
  size_t lds = 0;
  bool ok = false;
  for (int k = 0; k < 3 && !ok; ++k) {
    const int c = force_c > 0 ? force_c : cands[k];
    const int R = force_r > 0 ? force_r : 4;
    long long slots = 0;
    int maxpix = 0;
    for (int l = 0; l < L; ++l) {
      const int e = c * pl.s[l];
      slots += (long long)e * e;
      const int ww = e + 2 * R;
      maxpix = ww * ww > maxpix ? ww * ww : maxpix;
    }
    lds = (size_t)slots * 32 * 4 + (size_t)slots * P * 4 * 8 + (size_t)maxpix * 3 * 4 + (size_t)slots * 4;
    if (lds <= 150 * 1024 && slots <= 4096) {
      pl.c = c;
      pl.R = R;
      pl.maxslots = (int)slots;
      pl.maxpix = maxpix;
      ok = true;
    }
    if (force_c > 0) break;
  }
  if (!ok) return CGG_EUNSUPPORTED;
  pl.tx = (lv.w[lc] + pl.c - 1) / pl.c;
  pl.ty = (lv.h[lc] + pl.c - 1) / pl.c;
  pl.ntile = pl.tx * pl.ty;
  const long long nblk = (long long)B * H * pl.ntile * L;
  if (nblk >= (1ll << 31)) return CGG_EUNSUPPORTED;
  const bool big = pl.maxslots * P > 1024;
  auto kern = big ? cgg_msda_bwd_sorted_kernel<512> : cgg_msda_bwd_sorted_kernel<256>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    cgg_set_error("cgg_msda_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(big ? 512 : 256), lds, s, lv, pl, loc, attw, gout, gvalue, Nv, H, L, Nq, P);
  CGG_CHECK_LAUNCH("cgg_msda_backward(sorted scatter)");
  return CGG_OK;
}
