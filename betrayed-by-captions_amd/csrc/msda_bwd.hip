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

// NT threads; every thread owns at most MAXIT taps (items) whose geometry stays in registers across the sort's barriers
// (PT = points per level as a compile-time constant, 0 = the run-time value: the tap -> (slot, point) split is a division otherwise)
template <int NT, int MAXIT, int PT>
__global__ __launch_bounds__(NT) void cgg_msda_bwd_sorted_kernel(MsdaLevels lv, MsdaSortPlan pl, const float* __restrict__ loc,
                                                                const float* __restrict__ attw, const float* __restrict__ gout,
                                                                float* __restrict__ gvalue, int Nv, int H, int L, int Nq, int P_rt) {
  constexpr int D = 32;
  const int P = PT ? PT : P_rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;                                                       // [maxslots][32] grad_out rows of the tile's queries
  uint2* rec = reinterpret_cast<uint2*>(gs + (size_t)pl.maxslots * D);    // [maxslots * P * 4]: sorted in-window records from the
                                                                          // front, out-of-window records from the back
  const int cap = pl.maxslots * P * 4 + 3 * pl.maxpix;                    // + up to 3 zero records per window pixel (groups of 4)
  uint32_t* cnt = reinterpret_cast<uint32_t*>(rec + (size_t)cap);
  uint32_t* beg = cnt + pl.maxpix;
  int* qn = reinterpret_cast<int*>(beg + pl.maxpix);                      // [maxslots] query index of a slot
  __shared__ uint32_t wsum[NT / 64];
  __shared__ uint32_t nfb;                                                // out-of-window corners of this workgroup
  const int tid = threadIdx.x;
  // workgroup order: (image, tile) slowest, then head, then destination level -- the 8 x L workgroups that read the same rows of
  // grad_out / sampling_loc / attn_weight (one 128-B / 32-B / 16-B piece each of the tile's 1-KB / 768-B / 384-B rows) run side by
  // side on one XCD, so the rows come from HBM once (first version: level fastest, head slow -> every line re-fetched per head,
  // ~3.2 GB of L2 fills per call at configs[2] shapes)
  const int bid = cgg_xcd_remap(blockIdx.x, gridDim.x);
  const int ld = bid % L;                                                 // destination level
  const int h = (bid / L) % H;
  const int tile = (bid / (L * H)) % pl.ntile;
  const int b = bid / (L * H * pl.ntile);
  const int tyi = tile / pl.tx, txi = tile % pl.tx;
  const size_t rowstride = (size_t)H * D;

  // ---- slots: the tile's queries, all query levels flattened ----
  int nslots = 0;
  for (int l = 0; l < L; ++l) {
    const int e = pl.c * pl.s[l];
    const int xx = txi * e, yy = tyi * e;
    const int tww = min(e, lv.w[l] - xx), thh = min(e, lv.h[l] - yy);
    const int ns = tww > 0 && thh > 0 ? tww * thh : 0;
    if ((tww & (tww - 1)) == 0) {            // (workgroup-uniform) power-of-two row: no integer division
      const int sh = __builtin_ctz(tww > 0 ? tww : 1);
      for (int i = tid; i < ns; i += NT) qn[nslots + i] = lv.start[l] + (yy + (i >> sh)) * lv.w[l] + xx + (i & (tww - 1));
    } else {
      for (int i = tid; i < ns; i += NT) qn[nslots + i] = lv.start[l] + (yy + i / tww) * lv.w[l] + xx + i % tww;
    }
    nslots += ns;
  }
  // window of the destination level
  const int Hd = lv.h[ld], Wd = lv.w[ld];
  const int ed = pl.c * pl.s[ld];
  const int ox = txi * ed - pl.R, oy = tyi * ed - pl.R;
  const int ww = ed + 2 * pl.R;
  const int npix = ww * ww;
  for (int i = tid; i < npix; i += NT) cnt[i] = 0u;
  if (tid == 0) nfb = 0u;
  __syncthreads();

  float* gvl = gvalue + ((size_t)b * Nv + lv.start[ld]) * rowstride + (size_t)h * D;
  const int nitems = nslots * P;
  // ---- this thread's taps: locations / weights requested first (the longest latency), then the grad_out rows are staged ----
  float tx_[MAXIT], ty_[MAXIT], tw_[MAXIT];
  int tslot[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int item = tid + it * NT;
    const bool live = item < nitems;
    const int slot = live ? item / P : 0;
    const int p = live ? item - slot * P : 0;
    const size_t idx = ((((size_t)b * Nq + qn[slot]) * H + h) * L + ld) * P + p;
    const float2 xy = *reinterpret_cast<const float2*>(loc + 2 * idx);
    tx_[it] = xy.x;
    ty_[it] = xy.y;
    tw_[it] = live ? attw[idx] : 0.f;
    tslot[it] = live ? slot : -1;
  }
  for (int i = tid; i < nslots * 8; i += NT) {
    const int slot = i >> 3, cq = i & 7;
    const f32x4 g = cgg_ld4(gout + ((size_t)b * Nq + qn[slot]) * rowstride + (size_t)h * D + cq * 4);
    *reinterpret_cast<f32x4*>(gs + slot * D + cq * 4) = g;
  }
  // ---- pass 1: corner geometry; histogram over the window pixels with the corner's RANK inside its pixel as the return value ----
  // per corner: dst >= 0: window pixel, -1: not a valid corner, <= -2: outside the window, -(row index in the level) - 2
  int dst[MAXIT][4];
  uint32_t rank[MAXIT][4];
  float cf[MAXIT][4];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const float him = ty_[it] * (float)Hd - 0.5f, wim = tx_[it] * (float)Wd - 0.5f;
    const bool in = tslot[it] >= 0 && (him > -1.f) && (wim > -1.f) && (him < (float)Hd) && (wim < (float)Wd);
    const float hf = floorf(him), wf = floorf(wim);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool vh0 = in && h0 >= 0, vh1 = in && (h0 + 1) <= Hd - 1;
    const bool vw0 = w0 >= 0, vw1 = (w0 + 1) <= Wd - 1;
    const bool k[4] = {vh0 && vw0, vh0 && vw1, vh1 && vw0, vh1 && vw1};
    cf[it][0] = tw_[it] * hh * hw;
    cf[it][1] = tw_[it] * hh * lw;
    cf[it][2] = tw_[it] * lh * hw;
    cf[it][3] = tw_[it] * lh * lw;
    const int wy0 = h0 - oy, wx0 = w0 - ox;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wy = wy0 + (q >> 1), wx = wx0 + (q & 1);
      const bool inw = (unsigned)wy < (unsigned)ww && (unsigned)wx < (unsigned)ww;
      dst[it][q] = !k[q] ? -1 : (inw ? wy * ww + wx : -((h0 + (q >> 1)) * Wd + w0 + (q & 1)) - 2);
      rank[it][q] = 0u;
      if (dst[it][q] >= 0) rank[it][q] = atomicAdd(&cnt[dst[it][q]], 1u);
    }
  }
  __syncthreads();       // histogram complete, grad_out rows staged
  // ---- exclusive scan of the histogram, every pixel's list rounded up to a multiple of FOUR records (the sum loop below walks
  //      groups of four without a tail test; the pad records are written after pass 2) ----
  {
    const int per = (npix + NT - 1) / NT;
    const int i0 = tid * per;
    uint32_t sum = 0;
    for (int i = 0; i < per; ++i)
      if (i0 + i < npix) sum += (cnt[i0 + i] + 3u) & ~3u;
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
        run += (cnt[i0 + i] + 3u) & ~3u;
      }
  }
  __syncthreads();
  // ---- pass 2: the records into pixel order (position = pixel's begin + rank: no second atomic); corners outside the window are
  //      appended from the back of the same buffer as (slot | row << 12, coefficient) ----
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int d = dst[it][q];
      if (d >= 0) {
        rec[beg[d] + rank[it][q]] = make_uint2((uint32_t)tslot[it] * (D * 4u), __float_as_uint(cf[it][q]));      // .x = byte offset of the staged row
      } else if (d <= -2) {
        const uint32_t i = atomicAdd(&nfb, 1u);
        rec[cap - 1 - (int)i] = make_uint2((uint32_t)tslot[it] | ((uint32_t)(-d - 2) << 12), __float_as_uint(cf[it][q]));
      }
    }
  }
  __syncthreads();
  // pad records: coefficient 0 on the row of the pixel's FIRST record (a row that reaches this pixel anyway: a non-finite grad_out
  // row must not leak into pixels it does not touch through 0 x inf)
  for (int i = tid; i < npix; i += NT) {
    const uint32_t n = cnt[i];
    if (n & 3u) {
      const uint32_t b0 = beg[i], x0 = rec[b0].x;
      for (uint32_t k = n; k < ((n + 3u) & ~3u); ++k) rec[b0 + k] = make_uint2(x0, 0u);
    }
  }
  __syncthreads();
  // ---- destination-stationary sums: half-wave = one window pixel, lane = channel. A pixel's records are walked FOUR at a time: two
  //      16-byte record reads, four grad_out-row reads, four FMAs into independent sums (a one-record loop is a serial record ->
  //      row -> FMA chain of two LDS latencies per record). The kernel was VALU-bound (counters: 2760 VALU instructions per wave, the
  //      VALU pipes ~100 % busy): lists padded to groups of four, row byte offsets in the records and an incremental (row, column)
  //      of the window pixel take the tail selects, the index multiplies and the two integer divisions out of this loop ----
  const int lane = tid & 31, hwid = tid >> 5;
  constexpr int NHW = NT / 32;
  // (eight lanes per pixel with four channels each -- a wavefront walking 8 pixels at a time -- cut the VALU count by another
  // 20 % but its four dword atomics per lane are 32 scattered line requests per 8 pixels instead of 8 full lines: 2.5 ms vs 1.24)
  const char* gsb = reinterpret_cast<const char*>(gs) + lane * 4;
  // (walking a compacted list of the NON-EMPTY pixels instead -- one LDS read for begin / count / destination, no visits to empty
  // halo pixels -- measured slower, 1.44 vs 1.28 ms: this phase, 0.74 ms of the kernel, is bound by the LDS pipe, ~2/3 of it the
  // broadcast reads of the records themselves: 16 bytes x 64 lanes per two records whatever the number of distinct addresses)
  int wy = hwid / ww, wx = hwid - wy * ww;                     // window coordinates of this half-wave's pixel
  const int dy = NHW / ww, dx = NHW - dy * ww;                 // (dx < ww: at most one wrap per step)
  for (int pix = hwid; pix < npix; pix += NHW) {
    const int n = (int)cnt[pix];
    if (n > 0) {
      const uint4* r = reinterpret_cast<const uint4*>(rec + beg[pix]);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int i = 0; i < n; i += 4, r += 2) {
        const uint4 ra = r[0], rb = r[1];
        const float g0 = *reinterpret_cast<const float*>(gsb + ra.x), g1 = *reinterpret_cast<const float*>(gsb + ra.z);
        const float g2 = *reinterpret_cast<const float*>(gsb + rb.x), g3 = *reinterpret_cast<const float*>(gsb + rb.z);
        a0 = fmaf(__uint_as_float(ra.y), g0, a0);
        a1 = fmaf(__uint_as_float(ra.w), g1, a1);
        a2 = fmaf(__uint_as_float(rb.y), g2, a2);
        a3 = fmaf(__uint_as_float(rb.w), g3, a3);
      }
      // inside the image: only valid corners were counted
      atomicAdd(gvl + (size_t)((oy + wy) * Wd + ox + wx) * rowstride + lane, (a0 + a1) + (a2 + a3));
    }
    wy += dy;
    wx += dx;
    if (wx >= ww) {
      wx -= ww;
      ++wy;
    }
  }
  // ---- corners outside the window: one 128-byte atomic each (large learned offsets: correctness does not depend on locality) ----
  const int nf = (int)nfb;
  for (int i = hwid; i < nf; i += NHW) {
    const uint2 r0 = rec[cap - 1 - i];
    atomicAdd(gvl + (size_t)(r0.x >> 12) * rowstride + lane, __uint_as_float(r0.y) * gs[(r0.x & 0xfffu) * D + lane]);
  }
}

static int msda_sort_plan(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P, MsdaSortPlan& pl, size_t& lds);

bool msda_bwd_sorted_ok(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P) {
  MsdaSortPlan pl;
  size_t lds;
  return msda_sort_plan(lv, B, Nv, H, D, L, Nq, P, pl, lds) == CGG_OK;
}

static int msda_sort_plan(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P, MsdaSortPlan& pl, size_t& lds) {
  if (D != 32 || Nq != Nv || L < 1 || L > 8 || P < 1 || P > 16) return CGG_EUNSUPPORTED;
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
  // tile edge c (coarsest-level pixels) and halo R = 4 pixels (the reference's initialisation puts point p at p + 1 pixels from the
  // reference point: 4 covers it). Measured at configs[2] shapes (offset std 0.5 px, per backward call): c = 2 2.84 ms, c = 4
  // 4.04 ms (93 KB of LDS: one workgroup per CU), R = 2 / 3 with c = 2: 2.71 / 2.87 ms; the window flush (one 128-byte global atomic
  // per window pixel) is 0.06 ms of it -- the atomics are NOT the bound, the per-workgroup phase latencies are
  const int force_c = 0, force_r = 0;
  // c = 2 first: 25 KB of LDS per workgroup, five or six workgroups per CU overlap each other's load / sort / sum phases (measured
  // at configs[2] shapes, +-2 px offsets: c = 2 1.70 ms, c = 4 2.57 ms per call -- the 93-KB c = 4 tile runs one workgroup per CU)
  const int cands[3] = {2, 4, 1};
  lds = 0;
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
    lds = (size_t)slots * 32 * 4 + ((size_t)slots * P * 4 + 3 * (size_t)maxpix) * 8 + (size_t)maxpix * 2 * 4 + (size_t)slots * 4;
    if (lds <= 150 * 1024 && slots < 4096 && slots * P <= 3 * 512) {
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
  for (int l = 0; l < L; ++l)
    if ((long long)lv.h[l] * lv.w[l] >= (1ll << 20)) return CGG_EUNSUPPORTED;      // out-of-window records pack the row index in 20 bits
  return CGG_OK;
}

int msda_bwd_sorted_launch(const MsdaLevels& lv, const float* loc, const float* attw, const float* gout, float* gvalue, int B, int Nv,
                           int H, int D, int L, int Nq, int P, hipStream_t s) {
  MsdaSortPlan pl;
  size_t lds;
  const int rc = msda_sort_plan(lv, B, Nv, H, D, L, Nq, P, pl, lds);
  if (rc != CGG_OK) return rc;
  const long long nblk = (long long)B * H * pl.ntile * L;
  const int items = pl.maxslots * P;
  // 256 threads where two taps per thread cover the tile (c = 2: 336 taps), 512 for the c = 4 tile; 384-thread workgroups with one
  // tap per thread measured the same (2.96 vs 2.85 ms per backward call at configs[2] shapes)
  const int nt = items > 512 ? 512 : 256;
  auto kern = nt == 512 ? cgg_msda_bwd_sorted_kernel<512, 3, 0>
              : (items > 256 ? (P == 4 ? cgg_msda_bwd_sorted_kernel<256, 2, 4> : cgg_msda_bwd_sorted_kernel<256, 2, 0>)
                             : cgg_msda_bwd_sorted_kernel<256, 1, 0>);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    cgg_set_error("cgg_msda_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(nt), lds, s, lv, pl, loc, attw, gout, gvalue, Nv, H, L, Nq, P);
  CGG_CHECK_LAUNCH("cgg_msda_backward(sorted scatter)");
  return CGG_OK;
}
