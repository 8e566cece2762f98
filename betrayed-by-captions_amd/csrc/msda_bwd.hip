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
//
// Round 6, TWO PASSES (`cgg_msda_backward_hostlevels_ws`): with trained offsets of several pixels most corners leave the c = 2 / R = 4
// window and the single-pass form degenerates into one 128-byte atomic per corner (round 5: 2.0 ms per call at +-0.5 px, 8.5 ms at
// +-8 px). Pass A (MODE 1) is the kernel above except that out-of-window corners are only COUNTED, per (image, 4 x 4-coarse-pixel
// region, head, destination level), in a workspace; pass B (MODE 2) runs the same sort on c = 4 tiles with a 12-pixel halo over
// exactly those corners -- a workgroup whose counter is zero returns at once, a corner inside its query's pass-A window is skipped
// (the two passes evaluate the same expressions on the same inputs, so every corner is taken by exactly one of them) -- and only
// what leaves even that window goes to per-corner atomics. Where the offsets are small pass B costs its launch and an early exit.
#include "msda_common.h"

#define MSDA_LIGHT_MAX 768u      // pass B: regions with fewer corners left take the light (no-sort) form
#define MSDA_DEFER_MIN 256u      // pass A: a workgroup with fewer out-of-window corners (of its 1 344) scatters them itself

struct MsdaSortPlan {
  int c, R, tx, ty, ntile;
  int cA, RA;        // MODE 2: tile edge and halo of pass A (whose in-window corners this pass skips)
  int txB, ntileB;   // MODE 1 / 2: pass-B tiling (regions of cB = 2 cA coarse pixels) -- the overflow counters' geometry
  int txA, ntileA;   // MODE >= 2: pass-A tiling -- the geometry of the `deferred` flags
  int s[8];          // W_l / W_coarse
  int maxslots;      // queries of a full tile, all levels
  int maxpix;        // largest window, pixels
};

// NT threads; every thread owns at most MAXIT taps (items) whose geometry stays in registers across the sort's barriers
// (PT = points per level as a compile-time constant, 0 = the run-time value: the tap -> (slot, point) split is a division otherwise)
// MODE 0: single pass (out-of-window corners -> per-corner atomics); 1: pass A (they are counted into ovf); 2: pass B, sort form;
// 3: pass B, light form (no sort, no window: the region's few corners straight to per-corner atomics -- a small-LDS launch).
// One (tile, image, head, destination level) unit `bid`; `smem` = the workgroup's dynamic LDS.
template <int NT, int MAXIT, int PT, int MODE>
__device__ __forceinline__ void msda_sorted_region(const MsdaLevels& lv, const MsdaSortPlan& pl, const float* __restrict__ loc,
                                                   const float* __restrict__ attw, const float* __restrict__ gout,
                                                   float* __restrict__ gvalue, int Nv, int H, int L, int Nq, int P_rt,
                                                   uint32_t* __restrict__ ovf, uint32_t* __restrict__ aflags, const int bid, float* smem,
                                                   const int gvld) {
  constexpr int D = 32;
  constexpr bool light = MODE == 3;
  const int P = PT ? PT : P_rt;
  float* gs = smem;                                                       // [maxslots][32] grad_out rows of the tile's queries
  uint2* rec = reinterpret_cast<uint2*>(gs + (light ? 0 : (size_t)pl.maxslots * D));   // [maxslots * P * 4]: sorted in-window records
                                                                          // from the front, out-of-window records from the back
  const int cap = light ? (int)MSDA_LIGHT_MAX : pl.maxslots * P * 4 + 3 * pl.maxpix;   // + up to 3 zero records per window pixel
  const int wpix = light ? 0 : pl.maxpix;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(rec + (size_t)cap);
  uint32_t* beg = cnt + wpix;
  int* qn = reinterpret_cast<int*>(beg + wpix);                           // [maxslots] query index of a slot
  int* qa = qn + pl.maxslots;                                             // MODE 2: [maxslots] pass-A tile (x | y << 16) of a slot's query
  __shared__ uint32_t wsum[NT / 64];
  __shared__ uint32_t nfb;                                                // out-of-window corners of this workgroup
  const int tid = threadIdx.x;
  // workgroup order: (image, tile) slowest, then head, then destination level -- the 8 x L workgroups that read the same rows of
  // grad_out / sampling_loc / attn_weight (one 128-B / 32-B / 16-B piece each of the tile's 1-KB / 768-B / 384-B rows) run side by
  // side on one XCD, so the rows come from HBM once (first version: level fastest, head slow -> every line re-fetched per head,
  // ~3.2 GB of L2 fills per call at configs[2] shapes)
  const int ld = bid % L;                                                 // destination level
  const int h = (bid / L) % H;
  const int tile = (bid / (L * H)) % pl.ntile;
  const int b = bid / (L * H * pl.ntile);
  const int tyi = tile / pl.tx, txi = tile % pl.tx;
  const size_t rowstride = (size_t)H * D;
  // the overflow counter of this workgroup's region: pass A adds to its region's word, pass B reads its own
  uint32_t* myovf = nullptr;
  if constexpr (MODE == 1) myovf = ovf + (((size_t)b * pl.ntileB + (size_t)(tyi >> 1) * pl.txB + (txi >> 1)) * H + h) * L + ld;
  if constexpr (MODE >= 2) myovf = ovf + bid;                             // (pass B's unit numbering IS the counters' index)
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
    if constexpr (MODE >= 2) {               // the pass-A tile each query belonged to (edge cA s_l pixels of its level)
      const int ea = pl.cA * pl.s[l];
      for (int i = tid; i < ns; i += NT) qa[nslots + i] = ((xx + i % tww) / ea) | (((yy + i / tww) / ea) << 16);
    }
    nslots += ns;
  }
  // window of the destination level
  const int Hd = lv.h[ld], Wd = lv.w[ld];
  const int ed = pl.c * pl.s[ld];
  const int ox = txi * ed - pl.R, oy = tyi * ed - pl.R;
  const int ww = ed + 2 * pl.R;
  const int npix = ww * ww;
  if (!light)
    for (int i = tid; i < npix; i += NT) cnt[i] = 0u;
  if (tid == 0) nfb = 0u;
  __syncthreads();

  // grad_value rows may be padded like the value rows (gvld floats per pixel, 0 = H D): consecutive pixels' 128-byte atomics then
  // rotate over the L2 channels instead of hitting one 512-byte-aligned set
  const size_t gvrow = gvld ? (size_t)gvld : rowstride;
  float* gvl = gvalue + ((size_t)b * Nv + lv.start[ld]) * gvrow + (size_t)h * D;
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
  if (!light)
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
    // MODE 2: window of pass A around this query's pass-A tile, in destination-level pixels
    int ay0 = 0, ax0 = 0, aww = 0;
    bool deferred = true;                    // MODE >= 2: did this query's pass-A workgroup leave its out-of-window corners to us?
    if constexpr (MODE >= 2) {
      const int qat = qa[tslot[it] >= 0 ? tslot[it] : 0];
      deferred = aflags[(((size_t)b * pl.ntileA + (size_t)(qat >> 16) * pl.txA + (qat & 0xffff)) * H + h) * L + ld] != 0u;
      const int eda = pl.cA * pl.s[ld];
      ax0 = w0 - ((qat & 0xffff) * eda - pl.RA);
      ay0 = h0 - ((qat >> 16) * eda - pl.RA);
      aww = eda + 2 * pl.RA;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wy = wy0 + (q >> 1), wx = wx0 + (q & 1);
      const bool inw = !light && (unsigned)wy < (unsigned)ww && (unsigned)wx < (unsigned)ww;
      bool take = k[q];
      if constexpr (MODE >= 2)               // pass A has summed (or scattered) this corner already
        take = take && deferred && !((unsigned)(ay0 + (q >> 1)) < (unsigned)aww && (unsigned)(ax0 + (q & 1)) < (unsigned)aww);
      dst[it][q] = !take ? -1 : (inw ? wy * ww + wx : -((h0 + (q >> 1)) * Wd + w0 + (q & 1)) - 2);
      rank[it][q] = 0u;
      if (dst[it][q] >= 0) rank[it][q] = atomicAdd(&cnt[dst[it][q]], 1u);
    }
  }
  __syncthreads();       // histogram complete, grad_out rows staged
  if (!light)            // (workgroup-uniform: the barrier inside is taken by all or none)
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
  const int lane = tid & 31, hwid = tid >> 5;
  constexpr int NHW = NT / 32;
  if (!light) {
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
        atomicAdd(gvl + (size_t)((oy + wy) * Wd + ox + wx) * gvrow + lane, (a0 + a1) + (a2 + a3));
      }
      wy += dy;
      wx += dx;
      if (wx >= ww) {
        wx -= ww;
        ++wy;
      }
    }
  }
  // ---- corners outside the window: one 128-byte atomic each (large learned offsets: correctness does not depend on locality) ----
  // pass A: a workgroup with MANY of them leaves them to pass B (flag + the region's counter); a few are scattered right here -- a
  // second visit of the taps would cost more than the atomics (measured at offsets ~ N(0, 2 px): 2.9 vs 2.4 ms per call)
  if constexpr (MODE == 1) {
    if (nfb >= MSDA_DEFER_MIN) {                                          // (workgroup-uniform: nfb is final after the barrier above)
      if (tid == 0) {
        atomicAdd(myovf, nfb);
        aflags[bid] = 1u;
      }
      return;
    }
  }
  const int nf = (int)nfb;
  for (int i = hwid; i < nf; i += NHW) {
    const uint2 r0 = rec[cap - 1 - i];
    const uint32_t slot = r0.x & 0xfffu;
    const float g = light ? gout[((size_t)b * Nq + qn[slot]) * rowstride + (size_t)h * D + lane] : gs[slot * D + lane];
    atomicAdd(gvl + (size_t)(r0.x >> 12) * gvrow + lane, __uint_as_float(r0.y) * g);
  }
}

// grid-mapped launch (single pass / pass A): one workgroup per unit.
// workgroup order: (image, tile) slowest, then head, then destination level -- the 8 x L workgroups that read the same rows of
// grad_out / sampling_loc / attn_weight run side by side on one XCD, so the rows come from HBM once
template <int NT, int MAXIT, int PT, int MODE>
__global__ __launch_bounds__(NT) void cgg_msda_bwd_sorted_kernel(MsdaLevels lv, MsdaSortPlan pl, const float* __restrict__ loc,
                                                                const float* __restrict__ attw, const float* __restrict__ gout,
                                                                float* __restrict__ gvalue, int Nv, int H, int L, int Nq, int P_rt,
                                                                uint32_t* __restrict__ ovf, uint32_t* __restrict__ aflags, int gvld) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  msda_sorted_region<NT, MAXIT, PT, MODE>(lv, pl, loc, attw, gout, gvalue, Nv, H, L, Nq, P_rt, ovf, aflags,
                                          cgg_xcd_remap(blockIdx.x, gridDim.x), smem, gvld);
}

// pass B: PERSISTENT workgroups walk a list of units (built on the device by cgg_msda_bwd_classify_kernel from pass A's counters)
// through an atomic cursor -- no workgroup is dispatched for a region that has nothing left, whatever the offsets look like.
template <int NT, int MAXIT, int PT, int MODE>
__global__ __launch_bounds__(NT) void cgg_msda_bwd_list_kernel(MsdaLevels lv, MsdaSortPlan pl, const float* __restrict__ loc,
                                                              const float* __restrict__ attw, const float* __restrict__ gout,
                                                              float* __restrict__ gvalue, int Nv, int H, int L, int Nq, int P_rt,
                                                              uint32_t* __restrict__ ovf, uint32_t* __restrict__ aflags,
                                                              const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                              uint32_t* __restrict__ cursor, int gvld) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ uint32_t cur;
  const uint32_t n = *count;
  for (;;) {
    if (threadIdx.x == 0) cur = atomicAdd(cursor, 1u);
    __syncthreads();
    const uint32_t idx = cur;
    if (idx >= n) break;                                                  // (workgroup-uniform)
    msda_sorted_region<NT, MAXIT, PT, MODE>(lv, pl, loc, attw, gout, gvalue, Nv, H, L, Nq, P_rt, ovf, aflags, (int)list[idx], smem, gvld);
    __syncthreads();                                                      // the unit's LDS (and `cur`) are reused by the next one
  }
}

// header words of the two-pass workspace behind the nr counters: [0] light units, [1] sort units, [2] / [3] the lists' cursors
__global__ __launch_bounds__(256) void cgg_msda_bwd_classify_kernel(const uint32_t* __restrict__ ovf, uint32_t* __restrict__ hdr,
                                                                    uint32_t* __restrict__ light_list, uint32_t* __restrict__ sort_list,
                                                                    int nr) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nr) return;
  const uint32_t left = ovf[r];
  if (left == 0u) return;
  if (left < MSDA_LIGHT_MAX) light_list[atomicAdd(&hdr[0], 1u)] = (uint32_t)r;
  else sort_list[atomicAdd(&hdr[1], 1u)] = (uint32_t)r;
}

static int msda_sort_plan(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P, MsdaSortPlan& pl, size_t& lds,
                          int force_c = 0, int force_r = 0);

bool msda_bwd_sorted_ok(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P) {
  MsdaSortPlan pl;
  size_t lds;
  return msda_sort_plan(lv, B, Nv, H, D, L, Nq, P, pl, lds) == CGG_OK;
}

static int msda_sort_plan(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P, MsdaSortPlan& pl, size_t& lds,
                          int force_c, int force_r) {
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
    // grad_out rows | records (+ 3 pad records per window pixel) | histogram + begin | query index + pass-A tile of a slot
    lds = (size_t)slots * 32 * 4 + ((size_t)slots * P * 4 + 3 * (size_t)maxpix) * 8 + (size_t)maxpix * 2 * 4 + (size_t)slots * 4 * 2;
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
  pl.cA = pl.RA = 0;
  pl.txB = pl.ntileB = pl.txA = pl.ntileA = 0;
  const long long nblk = (long long)B * H * pl.ntile * L;
  if (nblk >= (1ll << 31)) return CGG_EUNSUPPORTED;
  for (int l = 0; l < L; ++l)
    if ((long long)lv.h[l] * lv.w[l] >= (1ll << 20)) return CGG_EUNSUPPORTED;      // out-of-window records pack the row index in 20 bits
  return CGG_OK;
}

template <int MODE>
static int msda_sorted_go(const MsdaLevels& lv, const MsdaSortPlan& pl, size_t lds, const float* loc, const float* attw, const float* gout,
                          float* gvalue, int B, int Nv, int H, int L, int Nq, int P, uint32_t* ovf, uint32_t* aflags, hipStream_t s,
                          int gvld) {
  const long long nblk = (long long)B * H * pl.ntile * L;
  const int items = pl.maxslots * P;
  // 256 threads where two taps per thread cover the tile (c = 2: 336 taps), 512 for the c = 4 tile; 384-thread workgroups with one
  // tap per thread measured the same (2.96 vs 2.85 ms per backward call at configs[2] shapes)
  const int nt = items > 512 ? 512 : 256;
  auto kern = nt == 512 ? cgg_msda_bwd_sorted_kernel<512, 3, 0, MODE>
              : (items > 256 ? (P == 4 ? cgg_msda_bwd_sorted_kernel<256, 2, 4, MODE> : cgg_msda_bwd_sorted_kernel<256, 2, 0, MODE>)
                             : cgg_msda_bwd_sorted_kernel<256, 1, 0, MODE>);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    cgg_set_error("cgg_msda_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(nt), lds, s, lv, pl, loc, attw, gout, gvalue, Nv, H, L, Nq, P, ovf, aflags, gvld);
  CGG_CHECK_LAUNCH("cgg_msda_backward(sorted scatter)");
  return CGG_OK;
}

int msda_bwd_sorted_launch(const MsdaLevels& lv, const float* loc, const float* attw, const float* gout, float* gvalue, int B, int Nv,
                           int H, int D, int L, int Nq, int P, hipStream_t s, int gvld) {
  MsdaSortPlan pl;
  size_t lds;
  const int rc = msda_sort_plan(lv, B, Nv, H, D, L, Nq, P, pl, lds);
  if (rc != CGG_OK) return rc;
  return msda_sorted_go<0>(lv, pl, lds, loc, attw, gout, gvalue, B, Nv, H, L, Nq, P, nullptr, nullptr, s, gvld);
}

// ---- two passes (see the header of this file): workspace = one counter per (image, pass-B tile, head, level) ----
#define MSDA_PASSB_C 4
#define MSDA_PASSB_R 12
static int msda_two_pass_plans(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P, MsdaSortPlan& pa, size_t& la,
                               MsdaSortPlan& pb, size_t& lb) {
  int rc = msda_sort_plan(lv, B, Nv, H, D, L, Nq, P, pa, la);
  if (rc != CGG_OK) return rc;
  if (pa.c * 2 != MSDA_PASSB_C) return CGG_EUNSUPPORTED;                     // (pass A took c = 4 or 1 itself: single pass)
  rc = msda_sort_plan(lv, B, Nv, H, D, L, Nq, P, pb, lb, MSDA_PASSB_C, MSDA_PASSB_R);
  if (rc != CGG_OK) return rc;
  pb.cA = pa.c;
  pb.RA = pa.R;
  pa.txB = pb.txB = pb.tx;
  pa.ntileB = pb.ntileB = pb.ntile;
  pa.txA = pb.txA = pa.tx;
  pa.ntileA = pb.ntileA = pa.ntile;
  return CGG_OK;
}

// workspace: [nr counters][16 header words][na `deferred` flags][nr light units][nr sort units], nr = B x pass-B tiles x H x L,
// na = B x pass-A tiles x H x L
static long long msda_two_pass_ws_bytes(long long nr, long long na) { return (3 * nr + 16 + na) * 4; }

long long msda_bwd_two_pass_workspace_bytes(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P) {
  MsdaSortPlan pa, pb;
  size_t la, lb;
  if (msda_two_pass_plans(lv, B, Nv, H, D, L, Nq, P, pa, la, pb, lb) != CGG_OK) return 0;
  return msda_two_pass_ws_bytes((long long)B * pb.ntile * H * L, (long long)B * pa.ntile * H * L);
}

template <int MODE>
static int msda_list_go(const MsdaLevels& lv, const MsdaSortPlan& pl, size_t lds, int nwg, const float* loc, const float* attw,
                        const float* gout, float* gvalue, int Nv, int H, int L, int Nq, int P, uint32_t* ovf, uint32_t* aflags,
                        const uint32_t* list, const uint32_t* count, uint32_t* cursor, hipStream_t s, int gvld) {
  auto kern = cgg_msda_bwd_list_kernel<512, 3, 0, MODE>;                     // (pass-B tiles: c = 4, 1 344 taps at P = 4)
  if (pl.maxslots * P > 3 * 512) return CGG_EUNSUPPORTED;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    cgg_set_error("cgg_msda_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), lds, s, lv, pl, loc, attw, gout, gvalue, Nv, H, L, Nq, P, ovf, aflags, list, count,
                     cursor, gvld);
  CGG_CHECK_LAUNCH("cgg_msda_backward(sorted scatter, second pass)");
  return CGG_OK;
}

int msda_bwd_sorted_launch_two_pass(const MsdaLevels& lv, const float* loc, const float* attw, const float* gout, float* gvalue, int B,
                                    int Nv, int H, int D, int L, int Nq, int P, void* ws, long long ws_bytes, hipStream_t s, int gvld) {
  MsdaSortPlan pa, pb;
  size_t la, lb;
  int rc = msda_two_pass_plans(lv, B, Nv, H, D, L, Nq, P, pa, la, pb, lb);
  const long long nr = rc == CGG_OK ? (long long)B * pb.ntile * H * L : 0;
  const long long na = rc == CGG_OK ? (long long)B * pa.ntile * H * L : 0;
  if (rc != CGG_OK || !ws || ws_bytes < msda_two_pass_ws_bytes(nr, na) || na >= (1ll << 31))
    return msda_bwd_sorted_launch(lv, loc, attw, gout, gvalue, B, Nv, H, D, L, Nq, P, s, gvld);
  uint32_t* ovf = (uint32_t*)ws;
  uint32_t* hdr = ovf + nr;
  uint32_t* aflags = hdr + 16;
  uint32_t* light_list = aflags + na;
  uint32_t* sort_list = light_list + nr;
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)(nr + 16 + na) * 4, s);        // counters + header + flags; the lists need no initialisation
  if (e != hipSuccess) {
    cgg_set_error("cgg_msda_backward: zeroing the overflow counters failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  rc = msda_sorted_go<1>(lv, pa, la, loc, attw, gout, gvalue, B, Nv, H, L, Nq, P, ovf, aflags, s, gvld);
  if (rc != CGG_OK) return rc;
  hipLaunchKernelGGL(cgg_msda_bwd_classify_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, ovf, hdr, light_list, sort_list,
                     (int)nr);
  CGG_CHECK_LAUNCH("cgg_msda_backward(classify)");
  // light form: records of < MSDA_LIGHT_MAX corners + the slot tables = ~9 KB of LDS, 4 workgroups of 512 threads per CU
  const size_t ll = (size_t)MSDA_LIGHT_MAX * 8 + (size_t)pb.maxslots * 4 * 2;
  rc = msda_list_go<3>(lv, pb, ll, 256 * 4, loc, attw, gout, gvalue, Nv, H, L, Nq, P, ovf, aflags, light_list, hdr + 0, hdr + 2, s, gvld);
  if (rc != CGG_OK) return rc;
  // sort form: ~137 KB of LDS, one workgroup per CU
  return msda_list_go<2>(lv, pb, lb, 256, loc, attw, gout, gvalue, Nv, H, L, Nq, P, ovf, aflags, sort_list, hdr + 1, hdr + 3, s, gvld);
}
