// Shared definitions of the MSDeformAttn kernels (msda.hip: forward + gather backward; msda_bwd.hip: sorted-scatter grad_value).
#pragma once
#include "cgg_common.h"

struct MsdaLevels {
  int h[8];
  int w[8];
  int start[8];
};

__device__ __forceinline__ f32x4 cgg_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 cgg_ld4(const uint16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  f32x4 r = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
             __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  return r;
}

// Four corner addresses (clamped into the map) + four weights (zeroed outside) of one sample.
struct MsdaTap {
  int o00, o01, o10, o11;  // row index (y*W+x) inside the level
  float w00, w01, w10, w11;
  float lh, lw;            // fractional parts (for backward)
  bool in;                 // sample inside (-1, H) x (-1, W)
};

__device__ __forceinline__ MsdaTap cgg_msda_tap(float x, float y, int Hl, int Wl) {
  MsdaTap t;
  const float him = y * (float)Hl - 0.5f;
  const float wim = x * (float)Wl - 0.5f;
  t.in = (him > -1.f) && (wim > -1.f) && (him < (float)Hl) && (wim < (float)Wl);
  const float hf = floorf(him), wf = floorf(wim);
  const int h0 = (int)hf, w0 = (int)wf;
  const int h1 = h0 + 1, w1 = w0 + 1;
  t.lh = him - hf;
  t.lw = wim - wf;
  const float hh = 1.f - t.lh, hw = 1.f - t.lw;
  const bool vh0 = t.in && h0 >= 0, vh1 = t.in && h1 <= Hl - 1;
  const bool vw0 = w0 >= 0, vw1 = w1 <= Wl - 1;
  t.w00 = (vh0 && vw0) ? hh * hw : 0.f;
  t.w01 = (vh0 && vw1) ? hh * t.lw : 0.f;
  t.w10 = (vh1 && vw0) ? t.lh * hw : 0.f;
  t.w11 = (vh1 && vw1) ? t.lh * t.lw : 0.f;
  const int ch0 = min(max(h0, 0), Hl - 1), ch1 = min(max(h1, 0), Hl - 1);
  const int cw0 = min(max(w0, 0), Wl - 1), cw1 = min(max(w1, 0), Wl - 1);
  t.o00 = ch0 * Wl + cw0;
  t.o01 = ch0 * Wl + cw1;
  t.o10 = ch1 * Wl + cw0;
  t.o11 = ch1 * Wl + cw1;
  return t;
}


// grad_value of the encoder's self-attention case by the sorted-scatter kernel (msda_bwd.hip); returns CGG_EUNSUPPORTED (without
// setting the error string) when the pyramid is not tileable -- the caller then takes the generic global-atomic kernel
bool msda_bwd_sorted_ok(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P);
long long msda_bwd_two_pass_workspace_bytes(const MsdaLevels& lv, int B, int Nv, int H, int D, int L, int Nq, int P);
int msda_bwd_sorted_launch_two_pass(const MsdaLevels& lv, const float* loc, const float* attw, const float* gout, float* gvalue, int B,
                                    int Nv, int H, int D, int L, int Nq, int P, void* ws, long long ws_bytes, hipStream_t s, int gvld = 0);
int msda_bwd_sorted_launch(const MsdaLevels& lv, const float* loc, const float* attw, const float* gout, float* gvalue, int B, int Nv,
                           int H, int D, int L, int Nq, int P, hipStream_t s, int gvld = 0);
