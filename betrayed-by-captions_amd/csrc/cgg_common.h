// Shared device/host helpers for libcgg_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cgg_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CGG_WAVE 64

// ---- error reporting (thread-local, no exceptions across the ABI) -------------------------------
void cgg_set_error(const char* fmt, ...);

#define CGG_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      cgg_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

#define CGG_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      cgg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

static inline bool cgg_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- bf16 helpers (round-to-nearest-even, NaN preserved) ----------------------------------------
// hardware conversion (the cast lowers to v_cvt_pk_bf16_f32 on gfx950, RNE); a software round costs
// ~15 VALU per element and made the conversion-heavy prologues latency-bound
__device__ __forceinline__ uint16_t cgg_f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float cgg_bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// split f into hi = bf16(f) and lo = bf16(f - hi): f ~= hi + lo with ~2^-17 relative residual
__device__ __forceinline__ void cgg_split_bf(float f, uint16_t& hi, uint16_t& lo) {
  hi = cgg_f2bf(f);
  lo = cgg_f2bf(f - cgg_bf2f(hi));
}

__device__ __forceinline__ uint32_t cgg_pack2(uint16_t a, uint16_t b) {
  return (uint32_t)a | ((uint32_t)b << 16);
}

// XCD-aware block remap: hardware places block b on XCD b % 8; give every XCD one contiguous chunk
// of the logical index space so neighbouring work shares that XCD's private L2 (bijective form,
// cdna_hip_programming.md "XCD swizzle must be bijective").
__device__ __forceinline__ int cgg_xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, idx = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
