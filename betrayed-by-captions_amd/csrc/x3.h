// "f32-class" contractions on the f16 matrix cores (parity mode of the path; runtime precision 'fp32').
//
// The reference computes every contraction of open_set/models/mask2former_head.py:763-849 in f32. gfx950's f32-input
// MFMA runs at 1/16 of the 16-bit rate, so parity mode splits each f32 operand into two f16 pieces and issues THREE
// v_mfma_f32_32x32x16_f16 per product into ONE f32 accumulator:
//
//     a' = s_a a = ah + al,   ah = f16(a'),  al = f16(a' - ah)          (residual computed exactly in f32)
//     sum_k a'b' ~= sum_k (al bh + ah bl + ah bh)                        (the dropped al bl term is <= 2^-22 |a'b'|)
//
// f16 carries 11 significand bits, so the pair (ah, al) holds a' to 22 bits (f32: 24); the 22-bit products are exact in the
// MFMA's f32 accumulation. Operands are pre-scaled by powers of two so that the low pieces stay well inside f16's range:
// activations by CGG_X3_ASCALE = 2^4 (|a| < 4094 required -- larger values overflow to inf / NaN, loudly), every weight row n
// by 2^e_n with max_k |w'_nk| in [2^10, 2^11). Low pieces that still fall below 2^-14 become f16 subnormals (absolute
// error <= 2^-25 of the pre-scaled value); v_cvt_pk_f16_f32 produces them and the f16 MFMA consumes them without flushing
// (checked on MI355X, scratch/x3/denorm_test.hip). The accumulator is un-scaled in the epilogue by colscale[n] =
// 2^-e_n / CGG_X3_ASCALE (exact). Measured against float64 the result is as accurate as an f32 GEMM (tests/test_x3_gpu.py);
// a 3 x bf16 split (8 + 8 bits, round 1-2) was 8-10 x worse than f32.
//
// Packed weight ("x3 image") of W [N, K] f32, K % 16 == 0, NT = ceil(N / 32), KS = K / 16:
//     hi  [NT][KS][64 lanes] x 16 B   B fragments of v_mfma_f32_32x32x16_f16: lane l holds W'[32 nt + (l & 31)][16 ks + 8 (l >> 5) .. + 7]
//     lo  [NT][KS][64 lanes] x 16 B   the residual pieces, same order
//     colscale [NT * 32] f32
// = cgg_x3_packed_bytes(N, K) bytes; rows >= N are zero with colscale 0.
#pragma once
#include "cgg_common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float cgg_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t cgg_u32x4;

#define CGG_X3_ASCALE 16.0f
#define CGG_X3_INV_ASCALE 0.0625f

// (a, b) -> packed f16 pair hi and residual pair lo of the values AS GIVEN (caller applies the pre-scale)
__device__ __forceinline__ void cgg_x3_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const cgg_f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);                       // v_cvt_pk_f16_f32 (RNE)
  const cgg_f32x2 r = v - __builtin_convertvector(h, cgg_f32x2);           // exact
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// 4 activations (un-scaled) -> 8-byte halves of an A-fragment slot
__device__ __forceinline__ void cgg_x3_split2_s(float a, float b, float sc, uint32_t& hi, uint32_t& lo);
__device__ __forceinline__ void cgg_x3_split4(const f32x4 v, uint2& hi, uint2& lo) {
  cgg_x3_split2_s(v[0], v[1], CGG_X3_ASCALE, hi.x, lo.x);
  cgg_x3_split2_s(v[2], v[3], CGG_X3_ASCALE, hi.y, lo.y);
}

// ---- per-tensor pre-scale (round 5) -----------------------------------------------------------------------------
// The fixed pre-scale 2^4 suits O(1) ACTIVATIONS. Gradients are not unit scale: with |g| ~ 1e-6 the hi piece of 16 g is already
// an f16 subnormal and the pair carries 5-10 bits instead of 22 (dW relative error 1e-3 at |g| = 1e-6, 1e-1 at 1e-8 in a CPU
// emulation). Operands whose magnitude is not known a priori (grad_output of the training linears / convolutions) are therefore
// pre-scaled by a power of two taken from the tensor's max |value| (cgg_absmax_f32, one pass): s = 2^(9 - floor(log2 amax)), so
// that s amax lies in [2^9, 2^10) like the packed weights; the epilogue multiplies by CGG_X3_ASCALE / s (exact). amax = 0 /
// denormal / non-finite keeps the default 2^4.
__device__ __forceinline__ float cgg_x3_scale_from_amax(float amax) {
  const uint32_t b = __builtin_bit_cast(uint32_t, amax);
  const int e = (int)((b >> 23) & 0xffu);
  if (e == 0 || e == 255) return CGG_X3_ASCALE;
  int se = 9 - (e - 127);
  se = se > 100 ? 100 : (se < -100 ? -100 : se);
  return __builtin_bit_cast(float, (uint32_t)(se + 127) << 23);
}
// (a, b) scaled by the power of two sc -> hi / lo pairs, same bits as cgg_x3_split2(a sc, b sc). The residual sc a - hi comes from
// ONE mixed-precision fma per value (v_fma_mixlo / mixhi_f16: f32 a, f32 sc, f16 hi half -> f16; sc a - hi is exact in f32, so the
// only rounding is the final one to f16, as before) instead of cvt_f32_f16 + subtract + cvt_pk, and the hi piece from one more (f16(sc a)): 2 VALU per value instead of 3.5 --
// the split is the main loop's VALU load in the kernels that take f32 operands (training GEMMs: 2.4 VALU per MFMA before).
// HAZARD (round 6): the pieces must not be consumed by an MFMA straight from these registers. The hazard recogniser does not look
// inside an asm block, so no wait states are inserted between the VALU writes here and a v_mfma that reads the register a few cycles
// later -- csrc/xattn_bwd.hip's first x3 form did exactly that and multiplied stale operands in some lanes (gradients off by 20 % ..
// 100 x). Every user in this library sends the pieces through LDS (or holds them across a barrier) first; a kernel that splits and
// multiplies in registers takes plain __builtin_convertvector conversions (xattn_bwd.hip `xb_split4`). (Measured: ONE wait state -- an
// `s_nop 0` tied to the outputs -- removes the error, but the volatile asm then costs more scheduling freedom than the two VALU
// instructions per value it saves: 124 vs 118 us at 1 024 keys, profiles/r6_xattn_bwd_x3.txt.)
__device__ __forceinline__ void cgg_x3_split2_s(float a, float b, float sc, uint32_t& hi, uint32_t& lo) {
  uint32_t h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a), "v"(sc));      // f16(sc a), RNE
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(b), "v"(sc));
  hi = h;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(sc), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(sc), "v"(hi));
  lo = l;
}
__device__ __forceinline__ void cgg_x3_split4_s(const f32x4 v, float sc, uint2& hi, uint2& lo) {
  cgg_x3_split2_s(v[0], v[1], sc, hi.x, lo.x);
  cgg_x3_split2_s(v[2], v[3], sc, hi.y, lo.y);
}
__device__ __forceinline__ void cgg_x3_split8_s(const f32x4 v0, const f32x4 v1, float sc, cgg_u32x4& hi, cgg_u32x4& lo) {
  uint2 h0, l0, h1, l1;
  cgg_x3_split4_s(v0, sc, h0, l0);
  cgg_x3_split4_s(v1, sc, h1, l1);
  hi = cgg_u32x4{h0.x, h0.y, h1.x, h1.y};
  lo = cgg_u32x4{l0.x, l0.y, l1.x, l1.y};
}

// 8 activations -> one 16-byte A-fragment slot each
__device__ __forceinline__ void cgg_x3_split8(const f32x4 v0, const f32x4 v1, cgg_u32x4& hi, cgg_u32x4& lo) {
  uint2 h0, l0, h1, l1;
  cgg_x3_split4(v0, h0, l0);
  cgg_x3_split4(v1, h1, l1);
  hi = cgg_u32x4{h0.x, h0.y, h1.x, h1.y};
  lo = cgg_u32x4{l0.x, l0.y, l1.x, l1.y};
}

// one activation -> its two f16 pieces (element-wise fragment stores)
__device__ __forceinline__ void cgg_x3_split1(float a, uint16_t& hi, uint16_t& lo) {
  const float s = a * CGG_X3_ASCALE;
  const _Float16 h = (_Float16)s;
  const _Float16 l = (_Float16)(s - (float)h);
  hi = __builtin_bit_cast(uint16_t, h);
  lo = __builtin_bit_cast(uint16_t, l);
}

// acc += A B^T for one k-step, A = (ah, al), B = (bh, bl): small terms first
__device__ __forceinline__ void cgg_x3_mfma(f32x16& acc, const cgg_u32x4 ah, const cgg_u32x4 al, const cgg_u32x4 bh,
                                            const cgg_u32x4 bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
}

// views of an x3 image
struct CggX3W {
  const cgg_u32x4* hi;
  const cgg_u32x4* lo;
  const float* scale;
};

static inline CggX3W cgg_x3_view(const void* packed, int N, int K) {
  const size_t frags = (size_t)((N + 31) / 32) * (K / 16) * 64;
  CggX3W w;
  w.hi = (const cgg_u32x4*)packed;
  w.lo = packed ? w.hi + frags : nullptr;
  w.scale = packed ? (const float*)(w.hi + 2 * frags) : nullptr;
  return w;
}

// ---- "x3a" rows: activations STORED pre-split (round 4) -------------------------------------------------------------------
// A channel-last f32 tensor whose rows are consumed by the x3 GEMM is kept in HBM in the form the MFMA wants: every group of 8
// consecutive channels (32 bytes) holds [8 x f16 hi | 8 x f16 lo] of the PRE-SCALED values a' = 16 a (hi = f16(a'), lo =
// f16(a' - hi)) instead of 8 floats -- same bytes, same addressing ((row * ld + channel) * 4 for channel % 8 == 0), so a
// 32-channel K chunk of a row is still one 128-byte line and goes HBM -> LDS by `buffer_load ... lds` with no conversion in
// the GEMM loop (csrc/x3s_gemm.hip). The GEMM result is bit-identical to splitting an f32 row in the kernel (the split IS what
// the kernel did); other consumers (residual adds, LayerNorm, up-sampling) read hi + lo, which carries 22 significant bits.
// Range: |a| < 4094 (f16 overflow of the pre-scaled value); producers raise the overflow flag (cgg_x3_overflow_flag).
__device__ __forceinline__ void cgg_x3a_split8_prescaled(const float* s, cgg_u32x4& hi, cgg_u32x4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cgg_x3_split2(s[2 * k], s[2 * k + 1], h[k], l[k]);
  hi = cgg_u32x4{h[0], h[1], h[2], h[3]};
  lo = cgg_u32x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ void cgg_x3a_encode8(const float* f, cgg_u32x4& hi, cgg_u32x4& lo) {
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = f[k] * CGG_X3_ASCALE;
  cgg_x3a_split8_prescaled(s, hi, lo);
}
// -> the PRE-SCALED values 16 a (exact: hi + lo has <= 23 significant bits)
__device__ __forceinline__ void cgg_x3a_decode8_prescaled(const cgg_u32x4 hi, const cgg_u32x4 lo, float* s) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (the element is copied to a scalar first: __builtin_bit_cast applied to the vector-element lvalue `hi[k]` read element 0
    // for every k with hipcc 7.2)
    const uint32_t hk = hi[k], lk = lo[k];
    const cgg_f32x2 h = __builtin_convertvector(__builtin_bit_cast(f16x2, hk), cgg_f32x2);
    const cgg_f32x2 l = __builtin_convertvector(__builtin_bit_cast(f16x2, lk), cgg_f32x2);
    s[2 * k] = h[0] + l[0];
    s[2 * k + 1] = h[1] + l[1];
  }
}
__device__ __forceinline__ void cgg_x3a_decode8(const cgg_u32x4 hi, const cgg_u32x4 lo, float* f) {
  cgg_x3a_decode8_prescaled(hi, lo, f);
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] *= CGG_X3_INV_ASCALE;
}
// 4 channels (half a group): 8-byte halves of the hi and lo pieces
__device__ __forceinline__ void cgg_x3a_decode4(const uint2 hi, const uint2 lo, f32x4& f) {
  const cgg_f32x2 h0 = __builtin_convertvector(__builtin_bit_cast(f16x2, hi.x), cgg_f32x2);
  const cgg_f32x2 l0 = __builtin_convertvector(__builtin_bit_cast(f16x2, lo.x), cgg_f32x2);
  const cgg_f32x2 h1 = __builtin_convertvector(__builtin_bit_cast(f16x2, hi.y), cgg_f32x2);
  const cgg_f32x2 l1 = __builtin_convertvector(__builtin_bit_cast(f16x2, lo.y), cgg_f32x2);
  f = f32x4{h0[0] + l0[0], h0[1] + l0[1], h1[0] + l1[0], h1[1] + l1[1]} * CGG_X3_INV_ASCALE;
}
// largest finite pre-scaled magnitude: beyond it the hi piece is inf
#define CGG_X3A_MAX 65504.0f
