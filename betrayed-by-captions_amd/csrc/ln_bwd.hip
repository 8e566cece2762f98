// Backward of y = LayerNorm(a + b) over 256 channels for the TRAINING encoder stream ([3P] BaseTransformerLayer: the residual add in
// front of each 'norm' of the MSDeformAttn encoder layers, open_set/models/mask2former_head.py:112-117; 16 x 21 504 rows at
// configs[2]). torch runs add / layer_norm forward as two passes and the backward as two more kernels (gamma / beta partials,
// grad input) at ~1.9 TB/s; here the forward is cgg_add_layernorm_ex (one pass) and the backward ONE pass:
//
//     s = a + b,  xhat = (s - mean(s)) * rstd(s)                                   (statistics recomputed from the row: nothing saved)
//     dy = dy32 + dy16a + dy16b   (any subset: the forward can also emit bf16(y) and bf16(y + pos), whose gradients arrive in bf16)
//     g = dy * gamma,   dx = rstd * (g - mean(g) - xhat * mean(g * xhat))           d/da = d/db = dx   (dx16 = bf16(dx), optional)
//     dgamma = sum_rows dy * xhat,   dbeta = sum_rows dy                            as per-workgroup partials (summed by the caller)
//
// 16 lanes share a row (lane sub = lane & 15 holds columns 4 sub + 64 k .. + 3, k = 0..3): the four row reductions are DPP adds
// inside a 16-lane row, a wavefront works on 4 rows at a time, the column sums stay in 32 registers per lane until the end.
#include "cgg_common.h"

#define LNB_ROWS 256          // rows per workgroup (4 wavefronts x 4 rows x 16 iterations)

__device__ __forceinline__ float lnb_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
  return v;
}

__device__ __forceinline__ f32x4 lnb_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 lnb_load4(const uint16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}

template <typename BT>
__global__ __launch_bounds__(256) void cgg_add_layernorm_bwd_kernel(const float* __restrict__ dy, const uint16_t* __restrict__ dy16a,
                                                                   const uint16_t* __restrict__ dy16b, const float* __restrict__ a,
                                                                   const BT* __restrict__ b, const float* __restrict__ gamma,
                                                                   float eps, float* __restrict__ dx, uint16_t* __restrict__ dx16,
                                                                   float* __restrict__ partial, int rows, float* __restrict__ dx_amax) {
  __shared__ float red[4][512];
  float lmax = 0.f;                                                          // max |dx| of this lane (dx_amax)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, rsub = lane >> 4;
  f32x4 g4[4], dg[4], db[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    g4[k] = *reinterpret_cast<const f32x4*>(gamma + 4 * sub + 64 * k);
    dg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int row0 = blockIdx.x * LNB_ROWS;
  constexpr float inv_n = 1.f / 256.f;
  for (int it = 0; it < LNB_ROWS / 16; ++it) {
    const int row = row0 + 16 * it + 4 * wave + rsub;
    const bool live = row < rows;
    const int rc = live ? row : rows - 1;                                    // clamped: the reductions stay convergent
    const size_t off = (size_t)rc * 256 + 4 * sub;
    f32x4 s[4], y[4];
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] = lnb_load4(a + off + 64 * k);
      if (b != nullptr) s[k] += lnb_load4(b + off + 64 * k);
      y[k] = dy != nullptr ? lnb_load4(dy + off + 64 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (dy16a != nullptr) y[k] += lnb_load4(dy16a + off + 64 * k);
      if (dy16b != nullptr) y[k] += lnb_load4(dy16b + off + 64 * k);
      sm += (s[k][0] + s[k][1]) + (s[k][2] + s[k][3]);
    }
    const float mean = lnb_row16_sum(sm) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] = s[k] - mean;
      q += (s[k][0] * s[k][0] + s[k][1] * s[k][1]) + (s[k][2] * s[k][2] + s[k][3] * s[k][3]);
    }
    const float rstd = rsqrtf(lnb_row16_sum(q) * inv_n + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] = s[k] * rstd;                                                    // xhat
      const f32x4 g = y[k] * g4[k];
      c1 += (g[0] + g[1]) + (g[2] + g[3]);
      const f32x4 gx = g * s[k];
      c2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
    }
    c1 = lnb_row16_sum(c1) * inv_n;
    c2 = lnb_row16_sum(c2) * inv_n;
    if (live) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 d = (y[k] * g4[k] - c1 - s[k] * c2) * rstd;
        *reinterpret_cast<f32x4*>(dx + off + 64 * k) = d;
        lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
        if (dx16)
          *reinterpret_cast<uint2*>(dx16 + off + 64 * k) =
              make_uint2(cgg_pack2(cgg_f2bf(d[0]), cgg_f2bf(d[1])), cgg_pack2(cgg_f2bf(d[2]), cgg_f2bf(d[3])));
        dg[k] += y[k] * s[k];
        db[k] += y[k];
      }
    }
  }
  // max |dx| for the per-tensor pre-scale of the contractions that consume dx as grad_output (x3.h): one atomic per wavefront
  if (dx_amax) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, sft, 64));
    if (lane == 0 && __float_as_uint(lmax) > __hip_atomic_load(reinterpret_cast<unsigned int*>(dx_amax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned int*>(dx_amax), __float_as_uint(lmax));      // (only when it can raise the running maximum)
  }
  // column sums: the 4 row groups of a wavefront (lanes sub, sub + 16, + 32, + 48), then the 4 wavefronts through LDS
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = dg[k][c], w = db[k][c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      w += __shfl_xor(w, 16);
      w += __shfl_xor(w, 32);
      if (rsub == 0) {
        red[wave][4 * sub + 64 * k + c] = v;
        red[wave][256 + 4 * sub + 64 * k + c] = w;
      }
    }
  __syncthreads();
  for (int i = tid; i < 512; i += 256)
    partial[(size_t)blockIdx.x * 512 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

extern "C" int64_t cgg_add_layernorm_backward_partials(int rows) { return rows > 0 ? (int64_t)((rows + LNB_ROWS - 1) / LNB_ROWS) : 0; }

static int lnb_launch(const float* dy, const void* dy16a, const void* dy16b, const float* a, const void* b, int b_dtype,
                      const float* gamma, float eps, float* dx, void* dx16, float* partial, int rows, int N, float* dx_amax,
                      cgg_stream_t stream) {
  CGG_REQUIRE((dy || dy16a || dy16b) && a && gamma && dx && partial, CGG_EINVAL, "cgg_add_layernorm_backward: null pointer");
  CGG_REQUIRE(rows > 0 && N == 256, CGG_EUNSUPPORTED, "cgg_add_layernorm_backward: rows=%d N=%d (only N = 256 is built)", rows, N);
  CGG_REQUIRE(b_dtype == CGG_F32 || b_dtype == CGG_BF16, CGG_EUNSUPPORTED, "cgg_add_layernorm_backward: b dtype %d", b_dtype);
  CGG_REQUIRE((!dy || cgg_aligned16(dy)) && (!dy16a || cgg_aligned16(dy16a)) && (!dy16b || cgg_aligned16(dy16b)) && cgg_aligned16(a) && (!b || cgg_aligned16(b)) && cgg_aligned16(gamma) && cgg_aligned16(dx) &&
                  (!dx16 || cgg_aligned16(dx16)),
              CGG_EALIGN, "cgg_add_layernorm_backward: 16-B alignment");
  const dim3 grid((rows + LNB_ROWS - 1) / LNB_ROWS), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dx_amax) {
    hipError_t e = hipMemsetAsync(dx_amax, 0, sizeof(float), s);
    CGG_REQUIRE(e == hipSuccess, (int)e, "cgg_add_layernorm_backward: memset failed");
  }
  if (b_dtype == CGG_BF16)
    hipLaunchKernelGGL(cgg_add_layernorm_bwd_kernel<uint16_t>, grid, block, 0, s, dy, (const uint16_t*)dy16a, (const uint16_t*)dy16b, a, (const uint16_t*)b, gamma, eps, dx,
                       (uint16_t*)dx16, partial, rows, dx_amax);
  else
    hipLaunchKernelGGL(cgg_add_layernorm_bwd_kernel<float>, grid, block, 0, s, dy, (const uint16_t*)dy16a, (const uint16_t*)dy16b, a, (const float*)b, gamma, eps, dx,
                       (uint16_t*)dx16, partial, rows, dx_amax);
  CGG_CHECK_LAUNCH("cgg_add_layernorm_backward");
  return CGG_OK;
}

extern "C" int cgg_add_layernorm_backward(const float* dy, const void* dy16a, const void* dy16b, const float* a, const void* b,
                                          int b_dtype, const float* gamma, float eps, float* dx, void* dx16, float* partial,
                                          int rows, int N, cgg_stream_t stream) {
  return lnb_launch(dy, dy16a, dy16b, a, b, b_dtype, gamma, eps, dx, dx16, partial, rows, N, nullptr, stream);
}

// ... and max |dx| as a device scalar (dx_amax, written by the call) for the x3 contractions that take dx as grad_output
extern "C" int cgg_add_layernorm_backward_amax(const float* dy, const void* dy16a, const void* dy16b, const float* a, const void* b,
                                               int b_dtype, const float* gamma, float eps, float* dx, void* dx16, float* partial,
                                               float* dx_amax, int rows, int N, cgg_stream_t stream) {
  CGG_REQUIRE(dx_amax, CGG_EINVAL, "cgg_add_layernorm_backward_amax: null dx_amax");
  return lnb_launch(dy, dy16a, dy16b, a, b, b_dtype, gamma, eps, dx, dx16, partial, rows, N, dx_amax, stream);
}
