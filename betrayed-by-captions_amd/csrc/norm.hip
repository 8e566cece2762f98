// GroupNorm (+ optional ReLU) for the pixel decoder's ConvModules (norm_cfg=dict(type='GN', num_groups=32),
// configs/instance/coco_b48n17.py:40; [3P] MSDeformAttnPixelDecoder input/lateral/output convs).
// HBM-bound: at 256x256 a (batch, group) row holds 524288 elements but there are only B*32 = 64 rows, so a
// row-per-block reduction leaves 3/4 of the chip idle. Here every row is split into chunks (pass 1: partial
// sum / sum of squares per chunk, full-chip), and pass 2 folds the partials and normalises.
#include "cgg_common.h"

#define GN_CHUNK 16384  // elements per partial

template <int VEC>
__global__ __launch_bounds__(256) void cgg_gn_partial_kernel(const float* __restrict__ x,
                                                             float* __restrict__ part, long long row_len,
                                                             int chunks) {
  const int row = blockIdx.y, chunk = blockIdx.x;
  const long long beg = (long long)chunk * GN_CHUNK;
  const long long end = min(row_len, beg + GN_CHUNK);
  const float* xr = x + (size_t)row * row_len;
  float s = 0.f, ss = 0.f;
  if (VEC == 4) {  // H*W % 4 == 0 and 16-B aligned rows -> float4 loads
    for (long long i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
      s += (v[0] + v[1]) + (v[2] + v[3]);
      ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  } else {
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
      const float v = xr[i];
      s += v;
      ss += v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  __shared__ float sm[8];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[wave] = s;
    sm[4 + wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)row * chunks + chunk) * 2] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    part[((size_t)row * chunks + chunk) * 2 + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void cgg_gn_apply_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ part,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ y, int C, int groups,
                                                           long long hw, int chunks, float eps, int relu) {
  // blockIdx.y = (b, channel), blockIdx.x = 1024-element span of that plane
  const int bc = blockIdx.y;
  const int c = bc % C, b = bc / C;
  const int cpg = C / groups;
  const int row = b * groups + c / cpg;
  double s = 0.0, ss = 0.0;
  for (int i = 0; i < chunks; ++i) {
    s += (double)part[((size_t)row * chunks + i) * 2];
    ss += (double)part[((size_t)row * chunks + i) * 2 + 1];
  }
  const double n = (double)cpg * (double)hw;
  const double mean = s / n;
  const double var = fmax(ss / n - mean * mean, 0.0);
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = gamma[c] * rstd;
  const float sh = beta[c] - (float)mean * a;
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i < hw) {
    const size_t off = (size_t)bc * hw + i;
    if (VEC == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r = v[e] * a + sh;
        v[e] = relu ? fmaxf(r, 0.f) : r;
      }
      *reinterpret_cast<f32x4*>(y + off) = v;
    } else {
      const float r = x[off] * a + sh;
      y[off] = relu ? fmaxf(r, 0.f) : r;
    }
  }
}

extern "C" int64_t cgg_group_norm_workspace_bytes(int B, int C, int H, int W, int groups) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups) return 0;
  const long long row_len = (long long)(C / groups) * H * W;
  const int chunks = (int)((row_len + GN_CHUNK - 1) / GN_CHUNK);
  return (int64_t)B * groups * chunks * 2 * (int64_t)sizeof(float);
}

extern "C" int cgg_group_norm(const float* x, const float* gamma, const float* beta, float* y, void* ws, int B,
                              int C, int H, int W, int groups, float eps, int relu, cgg_stream_t stream) {
  CGG_REQUIRE(x && gamma && beta && y && ws, CGG_EINVAL, "cgg_group_norm: null pointer");
  CGG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && groups > 0, CGG_EINVAL, "cgg_group_norm: bad sizes");
  CGG_REQUIRE(C % groups == 0, CGG_EINVAL, "cgg_group_norm: C=%d not divisible by groups=%d", C, groups);
  const long long hw = (long long)H * W;
  const bool vec = (hw % 4 == 0) && cgg_aligned16(x) && cgg_aligned16(y);
  const long long row_len = (long long)(C / groups) * hw;
  const int chunks = (int)((row_len + GN_CHUNK - 1) / GN_CHUNK);
  hipStream_t s = (hipStream_t)stream;
  if (vec) {
    hipLaunchKernelGGL(cgg_gn_partial_kernel<4>, dim3(chunks, B * groups), dim3(256), 0, s, x, (float*)ws,
                       row_len, chunks);
    hipLaunchKernelGGL(cgg_gn_apply_kernel<4>, dim3((unsigned)((hw / 4 + 255) / 256), B * C), dim3(256), 0, s,
                       x, (const float*)ws, gamma, beta, y, C, groups, hw, chunks, eps, relu);
  } else {
    hipLaunchKernelGGL(cgg_gn_partial_kernel<1>, dim3(chunks, B * groups), dim3(256), 0, s, x, (float*)ws,
                       row_len, chunks);
    hipLaunchKernelGGL(cgg_gn_apply_kernel<1>, dim3((unsigned)((hw + 255) / 256), B * C), dim3(256), 0, s, x,
                       (const float*)ws, gamma, beta, y, C, groups, hw, chunks, eps, relu);
  }
  CGG_CHECK_LAUNCH("cgg_group_norm");
  return CGG_OK;
}
