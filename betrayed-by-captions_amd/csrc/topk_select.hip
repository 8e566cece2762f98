// Top-k SELECTION per row (indices of the k largest values, in no particular order): the importance sampling of the mask losses --
// [3P] mmdet get_uncertain_point_coords_with_randomness (called at open_set/models/mask2former_head.py:605): of 3 x 12 544 random
// points per matched query the 9 408 most uncertain ones (largest -|logit|) are kept. torch.topk sorts (a segmented radix sort of
// 184 x 37 632 keys + gathers: 0.37 ms per layer, 3.7 ms per training step at configs[2]); the loss only needs the SET.
// One workgroup per row: 4 radix-select passes over the row (8 bits each, LDS histogram, the row stays in L2) find the k-th largest
// key, one compaction pass writes the indices (wave ballots + one LDS counter add per wave). Ties at the threshold: any of them.
#include "cgg_common.h"

__device__ __forceinline__ uint32_t tks_key(float x) {          // ascending float order == ascending unsigned order
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void cgg_topk_select_kernel(const float* __restrict__ x, int ld, int N, int k, long long* __restrict__ idx) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_mask, s_need, s_out, s_eq;
  const int tid = threadIdx.x;
  const float* row = x + (size_t)blockIdx.x * ld;
  long long* out = idx + (size_t)blockIdx.x * k;
  if (tid == 0) {
    s_prefix = 0u;
    s_mask = 0u;
    s_need = (uint32_t)k;
    s_out = 0u;
    s_eq = 0u;
  }
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const uint32_t prefix = s_prefix, mask = s_mask;
    for (int i = tid; i < N; i += 1024) {
      const uint32_t key = tks_key(row[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {                      // from the top digit down: the digit in which the `need`-th largest key lies
      uint32_t need = s_need, d = 255u;
      for (;; --d) {
        const uint32_t c = hist[d];
        if (c >= need || d == 0u) break;
        need -= c;
      }
      s_need = need;
      s_prefix = prefix | (d << shift);
      s_mask = mask | (255u << shift);
    }
    __syncthreads();
  }
  // threshold key T = s_prefix: every key > T is taken, of the keys == T the first s_need found
  const uint32_t T = s_prefix, need_eq = s_need;
  const int lane = tid & 63;
  for (int i0 = 0; i0 < N; i0 += 1024) {
    const int i = i0 + tid;
    const uint32_t key = i < N ? tks_key(row[i]) : 0u;
    const bool gt = i < N && key > T;
    bool eq = i < N && key == T;
    // ties: a ticket per equal key, the first need_eq tickets win
    if (eq) eq = atomicAdd(&s_eq, 1u) < need_eq;
    const bool take = gt || eq;
    const unsigned long long m = __ballot(take);
    uint32_t base = 0u;
    if (lane == 0 && m) base = atomicAdd(&s_out, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (take) out[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
  }
}

extern "C" int cgg_topk_select(const float* x, int ld, int rows, int N, int k, int64_t* idx, cgg_stream_t stream) {
  CGG_REQUIRE(x && idx, CGG_EINVAL, "cgg_topk_select: null pointer");
  CGG_REQUIRE(rows > 0 && N > 0 && k > 0 && k <= N && ld >= N, CGG_EINVAL, "cgg_topk_select: bad sizes (rows=%d N=%d k=%d ld=%d)", rows, N, k, ld);
  hipLaunchKernelGGL(cgg_topk_select_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, x, ld, N, k, (long long*)idx);
  CGG_CHECK_LAUNCH("cgg_topk_select");
  return CGG_OK;
}
