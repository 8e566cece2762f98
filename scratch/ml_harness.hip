#define CGG_ML_HARNESS 1
#include <cstdlib>
// standalone timing harness for the mask-logit kernel (experiments; not part of the product)
#include "../betrayed-by-captions_amd/csrc/mask_logits.hip"
#include "../betrayed-by-captions_amd/csrc/cgg_api.hip"
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void copy_mix(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nr, size_t nw) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = i0; i < nr; i += stride) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  for (size_t i = i0; i < nw; i += stride) dst[i] = acc;
}

int main(int argc, char** argv) {
  const int B = 2, Q = 100, C = 256, H = 256, W = 256, P = H * W, T = P / 32;
  float *E, *F, *out; void *hi, *lo;
  CK(hipMalloc(&E, (size_t)B * Q * C * 4)); CK(hipMalloc(&F, (size_t)B * C * P * 4));
  CK(hipMalloc(&out, (size_t)B * Q * P * 4)); CK(hipMalloc(&hi, (size_t)B * T * 32 * C * 2)); CK(hipMalloc(&lo, (size_t)B * T * 32 * C * 2));
  std::vector<float> h((size_t)B * C * P);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(F, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(E, h.data(), (size_t)B * Q * C * 4, hipMemcpyHostToDevice));
  cgg_pack_mask_feature(F, hi, lo, B, C, H, W, 1, 0);
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  for (int split = 0; split < 2; ++split) {
    for (int it = 0; it < 3; ++it) cgg_mask_logits(E, hi, split ? lo : nullptr, out, nullptr, B, Q, C, P, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev0, 0));
    const int N = 50;
    for (int it = 0; it < N; ++it) cgg_mask_logits(E, hi, split ? lo : nullptr, out, nullptr, B, Q, C, P, 0);
    CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1));
    float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
    double us = ms * 1e3 / N;
    double bytes = (double)B * ((double)C * P * (split ? 4 : 2) + Q * C * 4.0 + (double)Q * P * 4);
    printf("%s: %.1f us  %.0f GB/s  %.1f TF/s (%s)\n", split ? "split" : "bf16", us, bytes / us / 1e3, 2.0 * B * Q * C * P / us / 1e6, cgg_last_error_string());
  }
  // reference: plain streaming copy of the same byte mix (read 67 MB, write 52 MB)
  {
    const size_t nr = (size_t)B * T * 32 * C * 2 / 16, nw = (size_t)B * Q * P * 4 / 16;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(copy_mix, dim3(2048), dim3(256), 0, 0, (const uint4*)hi, (uint4*)out, nr, nw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev0, 0));
    for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(copy_mix, dim3(2048), dim3(256), 0, 0, (const uint4*)hi, (uint4*)out, nr, nw);
    CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1));
    float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
    printf("copy_mix: %.1f us  %.0f GB/s\n", ms * 1e3 / 50, (nr + nw) * 16.0 / (ms * 1e3 / 50) / 1e3);
  }
  return 0;
}
