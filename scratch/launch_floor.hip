#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1;} } while (0)
__global__ void empty_k(float* p) { extern __shared__ float sm[]; if (p && threadIdx.x == 9999) p[0] = sm[0]; }
__global__ __launch_bounds__(512, 2) void regs_k(float* p, int n) {
  float a[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) a[i] = p ? p[i * n] : (float)i;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 200; ++i) s += a[i] * (float)(i + threadIdx.x);
  if (p && threadIdx.x == 9999) p[0] = s;
}
int main() {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct { int grid, threads, lds; } cfgs[] = {{256, 512, 65536}, {256, 512, 0}, {256, 256, 65536}, {512, 512, 65536}, {2048, 256, 0}, {256, 512, 16384}};
  for (auto c : cfgs) {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_k, dim3(c.grid), dim3(c.threads), c.lds, 0, (float*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_k, dim3(c.grid), dim3(c.threads), c.lds, 0, (float*)nullptr);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("empty grid=%d threads=%d lds=%d : %.2f us/launch\n", c.grid, c.threads, c.lds, ms * 10);
  }
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(regs_k, dim3(256), dim3(512), 65536, 0, (float*)nullptr, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(regs_k, dim3(256), dim3(512), 65536, 0, (float*)nullptr, 1);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("regs_k 256x512 lds64k : %.2f us/launch\n", ms * 10);
  return 0;
}
