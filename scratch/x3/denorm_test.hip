// Does v_mfma_f32_32x32x16_f16 flush subnormal f16 inputs? Does v_cvt_f16_f32 / v_cvt_pkrtz produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void k(float aval, float bval, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = c[0];
    out[1] = (float)(_Float16)aval;
    auto p = __builtin_amdgcn_cvt_pkrtz(aval, aval);
    out[2] = (float)p[0];
  }
}
int main() {
  float* d; hipMalloc(&d, 64);
  float tests[][2] = {{9.5367431640625e-07f /*2^-20*/, 1024.f}, {5.9604644775390625e-08f /*2^-24*/, 16384.f}, {3.0e-6f, 100.f}, {1.0f, 1.0f}};
  for (auto& t : tests) {
    k<<<1, 64>>>(t[0], t[1], d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    double exact = 16.0 * (double)(float)(_Float16)t[0] * (double)t[1];
    printf("a=%.10g b=%g mfma=%.10g expect(16*f16(a)*b)=%.10g cvt=%.10g pkrtz=%.10g\n", t[0], t[1], h[0], exact, h[1], h[2]);
  }
  return 0;
}
