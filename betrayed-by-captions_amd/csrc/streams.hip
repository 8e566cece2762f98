// HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask) for the serving pipeline: the
// latency-bound decode stage gets its own CUs so that its 7-workgroup kernels are not queued behind the encode stage's
// full-device launches. The mask is given as 32-bit words, bit i = CU i enabled.
#include "cgg_common.h"

extern "C" int cgg_stream_create_cumask(const uint32_t* mask_host, int n_words, void** stream_out) {
  CGG_REQUIRE(mask_host && stream_out && n_words > 0, CGG_EINVAL, "cgg_stream_create_cumask: bad arguments");
  int enabled = 0;
  for (int i = 0; i < n_words; ++i) enabled += __builtin_popcount(mask_host[i]);
  CGG_REQUIRE(enabled >= 8, CGG_EINVAL, "cgg_stream_create_cumask: only %d CUs enabled", enabled);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_host);
  CGG_REQUIRE(e == hipSuccess, CGG_EUNSUPPORTED, "cgg_stream_create_cumask: %s", hipGetErrorString(e));
  *stream_out = (void*)s;
  return CGG_OK;
}

extern "C" int cgg_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
  return CGG_OK;
}
