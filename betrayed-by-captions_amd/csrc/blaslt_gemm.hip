// Plain library GEMM with the epilogue the ATen bindings do not expose: D = act(A B^T + C + bias) in ONE hipBLASLt call
// (HIPBLASLT_EPILOGUE_RELU_BIAS with beta = 1). Used for the third 1x1 convolution of a BN-folded [3P] mmdet ResNet
// Bottleneck on channel-last bf16 activations: `relu(conv3(x) + identity)` -- torch.addmm can add the residual OR
// _addmm_activation the bias + ReLU, not both, which cost a separate 200-MB residual / ReLU pass per block.
// hipBLASLt is reached through dlopen of the copy the process already uses (the path comes from the host side), so
// there is one library instance, one kernel cache and no link-time dependency.
#include "cgg_common.h"

#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

namespace {

struct LtApi {
  void* dl = nullptr;
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
  hipblasLtHandle_t handle = nullptr;
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
};

struct LtPlan {
  hipblasLtMatmulDesc_t desc;
  hipblasLtMatrixLayout_t a, b, c, d;
  hipblasLtMatmulAlgo_t algo;
  size_t ws;
};

LtApi g_lt;
std::mutex g_lt_mutex;
int g_tune_candidates = 1;        // > 1: time that many heuristic candidates at first use of a shape (outside captures)
float g_last_tune_us[2] = {0.f, 0.f};   // (heuristic top-1, chosen) of the most recent tuning, for reporting
std::map<std::tuple<int, int, int, int, int>, LtPlan> g_plans;   // (M, N, K, relu + 2 has_bias, has_res)

template <typename F>
bool lt_sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_lt.dl, name));
  return f != nullptr;
}

}  // namespace

extern "C" int cgg_blaslt_init(const char* libpath) {
  std::lock_guard<std::mutex> lock(g_lt_mutex);
  if (g_lt.handle) return CGG_OK;
  g_lt.dl = dlopen(libpath && libpath[0] ? libpath : "libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
  CGG_REQUIRE(g_lt.dl != nullptr, CGG_EUNSUPPORTED, "cgg_blaslt_init: dlopen(%s) failed: %s",
              libpath ? libpath : "libhipblaslt.so", dlerror());
  const bool ok = lt_sym(g_lt.Create, "hipblasLtCreate") && lt_sym(g_lt.DescCreate, "hipblasLtMatmulDescCreate") &&
                  lt_sym(g_lt.DescSet, "hipblasLtMatmulDescSetAttribute") &&
                  lt_sym(g_lt.LayoutCreate, "hipblasLtMatrixLayoutCreate") &&
                  lt_sym(g_lt.PrefCreate, "hipblasLtMatmulPreferenceCreate") &&
                  lt_sym(g_lt.PrefSet, "hipblasLtMatmulPreferenceSetAttribute") &&
                  lt_sym(g_lt.Heuristic, "hipblasLtMatmulAlgoGetHeuristic") && lt_sym(g_lt.Matmul, "hipblasLtMatmul");
  CGG_REQUIRE(ok, CGG_EUNSUPPORTED, "cgg_blaslt_init: hipBLASLt entry points missing in %s", libpath ? libpath : "(default)");
  CGG_REQUIRE(g_lt.Create(&g_lt.handle) == HIPBLAS_STATUS_SUCCESS, CGG_ELIBRARY, "cgg_blaslt_init: hipblasLtCreate failed");
  g_lt.workspace_bytes = 32u << 20;
  if (hipMalloc(&g_lt.workspace, g_lt.workspace_bytes) != hipSuccess) {
    g_lt.workspace = nullptr;
    g_lt.workspace_bytes = 0;
  }
  return CGG_OK;
}

// y[M, N] = act(x[M, K] w[N, K]^T + bias[N] + res[M, N]); all bf16, row-major, f32 accumulation. res nullable.
extern "C" int cgg_gemm_bias_res_act_bf16(const void* x, const void* w, const void* bias, const void* res, void* y, int M,
                                          int N, int K, int relu, cgg_stream_t stream) {
  CGG_REQUIRE(x && w && y, CGG_EINVAL, "cgg_gemm_bias_res_act_bf16: null pointer");
  CGG_REQUIRE(bias || !relu, CGG_EUNSUPPORTED, "cgg_gemm_bias_res_act_bf16: ReLU without bias is not built");
  CGG_REQUIRE(M > 0 && N > 0 && K > 0, CGG_EINVAL, "cgg_gemm_bias_res_act_bf16: bad sizes");
  CGG_REQUIRE(g_lt.handle != nullptr, CGG_EINVAL, "cgg_gemm_bias_res_act_bf16: call cgg_blaslt_init first");
  CGG_REQUIRE(cgg_aligned16(x) && cgg_aligned16(w) && cgg_aligned16(res) && cgg_aligned16(y) && K % 8 == 0 && N % 8 == 0,
              CGG_EALIGN, "cgg_gemm_bias_res_act_bf16: 16-B aligned operands, K and N multiples of 8");
  LtPlan plan;
  {
    std::lock_guard<std::mutex> lock(g_lt_mutex);
    const auto key = std::make_tuple(M, N, K, (relu ? 1 : 0) + (bias ? 2 : 0), res ? 1 : 0);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
      // column-major view: D^T (N x M, ld N) = op_T(W as K x N, ld K) * (x^T as K x M, ld K)
      LtPlan p;
      bool ok = g_lt.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
      const int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
      ok = ok && g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS;
      const uint32_t epi = !bias ? HIPBLASLT_EPILOGUE_DEFAULT : (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS);
      ok = ok && g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS;
      const int32_t bias_type = HIP_R_16BF;
      ok = ok && g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_type, sizeof(bias_type)) ==
                     HIPBLAS_STATUS_SUCCESS;
      ok = ok && g_lt.LayoutCreate(&p.a, HIP_R_16BF, K, N, K) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && g_lt.LayoutCreate(&p.b, HIP_R_16BF, K, M, K) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && g_lt.LayoutCreate(&p.c, HIP_R_16BF, N, M, N) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && g_lt.LayoutCreate(&p.d, HIP_R_16BF, N, M, N) == HIPBLAS_STATUS_SUCCESS;
      CGG_REQUIRE(ok, CGG_ELIBRARY, "cgg_gemm_bias_res_act_bf16: hipBLASLt descriptor setup failed");
      // the heuristic needs the bias pointer attribute to be present to pick a bias-capable solution
      if (bias) ok = g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) == HIPBLAS_STATUS_SUCCESS;
      hipblasLtMatmulPreference_t pref;
      ok = ok && g_lt.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
      const uint64_t wsmax = g_lt.workspace_bytes;
      ok = ok && g_lt.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax)) == HIPBLAS_STATUS_SUCCESS;
      constexpr int MAXC = 32;
      hipblasLtMatmulHeuristicResult_t heur[MAXC];
      int found = 0;
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
      const int want = (cap == hipStreamCaptureStatusNone) ? (g_tune_candidates < 1 ? 1 : (g_tune_candidates > MAXC ? MAXC : g_tune_candidates)) : 1;
      ok = ok && g_lt.Heuristic(g_lt.handle, p.desc, p.a, p.b, p.c, p.d, pref, want, heur, &found) == HIPBLAS_STATUS_SUCCESS;
      CGG_REQUIRE(ok && found > 0, CGG_EUNSUPPORTED, "cgg_gemm_bias_res_act_bf16: no hipBLASLt solution for %dx%dx%d", M, N, K);
      int best = 0;
      if (found > 1) {
        // the library's ranking is a model; the shapes here are few and fixed, so measure: 2 warm + 5 timed launches each
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        const float alpha = 1.f, beta = res ? 1.f : 0.f;
        float best_ms = 1e30f;
        for (int c = 0; c < found; ++c) {
          if (heur[c].state != HIPBLAS_STATUS_SUCCESS || heur[c].workspaceSize > g_lt.workspace_bytes) continue;
          bool good = true;
          for (int it = 0; it < 7 && good; ++it) {
            if (it == 2) (void)hipEventRecord(e0, (hipStream_t)stream);
            good = g_lt.Matmul(g_lt.handle, p.desc, &alpha, w, p.a, x, p.b, &beta, res ? res : y, p.c, y, p.d, &heur[c].algo,
                               g_lt.workspace, g_lt.workspace_bytes, (hipStream_t)stream) == HIPBLAS_STATUS_SUCCESS;
          }
          (void)hipEventRecord(e1, (hipStream_t)stream);
          (void)hipEventSynchronize(e1);
          float ms = 0.f;
          if (!good || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
          if (c == 0) g_last_tune_us[0] = ms * 200.f;
          if (ms < best_ms) { best_ms = ms; best = c; }
        }
        g_last_tune_us[1] = best_ms * 200.f;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
      }
      p.algo = heur[best].algo;
      p.ws = heur[best].workspaceSize;
      it = g_plans.emplace(key, p).first;
    }
    plan = it->second;
    // the bias pointer is an attribute of the (shared) descriptor: set it under the lock, launch under the lock
    const bool ok = !bias || g_lt.DescSet(plan.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) == HIPBLAS_STATUS_SUCCESS;
    CGG_REQUIRE(ok, CGG_ELIBRARY, "cgg_gemm_bias_res_act_bf16: bias pointer");
    const float alpha = 1.f, beta = res ? 1.f : 0.f;
    const hipblasStatus_t st =
        g_lt.Matmul(g_lt.handle, plan.desc, &alpha, w, plan.a, x, plan.b, &beta, res ? res : y, plan.c, y, plan.d,
                    &plan.algo, g_lt.workspace, g_lt.workspace_bytes, (hipStream_t)stream);
    CGG_REQUIRE(st == HIPBLAS_STATUS_SUCCESS, CGG_ELIBRARY, "cgg_gemm_bias_res_act_bf16: hipblasLtMatmul status %d", (int)st);
  }
  return CGG_OK;
}

// n > 1: at the first (non-captured) use of a shape, time the first n heuristic candidates and keep the fastest.
extern "C" int cgg_blaslt_set_tuning(int n_candidates) {
  std::lock_guard<std::mutex> lock(g_lt_mutex);
  g_tune_candidates = n_candidates;
  return CGG_OK;
}

// (heuristic top-1 us, chosen us) of the most recent tuning run
extern "C" int cgg_blaslt_last_tuning(float* top1_us, float* chosen_us) {
  if (top1_us) *top1_us = g_last_tune_us[0];
  if (chosen_us) *chosen_us = g_last_tune_us[1];
  return CGG_OK;
}
