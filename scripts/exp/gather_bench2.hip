// Micro-benchmark 2 (not part of the product): 64-B rows + separate 4-B w gather vs fused 128-B rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// LANES lanes x float4 per row (row = LANES*16 bytes); K rows per lane group; optional separate w gather
template <int LANES, int K, bool WITH_W>
__global__ __launch_bounds__(256) void gatherR(const float4* __restrict__ embed, const float* __restrict__ w,
                                               const int* __restrict__ idx, int64_t n, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t g = t / LANES;
  int c = (int)(t % LANES);
  if (g * K >= n) return;
  int id[K];
  float4 ev[K];
  float wv[K];
#pragma unroll
  for (int u = 0; u < K; ++u) id[u] = (g * K + u < n) ? idx[g * K + u] : 0;
#pragma unroll
  for (int u = 0; u < K; ++u) {
    ev[u] = embed[(int64_t)id[u] * LANES + c];
    wv[u] = (WITH_W && (u % LANES) == c) ? w[id[u]] : 0.f;
  }
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < K; ++u) acc += ev[u].x + ev[u].y + ev[u].z + ev[u].w + wv[u];
  for (int o = 1; o < LANES; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (c == 0) out[g] = acc;
}

template <typename Fn>
float time_graph(Fn launch, hipStream_t st, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  launch();
  CK(hipStreamSynchronize(st));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / reps;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int64_t V = 10000000;
  float4* embed; CK(hipMalloc(&embed, V * 256)); CK(hipMemset(embed, 0, V * 256));  // largest row variant = 256 B
  float* w; CK(hipMalloc(&w, V * 4)); CK(hipMemset(w, 0, V * 4));
  for (int64_t n : {212992LL, 851968LL, 3407872LL}) {
    std::vector<int> h(n);
    std::mt19937_64 rng(1);
    for (auto& x : h) x = (int)(rng() % (uint64_t)V);
    int* idx; CK(hipMalloc(&idx, n * 4)); CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, n * 4));
    printf("n=%lld lookups\n", (long long)n);
#define RUN(LANES, K, WW, label)                                                                              \
  {                                                                                                         \
    int64_t groups = (n + K - 1) / K;                                                                       \
    float us = time_graph([&]() { hipLaunchKernelGGL((gatherR<LANES, K, WW>), dim3((unsigned)((groups * LANES + 255) / 256)), dim3(256), 0, st, embed, w, idx, n, out); }, st, 20); \
    printf("  %-34s : %7.2f us  %6.1f Glookups/s\n", label, us, n / us / 1e3);                               \
  }
    RUN(4, 1, false, "64B rows, K=1")
    RUN(4, 4, false, "64B rows, K=4")
    RUN(4, 4, true, "64B rows + separate w, K=4")
    RUN(4, 13, true, "64B rows + separate w, K=13")
    RUN(8, 1, false, "128B rows, K=1")
    RUN(8, 4, false, "128B rows, K=4")
    RUN(8, 8, false, "128B rows, K=8")
    RUN(2, 4, false, "32B rows, K=4")
    RUN(16, 2, false, "256B rows, K=2")
    CK(hipFree(idx)); CK(hipFree(out));
  }
  return 0;
}
