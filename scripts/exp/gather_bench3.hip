// Random row reads WITHOUT cache reuse: every launch of the timed graph takes another set of ids (NSETS x 213k lookups x
// one 128-byte line = 870 MB of distinct lines per graph replay, well beyond L2 (32 MB) and the 256-MB memory-side
// cache), unlike gather_bench2.hip, which replays one id set and therefore measures cached reads for small n.
// hipcc --offload-arch=gfx950 -O3 gather_bench3.hip -o gather_bench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LANES, int K>   // LANES x 16 B per row, K rows per lane group
__global__ __launch_bounds__(256) void gatherR(const float4* __restrict__ tab, int row_f4, const int* __restrict__ idx,
                                               int64_t n, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t g = t / LANES;
  int c = (int)(t % LANES);
  if (g * K >= n) return;
  int id[K];
  float4 v[K];
#pragma unroll
  for (int u = 0; u < K; ++u) id[u] = (g * K + u < n) ? idx[g * K + u] : 0;
#pragma unroll
  for (int u = 0; u < K; ++u) v[u] = tab[(int64_t)id[u] * row_f4 + c];
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < K; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  for (int o = 1; o < LANES; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (c == 0) out[g] = acc;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int64_t V = 10000000, n = 212992;
  const int NSETS = 32;
  float4* tab; CK(hipMalloc(&tab, V * 128)); CK(hipMemset(tab, 0, V * 128));
  std::vector<int*> sets(NSETS);
  std::mt19937_64 rng(7);
  for (int s = 0; s < NSETS; ++s) {
    std::vector<int> h(n);
    for (auto& x : h) x = (int)(rng() % (uint64_t)V);
    CK(hipMalloc(&sets[s], n * 4)); CK(hipMemcpy(sets[s], h.data(), n * 4, hipMemcpyHostToDevice));
  }
  float* out; CK(hipMalloc(&out, n * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](auto launch, const char* label, bool rotate) -> int {
    hipGraph_t gr; hipGraphExec_t ge;
    for (int s = 0; s < NSETS; ++s) launch(sets[rotate ? s : 0]);
    CK(hipStreamSynchronize(st));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < NSETS; ++s) launch(sets[rotate ? s : 0]);
    CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    float us = best * 1e3f / NSETS;
    printf("  %-44s : %7.2f us  %6.1f Glookups/s  %6.2f TB/s of 128-B lines\n", label, us, n / us / 1e3, n * 128.0 / us / 1e6);
    return 0;
  };
#define RUN(LANES, K, ROWB, label)                                                                                        \
  for (int rot = 0; rot < 2; ++rot)                                                                                       \
    run([&](const int* ix) { int64_t groups = (n + K - 1) / K;                                                            \
          hipLaunchKernelGGL((gatherR<LANES, K>), dim3((unsigned)((groups * LANES + 255) / 256)), dim3(256), 0, st, tab,  \
                             ROWB / 16, ix, n, out); }, rot ? label " (fresh ids every launch)" : label " (same ids: cached)", rot)
  RUN(4, 4, 64, "64B rows K=4");
  RUN(8, 4, 128, "128B rows K=4");
  RUN(8, 13, 128, "128B rows K=13");
  RUN(4, 13, 64, "64B rows K=13");
  return 0;
}
