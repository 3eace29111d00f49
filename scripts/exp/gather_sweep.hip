// Random 128-byte line reads without cache reuse, swept over the number of lookups per launch (the B sweep SURVEY.md
// section 7 asks for): n = B x 26 lookups, B = 8k .. 128k, fresh ids every launch (>= 512 MB of distinct lines between two
// uses of an id set), table 10M x 128 B = 1.28 GB.  Per n: the lookup rate of the bare read (sum to one float per group),
// for several (rows in flight per lane group, workgroup size) shapes.  Answers: is ~31 G lines/s at n = 213k a chip limit
// or the bandwidth-delay regime of a 7-us kernel?
// hipcc --offload-arch=gfx950 -O3 gather_sweep.hip -o gather_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LANES, int K>   // LANES x 16 B per row, K rows in flight per lane group
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

// read + write (a materialising gather): 128-B rows copied to a dense output
template <int K>
__global__ __launch_bounds__(256) void gatherW(const float4* __restrict__ tab, const int* __restrict__ idx, int64_t n,
                                               float4* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t g = t / 8;
  int c = (int)(t % 8);
  if (g * K >= n) return;
  int id[K];
  float4 v[K];
#pragma unroll
  for (int u = 0; u < K; ++u) id[u] = (g * K + u < n) ? idx[g * K + u] : 0;
#pragma unroll
  for (int u = 0; u < K; ++u) v[u] = tab[(int64_t)id[u] * 8 + c];
#pragma unroll
  for (int u = 0; u < K; ++u) if (g * K + u < n) out[(g * K + u) * 8 + c] = v[u];
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int64_t V = 10000000;
  float4* tab; CK(hipMalloc(&tab, V * 128)); CK(hipMemset(tab, 1, V * 128));
  const int64_t NMAX = 131072 * 26;
  const int64_t POOL = NMAX * 4;                       // ids drawn once; a launch takes a window of them
  std::vector<int> h(POOL);
  std::mt19937_64 rng(7);
  for (auto& x : h) x = (int)(rng() % (uint64_t)V);
  int* pool; CK(hipMalloc(&pool, POOL * 4)); CK(hipMemcpy(pool, h.data(), POOL * 4, hipMemcpyHostToDevice));
  float* out; CK(hipMalloc(&out, NMAX * 128));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int Bs[] = {8192, 16384, 32768, 65536, 131072};
  printf("{\"table\": \"10M x 128 B\", \"sweep\": [\n");
  bool first = true;
  for (int B : Bs) {
    const int64_t n = (int64_t)B * 26;
    const int nsets = (int)(POOL / n);                  // distinct windows: nsets * n * 128 B >= 1.7 GB of lines per graph
    auto run = [&](auto launch, const char* label, double bytes_per_lookup) -> int {
      hipGraph_t gr; hipGraphExec_t ge;
      for (int s = 0; s < nsets; ++s) launch(pool + (int64_t)s * n);
      CK(hipStreamSynchronize(st));
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < nsets; ++s) launch(pool + (int64_t)s * n);
      CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      float best = 1e9f, sum = 0.f;
      const int IT = 5;
      for (int it = 0; it < IT; ++it) {
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        sum += ms;
      }
      const float us = best * 1e3f / nsets;
      printf("%s  {\"B\": %d, \"n_lookups\": %lld, \"kernel\": \"%s\", \"us\": %.2f, \"G_lookups_per_s\": %.1f, "
             "\"TBps_lines\": %.2f}", first ? "" : ",\n", B, (long long)n, label, us, n / us / 1e3,
             n * bytes_per_lookup / us / 1e6);
      first = false;
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
      return 0;
    };
#define RUN(LANES, K, label)                                                                                              \
    run([&](const int* ix) { int64_t groups = (n + K - 1) / K;                                                            \
          hipLaunchKernelGGL((gatherR<LANES, K>), dim3((unsigned)((groups * LANES + 255) / 256)), dim3(256), 0, st, tab,  \
                             8, ix, n, out); }, label, 128.0)
    RUN(8, 1, "read 128B K=1");
    RUN(8, 2, "read 128B K=2");
    RUN(8, 4, "read 128B K=4");
    RUN(8, 8, "read 128B K=8");
    RUN(8, 13, "read 128B K=13");
    RUN(8, 26, "read 128B K=26");
    RUN(4, 4, "read first 64B of the line K=4");
    RUN(4, 13, "read first 64B of the line K=13");
    run([&](const int* ix) { int64_t groups = (n + 3) / 4;
          hipLaunchKernelGGL((gatherW<4>), dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256), 0, st, tab, ix, n,
                             (float4*)out); }, "read+write 128B K=4", 256.0);
  }
  printf("\n]}\n");
  return 0;
}
