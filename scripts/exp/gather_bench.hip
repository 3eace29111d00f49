// Micro-benchmark (not part of the product): random 64-B row gather variants on MI355X.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gb scripts/exp/gather_bench.hip && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// variant A: LPE lanes per example, ids via LDS, NL loads in flight
template <int SPLIT, int NL, int MINW>
__global__ __launch_bounds__(256, MINW) void gatherA(const float4* __restrict__ embed, int64_t V,
                                                     const int64_t* __restrict__ idx, int64_t B, int F,
                                                     float* __restrict__ z) {
  constexpr int LPR = 4, LPE = LPR * SPLIT, EPW = 256 / LPE;
  extern __shared__ int ids_lds[];
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * EPW;
  const int n_ex = (B - b0 < EPW) ? (int)(B - b0) : EPW;
  for (int i = tid; i < n_ex * F; i += 256) ids_lds[i] = (int)idx[b0 * F + i];
  __syncthreads();
  const int e = tid / LPE, q = tid % LPE, c = q % LPR, s = q / LPR;
  if (e >= n_ex) return;
  const int FP = (F + SPLIT - 1) / SPLIT;
  const int f_begin = s * FP, f_end = (f_begin + FP < F) ? f_begin + FP : F;
  const int* my = ids_lds + e * F;
  float4 S = make_float4(0, 0, 0, 0), Q = make_float4(0, 0, 0, 0);
  for (int f0 = f_begin; f0 < f_end; f0 += NL) {
    int id[NL];
    float4 ev[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) id[u] = (f0 + u < f_end) ? my[f0 + u] : -1;
#pragma unroll
    for (int u = 0; u < NL; ++u) ev[u] = id[u] >= 0 ? embed[(int64_t)id[u] * LPR + c] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      S.x += ev[u].x; S.y += ev[u].y; S.z += ev[u].z; S.w += ev[u].w;
      Q.x += ev[u].x * ev[u].x; Q.y += ev[u].y * ev[u].y; Q.z += ev[u].z * ev[u].z; Q.w += ev[u].w * ev[u].w;
    }
  }
  float part = (S.x * S.x - Q.x) + (S.y * S.y - Q.y) + (S.z * S.z - Q.z) + (S.w * S.w - Q.w);
  for (int o = 1; o < LPE; o <<= 1) part += __shfl_xor(part, o, 64);
  if (q == 0) z[b0 + e] = part;
}

// variant B: flat -- one lane group (4 lanes) per LOOKUP, no per-example reduction (pure gather rate)
__global__ __launch_bounds__(256) void gatherB(const float4* __restrict__ embed, int64_t V,
                                               const int64_t* __restrict__ idx, int64_t n, float4* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * 4) return;
  int64_t r = t >> 2;
  int c = (int)(t & 3);
  out[t] = embed[idx[r] * 4 + c];
}

// variant C: like B but reduces 4 lookups per lane group and writes nothing but a checksum (no 13.6 MB store)
template <int K>
__global__ __launch_bounds__(256) void gatherC(const float4* __restrict__ embed, int64_t V,
                                               const int64_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t g = t >> 2;
  int c = (int)(t & 3);
  if (g * K >= n) return;
  int64_t id[K];
  float4 ev[K];
#pragma unroll
  for (int u = 0; u < K; ++u) id[u] = (g * K + u < n) ? idx[g * K + u] : 0;
#pragma unroll
  for (int u = 0; u < K; ++u) ev[u] = embed[id[u] * 4 + c];
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < K; ++u) acc += ev[u].x + ev[u].y + ev[u].z + ev[u].w;
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if (c == 0) out[g] = acc;
}

template <typename Fn>
float time_graph(Fn launch, hipStream_t st, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  launch();  // warm
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
  return best * 1e3f / reps;  // us per launch
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int F = 26;
  for (int64_t V : {1000000LL, 10000000LL, 40000000LL}) {
    float4* embed; CK(hipMalloc(&embed, V * 64));
    CK(hipMemset(embed, 0, V * 64));
    for (int64_t B : {8192LL, 32768LL, 131072LL}) {
      int64_t n = B * F;
      std::vector<int64_t> h(n);
      std::mt19937_64 rng(1);
      for (auto& x : h) x = (int64_t)(rng() % (uint64_t)V);
      int64_t* idx; CK(hipMalloc(&idx, n * 8)); CK(hipMemcpy(idx, h.data(), n * 8, hipMemcpyHostToDevice));
      float* z; CK(hipMalloc(&z, n * 4));
      float4* out; CK(hipMalloc(&out, n * 64));
      double mb = n * 72.0 / 1e6;
      printf("V=%lld B=%lld (%.1f MB ids+rows)\n", (long long)V, (long long)B, mb);
#define RUNA(SPLIT, NL, MINW)                                                                                   \
  {                                                                                                            \
    constexpr int EPW = 256 / (4 * SPLIT);                                                                     \
    float us = time_graph([&]() { hipLaunchKernelGGL((gatherA<SPLIT, NL, MINW>), dim3((unsigned)((B + EPW - 1) / EPW)), dim3(256), EPW * F * 4, st, embed, V, idx, B, F, z); }, st, 20); \
    printf("  A split=%d NL=%d minw=%d : %7.2f us  %7.1f GB/s\n", SPLIT, NL, MINW, us, mb / us * 1e3 / 1e3);     \
  }
      RUNA(1, 8, 1) RUNA(1, 26, 1) RUNA(2, 13, 1) RUNA(2, 16, 2) RUNA(4, 7, 1) RUNA(2, 13, 4)
      {
        float us = time_graph([&]() { hipLaunchKernelGGL(gatherB, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, st, embed, V, idx, n, out); }, st, 20);
        printf("  B flat gather+store        : %7.2f us  %7.1f GB/s (ids+rows) \n", us, mb / us);
      }
      {
        float us = time_graph([&]() { hipLaunchKernelGGL(gatherC<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, embed, V, idx, n, z); }, st, 20);
        printf("  C flat 4 rows/lane-group   : %7.2f us  %7.1f GB/s\n", us, mb / us);
        us = time_graph([&]() { hipLaunchKernelGGL(gatherC<13>, dim3((unsigned)(((n + 12) / 13 * 4 + 255) / 256)), dim3(256), 0, st, embed, V, idx, n, z); }, st, 20);
        printf("  C flat 13 rows/lane-group  : %7.2f us  %7.1f GB/s\n", us, mb / us);
        us = time_graph([&]() { hipLaunchKernelGGL(gatherC<1>, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, st, embed, V, idx, n, z); }, st, 20);
        printf("  C flat 1 row/lane-group    : %7.2f us  %7.1f GB/s\n", us, mb / us);
      }
      CK(hipFree(idx)); CK(hipFree(z)); CK(hipFree(out));
    }
    CK(hipFree(embed));
  }
  return 0;
}
