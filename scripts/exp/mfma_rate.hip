// EXPERIMENT: issue rate of dependent v_mfma_f32_32x32x2_f32 chains, 1 wave per SIMD (256 WGs x 256 threads), for short
// (N=52) and long (N=5200) chains, back-to-back launches.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void chain(float* out, int n, float a, float b) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 1.2345e30f) out[0] = s;
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs : {256, 512}) {
    for (int n : {52, 520, 5200, 52000}) {
      int reps = n < 1000 ? 200 : 20;
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, 0, d, n, 1.0f, 1e-9f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, 0, d, n, 1.0f, 1e-9f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double us = ms * 1e3 / reps;
      printf("wgs %d  n %6d  %9.2f us/launch  %7.2f ns per MFMA (incl. launch)  -> %.2f GHz if 64 cyc\n", wgs, n, us,
             us * 1e3 / n, 64.0 * n / (us * 1e3));
    }
  }
  return 0;
}
