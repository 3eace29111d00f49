// Shared helpers for the gfx950 kernels of libmi355rec.so.  Wave = 64 lanes (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mi355rec.h"

#define REC_WAVE 64

#define REC_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e_ = hipGetLastError();          \
    if (e_ != hipSuccess) return (int)e_;       \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave-level sum over a power-of-two group of `width` adjacent lanes (xor butterfly: every lane gets the sum)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// accurate form used where parity at 1e-6 matters (logits can be large)
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
