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

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane, by four v_add_f32 with DPP operands
// (quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_mirror) instead of four ds_bpermute round trips.  After the
// two quad steps every lane of a quad holds the quad's sum, so the mirror steps pair lanes that hold the SAME partial
// on each side: all 16 lanes end with bit-identical sums (a + b == b + a).
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_mov_f32<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov_f32<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov_f32<0x141>(v);    // row_half_mirror: lane i <-> 7 - i inside each half row
  v += dpp_mov_f32<0x140>(v);    // row_mirror: lane i <-> 15 - i
  return v;
}


// Keras Adam on one element with the operation sequence pinned: every multiply-add is an EXPLICIT fma (the compiler may
// fuse `a*b + c` or not, differently in every kernel; an explicit fma it can only keep) and what is left offers nothing to
// fuse.  Every kernel that applies a step -- the dense sweep, the touched-row kernels, the update inside the fused post
// launch, the catch-up of lazily evaluated rows -- therefore produces the same bits:
//   touched   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; x <- x - lr_t m / (sqrt(v) + eps)
//   untouched m <- b1 m            ; v <- b2 v              ; x <- x - lr_t m / (sqrt(v) + eps)   (Keras' dense sweep)
// The quotient: hardware square root and reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp each) and ONE pinned fma, not the
// correctly rounded sqrtf() and `/` (35 instruction slots per element and step against 14 -- and the lazily evaluated Adam
// replays ~47 steps per row it touches at 10M rows: that kernel is arithmetic-bound).  m and v are unaffected; the step
// lr_t m / (sqrt(v) + eps) is off by <= ~3 ulp of ITSELF, i.e. far below one ulp of x: x equals the correctly rounded
// evaluation in most steps and differs by one ulp of x in the rest (tests hold the tables to the oracle within 2e-5).
// -DREC_ADAM_IEEE restores the correctly rounded form in every kernel at once.
__device__ __forceinline__ float adam_step_x(float x, float m, float v, float lr_t, float eps) {
  const float num = lr_t * m;                        // a product that feeds a division: nothing to fuse
#ifdef REC_ADAM_IEEE
  const float den = sqrtf(v) + eps;
  return x - num / den;
#else
  const float den = __builtin_amdgcn_sqrtf(v) + eps;
  return __builtin_fmaf(-num, __builtin_amdgcn_rcpf(den), x);
#endif
}
__device__ __forceinline__ void adam_touch(float& x, float& m, float& v, float g, float lr_t, float b1, float b2,
                                           float eps) {
  m = __builtin_fmaf(g, 1.f - b1, m * b1);
  v = __builtin_fmaf(g * g, 1.f - b2, v * b2);
  x = adam_step_x(x, m, v, lr_t, eps);
}
// dense parameters (Keras' dense apply): m <- m + (g - m)(1 - b1) ; v <- v + (g^2 - v)(1 - b2)
__device__ __forceinline__ void adam_dense_elem(float& x, float& m, float& v, float g, float lr_t, float b1, float b2,
                                                float eps) {
  m = __builtin_fmaf(g - m, 1.f - b1, m);
  v = __builtin_fmaf(__builtin_fmaf(g, g, -v), 1.f - b2, v);
  x = adam_step_x(x, m, v, lr_t, eps);
}
__device__ __forceinline__ void adam_decay(float& x, float& m, float& v, float lr_t, float b1, float b2, float eps) {
  m = m * b1;
  v = v * b2;
  x = adam_step_x(x, m, v, lr_t, eps);
}

