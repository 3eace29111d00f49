// Row-local and elementwise kernels of the path (all HBM-bound; one pass over their operands):
//   CrossLayer vector mode fwd/bwd        3.DCN/CustomLayers.py:195-203      (K7)
//   two-tower cosine score fwd/bwd        2.FM/CustomLayers.py:233-234       (K10)
//   Keras BinaryCrossentropy fwd+bwd      2.FM/ModelManager.py:100,175       (K11)
//   Keras Adam dense / sparse (=dense sweep) / lazy rows   2.FM/ModelManager.py:104,178-179  (K12)
//   activation backward, column sums, axpby, column-block copies (dense backward plumbing)
#include "common.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------
// small elementwise helpers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(int act, const float* __restrict__ post,
                                                      const float* __restrict__ dpost, float* __restrict__ dpre,
                                                      int64_t n) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float y = post[t], g = dpost[t];
  switch (act) {
    case REC_ACT_RELU: g = y > 0.f ? g : 0.f; break;
    case REC_ACT_SIGMOID: g = g * y * (1.f - y); break;
    case REC_ACT_TANH: g = g * (1.f - y * y); break;
    default: break;
  }
  dpre[t] = g;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(int act, const float* __restrict__ x, const float* __restrict__ x2, float* __restrict__ y,
                                                      int64_t n) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float v = x[t];
  if (x2) v += x2[t];
  switch (act) {
    case REC_ACT_RELU: v = fmaxf(v, 0.f); break;
    case REC_ACT_SIGMOID: v = sigmoid_acc(v); break;
    case REC_ACT_TANH: v = tanhf(v); break;
    default: break;
  }
  y[t] = v;
}

// MatrixCrossLayer backward, elementwise part of one layer: H = G (.) X0 ; dX0 (+)= G (.) U
__global__ __launch_bounds__(256) void cross_mat_bwd_elem_kernel(const float* __restrict__ g, const float* __restrict__ x0,
                                                                 const float* __restrict__ u, float* __restrict__ h,
                                                                 float* __restrict__ gx0, int accumulate, int64_t n) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float gg = g[t];
  h[t] = gg * x0[t];
  float a = accumulate ? gx0[t] : 0.f;
  gx0[t] = a + gg * u[t];
}

// out[j] = sum_i X[i,j], two stages with a fixed summation order: stage 1 = (32 columns x 8 row lanes) per
// workgroup over one of `nrb` row blocks -> part[rb][N]; stage 2 adds the row blocks in order.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t M, int64_t N, int64_t ldx,
                                                     int64_t rows_per_block, float* __restrict__ part) {
  __shared__ float sh[8][33];
  int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  int64_t col = (int64_t)blockIdx.x * 32 + c;
  int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float acc = 0.f;
  if (col < N)
    for (int64_t i = r0 + r; i < r1; i += 8) acc += X[i * ldx + col];
  sh[r][c] = acc;
  __syncthreads();
  if (r == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += sh[q][c];
    part[(int64_t)blockIdx.y * N + col] = s;
  }
}

// second stage: 32 columns x 8 row slices per workgroup (slice r adds partials r, r+8, ...), slices combined in order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nrb, int64_t N,
                                                           float* __restrict__ out) {
  __shared__ float sh[8][33];
  int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  int64_t col = (int64_t)blockIdx.x * 32 + c;
  float acc = 0.f;
  if (col < N) {
#pragma unroll 4
    for (int q = r; q < nrb; q += 8) acc += part[(int64_t)q * N + col];
  }
  sh[r][c] = acc;
  __syncthreads();
  if (r == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += sh[q][c];
    out[col] = s;
  }
}

// Short matrices (M <= 1024 rows) in ONE launch: 32 columns x 32 row lanes per workgroup over all rows, four independent
// partial sums per lane, lanes and partials combined in a fixed order.  (Measured at M = 4096, N = 36..200 -- the bias
// gradients of the DIN step -- one launch with 128 dependent rounds per lane takes 13.5 us against 7.1 + 4.6 us for the
// two stages, whose first stage spreads the rows over 32 workgroups: the switch stays at 1024 rows.)
__global__ __launch_bounds__(1024) void colsum_onepass_kernel(const float* __restrict__ X, int64_t M, int64_t N,
                                                              int64_t ldx, float* __restrict__ out) {
  __shared__ float sh[32][33];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int64_t col = (int64_t)blockIdx.x * 32 + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < N) {
    const float* p = X + col;
    int64_t i = r;
    for (; i + 96 < M; i += 128) {
      a0 += p[i * ldx];
      a1 += p[(i + 32) * ldx];
      a2 += p[(i + 64) * ldx];
      a3 += p[(i + 96) * ldx];
    }
    for (; i < M; i += 32) a0 += p[i * ldx];
  }
  sh[r][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (r == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += sh[q][c];
    out[col] = s;
  }
}

// Both stages in ONE launch: every workgroup leaves its partial, and the one that arrives last at the column block's
// counter (integer atomics: exact) adds the partials of all row blocks -- in the same fixed order as colsum_final_kernel,
// so the result does not depend on which workgroup that is -- and puts the counter back to zero.  The bias gradients of
// the MLP / attention layers are 10-15 such sums per train step, each of them two launches at the launch floor before.
__global__ __launch_bounds__(256) void colsum_fused_kernel(const float* __restrict__ X, int64_t M, int64_t N, int64_t ldx,
                                                           int64_t rows_per_block, float* __restrict__ part, int nrb,
                                                           int* __restrict__ counters, float* __restrict__ out) {
  __shared__ float sh[8][33];
  __shared__ int last_s;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int64_t col = (int64_t)blockIdx.x * 32 + c;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float acc = 0.f;
  if (col < N)
    for (int64_t i = r0 + r; i < r1; i += 8) acc += X[i * ldx + col];
  sh[r][c] = acc;
  __syncthreads();
  // Partials and the counter are accessed with device-scope (write-through / cache-bypassing) operations, so no L2
  // write-back is needed to hand them from one workgroup to another: a release fence per workgroup costs an L2 flush
  // each, and thousands of them made the wide sums (DIN's [4096, 3492], CrossNet's [16384, 835]) slower than two launches.
  if (r == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += sh[q][c];
    __hip_atomic_store(&part[(int64_t)blockIdx.y * N + col], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_s_waitcnt(0);                     // this wave's partials have reached the coherent level
  __syncthreads();
  if (threadIdx.x == 0)
    last_s = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nrb - 1;
  __syncthreads();
  if (!last_s) return;                               // workgroup-uniform
  acc = 0.f;
  if (col < N) {
#pragma unroll 4
    for (int q = r; q < nrb; q += 8)
      acc += __hip_atomic_load(&part[(int64_t)q * N + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sh[r][c] = acc;
  __syncthreads();
  if (r == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += sh[q][c];
    out[col] = s;
  }
  if (threadIdx.x == 0) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void axpby_kernel(float a, const float* __restrict__ x, float b,
                                                    float* __restrict__ y, int64_t n) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  y[t] = (b == 0.f) ? a * x[t] : a * x[t] + b * y[t];
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int64_t lds_,
                                                        float* __restrict__ dst, int64_t ldd, int64_t rows,
                                                        int64_t w) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * w) return;
  int64_t r = t / w, c = t - r * w;
  dst[r * ldd + c] = src[r * lds_ + c];
}

// ------------------------------------------------------------------------------------------------
// K7 CrossNet vector mode: one wave per example row, the row lives in registers across all L layers
// (x0 read once, y written once).  lane owns dims lane, lane+64, ...; D <= 64*MAXJ.
// ------------------------------------------------------------------------------------------------
constexpr int MAXJ = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void cross_vec_fwd_kernel(const float* __restrict__ x0, int64_t B, int D, int L,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ y, float* __restrict__ xs) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float a0[MAXJ], xl[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    int d = lane + 64 * j;
    a0[j] = d < D ? x0[row * D + d] : 0.f;
    xl[j] = a0[j];
  }
  for (int l = 0; l < L; ++l) {
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      int d = lane + 64 * j;
      if (d < D) {
        if (xs) xs[((int64_t)l * B + row) * D + d] = xl[j];
        dot += xl[j] * w[l * D + d];
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      int d = lane + 64 * j;
      if (d < D) xl[j] = a0[j] * dot + b[l * D + d] + xl[j];
    }
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    int d = lane + 64 * j;
    if (d < D) y[row * D + d] = xl[j];
  }
}

// backward, one launch per layer (reverse order): rows are split over `nchunk` workgroups, each wave walks
// its rows with the layer's w in registers, updates the running gradient g and gx0 in place and keeps the
// dw/db partial sums of its rows in registers; a second kernel adds the per-wave partials in a fixed order.
__global__ __launch_bounds__(256) void cross_vec_bwd_layer_kernel(
    const float* __restrict__ x0, int64_t B, int D, const float* __restrict__ wl, const float* __restrict__ xl_saved,
    float* __restrict__ g /* in/out [B,D] */, float* __restrict__ gx0 /* accum [B,D] */,
    float* __restrict__ part /* [nchunk, 2, D] */, int rows_per_wg, int first_layer) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg;
  int64_t r_end = r_begin + rows_per_wg < B ? r_begin + rows_per_wg : B;
  float dw[MAXJ], db[MAXJ], wv[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    int d = lane + 64 * j;
    dw[j] = 0.f;
    db[j] = 0.f;
    wv[j] = d < D ? wl[d] : 0.f;
  }
  for (int64_t row = r_begin + wave; row < r_end; row += 4) {
    float gv[MAXJ], xv[MAXJ], a0[MAXJ];
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      int d = lane + 64 * j;
      bool ok = d < D;
      gv[j] = ok ? g[row * D + d] : 0.f;
      xv[j] = ok ? xl_saved[row * D + d] : 0.f;
      a0[j] = ok ? x0[row * D + d] : 0.f;
      s += xv[j] * wv[j];
      t += gv[j] * a0[j];
    }
    s = wave_sum(s);
    t = wave_sum(t);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      int d = lane + 64 * j;
      if (d < D) {
        dw[j] += t * xv[j];
        db[j] += gv[j];
        float acc = first_layer ? 0.f : gx0[row * D + d];
        gx0[row * D + d] = acc + gv[j] * s;
        g[row * D + d] = gv[j] + t * wv[j];
      }
    }
  }
  // per-wave partials -> global (wave-major); a second kernel adds them in a fixed order
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    int d = lane + 64 * j;
    if (d < D) {
      int64_t slot = (int64_t)blockIdx.x * 4 + wave;
      part[(slot * 2 + 0) * D + d] = dw[j];
      part[(slot * 2 + 1) * D + d] = db[j];
    }
  }
}

__global__ __launch_bounds__(256) void cross_vec_bwd_reduce_kernel(const float* __restrict__ part, int nslot, int D,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * D) return;
  int which = t / D, d = t - which * D;
  float acc = 0.f;
  for (int s = 0; s < nslot; ++s) acc += part[((int64_t)s * 2 + which) * D + d];
  (which == 0 ? dw : db)[d] = acc;
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                          int64_t n) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) y[t] += x[t];
}

// ------------------------------------------------------------------------------------------------
// K10 cosine
// ------------------------------------------------------------------------------------------------
template <int GW>
__global__ __launch_bounds__(256) void cosine_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         int64_t B, int d, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t row = t / GW;
  int c = (int)(t % GW);
  if (row >= B) return;
  float uu = 0.f, vv = 0.f, uv = 0.f;
  for (int k = c; k < d; k += GW) {
    float a = u[row * d + k], b = v[row * d + k];
    uu += a * a;
    vv += b * b;
    uv += a * b;
  }
  uu = group_sum<GW>(uu);
  vv = group_sum<GW>(vv);
  uv = group_sum<GW>(uv);
  if (c == 0) {
    float ru = rsqrtf(fmaxf(uu, 1e-12f)), rv = rsqrtf(fmaxf(vv, 1e-12f));
    out[row] = (1.f - uv * ru * rv) * 0.5f;
  }
}

template <int GW>
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         int64_t B, int d, const float* __restrict__ gout,
                                                         float* __restrict__ gu, float* __restrict__ gv) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t row = t / GW;
  int c = (int)(t % GW);
  if (row >= B) return;
  float uu = 0.f, vv = 0.f, uv = 0.f;
  for (int k = c; k < d; k += GW) {
    float a = u[row * d + k], b = v[row * d + k];
    uu += a * a;
    vv += b * b;
    uv += a * b;
  }
  uu = group_sum<GW>(uu);
  vv = group_sum<GW>(vv);
  uv = group_sum<GW>(uv);
  // out = (1 - c)/2, c = u_hat.v_hat ; dc/du = (v_hat - c u_hat)/|u|  (zero where the norm is clamped)
  float ru = rsqrtf(fmaxf(uu, 1e-12f)), rv = rsqrtf(fmaxf(vv, 1e-12f));
  float cs = uv * ru * rv;
  float gc = -0.5f * gout[row];
  bool cu = uu > 1e-12f, cv = vv > 1e-12f;
  for (int k = c; k < d; k += GW) {
    float a = u[row * d + k], b = v[row * d + k];
    float uh = a * ru, vh = b * rv;
    // clamped norm: u_hat = u * 1e6 is linear in u, d(u_hat)/du = 1e6 (no projection term)
    gu[row * d + k] = cu ? gc * (vh - cs * uh) * ru : gc * vh * ru;
    gv[row * d + k] = cv ? gc * (uh - cs * vh) * rv : gc * uh * rv;
  }
}

// ------------------------------------------------------------------------------------------------
// K11 BCE: one workgroup of 1024 threads, fixed-order tree => deterministic loss
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ y, const float* __restrict__ p, int64_t n,
                                                   float* __restrict__ loss, float* __restrict__ dp,
                                                   float* __restrict__ dz) {
  __shared__ float sh[16];
  const float eps = 1e-7f;
  float inv_n = 1.0f / (float)n;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    float yy = y[i], pp = p[i];
    float pc = fminf(fmaxf(pp, eps), 1.f - eps);
    acc += -(yy * logf(pc + eps) + (1.f - yy) * logf(1.f - pc + eps));
    if (dp || dz) {
      float inside = (pp >= eps && pp <= 1.f - eps) ? 1.f : 0.f;
      float g = -(yy / (pc + eps) - (1.f - yy) / (1.f - pc + eps)) * inside * inv_n;
      if (dp) dp[i] = g;
      if (dz) dz[i] = g * pp * (1.f - pp);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sh[q];
    if (loss) *loss = s * inv_n;
  }
}

// ------------------------------------------------------------------------------------------------
// L2 on the embedding rows a batch used (5.DIN/ModelManager.py:176-190): factor * l2_loss(table[unique ids]),
// l2_loss(x) = sum(x^2)/2.  rows_out[u,:] = factor * table[uniq_ids[u],:] (its gradient; zero on the padded tail),
// per-workgroup partial sums of squares in a fixed order, then one workgroup adds the partials in order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_rows_kernel(const float* __restrict__ table, int64_t ld, int E,
                                                      const int64_t* __restrict__ uniq_ids,
                                                      const int64_t* __restrict__ n_uniq, int64_t n, float factor,
                                                      float* __restrict__ rows_out, float* __restrict__ partial) {
  __shared__ float sh[4];
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float sq = 0.f;
  if (t < n * E) {
    int64_t u = t / E;
    int d = (int)(t - u * E);
    float x = 0.f;
    if (u < *n_uniq) x = table[uniq_ids[u] * ld + d];
    rows_out[t] = factor * x;
    sq = x * x;
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(1024) void l2_final_kernel(const float* __restrict__ partial, int64_t nb, float factor,
                                                        float* __restrict__ loss) {
  __shared__ float sh[16];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < nb; i += 1024) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sh[q];
    *loss = 0.5f * factor * s;
  }
}

// ------------------------------------------------------------------------------------------------
// K12 Adam
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_dense_kernel(float* __restrict__ var, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ g, int64_t n,
                                                         float lr_t, float b1, float b2, float eps) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float xx = var[t], mm = m[t], vv = v[t];
  adam_dense_elem(xx, mm, vv, g[t], lr_t, b1, b2, eps);
  m[t] = mm;
  v[t] = vv;
  var[t] = xx;
}

// touched rows: new (var, m, v) computed from the ORIGINAL state into `side` [cap, 3, E]
__global__ __launch_bounds__(256) void adam_rows_side_kernel(const float* __restrict__ var, const float* __restrict__ m,
                                                             const float* __restrict__ v, int64_t V, int E,
                                                             int64_t ld, const int64_t* __restrict__ ids,
                                                             const float* __restrict__ g, const int64_t* n_uniq,
                                                             int64_t cap, float* __restrict__ side, float lr_t,
                                                             float b1, float b2, float eps) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= cap * E) return;
  int64_t u = t / E;
  int d = (int)(t - u * E);
  if (u >= *n_uniq) return;
  int64_t id = ids[u];
  if ((uint64_t)id >= (uint64_t)V) return;
  float xx = var[id * ld + d], mm = m[id * E + d], vv = v[id * E + d];
  adam_touch(xx, mm, vv, g[t], lr_t, b1, b2, eps);
  side[(u * 3 + 0) * E + d] = xx;
  side[(u * 3 + 1) * E + d] = mm;
  side[(u * 3 + 2) * E + d] = vv;
}

// all rows: the untouched-row form of the dense sweep (streaming; float4 when E and ld allow).  var has row
// stride ld (fused-table layout), m and v are dense [V,E].
__global__ __launch_bounds__(256) void adam_sweep_vec_kernel(float* __restrict__ var, int64_t ld, int lpr,
                                                             float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                                             float lr_t, float b1, float b2, float eps) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (; t < n4; t += stride) {
    int64_t r = t / lpr;
    int c = (int)(t - r * lpr);
    float4* xp = reinterpret_cast<float4*>(var + r * ld + 4 * c);
    float4 mm = m[t], vv = v[t], x = *xp;
    adam_decay(x.x, mm.x, vv.x, lr_t, b1, b2, eps);
    adam_decay(x.y, mm.y, vv.y, lr_t, b1, b2, eps);
    adam_decay(x.z, mm.z, vv.z, lr_t, b1, b2, eps);
    adam_decay(x.w, mm.w, vv.w, lr_t, b1, b2, eps);
    m[t] = mm;
    v[t] = vv;
    *xp = x;
  }
}

__global__ __launch_bounds__(256) void adam_sweep_scalar_kernel(float* __restrict__ var, int64_t ld, int E,
                                                                float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                                float lr_t, float b1, float b2, float eps) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (; t < n; t += stride) {
    int64_t r = t / E;
    int d = (int)(t - r * E);
    float mm = m[t], vv = v[t], xx = var[r * ld + d];
    adam_decay(xx, mm, vv, lr_t, b1, b2, eps);
    m[t] = mm;
    v[t] = vv;
    var[r * ld + d] = xx;
  }
}

__global__ __launch_bounds__(256) void adam_rows_patch_kernel(float* __restrict__ var, float* __restrict__ m,
                                                              float* __restrict__ v, int64_t V, int E, int64_t ld,
                                                              const int64_t* __restrict__ ids, const int64_t* n_uniq,
                                                              int64_t cap, const float* __restrict__ side) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= cap * E) return;
  int64_t u = t / E;
  int d = (int)(t - u * E);
  if (u >= *n_uniq) return;
  int64_t id = ids[u];
  if ((uint64_t)id >= (uint64_t)V) return;
  var[id * ld + d] = side[(u * 3 + 0) * E + d];
  m[id * E + d] = side[(u * 3 + 1) * E + d];
  v[id * E + d] = side[(u * 3 + 2) * E + d];
}

__global__ __launch_bounds__(256) void adam_rows_lazy_kernel(float* __restrict__ var, float* __restrict__ m,
                                                             float* __restrict__ v, int64_t V, int E, int64_t ld,
                                                             const int64_t* __restrict__ ids, const float* __restrict__ g,
                                                             const int64_t* n_uniq, int64_t cap, float lr_t, float b1,
                                                             float b2, float eps) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= cap * E) return;
  int64_t u = t / E;
  int d = (int)(t - u * E);
  if (u >= *n_uniq) return;
  int64_t id = ids[u];
  if ((uint64_t)id >= (uint64_t)V) return;
  float xx = var[id * ld + d], mm = m[id * E + d], vv = v[id * E + d];
  adam_touch(xx, mm, vv, g[t], lr_t, b1, b2, eps);
  m[id * E + d] = mm;
  v[id * E + d] = vv;
  var[id * ld + d] = xx;
}

// ---- device-side step size: *step += 1, *lr_t = table[min(step, n) - 1].  The table holds Keras' bias-corrected step size
// of steps 1..n as the host computes it (adam_lr_t: float32 pow), so a graph-replayed train step uses bit for bit the
// values an eagerly enqueued one gets passed; beyond the table the correction factors are 1 in float32 (b2^t < 2^-24).
__global__ void adam_advance_kernel(int64_t* __restrict__ step, const float* __restrict__ tab, int64_t n,
                                    float* __restrict__ lr_t) {
  const int64_t t = *step + 1;
  *step = t;
  *lr_t = tab[(t < n ? t : n) - 1];
}

// Adam on several small dense parameters in ONE launch (the 7 dense parameters of DeepFM were 7 launches at the launch
// floor); arithmetic of adam_dense_kernel, step size from device memory
constexpr int ADAM_MULTI_MAX = 16;
struct AdamMulti {
  float* var[ADAM_MULTI_MAX]; float* m[ADAM_MULTI_MAX]; float* v[ADAM_MULTI_MAX]; const float* g[ADAM_MULTI_MAX];
  int64_t end[ADAM_MULTI_MAX];      // running element count: tensor i covers [end[i-1], end[i])
  int n;
};
__global__ __launch_bounds__(256) void adam_dense_multi_kernel(AdamMulti a, const float* __restrict__ lr_t_dev, float b1,
                                                               float b2, float eps) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.end[a.n - 1]) return;
  int i = 0;
  while (t >= a.end[i]) ++i;
  const int64_t e = t - (i ? a.end[i - 1] : 0);
  const float lr_t = *lr_t_dev;
  float xx = a.var[i][e], mm = a.m[i][e], vv = a.v[i][e];
  adam_dense_elem(xx, mm, vv, a.g[i][e], lr_t, b1, b2, eps);
  a.m[i][e] = mm;
  a.v[i][e] = vv;
  a.var[i][e] = xx;
}

inline float adam_lr_t(float lr, float b1, float b2, int64_t t) {
  // float32 arithmetic as Keras does (tf.pow on float32 scalars)
  float b1p = powf(b1, (float)t), b2p = powf(b2, (float)t);
  return lr * sqrtf(1.f - b2p) / (1.f - b1p);
}

}  // namespace

extern "C" int rec_version(void) { return 100; }

extern "C" int rec_act_bwd_f32(int act, const float* post, const float* dpost, float* dpre, int64_t n,
                               void* stream) {
  if (!post || !dpost || !dpre || n < 0 || act < REC_ACT_NONE || act > REC_ACT_TANH) return REC_E_ARG;
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), act, post,
                     dpost, dpre, n);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_act_fwd_f32(int act, const float* x, const float* x2, float* y, int64_t n, void* stream) {
  if (n < 0 || act < REC_ACT_NONE || act > REC_ACT_TANH) return REC_E_ARG;
  if (n == 0) return REC_OK;
  if (!x || !y) return REC_E_ARG;
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), act, x, x2, y, n);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_crossnet_mat_bwd_elem_f32(const float* g, const float* x0, const float* u, float* h, float* gx0,
                                             int accumulate, int64_t n, void* stream) {
  if (n < 0) return REC_E_ARG;
  if (n == 0) return REC_OK;
  if (!g || !x0 || !u || !h || !gx0) return REC_E_ARG;
  hipLaunchKernelGGL(cross_mat_bwd_elem_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), g,
                     x0, u, h, gx0, accumulate, n);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

static inline int colsum_row_blocks(int64_t M) {
  int64_t nrb = ceil_div64(M, 128);
  return (int)(nrb < 1 ? 1 : (nrb > 128 ? 128 : nrb));
}

extern "C" size_t rec_colsum_workspace_bytes(int64_t M, int64_t N) {
  return sizeof(float) * (size_t)colsum_row_blocks(M) * (size_t)(N > 0 ? N : 1);
}

extern "C" int rec_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, float* workspace,
                              void* stream) {
  if (!X || !out || M < 0 || N <= 0 || ldx < N) return REC_E_ARG;
  int nrb = colsum_row_blocks(M);
  if (nrb > 1 && !workspace) return REC_E_WORKSPACE;
  int64_t rpb = ceil_div64(M > 0 ? M : 1, nrb);
  hipStream_t st = as_stream(stream);
  if (M <= 1024) {
    hipLaunchKernelGGL(colsum_onepass_kernel, dim3((unsigned)ceil_div64(N, 32)), dim3(1024), 0, st, X, M, N, ldx, out);
    REC_LAUNCH_CHECK();
    return REC_OK;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div64(N, 32), (unsigned)nrb), dim3(256), 0, st, X, M, N, ldx,
                     rpb, nrb > 1 ? workspace : out);
  REC_LAUNCH_CHECK();
  if (nrb > 1) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div64(N, 32)), dim3(256), 0, st, workspace, nrb, N,
                       out);
    REC_LAUNCH_CHECK();
  }
  return REC_OK;
}

extern "C" int rec_colsum_fused_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, float* workspace,
                                    int* counters, void* stream) {
  if (!X || !out || !counters || M < 0 || N <= 0 || ldx < N) return REC_E_ARG;
  int nrb = colsum_row_blocks(M);
  if (!workspace) return REC_E_WORKSPACE;
  int64_t rpb = ceil_div64(M > 0 ? M : 1, nrb);
  hipLaunchKernelGGL(colsum_fused_kernel, dim3((unsigned)ceil_div64(N, 32), (unsigned)nrb), dim3(256), 0,
                     as_stream(stream), X, M, N, ldx, rpb, workspace, nrb, counters, out);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_axpby_f32(float a, const float* x, float b, float* y, int64_t n, void* stream) {
  if (!x || !y || n < 0) return REC_E_ARG;
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), a, x, b, y, n);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_copy_cols_f32(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows, int64_t w,
                                 void* stream) {
  if (!src || !dst || rows < 0 || w < 0 || lds_ < w || ldd < w) return REC_E_ARG;
  if (rows * w == 0) return REC_OK;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)ceil_div64(rows * w, 256)), dim3(256), 0, as_stream(stream), src,
                     lds_, dst, ldd, rows, w);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_crossnet_vec_fwd_f32(const float* x0, int64_t B, int D, int L, const float* w, const float* b,
                                        float* y, float* xs, void* stream) {
  if (!x0 || !w || !b || !y || B < 0 || D <= 0 || L < 0) return REC_E_ARG;
  if (D > 64 * MAXJ) return REC_E_UNSUPPORTED;
  if (B == 0) return REC_OK;
  hipLaunchKernelGGL(cross_vec_fwd_kernel, dim3((unsigned)ceil_div64(B, 4)), dim3(256), 0, as_stream(stream), x0, B, D,
                     L, w, b, y, xs);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

static inline int cross_bwd_chunks(int64_t B) {
  int64_t c = ceil_div64(B, 64);  // >= 64 rows per workgroup
  return (int)(c < 1 ? 1 : (c > 512 ? 512 : c));
}

extern "C" size_t rec_crossnet_vec_bwd_workspace_bytes(int64_t B, int D, int L) {
  (void)L;
  size_t g = sizeof(float) * (size_t)B * D;                               // running gradient g
  size_t part = sizeof(float) * (size_t)cross_bwd_chunks(B) * 4 * 2 * D;  // per-wave dw/db partials
  return g + part + 512;
}

extern "C" int rec_crossnet_vec_bwd_f32(const float* x0, int64_t B, int D, int L, const float* w, const float* xs,
                                        const float* gy, float* gx0, float* dw, float* db, void* workspace,
                                        void* stream) {
  if (!x0 || !w || !xs || !gy || !gx0 || !dw || !db || !workspace || B < 0 || D <= 0 || L < 0) return REC_E_ARG;
  if (D > 64 * MAXJ) return REC_E_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (B == 0) return REC_OK;
  float* g = (float*)workspace;
  float* part = g + (size_t)B * D;
  int nchunk = cross_bwd_chunks(B);
  int rows_per_wg = (int)ceil_div64(B, nchunk);
  hipError_t e = hipMemcpyAsync(g, gy, sizeof(float) * (size_t)B * D, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  if (L == 0) {
    e = hipMemcpyAsync(gx0, gy, sizeof(float) * (size_t)B * D, hipMemcpyDeviceToDevice, st);
    return (int)e;
  }
  for (int l = L - 1; l >= 0; --l) {
    hipLaunchKernelGGL(cross_vec_bwd_layer_kernel, dim3(nchunk), dim3(256), 0, st, x0, B, D, w + (size_t)l * D,
                       xs + (size_t)l * B * D, g, gx0, part, rows_per_wg, l == L - 1 ? 1 : 0);
    REC_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_vec_bwd_reduce_kernel, dim3((unsigned)ceil_div64(2 * D, 256)), dim3(256), 0, st, part,
                       nchunk * 4, D, dw + (size_t)l * D, db + (size_t)l * D);
    REC_LAUNCH_CHECK();
  }
  // x_0 is also the input of layer 0: gx0 += g
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)ceil_div64(B * D, 256)), dim3(256), 0, st, gx0, g,
                     (int64_t)B * D);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

#define COS_DISPATCH(KERN, ...)                                                                                  \
  do {                                                                                                           \
    if (d <= 1) hipLaunchKernelGGL(KERN<1>, dim3((unsigned)ceil_div64(B * 1, 256)), dim3(256), 0, st, __VA_ARGS__);      \
    else if (d <= 2) hipLaunchKernelGGL(KERN<2>, dim3((unsigned)ceil_div64(B * 2, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else if (d <= 4) hipLaunchKernelGGL(KERN<4>, dim3((unsigned)ceil_div64(B * 4, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else if (d <= 8) hipLaunchKernelGGL(KERN<8>, dim3((unsigned)ceil_div64(B * 8, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else if (d <= 16) hipLaunchKernelGGL(KERN<16>, dim3((unsigned)ceil_div64(B * 16, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else if (d <= 32) hipLaunchKernelGGL(KERN<32>, dim3((unsigned)ceil_div64(B * 32, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERN<64>, dim3((unsigned)ceil_div64(B * 64, 256)), dim3(256), 0, st, __VA_ARGS__);           \
  } while (0)

extern "C" int rec_cosine_fwd_f32(const float* u, const float* i, int64_t B, int d, float* out, void* stream) {
  if (!u || !i || !out || B < 0 || d <= 0) return REC_E_ARG;
  if (B == 0) return REC_OK;
  hipStream_t st = as_stream(stream);
  COS_DISPATCH(cosine_fwd_kernel, u, i, B, d, out);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_cosine_bwd_f32(const float* u, const float* i, int64_t B, int d, const float* gout, float* gu,
                                  float* gi, void* stream) {
  if (!u || !i || !gout || !gu || !gi || B < 0 || d <= 0) return REC_E_ARG;
  if (B == 0) return REC_OK;
  hipStream_t st = as_stream(stream);
  COS_DISPATCH(cosine_bwd_kernel, u, i, B, d, gout, gu, gi);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_bce_fwd_bwd_f32(const float* y, const float* p, int64_t n, float* loss, float* dp, float* dz,
                                   void* stream) {
  if (!y || !p || n <= 0) return REC_E_ARG;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, as_stream(stream), y, p, n, loss, dp, dz);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_adam_dense_f32(float* var, float* m, float* v, const float* g, int64_t n, int64_t t, float lr,
                                  float b1, float b2, float eps, void* stream) {
  if (!var || !m || !v || !g || n < 0 || t < 1) return REC_E_ARG;
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), var, m, v,
                     g, n, adam_lr_t(lr, b1, b2, t), b1, b2, eps);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" float rec_adam_lr_t_f32(float lr, float b1, float b2, int64_t t) { return adam_lr_t(lr, b1, b2, t); }

extern "C" int rec_adam_advance_f32(int64_t* step_dev, const float* lr_table, int64_t n_table, float* lr_t_dev,
                                    void* stream) {
  if (!step_dev || !lr_table || !lr_t_dev || n_table <= 0) return REC_E_ARG;
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), step_dev, lr_table, n_table, lr_t_dev);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_adam_dense_multi_f32(int n_tensors, float* const* var, float* const* m, float* const* v,
                                        const float* const* g, const int64_t* numel, const float* lr_t_dev, float b1,
                                        float b2, float eps, void* stream) {
  if (n_tensors <= 0 || !var || !m || !v || !g || !numel || !lr_t_dev) return REC_E_ARG;
  if (n_tensors > ADAM_MULTI_MAX) return REC_E_UNSUPPORTED;
  AdamMulti a{};
  int64_t tot = 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (!var[i] || !m[i] || !v[i] || !g[i] || numel[i] < 0) return REC_E_ARG;
    a.var[i] = var[i]; a.m[i] = m[i]; a.v[i] = v[i]; a.g[i] = g[i];
    tot += numel[i];
    a.end[i] = tot;
  }
  a.n = n_tensors;
  if (tot == 0) return REC_OK;
  hipLaunchKernelGGL(adam_dense_multi_kernel, dim3((unsigned)ceil_div64(tot, 256)), dim3(256), 0, as_stream(stream), a,
                     lr_t_dev, b1, b2, eps);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_adam_sparse_keras_f32(float* var, int64_t ld, float* m, float* v, int64_t V, int E,
                                         const int64_t* uniq_ids, const float* g_rows, const int64_t* n_uniq,
                                         int64_t cap, float* side, int64_t t, float lr, float b1, float b2, float eps,
                                         void* stream) {
  if (!var || !m || !v || !uniq_ids || !g_rows || !n_uniq || !side || V <= 0 || E <= 0 || ld < E || cap < 0 || t < 1)
    return REC_E_ARG;
  hipStream_t st = as_stream(stream);
  float lr_t = adam_lr_t(lr, b1, b2, t);
  if (cap > 0) {
    hipLaunchKernelGGL(adam_rows_side_kernel, dim3((unsigned)ceil_div64(cap * E, 256)), dim3(256), 0, st, var, m, v, V,
                       E, ld, uniq_ids, g_rows, n_uniq, cap, side, lr_t, b1, b2, eps);
    REC_LAUNCH_CHECK();
  }
  int64_t n = V * E;
  bool vec = E % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(var) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(m) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
  if (vec) {
    int64_t n4 = n / 4;
    int64_t blocks = ceil_div64(n4, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(adam_sweep_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, st, var, ld, E / 4, (float4*)m,
                       (float4*)v, n4, lr_t, b1, b2, eps);
  } else {
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(adam_sweep_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, var, ld, E, m, v, n, lr_t,
                       b1, b2, eps);
  }
  REC_LAUNCH_CHECK();
  if (cap > 0) {
    hipLaunchKernelGGL(adam_rows_patch_kernel, dim3((unsigned)ceil_div64(cap * E, 256)), dim3(256), 0, st, var, m, v, V,
                       E, ld, uniq_ids, n_uniq, cap, side);
    REC_LAUNCH_CHECK();
  }
  return REC_OK;
}

// Both tables of an FM-family layer in ONE sweep when they share the fused [embed(E) | w | pad] rows: lanes 0..E/4-1
// of a row take 16 bytes of embed each, lane E/4 takes w.  Swept separately, w (row stride ld, one float used per
// 128-byte line) costs a full line read + write per row: 0.52 ms of the 1.7-ms train step at 10M rows.
__global__ __launch_bounds__(256) void adam_sweep_pair_kernel(float* __restrict__ base, int64_t ld, int lpr,
                                                              float4* __restrict__ m_e, float4* __restrict__ v_e,
                                                              float* __restrict__ m_w, float* __restrict__ v_w,
                                                              int64_t V, float lr_t, float b1, float b2, float eps) {
  const int lanes = lpr + 1;
  const int64_t n = V * lanes;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; t < n; t += stride) {
    int64_t r = t / lanes;
    int c = (int)(t - r * lanes);
    if (c < lpr) {
      float4* xp = reinterpret_cast<float4*>(base + r * ld + 4 * c);
      int64_t i = r * lpr + c;
      float4 mm = m_e[i], vv = v_e[i], x = *xp;
      adam_decay(x.x, mm.x, vv.x, lr_t, b1, b2, eps);
      adam_decay(x.y, mm.y, vv.y, lr_t, b1, b2, eps);
      adam_decay(x.z, mm.z, vv.z, lr_t, b1, b2, eps);
      adam_decay(x.w, mm.w, vv.w, lr_t, b1, b2, eps);
      m_e[i] = mm;
      v_e[i] = vv;
      *xp = x;
    } else {
      float* xp = base + r * ld + 4 * lpr;
      float mm = m_w[r], vv = v_w[r], xx = *xp;
      adam_decay(xx, mm, vv, lr_t, b1, b2, eps);
      m_w[r] = mm;
      v_w[r] = vv;
      *xp = xx;
    }
  }
}

extern "C" int rec_adam_sparse_keras_pair_f32(float* fused, int64_t ld, float* m_e, float* v_e, float* m_w, float* v_w,
                                              int64_t V, int E, const int64_t* uniq_ids, const float* g_e_rows,
                                              const float* g_w_rows, const int64_t* n_uniq, int64_t cap, float* side_e,
                                              float* side_w, int64_t t, float lr, float b1, float b2, float eps,
                                              void* stream) {
  if (!fused || !m_e || !v_e || !m_w || !v_w || !uniq_ids || !g_e_rows || !g_w_rows || !n_uniq || !side_e || !side_w ||
      V <= 0 || E <= 0 || (E & 3) != 0 || ld < E + 1 || (ld & 3) != 0 || cap < 0 || t < 1)
    return REC_E_ARG;
  if (((reinterpret_cast<uintptr_t>(fused) | reinterpret_cast<uintptr_t>(m_e) | reinterpret_cast<uintptr_t>(v_e)) & 15) != 0)
    return REC_E_ARG;
  hipStream_t st = as_stream(stream);
  float lr_t = adam_lr_t(lr, b1, b2, t);
  float* w = fused + E;
  if (cap > 0) {          // touched rows: new values from the ORIGINAL state into the side buffers, patched after the sweep
    hipLaunchKernelGGL(adam_rows_side_kernel, dim3((unsigned)ceil_div64(cap * E, 256)), dim3(256), 0, st, fused, m_e,
                       v_e, V, E, ld, uniq_ids, g_e_rows, n_uniq, cap, side_e, lr_t, b1, b2, eps);
    hipLaunchKernelGGL(adam_rows_side_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, st, w, m_w, v_w, V, 1,
                       ld, uniq_ids, g_w_rows, n_uniq, cap, side_w, lr_t, b1, b2, eps);
    REC_LAUNCH_CHECK();
  }
  int64_t blocks = ceil_div64(V * (E / 4 + 1), 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(adam_sweep_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, st, fused, ld, E / 4, (float4*)m_e,
                     (float4*)v_e, m_w, v_w, V, lr_t, b1, b2, eps);
  REC_LAUNCH_CHECK();
  if (cap > 0) {
    hipLaunchKernelGGL(adam_rows_patch_kernel, dim3((unsigned)ceil_div64(cap * E, 256)), dim3(256), 0, st, fused, m_e,
                       v_e, V, E, ld, uniq_ids, n_uniq, cap, side_e);
    hipLaunchKernelGGL(adam_rows_patch_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, st, w, m_w, v_w, V, 1,
                       ld, uniq_ids, n_uniq, cap, side_w);
    REC_LAUNCH_CHECK();
  }
  return REC_OK;
}

extern "C" int rec_adam_rows_f32(float* var, int64_t ld, float* m, float* v, int64_t V, int E, const int64_t* uniq_ids,
                                 const float* g_rows, const int64_t* n_uniq, int64_t cap, int64_t t, float lr, float b1,
                                 float b2, float eps, void* stream) {
  if (!var || !m || !v || !uniq_ids || !g_rows || !n_uniq || V <= 0 || E <= 0 || ld < E || cap < 0 || t < 1)
    return REC_E_ARG;
  if (cap == 0) return REC_OK;
  hipLaunchKernelGGL(adam_rows_lazy_kernel, dim3((unsigned)ceil_div64(cap * E, 256)), dim3(256), 0, as_stream(stream),
                     var, m, v, V, E, ld, uniq_ids, g_rows, n_uniq, cap, adam_lr_t(lr, b1, b2, t), b1, b2, eps);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" size_t rec_l2_rows_workspace_bytes(int64_t n, int E) {
  if (n <= 0 || E <= 0) return 256;
  return sizeof(float) * (size_t)ceil_div64(n * E, 256) + 256;
}

extern "C" int rec_l2_rows_f32(const float* table, int64_t ld, int64_t V, int E, const int64_t* uniq_ids,
                               const int64_t* n_uniq, int64_t n, float factor, float* rows_out, float* loss,
                               float* workspace, void* stream) {
  if (V <= 0 || E <= 0 || ld < E || n < 0 || !loss) return REC_E_ARG;
  hipStream_t st = as_stream(stream);
  if (n == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
  if (!table || !uniq_ids || !n_uniq || !rows_out || !workspace) return REC_E_ARG;
  int64_t nb = ceil_div64(n * E, 256);
  hipLaunchKernelGGL(l2_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, table, ld, E, uniq_ids, n_uniq, n, factor,
                     rows_out, workspace);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(l2_final_kernel, dim3(1), dim3(1024), 0, st, workspace, nb, factor, loss);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// tf.keras.layers.BatchNormalization on [B,N] (axis=-1, non-fused path: biased batch variance for both the
// normalisation and the moving average), as NFM applies it to [bi-interaction | continuous]
// (3.DCN/CustomLayers.py:466,504) and MLPLayer(is_batch_norm=True) after BiasAdd (2.FM/CustomLayers.py:78-79).
// N is a layer width (tens), B the batch, so the batch axis carries the parallelism: slabs of BN_ROWS rows produce
// per-column (mean, M2) partials (two passes over registers, like tf.nn.moments), one small workgroup merges them
// in a fixed order (Chan's parallel-variance update: no E[x^2]-E[x]^2 cancellation), the normalisation is
// elementwise.  Deterministic; workspace = rec_batchnorm_workspace_bytes(B, N).
// ------------------------------------------------------------------------------------------------
#define BN_RPT 8      // rows per thread

__host__ __device__ static inline int bn_cols_pow2(int N) {
  int p = 1;
  while (p < N && p < 256) p <<= 1;
  return p;
}

// merge (n_b, mean_b, M2_b) into (n, mean, M2)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& M2, float nb, float meanb, float M2b) {
  if (nb == 0.f) return;
  float nt = n + nb;
  float d = meanb - mean;
  mean += d * (nb / nt);
  M2 += M2b + d * d * (n * nb / nt);
  n = nt;
}

// grid (slabs, column tiles); thread = (row slot rs, column c) with NP = power of two >= min(N,256) columns per tile
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t ld_x, int64_t B, int N,
                                                         int NP, float* __restrict__ part) {
  __shared__ float sm[3][256];
  const int RS = 256 / NP;
  const int c = threadIdx.x % NP, rs = threadIdx.x / NP;
  const int col = blockIdx.y * NP + c;
  const int64_t r0 = (int64_t)blockIdx.x * RS * BN_RPT;
  float v[BN_RPT];
  float n = 0.f, s = 0.f;
#pragma unroll
  for (int k = 0; k < BN_RPT; ++k) {
    int64_t r = r0 + rs + (int64_t)k * RS;
    bool in = r < B && col < N;
    v[k] = in ? x[r * ld_x + col] : 0.f;
    if (in) { n += 1.f; s += v[k]; }
  }
  float mean = n > 0.f ? s / n : 0.f, M2 = 0.f;
#pragma unroll
  for (int k = 0; k < BN_RPT; ++k) {
    int64_t r = r0 + rs + (int64_t)k * RS;
    if (r < B && col < N) { float d = v[k] - mean; M2 += d * d; }
  }
  sm[0][threadIdx.x] = n; sm[1][threadIdx.x] = mean; sm[2][threadIdx.x] = M2;
  __syncthreads();
  if (rs == 0 && col < N) {
    for (int q = 1; q < RS; ++q) chan_merge(n, mean, M2, sm[0][q * NP + c], sm[1][q * NP + c], sm[2][q * NP + c]);
    float* o = part + ((int64_t)blockIdx.x * N + col) * 3;
    o[0] = n; o[1] = mean; o[2] = M2;
  }
}

// one workgroup per column: merge the slab partials (fixed order: strided per thread, then a tree over threads)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ part, int n_slabs, int N, float eps,
                                                       float momentum, int training, float* __restrict__ moving_mean,
                                                       float* __restrict__ moving_var, float* __restrict__ stats) {
  __shared__ float sm[3][256];
  const int col = blockIdx.x;
  float mean, var;
  if (training) {
    float n = 0.f, m = 0.f, M2 = 0.f;
    for (int sidx = threadIdx.x; sidx < n_slabs; sidx += 256) {
      const float* p = part + ((int64_t)sidx * N + col) * 3;
      chan_merge(n, m, M2, p[0], p[1], p[2]);
    }
    sm[0][threadIdx.x] = n; sm[1][threadIdx.x] = m; sm[2][threadIdx.x] = M2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) {
        chan_merge(n, m, M2, sm[0][threadIdx.x + o], sm[1][threadIdx.x + o], sm[2][threadIdx.x + o]);
        sm[0][threadIdx.x] = n; sm[1][threadIdx.x] = m; sm[2][threadIdx.x] = M2;
      }
      __syncthreads();
    }
    mean = sm[1][0];
    var = sm[2][0] / sm[0][0];
    if (threadIdx.x == 0) {
      moving_mean[col] = moving_mean[col] * momentum + mean * (1.f - momentum);
      moving_var[col] = moving_var[col] * momentum + var * (1.f - momentum);
    }
  } else {
    mean = moving_mean[col];
    var = moving_var[col];
  }
  if (threadIdx.x == 0) {
    stats[col] = mean;
    stats[N + col] = 1.0f / sqrtf(var + eps);
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t ld_x, int64_t B, int N,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, float* __restrict__ xhat,
                                                       float* __restrict__ rstd_out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < N && rstd_out) rstd_out[t] = stats[N + t];
  if (t >= B * N) return;
  int64_t r = t / N;
  int c = (int)(t - r * N);
  float h = (x[r * ld_x + c] - stats[c]) * stats[N + c];
  if (xhat) xhat[t] = h;
  y[t] = h * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
}

// backward partials: per slab and column  sum g  and  sum g*xhat
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ g, const float* __restrict__ xhat,
                                                             int64_t B, int N, int NP, float* __restrict__ part) {
  __shared__ float sm[2][256];
  const int RS = 256 / NP;
  const int c = threadIdx.x % NP, rs = threadIdx.x / NP;
  const int col = blockIdx.y * NP + c;
  const int64_t r0 = (int64_t)blockIdx.x * RS * BN_RPT;
  float s = 0.f, sh = 0.f;
#pragma unroll
  for (int k = 0; k < BN_RPT; ++k) {
    int64_t r = r0 + rs + (int64_t)k * RS;
    if (r < B && col < N) {
      float gv = g[r * N + col];
      s += gv;
      sh += gv * xhat[r * N + col];
    }
  }
  sm[0][threadIdx.x] = s; sm[1][threadIdx.x] = sh;
  __syncthreads();
  if (rs == 0 && col < N) {
    for (int q = 1; q < RS; ++q) { s += sm[0][q * NP + c]; sh += sm[1][q * NP + c]; }
    float* o = part + ((int64_t)blockIdx.x * N + col) * 2;
    o[0] = s; o[1] = sh;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ part, int n_slabs, int N,
                                                           float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                           float* __restrict__ sums) {
  __shared__ float sm[2][256];
  const int col = blockIdx.x;
  float s = 0.f, sh = 0.f;
  for (int sidx = threadIdx.x; sidx < n_slabs; sidx += 256) {
    const float* p = part + ((int64_t)sidx * N + col) * 2;
    s += p[0];
    sh += p[1];
  }
  sm[0][threadIdx.x] = s; sm[1][threadIdx.x] = sh;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + o];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[col] = sm[0][0];
    sums[N + col] = sm[1][0];
    if (gbeta) gbeta[col] = sm[0][0];
    if (ggamma) ggamma[col] = sm[1][0];
  }
}

// training: gx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat));  inference: gx = gamma*rstd*g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, int64_t B, int N,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, int training,
                                                           float* __restrict__ gx) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= B * N) return;
  int c = (int)(t % N);
  float k = (gamma ? gamma[c] : 1.f) * rstd[c];
  float ms = training ? sums[c] / (float)B : 0.f, msh = training ? sums[N + c] / (float)B : 0.f;
  gx[t] = k * (g[t] - ms - xhat[t] * msh);
}

static inline int64_t bn_slabs(int64_t B, int N) {
  int rs = 256 / bn_cols_pow2(N);
  return ceil_div64(B, (int64_t)rs * BN_RPT);
}

extern "C" size_t rec_batchnorm_workspace_bytes(int64_t B, int N) {
  if (B <= 0 || N <= 0) return 256;
  return sizeof(float) * ((size_t)bn_slabs(B, N) * N * 3 + 2 * (size_t)N) + 256;
}

extern "C" int rec_batchnorm_fwd_f32(const float* x, int64_t ld_x, int64_t B, int N, const float* gamma,
                                     const float* beta, float eps, float momentum, int training, float* moving_mean,
                                     float* moving_var, float* y, float* xhat, float* rstd, void* workspace,
                                     void* stream) {
  if (B < 0 || N <= 0 || ld_x < N) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!x || !moving_mean || !moving_var || !y || !workspace) return REC_E_ARG;
  hipStream_t st = as_stream(stream);
  const int NP = bn_cols_pow2(N);
  const int64_t slabs = bn_slabs(B, N);
  float* part = static_cast<float*>(workspace);
  float* stats = part + slabs * N * 3;
  if (training) {
    hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)slabs, (unsigned)((N + NP - 1) / NP)), dim3(256), 0, st, x,
                       ld_x, B, N, NP, part);
    REC_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)N), dim3(256), 0, st, part, (int)slabs, N, eps, momentum, training,
                     moving_mean, moving_var, stats);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div64(B * N, 256)), dim3(256), 0, st, x, ld_x, B, N, stats,
                     gamma, beta, y, xhat, rstd);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_batchnorm_bwd_f32(const float* g, const float* xhat, const float* rstd, int64_t B, int N,
                                     const float* gamma, int training, float* gx, float* ggamma, float* gbeta,
                                     void* workspace, void* stream) {
  if (B < 0 || N <= 0) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!g || !xhat || !rstd || !gx || !workspace) return REC_E_ARG;
  hipStream_t st = as_stream(stream);
  const int NP = bn_cols_pow2(N);
  const int64_t slabs = bn_slabs(B, N);
  float* part = static_cast<float*>(workspace);
  float* sums = part + slabs * N * 3;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((unsigned)slabs, (unsigned)((N + NP - 1) / NP)), dim3(256), 0, st, g,
                     xhat, B, N, NP, part);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((unsigned)N), dim3(256), 0, st, part, (int)slabs, N, ggamma, gbeta, sums);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div64(B * N, 256)), dim3(256), 0, st, g, xhat, rstd, B, N,
                     gamma, sums, training, gx);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
