// DIN target attention (5.DIN/CustomLayers.py:163-180 DinActivationLayer, :256-282 masked sum pooling) and the
// row-wise pieces of DIN's final MLP (make_mlp_layer, :142-160: LayerNormalization, Dice / PReLU, softmax).
//
// The reference materialises, for every (example, time step), concat([q, q-k, k, vec(k q^T)]) of width 3D + D^2
// (9504 at D = 96) and pushes it through Dense(36): 280 GFLOP and 15.6 GB of intermediates per batch at config E.
// The ActivationUnit is bilinear in (q, k), so it is factorised instead:
//     pre[b,t,:] = c_b + k_t . Eff_b,   Eff_b = (W_k - W_d) + M_b,   M_b[i,o] = sum_j q_j W_o[i,j,o],
//     c_b = q (W_q + W_d) + b1
// M_b and c_b come from ONE fp32 MFMA GEMM per batch (q . [Wo_r | W_q+W_d]); the kernel below then does, per
// example, the gather of the T key rows, the [T,D] x [D,H] product against Eff_b held in LDS, the activation, the
// score, the (reference-quirk) mask and the weighted sum pooling -- keys are never written to HBM in the forward.
#include "common.h"
#include <math.h>

namespace {

constexpr float BN_EPS = 1e-3f;   // keras BatchNormalization epsilon (inside Dice)
constexpr float LN_EPS = 1e-3f;   // keras LayerNormalization epsilon

enum { DACT_NONE = 0, DACT_RELU = 1, DACT_SIGMOID = 2, DACT_TANH = 3, DACT_DICE = 4, DACT_PRELU = 5 };

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// y = act(x) with per-feature parameters; also d y/d x and d y/d alpha
__device__ __forceinline__ float feat_act(int kind, float x, float alpha, float mean, float var, float* dydx,
                                          float* dyda) {
  float y = x, dx = 1.f, da = 0.f;
  switch (kind) {
    case DACT_RELU: y = fmaxf(x, 0.f); dx = x > 0.f ? 1.f : 0.f; break;
    case DACT_SIGMOID: y = sigmoid_acc(x); dx = y * (1.f - y); break;
    case DACT_TANH: y = tanhf(x); dx = 1.f - y * y; break;
    case DACT_DICE: {
      // Dice (5.DIN/CustomLayers.py:193-196), BN(center=False, scale=False) in inference mode
      float r = rsqrtf(var + BN_EPS);
      float p = sigmoid_acc((x - mean) * r);
      y = alpha * (1.f - p) * x + p * x;
      float dp = p * (1.f - p) * r;
      dx = alpha * (1.f - p) + p + x * dp * (1.f - alpha);
      da = (1.f - p) * x;
      break;
    }
    case DACT_PRELU: y = x > 0.f ? x : alpha * x; dx = x > 0.f ? 1.f : alpha; da = x > 0.f ? 0.f : x; break;
    default: break;
  }
  if (dydx) *dydx = dx;
  if (dyda) *dyda = da;
  return y;
}

// ------------------------------------------------------------------------------------------------
// weight preparation: W1 [3D + D*D, H], b1 [H]  ->  Wcat [D, D*H + H] = [Wo_r | W_q + W_d],  Wkd [D,H] = W_k - W_d,
// bext [D*H + H] = [0 | b1], with Wo_r[j, i*H + o] = W1[3D + i*D + j, o].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void din_prep_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D,
                                                       int H, float* __restrict__ Wcat, float* __restrict__ Wkd,
                                                       float* __restrict__ bext) {
  int64_t N = (int64_t)D * H + H;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < (int64_t)D * N) {
    int64_t j = t / N, n = t - j * N;
    float v;
    if (n < (int64_t)D * H) {
      int64_t i = n / H, o = n - i * H;
      v = W1[(3 * (int64_t)D + i * D + j) * H + o];
    } else {
      int64_t o = n - (int64_t)D * H;
      v = W1[j * H + o] + W1[((int64_t)D + j) * H + o];
    }
    Wcat[t] = v;
  }
  if (t < (int64_t)D * H) {
    int64_t i = t / H, o = t - i * H;
    Wkd[t] = W1[(2 * (int64_t)D + i) * H + o] - W1[((int64_t)D + i) * H + o];
  }
  if (t < N) bext[t] = t < (int64_t)D * H ? 0.f : b1[t - (int64_t)D * H];
}

// gradient of the preparation: gW1 from gWcat [D, D*H+H] and gWkd [D,H]
__global__ __launch_bounds__(256) void din_prep_bwd_kernel(const float* __restrict__ gWcat, const float* __restrict__ gWkd,
                                                           int D, int H, float* __restrict__ gW1) {
  int64_t rows = 3 * (int64_t)D + (int64_t)D * D;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * H) return;
  int64_t r = t / H, o = t - r * H;
  int64_t N = (int64_t)D * H + H;
  float v;
  if (r < D) {                       // W_q
    v = gWcat[r * N + (int64_t)D * H + o];
  } else if (r < 2 * (int64_t)D) {   // W_d
    int64_t i = r - D;
    v = gWcat[i * N + (int64_t)D * H + o] - gWkd[i * H + o];
  } else if (r < 3 * (int64_t)D) {   // W_k
    v = gWkd[(r - 2 * (int64_t)D) * H + o];
  } else {                           // W_o[i,j,o]
    int64_t ij = r - 3 * (int64_t)D;
    int64_t i = ij / D, j = ij - i * D;
    v = gWcat[j * N + i * H + o];
  }
  gW1[t] = v;
}

// ------------------------------------------------------------------------------------------------
// attention forward / backward: one workgroup (4 waves) per example, waves take time steps round-robin.
// LDS (floats): Eff [D][H+1] | c [H] | kbuf [4][D] | red [4][D]            (+ backward: gEff [4][D][H+1] | gp [4][H])
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* embed; int64_t ld; int64_t V; int E; int C;
  const int64_t* series; int T;
  const float* Mext; const float* Wkd;
  int act; const float* alpha; const float* mean; const float* var;
  const float* w2; const float* b2;
  int64_t padding_index; int mask_valid;
};

__device__ __forceinline__ void load_key(const AttnArgs& a, int64_t b, int t, int lane, int D, float* kb, bool* bad) {
  for (int d = lane; d < D; d += 64) {
    int r = d / a.E, col = d - r * a.E;
    int64_t id = a.series[((int64_t)b * a.T + t) * a.C + r];
    float v = 0.f;
    if ((uint64_t)id < (uint64_t)a.V) v = a.embed[id * a.ld + col];
    else *bad = true;
    kb[d] = v;
  }
}

__global__ __launch_bounds__(256) void din_attn_fwd_kernel(AttnArgs a, int D, int H, float* __restrict__ scores,
                                                           float* __restrict__ pooled, int* oob) {
  extern __shared__ float lds[];
  const int HS = H + 1;
  float* Eff = lds;
  float* cvec = Eff + D * HS;
  float* kbuf = cvec + H;
  float* red = kbuf + 4 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int64_t NM = (int64_t)D * H + H;
  for (int i = tid; i < D * H; i += 256) {
    int r = i / H, o = i - r * H;
    Eff[r * HS + o] = a.Wkd[i] + a.Mext[b * NM + i];
  }
  for (int o = tid; o < H; o += 256) cvec[o] = a.Mext[b * NM + (int64_t)D * H + o];
  __syncthreads();
  float* kb = kbuf + wave * D;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // pooled dims lane, lane+64, lane+128, lane+192
  bool bad = false;
  const float al = (lane < H && a.alpha) ? a.alpha[lane] : 0.f;
  const float mu = (lane < H && a.mean) ? a.mean[lane] : 0.f;
  const float vr = (lane < H && a.var) ? a.var[lane] : 1.f;
  const float w2 = lane < H ? a.w2[lane] : 0.f;
  const float b2 = a.b2[0];
  for (int t0 = 0; t0 < a.T; t0 += 4) {
    int t = t0 + wave;
    bool live = t < a.T;
    if (live) load_key(a, b, t, lane, D, kb, &bad);
    __syncthreads();
    if (live) {
      float pre = 0.f;
      if (lane < H) {
        pre = cvec[lane];
        for (int i = 0; i < D; ++i) pre += kb[i] * Eff[i * HS + lane];
      }
      float h = lane < H ? feat_act(a.act, pre, al, mu, vr, nullptr, nullptr) : 0.f;
      float s = wave_sum64(h * w2) + b2;
      int64_t id0 = a.series[((int64_t)b * a.T + t) * a.C];
      bool pad = id0 == a.padding_index;
      float m = (a.mask_valid ? !pad : pad) ? 1.f : 0.f;   // reference quirk: mask = (id == padding)
      if (lane == 0) scores[b * a.T + t] = s;
      float ms = m * s;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int d = lane + 64 * j;
        if (d < D) acc[j] += ms * kb[d];
      }
    }
    __syncthreads();
  }
  if (bad && oob) *oob = 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int d = lane + 64 * j;
    if (d < D) red[wave * D + d] = acc[j];
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) pooled[b * D + d] = red[d] + red[D + d] + red[2 * D + d] + red[3 * D + d];
}

__global__ __launch_bounds__(256) void din_attn_bwd_kernel(AttnArgs a, int D, int H, const float* __restrict__ scores,
                                                           const float* __restrict__ gpooled,
                                                           float* __restrict__ gkeys /* [B,T,D] */,
                                                           float* __restrict__ gMext /* [B, D*H+H] */,
                                                           float* __restrict__ gw2p /* [B,H] */,
                                                           float* __restrict__ galphap /* [B,H] */,
                                                           float* __restrict__ gb2p /* [B] */) {
  extern __shared__ float lds[];
  const int HS = H + 1;
  float* Eff = lds;
  float* cvec = Eff + D * HS;
  float* kbuf = cvec + H;
  float* gpl = kbuf + 4 * D;          // g_pooled [D]
  float* gpb = gpl + D;               // gpre per wave [4][H]
  float* gEff = gpb + 4 * H;          // [4][D][HS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int64_t NM = (int64_t)D * H + H;
  for (int i = tid; i < D * H; i += 256) {
    int r = i / H, o = i - r * H;
    Eff[r * HS + o] = a.Wkd[i] + a.Mext[b * NM + i];
  }
  for (int o = tid; o < H; o += 256) cvec[o] = a.Mext[b * NM + (int64_t)D * H + o];
  for (int d = tid; d < D; d += 256) gpl[d] = gpooled[b * D + d];
  for (int i = tid; i < 4 * D * HS; i += 256) gEff[i] = 0.f;
  __syncthreads();
  float* kb = kbuf + wave * D;
  float* gp = gpb + wave * H;
  float* gE = gEff + wave * D * HS;
  bool bad = false;
  const float al = (lane < H && a.alpha) ? a.alpha[lane] : 0.f;
  const float mu = (lane < H && a.mean) ? a.mean[lane] : 0.f;
  const float vr = (lane < H && a.var) ? a.var[lane] : 1.f;
  const float w2 = lane < H ? a.w2[lane] : 0.f;
  float gc = 0.f, gw2 = 0.f, gal = 0.f, gb2 = 0.f;
  for (int t0 = 0; t0 < a.T; t0 += 4) {
    int t = t0 + wave;
    bool live = t < a.T;
    if (live) load_key(a, b, t, lane, D, kb, &bad);
    __syncthreads();
    float m = 0.f, s = 0.f, gs = 0.f;
    if (live) {
      int64_t id0 = a.series[((int64_t)b * a.T + t) * a.C];
      bool pad = id0 == a.padding_index;
      m = (a.mask_valid ? !pad : pad) ? 1.f : 0.f;
      s = scores[b * a.T + t];
      float dot = 0.f;
      for (int d = lane; d < D; d += 64) dot += gpl[d] * kb[d];
      gs = m * wave_sum64(dot);                          // d L / d score_t
      float pre = 0.f, dydx = 0.f, dyda = 0.f, h = 0.f;
      if (lane < H) {
        pre = cvec[lane];
        for (int i = 0; i < D; ++i) pre += kb[i] * Eff[i * HS + lane];
        h = feat_act(a.act, pre, al, mu, vr, &dydx, &dyda);
      }
      float gh = gs * w2;
      float gpre = gh * dydx;
      if (lane < H) {
        gw2 += gs * h;
        gal += gh * dyda;
        gc += gpre;
        gp[lane] = gpre;
        for (int i = 0; i < D; ++i) gE[i * HS + lane] += kb[i] * gpre;
      }
      if (lane == 0) gb2 += gs;
    }
    __syncthreads();
    if (live) {
      for (int d = lane; d < D; d += 64) {
        float g = m * s * gpl[d];
        for (int o = 0; o < H; ++o) g += Eff[d * HS + o] * gp[o];
        gkeys[((int64_t)b * a.T + t) * D + d] = g;
      }
    }
    __syncthreads();
  }
  // cross-wave reductions in a fixed order
  for (int i = tid; i < D * H; i += 256) {
    int r = i / H, o = i - r * H;
    int k = r * HS + o;
    gMext[b * NM + i] = gEff[k] + gEff[D * HS + k] + gEff[2 * D * HS + k] + gEff[3 * D * HS + k];
  }
  __syncthreads();
  float* red = gEff;                                     // reuse: [4][4][H]
  if (lane < H) {
    red[(wave * 4 + 0) * H + lane] = gc;
    red[(wave * 4 + 1) * H + lane] = gw2;
    red[(wave * 4 + 2) * H + lane] = gal;
    red[(wave * 4 + 3) * H + lane] = lane == 0 ? gb2 : 0.f;
  }
  __syncthreads();
  if (tid < H) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int wv = 0; wv < 4; ++wv) {
      s0 += red[(wv * 4 + 0) * H + tid];
      s1 += red[(wv * 4 + 1) * H + tid];
      s2 += red[(wv * 4 + 2) * H + tid];
    }
    gMext[b * NM + (int64_t)D * H + tid] = s0;
    gw2p[b * H + tid] = s1;
    galphap[b * H + tid] = s2;
  }
  if (tid == 0) gb2p[b] = red[3 * H] + red[(4 + 3) * H] + red[(8 + 3) * H] + red[(12 + 3) * H];
  (void)bad;
}

// ------------------------------------------------------------------------------------------------
// per-feature activations on [M,N], LayerNormalization, softmax (rows of N <= 1024: one wave per row)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void feat_act_fwd_kernel(int kind, const float* __restrict__ x, const float* alpha,
                                                           const float* mean, const float* var, float* __restrict__ y,
                                                           int64_t M, int N) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * N) return;
  int n = (int)(t % N);
  y[t] = feat_act(kind, x[t], alpha ? alpha[n] : 0.f, mean ? mean[n] : 0.f, var ? var[n] : 1.f, nullptr, nullptr);
}

__global__ __launch_bounds__(256) void feat_act_bwd_kernel(int kind, const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* alpha, const float* mean, const float* var,
                                                           float* __restrict__ gx, float* __restrict__ ga_elem, int64_t M,
                                                           int N) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * N) return;
  int n = (int)(t % N);
  float dx, da;
  feat_act(kind, x[t], alpha ? alpha[n] : 0.f, mean ? mean[n] : 0.f, var ? var[n] : 1.f, &dx, &da);
  float g = gy[t];
  gx[t] = g * dx;
  if (ga_elem) ga_elem[t] = g * da;
}

constexpr int LN_MAXJ = 16;

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int64_t M, int N,
                                                            float* __restrict__ y, float* __restrict__ xhat,
                                                            float* __restrict__ rstd) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[LN_MAXJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    v[j] = n < N ? x[row * N + n] : 0.f;
    s += v[j];
  }
  float mu = wave_sum64(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    float d = n < N ? v[j] - mu : 0.f;
    q += d * d;
  }
  float r = rsqrtf(wave_sum64(q) / (float)N + LN_EPS);
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    if (n < N) {
      float xh = (v[j] - mu) * r;
      if (xhat) xhat[row * N + n] = xh;
      y[row * N + n] = xh * gamma[n] + beta[n];
    }
  }
  if (rstd && lane == 0) rstd[row] = r;
}

// gx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = gy*gamma;  also gy*xhat per element (for d gamma)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            int64_t M, int N, float* __restrict__ gx,
                                                            float* __restrict__ gg_elem) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float g[LN_MAXJ], xh[LN_MAXJ];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    float gyv = n < N ? gy[row * N + n] : 0.f;
    xh[j] = n < N ? xhat[row * N + n] : 0.f;
    g[j] = n < N ? gyv * gamma[n] : 0.f;
    s1 += g[j];
    s2 += g[j] * xh[j];
    if (n < N && gg_elem) gg_elem[row * N + n] = gyv * xh[j];
  }
  s1 = wave_sum64(s1) / (float)N;
  s2 = wave_sum64(s2) / (float)N;
  float r = rstd[row];
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    if (n < N) gx[row * N + n] = r * (g[j] - s1 - xh[j] * s2);
  }
}

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, int64_t M, int N,
                                                          float* __restrict__ y) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[LN_MAXJ];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    v[j] = n < N ? x[row * N + n] : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    v[j] = n < N ? expf(v[j] - mx) : 0.f;
    s += v[j];
  }
  s = wave_sum64(s);
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int n = lane + 64 * j;
    if (n < N) y[row * N + n] = v[j] / s;
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, int64_t M,
                                                          int N, float* __restrict__ gx) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s += y[row * N + n] * gy[row * N + n];
  s = wave_sum64(s);
  for (int n = lane; n < N; n += 64) gx[row * N + n] = y[row * N + n] * (gy[row * N + n] - s);
}

size_t attn_lds_bytes(int D, int H, bool bwd) {
  size_t f = (size_t)D * (H + 1) + H + 4 * (size_t)D;
  if (bwd) f += (size_t)D + 4 * (size_t)H + 4 * (size_t)D * (H + 1);
  else f += 4 * (size_t)D;
  return f * sizeof(float);
}

bool attn_args_ok(int D, int H, int E, int C, int T) {
  return D > 0 && H > 0 && E > 0 && C > 0 && T > 0 && D == E * C && H <= 64 && D <= 256;
}

}  // namespace

extern "C" int rec_din_prepare_f32(const float* W1, const float* b1, int D, int H, float* Wcat, float* Wkd, float* bext,
                                   void* stream) {
  if (!W1 || !b1 || !Wcat || !Wkd || !bext || D <= 0 || H <= 0) return REC_E_ARG;
  int64_t total = (int64_t)D * ((int64_t)D * H + H);
  hipLaunchKernelGGL(din_prep_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, as_stream(stream), W1, b1, D,
                     H, Wcat, Wkd, bext);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_din_prepare_bwd_f32(const float* gWcat, const float* gWkd, int D, int H, float* gW1, void* stream) {
  if (!gWcat || !gWkd || !gW1 || D <= 0 || H <= 0) return REC_E_ARG;
  int64_t total = (3 * (int64_t)D + (int64_t)D * D) * H;
  hipLaunchKernelGGL(din_prep_bwd_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, as_stream(stream), gWcat,
                     gWkd, D, H, gW1);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_din_attn_fwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series,
                                    int64_t B, int T, const float* Mext, const float* Wkd, int H, int act,
                                    const float* alpha, const float* mean, const float* var, const float* w2,
                                    const float* b2, int64_t padding_index, int mask_valid, float* scores,
                                    float* pooled, int* oob_flag, void* stream) {
  int D = E * C;
  if (B < 0 || !attn_args_ok(D, H, E, C, T) || ld < E || V <= 0) return REC_E_ARG;
  if (act < DACT_NONE || act > DACT_PRELU) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!embed || !series || !Mext || !Wkd || !w2 || !b2 || !scores || !pooled) return REC_E_ARG;
  if ((act == DACT_DICE && (!alpha || !mean || !var)) || (act == DACT_PRELU && !alpha)) return REC_E_ARG;
  size_t lds = attn_lds_bytes(D, H, false);
  if (lds > 64 * 1024) return REC_E_UNSUPPORTED;
  AttnArgs a{embed, ld, V, E, C, series, T, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid};
  hipLaunchKernelGGL(din_attn_fwd_kernel, dim3((unsigned)B), dim3(256), lds, as_stream(stream), a, D, H, scores, pooled,
                     oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_din_attn_bwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series,
                                    int64_t B, int T, const float* Mext, const float* Wkd, int H, int act,
                                    const float* alpha, const float* mean, const float* var, const float* w2,
                                    const float* b2, int64_t padding_index, int mask_valid, const float* scores,
                                    const float* gpooled, float* gkeys, float* gMext, float* gw2p, float* galphap,
                                    float* gb2p, void* stream) {
  int D = E * C;
  if (B < 0 || !attn_args_ok(D, H, E, C, T) || ld < E || V <= 0) return REC_E_ARG;
  if (act < DACT_NONE || act > DACT_PRELU) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!embed || !series || !Mext || !Wkd || !w2 || !b2 || !scores || !gpooled || !gkeys || !gMext || !gw2p ||
      !galphap || !gb2p)
    return REC_E_ARG;
  size_t lds = attn_lds_bytes(D, H, true);
  if (lds > 150 * 1024) return REC_E_UNSUPPORTED;
  AttnArgs a{embed, ld, V, E, C, series, T, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid};
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(din_attn_bwd_kernel, dim3((unsigned)B), dim3(256), lds, as_stream(stream), a, D, H, scores, gpooled,
                     gkeys, gMext, gw2p, galphap, gb2p);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_feat_act_fwd_f32(int kind, const float* x, const float* alpha, const float* mean, const float* var,
                                    float* y, int64_t M, int N, void* stream) {
  if (M < 0 || N <= 0 || kind < DACT_NONE || kind > DACT_PRELU) return REC_E_ARG;
  if (M == 0) return REC_OK;
  if (!x || !y || (kind == DACT_DICE && (!alpha || !mean || !var)) || (kind == DACT_PRELU && !alpha)) return REC_E_ARG;
  hipLaunchKernelGGL(feat_act_fwd_kernel, dim3((unsigned)ceil_div64(M * N, 256)), dim3(256), 0, as_stream(stream), kind, x,
                     alpha, mean, var, y, M, N);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_feat_act_bwd_f32(int kind, const float* x, const float* gy, const float* alpha, const float* mean,
                                    const float* var, float* gx, float* ga_elem, int64_t M, int N, void* stream) {
  if (M < 0 || N <= 0 || kind < DACT_NONE || kind > DACT_PRELU) return REC_E_ARG;
  if (M == 0) return REC_OK;
  if (!x || !gy || !gx || (kind == DACT_DICE && (!alpha || !mean || !var)) || (kind == DACT_PRELU && !alpha))
    return REC_E_ARG;
  hipLaunchKernelGGL(feat_act_bwd_kernel, dim3((unsigned)ceil_div64(M * N, 256)), dim3(256), 0, as_stream(stream), kind, x,
                     gy, alpha, mean, var, gx, ga_elem, M, N);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, int64_t M, int N, float* y,
                                     float* xhat, float* rstd, void* stream) {
  if (M < 0 || N <= 0) return REC_E_ARG;
  if (N > 64 * LN_MAXJ) return REC_E_UNSUPPORTED;
  if (M == 0) return REC_OK;
  if (!x || !gamma || !beta || !y) return REC_E_ARG;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)ceil_div64(M, 4)), dim3(256), 0, as_stream(stream), x, gamma,
                     beta, M, N, y, xhat, rstd);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_layernorm_bwd_f32(const float* gy, const float* xhat, const float* rstd, const float* gamma, int64_t M,
                                     int N, float* gx, float* gg_elem, void* stream) {
  if (M < 0 || N <= 0) return REC_E_ARG;
  if (N > 64 * LN_MAXJ) return REC_E_UNSUPPORTED;
  if (M == 0) return REC_OK;
  if (!gy || !xhat || !rstd || !gamma || !gx) return REC_E_ARG;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)ceil_div64(M, 4)), dim3(256), 0, as_stream(stream), gy, xhat,
                     rstd, gamma, M, N, gx, gg_elem);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_softmax_fwd_f32(const float* x, int64_t M, int N, float* y, void* stream) {
  if (M < 0 || N <= 0) return REC_E_ARG;
  if (N > 64 * LN_MAXJ) return REC_E_UNSUPPORTED;
  if (M == 0) return REC_OK;
  if (!x || !y) return REC_E_ARG;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)ceil_div64(M, 4)), dim3(256), 0, as_stream(stream), x, M, N, y);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_softmax_bwd_f32(const float* y, const float* gy, int64_t M, int N, float* gx, void* stream) {
  if (M < 0 || N <= 0) return REC_E_ARG;
  if (M == 0) return REC_OK;
  if (!y || !gy || !gx) return REC_E_ARG;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)ceil_div64(M, 4)), dim3(256), 0, as_stream(stream), y, gy, M, N,
                     gx);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
