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
#include <stdlib.h>
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
// attention forward / backward: one workgroup (4 waves) per example.  The time axis is cut into chunks of TC = 64 steps
// whose key rows are gathered into LDS (KT, zero-padded); the [T,D] x [D,H] product against Eff_b and, in the backward,
// gEff = K^T . gpre and gkeys = gpre . Eff^T run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: lane l holds
// A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D: col = l&15, row = 4*(l>>4) + reg).  Wave w owns rows 16w..16w+15 of
// the chunk for everything that is per time step; all dimensions are padded to multiples of 16 with zeros in LDS, so
// padded rows / columns contribute exact zeros.  Keys are never written to HBM in the forward.
// LDS (floats): Eff [Dp][HS] | cvec [Hp] | KT [TC][DS] | msb [TC] | maskb [TC] | red [4][Dp]
//               backward adds: GP [TC][HS] | gsb [TC] | gpl [Dp] | redh [4][4][Hp]
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TC = 64;

struct AttnArgs {
  const float* embed; int64_t ld; int64_t V; int E; int C;
  const int64_t* series; int T;
  const float* Mext; const float* Wkd;
  int act; const float* alpha; const float* mean; const float* var;
  const float* w2; const float* b2;
  int64_t padding_index; int mask_valid;
  int stop;      // diagnostics only (REC_DIN_STOP, scripts/exp/din_{fwd,bwd}_phases.sh): forward 1-4 = leave after that
                 // phase, backward 12-14 = skip the later steps; 0 = run everything
};

struct AttnLds {
  int Dp, Hp, DS, HS;
  size_t eff, cvec, kt, msb, maskb, red, gp, gsb, gpl, redh, total;   // offsets in floats
};

__host__ __device__ inline AttnLds attn_layout(int D, int H, bool bwd) {
  AttnLds L;
  L.Dp = (D + 15) & ~15;
  L.Hp = (H + 15) & ~15;
  L.DS = L.Dp + 1;
  L.HS = L.Hp + 1;
  size_t o = 0;
  L.eff = o; o += (size_t)L.Dp * L.HS;
  L.cvec = o; o += L.Hp;
  L.kt = o; o += (size_t)TC * L.DS;
  L.msb = o; o += TC;
  L.maskb = o; o += TC;
  L.red = o; o += 4 * (size_t)L.Dp;
  L.gp = o; if (bwd) o += (size_t)TC * L.HS;
  L.gsb = o; if (bwd) o += TC;
  L.gpl = o; if (bwd) o += L.Dp;
  L.redh = o; if (bwd) o += 16 * (size_t)L.Hp;
  L.total = o;
  return L;
}

// Eff = Wkd + M_b (zero-padded), c_b; zero the padding columns of KT once
__device__ __forceinline__ void attn_load_eff(const AttnArgs& a, int64_t b, int D, int H, const AttnLds& L, float* lds) {
  float* Eff = lds + L.eff;
  float* cvec = lds + L.cvec;
  float* KT = lds + L.kt;
  const int64_t NM = (int64_t)D * H + H;
  const float* mrow = a.Mext + b * NM;
  const int DH = D * H;
  if ((DH & 3) == 0 && (NM & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(a.Mext) | reinterpret_cast<uintptr_t>(a.Wkd)) & 15) == 0) {
    // the row [Eff part D*H | c part H] is streamed as float4 (4 per lane in flight) and scattered into the padded tile
    for (int i = threadIdx.x; i < L.Dp * L.HS; i += 256) {           // padding entries first (disjoint from the rest)
      int r = i / L.HS, o = i - r * L.HS;
      if (r >= D || o >= H) Eff[i] = 0.f;
    }
    for (int o = H + threadIdx.x; o < L.Hp; o += 256) cvec[o] = 0.f;
    const float4* m4 = reinterpret_cast<const float4*>(mrow);
    const float4* w4 = reinterpret_cast<const float4*>(a.Wkd);
    const int n4 = (int)(NM >> 2), dh4 = DH >> 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * 4) {
      float4 mv[4], wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * 256;
        mv[u] = wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) {
          mv[u] = m4[i];
          if (i < dh4) wv[u] = w4[i];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * 256;
        if (i >= n4) continue;
        float e[4] = {mv[u].x + wv[u].x, mv[u].y + wv[u].y, mv[u].z + wv[u].z, mv[u].w + wv[u].w};
        int f = 4 * i;
        if (i < dh4) {
          int r = f / H, o = f - r * H;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            Eff[r * L.HS + o] = e[j];
            if (++o == H) { o = 0; ++r; }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) cvec[f - DH + j] = e[j];
        }
      }
    }
  } else {
    const int n_eff = L.Dp * L.HS;
    for (int i0 = threadIdx.x; i0 < n_eff; i0 += 256 * 6) {
      float wv[6], mv[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        int i = i0 + u * 256;
        wv[u] = mv[u] = 0.f;
        if (i < n_eff) {
          int r = i / L.HS, o = i - r * L.HS;
          if (r < D && o < H) { wv[u] = a.Wkd[r * H + o]; mv[u] = mrow[(int64_t)r * H + o]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        int i = i0 + u * 256;
        if (i < n_eff) Eff[i] = wv[u] + mv[u];
      }
    }
    for (int o = threadIdx.x; o < L.Hp; o += 256) cvec[o] = o < H ? mrow[(int64_t)DH + o] : 0.f;
  }
  const int padc = L.DS - D;                               // columns D .. DS-1 of every KT row stay zero
  for (int i = threadIdx.x; i < TC * padc; i += 256) KT[(i / padc) * L.DS + D + i % padc] = 0.f;
}

// key rows t0 .. t0+TC-1 -> KT (rows beyond T and rows of out-of-range ids: zeros), mask of every row -> maskb.
// One lane per 16 bytes of a row, six independent loads per lane issued before the first is consumed: a lane that
// walked a whole 128-byte row by itself (load, wait, store, eight times) made this phase eight dependent memory round
// trips per chunk, and with 2-3 workgroups per CU that latency WAS the kernel (forward 160 us at config E).
__device__ __forceinline__ void attn_gather(const AttnArgs& a, int64_t b, int t0, int D, const AttnLds& L, float* lds,
                                            bool* bad) {
  float* KT = lds + L.kt;
  float* maskb = lds + L.maskb;
  const int E = a.E, C = a.C;
  const int64_t* ser = a.series + (int64_t)b * a.T * C;
  for (int row = threadIdx.x; row < TC; row += 256) {
    int t = t0 + row;
    float m = 0.f;
    if (t < a.T) {
      bool pad = ser[(int64_t)t * C] == a.padding_index;
      m = (a.mask_valid ? !pad : pad) ? 1.f : 0.f;           // reference quirk: mask = (id == padding)
    }
    maskb[row] = m;
  }
  constexpr int NB = 6;
  if ((E & 3) == 0 && (a.ld & 3) == 0) {
    const int E4 = E >> 2, total = TC * C * E4;
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * NB) {
      float4 v[NB];
      int off[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        int i = i0 + u * 256;
        off[u] = -1;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total) {
          int rr = i / E4, e4 = i - rr * E4;
          int row = rr / C, r = rr - row * C;
          int t = t0 + row;
          off[u] = row * L.DS + r * E + 4 * e4;
          if (t < a.T) {
            int64_t id = ser[(int64_t)t * C + r];
            if ((uint64_t)id < (uint64_t)a.V) v[u] = *reinterpret_cast<const float4*>(a.embed + id * a.ld + 4 * e4);
            else *bad = true;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (off[u] < 0) continue;
        float* dst = KT + off[u];
        dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
      }
    }
  } else {
    const int total = TC * C * E;
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * NB) {
      float v[NB];
      int off[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        int i = i0 + u * 256;
        off[u] = -1;
        v[u] = 0.f;
        if (i < total) {
          int rr = i / E, e = i - rr * E;
          int row = rr / C, r = rr - row * C;
          int t = t0 + row;
          off[u] = row * L.DS + r * E + e;
          if (t < a.T) {
            int64_t id = ser[(int64_t)t * C + r];
            if ((uint64_t)id < (uint64_t)a.V) v[u] = a.embed[id * a.ld + e];
            else *bad = true;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u)
        if (off[u] >= 0) KT[off[u]] = v[u];
    }
  }
}

// pre-activation tile of this wave's 16 rows: acc[n][r] = sum_d KT[r0 + 4g + r][d] * Eff[d][16n + l15]
template <int NT>
__device__ __forceinline__ void attn_pre_gemm(const float* KT, const float* Eff, const AttnLds& L, int r0, int l15, int g,
                                              f32x4 (&acc)[NT]) {
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* ap = KT + (r0 + l15) * L.DS + g;
  const float* bp = Eff + g * L.HS + l15;
  const int nk = L.Dp >> 2;
  float ac = ap[0], bc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bc[n] = bp[16 * n];
  for (int k = 0; k + 1 < nk; ++k) {
    float an = ap[4 * (k + 1)], bn[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bn[n] = bp[4 * (k + 1) * L.HS + 16 * n];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, bc[n], acc[n], 0, 0, 0);
    ac = an;
#pragma unroll
    for (int n = 0; n < NT; ++n) bc[n] = bn[n];
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, bc[n], acc[n], 0, 0, 0);
}

template <int NT>
__global__ __launch_bounds__(256) void din_attn_fwd_kernel(AttnArgs a, int D, int H, float* __restrict__ scores,
                                                           float* __restrict__ pooled, int* oob) {
  extern __shared__ float lds[];
  const AttnLds L = attn_layout(D, H, false);
  float* Eff = lds + L.eff;
  float* cvec = lds + L.cvec;
  float* KT = lds + L.kt;
  float* msb = lds + L.msb;
  float* maskb = lds + L.maskb;
  float* red = lds + L.red;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, r0 = wave * 16;
  const int64_t b = blockIdx.x;
  attn_load_eff(a, b, D, H, L, lds);
  float al[NT], mu[NT], vr[NT], w2h[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int h = 16 * n + l15;
    bool ok = h < H;
    al[n] = (ok && a.alpha) ? a.alpha[h] : 0.f;
    mu[n] = (ok && a.mean) ? a.mean[h] : 0.f;
    vr[n] = (ok && a.var) ? a.var[h] : 1.f;
    w2h[n] = ok ? a.w2[h] : 0.f;                           // padded units contribute nothing to the score
  }
  const float b2 = a.b2[0];
  float pacc[4] = {0.f, 0.f, 0.f, 0.f};                    // pooled dims lane, lane+64, lane+128, lane+192
  bool bad = false;
  if (a.stop == 1) return;
  for (int t0 = 0; t0 < a.T; t0 += TC) {
    __syncthreads();                                       // Eff ready / previous chunk consumed
    attn_gather(a, b, t0, D, L, lds, &bad);
    __syncthreads();
    if (a.stop == 2) continue;
    if (t0 + r0 >= a.T) continue;                          // wave-uniform: all 16 rows of this wave are beyond T (no
                                                           // barrier below in this iteration, zero contribution)
    f32x4 acc[NT];
    attn_pre_gemm<NT>(KT, Eff, L, r0, l15, g, acc);
    if (a.stop == 3) { if (acc[0][0] == 12345.f) pacc[0] += 1.f; continue; }
    float sr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float cv = cvec[16 * n + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sr[r] += feat_act(a.act, acc[n][r] + cv, al[n], mu[n], vr[n], nullptr, nullptr) * w2h[n];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sr[r] += __shfl_xor(sr[r], o, 64);      // over the 16 unit lanes
      sr[r] += b2;
    }
    if (l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = r0 + 4 * g + r, t = t0 + row;
        if (t < a.T) scores[b * a.T + t] = sr[r];
        msb[row] = maskb[row] * sr[r];
      }
    }
    if (a.stop == 4) continue;
    // masked weighted sum of this wave's 16 key rows (msb of these rows was written by this wave)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int d = lane + 64 * j;
      if (d < L.Dp) {
        float p = 0.f;
#pragma unroll
        for (int row = 0; row < 16; ++row) p += msb[r0 + row] * KT[(r0 + row) * L.DS + d];
        pacc[j] += p;
      }
    }
  }
  if (bad && oob) *oob = 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int d = lane + 64 * j;
    if (d < L.Dp) red[wave * L.Dp + d] = pacc[j];
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256)
    pooled[b * D + d] = ((red[d] + red[L.Dp + d]) + red[2 * L.Dp + d]) + red[3 * L.Dp + d];
}

template <int NT>
__global__ __launch_bounds__(256) void din_attn_bwd_kernel(AttnArgs a, int D, int H, const float* __restrict__ scores,
                                                           const float* __restrict__ gpooled,
                                                           float* __restrict__ gkeys /* [B,T,D] */,
                                                           float* __restrict__ gMext /* [B, D*H+H] */,
                                                           float* __restrict__ gw2p /* [B,H] */,
                                                           float* __restrict__ galphap /* [B,H] */,
                                                           float* __restrict__ gb2p /* [B] */) {
  extern __shared__ float lds[];
  const AttnLds L = attn_layout(D, H, true);
  float* Eff = lds + L.eff;
  float* cvec = lds + L.cvec;
  float* KT = lds + L.kt;
  float* msb = lds + L.msb;
  float* maskb = lds + L.maskb;
  float* GP = lds + L.gp;
  float* gsb = lds + L.gsb;
  float* gpl = lds + L.gpl;
  float* redh = lds + L.redh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, r0 = wave * 16;
  const int64_t b = blockIdx.x;
  const int64_t NM = (int64_t)D * H + H;
  attn_load_eff(a, b, D, H, L, lds);
  for (int d = tid; d < L.Dp; d += 256) gpl[d] = d < D ? gpooled[b * D + d] : 0.f;
  float al[NT], mu[NT], vr[NT], w2h[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int h = 16 * n + l15;
    bool ok = h < H;
    al[n] = (ok && a.alpha) ? a.alpha[h] : 0.f;
    mu[n] = (ok && a.mean) ? a.mean[h] : 0.f;
    vr[n] = (ok && a.var) ? a.var[h] : 1.f;
    w2h[n] = ok ? a.w2[h] : 0.f;
  }
  // gEff row tiles of this wave: wave, wave + 4, ... (Dp/16 <= 16 tiles)
  constexpr int MTW = 4;
  const int nmt = L.Dp >> 4;
  f32x4 ge[MTW][NT];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) ge[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gc[NT], gw2a[NT], gala[NT], gb2a = 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) { gc[n] = 0.f; gw2a[n] = 0.f; gala[n] = 0.f; }
  bool bad = false;
  for (int t0 = 0; t0 < a.T; t0 += TC) {
    __syncthreads();
    attn_gather(a, b, t0, D, L, lds, &bad);
    __syncthreads();
    const bool wave_live = t0 + r0 < a.T && a.stop != 12;  // wave-uniform: some of this wave's 16 rows are real steps
    if (wave_live) {
    // (1) pre-activations of this wave's rows
    f32x4 acc[NT];
    attn_pre_gemm<NT>(KT, Eff, L, r0, l15, g, acc);
    // (2) d L / d score of row l15 of this wave: mask * <g_pooled, k_t>; (mask*score) kept for the keys' gradient
    {
      float dot = 0.f;
      const float* kr = KT + (r0 + l15) * L.DS + g;
#pragma unroll 8
      for (int j = 0; j < (L.Dp >> 2); ++j) dot += gpl[4 * j + g] * kr[4 * j];
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      if (g == 0) {
        int row = r0 + l15, t = t0 + row;
        float m = maskb[row];
        gsb[row] = m * dot;
        msb[row] = t < a.T ? m * scores[b * a.T + t] : 0.f;
      }
    }
    float gsr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gsr[r] = gsb[r0 + 4 * g + r];
    // (3) through the score layer and the activation; gpre tile -> LDS
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float cv = cvec[16 * n + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float dydx, dyda;
        float hv = feat_act(a.act, acc[n][r] + cv, al[n], mu[n], vr[n], &dydx, &dyda);
        float gh = gsr[r] * w2h[n];
        float gpre = gh * dydx;
        gw2a[n] += gsr[r] * hv;
        gala[n] += gh * dyda;
        gc[n] += gpre;
        GP[(r0 + 4 * g + r) * L.HS + 16 * n + l15] = gpre;
      }
    }
    if (l15 == 0) gb2a += (gsr[0] + gsr[1]) + (gsr[2] + gsr[3]);
    }
    __syncthreads();                                       // GP of the chunk's live rows
    // (4) gEff += K^T . gpre over the chunk's real steps (groups of 4; rows beyond T inside a group are zero in KT)
    if (a.stop != 12 && a.stop != 13) {
      const int live_rows = a.T - t0 < TC ? a.T - t0 : TC;
      const int nk = (live_rows + 3) >> 2;
      for (int k = 0; k < nk; ++k) {
        float bv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = GP[(4 * k + g) * L.HS + 16 * n + l15];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          int mt = wave + 4 * i;
          if (mt < nmt) {
            float av = KT[(4 * k + g) * L.DS + 16 * mt + l15];
#pragma unroll
            for (int n = 0; n < NT; ++n) ge[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[n], ge[i][n], 0, 0, 0);
          }
        }
      }
    }
    // (5) gkeys of this wave's rows: mask*score*g_pooled + gpre . Eff^T
    if (wave_live && a.stop != 13 && a.stop != 14) {
      const int nkh = L.Hp >> 2;
      float msr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) msr[r] = msb[r0 + 4 * g + r];
      for (int dt = 0; dt < nmt; ++dt) {
        f32x4 ak = {0.f, 0.f, 0.f, 0.f};
        const float* ap = GP + (r0 + l15) * L.HS + g;
        const float* bp = Eff + (16 * dt + l15) * L.HS + g;
#pragma unroll 4
        for (int k = 0; k < nkh; ++k) ak = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * k], bp[4 * k], ak, 0, 0, 0);
        int d = 16 * dt + l15;
        if (d < D) {
          float gp_d = gpl[d];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int t = t0 + r0 + 4 * g + r;
            if (t < a.T) gkeys[((int64_t)b * a.T + t) * D + d] = msr[r] * gp_d + ak[r];
          }
        }
      }
    }
  }
  // gEff -> gMext[b, d*H + h]
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    int mt = wave + 4 * i;
    if (mt < nmt) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        int h = 16 * n + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int d = 16 * mt + 4 * g + r;
          if (d < D && h < H) gMext[b * NM + (int64_t)d * H + h] = ge[i][n][r];
        }
      }
    }
  }
  // per-unit sums: over the row groups of the wave (lanes g), then over the waves in a fixed order
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    gc[n] += __shfl_xor(gc[n], 16, 64); gc[n] += __shfl_xor(gc[n], 32, 64);
    gw2a[n] += __shfl_xor(gw2a[n], 16, 64); gw2a[n] += __shfl_xor(gw2a[n], 32, 64);
    gala[n] += __shfl_xor(gala[n], 16, 64); gala[n] += __shfl_xor(gala[n], 32, 64);
  }
  gb2a += __shfl_xor(gb2a, 16, 64);
  gb2a += __shfl_xor(gb2a, 32, 64);
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      int h = 16 * n + l15;
      redh[(wave * 4 + 0) * L.Hp + h] = gc[n];
      redh[(wave * 4 + 1) * L.Hp + h] = gw2a[n];
      redh[(wave * 4 + 2) * L.Hp + h] = gala[n];
    }
    if (l15 == 0) redh[(wave * 4 + 3) * L.Hp] = gb2a;
  }
  __syncthreads();
  if (tid < H) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int wv = 0; wv < 4; ++wv) {
      s0 += redh[(wv * 4 + 0) * L.Hp + tid];
      s1 += redh[(wv * 4 + 1) * L.Hp + tid];
      s2 += redh[(wv * 4 + 2) * L.Hp + tid];
    }
    gMext[b * NM + (int64_t)D * H + tid] = s0;
    gw2p[b * H + tid] = s1;
    galphap[b * H + tid] = s2;
  }
  if (tid == 0)
    gb2p[b] = ((redh[3 * L.Hp] + redh[(4 + 3) * L.Hp]) + redh[(8 + 3) * L.Hp]) + redh[(12 + 3) * L.Hp];
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

size_t attn_lds_bytes(int D, int H, bool bwd) { return attn_layout(D, H, bwd).total * sizeof(float); }

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
  if (lds > 150 * 1024) return REC_E_UNSUPPORTED;
#ifdef REC_DEBUG_PHASE_STOPS   // profiling builds only (scripts/exp/*_phases.sh): the kernel stops after phase N
  static const int din_stop = getenv("REC_DIN_STOP") ? atoi(getenv("REC_DIN_STOP")) : 0;
#else
  constexpr int din_stop = 0;
#endif
  AttnArgs a{embed, ld, V, E, C, series, T, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid, din_stop};
#define LAUNCH_FWD(NT)                                                                                            \
  do {                                                                                                            \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_fwd_kernel<NT>),                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
    if (e != hipSuccess) return (int)e;                                                                           \
    hipLaunchKernelGGL(din_attn_fwd_kernel<NT>, dim3((unsigned)B), dim3(256), lds, as_stream(stream), a, D, H,    \
                       scores, pooled, oob_flag);                                                                 \
  } while (0)
  switch ((H + 15) / 16) {
    case 1: LAUNCH_FWD(1); break;
    case 2: LAUNCH_FWD(2); break;
    case 3: LAUNCH_FWD(3); break;
    default: LAUNCH_FWD(4); break;
  }
#undef LAUNCH_FWD
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
#ifdef REC_DEBUG_PHASE_STOPS   // profiling builds only (scripts/exp/*_phases.sh): the kernel stops after phase N
  static const int din_stop = getenv("REC_DIN_STOP") ? atoi(getenv("REC_DIN_STOP")) : 0;
#else
  constexpr int din_stop = 0;
#endif
  AttnArgs a{embed, ld, V, E, C, series, T, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid, din_stop};
#define LAUNCH_BWD(NT)                                                                                            \
  do {                                                                                                            \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_bwd_kernel<NT>),                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
    if (e != hipSuccess) return (int)e;                                                                           \
    hipLaunchKernelGGL(din_attn_bwd_kernel<NT>, dim3((unsigned)B), dim3(256), lds, as_stream(stream), a, D, H,    \
                       scores, gpooled, gkeys, gMext, gw2p, galphap, gb2p);                                       \
  } while (0)
  switch ((H + 15) / 16) {
    case 1: LAUNCH_BWD(1); break;
    case 2: LAUNCH_BWD(2); break;
    case 3: LAUNCH_BWD(3); break;
    default: LAUNCH_BWD(4); break;
  }
#undef LAUNCH_BWD
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
