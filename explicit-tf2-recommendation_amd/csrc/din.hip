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

// The same inside the attention kernels, where 12 (forward) / 12 x 3 (backward) evaluations per lane and tile made the
// VALU the busiest unit: hardware exp2 / reciprocal (1 ulp each; scores stay within 1e-6 of the accurate form), the
// Dice scale 1/sqrt(var + eps) precomputed per unit (rstd).
__device__ __forceinline__ float sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <bool GRAD>
__device__ __forceinline__ float feat_act_hw(int kind, float x, float alpha, float mean, float rstd, float& dydx,
                                             float& dyda) {
  float y = x, dx = 1.f, da = 0.f;
  switch (kind) {
    case DACT_RELU: y = fmaxf(x, 0.f); dx = x > 0.f ? 1.f : 0.f; break;
    case DACT_SIGMOID: y = sigmoid_hw(x); dx = y * (1.f - y); break;
    case DACT_TANH: y = tanhf(x); dx = 1.f - y * y; break;
    case DACT_DICE: {
      float p = sigmoid_hw((x - mean) * rstd);
      float q = 1.f - p;
      y = alpha * q * x + p * x;
      if (GRAD) {
        dx = alpha * q + p + x * (p * q * rstd) * (1.f - alpha);
        da = q * x;
      }
      break;
    }
    case DACT_PRELU: y = x > 0.f ? x : alpha * x; dx = x > 0.f ? 1.f : alpha; da = x > 0.f ? 0.f : x; break;
    default: break;
  }
  if (GRAD) { dydx = dx; dyda = da; }
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
// attention forward / backward: one workgroup (4 waves) per example, and inside it every WAVE works on its own tiles of
// 16 time steps (tiles wave, wave + 4, ...) from the gather to the stores: no workgroup barrier inside the time loop, so
// the gather latency of one wave hides behind the matrix work of the eleven others on the CU.  The tile's key rows sit in
// a wave-private slab of LDS (zero-padded); pre = K.Eff, gEff = K^T.gpre and gkeys = gpre.Eff^T run on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D: col = l&15, row =
// 4*(l>>4) + reg).  The contraction index of every product is PERMUTED so that a lane's four k steps are one 16-byte LDS
// read (k step c of block q is d = 16q + 4g + c for lane group g) or registers it already holds (the D layout of `pre`
// IS the B layout of K^T.gpre with time step 4g + c as k step c); with row strides = 4 (mod 16) floats every one of those
// reads and the D-layout writes touch 64 distinct banks.  All dimensions are padded to multiples of 16 with zeros in
// LDS, so padded rows / columns contribute exact zeros.  Keys are never written to HBM in the forward.
// LDS (floats): Eff [Dp][HS] | cvec [Hp] | gpl [Dp] | KT [64][DS] | GP [64][HS] | msb, maskb, gsb [64] | red / redh
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TW = 16;              // time steps per wave tile
constexpr int TC = 4 * TW;          // ... per round of the workgroup
constexpr int NBR = 6;              // 16-byte pieces of key rows a lane holds per tile (16 rows x D <= 96 floats)

struct AttnArgs {
  const float* embed; int64_t ld; int64_t V; int E; int C;
  const int64_t* series; int T;
  const float* Mext; const float* Wkd;
  int act; const float* alpha; const float* mean; const float* var;
  const float* w2; const float* b2;
  int64_t padding_index; int mask_valid;
};

// LDS layout (floats).  Every extent is a compile-time constant (ND = padded D / 16, NT = padded H / 16): the row strides
// fold into the offset fields of the LDS instructions instead of living in address registers.
template <int NT, int ND, bool BWD>
struct AL {
  static constexpr int Dp = 16 * ND, Hp = 16 * NT;
  static constexpr int DS = Dp + 4, HS = Hp + 4;           // = 4 (mod 16)
  static constexpr int eff = 0;
  static constexpr int cvec = eff + Dp * HS;
  static constexpr int gpl = cvec + Hp;
  static constexpr int kt = gpl + Dp;
  static constexpr int gp = kt + TC * DS;
  static constexpr int msb = gp + (BWD ? TC * HS : 0);
  static constexpr int maskb = msb + TC;
  static constexpr int gsb = maskb + TC;
  static constexpr int red = gsb + TC;
  static constexpr int total = red + (BWD ? 16 * Hp : 4 * Dp);
};

// LDS traffic of ONE wave (its own slab): program order inside the wave is enough, the compiler must not reorder
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Eff = Wkd + M_b (zero-padded), c_b; zero the padding columns of KT once
template <class L>
__device__ __forceinline__ void attn_load_eff(const AttnArgs& a, int64_t b, int D, int H, float* lds) {
  float* Eff = lds + L::eff;
  float* cvec = lds + L::cvec;
  float* KT = lds + L::kt;
  const int64_t NM = (int64_t)D * H + H;
  const float* mrow = a.Mext + b * NM;
  const int DH = D * H;
  if ((DH & 3) == 0 && (NM & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(a.Mext) | reinterpret_cast<uintptr_t>(a.Wkd)) & 15) == 0) {
    // the row [Eff part D*H | c part H] is streamed as float4 (4 per lane in flight) and scattered into the padded tile
    // padding entries first (disjoint from the rest): columns H.. of the real rows, then the rows beyond D
    for (int r = threadIdx.x >> 4; r < D; r += 16)
      for (int o = H + (threadIdx.x & 15); o < L::HS; o += 16) Eff[r * L::HS + o] = 0.f;
    for (int i = D * L::HS + threadIdx.x; i < L::Dp * L::HS; i += 256) Eff[i] = 0.f;
    for (int o = H + threadIdx.x; o < L::Hp; o += 256) cvec[o] = 0.f;
    const int h_magic = (1 << 20) / H + 1;                 // f / H for f < 2^20 / H (D*H <= 16384)
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
          int r = (int)(((uint64_t)(unsigned)f * (unsigned)h_magic) >> 20), o = f - r * H;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            Eff[r * L::HS + o] = e[j];
            if (++o == H) { o = 0; ++r; }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) cvec[f - DH + j] = e[j];
        }
      }
    }
  } else {
    constexpr int n_eff = L::Dp * L::HS;
    for (int i0 = threadIdx.x; i0 < n_eff; i0 += 256 * 6) {
      float wv[6], mv[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        int i = i0 + u * 256;
        wv[u] = mv[u] = 0.f;
        if (i < n_eff) {
          int r = i / L::HS, o = i - r * L::HS;
          if (r < D && o < H) { wv[u] = a.Wkd[r * H + o]; mv[u] = mrow[(int64_t)r * H + o]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        int i = i0 + u * 256;
        if (i < n_eff) Eff[i] = wv[u] + mv[u];
      }
    }
    for (int o = threadIdx.x; o < L::Hp; o += 256) cvec[o] = o < H ? mrow[(int64_t)DH + o] : 0.f;
  }
  const int padc = L::DS - D;                              // columns D .. DS-1 of every KT row stay zero
  for (int i = threadIdx.x; i < TC * padc; i += 256) KT[(i / padc) * L::DS + D + i % padc] = 0.f;
}

// Key rows t0 .. t0+15 of one wave tile, in two halves so that the rows of the NEXT tile travel while this one is worked
// on: `issue` starts the loads (ids, then up to NBR independent 16-byte row pieces per lane, 8 lanes = one 128-byte row
// at E = 32; rows beyond T and rows of out-of-range ids: zeros), `commit` puts them into the wave's slab KTw [16][DS].
struct RowRegs {
  float4 v[NBR];
  int off[NBR];     // LDS offset of the piece inside the slab, -1: this lane has no piece u
  unsigned okm;     // bit u: piece u is a real row (inside the series, id in range); else it is stored as zeros
  int64_t mid;      // lanes 0..15: series id (table 0) of row `lane` -- the mask
  float sc;         // lanes 0..15 (backward): the forward's score of that row
  bool row_in;      // lanes 0..15: row `lane` lies inside the series
};
// Every load is issued unconditionally from a clamped (always valid) address and judged afterwards: a load inside a
// guarded block is waited for before the next one is issued, which turned the six ids and six row pieces of a tile into
// twelve dependent memory round trips.
template <class L>
__device__ __forceinline__ void attn_rows_issue(const AttnArgs& a, int64_t b, int t0, int lane, int e4_shift,
                                                int c_magic, const float* scores, RowRegs& R, bool* bad) {
  const int E = a.E, C = a.C;
  const int64_t* ser = a.series + (int64_t)b * a.T * C;
  const int total = (TW * C) << e4_shift;
  {
    const int t = t0 + (lane & (TW - 1));
    const int tc = t < a.T ? t : a.T - 1;
    R.row_in = lane < TW && t < a.T;
    R.mid = ser[(int64_t)tc * C];
    R.sc = scores ? scores[b * a.T + tc] : 0.f;
  }
  int64_t id[NBR];
  int e4s[NBR];
  bool live[NBR];
#pragma unroll
  for (int u = 0; u < NBR; ++u) {
    const int i = lane + u * 64;
    const bool mine = i < total;
    const int ic = mine ? i : 0;
    const int rr = ic >> e4_shift, e4 = ic - (rr << e4_shift);   // (row, table) pair and 16-byte piece of its row
    const int row = (rr * c_magic) >> 16, r = rr - row * C;      // rr / C for rr < 1024 (c_magic = 65536 / C + 1)
    const int t = t0 + row;
    R.off[u] = mine ? row * L::DS + r * E + 4 * e4 : -1;
    e4s[u] = 4 * e4;
    live[u] = mine && t < a.T;                                   // beyond the series: a zero row, not an error
    id[u] = ser[(int64_t)(t < a.T ? t : a.T - 1) * C + r];
  }
  R.okm = 0u;
#pragma unroll
  for (int u = 0; u < NBR; ++u) {
    const bool inr = (uint64_t)id[u] < (uint64_t)a.V;
    if (live[u] && !inr) *bad = true;
    if (live[u] && inr) R.okm |= 1u << u;
    R.v[u] = *reinterpret_cast<const float4*>(a.embed + (inr ? id[u] : 0) * a.ld + e4s[u]);
  }
}
// mask of row `lane` (lanes 0..15) from the id that came with the rows; reference quirk: mask = (id == padding)
__device__ __forceinline__ float attn_mask_of(const AttnArgs& a, const RowRegs& R) {
  if (!R.row_in) return 0.f;
  const bool pad = R.mid == a.padding_index;
  return (a.mask_valid ? !pad : pad) ? 1.f : 0.f;
}
__device__ __forceinline__ void attn_rows_commit(const RowRegs& R, float* KTw) {
#pragma unroll
  for (int u = 0; u < NBR; ++u) {
    if (R.off[u] >= 0) {
      const bool ok = (R.okm >> u) & 1u;
      *reinterpret_cast<float4*>(KTw + R.off[u]) = ok ? R.v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__device__ __forceinline__ void attn_mask_rows(const AttnArgs& a, int64_t b, int t0, int lane, float* maskw) {
  if (lane < TW) {
    const int64_t* ser = a.series + (int64_t)b * a.T * a.C;
    int t = t0 + lane;
    float m = 0.f;
    if (t < a.T) {
      bool pad = ser[(int64_t)t * a.C] == a.padding_index;
      m = (a.mask_valid ? !pad : pad) ? 1.f : 0.f;         // reference quirk: mask = (id == padding)
    }
    maskw[lane] = m;
  }
}
// any E / ld / D: the tile's rows element by element, loads and stores in rounds of NBR per lane (no prefetch)
template <class L>
__device__ __forceinline__ void attn_rows_slow(const AttnArgs& a, int64_t b, int t0, int lane, float* KTw, bool* bad) {
  const int E = a.E, C = a.C;
  const int64_t* ser = a.series + (int64_t)b * a.T * C;
  const int total = TW * C * E;
  for (int i0 = lane; i0 < total; i0 += 64 * NBR) {
    float v[NBR];
    int off[NBR];
#pragma unroll
    for (int u = 0; u < NBR; ++u) {
      int i = i0 + u * 64;
      off[u] = -1;
      v[u] = 0.f;
      if (i < total) {
        int rr = i / E, e = i - rr * E;
        int row = rr / C, r = rr - row * C;
        int t = t0 + row;
        off[u] = row * L::DS + r * E + e;
        if (t < a.T) {
          int64_t id = ser[(int64_t)t * C + r];
          if ((uint64_t)id < (uint64_t)a.V) v[u] = a.embed[id * a.ld + e];
          else *bad = true;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NBR; ++u)
      if (off[u] >= 0) KTw[off[u]] = v[u];
  }
}

// pre-activation tile of the wave's 16 rows: acc[n][r] = sum_d KTw[4g + r][d] * Eff[d][16n + l15]; with DOT also
// dot = sum over this lane's d of gpl[d] * KTw[l15][d] (the caller adds the four lane groups)
template <class L, int NT, int ND, bool DOT>
__device__ __forceinline__ void attn_pre_gemm(const float* KTw, const float* Eff, const float* gpl, int l15, int g,
                                              f32x4 (&acc)[NT], float& dot) {
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* ap = KTw + l15 * L::DS + 4 * g;
  const float* bp = Eff + 4 * g * L::HS + l15;
  const float* gq = gpl + 4 * g;
#pragma unroll(DOT ? 1 : 2)
  for (int q = 0; q < ND; ++q) {
    const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * q);
    const float ac[4] = {a4.x, a4.y, a4.z, a4.w};
    float bc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int n = 0; n < NT; ++n) bc[c][n] = bp[(16 * q + c) * L::HS + 16 * n];
    if (DOT) {
      const float4 gp4 = *reinterpret_cast<const float4*>(gq + 16 * q);
      dot += (gp4.x * ac[0] + gp4.y * ac[1]) + (gp4.z * ac[2] + gp4.w * ac[3]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[c], bc[c][n], acc[n], 0, 0, 0);
  }
}

template <int NT, int ND>
__global__ __launch_bounds__(256, 3) void din_attn_fwd_kernel(AttnArgs a, int D, int H, float* __restrict__ scores,
                                                              float* __restrict__ pooled, int* oob) {
  using L = AL<NT, ND, false>;
  extern __shared__ float lds[];
  float* Eff = lds + L::eff;
  float* cvec = lds + L::cvec;
  float* red = lds + L::red;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, r0 = wave * TW;
  float* KTw = lds + L::kt + r0 * L::DS;                    // the wave's slab
  float* msw = lds + L::msb + r0;
  float* maskw = lds + L::maskb + r0;
  const int64_t b = blockIdx.x;
  bool bad = false;
  // fast rows: 16-byte pieces, at most NBR per lane and tile -> the first two tiles of the wave are requested before
  // anything else happens, every later one two tiles ahead of its use
  // fast rows: 16-byte pieces, a power of two of them per row, at most NBR per lane and tile
  const int e4_shift = 31 - __clz(a.E >> 2 > 0 ? a.E >> 2 : 1), c_magic = 65536 / a.C + 1;
  const bool fast = (a.E & 3) == 0 && (a.ld & 3) == 0 && (4 << e4_shift) == a.E && TW * a.C * (a.E >> 2) <= 64 * NBR;
  RowRegs R0, R1;
  if (fast) {
    if (r0 < a.T) attn_rows_issue<L>(a, b, r0, lane, e4_shift, c_magic, nullptr, R0, &bad);
    if (r0 + TC < a.T) attn_rows_issue<L>(a, b, r0 + TC, lane, e4_shift, c_magic, nullptr, R1, &bad);
  }
  attn_load_eff<L>(a, b, D, H, lds);
  float al[NT], mu[NT], vr[NT], w2h[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int h = 16 * n + l15;
    bool ok = h < H;
    al[n] = (ok && a.alpha) ? a.alpha[h] : 0.f;
    mu[n] = (ok && a.mean) ? a.mean[h] : 0.f;
    vr[n] = (ok && a.var) ? rsqrtf(a.var[h] + BN_EPS) : 1.f;   // Dice: 1 / sqrt(var + eps)
    w2h[n] = ok ? a.w2[h] : 0.f;                           // padded units contribute nothing to the score
  }
  const float b2 = a.b2[0];
  constexpr int NP = (L::Dp + 63) / 64;
  float pacc[NP];                                          // pooled dims lane, lane+64, ...
#pragma unroll
  for (int j = 0; j < NP; ++j) pacc[j] = 0.f;
  __syncthreads();                                         // Eff, cvec and the zero padding are in place

  auto tile = [&](RowRegs& R, int t0) {
    wave_lds_sync();                                       // the previous tile has been consumed
    if (fast) {
      attn_rows_commit(R, KTw);
      if (lane < TW) maskw[lane] = attn_mask_of(a, R);
      if (t0 + 2 * TC < a.T) attn_rows_issue<L>(a, b, t0 + 2 * TC, lane, e4_shift, c_magic, nullptr, R, &bad);
    } else {
      attn_rows_slow<L>(a, b, t0, lane, KTw, &bad);
      attn_mask_rows(a, b, t0, lane, maskw);
    }
    wave_lds_sync();
    f32x4 acc[NT];
    float unused = 0.f;
    attn_pre_gemm<L, NT, ND, false>(KTw, Eff, nullptr, l15, g, acc, unused);
    float sr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float cv = cvec[16 * n + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sr[r] += feat_act_hw<false>(a.act, acc[n][r] + cv, al[n], mu[n], vr[n], unused, unused) * w2h[n];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sr[r] = row16_allsum(sr[r]) + b2;       // over the 16 unit lanes
    if (l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r, t = t0 + row;
        if (t < a.T) scores[b * a.T + t] = sr[r];
        msw[row] = maskw[row] * sr[r];
      }
    }
    wave_lds_sync();
    // masked weighted sum of the tile's 16 key rows
    float ms[TW];
#pragma unroll
    for (int q4 = 0; q4 < TW / 4; ++q4) {
      const float4 m4 = *reinterpret_cast<const float4*>(msw + 4 * q4);
      ms[4 * q4] = m4.x; ms[4 * q4 + 1] = m4.y; ms[4 * q4 + 2] = m4.z; ms[4 * q4 + 3] = m4.w;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      int d = lane + 64 * j;
      if (d < L::Dp) {
        float p = 0.f;
#pragma unroll
        for (int row = 0; row < TW; ++row) p += ms[row] * KTw[row * L::DS + d];
        pacc[j] += p;
      }
    }
  };
#pragma unroll 1
  for (int t0 = r0; t0 < a.T; t0 += 2 * TC) {
    tile(R0, t0);
    if (t0 + TC < a.T) tile(R1, t0 + TC);
  }
  if (bad && oob) *oob = 1;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int d = lane + 64 * j;
    if (d < L::Dp) red[wave * L::Dp + d] = pacc[j];
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256)
    pooled[b * D + d] = ((red[d] + red[L::Dp + d]) + red[2 * L::Dp + d]) + red[3 * L::Dp + d];
}

// NMT: d tiles of gEff a wave accumulates at a time (its 16 time steps x all units); NPASS > 1 (D > 128): several passes
// over the series, the later ones only for their gEff tiles
template <int NT, int NMT, int NPASS>
__global__ __launch_bounds__(256, 2) void din_attn_bwd_kernel(AttnArgs a, int D, int H, const float* __restrict__ scores,
                                                              const float* __restrict__ gpooled,
                                                              float* __restrict__ gkeys /* [B,T,D] */,
                                                              float* __restrict__ gMext /* [B, D*H+H] */,
                                                              float* __restrict__ gw2p /* [B,H] */,
                                                              float* __restrict__ galphap /* [B,H] */,
                                                              float* __restrict__ gb2p /* [B] */) {
  constexpr int ND = NMT * NPASS;
  using L = AL<NT, ND, true>;
  extern __shared__ float lds[];
  float* Eff = lds + L::eff;
  float* cvec = lds + L::cvec;
  float* gpl = lds + L::gpl;
  float* redh = lds + L::red;
  float* GE = lds + L::kt;                                  // [16 * NMT][Hp]: aliases the slabs, behind barriers
  static_assert(16 * NMT * L::Hp <= TC * (L::DS + L::HS), "gEff staging must fit into the slabs");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, r0 = wave * TW;
  float* KTw = lds + L::kt + r0 * L::DS;
  float* GPw = lds + L::gp + r0 * L::HS;
  float* msw = lds + L::msb + r0;
  float* maskw = lds + L::maskb + r0;
  float* gsw = lds + L::gsb + r0;
  const int64_t b = blockIdx.x;
  const int64_t NM = (int64_t)D * H + H;
  bool bad = false;
  // fast rows: 16-byte pieces, a power of two of them per row, at most NBR per lane and tile
  const int e4_shift = 31 - __clz(a.E >> 2 > 0 ? a.E >> 2 : 1), c_magic = 65536 / a.C + 1;
  const bool fast = (a.E & 3) == 0 && (a.ld & 3) == 0 && (4 << e4_shift) == a.E && TW * a.C * (a.E >> 2) <= 64 * NBR;
  RowRegs R;
  attn_load_eff<L>(a, b, D, H, lds);
  for (int d = tid; d < L::Dp; d += 256) gpl[d] = d < D ? gpooled[b * D + d] : 0.f;
  float al[NT], mu[NT], vr[NT], w2h[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int h = 16 * n + l15;
    bool ok = h < H;
    al[n] = (ok && a.alpha) ? a.alpha[h] : 0.f;
    mu[n] = (ok && a.mean) ? a.mean[h] : 0.f;
    vr[n] = (ok && a.var) ? rsqrtf(a.var[h] + BN_EPS) : 1.f;   // Dice: 1 / sqrt(var + eps)
    w2h[n] = ok ? a.w2[h] : 0.f;
  }
  float gc[NT], gw2a[NT], gala[NT], gb2a = 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) { gc[n] = 0.f; gw2a[n] = 0.f; gala[n] = 0.f; }
  __syncthreads();                                         // Eff, cvec, gpl and the zero padding are in place
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    const bool first = pass == 0;
    const int mt0 = pass * NMT;
    f32x4 ge[NMT][NT];
#pragma unroll
    for (int i = 0; i < NMT; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) ge[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t0 = r0; t0 < a.T; t0 += TC) {
      wave_lds_sync();
      float m_row = 0.f, sc_row = 0.f;                      // lanes 0..15: mask and forward score of row `lane`
      if (fast) {
        attn_rows_issue<L>(a, b, t0, lane, e4_shift, c_magic, scores, R, &bad);
        attn_rows_commit(R, KTw);
        m_row = attn_mask_of(a, R);
        sc_row = R.row_in ? R.sc : 0.f;
      } else {
        attn_rows_slow<L>(a, b, t0, lane, KTw, &bad);
        attn_mask_rows(a, b, t0, lane, maskw);
      }
      wave_lds_sync();
      if (!fast && lane < TW) {
        m_row = maskw[lane];
        sc_row = t0 + lane < a.T ? scores[b * a.T + t0 + lane] : 0.f;
      }
      // (1) pre-activations of the tile, (2) d L / d score of row l15: mask * <g_pooled, k_t>
      f32x4 acc[NT];
      float dot = 0.f;
      attn_pre_gemm<L, NT, ND, true>(KTw, Eff, gpl, l15, g, acc, dot);
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      if (g == 0) {                                        // lanes 0..15 = rows 0..15
        gsw[l15] = m_row * dot;
        msw[l15] = m_row * sc_row;                         // (mask*score) for the keys' gradient
      }
      wave_lds_sync();
      float gsr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) gsr[r] = gsw[4 * g + r];
      // (3) through the score layer and the activation: gpre stays in acc (the B operand of (4)) and goes to the
      // wave's GP slab (the A operand of (5) needs it with the time step on the other lane index)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float cv = cvec[16 * n + l15];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dydx, dyda;
          float hv = feat_act_hw<true>(a.act, acc[n][r] + cv, al[n], mu[n], vr[n], dydx, dyda);
          float gh = gsr[r] * w2h[n];
          float gpre = gh * dydx;
          if (first) {
            gw2a[n] += gsr[r] * hv;
            gala[n] += gh * dyda;
            gc[n] += gpre;
            GPw[(4 * g + r) * L::HS + 16 * n + l15] = gpre;
          }
          acc[n][r] = gpre;
        }
      }
      if (first && l15 == 0) gb2a += (gsr[0] + gsr[1]) + (gsr[2] + gsr[3]);
      wave_lds_sync();
      // (4) gEff[d][h] += sum over the tile's steps of K[t][d] * gpre[t][h]; k step c = time step 4g + c
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* kr = KTw + (4 * g + c) * L::DS + 16 * mt0 + l15;
#pragma unroll
        for (int i = 0; i < NMT; ++i) {
          const float av = kr[16 * i];
#pragma unroll
          for (int n = 0; n < NT; ++n) ge[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, acc[n][c], ge[i][n], 0, 0, 0);
        }
      }
      // (5) gkeys of the tile's rows: mask*score*g_pooled + gpre . Eff^T; k step c of block p = unit 16p + 4g + c
      if (first) {
        float4 ga[NT];
#pragma unroll
        for (int p = 0; p < NT; ++p) ga[p] = *reinterpret_cast<const float4*>(GPw + l15 * L::HS + 16 * p + 4 * g);
        float msr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) msr[r] = msw[4 * g + r];
        float* gk = gkeys + ((int64_t)b * a.T + t0 + 4 * g) * D + l15;
#pragma unroll 1
        for (int dt = 0; dt < ND; ++dt) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f};
          const float* bp = Eff + (16 * dt + l15) * L::HS + 4 * g;
#pragma unroll
          for (int p = 0; p < NT; ++p) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 16 * p);
            ak = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[p].x, b4.x, ak, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[p].y, b4.y, ak, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[p].z, b4.z, ak, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[p].w, b4.w, ak, 0, 0, 0);
          }
          const int d = 16 * dt + l15;
          if (d < D) {
            const float gp_d = gpl[d];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t0 + 4 * g + r < a.T) gk[(int64_t)r * D + 16 * dt] = msr[r] * gp_d + ak[r];
          }
        }
      }
    }
    // gEff of this pass: the four waves' partial tiles are added in wave order through LDS, then the rows leave as one
    // contiguous block of gMext[b, d*H + h]
    __syncthreads();                                       // every wave is done with its slabs
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < NMT; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* p = GE + (16 * i + 4 * g + r) * L::Hp + 16 * n + l15;
              *p = (w == 0 ? 0.f : *p) + ge[i][n][r];
            }
      }
      __syncthreads();
    }
    {
      const int d_lo = 16 * mt0;
      const int rows = (D - d_lo < 16 * NMT) ? D - d_lo : 16 * NMT;
      float* dst = gMext + b * NM + (int64_t)d_lo * H;
      for (int dl = wave; dl < rows; dl += 4)
        for (int h = lane; h < H; h += 64) dst[dl * H + h] = GE[dl * L::Hp + h];
    }
    __syncthreads();                                       // GE consumed before the slabs are written again
    if (pass + 1 < NPASS) {                                 // the alias clobbered the zero padding of the key slabs
      float* KT = lds + L::kt;
      const int padc = L::DS - D;
      for (int i = tid; i < TC * padc; i += 256) KT[(i / padc) * L::DS + D + i % padc] = 0.f;
      __syncthreads();
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
  if (g == 0) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      int h = 16 * n + l15;
      redh[(wave * 4 + 0) * L::Hp + h] = gc[n];
      redh[(wave * 4 + 1) * L::Hp + h] = gw2a[n];
      redh[(wave * 4 + 2) * L::Hp + h] = gala[n];
    }
    if (l15 == 0) redh[(wave * 4 + 3) * L::Hp] = gb2a;
  }
  __syncthreads();
  if (tid < H) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int wv = 0; wv < 4; ++wv) {
      s0 += redh[(wv * 4 + 0) * L::Hp + tid];
      s1 += redh[(wv * 4 + 1) * L::Hp + tid];
      s2 += redh[(wv * 4 + 2) * L::Hp + tid];
    }
    gMext[b * NM + (int64_t)D * H + tid] = s0;
    gw2p[b * H + tid] = s1;
    galphap[b * H + tid] = s2;
  }
  if (tid == 0)
    gb2p[b] = ((redh[3 * L::Hp] + redh[(4 + 3) * L::Hp]) + redh[(8 + 3) * L::Hp]) + redh[(12 + 3) * L::Hp];
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

// padded-D classes (ND = padded D / 16): 2, 6, 8 (one pass of the backward) and 16 (two passes of 8 tiles)
template <int NT, int ND>
static int launch_attn_fwd(const AttnArgs& a, int64_t B, int D, int H, float* scores, float* pooled, int* oob,
                           hipStream_t st) {
  constexpr size_t lds = sizeof(float) * AL<NT, ND, false>::total;
  if constexpr (lds > 150 * 1024) return REC_E_UNSUPPORTED;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_fwd_kernel<NT, ND>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((din_attn_fwd_kernel<NT, ND>), dim3((unsigned)B), dim3(256), lds, st, a, D, H, scores, pooled, oob);
  return REC_OK;
}
template <int NT, int NMT, int NPASS>
static int launch_attn_bwd(const AttnArgs& a, int64_t B, int D, int H, const float* scores, const float* gpooled,
                           float* gkeys, float* gMext, float* gw2p, float* galphap, float* gb2p, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * AL<NT, NMT * NPASS, true>::total;
  if constexpr (lds > 150 * 1024) return REC_E_UNSUPPORTED;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_bwd_kernel<NT, NMT, NPASS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((din_attn_bwd_kernel<NT, NMT, NPASS>), dim3((unsigned)B), dim3(256), lds, st, a, D, H, scores,
                     gpooled, gkeys, gMext, gw2p, galphap, gb2p);
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
  AttnArgs a{embed, ld, V, E, C, series, T, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid};
  hipStream_t st = as_stream(stream);
  const int nd = (D + 15) / 16;
  int rc;
#define FWD_ND(NT)                                                                                     \
  (nd <= 2 ? launch_attn_fwd<NT, 2>(a, B, D, H, scores, pooled, oob_flag, st)                           \
   : nd <= 6 ? launch_attn_fwd<NT, 6>(a, B, D, H, scores, pooled, oob_flag, st)                         \
   : nd <= 8 ? launch_attn_fwd<NT, 8>(a, B, D, H, scores, pooled, oob_flag, st)                         \
             : launch_attn_fwd<NT, 16>(a, B, D, H, scores, pooled, oob_flag, st))
  switch ((H + 15) / 16) {
    case 1: rc = FWD_ND(1); break;
    case 2: rc = FWD_ND(2); break;
    case 3: rc = FWD_ND(3); break;
    default: rc = FWD_ND(4); break;
  }
#undef FWD_ND
  if (rc != REC_OK) return rc;
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
  AttnArgs a{embed, ld, V, E, C, series, T, Mext, Wkd, act, alpha, mean, var, w2, b2, padding_index, mask_valid};
  hipStream_t st = as_stream(stream);
  const int nd = (D + 15) / 16;
  int rc;
#define BWD_ND(NT)                                                                                                   \
  (nd <= 2 ? launch_attn_bwd<NT, 2, 1>(a, B, D, H, scores, gpooled, gkeys, gMext, gw2p, galphap, gb2p, st)            \
   : nd <= 6 ? launch_attn_bwd<NT, 6, 1>(a, B, D, H, scores, gpooled, gkeys, gMext, gw2p, galphap, gb2p, st)          \
   : nd <= 8 ? launch_attn_bwd<NT, 8, 1>(a, B, D, H, scores, gpooled, gkeys, gMext, gw2p, galphap, gb2p, st)          \
             : launch_attn_bwd<NT, 8, 2>(a, B, D, H, scores, gpooled, gkeys, gMext, gw2p, galphap, gb2p, st))
  switch ((H + 15) / 16) {
    case 1: rc = BWD_ND(1); break;
    case 2: rc = BWD_ND(2); break;
    case 3: rc = BWD_ND(3); break;
    default: rc = BWD_ND(4); break;
  }
#undef BWD_ND
  if (rc != REC_OK) return rc;
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
