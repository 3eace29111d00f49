// fp32-exact dense GEMM on the gfx950 matrix cores: C[M,N] = epi(op(A).op(B)).
// Serves MLPLayer MatMul+BiasAdd+activation (2.FM/CustomLayers.py:74-81), Keras Dense
// (3.DCN/CustomLayers.py:158-167), MatrixCrossLayer x0*(W x_l + b) + x_l (3.DCN/CustomLayers.py:301-303)
// and the three backward GEMMs of each (dX = dY.K^T, dK = X^T.dY with split-K over the batch).
//
// v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain (no xf32/TF32
// shortcut exists on gfx950) -- what the 1e-5 parity through three multiplicative CrossNet layers needs.
// Tile 64x64x16, 4 waves as 2x2, one 32x32 accumulator (16 VGPRs) per wave.  LDS tiles are k-major
// ([k][m], [k][n]) so that an MFMA operand fetch is one conflict-free ds_read_b32 per lane.
// Loads are scalar and guarded, so any M, N, K and any leading dimension work (D = 323, 835 are odd).
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 1;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float epilogue(int epi, float acc, int64_t gm, int64_t gn, const float* bias,
                                          const float* e0, int64_t lde0, const float* e1, int64_t lde1,
                                          float* aux = nullptr, int64_t ldaux = 0) {
  switch (epi) {
    case REC_EPI_BIAS: return acc + bias[gn];
    case REC_EPI_BIAS_RELU: return fmaxf(acc + bias[gn], 0.f);
    case REC_EPI_BIAS_SIGMOID: return sigmoid_acc(acc + bias[gn]);
    case REC_EPI_BIAS_TANH: return tanhf(acc + bias[gn]);
    case REC_EPI_CROSS: {
      float u = acc + bias[gn];
      if (aux) aux[gm * ldaux + gn] = u;      // U_l = x_l W_l^T + b_l, kept for the backward pass
      return e0[gm * lde0 + gn] * u + e1[gm * lde1 + gn];
    }
    case REC_EPI_ADD: return acc + e1[gm * lde1 + gn];
    default: return acc;
  }
}

template <int TA, int TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A,
                                                       int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc, int epi,
                                                       const float* __restrict__ bias, const float* __restrict__ e0,
                                                       int64_t lde0, const float* __restrict__ e1, int64_t lde1,
                                                       int64_t kchunk, float* __restrict__ ws, float* __restrict__ aux) {
  __shared__ float As[BK][BM + PAD];
  __shared__ float Bs[BK][BN + PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
  const int64_t k_begin = (int64_t)blockIdx.z * kchunk;
  const int64_t k_end = (k_begin + kchunk < K) ? k_begin + kchunk : K;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int i = 0; i < (BM * BK) / 256; ++i) {
      int e = tid + i * 256;
      int m, k;
      if (TA == 0) { m = e / BK; k = e % BK; } else { k = e / BM; m = e % BM; }
      int64_t gm = m0 + m, gk = k0 + k;
      float v = 0.f;
      if (gm < M && gk < k_end) v = (TA == 0) ? A[gm * lda + gk] : A[gk * lda + gm];
      As[k][m] = v;
    }
#pragma unroll
    for (int i = 0; i < (BN * BK) / 256; ++i) {
      int e = tid + i * 256;
      int n, k;
      if (TB == 0) { k = e / BN; n = e % BN; } else { n = e / BK; k = e % BK; }
      int64_t gn = n0 + n, gk = k0 + k;
      float v = 0.f;
      if (gn < N && gk < k_end) v = (TB == 0) ? B[gk * ldb + gn] : B[gn * ldb + gk];
      Bs[k][n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int col = lane & 31;
  const int64_t gn = n0 + wn * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int64_t gm = m0 + wm * 32 + row;
    if (gm < M && gn < N) {
      if (ws) {
        ws[((int64_t)blockIdx.z * M + gm) * N + gn] = acc[r];
      } else {
        C[gm * ldc + gn] = epilogue(epi, acc[r], gm, gn, bias, e0, lde0, e1, lde1, aux, ldc);
      }
    }
  }
}

// ---- tall-skinny variant: N <= 64 (MLP layers of width 32, 8, 2, 1: every `units` list of the reference), A and B
// row-major.  With a single column of tiles no A element is shared between waves, so nothing is staged: lane
// (row, h) reads its 8 consecutive k values of A straight into registers and the matching B values (128-byte
// coalesced, L2-resident) -- the MFMA only needs both operands to agree on which k sits in which slot.  No LDS, no
// barrier in the K loop; the 4 waves of a workgroup split K (the 64x64 kernel ran 47 dependent load->barrier->MFMA
// rounds on 128 workgroups for [8192,741]x[741,32]: ~100 us) and add their partials in wave order.
template <int NT>
__global__ __launch_bounds__(256) void gemm_f32_skinny_kernel(int64_t M, int64_t N, int64_t K,
                                                              const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ C, int64_t ldc, int epi,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ e0, int64_t lde0,
                                                              const float* __restrict__ e1, int64_t lde1,
                                                              float* __restrict__ aux) {
  __shared__ float part[4][32][32 * NT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  const int64_t kper = ((((K + 15) >> 4) + 3) >> 2) << 4;          // 16-deep steps, split over the 4 waves
  const int64_t k_begin = wave * kper;
  const int64_t k_end = (k_begin + kper < K) ? k_begin + kper : K;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool rok = m0 + row < M;
  const float* arow = A + (m0 + row) * lda;
  for (int64_t k0 = k_begin; k0 < k_end; k0 += 32) {        // two 16-deep steps per round: 16 + 16*NT loads in flight
    float a[2][8], b[2][NT][8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int64_t kb = k0 + 16 * c + 8 * h;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bool kok = kb + s < k_end;
        a[c][s] = (rok && kok) ? arow[kb + s] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          b[c][t][s] = (kok && 32 * t + row < N) ? B[(kb + s) * ldb + 32 * t + row] : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s], b[c][t][s], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][32 * t + row] = acc[t][r];
  __syncthreads();
  for (int e = tid; e < 32 * 32 * NT; e += 256) {
    int r = e / (32 * NT), c = e - r * (32 * NT);
    int64_t gm = m0 + r;
    if (gm < M && c < N) {
      float v = (part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c]);
      C[gm * ldc + c] = epilogue(epi, v, gm, c, bias, e0, lde0, e1, lde1, aux, ldc);
    }
  }
}

// ---- large-tile variant: 128x128x16 block tile, 4 waves as 2x2, each wave a 64x64 patch = 2x2 MFMA 32x32 tiles
// (64 accumulator registers), the next K-tile prefetched into registers while the current one is multiplied (one
// barrier pair per K-tile, global latency hidden behind 32 MFMAs per wave).  Used when both M and N are large
// (CrossNet matrix mode, DIN q.Wcat); same guarded scalar loads, so any shape / leading dimension works.
constexpr int LM = 128, LN = 128, LK = 16, LPAD = 4;

template <int TA, int TB>
__global__ __launch_bounds__(256, 3) void gemm_f32_big_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A,
                                                           int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                           float* __restrict__ C, int64_t ldc, int epi,
                                                           const float* __restrict__ bias, const float* __restrict__ e0,
                                                           int64_t lde0, const float* __restrict__ e1, int64_t lde1,
                                                           int64_t kchunk, float* __restrict__ ws, float* __restrict__ aux) {
  __shared__ float As[LK][LM + LPAD];
  __shared__ float Bs[LK][LN + LPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * LM, n0 = (int64_t)blockIdx.y * LN;
  const int64_t k_begin = (int64_t)blockIdx.z * kchunk;
  const int64_t k_end = (k_begin + kchunk < K) ? k_begin + kchunk : K;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int PER = (LM * LK) / 256;   // 8 elements of each operand per thread and K-tile
  float ra[PER], rb[PER];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int e = tid + i * 256;
      int m, k;
      if (TA == 0) { m = e / LK; k = e % LK; } else { k = e / LM; m = e % LM; }
      int64_t gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < M && gk < k_end) ? ((TA == 0) ? A[gm * lda + gk] : A[gk * lda + gm]) : 0.f;
      int n, k2;
      if (TB == 0) { k2 = e / LN; n = e % LN; } else { n = e / LK; k2 = e % LK; }
      int64_t gn = n0 + n, gk2 = k0 + k2;
      rb[i] = (gn < N && gk2 < k_end) ? ((TB == 0) ? B[gk2 * ldb + gn] : B[gn * ldb + gk2]) : 0.f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int e = tid + i * 256;
      if (TA == 0) As[e % LK][e / LK] = ra[i]; else As[e / LM][e % LM] = ra[i];
      if (TB == 0) Bs[e / LN][e % LN] = rb[i]; else Bs[e % LK][e / LK] = rb[i];
    }
  };

  if (k_begin < k_end) fetch(k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += LK) {
    stage();
    __syncthreads();
    if (k0 + LK < k_end) fetch(k0 + LK);               // in flight while this tile is multiplied
#pragma unroll
    for (int kk = 0; kk < LK; kk += 2) {
      float a0 = As[kk + (lane >> 5)][wm * 64 + (lane & 31)];
      float a1 = As[kk + (lane >> 5)][wm * 64 + 32 + (lane & 31)];
      float b0 = Bs[kk + (lane >> 5)][wn * 64 + (lane & 31)];
      float b1 = Bs[kk + (lane >> 5)][wn * 64 + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  const int col = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t gn = n0 + wn * 64 + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int64_t gm = m0 + wm * 64 + i * 32 + row;
        if (gm < M && gn < N) {
          if (ws) ws[((int64_t)blockIdx.z * M + gm) * N + gn] = acc[i][j][r];
          else C[gm * ldc + gn] = epilogue(epi, acc[i][j][r], gm, gn, bias, e0, lde0, e1, lde1, aux, ldc);
        }
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int split, int64_t M,
                                                            int64_t N, float* __restrict__ C, int64_t ldc, int epi,
                                                            const float* __restrict__ bias, const float* __restrict__ e0,
                                                            int64_t lde0, const float* __restrict__ e1, int64_t lde1,
                                                            float* __restrict__ aux) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * N) return;
  int64_t gm = t / N, gn = t - gm * N;
  // fixed order: four chains (slice z -> chain z mod 4) keep several loads in flight, then one fixed tree
  const int64_t MN = M * N;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int z = 0;
  for (; z + 4 <= split; z += 4) {
    a0 += ws[(int64_t)z * MN + t];
    a1 += ws[(int64_t)(z + 1) * MN + t];
    a2 += ws[(int64_t)(z + 2) * MN + t];
    a3 += ws[(int64_t)(z + 3) * MN + t];
  }
  if (z < split) a0 += ws[(int64_t)z * MN + t];
  if (z + 1 < split) a1 += ws[(int64_t)(z + 1) * MN + t];
  if (z + 2 < split) a2 += ws[(int64_t)(z + 2) * MN + t];
  float acc = (a0 + a1) + (a2 + a3);
  C[gm * ldc + gn] = epilogue(epi, acc, gm, gn, bias, e0, lde0, e1, lde1, aux, ldc);
}

}  // namespace

extern "C" int rec_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, int epilogue_kind,
                            const float* bias, const float* e0, int64_t lde0, const float* e1, int64_t lde1,
                            int split_k, float* workspace, float* aux, void* stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0) return REC_E_ARG;
  if (epilogue_kind < REC_EPI_NONE || epilogue_kind > REC_EPI_ADD) return REC_E_ARG;
  if (epilogue_kind >= REC_EPI_BIAS && epilogue_kind <= REC_EPI_CROSS && !bias) return REC_E_ARG;
  if (epilogue_kind == REC_EPI_CROSS && (!e0 || !e1)) return REC_E_ARG;
  if (epilogue_kind == REC_EPI_ADD && !e1) return REC_E_ARG;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return REC_E_ARG;
  if (M == 0 || N == 0) return REC_OK;
  if (split_k < 1) split_k = 1;
  if (split_k > 1 && !workspace) return REC_E_WORKSPACE;
  int64_t kchunk = K;
  if (split_k > 1) {
    kchunk = ((ceil_div64(K, split_k) + BK - 1) / BK) * BK;
    if (kchunk < BK) kchunk = BK;
    split_k = (int)ceil_div64(K, kchunk);
    if (split_k < 1) split_k = 1;
  }
  float* ws = split_k > 1 ? workspace : nullptr;
  hipStream_t st = as_stream(stream);
  // 128x128 tiles (register prefetch, 4x the MFMA work per barrier) only when they fill a good part of the chip;
  // a mid-size product ([4096,352] x [352,200]: 64 big tiles) runs on 64x64 tiles instead (256 workgroups)
  const bool big = M > 64 && N > 64 && ceil_div64(M, LM) * ceil_div64(N, LN) * split_k >= 160;
  if (!transA && !transB && N <= 64 && M >= 256 && split_k == 1) {
    dim3 grid((unsigned)ceil_div64(M, 32));
    if (N <= 32)
      hipLaunchKernelGGL((gemm_f32_skinny_kernel<1>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc,
                         epilogue_kind, bias, e0, lde0, e1, lde1, aux);
    else
      hipLaunchKernelGGL((gemm_f32_skinny_kernel<2>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc,
                         epilogue_kind, bias, e0, lde0, e1, lde1, aux);
    REC_LAUNCH_CHECK();
    return REC_OK;
  }
  if (big) {
    dim3 grid((unsigned)ceil_div64(M, LM), (unsigned)ceil_div64(N, LN), (unsigned)split_k);
    if (grid.y > 65535u) return REC_E_UNSUPPORTED;
#define LAUNCHB(TA, TB)                                                                                       \
  hipLaunchKernelGGL((gemm_f32_big_kernel<TA, TB>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc,  \
                     epilogue_kind, bias, e0, lde0, e1, lde1, kchunk, ws, aux)
    if (!transA && !transB) LAUNCHB(0, 0);
    else if (!transA && transB) LAUNCHB(0, 1);
    else if (transA && !transB) LAUNCHB(1, 0);
    else LAUNCHB(1, 1);
#undef LAUNCHB
  } else {
    dim3 grid((unsigned)ceil_div64(M, BM), (unsigned)ceil_div64(N, BN), (unsigned)split_k);
    if (grid.y > 65535u) return REC_E_UNSUPPORTED;
#define LAUNCH(TA, TB)                                                                                        \
  hipLaunchKernelGGL((gemm_f32_kernel<TA, TB>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc,      \
                     epilogue_kind, bias, e0, lde0, e1, lde1, kchunk, ws, aux)
    if (!transA && !transB) LAUNCH(0, 0);
    else if (!transA && transB) LAUNCH(0, 1);
    else if (transA && !transB) LAUNCH(1, 0);
    else LAUNCH(1, 1);
#undef LAUNCH
  }
  REC_LAUNCH_CHECK();
  if (ws) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div64(M * N, 256)), dim3(256), 0, st, ws, split_k,
                       M, N, C, ldc, epilogue_kind, bias, e0, lde0, e1, lde1, aux);
    REC_LAUNCH_CHECK();
  }
  return REC_OK;
}
