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

  // the next K-tile travels (global -> registers) while the current one is multiplied: with one 64x64 tile per CU -- a
  // mid-size product such as [4096,352] x [352,200] fills the chip exactly once -- nothing else hides the ~2 us of a
  // load -> LDS -> barrier -> MFMA round (22 rounds: 51 us for 0.58 GFLOP)
  constexpr int PERA = (BM * BK) / 256, PERB = (BN * BK) / 256;
  float ra[PERA], rb[PERB];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < PERA; ++i) {
      int e = tid + i * 256;
      int m, k;
      if (TA == 0) { m = e / BK; k = e % BK; } else { k = e / BM; m = e % BM; }
      int64_t gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < M && gk < k_end) ? ((TA == 0) ? A[gm * lda + gk] : A[gk * lda + gm]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PERB; ++i) {
      int e = tid + i * 256;
      int n, k;
      if (TB == 0) { k = e / BN; n = e % BN; } else { n = e / BK; k = e % BK; }
      int64_t gn = n0 + n, gk = k0 + k;
      rb[i] = (gn < N && gk < k_end) ? ((TB == 0) ? B[gk * ldb + gn] : B[gn * ldb + gk]) : 0.f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < PERA; ++i) {
      int e = tid + i * 256;
      if (TA == 0) As[e % BK][e / BK] = ra[i]; else As[e / BM][e % BM] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PERB; ++i) {
      int e = tid + i * 256;
      if (TB == 0) Bs[e / BN][e % BN] = rb[i]; else Bs[e % BK][e / BK] = rb[i];
    }
  };
  if (k_begin < k_end) fetch(k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < k_end) fetch(k0 + BK);
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
  __shared__ __attribute__((aligned(16))) float As[LK][LM + LPAD];
  __shared__ __attribute__((aligned(16))) float Bs[LK][LN + LPAD];
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
  constexpr int PER4 = PER / 4;          // ... as 16-byte pieces when the operand is k-major in memory
  float ra[PER], rb[PER];
  // A k-major operand (A with TA = 1: A[k][m]; B with TB = 0: B[k][n]) is fetched as 16-byte pieces along its
  // contiguous index -- 4-byte aligned, the rows of a [*, 835] matrix start anywhere -- and staged with one 16-byte LDS
  // store: a quarter of the load and store instructions of the element-wise path, which remains for the pieces that
  // straddle the edge of the matrix or of the K range.  (H^T.X at D = 835: 358 -> see DESIGN.md section 5.)
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  auto fetch_kmajor = [&](const float* __restrict__ P, int64_t ld, int64_t x0, int64_t X, int64_t k0, float* r) {
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int e = tid + i * 256;
      const int k = e / (LM / 4), x = 4 * (e % (LM / 4));
      const int64_t gx = x0 + x, gk = k0 + k;
      if (gx + 3 < X && gk < k_end) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(P + gk * ld + gx);
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) r[4 * i + c] = (gx + c < X && gk < k_end) ? P[gk * ld + gx + c] : 0.f;
      }
    }
  };
  auto fetch = [&](int64_t k0) {
    if (TA == 1) fetch_kmajor(A, lda, m0, M, k0, ra);
    if (TB == 0) fetch_kmajor(B, ldb, n0, N, k0, rb);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int e = tid + i * 256;
      if (TA == 0) {
        int m = e / LK, k = e % LK;
        int64_t gm = m0 + m, gk = k0 + k;
        ra[i] = (gm < M && gk < k_end) ? A[gm * lda + gk] : 0.f;
      }
      if (TB == 1) {
        int n = e / LK, k2 = e % LK;
        int64_t gn = n0 + n, gk2 = k0 + k2;
        rb[i] = (gn < N && gk2 < k_end) ? B[gn * ldb + gk2] : 0.f;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int e = tid + i * 256;
      const int k = e / (LM / 4), x = 4 * (e % (LM / 4));
      if (TA == 1) *reinterpret_cast<float4*>(&As[k][x]) = make_float4(ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]);
      if (TB == 0) *reinterpret_cast<float4*>(&Bs[k][x]) = make_float4(rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int e = tid + i * 256;
      if (TA == 0) As[e % LK][e / LK] = ra[i];
      if (TB == 1) Bs[e % LK][e / LK] = rb[i];
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

// ---- row-panel variant for a tall product with a modest N (CrossNet matrix mode: M = batch 16384, N = K = 323 / 835;
// every layer of it and its dX = H.W backward): ONE workgroup owns 64 rows x ALL N columns, so the grid is M/64
// workgroups -- exactly one per CU at M = 16384 -- and every CU multiplies the same number of 32x32 blocks (a 128x128
// tiling of [16384, 835] leaves 896 tiles for 256 CUs: a fourth round that is 17 % full, and 7 % of padding columns).
// 8 waves = 2 row blocks x 4 column groups; column block j belongs to group j & 3, a wave holds up to 7 accumulator
// tiles (112 VGPRs) and reads 1 A + up to 7 B operands per 2-deep k step.  A (64 x 16) and B (16 x N) slabs are
// double-buffered in LDS (k-major, conflict-free operand reads), the next slab's global loads are in flight while the
// current one is multiplied (one barrier per slab).  B is re-read from L2 by every workgroup (N*K*4 bytes <= 2.8 MB).
constexpr int PNB = 28, PPAD = 4;      // max column blocks (N <= 896), LDS row pad

// PMt rows per workgroup (32 or 64), PKt slab depth, MAXB column blocks per wave; threads = PMt * 8 (waves = PMt/32 row
// blocks x 4 column groups).  <64,16>: one 8-wave workgroup per CU with 124 KB of LDS; <32,8>: two 4-wave workgroups
// per CU (60 KB each) whose barriers and staging phases overlap each other's MFMAs.
template <int TA, int TB, int MAXB, int PMt, int PKt>
__global__ __launch_bounds__(PMt * 8) void gemm_f32_panel_kernel(int64_t M, int64_t N, int64_t Kall,
                                                                 const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc, int epi,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ e0, int64_t lde0,
                                                                 const float* __restrict__ e1, int64_t lde1,
                                                                 float* __restrict__ aux, int64_t kchunk,
                                                                 float* __restrict__ ws, int64_t ncp) {
  // column panels (ncp > 0: a short-K product whose N exceeds one panel, e.g. DIN's q.Wcat with K = 96, N = 3492):
  // blockIdx.z takes columns [z*ncp, z*ncp + ncp) -- everything column-indexed moves, the A panel is read again
  if (ncp > 0) {
    const int64_t n_off = (int64_t)blockIdx.z * ncp;
    B += TB ? n_off * ldb : n_off;
    C += n_off;
    if (bias) bias += n_off;
    if (e0) e0 += n_off;
    if (e1) e1 += n_off;
    if (aux) aux += n_off;
    N = N - n_off < ncp ? N - n_off : ncp;
  }
  // split-K (weight gradient H^T.X: M = N = D, K = batch): blockIdx.y takes k in [kb, kb + K) and leaves its partial in
  // ws[blockIdx.y]; splitk_reduce_kernel adds the slices in order.  Inside the kernel K and the operand pointers are
  // those of the slice.
  const int64_t kb = (int64_t)blockIdx.y * kchunk;
  const int64_t K = (kb + kchunk < Kall ? kb + kchunk : Kall) - kb;
  A += TA ? kb * lda : kb;
  B += TB ? kb : kb * ldb;
  // every wave multiplies exactly MAXB blocks per k step (no guards around the MFMAs); blocks beyond ceil(N/32) read
  // zero columns of the slab
  extern __shared__ __attribute__((aligned(16))) float plds[];
  constexpr int NTH = PMt * 8;
  constexpr int NP = MAXB * 4 * 32;                  // padded column count
  constexpr int bs = NP + PPAD;                      // row stride of a B slab
  constexpr int as_ld = PMt + PPAD;
  constexpr int KQ = PKt / 4;                        // float4 pieces along k
  float* As = plds;                                  // [2][PKt][as_ld]
  float* Bs = plds + 2 * PKt * as_ld;                // [2][PKt][bs]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % (PMt / 32), wn = wave / (PMt / 32);
  const int lo = lane & 31, hi = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * PMt;
  f32x16 acc[MAXB];
#pragma unroll
  for (int j = 0; j < MAXB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // global -> register staging of one slab.  A (row-major): PMt rows x PKt k as float4 pieces along k.  B, TB = 1
  // (B[k][n] = Bm[n][k], rows of K contiguous): float4 along k, item = (n, k4).  B, TB = 0 (B[k][n] = Bm[k][n], rows of N
  // contiguous): one float per item, for every k of the slab the thread takes columns tid, tid + NTH, ...
  // Loads are unconditional (clamped addresses; what lies outside M x N never reaches the result) and 16 bytes wide at
  // 4-byte alignment (rows of an odd K = 323 / 835 start anywhere): a guarded element-wise load would put a branch
  // around every load and wait for each before the next.  Only the last, partial slab takes the element-wise path.
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr int BMAX = (NP * KQ + NTH - 1) / NTH;    // float4 items per thread, TB = 1
  constexpr int NC = (NP + NTH - 1) / NTH;           // columns per thread, TB = 0
  constexpr int AITEMS = PMt * KQ;                   // TA = 0: float4 pieces of the A slab
  constexpr int AT = (PMt * PKt + NTH - 1) / NTH;    // TA = 1: floats per thread (A is k-major in memory: [K][M])
  float4 ra;
  float rat[TA ? AT : 1];
  float4 rbv[TB ? BMAX : 1];
  float rbs[TB ? 1 : PKt][TB ? 1 : NC];
  const int n_items = (int)N * KQ;                   // TB = 1
  const int am = tid / KQ, ak4 = (tid % KQ) * 4;
  const float* arow = TA ? A : A + (m0 + am < M ? m0 + am : M - 1) * lda;
  const int tm = tid % PMt, tk = tid / PMt;          // TA = 1: column m and first k row of this thread
  const int64_t tmc = m0 + tm < M ? m0 + tm : M - 1;
  int bc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) bc[c] = tid + c * NTH < (int)N ? tid + c * NTH : (int)N - 1;
  // The staging work of a slab is cut into NCH = PKt/2 chunks, one per k step of the multiply loop: chunk c writes its
  // share of the NEXT slab (held in registers since the slab before) to the other LDS buffer and at once re-issues the
  // global loads of the slab after that into the same registers.  Placed between the MFMA groups, the LDS writes, the
  // loads and their address arithmetic run under the matrix pipe instead of after it (all eight waves stage at the same
  // time -- behind the one barrier per slab -- so nothing else would hide them).  The pipelined loop only sees FULL
  // slabs and has no data-dependent branch (a slab index past the end is clamped to the last full slab: loaded again,
  // staged into a buffer nobody reads); the K % PKt tail is one plain, guarded step after it.
  constexpr int NCH = PKt / 2;
  constexpr int BPC = (BMAX + NCH - 1) / NCH;        // TB = 1: float4 items per chunk
  constexpr int KPC = PKt / NCH;                     // TB = 0: k rows per chunk (= 2)
  auto fetch_chunk = [&](int64_t k0, int c) {
    if (TA) {
      if (c == NCH - 1) {
#pragma unroll
        for (int q = 0; q < AT; ++q) rat[q] = A[(k0 + tk + q * (NTH / PMt)) * lda + tmc];
      }
    } else if (c == NCH - 1 && tid < AITEMS) {
      const f32x4u v = *reinterpret_cast<const f32x4u*>(arow + k0 + ak4);
      ra = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (TB) {
#pragma unroll
      for (int q = 0; q < BPC; ++q) {
        const int i = c * BPC + q;
        if (i < BMAX) {
          const int it = tid + i * NTH;
          const int itc = it < n_items ? it : n_items - 1;
          const int n = itc / KQ, k4 = (itc % KQ) * 4;
          const f32x4u v = *reinterpret_cast<const f32x4u*>(B + (int64_t)n * ldb + k0 + k4);
          rbv[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < KPC; ++q) {
        const int k = c * KPC + q;
        const float* p = B + (k0 + k) * ldb;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) rbs[k][cc] = p[bc[cc]];
      }
    }
  };
  auto stage_chunk = [&](int buf, int c) {
    float* as = As + buf * PKt * as_ld;
    float* bsb = Bs + buf * PKt * bs;
    if (TA) {
      if (c == NCH - 1) {
#pragma unroll
        for (int q = 0; q < AT; ++q) as[(tk + q * (NTH / PMt)) * as_ld + tm] = rat[q];
      }
    } else if (c == NCH - 1 && tid < AITEMS) {
      as[(ak4 + 0) * as_ld + am] = ra.x;
      as[(ak4 + 1) * as_ld + am] = ra.y;
      as[(ak4 + 2) * as_ld + am] = ra.z;
      as[(ak4 + 3) * as_ld + am] = ra.w;
    }
    if (TB) {
#pragma unroll
      for (int q = 0; q < BPC; ++q) {
        const int i = c * BPC + q;
        const int it = tid + i * NTH;
        if (i < BMAX && it < n_items) {
          const int n = it / KQ, k4 = (it % KQ) * 4;
          bsb[(k4 + 0) * bs + n] = rbv[i].x;
          bsb[(k4 + 1) * bs + n] = rbv[i].y;
          bsb[(k4 + 2) * bs + n] = rbv[i].z;
          bsb[(k4 + 3) * bs + n] = rbv[i].w;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < KPC; ++q) {
        const int k = c * KPC + q;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
          if (tid + cc * NTH < (int)N) bsb[k * bs + tid + cc * NTH] = rbs[k][cc];
      }
    }
  };
  // columns N .. NP-1 of both B buffers are read by the last column blocks and never staged: zero them once
  for (int i = tid; i < 2 * PKt * (NP - (int)N); i += NTH) {
    const int r = i / (NP - (int)N), cix = i - r * (NP - (int)N);
    Bs[r * bs + (int)N + cix] = 0.f;
  }

  // multiply one slab out of LDS buffer `buf`; operands of k step s+1 are read while the MFMAs of step s issue
  // (explicit two-deep register pipeline); hook(c) runs behind the MFMA group of k step c
  auto multiply = [&](int buf, auto hook) {
    const float* as = As + buf * PKt * as_ld + wm * 32 + lo;
    const float* bsb = Bs + buf * PKt * bs + wn * 32 + lo;
    float a_c = as[hi * as_ld], b_c[MAXB];
#pragma unroll
    for (int j = 0; j < MAXB; ++j) b_c[j] = bsb[hi * bs + j * 128];
#pragma unroll
    for (int kk = 0; kk < PKt; kk += 2) {
      float a_n = 0.f, b_n[MAXB];
      if (kk + 2 < PKt) {
        a_n = as[(kk + 2 + hi) * as_ld];
#pragma unroll
        for (int j = 0; j < MAXB; ++j) b_n[j] = bsb[(kk + 2 + hi) * bs + j * 128];
      }
      __builtin_amdgcn_sched_barrier(0);               // keep the reads above in front of this step's MFMAs
#pragma unroll
      for (int j = 0; j < MAXB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[j], acc[j], 0, 0, 0);
      hook(kk >> 1);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 2 < PKt) {
        a_c = a_n;
#pragma unroll
        for (int j = 0; j < MAXB; ++j) b_c[j] = b_n[j];
      }
    }
  };

  const int64_t Kfull = (K / PKt) * PKt;
  int buf = 0;
  if (Kfull > 0) {
    const int64_t klast = Kfull - PKt;
    // prologue: slab 0 -> LDS buffer 0, slab 1 -> registers
#pragma unroll
    for (int c = 0; c < NCH; ++c) fetch_chunk(0, c);
#pragma unroll
    for (int c = 0; c < NCH; ++c) stage_chunk(0, c);
    {
      const int64_t k1 = PKt < klast ? PKt : klast;
#pragma unroll
      for (int c = 0; c < NCH; ++c) fetch_chunk(k1, c);
    }
    __syncthreads();
    for (int64_t k0 = 0; k0 < Kfull; k0 += PKt) {
      const int64_t k2 = k0 + 2 * PKt < klast ? k0 + 2 * PKt : klast;
      multiply(buf, [&](int c) {
        stage_chunk(buf ^ 1, c);                         // next slab: registers -> the other buffer (free since the barrier)
        fetch_chunk(k2, c);                              // the slab after next into the freed registers
      });
      __syncthreads();
      buf ^= 1;
    }
  }
  if (Kfull < K) {
    // tail slab (K % PKt deep): guarded element-wise loads straight to LDS, zero beyond K
    float* as = As + buf * PKt * as_ld;
    float* bsb = Bs + buf * PKt * bs;
    for (int i = tid; i < PMt * PKt; i += NTH) {
      const int m = i / PKt, k = i - m * PKt;
      const int64_t gm = m0 + m, gk = Kfull + k;
      as[k * as_ld + m] = (gm < M && gk < K) ? (TA ? A[gk * lda + gm] : A[gm * lda + gk]) : 0.f;
    }
    for (int i = tid; i < PKt * (int)N; i += NTH) {
      int k, n;
      if (TB) { n = i / PKt; k = i - n * PKt; } else { k = i / (int)N; n = i - k * (int)N; }
      const int64_t gk = Kfull + k;
      bsb[k * bs + n] = gk < K ? (TB ? B[(int64_t)n * ldb + gk] : B[gk * ldb + n]) : 0.f;
    }
    __syncthreads();
    multiply(buf, [](int) {});
  }

  // epilogue, one column block at a time: the kind is tested once per block (not per element), the side operands of
  // the block's 16 rows are all requested before the first is used, addresses are clamped instead of guarded
#pragma unroll
  for (int j = 0; j < MAXB; ++j) {
    const int64_t gn = (int64_t)(wn + 4 * j) * 32 + lo;
    const bool cok = gn < N;
    const int64_t gnc = cok ? gn : N - 1;
    const int64_t mb = m0 + wm * 32 + 4 * hi;                  // row of register r: mb + (r&3) + 8*(r>>2)
    if (ws) {                                                  // split-K slice: plain partial, no epilogue
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gm = mb + (r & 3) + 8 * (r >> 2);
        if (gm < M && cok) ws[((int64_t)blockIdx.y * M + gm) * N + gn] = acc[j][r];
      }
      continue;
    }
    const float bj = (epi >= REC_EPI_BIAS && epi <= REC_EPI_CROSS) ? bias[gnc] : 0.f;
    float x0[16], x1[16];
    if (epi == REC_EPI_CROSS || epi == REC_EPI_ADD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t gm = mb + (r & 3) + 8 * (r >> 2);
        gm = gm < M ? gm : M - 1;
        x1[r] = e1[gm * lde1 + gnc];
        x0[r] = epi == REC_EPI_CROSS ? e0[gm * lde0 + gnc] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gm = mb + (r & 3) + 8 * (r >> 2);
      const float u = acc[j][r] + bj;
      float v = u;
      if (epi == REC_EPI_BIAS_RELU) v = fmaxf(u, 0.f);
      else if (epi == REC_EPI_BIAS_SIGMOID) v = sigmoid_acc(u);
      else if (epi == REC_EPI_BIAS_TANH) v = tanhf(u);
      else if (epi == REC_EPI_CROSS) v = x0[r] * u + x1[r];
      else if (epi == REC_EPI_ADD) v = u + x1[r];
      if (gm < M && cok) {
        if (epi == REC_EPI_CROSS && aux) aux[gm * ldc + gn] = u;
        C[gm * ldc + gn] = v;
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
  // row panels x all N: a tall product with a modest N and row-major A (CrossNet matrix mode and its dX, split_k = 1)
  // ... or a wider N in several column panels when K is short (the A panel is then cheap to read once per column panel)
  const int64_t ncol_panels = N <= PNB * 32 ? 1 : ceil_div64(N, PNB * 32);
  const int64_t ncp = ncol_panels == 1 ? 0 : ((ceil_div64(N, ncol_panels) + 31) / 32) * 32;
  const bool panel_fwd = !transA && split_k == 1 && N > 64 &&
                         ((ncol_panels == 1 && M >= 64 * 128) ||
                          (ncol_panels > 1 && ncol_panels <= 64 && K <= 256 && ceil_div64(M, 64) * ncol_panels >= 192));
  // (measured and not taken: M = N = 835, K = 16384 as 14 panels x 19 K slices ran 456 us against 358 us on the 128x128
  // split-K tiles -- every slice's 2.9 MB of X is fetched by the L2 of each of the 8 XCDs its 14 panels land on)
  if (panel_fwd) {
    const int maxb = (int)ceil_div64(ceil_div64(ncp > 0 ? ncp : N, 32), 4);   // column blocks per wave, 1..7
    constexpr int PMt = 64, PKt = 16;
    const size_t lds = sizeof(float) * (2 * PKt * (PMt + PPAD) + 2 * PKt * ((size_t)maxb * 128 + PPAD));
    const int64_t panels = ceil_div64(M, PMt);
    // (the kernel can also take K slices of a k-major A -- blockIdx.y, partials in ws -- for the weight gradient H^T.X;
    // measured slower than the 128x128 split-K tiles, see above, so no slice is ever launched from here)
    const int psplit = 1;
    const int64_t pchunk = K;
    float* pws = nullptr;
    dim3 grid((unsigned)panels, (unsigned)psplit, (unsigned)ncol_panels);
#define PANEL(TAv, TBv, MB)                                                                                          \
  do {                                                                                                               \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_panel_kernel<TAv, TBv, MB, PMt, PKt>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    if (e_ != hipSuccess) return (int)e_;                                                                            \
    hipLaunchKernelGGL((gemm_f32_panel_kernel<TAv, TBv, MB, PMt, PKt>), grid, dim3(PMt * 8), lds, st, M, N, K, A, lda, \
                       B, ldb, C, ldc, epilogue_kind, bias, e0, lde0, e1, lde1, aux, pchunk, pws, ncp);              \
  } while (0)
#define PANEL_MB(TAv, TBv)                                                                                           \
  switch (maxb) {                                                                                                    \
    case 1: PANEL(TAv, TBv, 1); break;                                                                               \
    case 2: PANEL(TAv, TBv, 2); break;                                                                               \
    case 3: PANEL(TAv, TBv, 3); break;                                                                               \
    case 4: PANEL(TAv, TBv, 4); break;                                                                               \
    case 5: PANEL(TAv, TBv, 5); break;                                                                               \
    case 6: PANEL(TAv, TBv, 6); break;                                                                               \
    default: PANEL(TAv, TBv, 7); break;                                                                              \
  }
    if (transB) { PANEL_MB(0, 1); } else { PANEL_MB(0, 0); }
#undef PANEL_MB
#undef PANEL
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
