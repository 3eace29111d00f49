// Fused DeepFM train step (2.FM/CustomLayers.py:279-308 under 2.FM/ModelManager.py:171-177) for the reference's
// default head (embedding_dims = 16, mlp_dims = [32, 8]) on the fused [embed(16) | w | pad] 128-byte row layout.
//
// The generic path runs ~35 launch-bound kernels per step (profiles/r01_v1_*): at batch 8192 every small kernel costs
// ~5 us while the whole gather is ~4 us of HBM time.  Here a step is two launches on the main stream plus the sort of an
// upcoming batch's ids on a second one:
//
//   deepfm_fwd_bwd_kernel   one workgroup (4 waves) per 32 examples: index assembly straight from the F feature
//                           columns, gather of the 128-B rows (26 independent 16-B loads in flight per lane), FM,
//                           MLP 416->32 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, operands from LDS, K split
//                           over the waves), the two small layers + sigmoid + Keras BCE + their backward on the VALU,
//                           dX = dpre1 . K0^T and the per-workgroup dK0 partial on the matrix cores (one 32-wide tile
//                           per wave at a time), and the IndexedSlices values dz*(S - e) + dX written once as full
//                           128-byte lines.  The embedding rows never leave LDS.
//   deepfm_post_kernel      ONE launch, two jobs side by side: the fixed-order sum of the per-workgroup partials (dK0,
//                           dK1, biases, loss) and the segment sums of embed and w gradients over the batch's
//                           de-duplication plan + global compaction (column counts prefix).  (deepfm_reduce_kernel
//                           and colseg_sum_kernel are the same two jobs as separate launches.)
//   colsort_*_kernel (3)    de-duplication plan: the DataGenerator contract (2.FM/DataGenerator.py:76-88) gives every
//                           feature column its own contiguous id range, so duplicates only occur inside a column:
//                           each column (B <= 16384 ids) is sorted on its own as 32-bit (key << PB | position) words
//                           -- 1024-id chunks by a bitonic network (registers / wave shuffles / LDS), chunks merged
//                           by ranking (binary searches in LDS), then run detection, all on a (B/1024) x columns
//                           grid.  Depends on ids only: the engine runs it ahead, on a second stream, for up to two
//                           upcoming batches per call.
//
// Everything is deterministic (no float atomics): per-workgroup partials + fixed-order reductions, stable sort keys.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EX = 32;        // examples per workgroup
constexpr int E16 = 16;       // embedding dims
constexpr int LD = 32;        // fused row stride (floats)
constexpr int U1 = 32, U2 = 8;
constexpr int HS = 33;        // padded row stride of the [*,32] LDS tiles
constexpr int SMALL = 320;    // floats of small partials per workgroup

struct Cols {
  const int64_t* p[REC_MAX_COLS];
};

struct FusedArgs {
  const float* table;         // fused rows [V, 32]
  int64_t V;
  const float* bias;
  const float* K0; const float* b0;   // [F*16,32], [32]
  const float* K1; const float* b1;   // [32,8], [8]
  const float* K2; const float* b2;   // [8,1], [1]
  const float* label;         // [B]
  int64_t B; int F;
  float* gz;                  // [B]     dL/dz
  float* vals;                // [B*F,16] IndexedSlices values of embed
  float* prob;                // [B] or null
  float* dK0part;             // [nwg, F*16*32]
  float* small;               // [nwg, SMALL]
  int* oob;
  int stop;                   // diagnostics only (REC_FUSED_STOP): leave after phase `stop` (0 = run everything)
};

__device__ __forceinline__ int xs_of(int F) { return F * E16 + 2; }

__global__ __launch_bounds__(256) void deepfm_fwd_bwd_kernel(Cols cols, FusedArgs a) {
  extern __shared__ float lds[];
  const int F = a.F, D = F * E16, XS = xs_of(F);
  float* XT = lds;                       // [EX][XS]      gathered embedding rows (later: dK0 exchange scratch)
  float* K0s = XT + EX * XS;             // [D][HS]
  float* H1s = K0s + D * HS;             // [EX][HS]      relu(h1)
  float* DP1 = H1s + EX * HS;            // [EX][HS]      d pre-activation of layer 1
  float* Ss = DP1 + EX * HS;             // [EX][16]
  // ids (P0-P1) and the first-order weights Wl (P1-P2) are dead before H1s / DP1 are first written (P3, P4):
  // they share that region (2*EX*F <= 2*EX*HS for F <= 33), which keeps the workgroup at ~134 KB of LDS so that a
  // 32-KB sort workgroup of the second stream can be co-resident on the CU
  float* Wl = H1s;                       // [EX*F]
  int* ids = reinterpret_cast<int*>(H1s + EX * F);   // [EX*F]
  float* h2s = Ss + EX * E16;            // [EX][8]
  float* dp2s = h2s + EX * U2;           // [EX][8]
  float* zfm = dp2s + EX * U2;           // [EX]
  float* dzs = zfm + EX;                 // [EX]
  float* lss = dzs + EX;                 // [EX]
  float* K1s = lss + EX;                 // [32][8]
  float* b0s = K1s + U1 * U2;            // [32]
  float* b1s = b0s + U1;                 // [8]
  float* K2s = b1s + U2;                 // [8]
  float* PT = K2s + U2;                  // [4 waves][EX][HS]  K-split partial tiles of layer 1 (P3)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ex0 = (int64_t)blockIdx.x * EX;
  const int n_ex = (a.B - ex0 < EX) ? (int)(a.B - ex0) : EX;

  // ---- P0: K0 (53 KB, the same for every workgroup) starts its trip from L2 into registers first, so that its
  // latency hides behind the id fetch and the row gather; ids of the 32 examples from the F feature columns go to
  // LDS as ids[f*32 + e] (one field per gather pass: no integer division anywhere)
  constexpr int MAXK = 14;                                   // F <= 28: D*8/256 = F/2 float4 per thread
  float4 kreg[MAXK];
#pragma unroll
  for (int q = 0; q < MAXK; ++q) {
    int i = tid + q * 256;
    kreg[q] = i < D * (U1 / 4) ? reinterpret_cast<const float4*>(a.K0)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  bool bad = false;
  for (int i = tid; i < EX * F; i += 256) {
    int f = i >> 5, e = i & 31;              // consecutive threads read consecutive examples of one column
    int v = -1;
    if (e < n_ex) {
      int64_t id = cols.p[f][ex0 + e];
      if ((uint64_t)id < (uint64_t)a.V) v = (int)id; else bad = true;
    }
    ids[i] = v;
  }
  if (bad && a.oob) *a.oob = 1;
  K1s[tid] = a.K1[tid];
  if (tid < U1) b0s[tid] = a.b0[tid];
  if (tid < U2) { b1s[tid] = a.b1[tid]; K2s[tid] = a.K2[tid]; }
  __syncthreads();
  if (a.stop == 1) return;

  // ---- P1: gather.  8 lanes x 16 B cover one 128-B row; pass `it` = field it of the 32 examples; every pass's
  // load is issued before anything is consumed (F independent 16-B loads in flight per lane)
  {
    const int c = tid & 7, e = tid >> 3;
    constexpr int MAXP = 28;
    float4 v[MAXP];
#pragma unroll
    for (int it = 0; it < MAXP; ++it) {
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < F && c <= 4) {
        int id = ids[it * 32 + e];
        if (id >= 0) v[it] = *reinterpret_cast<const float4*>(a.table + (int64_t)id * LD + 4 * c);
      }
    }
    // K0 registers -> LDS (rows of 33 floats) while the rows are in flight
#pragma unroll
    for (int q = 0; q < MAXK; ++q) {
      int i = tid + q * 256;
      if (i < D * (U1 / 4)) {
        float* dst = K0s + (i >> 3) * HS + (i & 7) * 4;
        dst[0] = kreg[q].x; dst[1] = kreg[q].y; dst[2] = kreg[q].z; dst[3] = kreg[q].w;
      }
    }
    float* xrow = XT + e * XS + 4 * c;                       // 8-byte aligned only (XS even): two 8-byte stores
#pragma unroll
    for (int it = 0; it < MAXP; ++it) {
      if (it < F) {
        if (c < 4) {
          float* dst = xrow + it * E16;
          reinterpret_cast<float2*>(dst)[0] = make_float2(v[it].x, v[it].y);
          reinterpret_cast<float2*>(dst)[1] = make_float2(v[it].z, v[it].w);
        } else if (c == 4) {
          Wl[it * 32 + e] = v[it].x;
        }
      }
    }
  }
  __syncthreads();
  if (a.stop == 2) return;

  // ---- P2: FM terms.  thread = (example e, dim pair d2)
  {
    const int e = tid >> 3, d2 = tid & 7;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    const float* xr = XT + e * XS + 2 * d2;
    for (int f = 0; f < F; ++f) {
      float2 x = *reinterpret_cast<const float2*>(xr + f * E16);
      s0 += x.x; s1 += x.y;
      q0 += x.x * x.x; q1 += x.y * x.y;
    }
    Ss[e * E16 + 2 * d2] = s0;
    Ss[e * E16 + 2 * d2 + 1] = s1;
    float part = (s0 * s0 - q0) + (s1 * s1 - q1);
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (d2 == 0) {
      float first = 0.f;
      for (int f = 0; f < F; ++f) first += Wl[f * 32 + e];
      zfm[e] = a.bias[0] + first + 0.5f * part;
    }
  }
  __syncthreads();                         // Wl / ids are dead from here on: their LDS becomes H1s / DP1
  if (a.stop == 3) return;

  // ---- P3: h1 = relu(X . K0 + b0) on the matrix cores, 32x32x2 tiles.  The output is ONE 32x32 tile, so the four
  // waves split K = 16F: wave w multiplies columns [4F*w, 4F*(w+1)) of X with the matching rows of K0 (2F steps of 2),
  // the four partial tiles meet in LDS and are added in wave order.  Lane l: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
  // D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
  const int lo = lane & 31, hi = lane >> 5;
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kbeg = wave * 4 * F;
    const float* ap = XT + lo * XS + kbeg + hi;              // + 2*step
    const float* bp = K0s + (kbeg + hi) * HS + lo;           // + 2*step*HS
    const int ns = 2 * F;                                    // steps of this wave
    // groups of 4 steps, software-pipelined with a STATIC number of LDS loads in flight (no guarded loads: a guard
    // makes the count of outstanding loads unknown to the compiler, which then waits for all of them before the
    // first MFMA of every group); the last group and the 2-step tail of an odd F are peeled
    const int ng = ns >> 2;
    float ac[4], bc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ac[u] = ap[2 * u]; bc[u] = bp[2 * u * HS]; }      // F >= 2: group 0 exists
    for (int gi = 0; gi + 1 < ng; ++gi) {
      float an[4], bn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        an[u] = ap[2 * (4 * (gi + 1) + u)];
        bn[u] = bp[2 * (4 * (gi + 1) + u) * HS];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u], acc, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { ac[u] = an[u]; bc[u] = bn[u]; }
    }
    if (ng > 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u], acc, 0, 0, 0);
    }
    for (int sn = 4 * ng; sn < ns; ++sn)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * sn], bp[2 * sn * HS], acc, 0, 0, 0);
    float* pt = PT + wave * (EX * HS);
#pragma unroll
    for (int r = 0; r < 16; ++r) pt[((r & 3) + 8 * (r >> 2) + 4 * hi) * HS + lo] = acc[r];
  }
  __syncthreads();
  {
    const int e = tid >> 3, u4 = (tid & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int o = e * HS + u4 + q;
      float h = ((PT[o] + PT[EX * HS + o]) + PT[2 * EX * HS + o]) + PT[3 * EX * HS + o];     // wave order: fixed
      H1s[o] = fmaxf(h + b0s[u4 + q], 0.f);
    }
  }
  __syncthreads();
  if (a.stop == 4) return;

  // ---- P4: layers 32->8->1, sigmoid, Keras BCE and their backward.  thread = (example e, unit u)
  {
    const int e = tid >> 3, u = tid & 7;
    const bool valid = e < n_ex;
    float h2 = b1s[u];
#pragma unroll
    for (int k = 0; k < U1; ++k) h2 += H1s[e * HS + k] * K1s[k * U2 + u];
    h2 = fmaxf(h2, 0.f);
    float dnn = h2 * K2s[u];
    dnn += __shfl_xor(dnn, 1, 64);
    dnn += __shfl_xor(dnn, 2, 64);
    dnn += __shfl_xor(dnn, 4, 64);
    float z = zfm[e] + dnn + a.b2[0];
    float p = sigmoid_acc(z);
    float y = valid ? a.label[ex0 + e] : 0.f;
    const float eps = 1e-7f;
    float pc = fminf(fmaxf(p, eps), 1.f - eps);
    float le = -(y * logf(pc + eps) + (1.f - y) * logf(1.f - pc + eps));
    float inside = (p >= eps && p <= 1.f - eps) ? 1.f : 0.f;
    float dz = -(y / (pc + eps) - (1.f - y) / (1.f - pc + eps)) * inside * p * (1.f - p) / (float)a.B;
    if (!valid) { dz = 0.f; le = 0.f; }
    float dp2 = h2 > 0.f ? dz * K2s[u] : 0.f;
    float dh1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int uu = 0; uu < U2; ++uu) {
      float vv = __shfl(dp2, (lane & ~7) | uu, 64);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) dh1[kq] += vv * K1s[(4 * u + kq) * U2 + uu];
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      int k = 4 * u + kq;
      DP1[e * HS + k] = H1s[e * HS + k] > 0.f ? dh1[kq] : 0.f;
    }
    h2s[e * U2 + u] = h2;
    dp2s[e * U2 + u] = dp2;
    if (u == 0) {
      dzs[e] = dz;
      lss[e] = le;
      if (valid) {
        a.gz[ex0 + e] = dz;
        if (a.prob) a.prob[ex0 + e] = p;
      }
    }
  }
  __syncthreads();

  // ---- small per-workgroup partials (fixed order over the 32 examples).  The single-thread sums are spread over the
  // four waves (no barrier follows: a wave goes on to P5 as soon as its own share is stored)
  {
    float* sm = a.small + (int64_t)blockIdx.x * SMALL;
    const int k = tid >> 3, u = tid & 7;
    float s = 0.f;
#pragma unroll 8
    for (int e = 0; e < EX; ++e) s += H1s[e * HS + k] * dp2s[e * U2 + u];
    sm[tid] = s;                                             // dK1 [32][8]
    if (wave == 1 && lane < U1) {
      float t = 0.f;
#pragma unroll 8
      for (int e = 0; e < EX; ++e) t += DP1[e * HS + lane];
      sm[256 + lane] = t;                                    // db0
    }
    if (wave == 2 && lane < U2) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
      for (int e = 0; e < EX; ++e) { t1 += dp2s[e * U2 + lane]; t2 += h2s[e * U2 + lane] * dzs[e]; }
      sm[288 + lane] = t1;                                   // db1
      sm[296 + lane] = t2;                                   // dK2
    }
    if (wave == 3 && lane == 0) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
      for (int e = 0; e < EX; ++e) { t1 += dzs[e]; t2 += lss[e]; }
      sm[304] = t1;                                          // db2 = dbias
      sm[305] = t2;                                          // sum of per-example BCE terms
    }
  }

  if (a.stop == 5) return;

  // ---- P5: dX = dpre1 . K0^T on the matrix cores, fused with the IndexedSlices values.  N = 16F is cut into tiles of
  // 32 columns = two fields; wave w takes tiles w, w+4, ...; K = 32 units = 16 steps.  A (dpre1, the same for every
  // tile) stays in registers.  D rows are examples, columns 32 consecutive floats of the example's values row: every
  // accumulator register leaves as two full 128-byte lines, no transpose.
  const int n_tiles = (F + 1) / 2;
  {
    float av[16], dzr[16], sr[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) av[s2] = DP1[lo * HS + 2 * s2 + hi];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int e = (r & 3) + 8 * (r >> 2) + 4 * hi;
      dzr[r] = dzs[e];
      sr[r] = Ss[e * E16 + (lo & 15)];
    }
    for (int t = wave; t < n_tiles; t += 4) {
      const float* bp = K0s + (t * 32 + lo) * HS + hi;       // B[k = unit 2s+hi][n = lo] = K0[t*32 + lo][2s + hi]
      const float* xp = XT + (4 * hi) * XS + t * 32 + lo;    // x of (row r, this lane): + ((r&3) + 8*(r>>2)) * XS
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float bv[16], xv[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) bv[s2] = bp[2 * s2];
#pragma unroll
      for (int r = 0; r < 16; ++r) xv[r] = xp[((r & 3) + 8 * (r >> 2)) * XS];     // all loads before the first MFMA
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
      float* vp = a.vals + ((ex0 + 4 * hi) * F + 2 * t) * E16 + lo;
      if (n_ex == EX && 2 * t + 1 < F) {                     // whole tile inside the batch and the fields: no guards
#pragma unroll
        for (int r = 0; r < 16; ++r)
          vp[(int64_t)((r & 3) + 8 * (r >> 2)) * F * E16] = dzr[r] * (sr[r] - xv[r]) + acc[r];
      } else if (2 * t + (lo >> 4) < F) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int e = (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (e < n_ex) vp[(int64_t)((r & 3) + 8 * (r >> 2)) * F * E16] = dzr[r] * (sr[r] - xv[r]) + acc[r];
        }
      }
    }
  }

  if (a.stop == 6) return;

  // ---- P6: per-workgroup dK0 = X^T . dpre1 on the matrix cores.  M = 16F rows of K0 in tiles of 32, wave w takes tiles
  // w, w+4, ...; K = the 32 examples = 16 steps; B (dpre1) stays in registers.  P5 and P6 only read LDS: no barrier.
  {
    float* part = a.dK0part + (int64_t)blockIdx.x * D * U1;
    float bv[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) bv[s2] = DP1[(2 * s2 + hi) * HS + lo];
    for (int t = wave; t < n_tiles; t += 4) {
      // A[i = row lo of the tile][k = example 2s+hi].  For an odd F the upper half of the last tile does not exist:
      // those lanes read the start of the next LDS row instead, which only reaches D rows that are never stored
      const float* ap = XT + hi * XS + t * 32 + lo;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float xv[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) xv[s2] = ap[2 * s2 * XS];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s2], bv[s2], acc, 0, 0, 0);
      float* pp = part + (t * 32 + 4 * hi) * U1 + lo;
      if (t * 32 + 32 <= D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pp[((r & 3) + 8 * (r >> 2)) * U1] = acc[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi < D) pp[((r & 3) + 8 * (r >> 2)) * U1] = acc[r];
      }
    }
  }
}

// fixed-order sum of the per-workgroup partials.  1024 threads = 16 slices x 64 lanes; lane owns 4 consecutive
// outputs (float4), slice q adds workgroups q, q+16, q+32, ... (independent 16-B loads), then the 16 slices are
// added in slice order through LDS.
struct ReduceArgs {
  const float* dK0part; const float* small; int nwg; int D; int64_t B;
  float* dK0; float* dK1; float* db0; float* db1; float* dK2; float* db2; float* dbias; float* loss;
};

__device__ __forceinline__ void reduce_body(const ReduceArgs& r, int bidx) {
  const float* __restrict__ dK0part = r.dK0part;
  const float* __restrict__ small = r.small;
  const int nwg = r.nwg, D = r.D;
  const int64_t B = r.B;
  float* __restrict__ dK0 = r.dK0; float* __restrict__ dK1 = r.dK1; float* __restrict__ db0 = r.db0;
  float* __restrict__ db1 = r.db1; float* __restrict__ dK2 = r.dK2; float* __restrict__ db2 = r.db2;
  float* __restrict__ dbias = r.dbias; float* __restrict__ loss = r.loss;
  __shared__ float4 red[16][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)D * U1;                  // multiple of 4
  const int nb0 = (int)((n0 / 4 + 63) / 64);           // blocks that cover dK0
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t e4 = 0;
  bool is_small = bidx >= nb0;
  if (!is_small) {
    e4 = (int64_t)bidx * 64 + lane;                    // float4 index into dK0
    if (e4 * 4 < n0)
      for (int w = q; w < nwg; w += 16) {
        float4 x = *reinterpret_cast<const float4*>(dK0part + (int64_t)w * n0 + e4 * 4);
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
  } else {
    e4 = (int64_t)(bidx - nb0) * 64 + lane;             // float4 index into the SMALL block
    if (e4 * 4 < SMALL)
      for (int w = q; w < nwg; w += 16) {
        float4 x = *reinterpret_cast<const float4*>(small + (int64_t)w * SMALL + e4 * 4);
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
  }
  red[q][lane] = acc;
  __syncthreads();
  if (q != 0) return;
  float4 s = red[0][lane];
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    float4 x = red[k][lane];
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  if (!is_small) {
    if (e4 * 4 < n0) *reinterpret_cast<float4*>(dK0 + e4 * 4) = s;
    return;
  }
  float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int k = (int)(e4 * 4) + i;
    if (k < 256) dK1[k] = sv[i];
    else if (k < 288) db0[k - 256] = sv[i];
    else if (k < 296) db1[k - 288] = sv[i];
    else if (k < 304) dK2[k - 296] = sv[i];
    else if (k == 304) { db2[0] = sv[i]; dbias[0] = sv[i]; }
    else if (k == 305) loss[0] = sv[i] / (float)B;
  }
}

__global__ __launch_bounds__(1024) void deepfm_reduce_kernel(ReduceArgs r) { reduce_body(r, (int)blockIdx.x); }

size_t fused_lds_bytes(int F) {
  size_t D = (size_t)F * E16;
  size_t f = (size_t)EX * (D + 2) + D * HS + 2 * (size_t)EX * HS + (size_t)EX * E16 + 2 * (size_t)EX * U2 +
             3 * (size_t)EX + U1 * U2 + U1 + 2 * U2 + 4 * (size_t)EX * HS;
  return f * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// per-column sort of the de-duplication plan, three short kernels so that ~200 CUs work on it instead of F:
//   colsort_chunk_kernel   each 1024-id chunk of a column: bitonic network in registers / wave shuffles / 3 LDS
//                          exchange stages on 32-bit words (key << pos_bits | example)
//   colsort_rank_kernel    final position of a word = sum over the column's sorted chunks of #(words < it)
//                          (binary searches in LDS; words are unique, so positions are too) -> scatter
//   colsort_heads_kernel   one workgroup per column: run heads, scan, perm / col_uid / col_seg / col_nu
// ------------------------------------------------------------------------------------------------
constexpr int CHK = 1024;      // ids per sort chunk (256 threads x 4)
constexpr uint32_t PADW = 0xFFFFFFFFu;

struct ColSortArgs {
  int64_t B; int F; int64_t V; int key_bits; int pos_bits; int nch;
  uint32_t* chunks;     // [F][nch][CHK] sorted chunks
  uint32_t* sorted;     // [F][B]
  int32_t* perm;        // [F][B]  sorted position -> example
  int64_t* col_uid;     // [F][B]  unique ids of the column, ascending (first col_nu[f] valid)
  int32_t* col_seg;     // [F][B+1] run starts in the column's sorted order (tail = B)
  int32_t* col_nu;      // [F]
  int* bad;
};

__global__ __launch_bounds__(256) void colsort_chunk_kernel(Cols cols, const int64_t* __restrict__ col_lo, ColSortArgs a) {
  constexpr int NPT = 4;
  __shared__ uint32_t buf[2][CHK];
  const int tid = threadIdx.x;
  const int ch = blockIdx.x, f = blockIdx.y;
  const int64_t lo = col_lo[f];
  uint32_t v[NPT];
  bool bad = false;
#pragma unroll
  for (int r = 0; r < NPT; ++r) {
    int64_t b = (int64_t)ch * CHK + tid * NPT + r;
    uint32_t w = PADW;
    if (b < a.B) {
      int64_t id = cols.p[f][b];
      int64_t key = id - lo;
      if (key < 0 || key >= (int64_t(1) << a.key_bits) || (uint64_t)id >= (uint64_t)a.V) {
        bad = true;
        key = key < 0 ? 0 : (int64_t(1) << a.key_bits) - 1;
      }
      w = ((uint32_t)key << a.pos_bits) | (uint32_t)b;
    }
    v[r] = w;
  }
  if (bad && a.bad) *a.bad = 1;
  int pp = 0;
  for (int k = 2; k <= CHK; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < NPT) {                                   // both elements in this thread's registers
#pragma unroll
        for (int r = 0; r < NPT; ++r) {
          int pr = r ^ j;
          if (pr > r) {
            bool asc = ((tid * NPT + r) & k) == 0;
            uint32_t x = v[r], y = v[pr];
            bool sw = asc ? (x > y) : (x < y);
            v[r] = sw ? y : x;
            v[pr] = sw ? x : y;
          }
        }
      } else {
        int jt = j / NPT;                              // partner thread = tid ^ jt, same register index
        bool lower = (tid & jt) == 0;
        bool asc = ((tid * NPT) & k) == 0;             // k > j >= NPT: the same for the thread's 4 elements
        bool keep_min = (lower == asc);
        if (jt < 64) {                                 // partner in the same wave
#pragma unroll
          for (int r = 0; r < NPT; ++r) {
            uint32_t y = (uint32_t)__shfl_xor((int)v[r], jt, 64);
            v[r] = keep_min ? (v[r] < y ? v[r] : y) : (v[r] > y ? v[r] : y);
          }
        } else {                                       // partner in another wave: exchange through LDS
          uint32_t* mine = buf[pp] + tid * NPT;
#pragma unroll
          for (int r = 0; r < NPT; ++r) mine[r] = v[r];
          __syncthreads();
          const uint32_t* other = buf[pp] + (tid ^ jt) * NPT;
#pragma unroll
          for (int r = 0; r < NPT; ++r) {
            uint32_t y = other[r];
            v[r] = keep_min ? (v[r] < y ? v[r] : y) : (v[r] > y ? v[r] : y);
          }
          pp ^= 1;                                     // ping-pong: one barrier per exchange stage
        }
      }
    }
  }
  uint32_t* out = a.chunks + ((int64_t)f * a.nch + ch) * CHK + tid * NPT;
#pragma unroll
  for (int r = 0; r < NPT; ++r) out[r] = v[r];
}

__global__ __launch_bounds__(256) void colsort_rank_kernel(ColSortArgs a) {
  // the other chunks of the column pass through LDS one at a time (two 4-KB buffers, one barrier per chunk): 8 KB
  // of LDS, so these workgroups fit next to the 134-KB workgroups of the fused kernel on a CU
  __shared__ uint32_t buf[2][CHK];
  const int tid = threadIdx.x;
  const int ch = blockIdx.x, f = blockIdx.y;
  const uint32_t* col = a.chunks + (int64_t)f * a.nch * CHK;
  uint32_t x[4];
  int pos[4];
  {
    uint4 own = reinterpret_cast<const uint4*>(col + (int64_t)ch * CHK)[tid];
    x[0] = own.x; x[1] = own.y; x[2] = own.z; x[3] = own.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) pos[r] = tid * 4 + r;    // words < x in the own (sorted, unique) chunk
  }
  int pp = 0;
  for (int c2 = 0; c2 < a.nch; ++c2) {
    if (c2 == ch) continue;
    reinterpret_cast<uint4*>(buf[pp])[tid] = reinterpret_cast<const uint4*>(col + (int64_t)c2 * CHK)[tid];
    __syncthreads();
    const uint32_t* cc = buf[pp];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int lo = 0;                                        // lower bound: number of words < x
#pragma unroll
      for (int step = CHK / 2; step > 0; step >>= 1)
        if (cc[lo + step - 1] < x[r]) lo += step;
      if (cc[lo] < x[r]) ++lo;                           // CHK is a power of two: one last probe
      pos[r] += lo;
    }
    pp ^= 1;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (x[r] != PADW) a.sorted[(int64_t)f * a.B + pos[r]] = x[r];
}

// One workgroup per 1024 sorted positions of a column (grid nch x F, like the two kernels before it), so that ~200 CUs
// share the work instead of F.  Every workgroup counts the run heads of the WHOLE column itself (B <= 16384 words out
// of L2: heads before its slice = its rank offset, heads in all = col_nu) -- no second pass, no cross-workgroup wait.
__global__ __launch_bounds__(256) void colsort_heads_kernel(const int64_t* __restrict__ col_lo, ColSortArgs a) {
  __shared__ int wred[2][4];
  __shared__ int wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x, f = blockIdx.y;
  const int64_t B = a.B;
  const int64_t lo = col_lo[f];
  const uint32_t* srt = a.sorted + (int64_t)f * B;
  const int pb = a.pos_bits;
  const uint32_t pmask = (1u << pb) - 1u;
  const int64_t s_begin = (int64_t)q * CHK;
  // heads of the column before this slice / in total
  int before = 0, all = 0;
  for (int64_t s = tid; s < B; s += 256) {
    int h = (s == 0 || (srt[s] >> pb) != (srt[s - 1] >> pb)) ? 1 : 0;
    all += h;
    if (s < s_begin) before += h;
  }
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o, 64);
    all += __shfl_xor(all, o, 64);
  }
  if (lane == 0) { wred[0][wave] = before; wred[1][wave] = all; }
  __syncthreads();
  before = wred[0][0] + wred[0][1] + wred[0][2] + wred[0][3];
  all = wred[1][0] + wred[1][1] + wred[1][2] + wred[1][3];
  // own slice: thread owns 4 consecutive sorted positions
  const int64_t s0 = s_begin + (int64_t)tid * 4;
  uint32_t v[4];
  bool hd[4];
  int heads = 0;
  uint32_t prev = (s0 > 0 && s0 - 1 < B) ? srt[s0 - 1] : PADW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t s = s0 + r;
    bool valid = s < B;
    v[r] = valid ? srt[s] : PADW;
    uint32_t pk = (r == 0 ? prev : v[r - 1]) >> pb;
    hd[r] = valid && (s == 0 || (v[r] >> pb) != pk);
    heads += hd[r] ? 1 : 0;
    if (valid) a.perm[(int64_t)f * B + s] = (int32_t)(v[r] & pmask);
  }
  int incl = heads;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wtot[w];
  int rank = before + woff + incl - heads;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t s = s0 + r;
    if (hd[r]) {
      a.col_uid[(int64_t)f * B + rank] = lo + (int64_t)(v[r] >> pb);
      a.col_seg[(int64_t)f * (B + 1) + rank] = (int32_t)s;
      ++rank;
    }
    if (s < B && s + 1 >= all) a.col_seg[(int64_t)f * (B + 1) + s + 1] = (int32_t)B;   // tail [all .. B] = B
  }
  if (q == 0 && tid == 0) a.col_nu[f] = all;
}

// ------------------------------------------------------------------------------------------------
// segment sums of both tables + global compaction.  One lane group (4 lanes x float4) per (column, local run).
// Long runs: the first 16 rows per group directly; what is left of a long run is summed by the whole wave.
// ------------------------------------------------------------------------------------------------
struct ColSegArgs {
  const float4* vals; const float* gz; const int32_t* perm; const int64_t* col_uid; const int32_t* col_seg;
  const int32_t* col_nu; int64_t B; int F; int64_t* uniq_ids; float4* g_embed; float* g_w; int64_t* n_uniq; int packed;
};

__device__ __forceinline__ void colseg_body(const ColSegArgs& k, int bidx) {
  const float4* __restrict__ vals = k.vals;
  const float* __restrict__ gz = k.gz;
  const int32_t* __restrict__ perm = k.perm;
  const int64_t* __restrict__ col_uid = k.col_uid;
  const int32_t* __restrict__ col_seg = k.col_seg;
  const int32_t* __restrict__ col_nu = k.col_nu;
  const int64_t B = k.B;
  const int F = k.F;
  int64_t* __restrict__ uniq_ids = k.uniq_ids;
  float4* __restrict__ g_embed = k.g_embed;
  float* __restrict__ g_w = k.g_w;
  int64_t* __restrict__ n_uniq = k.n_uniq;
  const int packed = k.packed;
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = tid & 3;                       // float4 chunk of the 16-float row
  const int64_t grp = ((int64_t)bidx * blockDim.x + tid) >> 2;     // (f, u_local) = (grp / B, grp % B)
  const int f = (int)(grp / B);
  int u = (int)(grp - (int64_t)f * B);
  if ((B & 63) == 0) {
    // Long runs (a hot id) are worked off by a wave one after the other.  When hot ids are neighbours (the synthetic
    // Zipf draws make the smallest ids of a field the frequent ones) they would all fall to the same wave: deal the
    // runs out so that a wave (16 lane groups) takes four consecutive runs from each quarter of the column -- the
    // first 64 runs then go to 16 different waves, a workgroup still writes 4 KB pieces.  Measured (B=8192, F=26):
    // Zipf(1.05) 113 -> 101 us/step, uniform unchanged.  (Summing a long run with the whole workgroup instead:
    // Zipf 89 us but uniform +5 us -- not taken.)
    const int w = u >> 4, g = u & 15;
    u = (g >> 2) * (int)(B >> 2) + 4 * w + (g & 3);
  }
  const bool in_range = f < F;
  __shared__ int nu_s[REC_MAX_COLS];
  if (tid < F) nu_s[tid] = col_nu[tid];
  __syncthreads();
  int64_t before = 0, total = 0;               // unique ids in earlier columns / in all columns
  for (int q = 0; q < F; ++q) {
    int nq = nu_s[q];
    if (q < f) before += nq;
    total += nq;
  }
  const int nu = in_range ? nu_s[f] : 0;
  const bool live = in_range && u < nu;
  int s0 = 0, s1 = 0;
  if (live) {
    s0 = col_seg[(int64_t)f * (B + 1) + u];
    s1 = col_seg[(int64_t)f * (B + 1) + u + 1];
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float accw = 0.f;
  const int32_t* pf = perm + (int64_t)f * B;
  int s_end = s1 < s0 + 16 ? s1 : s0 + 16;
  for (int s = s0; s < s_end; ++s) {
    int64_t b = pf[s];
    float4 x = vals[(b * F + f) * 4 + c];
    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    accw += gz[b];
  }
  // long runs, one at a time, by all 16 lane groups of the wave (wave-uniform loop)
  unsigned long long longm = __ballot(live && (s1 - s0) > 16 && c == 0);
  while (longm) {
    int src = __ffsll((long long)longm) - 1;   // lane (c == 0) of the group that owns the run
    longm &= longm - 1;
    int rs0 = __shfl(s0, src, 64) + 16, rs1 = __shfl(s1, src, 64);
    int rf = __shfl(f, src, 64);
    const int32_t* rp = perm + (int64_t)rf * B;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
    float pw = 0.f;
#pragma unroll 4
    for (int s = rs0 + (lane >> 2); s < rs1; s += 16) {    // independent loads: several iterations in flight
      int64_t b = rp[s];
      float4 x = vals[(b * F + rf) * 4 + c];
      pa.x += x.x; pa.y += x.y; pa.z += x.z; pa.w += x.w;
      pw += gz[b];
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {         // fixed butterfly over the 16 groups (same chunk lanes)
      pa.x += __shfl_xor(pa.x, o, 64); pa.y += __shfl_xor(pa.y, o, 64);
      pa.z += __shfl_xor(pa.z, o, 64); pa.w += __shfl_xor(pa.w, o, 64);
      pw += __shfl_xor(pw, o, 64);
    }
    if ((lane & ~3) == src) {
      acc.x += pa.x; acc.y += pa.y; acc.z += pa.z; acc.w += pa.w;
      accw += pw;
    }
  }
  if (!in_range) return;
  int64_t dst;
  int64_t idv;
  if (live) {
    dst = before + u;
    idv = col_uid[(int64_t)f * B + u];
  } else {
    // padded tail: slot = total + rank among the non-live groups; id = the smallest id of column 0
    dst = total + ((int64_t)f * B - before) + (u - nu);
    idv = col_uid[0];
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    accw = 0.f;
  }
  if (packed) {                                  // rows of 20 floats: [embed 16 | w | 0 0 0] (one exchange buffer)
    g_embed[dst * 5 + c] = acc;
    if (c == 0) g_embed[dst * 5 + 4] = make_float4(accw, 0.f, 0.f, 0.f);
  } else {
    g_embed[dst * 4 + c] = acc;
    if (c == 0) g_w[dst] = accw;
  }
  if (c == 0) uniq_ids[dst] = idv;
  if (grp == 0 && c == 0) *n_uniq = total;
}

__global__ __launch_bounds__(256) void colseg_sum_kernel(ColSegArgs k) { colseg_body(k, (int)blockIdx.x); }

// reduction of the workgroup partials and the segment sums only depend on the fused kernel, not on each other: one
// launch, the first nb_reduce workgroups (1024 threads) reduce, the others sum segments -- they run side by side
__global__ __launch_bounds__(1024) void deepfm_post_kernel(ReduceArgs r, ColSegArgs k, int nb_reduce) {
  if ((int)blockIdx.x < nb_reduce) reduce_body(r, (int)blockIdx.x);
  else colseg_body(k, (int)blockIdx.x - nb_reduce);
}

}  // namespace

extern "C" size_t rec_deepfm_fused_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  size_t nwg = (size_t)ceil_div64(B, EX);
  return sizeof(float) * nwg * ((size_t)F * E16 * U1 + SMALL) + 256;
}

static int launch_fused(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F, int64_t B,
                        const float* bias, const float* K0, const float* b0, const float* K1, const float* b1,
                        const float* K2, const float* b2, const float* label, float* gz, float* vals, float* prob,
                        float* dK0, float* db0, float* dK1, float* db1, float* dK2, float* db2, float* dbias,
                        float* loss, int* oob_flag, void* workspace, void* stream, const ColSegArgs* seg,
                        bool main_only = false) {
  if (B <= 0 || F <= 0 || V <= 0) return REC_E_ARG;
  if (ld != LD || F > 28 || F > REC_MAX_COLS || V >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;
  if (!table || !cols_host || !bias || !K0 || !b0 || !K1 || !b1 || !K2 || !b2 || !label || !gz || !vals || !dK0 ||
      !db0 || !dK1 || !db1 || !dK2 || !db2 || !dbias || !loss || !workspace)
    return REC_E_ARG;
  if ((reinterpret_cast<uintptr_t>(table) & 15) != 0) return REC_E_UNSUPPORTED;
  size_t lds = fused_lds_bytes(F);
  if (lds > 160 * 1024) return REC_E_UNSUPPORTED;
  Cols cp;
  for (int f = 0; f < F; ++f) {
    if (!cols_host[f]) return REC_E_ARG;
    cp.p[f] = cols_host[f];
  }
  hipStream_t st = as_stream(stream);
  int nwg = (int)ceil_div64(B, EX);
  float* dK0part = (float*)workspace;
  float* small = dK0part + (size_t)nwg * F * E16 * U1;
#ifdef REC_DEBUG_PHASE_STOPS   // profiling builds only (scripts/exp/*_phases.sh): the kernel stops after phase N
  static const int stop = getenv("REC_FUSED_STOP") ? atoi(getenv("REC_FUSED_STOP")) : 0;
#else
  constexpr int stop = 0;
#endif
  FusedArgs a{table, V, bias, K0, b0, K1, b1, K2, b2, label, B, F, gz, vals, prob, dK0part, small, oob_flag, stop};
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(deepfm_fwd_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(deepfm_fwd_bwd_kernel, dim3(nwg), dim3(256), lds, st, cp, a);
  REC_LAUNCH_CHECK();
  if (main_only) return REC_OK;
  int D = F * E16;
  unsigned nb = (unsigned)ceil_div64((int64_t)D * U1 / 4, 64) + (unsigned)ceil_div64(SMALL / 4, 64);
  ReduceArgs r{dK0part, small, nwg, D, B, dK0, dK1, db0, db1, dK2, db2, dbias, loss};
  if (seg) {
    unsigned nbs = (unsigned)ceil_div64(B * F * 4, 1024);
    hipLaunchKernelGGL(deepfm_post_kernel, dim3(nb + nbs), dim3(1024), 0, st, r, *seg, (int)nb);
  } else {
    hipLaunchKernelGGL(deepfm_reduce_kernel, dim3(nb), dim3(1024), 0, st, r);
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_deepfm_fused_fwd_bwd_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host,
                                            int F, int64_t B, const float* bias, const float* K0, const float* b0,
                                            const float* K1, const float* b1, const float* K2, const float* b2,
                                            const float* label, float* gz, float* vals, float* prob, float* dK0,
                                            float* db0, float* dK1, float* db1, float* dK2, float* db2, float* dbias,
                                            float* loss, int* oob_flag, void* workspace, void* stream) {
  return launch_fused(table, ld, V, cols_host, F, B, bias, K0, b0, K1, b1, K2, b2, label, gz, vals, prob, dK0, db0, dK1,
                      db1, dK2, db2, dbias, loss, oob_flag, workspace, stream, nullptr);
}

extern "C" int rec_deepfm_fused_step_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host,
                                         int F, int64_t B, const float* bias, const float* K0, const float* b0,
                                         const float* K1, const float* b1, const float* K2, const float* b2,
                                         const float* label, float* gz, float* vals, float* prob, float* dK0, float* db0,
                                         float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                         int* oob_flag, void* workspace, const int32_t* perm, const int64_t* col_uid,
                                         const int32_t* col_seg, const int32_t* col_nu, int64_t* uniq_ids,
                                         float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, int packed,
                                         void* stream) {
  if (!perm || !col_uid || !col_seg || !col_nu || !uniq_ids || !g_embed_rows || !n_uniq || (!packed && !g_w_rows))
    return REC_E_ARG;
  if ((reinterpret_cast<uintptr_t>(vals) & 15) != 0 || (reinterpret_cast<uintptr_t>(g_embed_rows) & 15) != 0)
    return REC_E_UNSUPPORTED;
  ColSegArgs k{(const float4*)vals, gz, perm, col_uid, col_seg, col_nu, B, F, uniq_ids, (float4*)g_embed_rows,
               packed ? (float*)nullptr : g_w_rows, n_uniq, packed ? 1 : 0};
  return launch_fused(table, ld, V, cols_host, F, B, bias, K0, b0, K1, b1, K2, b2, label, gz, vals, prob, dK0, db0, dK1,
                      db1, dK2, db2, dbias, loss, oob_flag, workspace, stream, &k);
}

// the two halves of rec_deepfm_fused_step_f32 as separate calls: a caller that builds the plan on another stream can
// put its wait between them, so that only the second half depends on the plan
extern "C" int rec_deepfm_fused_main_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host,
                                         int F, int64_t B, const float* bias, const float* K0, const float* b0,
                                         const float* K1, const float* b1, const float* K2, const float* b2,
                                         const float* label, float* gz, float* vals, float* prob, int* oob_flag,
                                         void* workspace, void* stream) {
  float dummy = 0.f;
  float* d = &dummy;                      // gradient outputs are written by the second half only
  return launch_fused(table, ld, V, cols_host, F, B, bias, K0, b0, K1, b1, K2, b2, label, gz, vals, prob, d, d, d, d, d,
                      d, d, d, oob_flag, workspace, stream, nullptr, true);
}

extern "C" int rec_deepfm_fused_post_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                                         float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                         void* workspace, const int32_t* perm, const int64_t* col_uid,
                                         const int32_t* col_seg, const int32_t* col_nu, int64_t* uniq_ids,
                                         float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, int packed,
                                         void* stream) {
  if (B <= 0 || F <= 0 || F > 28) return REC_E_ARG;
  if (!gz || !vals || !dK0 || !db0 || !dK1 || !db1 || !dK2 || !db2 || !dbias || !loss || !workspace || !perm ||
      !col_uid || !col_seg || !col_nu || !uniq_ids || !g_embed_rows || !n_uniq || (!packed && !g_w_rows))
    return REC_E_ARG;
  if ((reinterpret_cast<uintptr_t>(vals) & 15) != 0 || (reinterpret_cast<uintptr_t>(g_embed_rows) & 15) != 0)
    return REC_E_UNSUPPORTED;
  int nwg = (int)ceil_div64(B, EX);
  int D = F * E16;
  float* dK0part = (float*)workspace;
  float* small = dK0part + (size_t)nwg * F * E16 * U1;
  unsigned nb = (unsigned)ceil_div64((int64_t)D * U1 / 4, 64) + (unsigned)ceil_div64(SMALL / 4, 64);
  unsigned nbs = (unsigned)ceil_div64(B * F * 4, 1024);
  ReduceArgs r{dK0part, small, nwg, D, B, dK0, dK1, db0, db1, dK2, db2, dbias, loss};
  ColSegArgs k{(const float4*)vals, gz, perm, col_uid, col_seg, col_nu, B, F, uniq_ids, (float4*)g_embed_rows,
               packed ? (float*)nullptr : g_w_rows, n_uniq, packed ? 1 : 0};
  hipLaunchKernelGGL(deepfm_post_kernel, dim3(nb + nbs), dim3(1024), 0, as_stream(stream), r, k, (int)nb);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" size_t rec_colsort_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  int64_t np = CHK;
  while (np < B) np <<= 1;
  return sizeof(uint32_t) * (size_t)F * ((size_t)np + (size_t)B) + 256;
}

extern "C" int rec_colsort_plan_i64(const int64_t* const* cols_host, int F, int64_t B, int64_t V, const int64_t* col_lo,
                                    int64_t max_key, int32_t* perm, int64_t* col_uid, int32_t* col_seg, int32_t* col_nu,
                                    int* bad_flag, void* workspace, void* stream) {
  if (!cols_host || !col_lo || !perm || !col_uid || !col_seg || !col_nu || !workspace || F <= 0 || B <= 0 || V <= 0 ||
      max_key < 0)
    return REC_E_ARG;
  if (F > REC_MAX_COLS || B > 16384) return REC_E_UNSUPPORTED;
  int pos_bits = 1, key_bits = 1;
  while ((int64_t(1) << pos_bits) < B) ++pos_bits;
  while ((int64_t(1) << key_bits) <= max_key) ++key_bits;
  if (key_bits + pos_bits > 32) return REC_E_UNSUPPORTED;
  // the pad word 0xFFFFFFFF must be larger than every real (key, position) word
  if ((((uint64_t)max_key << pos_bits) | (uint64_t)(B - 1)) >= 0xFFFFFFFFull) return REC_E_UNSUPPORTED;
  Cols cp;
  for (int f = 0; f < F; ++f) {
    if (!cols_host[f]) return REC_E_ARG;
    cp.p[f] = cols_host[f];
  }
  int64_t np = CHK;
  while (np < B) np <<= 1;
  int nch = (int)(np / CHK);
  uint32_t* chunks = (uint32_t*)workspace;
  uint32_t* sorted = chunks + (size_t)F * np;
  ColSortArgs a{B, F, V, key_bits, pos_bits, nch, chunks, sorted, perm, col_uid, col_seg, col_nu, bad_flag};
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(colsort_chunk_kernel, dim3(nch, F), dim3(256), 0, st, cp, col_lo, a);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsort_rank_kernel, dim3(nch, F), dim3(256), 0, st, a);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsort_heads_kernel, dim3(nch, F), dim3(256), 0, st, col_lo, a);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_colseg_sum_f32(const float* vals, const float* gz, const int32_t* perm, const int64_t* col_uid,
                                  const int32_t* col_seg, const int32_t* col_nu, int64_t B, int F, int64_t* uniq_ids,
                                  float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, void* stream) {
  if (!vals || !gz || !perm || !col_uid || !col_seg || !col_nu || !uniq_ids || !g_embed_rows || !g_w_rows || !n_uniq ||
      B <= 0 || F <= 0)
    return REC_E_ARG;
  if ((reinterpret_cast<uintptr_t>(vals) & 15) != 0 || (reinterpret_cast<uintptr_t>(g_embed_rows) & 15) != 0)
    return REC_E_UNSUPPORTED;
  int64_t groups = B * F;
  ColSegArgs k{(const float4*)vals, gz, perm, col_uid, col_seg, col_nu, B, F, uniq_ids, (float4*)g_embed_rows,
               g_w_rows, n_uniq, 0};
  hipLaunchKernelGGL(colseg_sum_kernel, dim3((unsigned)ceil_div64(groups * 4, 256)), dim3(256), 0, as_stream(stream), k);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_colseg_sum_packed_f32(const float* vals, const float* gz, const int32_t* perm, const int64_t* col_uid,
                                         const int32_t* col_seg, const int32_t* col_nu, int64_t B, int F,
                                         int64_t* uniq_ids, float* g_rows, int64_t* n_uniq, void* stream) {
  if (!vals || !gz || !perm || !col_uid || !col_seg || !col_nu || !uniq_ids || !g_rows || !n_uniq || B <= 0 || F <= 0)
    return REC_E_ARG;
  if ((reinterpret_cast<uintptr_t>(vals) & 15) != 0 || (reinterpret_cast<uintptr_t>(g_rows) & 15) != 0)
    return REC_E_UNSUPPORTED;
  int64_t groups = B * F;
  ColSegArgs k{(const float4*)vals, gz, perm, col_uid, col_seg, col_nu, B, F, uniq_ids, (float4*)g_rows,
               (float*)nullptr, n_uniq, 1};
  hipLaunchKernelGGL(colseg_sum_kernel, dim3((unsigned)ceil_div64(groups * 4, 256)), dim3(256), 0, as_stream(stream), k);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
