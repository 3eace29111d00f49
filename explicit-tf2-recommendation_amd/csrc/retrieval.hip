// Retrieval step that follows the DSSM towers: exact k nearest items of every user vector.
// The reference L2-normalises the item vectors, builds a scikit-learn BallTree over them and queries it with the raw
// user vector (2.FM/OfflineLoader.py:129-162, 2.FM/OnlineServer.py:53-75); BallTree.query is exact, so the answer is
// the brute-force one -- the k smallest ||u - i_hat||_2, ascending -- and that is what runs here:
//
//   l2_normalize_rows_kernel   i_hat = i / ||i||                                    (once per item table)
//   topk_scan_kernel<K, D, QW> one ITEM per lane (its vector in registers, 64 consecutive rows per wave step: coalesced),
//                              the QW queries of the wave broadcast from LDS one after the other: 3*D VALU operations
//                              per (query, item) pair and wave-uniform control flow -- the threshold of query q (its
//                              current K-th distance^2) lives in lane q of one register, `ballot(d2 < thr)` is zero
//                              for almost every step, and the rare survivor is put into the query's sorted list in
//                              LDS by the whole wave (position = popcount of a ballot, shift by one shuffle).  Items
//                              are visited in index order and an item only displaces a strictly larger distance, so
//                              equal distances keep the lower index.  Every wave scans its own sub-slab of items.
//   topk_merge_kernel<K>       per query: the (ascending) lists of all sub-slabs merged in index order, sqrt, int64
//
// fp32 throughout; the fp32 matrix cores have the same peak as the vector units on gfx950 and a pairwise distance needs
// no GEMM.  HBM: the item table is read once per block of QW queries (QW * K = 1024: 8 KB of lists per wave in LDS).
#include "common.h"
#include <float.h>
#include <math.h>

namespace {

constexpr int STEP = 128;      // items per wave pass (two steps of one item per lane)

__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, int64_t n, int d, int64_t ld_in,
                                                                float* __restrict__ y, int64_t ld_out) {
  int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  float s = 0.f;
  for (int j = 0; j < d; ++j) {
    float v = x[r * ld_in + j];
    s += v * v;
  }
  float inv = 1.0f / sqrtf(s);
  for (int j = 0; j < d; ++j) y[r * ld_out + j] = x[r * ld_in + j] * inv;
}

template <int K, int D, int QW>
__global__ __launch_bounds__(256) void topk_scan_kernel(const float* __restrict__ queries, int64_t nq, int d, int64_t ldq,
                                                        const float* __restrict__ items, int64_t n, int64_t ldi,
                                                        int64_t sub, float* __restrict__ cand_v,
                                                        int* __restrict__ cand_i) {
  // LDS: queries [QW][D] | per wave: list values [QW][K], list indices [QW][K]
  extern __shared__ float lds[];
  float* qs = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* lv = qs + QW * D + (size_t)wave * QW * K * 2;
  int* li = reinterpret_cast<int*>(lv + QW * K);
  const int64_t q0 = (int64_t)blockIdx.y * QW;
  for (int e = tid; e < QW * D; e += 256) {
    int qq = e / D, j = e - qq * D;
    qs[e] = (q0 + qq < nq && j < d) ? queries[(q0 + qq) * ldq + j] : 0.f;
  }
  for (int e = lane; e < QW * K; e += 64) { lv[e] = FLT_MAX; li[e] = -1; }
  __syncthreads();
  const int64_t ss = ((int64_t)blockIdx.x * 4 + wave) * sub;         // this wave's sub-slab [ss, se)
  const int64_t se = ss + sub < n ? ss + sub : n;
  float thr = FLT_MAX;                                               // lane q: K-th best distance^2 of query q so far
  // the rare survivor of the threshold test: the whole wave puts (v, idx) into query qq's ascending list.  Everything
  // after the two LDS reads stays in the vector unit: v_readlane for the broadcasts, a DPP wave shift for "take your
  // left neighbour's element" (an LDS-routed shuffle would add a round trip each)
  auto insert = [&](int qq, float v, int idx) {
    float* row = lv + qq * K;
    int* rowi = li + qq * K;
    float ev = lane < K ? row[lane] : FLT_MAX;                       // lanes 0..K-1 hold the list; K <= 64
    int ei = lane < K ? rowi[lane] : -1;
    const float last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ev), K - 1));
    if (!(v < last)) return;                                         // an earlier survivor of this step raised the bar
    const int pos = __popcll(__ballot(lane < K && !(v < ev)));       // elements <= v stay in front: ties keep the earlier
    const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ev), 0x138, 0xf, 0xf, false));
    const int pi = __builtin_amdgcn_update_dpp(0, ei, 0x138, 0xf, 0xf, false);                 // wave_shr:1
    const float nv = lane < pos ? ev : (lane == pos ? v : pv);
    if (lane < K) {
      row[lane] = nv;
      rowi[lane] = lane < pos ? ei : (lane == pos ? idx : pi);
    }
    const float nthr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nv), K - 1));
    if (lane == qq) thr = nthr;
  };
  // two steps of 64 items per pass (one item per lane and step, vectors in registers), queries four at a time: their
  // vectors are the same for every lane (broadcast LDS reads, issued together and used for 128 pairs each); the
  // threshold of query q sits in lane q of `thr` and is read with v_readlane (q is wave-uniform)
  for (int64_t t0 = ss; t0 < se; t0 += 128) {
    bool have[2];
    float x[2][D];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int64_t it = t0 + 64 * sb + lane;
      have[sb] = it < se;
#pragma unroll
      for (int j = 0; j < D; ++j) x[sb][j] = (have[sb] && j < d) ? items[it * ldi + j] : 0.f;
    }
#pragma unroll 1
    for (int qq0 = 0; qq0 < QW; qq0 += 4) {
      float acc[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* qv = qs + (qq0 + u) * D;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const float qj = qv[j];
          const float d0 = qj - x[0][j], d1 = qj - x[1][j];
          a0 += d0 * d0;
          a1 += d1 * d1;
        }
        acc[u][0] = a0;
        acc[u][1] = a1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qq = qq0 + u;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {                             // step 0's items come before step 1's: index order
          const float tq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(thr), qq));
          unsigned long long m = __ballot(have[sb] && acc[u][sb] < tq);
          while (m) {                                                // rare; survivors in lane (= index) order
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            insert(qq, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc[u][sb]), src)),
                   (int)(t0 - ss) + 64 * sb + src);
          }
        }
      }
    }
  }
  // lists out: [sub-slab][query][K]
  const int64_t sl = (int64_t)blockIdx.x * 4 + wave;
  for (int e = lane; e < QW * K; e += 64) {
    int qq = e / K;
    if (q0 + qq < nq) {
      cand_v[(sl * nq + q0 + qq) * K + (e - qq * K)] = lv[e];
      cand_i[(sl * nq + q0 + qq) * K + (e - qq * K)] = li[e];
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                         int64_t nq, int nslab, int64_t slab, int k,
                                                         int64_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  float bv[K];
  int64_t gi[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bv[j] = FLT_MAX; bi[j] = -1; }
  // bi holds (slab << 8 | position) would overflow for big slabs: merge on a (slab, local index) pair instead
  int bs[K];
#pragma unroll
  for (int j = 0; j < K; ++j) bs[j] = 0;
  for (int s = 0; s < nslab; ++s) {                        // slab order = index order: ties keep the lower index
    const float* v = cand_v + ((int64_t)s * nq + q) * K;
    const int* ix = cand_i + ((int64_t)s * nq + q) * K;
    for (int j = 0; j < K; ++j) {
      float x = v[j];
      if (!(x < bv[K - 1])) break;                         // the list is ascending: nothing further can enter
      int li = ix[j], ls = s;
#pragma unroll
      for (int m = 0; m < K; ++m) {
        bool sw = x < bv[m];
        float tv = bv[m];
        int ti = bi[m], ts = bs[m];
        bv[m] = sw ? x : tv;
        bi[m] = sw ? li : ti;
        bs[m] = sw ? ls : ts;
        x = sw ? tv : x;
        li = sw ? ti : li;
        ls = sw ? ts : ls;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) gi[j] = bi[j] < 0 ? -1 : (int64_t)bs[j] * slab + bi[j];
  for (int j = 0; j < k; ++j) {
    // static register indices: pick element j with a select chain
    float v = 0.f;
    int64_t id = -1;
#pragma unroll
    for (int m = 0; m < K; ++m) {
      v = m == j ? bv[m] : v;
      id = m == j ? gi[m] : id;
    }
    out_idx[q * k + j] = id;
    out_dist[q * k + j] = id < 0 ? INFINITY : sqrtf(v);
  }
}

inline int qw_of(int K) { return K <= 16 ? 64 : (K <= 32 ? 32 : 16); }   // queries per wave: 8 KB of lists per wave

// workgroups along the item axis: enough to fill the chip, at least 16 steps per wave
int pick_slabs(int64_t n, int64_t nq, int K) {
  int64_t qblocks = ceil_div64(nq, qw_of(K));
  int64_t want = ceil_div64(1024, qblocks);
  int64_t by_size = ceil_div64(n, 4 * 16 * STEP);
  int64_t s = want < by_size ? want : by_size;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

}  // namespace

extern "C" int rec_l2_normalize_rows_f32(const float* x, int64_t n, int d, int64_t ld_in, float* y, int64_t ld_out,
                                         void* stream) {
  if (n < 0 || d <= 0 || ld_in < d || ld_out < d) return REC_E_ARG;
  if (n == 0) return REC_OK;
  if (!x || !y) return REC_E_ARG;
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), x, n, d,
                     ld_in, y, ld_out);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" size_t rec_topk_l2_workspace_bytes(int64_t nq, int64_t n, int k) {
  if (nq <= 0 || n <= 0 || k <= 0) return 256;
  int K = k <= 16 ? 16 : (k <= 32 ? 32 : 64);
  return (size_t)pick_slabs(n, nq, K) * 4 * (size_t)nq * K * (sizeof(float) + sizeof(int)) + 256;
}

template <int K, int D, int QW>
static int launch_scan(const float* queries, int64_t nq, int d, int64_t ldq, const float* items, int64_t n, int64_t ldi,
                       int nslab, int64_t sub, float* cand_v, int* cand_i, hipStream_t st) {
  size_t lds = sizeof(float) * ((size_t)QW * D + 4 * (size_t)QW * K * 2);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_scan_kernel<K, D, QW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)nslab, (unsigned)ceil_div64(nq, QW));
  hipLaunchKernelGGL((topk_scan_kernel<K, D, QW>), grid, dim3(256), lds, st, queries, nq, d, ldq, items, n, ldi, sub,
                     cand_v, cand_i);
  return REC_OK;
}

template <int K, int QW>
static int launch_scan_d(const float* queries, int64_t nq, int d, int64_t ldq, const float* items, int64_t n, int64_t ldi,
                         int nslab, int64_t sub, float* cand_v, int* cand_i, hipStream_t st) {
  if (d <= 8) return launch_scan<K, 8, QW>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
  if (d <= 16) return launch_scan<K, 16, QW>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
  if (d <= 32) return launch_scan<K, 32, QW>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
  return launch_scan<K, 64, QW>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
}

extern "C" int rec_topk_l2_f32(const float* queries, int64_t nq, int d, int64_t ldq, const float* items, int64_t n,
                               int64_t ldi, int k, int64_t* out_idx, float* out_dist, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (nq < 0 || n <= 0 || d <= 0 || k <= 0 || ldq < d || ldi < d) return REC_E_ARG;
  if (k > 64 || d > 64 || n >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;
  if (nq == 0) return REC_OK;
  if (!queries || !items || !out_idx || !out_dist || !workspace) return REC_E_ARG;
  if (workspace_bytes < rec_topk_l2_workspace_bytes(nq, n, k)) return REC_E_WORKSPACE;
  const int K = k <= 16 ? 16 : (k <= 32 ? 32 : 64);
  const int nslab = pick_slabs(n, nq, K);
  const int nsub = nslab * 4;                                        // one sub-slab per wave
  const int64_t sub = ceil_div64(ceil_div64(n, nsub), STEP) * STEP;
  float* cand_v = (float*)workspace;
  int* cand_i = (int*)(cand_v + (size_t)nsub * nq * K);
  hipStream_t st = as_stream(stream);
  int rc;
  if (K == 16) rc = launch_scan_d<16, 64>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
  else if (K == 32) rc = launch_scan_d<32, 32>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
  else rc = launch_scan_d<64, 16>(queries, nq, d, ldq, items, n, ldi, nslab, sub, cand_v, cand_i, st);
  if (rc != REC_OK) return rc;
  REC_LAUNCH_CHECK();
  unsigned mg = (unsigned)ceil_div64(nq, 256);
  if (K == 16)
    hipLaunchKernelGGL(topk_merge_kernel<16>, dim3(mg), dim3(256), 0, st, cand_v, cand_i, nq, nsub, sub, k, out_idx,
                       out_dist);
  else if (K == 32)
    hipLaunchKernelGGL(topk_merge_kernel<32>, dim3(mg), dim3(256), 0, st, cand_v, cand_i, nq, nsub, sub, k, out_idx,
                       out_dist);
  else
    hipLaunchKernelGGL(topk_merge_kernel<64>, dim3(mg), dim3(256), 0, st, cand_v, cand_i, nq, nsub, sub, k, out_idx,
                       out_dist);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
