// De-duplication of an IndexedSlices gradient: sort lookups by id, find the runs, sum the rows of a run.
// Replaces tf.unique + unsorted_segment_sum inside Keras' optimizer (2.FM/ModelManager.py:178-179) and the
// tf.unique of 5.DIN/ModelManager.py:185-186.  Deterministic: a stable radix sort puts the rows of one id
// in ascending position order and every run is summed in that order -- no float atomics.
//
// The key/position radix sort is rocPRIM's device primitive (header-only, compiled here for gfx950); the
// run detection, compaction and the segment sums are hand-written.
//
// Hot ids (a Zipf head, or DIN's padding id that fills half of every behaviour series) give runs of 10^5 rows;
// a run longer than LONG rows is therefore cut into chunks of CH sorted positions that separate workgroups sum
// (fixed slot order + fixed LDS tree), and the owner of the run adds the chunk partials in chunk order.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int TILE = 1024;  // sorted keys per workgroup in the run-detection kernels (256 threads x 4)
constexpr int CH = 256;     // chunk of sorted positions in the long-run path
constexpr int LONG = 256;   // runs longer than this take the chunked path (LONG >= CH: <= 2 long runs per chunk)

struct Layout {
  size_t keys_in, keys_out, pos_in, tile_heads, sort_tmp, sort_tmp_bytes, total;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// rocPRIM's radix sort switches from merge sort to its Onesweep algorithm above 2^20 items, and a captured hipGraph
// that contains Onesweep faults when it is REPLAYED (memory aperture violation, seen with n = 1.27M: the DIN table
// gradient at B = 4096, T = 100).  While the stream is being captured the merge-sort path is therefore kept for every size.
// Block sorts of 4096 keys (512 threads x 8) up to 2^19 keys and of 8192 keys above, instead of the tuned default: two
// or three merge passes (four or six launches) fewer (measured: DeepFM generic step 0.326 -> 0.308 ms at 213k keys with
// 4096, DIN config E 1.588 -> 1.548 ms at 1.27M keys with 8192).
using MergeSortOnly = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 8>,
                                                 rocprim::default_config, (size_t(1) << 40)>;
using MergeSortOnlyLarge = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 16>,
                                                      rocprim::default_config, (size_t(1) << 40)>;
constexpr int64_t MERGE_LARGE_N = int64_t(1) << 19;

Layout make_layout(int64_t n) {
  Layout L;
  size_t off = 0;
  L.keys_in = off; off = align256(off + sizeof(uint32_t) * n);
  L.keys_out = off; off = align256(off + sizeof(uint32_t) * n);
  L.pos_in = off; off = align256(off + sizeof(int32_t) * n);
  L.tile_heads = off; off = align256(off + sizeof(int32_t) * (size_t)(ceil_div64(n, TILE) + 1));
  size_t tmp = 0, tmp2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr,
                            (int32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
  (void)rocprim::radix_sort_pairs<MergeSortOnly>(nullptr, tmp2, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                                 (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
  if (tmp2 > tmp) tmp = tmp2;
  (void)rocprim::radix_sort_pairs<MergeSortOnlyLarge>(nullptr, tmp2, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                                      (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0, 32,
                                                      (hipStream_t)0);
  if (tmp2 > tmp) tmp = tmp2;
  L.sort_tmp = off;
  L.sort_tmp_bytes = tmp;
  off = align256(off + tmp);
  L.total = off;
  return L;
}

__global__ __launch_bounds__(256) void prep_keys_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                        uint32_t* __restrict__ keys, int32_t* __restrict__ pos) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  keys[t] = (uint32_t)ids[t];
  pos[t] = (int32_t)t;
}

__device__ __forceinline__ int block_sum_256(int v, int* sh) {
  // sh: 4 ints.  Returns the block total to every thread.
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void count_heads_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                          int32_t* __restrict__ tile_heads) {
  __shared__ int sh[4];
  int64_t base = (int64_t)blockIdx.x * TILE;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < TILE / 256; ++k) {
    int64_t i = base + threadIdx.x + k * 256;
    if (i < n) cnt += (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
  }
  int tot = block_sum_256(cnt, sh);
  if (threadIdx.x == 0) tile_heads[blockIdx.x] = tot;
}

// One workgroup per tile of sorted keys: global rank of every run head = heads in earlier tiles (summed
// redundantly by each workgroup, a few hundred ints) + exclusive rank inside the tile.
__global__ __launch_bounds__(256) void finalize_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                       const int32_t* __restrict__ tile_heads, int n_tiles,
                                                       int64_t* __restrict__ uniq_ids,
                                                       int32_t* __restrict__ seg_start,
                                                       int64_t* __restrict__ n_uniq) {
  __shared__ int sh[4];
  __shared__ int wave_tot[4];
  int before = 0, all = 0;
  for (int t = threadIdx.x; t < n_tiles; t += 256) {
    int h = tile_heads[t];
    all += h;
    if (t < (int)blockIdx.x) before += h;
  }
  before = block_sum_256(before, sh);
  all = block_sum_256(all, sh);
  // thread owns 4 consecutive sorted positions
  int64_t i0 = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * 4;
  int flag[4];
  uint32_t key[4];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int64_t i = i0 + k;
    flag[k] = 0;
    key[k] = 0;
    if (i < n) {
      key[k] = keys[i];
      flag[k] = (i == 0 || key[k] != keys[i - 1]) ? 1 : 0;
    }
    mine += flag[k];
  }
  // exclusive scan of `mine` over the 256 threads
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int q = 0; q < wave; ++q) woff += wave_tot[q];
  int rank = before + woff + incl - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int64_t i = i0 + k;
    if (i < n && flag[k]) {
      uniq_ids[rank] = (int64_t)key[k];
      seg_start[rank] = (int32_t)i;
      ++rank;
    }
  }
  // tail padding: slots >= n_uniq become empty runs of a valid id (keys[0] = the smallest id)
  uint32_t pad = keys[0];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int64_t i = i0 + k;
    if (i < n && i >= all) uniq_ids[i] = (int64_t)pad;
    if (i < n && i + 1 >= all) seg_start[i + 1] = (int32_t)n;  // covers seg_start[n_uniq .. n]
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *n_uniq = all;
  }
}

// ---- long runs: chunk partials.  One workgroup per chunk of CH sorted positions; threads = (slot, dim) with
// GE = power of two >= E dims per slot.  part[(chunk*2 + which)*E + d]; which = 0: the run that covers the chunk's
// first position, 1: a long run that starts inside the chunk.
template <int GE>
__global__ __launch_bounds__(256) void segsum_chunk_kernel(const float* __restrict__ vals, int E,
                                                           const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ seg_start, int64_t n,
                                                           int32_t row_div, float* __restrict__ part) {
  constexpr int NS = 256 / GE;   // row slots
  __shared__ float red[256];
  const int tid = threadIdx.x, slot = tid / GE, d = tid % GE;
  const int c0 = blockIdx.x * CH;
  const int c1 = (c0 + CH < n) ? c0 + CH : (int)n;
  // largest u with seg_start[u] <= c0 (the padded tail holds n > c0, so it is never chosen): 256-ary search, the
  // workgroup probes 256 evenly spaced candidates per round (3 rounds for n <= 16M instead of 24 dependent loads)
  __shared__ int probe_cnt[4];
  int lo = 0, span = (int)n + 1;                 // candidates [lo, lo + span); seg_start[lo] <= c0 holds throughout
  while (span > 1) {
    int step = (span + 255) >> 8;
    int idx = lo + tid * step;
    bool ok = tid * step < span && seg_start[idx] <= c0;
    unsigned long long m = __ballot(ok);
    __syncthreads();
    if ((tid & 63) == 0) probe_cnt[tid >> 6] = __popcll(m);
    __syncthreads();
    int cnt = probe_cnt[0] + probe_cnt[1] + probe_cnt[2] + probe_cnt[3];     // monotone: the first cnt probes hold
    int nlo = lo + (cnt - 1) * step;
    int nspan = lo + span - nlo;
    span = nspan < step ? nspan : step;
    lo = nlo;
  }
  // Only runs longer than LONG (>= CH) matter here, and at most two of them touch a chunk: the run that covers c0 and
  // the last run that starts inside the chunk.  All runs that start before c1 (<= 256 from `lo` on) are tested at
  // once, one per thread, instead of being walked one after the other.
  __shared__ int long_s0[2], long_s1[2];
  if (tid < 2) long_s0[tid] = -1;
  __syncthreads();
  {
    int u = lo + tid;
    if (u < (int)n) {
      int s0 = seg_start[u];
      if (s0 < c1) {
        int s1 = seg_start[u + 1];
        if (s1 - s0 > LONG) {
          int which = s0 <= c0 ? 0 : 1;
          long_s0[which] = s0;
          long_s1[which] = s1;
        }
      }
    }
  }
  __syncthreads();
  for (int which = 0; which < 2; ++which) {
    int s0 = long_s0[which], s1 = long_s1[which];
    if (s0 < 0) continue;                                   // workgroup-uniform
    int a = s0 > c0 ? s0 : c0, b = s1 < c1 ? s1 : c1;
    float acc = 0.f;
    if (d < E)
      for (int s = a + slot; s < b; s += NS) acc += vals[(int64_t)(perm[s] / row_div) * E + d];
    red[tid] = acc;
    __syncthreads();
#pragma unroll
    for (int k = NS / 2; k > 0; k >>= 1) {   // fixed tree over the slots
      if (slot < k) red[tid] += red[tid + k * GE];
      __syncthreads();
    }
    if (slot == 0 && d < E) part[((int64_t)blockIdx.x * 2 + which) * E + d] = red[d];
    __syncthreads();
  }
}

// One lane group (LPR lanes x float4) per unique id; rows of the run are added in sorted (= position) order.
// A long run (DIN's / SIM's padding id: 600k rows = 2400 chunk partials) is finished by the WHOLE workgroup: its
// 256/lpr lane groups each add a fixed strided subset of the chunk partials (4 loads in flight per lane), and the
// groups meet in a fixed tree through LDS -- one lane group walking 2400 partials alone was a 250-us critical path.
__global__ __launch_bounds__(256) void segsum_vec_kernel(const float4* __restrict__ vals, int lpr,
                                                         const int32_t* __restrict__ perm,
                                                         const int32_t* __restrict__ seg_start, int64_t n,
                                                         int32_t row_div, const float4* __restrict__ part,
                                                         float4* __restrict__ out) {
  __shared__ int long_cnt;
  __shared__ int long_u[256];
  __shared__ float4 red[256];
  if (threadIdx.x == 0) long_cnt = 0;
  __syncthreads();
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n * lpr) {
    int64_t u = t / lpr;
    int c = (int)(t - u * lpr);
    int s0 = seg_start[u], s1 = seg_start[u + 1];
    if (s1 - s0 <= LONG) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = s0; s < s1; ++s) {
        int64_t src = perm[s] / row_div;
        float4 v = vals[src * lpr + c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      out[t] = acc;
    } else if (c == 0 || threadIdx.x == 0) {
      // the first lane of the run inside this workgroup enters it (a run whose lanes straddle two workgroups is
      // finished by both, with identical results)
      long_u[atomicAdd(&long_cnt, 1)] = (int)u;
    }
  }
  __syncthreads();
  const int n_long = long_cnt;                              // workgroup-uniform
  const int NG = 256 / lpr;                                 // lane groups that cooperate
  const int g = threadIdx.x / lpr, c = threadIdx.x - g * lpr;
  for (int q = 0; q < n_long; ++q) {
    const int u = long_u[q];
    const int s0 = seg_start[u], s1 = seg_start[u + 1];
    const int chb = s0 / CH, che = (s1 - 1) / CH;
    float4 a4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < NG) {
      // every chunk after the first starts inside the run: slot 0.  Group g takes chunks chb+1+g, +NG, ...; four
      // independent chains per lane
      int ch = chb + 1 + g;
      for (; ch + 3 * NG <= che; ch += 4 * NG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = part[((int64_t)(ch + j * NG) * 2) * lpr + c];
          a4[j].x += v.x; a4[j].y += v.y; a4[j].z += v.z; a4[j].w += v.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {                         // at most three chunks are left for this group
        int cj = ch + j * NG;
        if (cj <= che) {
          float4 v = part[((int64_t)cj * 2) * lpr + c];
          a4[j].x += v.x; a4[j].y += v.y; a4[j].z += v.z; a4[j].w += v.w;
        }
      }
      a4[0].x += a4[2].x; a4[0].y += a4[2].y; a4[0].z += a4[2].z; a4[0].w += a4[2].w;
      a4[1].x += a4[3].x; a4[1].y += a4[3].y; a4[1].z += a4[3].z; a4[1].w += a4[3].w;
      a4[0].x += a4[1].x; a4[0].y += a4[1].y; a4[0].z += a4[1].z; a4[0].w += a4[1].w;
    }
    __syncthreads();                                        // red[] of the previous run has been consumed
    if (g < NG) red[threadIdx.x] = a4[0];
    __syncthreads();
    int live = NG;                                          // fixed tree over the groups: g += g + ceil(live/2)
    while (live > 1) {
      int half = (live + 1) >> 1;
      if (g + half < live) {
        float4 o = red[(g + half) * lpr + c];
        float4 m = red[g * lpr + c];
        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
        red[g * lpr + c] = m;
      }
      live = half;
      __syncthreads();
    }
    if (g == 0) {
      int which = s0 <= chb * CH ? 0 : 1;
      float4 first = part[((int64_t)chb * 2 + which) * lpr + c];
      float4 m = red[c];
      first.x += m.x; first.y += m.y; first.z += m.z; first.w += m.w;
      out[(int64_t)u * lpr + c] = first;
    }
  }
}

__global__ __launch_bounds__(256) void segsum_scalar_kernel(const float* __restrict__ vals, int E,
                                                            const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ seg_start, int64_t n,
                                                            int32_t row_div, const float* __restrict__ part,
                                                            float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * E) return;
  int64_t u = t / E;
  int d = (int)(t - u * E);
  int s0 = seg_start[u], s1 = seg_start[u + 1];
  float acc = 0.f;
  if (s1 - s0 <= LONG) {
    for (int s = s0; s < s1; ++s) {
      int64_t src = perm[s] / row_div;
      acc += vals[src * E + d];
    }
  } else {
    for (int ch = s0 / CH; ch <= (s1 - 1) / CH; ++ch) {
      int which = s0 <= ch * CH ? 0 : 1;
      acc += part[((int64_t)ch * 2 + which) * E + d];
    }
  }
  out[t] = acc;
}

// ---- union of P ascending, duplicate-free id lists (what P requesters that de-duplicated their own batch send to
// the owner of a table shard).  Final position of an element = its index in its own list + for every other list the
// number of ids that sort before it (ties: the lower list first) -- binary searches, no sort.
constexpr int MAX_LISTS = 64;
__global__ __launch_bounds__(256) void merge_rank_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                         const int64_t* __restrict__ list_counts, int n_lists,
                                                         uint32_t* __restrict__ keys_out, int32_t* __restrict__ perm) {
  __shared__ int64_t off[MAX_LISTS + 1];
  if (threadIdx.x == 0) {
    int64_t a = 0;
    for (int q = 0; q < n_lists; ++q) {
      off[q] = a;
      int64_t c = list_counts[q];
      a += c < 0 ? 0 : c;
    }
    off[n_lists] = a;
  }
  __syncthreads();
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n || e >= off[n_lists]) return;
  int s = 0;
  while (s + 1 < n_lists && off[s + 1] <= e) ++s;
  const int64_t id = ids[e];
  int64_t rank = e - off[s];
  for (int q = 0; q < n_lists; ++q) {
    if (q == s) continue;
    int64_t lo = off[q], hi = off[q + 1];          // first position whose id is > id (q < s) or >= id (q > s)
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      int64_t v = ids[mid];
      bool before = q < s ? v <= id : v < id;
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo - off[q];
  }
  keys_out[rank] = (uint32_t)id;
  perm[rank] = (int32_t)e;
}

}  // namespace

extern "C" size_t rec_dedup_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return make_layout(n).total;
}

extern "C" int rec_dedup_plan_i64(const int64_t* ids, int64_t n, int64_t V, int64_t* uniq_ids,
                                  int32_t* seg_start, int32_t* perm, int64_t* n_uniq, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!ids || !uniq_ids || !seg_start || !perm || !n_uniq || !workspace || n < 0 || V <= 0) return REC_E_ARG;
  if (n >= (int64_t(1) << 31) || V > (int64_t(1) << 32)) return REC_E_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(n_uniq, 0, sizeof(int64_t), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(seg_start, 0, sizeof(int32_t), st);
    return (int)e;
  }
  Layout L = make_layout(n);
  if (workspace_bytes < L.total) return REC_E_WORKSPACE;
  char* ws = (char*)workspace;
  uint32_t* keys_in = (uint32_t*)(ws + L.keys_in);
  uint32_t* keys_out = (uint32_t*)(ws + L.keys_out);
  int32_t* pos_in = (int32_t*)(ws + L.pos_in);
  int32_t* tile_heads = (int32_t*)(ws + L.tile_heads);
  hipLaunchKernelGGL(prep_keys_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, ids, n, keys_in,
                     pos_in);
  REC_LAUNCH_CHECK();
  unsigned end_bit = 1;
  while (end_bit < 32 && (int64_t(1) << end_bit) < V) ++end_bit;
  size_t tmp = L.sort_tmp_bytes;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(st, &cap);
  if (e != hipSuccess) return (int)e;
  if (cap == hipStreamCaptureStatusActive && n >= MERGE_LARGE_N)
    e = rocprim::radix_sort_pairs<MergeSortOnlyLarge>(ws + L.sort_tmp, tmp, keys_in, keys_out, pos_in, perm, (size_t)n,
                                                      0u, end_bit, st);
  else if (cap == hipStreamCaptureStatusActive)
    e = rocprim::radix_sort_pairs<MergeSortOnly>(ws + L.sort_tmp, tmp, keys_in, keys_out, pos_in, perm, (size_t)n, 0u,
                                                 end_bit, st);
  else
    e = rocprim::radix_sort_pairs(ws + L.sort_tmp, tmp, keys_in, keys_out, pos_in, perm, (size_t)n, 0u, end_bit, st);
  if (e != hipSuccess) return (int)e;
  int n_tiles = (int)ceil_div64(n, TILE);
  hipLaunchKernelGGL(count_heads_kernel, dim3(n_tiles), dim3(256), 0, st, keys_out, n, tile_heads);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_kernel, dim3(n_tiles), dim3(256), 0, st, keys_out, n, tile_heads, n_tiles,
                     uniq_ids, seg_start, n_uniq);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" size_t rec_segment_sum_workspace_bytes(int64_t n, int E) {
  if (n <= 0 || E <= 0) return 256;
  return sizeof(float) * (size_t)ceil_div64(n, CH) * 2 * (size_t)E + 256;
}

extern "C" int rec_segment_sum_f32(const float* vals, int E, const int32_t* perm, const int32_t* seg_start, int64_t n,
                                   int32_t row_div, float* out, float* workspace, void* stream) {
  if (E <= 0 || n < 0 || row_div <= 0) return REC_E_ARG;
  if (n == 0) return REC_OK;
  if (!vals || !perm || !seg_start || !out || !workspace) return REC_E_ARG;
  if (E > 256) return REC_E_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  unsigned n_chunks = (unsigned)ceil_div64(n, CH);
#define CHUNK(GE) hipLaunchKernelGGL(segsum_chunk_kernel<GE>, dim3(n_chunks), dim3(256), 0, st, vals, E, perm, \
                                     seg_start, n, row_div, workspace)
  if (E <= 1) CHUNK(1);
  else if (E <= 2) CHUNK(2);
  else if (E <= 4) CHUNK(4);
  else if (E <= 8) CHUNK(8);
  else if (E <= 16) CHUNK(16);
  else if (E <= 32) CHUNK(32);
  else if (E <= 64) CHUNK(64);
  else if (E <= 128) CHUNK(128);
  else CHUNK(256);
#undef CHUNK
  REC_LAUNCH_CHECK();
  bool vec = E % 4 == 0 && (reinterpret_cast<uintptr_t>(vals) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
  if (vec) {
    int lpr = E / 4;
    hipLaunchKernelGGL(segsum_vec_kernel, dim3((unsigned)ceil_div64(n * lpr, 256)), dim3(256), 0, st,
                       (const float4*)vals, lpr, perm, seg_start, n, row_div, (const float4*)workspace, (float4*)out);
  } else {
    hipLaunchKernelGGL(segsum_scalar_kernel, dim3((unsigned)ceil_div64(n * E, 256)), dim3(256), 0, st, vals, E, perm,
                       seg_start, n, row_div, workspace, out);
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_dedup_plan_sorted_lists_i64(const int64_t* ids, int64_t n, const int64_t* list_counts, int n_lists,
                                               int64_t V, int64_t* uniq_ids, int32_t* seg_start, int32_t* perm,
                                               int64_t* n_uniq, void* workspace, size_t workspace_bytes, void* stream) {
  if (!ids || !list_counts || !uniq_ids || !seg_start || !perm || !n_uniq || !workspace || n < 0 || V <= 0 ||
      n_lists <= 0)
    return REC_E_ARG;
  if (n >= (int64_t(1) << 31) || V > (int64_t(1) << 32) || n_lists > MAX_LISTS) return REC_E_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(n_uniq, 0, sizeof(int64_t), st);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemsetAsync(seg_start, 0, sizeof(int32_t), st);
  }
  Layout L = make_layout(n);
  if (workspace_bytes < L.total) return REC_E_WORKSPACE;
  char* ws = (char*)workspace;
  uint32_t* keys_out = (uint32_t*)(ws + L.keys_out);
  int32_t* tile_heads = (int32_t*)(ws + L.tile_heads);
  hipLaunchKernelGGL(merge_rank_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, ids, n, list_counts,
                     n_lists, keys_out, perm);
  REC_LAUNCH_CHECK();
  int n_tiles = (int)ceil_div64(n, TILE);
  hipLaunchKernelGGL(count_heads_kernel, dim3(n_tiles), dim3(256), 0, st, keys_out, n, tile_heads);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_kernel, dim3(n_tiles), dim3(256), 0, st, keys_out, n, tile_heads, n_tiles, uniq_ids,
                     seg_start, n_uniq);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
