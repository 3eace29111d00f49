// Row-wise block sharding of an embedding table across the GPUs of a node (the
// reference is single-device, so this has no reference counterpart).  owner = id / rows_per_shard.
// rec_shard_bucketize_i64 is the stable partition that precedes the RCCL all-to-all of ids (C1);
// rec_permute_rows_f32 undoes it on the rows that come back (C2) and re-applies it to the row
// gradients that go out (C3).  Integer work: results are bit-exact and independent of timing.
#include "common.h"

namespace {

constexpr int TILE = 1024;
constexpr int MAX_SHARD = 64;

__global__ __launch_bounds__(256) void shard_count_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                          int64_t rows_per_shard, int n_shard,
                                                          int32_t* __restrict__ tile_counts, int* oob) {
  __shared__ int cnt[MAX_SHARD];
  if (threadIdx.x < MAX_SHARD) cnt[threadIdx.x] = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * TILE;
#pragma unroll
  for (int k = 0; k < TILE / 256; ++k) {
    int64_t i = base + threadIdx.x + k * 256;
    if (i < n) {
      int64_t id = ids[i];
      int64_t o = id >= 0 ? id / rows_per_shard : -1;
      if (o < 0 || o >= n_shard) {
        if (oob) *oob = 1;
      } else {
        atomicAdd(&cnt[(int)o], 1);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < n_shard) tile_counts[(int64_t)blockIdx.x * n_shard + threadIdx.x] = cnt[threadIdx.x];
}

// single workgroup of 1024 threads: column-wise exclusive scan of tile_counts[n_tiles][n_shard] + shard bases.
// Thread t owns tile (pass*1024 + t); one block scan per shard and pass, carries in LDS.
__global__ __launch_bounds__(1024) void shard_scan_kernel(int32_t* __restrict__ tile_counts, int n_tiles, int n_shard,
                                                          int64_t* __restrict__ send_counts) {
  __shared__ int64_t tot[MAX_SHARD];
  __shared__ int wsum[16];
  __shared__ int64_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // totals per shard
  for (int s = 0; s < n_shard; ++s) {
    int v = 0;
    for (int t = tid; t < n_tiles; t += 1024) v += tile_counts[(int64_t)t * n_shard + s];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (tid == 0) {
      int64_t a = 0;
      for (int q = 0; q < 16; ++q) a += wsum[q];
      tot[s] = a;
      send_counts[s] = a;
    }
  }
  __syncthreads();
  for (int s = 0; s < n_shard; ++s) {
    if (tid == 0) {
      int64_t base = 0;
      for (int q = 0; q < s; ++q) base += tot[q];
      carry = base;
    }
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += 1024) {
      int t = t0 + tid;
      int c = t < n_tiles ? tile_counts[(int64_t)t * n_shard + s] : 0;
      int incl = c;
      for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int woff = 0, all = 0;
      for (int q = 0; q < 16; ++q) {
        if (q < wave) woff += wsum[q];
        all += wsum[q];
      }
      int64_t base = carry;
      if (t < n_tiles) tile_counts[(int64_t)t * n_shard + s] = (int32_t)(base + woff + incl - c);   // now an offset
      __syncthreads();
      if (tid == 0) carry = base + all;
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void shard_scatter_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                            int64_t rows_per_shard, int n_shard,
                                                            const int32_t* __restrict__ tile_off,
                                                            int64_t* __restrict__ perm, int64_t* __restrict__ local_ids) {
  __shared__ int wave_tot[4];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread owns 4 CONSECUTIVE positions so that ranks follow position order
  int64_t i0 = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * 4;
  int own[4];
  int64_t idv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int64_t i = i0 + k;
    own[k] = -1;
    idv[k] = 0;
    if (i < n) {
      idv[k] = ids[i];
      int64_t o = idv[k] >= 0 ? idv[k] / rows_per_shard : -1;
      own[k] = (o >= 0 && o < n_shard) ? (int)o : -1;
    }
  }
  for (int s = 0; s < n_shard; ++s) {
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) mine += (own[k] == s);
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
    int64_t dst = (int64_t)tile_off[(int64_t)blockIdx.x * n_shard + s] + woff + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (own[k] == s) {
        perm[dst] = i0 + k;
        local_ids[dst] = idv[k] - (int64_t)s * rows_per_shard;
        ++dst;
      }
    }
  }
}

__global__ __launch_bounds__(256) void permute_rows_kernel(const float* __restrict__ in, const int64_t* __restrict__ perm,
                                                           int64_t n, int E, int scatter, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * E) return;
  int64_t i = t / E;
  int d = (int)(t - i * E);
  int64_t p = perm[i];
  if ((uint64_t)p >= (uint64_t)n) return;
  if (scatter) out[p * E + d] = in[t];
  else out[t] = in[p * E + d];
}

// ---- de-duplicate-first exchange plan.  Input: the per-column sort plan of a batch (rec_colsort_plan_i64; columns own
// ascending id ranges, so the concatenation of the columns' unique ids is globally ascending and, with the block
// partition, already grouped by owner).  One thread per (column, sorted position):
//   uidx[f][example]   = compact index of the lookup's id in that ascending list   (what the fused kernel gathers by)
//   uid_local[compact] = id - owner*rows_per_shard                                   (what is sent to the owner)
//   send_counts[owner] = unique ids of this batch owned by `owner`                  (integer atomics: exact)
__global__ __launch_bounds__(256) void shard_map_kernel(const int32_t* __restrict__ perm, const int64_t* __restrict__ col_uid,
                                                        const int32_t* __restrict__ col_seg, const int32_t* __restrict__ col_nu,
                                                        int64_t B, int F, int64_t rows_per_shard, int n_shard,
                                                        int64_t* __restrict__ uid_local, int64_t* __restrict__ uidx,
                                                        unsigned long long* __restrict__ send_counts,
                                                        int64_t* __restrict__ n_uniq, int* __restrict__ oob) {
  __shared__ int nu_s[REC_MAX_COLS];
  __shared__ int cnt[MAX_SHARD];
  const int tid = threadIdx.x;
  if (tid < F) nu_s[tid] = col_nu[tid];
  if (tid < MAX_SHARD) cnt[tid] = 0;
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + tid;
  const int f = (int)(t / B);
  const int s = (int)(t - (int64_t)f * B);
  if (f < F) {
    int64_t before = 0, total = 0;
    for (int q = 0; q < F; ++q) {
      if (q < f) before += nu_s[q];
      total += nu_s[q];
    }
    const int nu = nu_s[f];
    const int32_t* seg = col_seg + (int64_t)f * (B + 1);
    int lo = 0, hi = nu - 1;                       // largest u with seg[u] <= s   (seg[0] = 0)
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (seg[mid] <= s) lo = mid; else hi = mid - 1;
    }
    uidx[(int64_t)f * B + perm[(int64_t)f * B + s]] = before + lo;
    if (seg[lo] == s) {                            // run head: one thread per unique id
      int64_t id = col_uid[(int64_t)f * B + lo];
      int64_t o = id >= 0 ? id / rows_per_shard : -1;
      if (o < 0 || o >= n_shard) {
        if (oob) *oob = 1;
        o = 0;
      } else {
        atomicAdd(&cnt[(int)o], 1);
      }
      uid_local[before + lo] = id - o * rows_per_shard;
    }
    if (t == 0) *n_uniq = total;
  }
  __syncthreads();
  if (tid < n_shard && cnt[tid]) atomicAdd(&send_counts[tid], (unsigned long long)cnt[tid]);
}


// ---- the same plan for FIXED-CAPACITY exchanges: every owner gets a slab of `cap` slots, whatever the batch holds, so
// that the three payload exchanges of a step have constant sizes -- no count exchange, no host read, and the step can be
// captured in a hipGraph.  cap >= the unique ids one batch can hold for one owner: the sum over the fields that intersect
// the owner's block of min(B, overlap) (engine.ShardedDeepFMStep computes it from the field layout).
//   msg[o][0] = unique ids of this batch owned by o, msg[o][1] = 0, msg[o][2 + j] = j-th of them as owner-local id
//   uidx[f][example] = o * cap + j : the row of the lookup inside the [owners, cap] buffer the rows come back in
//   slot_map[rank in the batch's ascending unique list] = o * cap + j : where the post launch puts the gradient row
// The first unique id of owner o has rank start[o] = sum over the columns of (ids below o * rows_per_shard): one binary
// search per (owner, column) per workgroup, in LDS.
constexpr int MSG_HDR = 2;
__global__ __launch_bounds__(256) void shard_map_fixed_kernel(
    const int32_t* __restrict__ perm, const int64_t* __restrict__ col_uid, const int32_t* __restrict__ col_seg,
    const int32_t* __restrict__ col_nu, int64_t B, int F, int64_t rows_per_shard, int n_shard, int64_t cap,
    int64_t* __restrict__ msg, int64_t* __restrict__ uidx, int32_t* __restrict__ slot_map, int64_t* __restrict__ n_uniq,
    int* __restrict__ oob) {
  __shared__ int nu_s[REC_MAX_COLS];
  __shared__ int start_s[MAX_SHARD + 1];
  const int tid = threadIdx.x;
  if (tid < F) nu_s[tid] = col_nu[tid];
  if (tid <= n_shard) start_s[tid] = 0;
  __syncthreads();
  for (int j = tid; j < (n_shard - 1) * F; j += 256) {
    const int o = 1 + j / F, q = j - (o - 1) * F;
    const int64_t bound = (int64_t)o * rows_per_shard;
    const int64_t* cu = col_uid + (int64_t)q * B;
    int lo = 0, hi = nu_s[q];
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cu[mid] < bound) lo = mid + 1; else hi = mid;
    }
    if (lo) atomicAdd(&start_s[o], lo);
  }
  int64_t total = 0;
  for (int q = 0; q < F; ++q) total += nu_s[q];
  __syncthreads();
  if (tid == 0) start_s[n_shard] = (int)total;
  __syncthreads();
  const int64_t stride = cap + MSG_HDR;
  const int64_t t = (int64_t)blockIdx.x * 256 + tid;
  if (blockIdx.x == 0 && tid < n_shard) {
    int cnt = start_s[tid + 1] - start_s[tid];
    if (cnt > cap) {
      if (oob) *oob = 1;                          // capacity too small for this batch (caller's bound was wrong)
      cnt = (int)cap;
    }
    msg[(int64_t)tid * stride] = cnt;
    msg[(int64_t)tid * stride + 1] = 0;
  }
  if (t == 0) *n_uniq = total;
  const int f = (int)(t / B);
  const int s = (int)(t - (int64_t)f * B);
  if (f >= F) return;
  int64_t before = 0;
  for (int q = 0; q < f; ++q) before += nu_s[q];
  const int nu = nu_s[f];
  const int32_t* seg = col_seg + (int64_t)f * (B + 1);
  int lo = 0, hi = nu - 1;                         // largest u with seg[u] <= s   (seg[0] = 0)
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (seg[mid] <= s) lo = mid; else hi = mid - 1;
  }
  const int64_t id = col_uid[(int64_t)f * B + lo];
  int64_t o = id >= 0 ? id / rows_per_shard : -1;
  if (o < 0 || o >= n_shard) {
    if (oob) *oob = 1;
    o = id < 0 ? 0 : n_shard - 1;
  }
  int64_t j = before + lo - start_s[(int)o];
  if (j < 0 || j >= cap) j = 0;                    // only after an out-of-range id or a capacity overflow (flag is set)
  const int64_t slot = o * cap + j;
  uidx[(int64_t)f * B + perm[(int64_t)f * B + s]] = slot;
  if (seg[lo] == s) {                              // run head: one thread per unique id
    msg[o * stride + MSG_HDR + j] = id - o * rows_per_shard;
    slot_map[before + lo] = (int32_t)slot;
  }
}

}  // namespace

extern "C" size_t rec_shard_bucketize_workspace_bytes(int64_t n, int n_shard) {
  if (n < 0 || n_shard <= 0) return 0;
  return sizeof(int32_t) * (size_t)(ceil_div64(n, TILE) + 1) * (size_t)n_shard + 256;
}

extern "C" int rec_shard_bucketize_i64(const int64_t* ids, int64_t n, int64_t rows_per_shard, int n_shard,
                                       int64_t* perm, int64_t* send_counts, int64_t* local_ids, int* oob_flag,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!ids || !perm || !send_counts || !local_ids || !workspace || n < 0 || rows_per_shard <= 0 || n_shard <= 0)
    return REC_E_ARG;
  if (n_shard > MAX_SHARD) return REC_E_UNSUPPORTED;
  if (workspace_bytes < rec_shard_bucketize_workspace_bytes(n, n_shard)) return REC_E_WORKSPACE;
  hipStream_t st = as_stream(stream);
  if (n == 0) return (int)hipMemsetAsync(send_counts, 0, sizeof(int64_t) * n_shard, st);
  int n_tiles = (int)ceil_div64(n, TILE);
  int32_t* tile_counts = (int32_t*)workspace;
  hipLaunchKernelGGL(shard_count_kernel, dim3(n_tiles), dim3(256), 0, st, ids, n, rows_per_shard, n_shard, tile_counts,
                     oob_flag);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(shard_scan_kernel, dim3(1), dim3(1024), 0, st, tile_counts, n_tiles, n_shard, send_counts);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(shard_scatter_kernel, dim3(n_tiles), dim3(256), 0, st, ids, n, rows_per_shard, n_shard,
                     tile_counts, perm, local_ids);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_permute_rows_f32(const float* in, const int64_t* perm, int64_t n, int E, int scatter, float* out,
                                    void* stream) {
  if (!in || !perm || !out || n < 0 || E <= 0) return REC_E_ARG;
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)ceil_div64(n * E, 256)), dim3(256), 0, as_stream(stream), in,
                     perm, n, E, scatter, out);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_colsort_shard_map_i64(const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                                         const int32_t* col_nu, int64_t B, int F, int64_t rows_per_shard, int n_shard,
                                         int64_t* uid_local, int64_t* uidx, int64_t* send_counts, int64_t* n_uniq,
                                         int* oob_flag, void* stream) {
  if (!perm || !col_uid || !col_seg || !col_nu || !uid_local || !uidx || !send_counts || !n_uniq || B <= 0 || F <= 0 ||
      rows_per_shard <= 0 || n_shard <= 0)
    return REC_E_ARG;
  if (n_shard > MAX_SHARD || F > REC_MAX_COLS) return REC_E_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(send_counts, 0, sizeof(int64_t) * n_shard, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(shard_map_kernel, dim3((unsigned)ceil_div64(B * F, 256)), dim3(256), 0, st, perm, col_uid, col_seg,
                     col_nu, B, F, rows_per_shard, n_shard, uid_local, uidx, (unsigned long long*)send_counts, n_uniq,
                     oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_colsort_shard_map_fixed_i64(const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                                               const int32_t* col_nu, int64_t B, int F, int64_t rows_per_shard,
                                               int n_shard, int64_t cap, int64_t* msg, int64_t* uidx, int32_t* slot_map,
                                               int64_t* n_uniq, int* oob_flag, void* stream) {
  if (!perm || !col_uid || !col_seg || !col_nu || !msg || !uidx || !slot_map || !n_uniq || B <= 0 || F <= 0 ||
      rows_per_shard <= 0 || n_shard <= 0 || cap <= 0)
    return REC_E_ARG;
  if (n_shard > MAX_SHARD || F > REC_MAX_COLS || (int64_t)n_shard * cap >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;
  hipLaunchKernelGGL(shard_map_fixed_kernel, dim3((unsigned)ceil_div64(B * F, 256)), dim3(256), 0, as_stream(stream),
                     perm, col_uid, col_seg, col_nu, B, F, rows_per_shard, n_shard, cap, msg, uidx, slot_map, n_uniq,
                     oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// De-duplicate-first, fixed-capacity exchange map of a GENERIC lookup (any layer: DSSM towers, DCN, DIN's profile +
// behaviour series): given the sorted-unique plan of the flat id list (rec_dedup_plan_i64: uniq_ids ascending, seg_start,
// perm, *n_uniq) and the block partition owner = id / rows_per_shard, the ascending unique list is already grouped by
// owner, so owner o's ids are the slice [first(o), first(o+1)) found by binary search:
//   msg  [n_shard, 2 + cap] int64   word 0 = ids for this owner, word 1 = 0, then owner-local ids ascending (the layout
//                                   rec_emb_gather_lists_f32 / rec_dedup_plan_sorted_slabs_i64 read); slots beyond the
//                                   count keep what the caller put there (zeros)
//   slot [n] int64                  for lookup i: owner * cap + rank of its id inside the owner's slice = its row in the
//                                   [n_shard * cap, E] buffer the rows come back in
// One launch, two independent jobs (no cross-thread dependency): thread j < n_uniq writes unique j's message word,
// thread p < n finds the unique id of sorted position p (upper bound in seg_start) and writes slot[perm[p]].
// ------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void shard_slab_map_kernel(const int64_t* __restrict__ uniq, const int64_t* __restrict__ n_uniq_p,
                                                             const int32_t* __restrict__ seg_start,
                                                             const int32_t* __restrict__ perm, int64_t n,
                                                             int64_t rows_per_shard, int n_shard, int64_t cap,
                                                             int64_t* __restrict__ msg, int64_t* __restrict__ slot,
                                                             int64_t* __restrict__ uslot, int* __restrict__ flag) {
  __shared__ int64_t first_s[MAX_SHARD + 1];
  const int64_t nu = *n_uniq_p;
  if (threadIdx.x <= n_shard)                    // first unique index of every owner's slice (and the end)
    first_s[threadIdx.x] = threadIdx.x == n_shard ? nu : lower_bound_i64(uniq, nu, (int64_t)threadIdx.x * rows_per_shard);
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n_shard) {                             // headers
    const int64_t c = first_s[t + 1] - first_s[t];
    msg[t * (cap + 2)] = c < cap ? c : cap;
    msg[t * (cap + 2) + 1] = 0;
    if (c > cap && flag) *flag = 1;              // more unique ids for one owner than the exchange capacity
  }
  if (t < nu) {                                  // message word of unique t
    const int64_t id = uniq[t];
    int o = (int)(id / rows_per_shard);
    if (o < 0 || o >= n_shard) { o = o < 0 ? 0 : n_shard - 1; if (flag) *flag = 1; }
    const int64_t r = t - first_s[o];
    if (r < cap) msg[(int64_t)o * (cap + 2) + 2 + r] = id - (int64_t)o * rows_per_shard;
    if (uslot) uslot[t] = (int64_t)o * cap + (r < cap ? r : cap - 1);
  } else if (t < n && uslot) {                   // padded tail: one row PAST the buffer (the caller's dump row)
    uslot[t] = (int64_t)n_shard * cap;
  }
  if (t < n) {                                   // slot of the lookup behind sorted position t
    int64_t lo = 0, hi = nu;                     // largest j with seg_start[j] <= t
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)seg_start[mid] <= t) lo = mid + 1; else hi = mid;
    }
    const int64_t j = lo > 0 ? lo - 1 : 0;
    const int64_t id = uniq[j];
    int o = (int)(id / rows_per_shard);
    o = o < 0 ? 0 : (o >= n_shard ? n_shard - 1 : o);
    int64_t r = j - first_s[o];
    if (r >= cap) r = cap - 1;                   // overflow was flagged above: stay inside the buffer
    slot[perm[t]] = (int64_t)o * cap + r;
  }
}

}  // namespace

extern "C" int rec_shard_slab_map_i64(const int64_t* uniq_ids, const int64_t* n_uniq, const int32_t* seg_start,
                                      const int32_t* perm, int64_t n, int64_t rows_per_shard, int n_shard, int64_t cap,
                                      int64_t* msg, int64_t* slot, int* oob_flag, void* stream) {
  if (n < 0 || rows_per_shard <= 0 || n_shard <= 0 || cap <= 0) return REC_E_ARG;
  if (n_shard > MAX_SHARD) return REC_E_UNSUPPORTED;
  if (n == 0) return REC_OK;
  if (!uniq_ids || !n_uniq || !seg_start || !perm || !msg || !slot) return REC_E_ARG;
  const int64_t threads = n > n_shard ? n : n_shard;
  hipLaunchKernelGGL(shard_slab_map_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0, as_stream(stream),
                     uniq_ids, n_uniq, seg_start, perm, n, rows_per_shard, n_shard, cap, msg, slot, nullptr, oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_shard_slab_map_uslot_i64(const int64_t* uniq_ids, const int64_t* n_uniq, const int32_t* seg_start,
                                            const int32_t* perm, int64_t n, int64_t rows_per_shard, int n_shard,
                                            int64_t cap, int64_t* msg, int64_t* slot, int64_t* uslot, int* oob_flag,
                                            void* stream) {
  if (n < 0 || rows_per_shard <= 0 || n_shard <= 0 || cap <= 0) return REC_E_ARG;
  if (n_shard > MAX_SHARD) return REC_E_UNSUPPORTED;
  if (n == 0) return REC_OK;
  if (!uniq_ids || !n_uniq || !seg_start || !perm || !msg || !slot || !uslot) return REC_E_ARG;
  const int64_t threads = n > n_shard ? n : n_shard;
  hipLaunchKernelGGL(shard_slab_map_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0, as_stream(stream),
                     uniq_ids, n_uniq, seg_start, perm, n, rows_per_shard, n_shard, cap, msg, slot, uslot, oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
