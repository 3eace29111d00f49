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

// single workgroup: column-wise exclusive scan of tile_counts[n_tiles][n_shard] + shard bases
__global__ __launch_bounds__(64) void shard_scan_kernel(int32_t* __restrict__ tile_counts, int n_tiles, int n_shard,
                                                        int64_t* __restrict__ send_counts) {
  __shared__ int64_t tot[MAX_SHARD];
  int s = threadIdx.x;
  int64_t run = 0;
  if (s < n_shard) {
    for (int t = 0; t < n_tiles; ++t) run += tile_counts[(int64_t)t * n_shard + s];
    tot[s] = run;
    send_counts[s] = run;
  }
  __syncthreads();
  if (s < n_shard) {
    int64_t base = 0;
    for (int q = 0; q < s; ++q) base += tot[q];
    int64_t acc = base;
    for (int t = 0; t < n_tiles; ++t) {
      int c = tile_counts[(int64_t)t * n_shard + s];
      tile_counts[(int64_t)t * n_shard + s] = (int32_t)acc;  // now an offset
      acc += c;
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
  hipLaunchKernelGGL(shard_scan_kernel, dim3(1), dim3(64), 0, st, tile_counts, n_tiles, n_shard, send_counts);
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
