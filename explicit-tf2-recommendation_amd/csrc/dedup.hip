// De-duplication of an IndexedSlices gradient: sort lookups by id, find the runs, sum the rows of a run.
// Replaces tf.unique + unsorted_segment_sum inside Keras' optimizer (2.FM/ModelManager.py:178-179) and the
// tf.unique of 5.DIN/ModelManager.py:185-186.  Deterministic: a stable radix sort puts the rows of one id
// in ascending position order and every run is summed in that order -- no float atomics.
//
// Everything here is hand-written: the (id, position) sort is an LSD radix sort built for this job (below), the
// run detection, compaction and the segment sums follow it.  Each sort pass is three plain kernels (per-tile digit
// histogram, per-digit scan over the tiles, stable scatter) -- no memsets, no host-side state, no cross-workgroup
// hand-off inside a launch -- so a captured hipGraph of a plan replays like any other kernel chain, at every n.
// (Round 1 used rocPRIM here; its Onesweep path, taken above 2^20 keys, keeps a 4-byte block-id word and the
// look-back states alive through hipMemsetAsync nodes and faulted on graph replay.  Nothing of it is left.)
//
// Hot ids (a Zipf head, or DIN's padding id that fills half of every behaviour series) give runs of 10^5 rows;
// a run longer than LONG rows is therefore cut into chunks of CH sorted positions that separate workgroups sum
// (fixed slot order + fixed LDS tree), and the owner of the run adds the chunk partials in chunk order.
#include "common.h"
#include <cstring>

namespace {

constexpr int TILE = 1024;  // sorted keys per workgroup in the run-detection kernels (256 threads x 4)
constexpr int CH = 256;     // chunk of sorted positions in the long-run path
constexpr int LONG = 256;   // runs longer than this take the chunked path (LONG >= CH: <= 2 long runs per chunk)

// ---- LSD radix sort of (uint32 key, int32 position) pairs --------------------------------------------------------
// Tile = 2048 keys per workgroup (4 waves x 8 rounds x 64 lanes; element = tile0 + wave*512 + round*64 + lane, so the
// order (wave, round, lane) IS the input order and ranking in that order keeps the sort stable).  Digits are <= 8 bits;
// `bits` significant key bits are split evenly over ceil(bits/8) passes.  Pass 0 reads the int64 ids themselves and
// makes up the positions, so no key-preparation pass exists.
constexpr int RS_TILE = 2048, RS_ROUNDS = 8, RS_BINS = 256;

struct Layout {
  size_t keys_tmp, keys_out, pos_tmp, hist, tot, tile_heads, total;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

Layout make_layout(int64_t n) {
  Layout L;
  size_t off = 0;
  const size_t nb = (size_t)ceil_div64(n, RS_TILE);
  L.keys_tmp = off; off = align256(off + sizeof(uint32_t) * n);
  L.keys_out = off; off = align256(off + sizeof(uint32_t) * n);
  L.pos_tmp = off; off = align256(off + sizeof(int32_t) * n);
  L.hist = off; off = align256(off + sizeof(uint32_t) * RS_BINS * nb);
  L.tot = off; off = align256(off + sizeof(uint32_t) * RS_BINS);
  L.tile_heads = off; off = align256(off + sizeof(int32_t) * (size_t)(ceil_div64(n, TILE) + 1));
  L.total = off;
  return L;
}

template <bool FIRST>
__device__ __forceinline__ uint32_t rs_key(const int64_t* __restrict__ ids, const uint32_t* __restrict__ keys, int64_t i) {
  if (FIRST) return (uint32_t)ids[i];
  return keys[i];
}

// per-tile digit counts -> hist[digit * n_tiles + tile]
template <bool FIRST>
__global__ __launch_bounds__(256) void rs_hist_kernel(const int64_t* __restrict__ ids, const uint32_t* __restrict__ keys,
                                                      int64_t n, int shift, uint32_t dmask, int n_tiles,
                                                      uint32_t* __restrict__ hist) {
  __shared__ unsigned int cnt[RS_BINS];
  const int tid = threadIdx.x;
  cnt[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    int64_t i = base + r * 256 + tid;
    if (i < n) atomicAdd(&cnt[(rs_key<FIRST>(ids, keys, i) >> shift) & dmask], 1u);   // integer LDS atomics: exact
  }
  __syncthreads();
  hist[(int64_t)tid * n_tiles + blockIdx.x] = cnt[tid];
}

// one workgroup per digit: exclusive scan of that digit's counts over the tiles (in place) + the digit total
__global__ __launch_bounds__(256) void rs_scan_kernel(uint32_t* __restrict__ hist, int n_tiles,
                                                      uint32_t* __restrict__ tot) {
  __shared__ unsigned int wave_tot[4];
  __shared__ unsigned int carry_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* row = hist + (int64_t)blockIdx.x * n_tiles;
  unsigned int carry = 0;
  for (int c0 = 0; c0 < n_tiles; c0 += 256) {
    int i = c0 + tid;
    unsigned int v = i < n_tiles ? row[i] : 0u;
    unsigned int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned int woff = 0;
    for (int q = 0; q < wave; ++q) woff += wave_tot[q];
    if (i < n_tiles) row[i] = carry + woff + incl - v;
    if (tid == 255) carry_sh = carry + woff + incl;
    __syncthreads();
    carry = carry_sh;
  }
  if (tid == 0) tot[blockIdx.x] = carry;
}

// exclusive scan of 256 values held one per thread (result to every thread's own slot)
__device__ __forceinline__ unsigned int rs_excl_scan_256(unsigned int v, unsigned int* wave_tot /*[4]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned int woff = 0;
  for (int q = 0; q < wave; ++q) woff += wave_tot[q];
  return woff + incl - v;
}

// stable scatter of one tile: rank every key among the tile's keys of the same digit (wave-level match by ballots,
// per-wave digit counters in LDS), lay the tile out digit-sorted in LDS, then copy each digit's run to its global
// slot: tile-exclusive count of the digit (rs_scan_kernel) + the digit's base (scan of the totals, redone here).
template <bool FIRST>
__global__ __launch_bounds__(256) void rs_scatter_kernel(const int64_t* __restrict__ ids,
                                                         const uint32_t* __restrict__ keys_in,
                                                         const int32_t* __restrict__ pos_in, int64_t n, int shift,
                                                         int dbits, int n_tiles, const uint32_t* __restrict__ hist,
                                                         const uint32_t* __restrict__ tot,
                                                         uint32_t* __restrict__ keys_out, int32_t* __restrict__ pos_out) {
  __shared__ unsigned int cnt[4][RS_BINS];       // per-wave digit counters, then the wave's base inside the tile
  __shared__ unsigned int gpos[RS_BINS];         // global slot of the digit's run minus its start inside the tile
  __shared__ unsigned int wave_tot[4];
  __shared__ uint32_t st_key[RS_TILE];
  __shared__ int32_t st_pos[RS_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t dmask = (1u << dbits) - 1u;
  const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
  const int count = (int)((n - tile0 < RS_TILE) ? (n - tile0) : RS_TILE);
#pragma unroll
  for (int w = 0; w < 4; ++w) cnt[w][tid] = 0;
  __syncthreads();
  uint32_t key[RS_ROUNDS];
  int32_t pos[RS_ROUNDS];
  unsigned int loc[RS_ROUNDS];                   // rank among the wave's earlier keys of the same digit
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int e = wave * 512 + r * 64 + lane;
    const bool ok = e < count;
    key[r] = ok ? rs_key<FIRST>(ids, keys_in, tile0 + e) : 0u;
    pos[r] = ok ? (FIRST ? (int32_t)(tile0 + e) : pos_in[tile0 + e]) : 0;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int e = wave * 512 + r * 64 + lane;
    const bool ok = e < count;
    const uint32_t d = (key[r] >> shift) & dmask;
    unsigned long long m = __ballot(ok);
    for (int b = 0; b < dbits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    // lanes of one digit read the counter (one address: broadcast), the lowest of them adds the group's size; LDS
    // operations of a wave complete in program order, so the next round sees the sum
    unsigned int old = 0;
    if (ok) {
      old = cnt[wave][d];
      if ((m & lt) == 0) cnt[wave][d] = old + (unsigned int)__popcll(m);
    }
    loc[r] = old + (unsigned int)__popcll(m & lt);
  }
  __syncthreads();
  {
    const unsigned int c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
    const unsigned int dstart = rs_excl_scan_256(c0 + c1 + c2 + c3, wave_tot);      // digit's start inside the tile
    const unsigned int dbase = rs_excl_scan_256(tot[tid], wave_tot);                // digit's start in the output
    cnt[0][tid] = dstart;
    cnt[1][tid] = dstart + c0;
    cnt[2][tid] = dstart + c0 + c1;
    cnt[3][tid] = dstart + c0 + c1 + c2;
    gpos[tid] = dbase + hist[(int64_t)tid * n_tiles + blockIdx.x] - dstart;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int e = wave * 512 + r * 64 + lane;
    if (e < count) {
      const uint32_t d = (key[r] >> shift) & dmask;
      const unsigned int s = cnt[wave][d] + loc[r];
      st_key[s] = key[r];
      st_pos[s] = pos[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < count; i += 256) {
    const uint32_t k = st_key[i];
    const unsigned int o = gpos[(k >> shift) & dmask] + (unsigned int)i;
    keys_out[o] = k;
    pos_out[o] = st_pos[i];
  }
}

// keys_out / perm <- stable sort of ((uint32)ids[i], i) by the low `bits` key bits.  Only launches kernels.
int radix_sort_ids(const int64_t* ids, int64_t n, unsigned bits, char* ws, const Layout& L, uint32_t* keys_out,
                   int32_t* perm, hipStream_t st) {
  const int n_tiles = (int)ceil_div64(n, RS_TILE);
  uint32_t* keys_tmp = (uint32_t*)(ws + L.keys_tmp);
  int32_t* pos_tmp = (int32_t*)(ws + L.pos_tmp);
  uint32_t* hist = (uint32_t*)(ws + L.hist);
  uint32_t* tot = (uint32_t*)(ws + L.tot);
  if (bits < 1) bits = 1;
  const int passes = (int)((bits + 7) / 8);
  const int w = (int)((bits + passes - 1) / passes);
  const uint32_t* kin = nullptr;
  const int32_t* pin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * w;
    const int dbits = ((int)bits - shift < w) ? (int)bits - shift : w;
    const uint32_t dmask = (1u << dbits) - 1u;
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    uint32_t* kout = to_out ? keys_out : keys_tmp;
    int32_t* pout = to_out ? perm : pos_tmp;
    if (p == 0) {
      hipLaunchKernelGGL((rs_hist_kernel<true>), dim3(n_tiles), dim3(256), 0, st, ids, kin, n, shift, dmask, n_tiles, hist);
      REC_LAUNCH_CHECK();
      hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_BINS), dim3(256), 0, st, hist, n_tiles, tot);
      REC_LAUNCH_CHECK();
      hipLaunchKernelGGL((rs_scatter_kernel<true>), dim3(n_tiles), dim3(256), 0, st, ids, kin, pin, n, shift, dbits,
                         n_tiles, hist, tot, kout, pout);
    } else {
      hipLaunchKernelGGL((rs_hist_kernel<false>), dim3(n_tiles), dim3(256), 0, st, ids, kin, n, shift, dmask, n_tiles, hist);
      REC_LAUNCH_CHECK();
      hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_BINS), dim3(256), 0, st, hist, n_tiles, tot);
      REC_LAUNCH_CHECK();
      hipLaunchKernelGGL((rs_scatter_kernel<false>), dim3(n_tiles), dim3(256), 0, st, ids, kin, pin, n, shift, dbits,
                         n_tiles, hist, tot, kout, pout);
    }
    REC_LAUNCH_CHECK();
    kin = kout;
    pin = pout;
  }
  return REC_OK;
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

// n_dev (optional): the number of valid keys lives on the device (<= n, the capacity the grid was sized for)
__global__ __launch_bounds__(256) void count_heads_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                          int32_t* __restrict__ tile_heads,
                                                          const int64_t* __restrict__ n_dev = nullptr) {
  __shared__ int sh[4];
  if (n_dev) n = *n_dev < n ? *n_dev : n;
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
                                                       int64_t* __restrict__ n_uniq,
                                                       const int64_t* __restrict__ n_dev = nullptr) {
  __shared__ int sh[4];
  __shared__ int wave_tot[4];
  const int64_t cap = n;                          // slots of uniq_ids / seg_start (+1) to fill
  if (n_dev) n = *n_dev < n ? *n_dev : n;
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
  uint32_t pad = n > 0 ? keys[0] : 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int64_t i = i0 + k;
    if (i < cap && i >= all) uniq_ids[i] = (int64_t)pad;
    if (i < cap && i + 1 >= all) seg_start[i + 1] = (int32_t)n;  // covers seg_start[n_uniq .. cap]
  }
  if (i0 == 0 && all == 0) seg_start[0] = 0;
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


// The same union for lists that arrive in fixed-capacity slabs (rec_colsort_shard_map_fixed_i64's message layout): list q
// starts at msg[q*(hdr+cap) + hdr], msg[q*(hdr+cap)] of its cap slots are used.  perm[rank] = q*cap + j = the row of the
// element in the [n_lists*cap, .] payload buffer that travels beside the ids; *n_valid = total number of ids.
__global__ __launch_bounds__(256) void merge_rank_strided_kernel(const int64_t* __restrict__ msg, int n_lists,
                                                                 int64_t cap, int hdr, uint32_t* __restrict__ keys_out,
                                                                 int32_t* __restrict__ perm,
                                                                 int64_t* __restrict__ n_valid) {
  __shared__ int cnt[MAX_LISTS];
  const int64_t stride = cap + hdr;
  if ((int)threadIdx.x < n_lists) {
    int64_t c = msg[(int64_t)threadIdx.x * stride];
    cnt[threadIdx.x] = (int)(c < 0 ? 0 : (c > cap ? cap : c));
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t a = 0;
    for (int q = 0; q < n_lists; ++q) a += cnt[q];
    *n_valid = a;
  }
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int s = (int)(e / cap);
  if (s >= n_lists) return;
  const int64_t j = e - (int64_t)s * cap;
  if (j >= cnt[s]) return;
  const int64_t id = msg[(int64_t)s * stride + hdr + j];
  int64_t rank = j;
  for (int q = 0; q < n_lists; ++q) {
    if (q == s) continue;
    const int64_t* lst = msg + (int64_t)q * stride + hdr;
    int lo = 0, hi = cnt[q];                        // first position whose id is > id (q < s) or >= id (q > s)
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      int64_t v = lst[mid];
      bool before = q < s ? v <= id : v < id;
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
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
  uint32_t* keys_out = (uint32_t*)(ws + L.keys_out);
  int32_t* tile_heads = (int32_t*)(ws + L.tile_heads);
  unsigned end_bit = 1;
  while (end_bit < 32 && (int64_t(1) << end_bit) < V) ++end_bit;
  int rc = radix_sort_ids(ids, n, end_bit, ws, L, keys_out, perm, st);
  if (rc != REC_OK) return rc;
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

extern "C" int rec_dedup_plan_sorted_slabs_i64(const int64_t* msg, int n_lists, int64_t cap, int64_t V,
                                               int64_t* uniq_ids, int32_t* seg_start, int32_t* perm, int64_t* n_uniq,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!msg || !uniq_ids || !seg_start || !perm || !n_uniq || !workspace || cap <= 0 || V <= 0 || n_lists <= 0)
    return REC_E_ARG;
  const int64_t n = (int64_t)n_lists * cap;
  if (n >= (int64_t(1) << 31) || V > (int64_t(1) << 32) || n_lists > MAX_LISTS) return REC_E_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  Layout L = make_layout(n);
  if (workspace_bytes < L.total) return REC_E_WORKSPACE;
  char* ws = (char*)workspace;
  uint32_t* keys_out = (uint32_t*)(ws + L.keys_out);
  int32_t* tile_heads = (int32_t*)(ws + L.tile_heads);
  int64_t* n_valid = (int64_t*)(ws + L.tot);      // the radix sort's digit totals are not used on this path
  hipLaunchKernelGGL(merge_rank_strided_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, msg, n_lists, cap,
                     2, keys_out, perm, n_valid);
  REC_LAUNCH_CHECK();
  int n_tiles = (int)ceil_div64(n, TILE);
  hipLaunchKernelGGL(count_heads_kernel, dim3(n_tiles), dim3(256), 0, st, keys_out, n, tile_heads,
                     (const int64_t*)n_valid);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_kernel, dim3(n_tiles), dim3(256), 0, st, keys_out, n, tile_heads, n_tiles, uniq_ids,
                     seg_start, n_uniq, (const int64_t*)n_valid);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
