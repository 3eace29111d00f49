// Embedding lookup + FM interaction kernels (SURVEY.md rows K1-K4 values).
//   K1  index assembly        2.FM/CustomLayers.py:138-144
//   K2  Embedding gather      2.FM/CustomLayers.py:129-134,146-147
//   K3  FM sum-square trick   2.FM/CustomLayers.py:149-155  (fused with K2)
//   K4  per-lookup gradient values of the FM part (IndexedSlices values), 2.FM/ModelManager.py:176-177
//
// What bounds these kernels on MI355X (measured, scripts/exp/gather_bench*.hip): a random row read costs one
// 128-B line request whatever the row size (32, 64 and 128-B rows all run at ~50 G lookups/s chip-wide; 256-B
// rows and "64-B row + separate 4-B scalar" at half that).  So (1) tables carry a row stride (`ld`) and the
// FM layers keep `embed` and `w` of one id in ONE 128-B line (fused layout: row = [embed(E) | w | pad],
// ld = next_pow2(E+1) >= 16) -- the first-order weight then rides along for free; (2) a kernel has exactly two
// dependent memory phases (ids, then every row load of a lane issued back to back), because at batch 8192 the
// whole gather is ~4 us and any extra dependent round trip shows.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
struct ColPtrs {
  const int64_t* p[REC_MAX_COLS];
};

__global__ __launch_bounds__(256) void index_pack_kernel(ColPtrs cols, int F, int64_t rows, int64_t* X,
                                                         int64_t ldx, int64_t col0) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * F) return;
  int64_t r = t / F;
  int f = (int)(t - r * F);
  X[r * ldx + col0 + f] = cols.p[f][r];
}

extern "C" int rec_index_pack_i64(const int64_t* const* cols_host, int F, int64_t rows, int64_t* X,
                                  int64_t ldx, int64_t col0, void* stream) {
  if (!cols_host || F <= 0 || rows < 0 || ldx < col0 + F) return REC_E_ARG;
  if (rows == 0) return REC_OK;
  if (!X) return REC_E_ARG;
  for (int f0 = 0; f0 < F; f0 += REC_MAX_COLS) {
    int nf = F - f0 < REC_MAX_COLS ? F - f0 : REC_MAX_COLS;
    ColPtrs cp;
    for (int f = 0; f < nf; ++f) {
      if (!cols_host[f0 + f]) return REC_E_ARG;
      cp.p[f] = cols_host[f0 + f];
    }
    int64_t total = rows * nf;
    hipLaunchKernelGGL(index_pack_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0,
                       as_stream(stream), cp, nf, rows, X, ldx, col0 + f0);
    REC_LAUNCH_CHECK();
  }
  return REC_OK;
}

// Staging of input batches (model_manager.py, the compiled train loop): n device arrays of the same size -> consecutive
// slices of ONE block, one launch (a torch.stack of 26 columns costs ~35 us of host time, and the loop stages a batch
// per step).  16-byte pieces when everything is 16-byte aligned, 4-byte words otherwise.
struct BlockSrcs {
  const void* p[256];
};
template <typename T>
__global__ __launch_bounds__(256) void block_copy_kernel(BlockSrcs srcs, int64_t units, T* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= units) return;
  const int s = (int)blockIdx.y;
  out[(int64_t)s * units + t] = reinterpret_cast<const T*>(srcs.p[s])[t];
}

extern "C" int rec_block_copy(const void* const* srcs_host, int n, int64_t bytes_each, void* out, void* stream) {
  if (!srcs_host || n <= 0 || bytes_each < 0 || (bytes_each & 3) != 0) return REC_E_ARG;
  if (bytes_each == 0) return REC_OK;
  if (!out) return REC_E_ARG;
  for (int s0 = 0; s0 < n; s0 += 256) {
    const int ns = n - s0 < 256 ? n - s0 : 256;
    BlockSrcs b;
    bool vec = (bytes_each & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int s = 0; s < ns; ++s) {
      if (!srcs_host[s0 + s]) return REC_E_ARG;
      b.p[s] = srcs_host[s0 + s];
      vec = vec && (reinterpret_cast<uintptr_t>(b.p[s]) & 15) == 0;
    }
    char* o = reinterpret_cast<char*>(out) + (int64_t)s0 * bytes_each;
    if (vec) {
      const int64_t units = bytes_each / 16;
      hipLaunchKernelGGL(block_copy_kernel<uint4>, dim3((unsigned)ceil_div64(units, 256), ns), dim3(256), 0,
                         as_stream(stream), b, units, reinterpret_cast<uint4*>(o));
    } else {
      const int64_t units = bytes_each / 4;
      hipLaunchKernelGGL(block_copy_kernel<uint32_t>, dim3((unsigned)ceil_div64(units, 256), ns), dim3(256), 0,
                         as_stream(stream), b, units, reinterpret_cast<uint32_t*>(o));
    }
    REC_LAUNCH_CHECK();
  }
  return REC_OK;
}

// tf.keras.metrics.AUC(num_thresholds) as a streaming histogram (2.FM/ModelManager.py:106,180): bucket k of an example =
// the number of thresholds strictly below its prediction; hist [2][n_thr + 1] int64 = examples per (label > 0.5, bucket),
// accumulated with integer atomics (exact, order-independent).  Also adds the n_steps per-step losses onto *loss_acc
// (double; one thread, fixed order).
__global__ __launch_bounds__(256) void auc_hist_kernel(const float* __restrict__ prob, const float* __restrict__ label,
                                                       int64_t n, const float* __restrict__ thr, int n_thr,
                                                       unsigned long long* __restrict__ hist,
                                                       const float* __restrict__ loss_steps, int n_steps,
                                                       double* __restrict__ loss_acc) {
  extern __shared__ unsigned int h_s[];                    // [2][n_thr + 1]
  const int nb = 2 * (n_thr + 1);
  for (int i = threadIdx.x; i < nb; i += 256) h_s[i] = 0u;
  __syncthreads();
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const float p = prob[t];
    int lo = 0, hi = n_thr;                                // first threshold that is not below p
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (thr[mid] < p) lo = mid + 1; else hi = mid;
    }
    atomicAdd(&h_s[(label[t] > 0.5f ? n_thr + 1 : 0) + lo], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256)
    if (h_s[i]) atomicAdd(&hist[i], (unsigned long long)h_s[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && loss_steps && loss_acc) {
    double s = 0.0;
    for (int i = 0; i < n_steps; ++i) s += (double)loss_steps[i];
    *loss_acc += s;
  }
}

extern "C" int rec_auc_hist_update_f32(const float* prob, const float* label, int64_t n, const float* thresholds,
                                       int n_thresholds, int64_t* hist, const float* loss_steps, int n_steps,
                                       double* loss_acc, void* stream) {
  if (n < 0 || n_thresholds <= 0 || n_thresholds > 4096 || n_steps < 0) return REC_E_ARG;
  if (!hist || !thresholds || (n > 0 && (!prob || !label)) || (n_steps > 0 && (!loss_steps || !loss_acc))) return REC_E_ARG;
  if (n == 0 && n_steps == 0) return REC_OK;
  int64_t blocks = ceil_div64(n > 0 ? n : 1, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(auc_hist_kernel, dim3((unsigned)blocks), dim3(256), sizeof(unsigned int) * 2 * (n_thresholds + 1),
                     as_stream(stream), prob, label, n, thresholds, n_thresholds,
                     reinterpret_cast<unsigned long long*>(hist), loss_steps, n_steps, loss_acc);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// K2  plain gather (table row stride ld, output dense [n,E])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_vec4_kernel(const float* __restrict__ table, int64_t V, int lpr,
                                                          int64_t ld, const int64_t* __restrict__ idx, int64_t n,
                                                          float4* __restrict__ out, int* oob) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * lpr) return;
  int64_t r = t / lpr;
  int c = (int)(t - r * lpr);
  int64_t id = idx[r];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((uint64_t)id < (uint64_t)V) {
    v = *reinterpret_cast<const float4*>(table + id * ld + 4 * c);
  } else if (oob) {
    *oob = 1;
  }
  out[t] = v;
}

// Power-of-two rows (E = 4, 8, ..., 256: every embedding width of BASELINE.json's configs): LPR lanes x 16 B cover a
// row, so a wave instruction reads 64/LPR whole rows; each thread has UNR independent row loads in flight (ids first,
// then all rows, then all stores) and the row / lane split is a shift, not a 64-bit division.
template <int LPR, int UNR>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int64_t V, int64_t ld,
                                                          const int64_t* __restrict__ idx, int64_t n,
                                                          float4* __restrict__ out, int* oob) {
  constexpr int R = 256 / LPR;                       // rows per workgroup pass
  const int c = threadIdx.x & (LPR - 1), rs = threadIdx.x / LPR;
  const int64_t r0 = (int64_t)blockIdx.x * (R * UNR) + rs;
  int64_t id[UNR];
#pragma unroll
  for (int j = 0; j < UNR; ++j) {
    const int64_t r = r0 + (int64_t)j * R;
    id[j] = r < n ? idx[r] : -1;
  }
  float4 v[UNR];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < UNR; ++j) {
    const bool ok = (uint64_t)id[j] < (uint64_t)V;
    bad |= !ok && (r0 + (int64_t)j * R < n);
    v[j] = *reinterpret_cast<const float4*>(table + (ok ? id[j] : 0) * ld + 4 * c);
    if (!ok) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (bad && oob) *oob = 1;
#pragma unroll
  for (int j = 0; j < UNR; ++j) {
    const int64_t r = r0 + (int64_t)j * R;
    if (r < n) out[r * LPR + c] = v[j];
  }
}

// Owner-side gather of a fixed-capacity exchange (sharded step): n_lists id lists at a stride of hdr + cap int64 words,
// word 0 of a list = how many of its cap slots are used.  Row (q, j) of out [n_lists * cap, 4 * LPR] is read only when
// j < count_q; the unused slots are neither read nor written (nobody indexes them).
template <int LPR, int UNR>
__global__ __launch_bounds__(256) void gather_lists_kernel(const float* __restrict__ table, int64_t V, int64_t ld,
                                                           const int64_t* __restrict__ msg, int64_t cap, int hdr,
                                                           int64_t n, float4* __restrict__ out, int* oob, int nc) {
  constexpr int R = 256 / LPR;
  const int c = threadIdx.x & (LPR - 1), rs = threadIdx.x / LPR;
  const int64_t r0 = (int64_t)blockIdx.x * (R * UNR) + rs;
  const int64_t stride = cap + hdr;
  int64_t id[UNR];
  bool used[UNR];
#pragma unroll
  for (int j = 0; j < UNR; ++j) {
    const int64_t r = r0 + (int64_t)j * R;
    used[j] = false;
    id[j] = -1;
    if (r < n) {
      const int64_t q = r / cap, jj = r - q * cap;
      used[j] = jj < msg[q * stride];
      if (used[j]) id[j] = msg[q * stride + hdr + jj];
    }
  }
  float4 v[UNR];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < UNR; ++j) {
    const bool ok = (uint64_t)id[j] < (uint64_t)V;
    bad |= used[j] && !ok;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && c < nc) v[j] = *reinterpret_cast<const float4*>(table + id[j] * ld + 4 * c);
  }
  if (bad && oob) *oob = 1;
#pragma unroll
  for (int j = 0; j < UNR; ++j) {                      // rows of nc 16-byte pieces (nc = LPR: the whole row)
    const int64_t r = r0 + (int64_t)j * R;
    if (used[j] && c < nc) out[r * nc + c] = v[j];
  }
}

__global__ __launch_bounds__(256) void gather_scalar_kernel(const float* __restrict__ table, int64_t V, int E,
                                                            int64_t ld, const int64_t* __restrict__ idx, int64_t n,
                                                            float* __restrict__ out, int* oob) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * E) return;
  int64_t r = t / E;
  int d = (int)(t - r * E);
  int64_t id = idx[r];
  float v = 0.f;
  if ((uint64_t)id < (uint64_t)V) {
    v = table[id * ld + d];
  } else if (oob) {
    *oob = 1;
  }
  out[t] = v;
}

static inline bool vec4_ok(const void* p, int E, int64_t ld) {
  return E % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

extern "C" int rec_emb_gather_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* idx, int64_t n,
                                  float* out, int* oob_flag, void* stream) {
  if (V <= 0 || E <= 0 || ld < E || n < 0) return REC_E_ARG;
  if (n == 0) return REC_OK;
  if (!table || !idx || !out) return REC_E_ARG;
  if (vec4_ok(table, E, ld) && vec4_ok(out, E, E) && (E & (E - 1)) == 0 && E >= 4 && E <= 256) {
    const int lpr = E / 4;
    constexpr int UNR = 4;
    const unsigned grid = (unsigned)ceil_div64(n, (256 / lpr) * UNR);
#define GATHER_ROWS(L)                                                                                           \
  hipLaunchKernelGGL((gather_rows_kernel<L, UNR>), dim3(grid), dim3(256), 0, as_stream(stream), table, V, ld, idx, n, \
                     (float4*)out, oob_flag)
    switch (lpr) {
      case 1: GATHER_ROWS(1); break;
      case 2: GATHER_ROWS(2); break;
      case 4: GATHER_ROWS(4); break;
      case 8: GATHER_ROWS(8); break;
      case 16: GATHER_ROWS(16); break;
      case 32: GATHER_ROWS(32); break;
      default: GATHER_ROWS(64); break;
    }
#undef GATHER_ROWS
  } else if (vec4_ok(table, E, ld) && vec4_ok(out, E, E)) {
    int lpr = E / 4;
    hipLaunchKernelGGL(gather_vec4_kernel, dim3((unsigned)ceil_div64(n * lpr, 256)), dim3(256), 0,
                       as_stream(stream), table, V, lpr, ld, idx, n, (float4*)out, oob_flag);
  } else {
    hipLaunchKernelGGL(gather_scalar_kernel, dim3((unsigned)ceil_div64(n * E, 256)), dim3(256), 0,
                       as_stream(stream), table, V, E, ld, idx, n, out, oob_flag);
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_emb_gather_lists_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* msg,
                                        int n_lists, int64_t cap, float* out, int* oob_flag, void* stream) {
  if (V <= 0 || E <= 0 || ld < E || n_lists <= 0 || cap <= 0 || !table || !msg || !out) return REC_E_ARG;
  if (!(vec4_ok(table, E, ld) && vec4_ok(out, E, E) && E >= 4 && E <= 256)) return REC_E_UNSUPPORTED;
  const int nc = E / 4;                                // 16-byte pieces copied per row (rows of E floats in `out`)
  int lpr = 1;
  while (lpr < nc) lpr *= 2;                           // lanes per row: the next power of two
  constexpr int UNR = 4;
  const int64_t n = (int64_t)n_lists * cap;
  const unsigned grid = (unsigned)ceil_div64(n, (256 / lpr) * UNR);
#define GATHER_LISTS(L)                                                                                            \
  hipLaunchKernelGGL((gather_lists_kernel<L, UNR>), dim3(grid), dim3(256), 0, as_stream(stream), table, V, ld, msg, cap, \
                     2, n, (float4*)out, oob_flag, nc)
  switch (lpr) {
    case 1: GATHER_LISTS(1); break;
    case 2: GATHER_LISTS(2); break;
    case 4: GATHER_LISTS(4); break;
    case 8: GATHER_LISTS(8); break;
    case 16: GATHER_LISTS(16); break;
    case 32: GATHER_LISTS(32); break;
    default: GATHER_LISTS(64); break;
  }
#undef GATHER_LISTS
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// K2+K3 fused forward, vector form.
//   LPR   lanes (float4 each) that cover one table row of ld = 4*LPR floats
//   NE    = E/4 leading lanes of a row that carry embedding dims
//   fused : w[id] is float NE*4 of the same row (lane NE, component x); else w is a separate table
//   SPLIT lane groups share the F fields of one example (fills the chip at batch 8192)
// ids of the workgroup are read coalesced and parked in LDS as 32-bit row numbers (-1 = out of range).
// ------------------------------------------------------------------------------------------------
template <int LPR, int SPLIT, bool FUSED>
__global__ __launch_bounds__(256) void emb_fm_fwd_vec_kernel(
    const float* __restrict__ embed, int64_t ld_e, const float* __restrict__ w, int64_t ld_w,
    const float* __restrict__ bias, int64_t V, int NE, const int64_t* __restrict__ idx, int64_t B, int F,
    float* __restrict__ z, float* __restrict__ prob, float4* __restrict__ emb_out, float4* __restrict__ sumvec,
    int* oob) {
  constexpr int LPE = LPR * SPLIT;   // lanes per example
  constexpr int EPW = 256 / LPE;     // examples per workgroup
  constexpr int NL = 16;             // row loads in flight per lane
  extern __shared__ int ids_lds[];   // [EPW * F]
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * EPW;
  const int n_ex = (B - b0 < EPW) ? (int)(B - b0) : EPW;
  bool bad = false;
  for (int i = tid; i < n_ex * F; i += 256) {
    int64_t id = idx[b0 * F + i];
    bool ok = (uint64_t)id < (uint64_t)V;
    bad |= !ok;
    ids_lds[i] = ok ? (int)id : -1;
  }
  if (bad && oob) *oob = 1;
  __syncthreads();
  const int e = tid / LPE, q = tid % LPE, c = q % LPR, s = q / LPR;
  if (e >= n_ex) return;             // whole examples leave together, after the only barrier
  const int64_t b = b0 + e;
  const int FP = (F + SPLIT - 1) / SPLIT;
  const int f_begin = s * FP;
  const int f_end = (f_begin + FP < F) ? f_begin + FP : F;
  const int* my_ids = ids_lds + e * F;
  const bool emb_lane = c < NE;
  float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 Q = make_float4(0.f, 0.f, 0.f, 0.f);
  float first = 0.f;
  for (int f0 = f_begin; f0 < f_end; f0 += NL) {
    int id[NL];
    float4 ev[NL];
    float wv[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) id[u] = (f0 + u < f_end) ? my_ids[f0 + u] : -1;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      bool ok = id[u] >= 0;
      ev[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      wv[u] = 0.f;
      if (FUSED) {
        // every lane of the row group reads its 16 B of the line; lane NE holds w in .x
        if (ok && c <= NE) ev[u] = *reinterpret_cast<const float4*>(embed + (int64_t)id[u] * ld_e + 4 * c);
      } else {
        if (ok && emb_lane) ev[u] = *reinterpret_cast<const float4*>(embed + (int64_t)id[u] * ld_e + 4 * c);
        if (ok && (u % LPR) == c) wv[u] = w[(int64_t)id[u] * ld_w];
      }
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      if (FUSED && c == NE) {
        first += ev[u].x;
      } else if (emb_lane) {
        S.x += ev[u].x; S.y += ev[u].y; S.z += ev[u].z; S.w += ev[u].w;
        Q.x += ev[u].x * ev[u].x; Q.y += ev[u].y * ev[u].y; Q.z += ev[u].z * ev[u].z; Q.w += ev[u].w * ev[u].w;
        if (emb_out && f0 + u < f_end) emb_out[(b * F + f0 + u) * NE + c] = ev[u];
      }
      first += wv[u];
    }
  }
#pragma unroll
  for (int o = LPR; o < LPE; o <<= 1) {   // add the SPLIT partial sums of an example (lanes c, c+LPR, ...)
    S.x += __shfl_xor(S.x, o, 64); S.y += __shfl_xor(S.y, o, 64);
    S.z += __shfl_xor(S.z, o, 64); S.w += __shfl_xor(S.w, o, 64);
    Q.x += __shfl_xor(Q.x, o, 64); Q.y += __shfl_xor(Q.y, o, 64);
    Q.z += __shfl_xor(Q.z, o, 64); Q.w += __shfl_xor(Q.w, o, 64);
  }
  float part = (S.x * S.x - Q.x) + (S.y * S.y - Q.y) + (S.z * S.z - Q.z) + (S.w * S.w - Q.w);
  part = group_sum<LPR>(part);            // over the row's lanes (non-embedding lanes contribute 0)
  first = group_sum<LPE>(first);
  if (sumvec && s == 0 && emb_lane) sumvec[b * NE + c] = S;
  if (q == 0) {
    float zz = bias[0] + first + 0.5f * part;
    if (z) z[b] = zz;
    if (prob) prob[b] = sigmoid_acc(zz);
  }
}

// Generic form: any E <= 4*64 and any strides; a group of GW lanes (power of two >= min(E,64)) per example,
// lane c owns dims c, c+GW, ...
template <int GW, int NACC>
__global__ __launch_bounds__(256) void emb_fm_fwd_gen_kernel(
    const float* __restrict__ embed, int64_t ld_e, const float* __restrict__ w, int64_t ld_w,
    const float* __restrict__ bias, int64_t V, int E, const int64_t* __restrict__ idx, int64_t B, int F,
    float* __restrict__ z, float* __restrict__ prob, float* __restrict__ emb_out, float* __restrict__ sumvec,
    int* oob) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t b = t / GW;
  int c = (int)(t % GW);
  if (b >= B) return;
  const int64_t* ids = idx + b * F;
  float S[NACC], Q[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) S[a] = Q[a] = 0.f;
  float first = 0.f;
  bool bad = false;
  for (int f = 0; f < F; ++f) {
    int64_t id = ids[f];
    bool ok = (uint64_t)id < (uint64_t)V;
    bad |= !ok;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      int d = c + a * GW;
      float e = (ok && d < E) ? embed[id * ld_e + d] : 0.f;
      S[a] += e;
      Q[a] += e * e;
      if (emb_out && d < E) emb_out[(b * F + f) * E + d] = e;
    }
    if (ok && (f % GW) == c) first += w[id * ld_w];
  }
  if (bad && oob) *oob = 1;
  float part = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    part += S[a] * S[a] - Q[a];
    int d = c + a * GW;
    if (sumvec && d < E) sumvec[b * E + d] = S[a];
  }
  part = group_sum<GW>(part);
  first = group_sum<GW>(first);
  if (c == 0) {
    float zz = bias[0] + first + 0.5f * part;
    if (z) z[b] = zz;
    if (prob) prob[b] = sigmoid_acc(zz);
  }
}

namespace {

struct FwdArgs {
  const float* embed; int64_t ld_e; const float* w; int64_t ld_w; const float* bias; int64_t V; int E;
  const int64_t* idx; int64_t B; int F; float* z; float* prob; float* emb_out; float* sumvec; int* oob;
  hipStream_t st;
};

template <int LPR, int SPLIT, bool FUSED>
void launch_vec(const FwdArgs& a) {
  constexpr int EPW = 256 / (LPR * SPLIT);
  size_t lds = sizeof(int) * (size_t)EPW * (size_t)a.F;
  hipLaunchKernelGGL((emb_fm_fwd_vec_kernel<LPR, SPLIT, FUSED>), dim3((unsigned)ceil_div64(a.B, EPW)), dim3(256), lds,
                     a.st, a.embed, a.ld_e, a.w, a.ld_w, a.bias, a.V, a.E / 4, a.idx, a.B, a.F, a.z, a.prob,
                     (float4*)a.emb_out, (float4*)a.sumvec, a.oob);
}

// lanes of an example = LPR * SPLIT: split the fields of an example over 2 or 4 lane groups when that is needed
// to keep <= 16 row loads per lane or to fill the chip (>= ~1024 waves)
template <int LPR, bool FUSED>
bool dispatch_vec(const FwdArgs& a) {
  int split = 1;
  while (split < 4 && LPR * split * 2 <= 64 && (a.F + split - 1) / split > 1 &&
         ((a.F + split - 1) / split > 16 || a.B * LPR * split < 1024 * 64))
    split *= 2;
  if (sizeof(int) * (size_t)(256 / (LPR * split)) * (size_t)a.F > 60 * 1024) return false;
  if (split == 1) launch_vec<LPR, 1, FUSED>(a);
  else if (split == 2) launch_vec<LPR, 2, FUSED>(a);
  else launch_vec<LPR, (LPR * 4 <= 64 ? 4 : 2), FUSED>(a);
  return true;
}

template <bool FUSED>
bool dispatch_lpr(const FwdArgs& a, int lpr) {
  switch (lpr) {
    case 1: return dispatch_vec<1, FUSED>(a);
    case 2: return dispatch_vec<2, FUSED>(a);
    case 4: return dispatch_vec<4, FUSED>(a);
    case 8: return dispatch_vec<8, FUSED>(a);
    case 16: return dispatch_vec<16, FUSED>(a);
    default: return false;
  }
}

}  // namespace

#define FWD_GEN(GW, NACC)                                                                                      \
  hipLaunchKernelGGL((emb_fm_fwd_gen_kernel<GW, NACC>), dim3((unsigned)ceil_div64(B * GW, 256)), dim3(256), 0, \
                     as_stream(stream), embed, ld_e, w, ld_w, bias, V, E, idx, B, F, z, prob, emb_out, sumvec,  \
                     oob_flag)

extern "C" int rec_emb_fm_fwd_f32(const float* embed, int64_t ld_e, const float* w, int64_t ld_w, const float* bias,
                                  int64_t V, int E, const int64_t* idx, int64_t B, int F, float* z, float* prob,
                                  float* emb_out, float* sumvec, int* oob_flag, void* stream) {
  if (V <= 0 || E <= 0 || B < 0 || F <= 0 || ld_e < E || ld_w < 1) return REC_E_ARG;
  if (E > 256) return REC_E_UNSUPPORTED;
  if (B == 0) return REC_OK;
  if (!embed || !w || !bias || !idx) return REC_E_ARG;
  if (V >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;
  FwdArgs a{embed, ld_e, w, ld_w, bias, V, E, idx, B, F, z, prob, emb_out, sumvec, oob_flag, as_stream(stream)};
  bool done = false;
  bool out_ok = (!emb_out || (reinterpret_cast<uintptr_t>(emb_out) & 15) == 0) &&
                (!sumvec || (reinterpret_cast<uintptr_t>(sumvec) & 15) == 0);
  if (vec4_ok(embed, E, ld_e) && out_ok) {
    // fused layout: w is float E of the embed row and the row is a power-of-two number of float4 lanes
    bool fused = (w == embed + E) && ld_w == ld_e && ld_e > E && (ld_e & (ld_e - 1)) == 0 && ld_e <= 64;
    if (fused) done = dispatch_lpr<true>(a, (int)(ld_e / 4));
    if (!done && (E & (E - 1)) == 0 && E <= 64) done = dispatch_lpr<false>(a, E / 4);
  }
  if (!done) {
    if (E <= 1) FWD_GEN(1, 1);
    else if (E <= 2) FWD_GEN(2, 1);
    else if (E <= 4) FWD_GEN(4, 1);
    else if (E <= 8) FWD_GEN(8, 1);
    else if (E <= 16) FWD_GEN(16, 1);
    else if (E <= 32) FWD_GEN(32, 1);
    else if (E <= 64) FWD_GEN(64, 1);
    else if (E <= 128) FWD_GEN(64, 2);
    else FWD_GEN(64, 4);
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// K4 values
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emb_fm_bwd_vals_kernel(
    const float* __restrict__ embed, int64_t ld_e, int64_t V, int E, const int64_t* __restrict__ idx, int64_t B, int F,
    const float* __restrict__ gz, const float* __restrict__ sumvec, const float* __restrict__ emb_rows,
    const float* __restrict__ extra, float* __restrict__ dvals) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = B * F * E;
  if (t >= total) return;
  int64_t j = t / E;  // lookup
  int d = (int)(t - j * E);
  int64_t b = j / F;
  float e;
  if (emb_rows) {
    e = emb_rows[t];
  } else {
    int64_t id = idx[j];
    e = ((uint64_t)id < (uint64_t)V) ? embed[id * ld_e + d] : 0.f;
  }
  float v = gz[b] * (sumvec[b * E + d] - e);
  if (extra) v += extra[t];
  dvals[t] = v;
}

__global__ __launch_bounds__(256) void emb_fm_bwd_vals_vec_kernel(
    const float* __restrict__ embed, int64_t ld_e, int64_t V, int lpr, const int64_t* __restrict__ idx, int64_t B,
    int F, const float* __restrict__ gz, const float4* __restrict__ sumvec, const float4* __restrict__ emb_rows,
    const float4* __restrict__ extra, float4* __restrict__ dvals) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = B * F * lpr;
  if (t >= total) return;
  int64_t j = t / lpr;
  int c = (int)(t - j * lpr);
  int64_t b = j / F;
  float4 e;
  if (emb_rows) {
    e = emb_rows[t];
  } else {
    int64_t id = idx[j];
    e = ((uint64_t)id < (uint64_t)V) ? *reinterpret_cast<const float4*>(embed + id * ld_e + 4 * c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float g = gz[b];
  float4 s = sumvec[b * lpr + c];
  float4 v = make_float4(g * (s.x - e.x), g * (s.y - e.y), g * (s.z - e.z), g * (s.w - e.w));
  if (extra) {
    float4 x = extra[t];
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
  dvals[t] = v;
}

extern "C" int rec_emb_fm_bwd_vals_f32(const float* embed, int64_t ld_e, int64_t V, int E, const int64_t* idx,
                                       int64_t B, int F, const float* gz, const float* sumvec, const float* emb_rows,
                                       const float* extra, float* dvals, void* stream) {
  if (E <= 0 || F <= 0 || B < 0 || (embed && ld_e < E)) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if ((!embed && !emb_rows) || !idx || !gz || !sumvec || !dvals) return REC_E_ARG;
  bool al = (reinterpret_cast<uintptr_t>(sumvec) & 15) == 0 && (reinterpret_cast<uintptr_t>(dvals) & 15) == 0 &&
            (!emb_rows || (reinterpret_cast<uintptr_t>(emb_rows) & 15) == 0) &&
            (!extra || (reinterpret_cast<uintptr_t>(extra) & 15) == 0);
  if (E % 4 == 0 && al && (emb_rows || vec4_ok(embed, E, ld_e))) {
    int lpr = E / 4;
    hipLaunchKernelGGL(emb_fm_bwd_vals_vec_kernel, dim3((unsigned)ceil_div64(B * F * lpr, 256)), dim3(256), 0,
                       as_stream(stream), embed, ld_e, V, lpr, idx, B, F, gz, (const float4*)sumvec,
                       (const float4*)emb_rows, (const float4*)extra, (float4*)dvals);
  } else {
    hipLaunchKernelGGL(emb_fm_bwd_vals_kernel, dim3((unsigned)ceil_div64(B * F * E, 256)), dim3(256), 0,
                       as_stream(stream), embed, ld_e, V, E, idx, B, F, gz, sumvec, emb_rows, extra, dvals);
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}
