// Sibling interaction layers that share the embedding gather:
//   PNN inner product      2.FM/CustomLayers.py:729-745 (PNNLayer.call), :755-792 (SharedFieldsInteraction, IpnLayer)
//   NFM bi-interaction     3.DCN/CustomLayers.py:493-503
//   SIM GSU inner-product attention + sum pooling   7.SIM/CustomLayers.py:88-96
//   FFM field-aware second order                    2.FM/CustomLayers.py:398-425, 428-462
// Each is the gather of embedding.hip with a different epilogue, so each is ONE kernel forward (ids -> rows -> the
// layer's output, nothing materialised in between) and ONE kernel backward that produces the per-lookup gradient rows
// (IndexedSlices values); de-duplication is the shared plan + segment sum of dedup.hip.  All HBM-bound on the random
// row reads; none of them is GEMM-shaped at these sizes (F*(F-1)/2 dot products of length E per example).
#include "common.h"

// (i<j) -> position in the row-major upper triangle, the order tf.boolean_mask walks (2.FM/CustomLayers.py:768-771)
__device__ __forceinline__ int pair_index(int i, int j, int F) { return i * F - (i * (i + 1)) / 2 + (j - i - 1); }

// ------------------------------------------------------------------------------------------------
// PNN inner product.  out[b, f*E + d] = table[X[b,f], d];  out[b, F*E + p(i,j)] = <e_i, e_j>.
// A workgroup takes EX examples: ids -> LDS, rows -> LDS (row stride E+1: lanes that walk different rows hit
// different banks), then every output element is one lane's job and is written coalesced.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emb_ipn_fwd_kernel(const float* __restrict__ table, int64_t V, int E,
                                                          int64_t ld, const int64_t* __restrict__ X, int64_t B, int F,
                                                          int EX, int vec4, float* __restrict__ out, int64_t ld_out, int* oob) {
  extern __shared__ float ipn_lds[];
  const int ES = vec4 ? E + 4 : E + 1, P = F * (F - 1) / 2, W = F * E + P;
  float* rows = ipn_lds;                                    // [EX][F][ES]
  int* ids = reinterpret_cast<int*>(rows + EX * F * ES);    // [EX][F]
  unsigned char* pi = reinterpret_cast<unsigned char*>(ids + EX * F);   // [P]
  unsigned char* pj = pi + P;                                            // [P]
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * EX;
  const int n_ex = (B - b0 < EX) ? (int)(B - b0) : EX;
  bool bad = false;
  for (int i = tid; i < n_ex * F; i += 256) {
    int64_t id = X[b0 * F + i];
    bool ok = (uint64_t)id < (uint64_t)V;
    bad |= !ok;
    ids[i] = ok ? (int)id : -1;
  }
  if (bad && oob) *oob = 1;
  for (int i = tid; i < F - 1; i += 256) {
    int p = pair_index(i, i + 1, F);
    for (int j = i + 1; j < F; ++j, ++p) { pi[p] = (unsigned char)i; pj[p] = (unsigned char)j; }
  }
  __syncthreads();
  // every lane issues its (up to 4) row loads back to back before the first one is consumed: at this size the kernel
  // is a handful of dependent memory round trips, so the loads of a lane must not wait for each other
  if (vec4) {
    const int E4 = E >> 2, total = n_ex * F * E4;
    for (int i0 = tid; i0 < total; i0 += 4 * 256) {
      float4 v[4];
      int r[4], d4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * 256;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        r[u] = -1;
        if (i < total) {
          r[u] = i / E4;
          d4[u] = i - r[u] * E4;
          int id = ids[r[u]];
          if (id >= 0) v[u] = *reinterpret_cast<const float4*>(table + (int64_t)id * ld + 4 * d4[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r[u] < 0) continue;
        *reinterpret_cast<float4*>(rows + r[u] * ES + 4 * d4[u]) = v[u];
      }
    }
  } else {
    const int total = n_ex * F * E;
    for (int i0 = tid; i0 < total; i0 += 4 * 256) {
      float v[4];
      int r[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * 256;
        v[u] = 0.f;
        r[u] = -1;
        if (i < total) {
          r[u] = i / E;
          d[u] = i - r[u] * E;
          int id = ids[r[u]];
          if (id >= 0) v[u] = table[(int64_t)id * ld + d[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r[u] >= 0) rows[r[u] * ES + d[u]] = v[u];
    }
  }
  __syncthreads();
  for (int ex = 0; ex < n_ex; ++ex) {
    const float* re = rows + ex * F * ES;
    float* o = out + (b0 + ex) * ld_out;
    for (int k = tid; k < W; k += 256) {
      float v;
      if (k < F * E) {
        int f = k / E;
        v = re[k + f * (ES - E)];                           // f*ES + d  with  k = f*E + d
      } else {
        int p = k - F * E;
        const float* a = re + (int)pi[p] * ES;
        const float* c = re + (int)pj[p] * ES;
        v = 0.f;
        if (vec4) {                                         // independent 16-byte LDS reads, 4 MACs each
#pragma unroll 2
          for (int d = 0; d < E; d += 4) {
            float4 x = *reinterpret_cast<const float4*>(a + d);
            float4 y = *reinterpret_cast<const float4*>(c + d);
            v += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
          }
        } else {
#pragma unroll 4
          for (int d = 0; d < E; ++d) v += a[d] * c[d];
        }
      }
      o[k] = v;
    }
  }
}

// vals[b*F + i, d] = g[b, i*E + d] + sum_{j != i} g[b, F*E + p(i,j)] * e_j[d];  e is read back from the forward output
__global__ __launch_bounds__(256) void emb_ipn_bwd_kernel(const float* __restrict__ out, int64_t ld_out,
                                                          const float* __restrict__ g, int64_t ld_g, int64_t B, int F,
                                                          int E, int EX, float* __restrict__ vals) {
  extern __shared__ float ipn_lds[];
  const int ES = E + 1, FE = F * E, GW = FE + F * F;
  float* rows = ipn_lds;                 // [EX][F][ES]
  float* gl = rows + EX * F * ES;        // [EX][GW]: the flat part of the gradient row, then G[i][j] = g_pair(i,j)
  const int tid = threadIdx.x;           //           as a full symmetric F x F matrix with a zero diagonal
  const int64_t b0 = (int64_t)blockIdx.x * EX;
  const int n_ex = (B - b0 < EX) ? (int)(B - b0) : EX;
  // 8 independent loads per lane in flight (4 of `out`, 4 of `g`) -- see the forward kernel
  const int total = n_ex * GW;
  for (int i0 = tid; i0 < total; i0 += 4 * 256) {
    float vo[4], vg[4];
    int ex[4], k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + u * 256;
      ex[u] = -1;
      vo[u] = vg[u] = 0.f;
      if (i < total) {
        ex[u] = i / GW;
        k[u] = i - ex[u] * GW;
        int src = k[u];
        if (k[u] >= FE) {
          int ij = k[u] - FE, ii = ij / F, jj = ij - ii * F;
          src = ii == jj ? -1 : FE + (ii < jj ? pair_index(ii, jj, F) : pair_index(jj, ii, F));
        } else {
          vo[u] = out[(b0 + ex[u]) * ld_out + k[u]];
        }
        if (src >= 0) vg[u] = g[(b0 + ex[u]) * ld_g + src];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (ex[u] < 0) continue;
      gl[ex[u] * GW + k[u]] = vg[u];
      if (k[u] < FE) rows[ex[u] * F * ES + k[u] + k[u] / E] = vo[u];
    }
  }
  __syncthreads();
  for (int ex = 0; ex < n_ex; ++ex) {
    const float* re = rows + ex * F * ES;
    const float* gg = gl + ex * GW;
    float* vo = vals + (b0 + ex) * (int64_t)FE;
    for (int k = tid; k < FE; k += 256) {
      int i = k / E, d = k - i * E;
      const float* Gi = gg + FE + i * F;
      float acc = gg[k];
#pragma unroll 8
      for (int j = 0; j < F; ++j) acc += Gi[j] * re[j * ES + d];     // the diagonal entry is zero
      vo[k] = acc;
    }
  }
}

// Examples per workgroup.  A workgroup is load -> barrier -> compute -> store with nothing overlapped inside it, so
// the overlap has to come from MANY resident workgroups per CU: keep the LDS of one at <= 16 KB (>= 8 per CU) and the
// grid at >= 1024 workgroups; only a single example that needs more may take up to 64 KB.
static int ipn_examples_per_group(int64_t B, int F, int E, bool bwd, size_t* lds) {
  const int P = F * (F - 1) / 2;
  int EX = 8;
  while (EX > 1 && (B / EX) < 1024) EX >>= 1;
  for (;; EX >>= 1) {
    size_t bytes = bwd ? (size_t)EX * ((size_t)F * (E + 1) + (size_t)F * E + (size_t)F * F) * 4
                       : (size_t)EX * F * (E + 4) * 4 + (size_t)EX * F * 4 + 2 * (size_t)P + 16;
    if (bytes <= 16 * 1024 || EX == 1) {
      *lds = bytes;
      return (bytes <= 64 * 1024) ? EX : 0;
    }
  }
}

extern "C" int rec_emb_ipn_fwd_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* X, int64_t B, int F,
                                   float* out, int64_t ld_out, int* oob_flag, void* stream) {
  if (V <= 0 || V > INT32_MAX || E <= 0 || ld < E || B < 0 || F <= 0 || F > 255) return REC_E_ARG;
  if (ld_out < (int64_t)F * E + (int64_t)F * (F - 1) / 2) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!table || !X || !out) return REC_E_ARG;
  const int vec4 = (E & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0;
  size_t lds;
  int EX = ipn_examples_per_group(B, F, E, false, &lds);
  if (!EX) return REC_E_ARG;
  hipLaunchKernelGGL(emb_ipn_fwd_kernel, dim3((unsigned)ceil_div64(B, EX)), dim3(256), lds, as_stream(stream), table, V,
                     E, ld, X, B, F, EX, vec4, out, ld_out, oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_emb_ipn_bwd_vals_f32(const float* out, int64_t ld_out, const float* g, int64_t ld_g, int64_t B, int F,
                                        int E, float* vals, void* stream) {
  if (E <= 0 || B < 0 || F <= 0 || F > 255) return REC_E_ARG;
  const int64_t W = (int64_t)F * E + (int64_t)F * (F - 1) / 2;
  if (ld_out < W || ld_g < W) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!out || !g || !vals) return REC_E_ARG;
  size_t lds;
  int EX = ipn_examples_per_group(B, F, E, true, &lds);
  if (!EX) return REC_E_ARG;
  hipLaunchKernelGGL(emb_ipn_bwd_kernel, dim3((unsigned)ceil_div64(B, EX)), dim3(256), lds, as_stream(stream), out,
                     ld_out, g, ld_g, B, F, E, EX, vals);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// NFM bi-interaction pooling.  out[b, d] = 0.5 * ((sum_f e_fd)^2 - sum_f e_fd^2), S[b,d] = sum_f e_fd kept for the
// backward: vals[b*F + f, d] = g[b,d] * (S[b,d] - e_fd).  One lane per (example, 4 dims); the F row loads of a lane
// are independent and issued 8 at a time.
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void emb_bi_fwd_kernel(const float* __restrict__ table, int64_t V, int E, int64_t ld,
                                                         const int64_t* __restrict__ X, int64_t B, int F,
                                                         float* __restrict__ out, int64_t ld_out,
                                                         float* __restrict__ sumvec, int* oob) {
  constexpr int W = VEC ? 4 : 1;
  const int EL = E / W;                                    // lanes per example
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= B * EL) return;
  int64_t b = t / EL;
  int c = (int)(t - b * EL);
  const int64_t* ids = X + b * F;
  float S[W], Q[W];
#pragma unroll
  for (int a = 0; a < W; ++a) S[a] = Q[a] = 0.f;
  bool bad = false;
  for (int f0 = 0; f0 < F; f0 += 8) {
    float e[8][W];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int a = 0; a < W; ++a) e[u][a] = 0.f;
      if (f0 + u < F) {
        int64_t id = ids[f0 + u];
        if ((uint64_t)id < (uint64_t)V) {
          if constexpr (VEC) {
            float4 v = *reinterpret_cast<const float4*>(table + id * ld + 4 * c);
            e[u][0] = v.x; e[u][1] = v.y; e[u][2] = v.z; e[u][3] = v.w;
          } else {
            e[u][0] = table[id * ld + c];
          }
        } else {
          bad = true;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < W; ++a) { S[a] += e[u][a]; Q[a] += e[u][a] * e[u][a]; }
  }
  if (bad && oob) *oob = 1;
#pragma unroll
  for (int a = 0; a < W; ++a) {
    out[b * ld_out + c * W + a] = 0.5f * (S[a] * S[a] - Q[a]);
    sumvec[b * E + c * W + a] = S[a];
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void emb_bi_bwd_kernel(const float* __restrict__ table, int64_t V, int E, int64_t ld,
                                                         const int64_t* __restrict__ X, int64_t B, int F,
                                                         const float* __restrict__ g, int64_t ld_g,
                                                         const float* __restrict__ sumvec, float* __restrict__ vals) {
  constexpr int W = VEC ? 4 : 1;
  const int EL = E / W;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= B * F * EL) return;
  int64_t r = t / EL;                                      // lookup b*F + f
  int c = (int)(t - r * EL);
  int64_t b = r / F;
  int64_t id = X[r];
  bool ok = (uint64_t)id < (uint64_t)V;
  if constexpr (VEC) {
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) e = *reinterpret_cast<const float4*>(table + id * ld + 4 * c);
    float4 S = *reinterpret_cast<const float4*>(sumvec + b * E + 4 * c);
    const float* gg = g + b * ld_g + 4 * c;
    float4 o = make_float4(gg[0] * (S.x - e.x), gg[1] * (S.y - e.y), gg[2] * (S.z - e.z), gg[3] * (S.w - e.w));
    *reinterpret_cast<float4*>(vals + r * E + 4 * c) = o;
  } else {
    float e = ok ? table[id * ld + c] : 0.f;
    vals[r * E + c] = g[b * ld_g + c] * (sumvec[b * E + c] - e);
  }
}

extern "C" int rec_emb_bi_fwd_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* X, int64_t B, int F,
                                  float* out, int64_t ld_out, float* sumvec, int* oob_flag, void* stream) {
  if (V <= 0 || E <= 0 || ld < E || B < 0 || F <= 0 || ld_out < E) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!table || !X || !out || !sumvec) return REC_E_ARG;
  const bool vec = (E & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL((emb_bi_fwd_kernel<true>), dim3((unsigned)ceil_div64(B * (E / 4), 256)), dim3(256), 0,
                       as_stream(stream), table, V, E, ld, X, B, F, out, ld_out, sumvec, oob_flag);
  else
    hipLaunchKernelGGL((emb_bi_fwd_kernel<false>), dim3((unsigned)ceil_div64(B * E, 256)), dim3(256), 0,
                       as_stream(stream), table, V, E, ld, X, B, F, out, ld_out, sumvec, oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_emb_bi_bwd_vals_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* X, int64_t B,
                                       int F, const float* g, int64_t ld_g, const float* sumvec, float* vals,
                                       void* stream) {
  if (V <= 0 || E <= 0 || ld < E || B < 0 || F <= 0 || ld_g < E) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!table || !X || !g || !sumvec || !vals) return REC_E_ARG;
  const bool vec = (E & 3) == 0 && (ld & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(sumvec) |
                     reinterpret_cast<uintptr_t>(vals)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL((emb_bi_bwd_kernel<true>), dim3((unsigned)ceil_div64(B * F * (E / 4), 256)), dim3(256), 0,
                       as_stream(stream), table, V, E, ld, X, B, F, g, ld_g, sumvec, vals);
  else
    hipLaunchKernelGGL((emb_bi_bwd_kernel<false>), dim3((unsigned)ceil_div64(B * F * E, 256)), dim3(256), 0,
                       as_stream(stream), table, V, E, ld, X, B, F, g, ld_g, sumvec, vals);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------
// SIM GSU inner-product attention.  key k_t = concat_r embed[series[b,t,r]] (D = C*E), valid[b,t] = series[b,t,0] !=
// padding_index;  scores[b,t] = valid * <q_b, k_t>;  pooled[b,:] = sum_t scores[b,t] * k_t.
// One workgroup per example, its 4 waves share the time axis; lane l owns dims l, l+64, ...; TB time steps are in
// flight together per wave; padded steps are never read (their score and their gradient are zero by definition).
// backward: gs_t = <gpooled, k_t>;  gq = sum_t valid*gs_t*k_t;  gkeys[b,t,:] = scores_t*gpooled + valid*gs_t*q.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NPL, bool BWD>
__global__ __launch_bounds__(256) void ip_attn_kernel(const float* __restrict__ embed, int64_t ld, int64_t V, int E,
                                                      int C, const int64_t* __restrict__ series, int64_t B, int T,
                                                      const float* __restrict__ q, int64_t ld_q, int64_t padding_index,
                                                      float* __restrict__ scores, float* __restrict__ pooled,
                                                      int64_t ld_p, const float* __restrict__ gpooled, int64_t ld_gp,
                                                      float* __restrict__ gkeys, float* __restrict__ gq, int* oob) {
  constexpr int TB = 8;
  extern __shared__ int attn_ids[];                        // [T*C]; -1 = out of range, -2 = padded step
  const int D = C * E;
  float* partial = reinterpret_cast<float*>(attn_ids + T * C);       // [4 waves][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = blockIdx.x;
  bool bad = false;
  for (int i = threadIdx.x; i < T * C; i += 256) {
    int t = i / C;
    int64_t id = series[(b * T) * C + i];
    int64_t id0 = series[(b * T + t) * C];
    bool ok = (uint64_t)id < (uint64_t)V;
    bool pad = id0 == padding_index;
    bad |= !ok;
    attn_ids[i] = pad ? -2 : (ok ? (int)id : -1);
  }
  if (bad && oob) *oob = 1;
  __syncthreads();
  int rr[NPL], ee[NPL];
  float qv[NPL], gp[NPL], acc[NPL];
#pragma unroll
  for (int a = 0; a < NPL; ++a) {
    int d = lane + 64 * a;
    bool in = d < D;
    rr[a] = in ? d / E : 0;
    ee[a] = in ? d - rr[a] * E : 0;
    qv[a] = in ? q[b * ld_q + d] : 0.f;
    gp[a] = (BWD && in) ? gpooled[b * ld_gp + d] : 0.f;
    acc[a] = 0.f;
  }
  // the 4 waves take the chunks of TB steps round-robin (padding sits at the tail of a series: interleaving keeps the
  // waves balanced)
  for (int t0 = wave * TB; t0 < T; t0 += 4 * TB) {
    float k[TB][NPL];
    bool valid[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      int t = t0 + u;
      valid[u] = t < T && attn_ids[t * C] != -2;          // wave-uniform
#pragma unroll
      for (int a = 0; a < NPL; ++a) {
        k[u][a] = 0.f;
        if (valid[u] && lane + 64 * a < D) {
          int id = attn_ids[t * C + rr[a]];
          if (id >= 0) k[u][a] = embed[(int64_t)id * ld + ee[a]];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      int t = t0 + u;
      if (t >= T) break;
      float dot = 0.f;
      if (valid[u]) {
        float part = 0.f;
#pragma unroll
        for (int a = 0; a < NPL; ++a) part += (BWD ? gp[a] : qv[a]) * k[u][a];
        dot = wave_sum64(part);
      }
      if (!BWD) {
        if (lane == 0) scores[b * T + t] = dot;
#pragma unroll
        for (int a = 0; a < NPL; ++a) acc[a] += dot * k[u][a];
      } else {
        float s = valid[u] ? scores[b * T + t] : 0.f;
#pragma unroll
        for (int a = 0; a < NPL; ++a) {
          acc[a] += dot * k[u][a];
          if (lane + 64 * a < D) gkeys[(b * T + t) * D + lane + 64 * a] = s * gp[a] + dot * qv[a];
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NPL; ++a)
    if (lane + 64 * a < D) partial[wave * D + lane + 64 * a] = acc[a];
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float v = (partial[d] + partial[D + d]) + (partial[2 * D + d] + partial[3 * D + d]);
    if (!BWD) pooled[b * ld_p + d] = v;
    else gq[b * D + d] = v;
  }
}

#define IP_ATTN_LAUNCH(NPL, BWD)                                                                                   \
  hipLaunchKernelGGL((ip_attn_kernel<NPL, BWD>), dim3((unsigned)B), dim3(256), lds, as_stream(stream), \
                     embed, ld, V, E, C, series, B, T, q, ld_q, padding_index, scores, pooled, ld_p, gpooled, ld_gp,   \
                     gkeys, gq, oob_flag)

static int ip_attn_launch(bool bwd, const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series,
                          int64_t B, int T, const float* q, int64_t ld_q, int64_t padding_index, float* scores,
                          float* pooled, int64_t ld_p, const float* gpooled, int64_t ld_gp, float* gkeys, float* gq,
                          int* oob_flag, void* stream) {
  const int D = C * E;
  if (V <= 0 || V > INT32_MAX || E <= 0 || C <= 0 || ld < E || B < 0 || T <= 0 || D > 256 || ld_q < D)
    return REC_E_ARG;
  const size_t lds = (size_t)T * C * 4 + (size_t)4 * D * 4;
  if (lds > 64 * 1024) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!embed || !series || !q || !scores) return REC_E_ARG;
  const int npl = (D + 63) / 64;
  if (!bwd) {
    switch (npl) {
      case 1: IP_ATTN_LAUNCH(1, false); break;
      case 2: IP_ATTN_LAUNCH(2, false); break;
      case 3: IP_ATTN_LAUNCH(3, false); break;
      default: IP_ATTN_LAUNCH(4, false); break;
    }
  } else {
    switch (npl) {
      case 1: IP_ATTN_LAUNCH(1, true); break;
      case 2: IP_ATTN_LAUNCH(2, true); break;
      case 3: IP_ATTN_LAUNCH(3, true); break;
      default: IP_ATTN_LAUNCH(4, true); break;
    }
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_ip_attn_fwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series,
                                   int64_t B, int T, const float* q, int64_t ld_q, int64_t padding_index, float* scores,
                                   float* pooled, int64_t ld_pooled, int* oob_flag, void* stream) {
  if (B > 0 && (!pooled || ld_pooled < (int64_t)C * E)) return REC_E_ARG;
  return ip_attn_launch(false, embed, ld, V, E, C, series, B, T, q, ld_q, padding_index, scores, pooled, ld_pooled,
                        nullptr, 0, nullptr, nullptr, oob_flag, stream);
}

extern "C" int rec_ip_attn_bwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series,
                                   int64_t B, int T, const float* q, int64_t ld_q, int64_t padding_index,
                                   const float* scores, const float* gpooled, int64_t ld_gpooled, float* gkeys,
                                   float* gq, void* stream) {
  if (B > 0 && (!gpooled || !gkeys || !gq || ld_gpooled < (int64_t)C * E)) return REC_E_ARG;
  return ip_attn_launch(true, embed, ld, V, E, C, series, B, T, q, ld_q, padding_index, const_cast<float*>(scores),
                        nullptr, 0, gpooled, ld_gpooled, gkeys, gq, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// FFM, field-aware second order (FieldAwareInteractionLayer, 2.FM/CustomLayers.py:428-462; the loop form
// FFMRankingLayer.call :398-425 computes the same numbers from F separate tables).
// Table v [V, F, E]: v[id, c, :] is the vector id uses against field c, so ONE id's F vectors are contiguous
// (F*E*4 bytes = 13 lines at F=26, E=16) and the workgroup of an example reads whole rows.
//   z[b] = bias + sum_a w[X[b,a]] + sum_{a<c} < v[X[b,a], c, :], v[X[b,c], a, :] >
// Every element v[X[b,a], c, :] (c != a) is used exactly once, so nothing is staged: one lane per (pair, 4 dims).
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void ffm_fwd_kernel(const float* __restrict__ v, int64_t ld_v,
                                                      const float* __restrict__ w, int64_t ld_w,
                                                      const float* __restrict__ bias, int64_t V, int E,
                                                      const int64_t* __restrict__ X, int64_t B, int F,
                                                      float* __restrict__ z, float* __restrict__ prob, int* oob) {
  extern __shared__ int ffm_lds[];
  const int P = F * (F - 1) / 2;
  int* ids = ffm_lds;                                                    // [F]
  float* red = reinterpret_cast<float*>(ids + F);                        // [4]
  unsigned char* pi = reinterpret_cast<unsigned char*>(red + 4);         // [P]
  unsigned char* pj = pi + P;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  bool bad = false;
  for (int i = tid; i < F; i += 256) {
    int64_t id = X[b * F + i];
    bool ok = (uint64_t)id < (uint64_t)V;
    bad |= !ok;
    ids[i] = ok ? (int)id : -1;
  }
  if (bad && oob) *oob = 1;
  for (int i = tid; i < F - 1; i += 256) {
    int p = pair_index(i, i + 1, F);
    for (int j = i + 1; j < F; ++j, ++p) { pi[p] = (unsigned char)i; pj[p] = (unsigned char)j; }
  }
  __syncthreads();
  constexpr int W = VEC ? 4 : 1;
  const int EL = E / W;
  float acc = 0.f;
  for (int i = tid; i < P * EL; i += 256) {
    int p = i / EL, c4 = i - p * EL;
    int a = pi[p], c = pj[p];
    int ia = ids[a], ic = ids[c];
    if (ia < 0 || ic < 0) continue;                        // an out-of-range id contributes zeros
    const float* pa = v + (int64_t)ia * ld_v + c * E + c4 * W;
    const float* pc = v + (int64_t)ic * ld_v + a * E + c4 * W;
    if constexpr (VEC) {
      float4 x = *reinterpret_cast<const float4*>(pa);
      float4 y = *reinterpret_cast<const float4*>(pc);
      acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    } else {
      acc += pa[0] * pc[0];
    }
  }
  for (int i = tid; i < F; i += 256)
    if (ids[i] >= 0) acc += w[(int64_t)ids[i] * ld_w];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float zz = bias[0] + ((red[0] + red[1]) + (red[2] + red[3]));
    if (z) z[b] = zz;
    if (prob) prob[b] = sigmoid_acc(zz);
  }
}

// Rows of the de-duplicated gradient of v, straight from the plan of rec_dedup_plan_i64 over X (n = B*F lookups):
//   g_rows[u, c, :] = sum over the lookups (b,a) of unique id u, in plan order, of gz[b] * v[X[b,c], a, :]   (c != a)
// One workgroup per unique id (rows u >= n_uniq are zero-filled like every padded tail of the plan).
template <bool VEC>
__global__ __launch_bounds__(128) void ffm_bwd_rows_kernel(const float* __restrict__ v, int64_t ld_v, int64_t V, int E,
                                                           const int64_t* __restrict__ X, int F,
                                                           const float* __restrict__ gz,
                                                           const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ seg_start,
                                                           const int64_t* __restrict__ n_uniq,
                                                           float* __restrict__ g_rows) {
  constexpr int W = VEC ? 4 : 1;
  const int EL = E / W;
  const int64_t u = blockIdx.x;
  float* out = g_rows + u * (int64_t)F * E;
  const bool live = u < *n_uniq;
  const int s = live ? seg_start[u] : 0, e = live ? seg_start[u + 1] : 0;
  for (int item = threadIdx.x; item < F * EL; item += 128) {
    int c = item / EL, c4 = item - c * EL;
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.f;
    for (int r = s; r < e; ++r) {
      int lr = perm[r];
      int b = lr / F, a = lr - b * F;
      if (a == c) continue;
      int64_t id = X[(int64_t)b * F + c];
      if ((uint64_t)id >= (uint64_t)V) continue;
      float g = gz[b];
      const float* src = v + id * ld_v + a * E + c4 * W;
      if constexpr (VEC) {
        float4 x = *reinterpret_cast<const float4*>(src);
        acc[0] += g * x.x; acc[1] += g * x.y; acc[2] += g * x.z; acc[3] += g * x.w;
      } else {
        acc[0] += g * src[0];
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) out[c * E + c4 * W + k] = acc[k];
  }
}

extern "C" int rec_ffm_fwd_f32(const float* v, int64_t ld_v, const float* w, int64_t ld_w, const float* bias, int64_t V,
                               int E, const int64_t* X, int64_t B, int F, float* z, float* prob, int* oob_flag,
                               void* stream) {
  if (V <= 0 || V > INT32_MAX || E <= 0 || F <= 0 || F > 255 || ld_v < (int64_t)F * E || ld_w < 1 || B < 0)
    return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!v || !w || !bias || !X || (!z && !prob)) return REC_E_ARG;
  const size_t lds = (size_t)F * 4 + 16 + (size_t)F * (F - 1);
  const bool vec = (E & 3) == 0 && (ld_v & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL((ffm_fwd_kernel<true>), dim3((unsigned)B), dim3(256), lds, as_stream(stream), v, ld_v, w, ld_w,
                       bias, V, E, X, B, F, z, prob, oob_flag);
  else
    hipLaunchKernelGGL((ffm_fwd_kernel<false>), dim3((unsigned)B), dim3(256), lds, as_stream(stream), v, ld_v, w, ld_w,
                       bias, V, E, X, B, F, z, prob, oob_flag);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_ffm_bwd_rows_f32(const float* v, int64_t ld_v, int64_t V, int E, const int64_t* X, int64_t B, int F,
                                    const float* gz, const int32_t* perm, const int32_t* seg_start,
                                    const int64_t* n_uniq, float* g_rows, void* stream) {
  if (V <= 0 || E <= 0 || F <= 0 || ld_v < (int64_t)F * E || B < 0 || B * F > INT32_MAX) return REC_E_ARG;
  if (B == 0) return REC_OK;
  if (!v || !X || !gz || !perm || !seg_start || !n_uniq || !g_rows) return REC_E_ARG;
  const bool vec = (E & 3) == 0 && (ld_v & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(g_rows)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL((ffm_bwd_rows_kernel<true>), dim3((unsigned)(B * F)), dim3(128), 0, as_stream(stream), v, ld_v,
                       V, E, X, F, gz, perm, seg_start, n_uniq, g_rows);
  else
    hipLaunchKernelGGL((ffm_bwd_rows_kernel<false>), dim3((unsigned)(B * F)), dim3(128), 0, as_stream(stream), v, ld_v,
                       V, E, X, F, gz, perm, seg_start, n_uniq, g_rows);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
