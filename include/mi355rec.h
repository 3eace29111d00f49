/* mi355rec.h -- C ABI of libmi355rec.so: the MI355X (gfx950) embedding + feature-interaction engine.
 *
 * Drop-in boundary.  The reference (PatrickHwang/Explicit-tf2-Recommendation) is pure Python/TF2 and has
 * no FFI of its own; its boundary for this path is the Keras `Layer.__call__` protocol of
 * 2.FM/CustomLayers.py, 3.DCN/CustomLayers.py, 5.DIN/CustomLayers.py as driven by
 * 2.FM/ModelManager.py:87-96,171-181.  Each entry point below replaces the TF op sequence cited next to
 * it; the Python mirror of the Layer classes
 * (explicit-tf2-recommendation_amd/layers.py) is the only caller and binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - all matrices are dense row-major fp32, index tensors are int64 (as the reference's Inputs are,
 *     2.FM/ModelManager.py:92);
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no allocation, no
 *     synchronisation, graph-capture safe); inputs are borrowed for the duration of the enqueued work;
 *   - return value: 0 ok, <0 argument error (REC_E_*), >0 a hipError_t from the launch;
 *   - `oob_flag` (optional int32*): kernels never read outside a table -- an id outside [0,V) contributes
 *     a zero row and sets *oob_flag = 1, which the Python layer turns into IndexError (the reference
 *     raises InvalidArgumentError on CPU, aliased wrapError at 2.FM/CustomLayers.py:8).
 */
#ifndef MI355REC_H
#define MI355REC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REC_OK 0
#define REC_E_ARG (-1)
#define REC_E_UNSUPPORTED (-2)
#define REC_E_WORKSPACE (-3)

#define REC_MAX_COLS 128

/* activation kinds shared by the dense entry points */
enum { REC_ACT_NONE = 0, REC_ACT_RELU = 1, REC_ACT_SIGMOID = 2, REC_ACT_TANH = 3 };

/* epilogues of rec_gemm_f32 */
enum {
  REC_EPI_NONE = 0,      /* C = A.B                                   */
  REC_EPI_BIAS = 1,      /* C = A.B + bias[n]                          */
  REC_EPI_BIAS_RELU = 2, /* C = relu(A.B + bias[n])                    */
  REC_EPI_BIAS_SIGMOID = 3,
  REC_EPI_BIAS_TANH = 4,
  REC_EPI_CROSS = 5,     /* C = e0[m,n] * (A.B + bias[n]) + e1[m,n]    (MatrixCrossLayer, 3.DCN/CustomLayers.py:301-303) */
  REC_EPI_ADD = 6        /* C = A.B + e1[m,n]                          */
};

int rec_version(void);

/* ---- K1  index assembly: expand_dims + concat(axis=1) of F int64 columns (2.FM/CustomLayers.py:138-144).
 * cols_host: HOST array of F device pointers, each a column of `rows` int64 (a [B,1] or [B] tensor; for the
 * DIN series stack, 5.DIN/CustomLayers.py:258, a [B,T] tensor with rows = B*T).  Writes X[r*ldx + col0 + f]. */
int rec_index_pack_i64(const int64_t* const* cols_host, int F, int64_t rows, int64_t* X, int64_t ldx,
                       int64_t col0, void* stream);

/* Staging of input batches for the compiled train loop (2.FM/ModelManager.py:183-199 iterates a tf.data pipeline; here
 * the batches of a chunk of steps are copied into fixed device buffers so that the captured step sees constant addresses):
 * n device arrays of bytes_each bytes (a multiple of 4) -> out[s * bytes_each ...], one launch.  srcs_host: HOST array of
 * n device pointers. */
int rec_block_copy(const void* const* srcs_host, int n, int64_t bytes_each, void* out, void* stream);
/* tf.keras.metrics.AUC(num_thresholds) / Mean(loss) accumulated on the device (2.FM/ModelManager.py:106-107,180-181):
 * hist [2][n_thresholds + 1] int64 += examples per (label > 0.5, number of thresholds strictly below the prediction);
 * *loss_acc (double) += the n_steps per-step losses.  Integer atomics: exact and order-independent. */
int rec_auc_hist_update_f32(const float* prob, const float* label, int64_t n, const float* thresholds, int n_thresholds,
                            int64_t* hist, const float* loss_steps, int n_steps, double* loss_acc, void* stream);

/* Tables: row-major fp32 with a row stride `ld` >= E floats (dense table: ld = E).  The FM-family layers keep
 * `embed` [V,E] and `w` [V,1] of one id in ONE 128-byte line -- fused layout, row = [embed(E) | w | pad] with
 * ld = next_pow2(E+1) >= 16, passed as embed = base, w = base + E, ld_e = ld_w = ld -- because a random row
 * read costs one 128-B line request whatever its size (measured), so the first-order weight is free.
 *
 * ---- K2  Embedding(V,E)(X) -> gather (2.FM/CustomLayers.py:129-134,146-147).  out[i,:] = table[idx[i],:]. */
int rec_emb_gather_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* idx, int64_t n,
                       float* out, int* oob_flag, void* stream);

/* ---- K2+K3 fused: w(X), embed(X), reduce_sum / square / subtract / 0.5*reduce_sum
 * (2.FM/CustomLayers.py:146-153, 289-297).  z[b] = bias + sum_f w[X[b,f]] + 0.5*sum_d(S_d^2 - sum_f e_fd^2).
 * Optional outputs (NULL to skip): prob[b] = sigmoid(z[b]) (FMRankingLayer, :155); emb_out [B,F,E] (the
 * Flatten() input of DeepFM's DNN part, :300); sumvec [B,E] = S (saved for backward). */
int rec_emb_fm_fwd_f32(const float* embed, int64_t ld_e, const float* w, int64_t ld_w, const float* bias,
                       int64_t V, int E, const int64_t* idx, int64_t B, int F, float* z, float* prob,
                       float* emb_out, float* sumvec, int* oob_flag, void* stream);

/* ---- K4 (values): GradientTape gradient of the FM part w.r.t. the gathered rows, as the IndexedSlices
 * values TF produces (2.FM/ModelManager.py:176-177): dvals[b,f,:] = gz[b]*(S[b,:] - e[b,f,:]) + extra[b,f,:].
 * emb_rows (optional): the rows saved by the forward; NULL -> re-gather from `embed`.  extra (optional):
 * gradient arriving through the DNN part. */
int rec_emb_fm_bwd_vals_f32(const float* embed, int64_t ld_e, int64_t V, int E, const int64_t* idx, int64_t B,
                            int F, const float* gz, const float* sumvec, const float* emb_rows,
                            const float* extra, float* dvals, void* stream);

/* ---- K4 (de-duplication): what Keras' optimizer does to an IndexedSlices gradient before applying it
 * (tf.unique + unsorted_segment_sum); here ids come out ASCENDING and rows of one id are added in a fixed
 * order, so results are run-to-run bit-identical.
 * plan: ids[n] -> uniq_ids[n] (first *n_uniq valid, ascending; the tail is padded with uniq_ids[0]),
 *       seg_start[n+1] (row range of each unique id in the sorted order; empty for the tail),
 *       perm[n] (sorted position -> original position), n_uniq (device int64). */
size_t rec_dedup_workspace_bytes(int64_t n);
int rec_dedup_plan_i64(const int64_t* ids, int64_t n, int64_t V, int64_t* uniq_ids, int32_t* seg_start,
                       int32_t* perm, int64_t* n_uniq, void* workspace, size_t workspace_bytes,
                       void* stream);
/* out[u,:] = sum over s in [seg_start[u], seg_start[u+1]) of vals[perm[s] / row_div, :]   for u in [0,n).
 * row_div = 1 for per-lookup values; row_div = F broadcasts a per-example value (the w table, whose
 * per-lookup gradient is gz[b]). */
/* Same outputs as rec_dedup_plan_i64 for ids that arrive as n_lists ascending, duplicate-free lists laid end to end
 * (list_counts [n_lists], int64, on the device; their sum must be n): the owner-side union of what P requesters
 * send after de-duplicating their own batches.  Rank merge by binary search instead of a sort; rows of one id are
 * summed in list order.  workspace: rec_dedup_workspace_bytes(n). */
int rec_dedup_plan_sorted_lists_i64(const int64_t* ids, int64_t n, const int64_t* list_counts, int n_lists,
                                    int64_t V, int64_t* uniq_ids, int32_t* seg_start, int32_t* perm,
                                    int64_t* n_uniq, void* workspace, size_t workspace_bytes, void* stream);
size_t rec_segment_sum_workspace_bytes(int64_t n, int E);
int rec_segment_sum_f32(const float* vals, int E, const int32_t* perm, const int32_t* seg_start, int64_t n,
                        int32_t row_div, float* out, float* workspace, void* stream);

/* ---- K5/K6  dense: C[M,N] = epi(op(A).op(B)), fp32-exact MFMA (v_mfma_f32_32x32x2_f32).
 * op(A) is [M,K]: transA=0 -> A stored [M,K] (lda), transA=1 -> A stored [K,M].  op(B) is [K,N]:
 * transB=0 -> B stored [K,N], transB=1 -> B stored [N,K].  MatMul+BiasAdd+activation of MLPLayer
 * (2.FM/CustomLayers.py:74-81), Keras Dense (3.DCN/CustomLayers.py:158-167), MatrixCrossLayer
 * (3.DCN/CustomLayers.py:301-303, REC_EPI_CROSS with e0=x0, e1=x_l) and their backward GEMMs.
 * split_k > 1 needs workspace of split_k*M*N floats (partials are summed in a fixed order).
 * aux (optional, REC_EPI_CROSS only, leading dimension ldc): receives U = A.B + bias, kept for the backward pass. */
int rec_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                 const float* B, int64_t ldb, float* C, int64_t ldc, int epilogue, const float* bias,
                 const float* e0, int64_t lde0, const float* e1, int64_t lde1, int split_k,
                 float* workspace, float* aux, void* stream);

/* ---- K6 backward, elementwise part of one MatrixCrossLayer layer: h = g*x0 ; gx0 = (accumulate ? gx0 : 0) + g*u
 * (H = G (.) X0 feeds the dW and dX GEMMs; dX0 += G (.) U_l).  n = B*D. */
int rec_crossnet_mat_bwd_elem_f32(const float* g, const float* x0, const float* u, float* h, float* gx0,
                                  int accumulate, int64_t n, void* stream);

/* elementwise helpers of the dense forward / backward */
/* y = act(x + x2)  (x2 optional; tf.nn.sigmoid(fm_part + dnn_part), 2.FM/CustomLayers.py:155,305) */
int rec_act_fwd_f32(int act, const float* x, const float* x2, float* y, int64_t n, void* stream);
/* dpre = dpost * act'(post)   (in place allowed) */
int rec_act_bwd_f32(int act, const float* post, const float* dpost, float* dpre, int64_t n, void* stream);
/* out[j] = sum_i X[i,j]  (bias gradients), deterministic two-stage sum; workspace of
 * rec_colsum_workspace_bytes(M,N) bytes. */
size_t rec_colsum_workspace_bytes(int64_t M, int64_t N);
int rec_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, float* workspace, void* stream);
/* The same in ONE launch: `counters` = ceil(N/32) ints that are ZERO on entry and zero again on exit (the workgroup that
 * arrives last at a column block's counter adds the partials of all row blocks, in the fixed order of the two-stage
 * form: bit-identical results).  Calls that share counters must not run concurrently. */
int rec_colsum_fused_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, float* workspace, int* counters,
                         void* stream);
/* y = a*x + b*y over n elements */
int rec_axpby_f32(float a, const float* x, float b, float* y, int64_t n, void* stream);
/* dst[r, c0:c0+w] = src[r, 0:w]   (concat / split along the feature axis) */
int rec_copy_cols_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t w,
                      void* stream);

/* ---- K7  CrossLayer, vector mode (3.DCN/CustomLayers.py:195-203): x_{l+1} = x0*(x_l.w_l) + b_l + x_l.
 * w, b: [L,D].  xs (optional) saves x_0..x_{L-1} as [L,B,D] for backward.  y [B,D]. */
int rec_crossnet_vec_fwd_f32(const float* x0, int64_t B, int D, int L, const float* w, const float* b,
                             float* y, float* xs, void* stream);
/* gx0 [B,D], dw [L,D], db [L,D]; workspace: rec_crossnet_vec_bwd_workspace_bytes(B,D,L). */
size_t rec_crossnet_vec_bwd_workspace_bytes(int64_t B, int D, int L);
int rec_crossnet_vec_bwd_f32(const float* x0, int64_t B, int D, int L, const float* w, const float* xs,
                             const float* gy, float* gx0, float* dw, float* db, void* workspace,
                             void* stream);

/* ---- K10  two-tower score (2.FM/CustomLayers.py:233-234): out[b] = (1 - cos(u_b, i_b))/2,
 * l2norm = x*rsqrt(max(sum x^2, 1e-12)). */
int rec_cosine_fwd_f32(const float* u, const float* i, int64_t B, int d, float* out, void* stream);
int rec_cosine_bwd_f32(const float* u, const float* i, int64_t B, int d, const float* gout, float* gu,
                       float* gi, void* stream);

/* ---- K11  reduce_sum(BinaryCrossentropy()(y, p)) and its gradient (2.FM/ModelManager.py:100,175).
 * p: probabilities [n]; y: labels [n].  loss (device scalar) = mean of -(y log(clip(p)+eps) +
 * (1-y) log(1-clip(p)+eps)), eps = 1e-7.  dp (optional) = dL/dp; dz (optional) = dL/dp * p(1-p) (sigmoid head). */
int rec_bce_fwd_bwd_f32(const float* y, const float* p, int64_t n, float* loss, float* dp, float* dz,
                        void* stream);

/* ---- K12  Keras Adam (2.FM/ModelManager.py:104,178-179).  t = 1-based step.
 * dense: m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= lr_t*m/(sqrt(v)+eps). */
int rec_adam_dense_f32(float* var, float* m, float* v, const float* g, int64_t n, int64_t t, float lr,
                       float b1, float b2, float eps, void* stream);
/* Keras' bias-corrected step size lr * sqrt(1 - b2^t) / (1 - b1^t) in float32 as the entry points above compute it from
 * (t, lr, b1, b2) -- for a caller that keeps the values of steps 1..n in a device table. */
float rec_adam_lr_t_f32(float lr, float b1, float b2, int64_t t);
/* Device-side step counter: *step_dev += 1, *lr_t_dev = lr_table[min(*step_dev, n_table) - 1] (one tiny launch).  With it
 * a train step holds no per-step host scalar and can be captured in a hipGraph. */
int rec_adam_advance_f32(int64_t* step_dev, const float* lr_table, int64_t n_table, float* lr_t_dev, void* stream);
/* rec_deepfm_fused3_main_direct_f32 (below) with that advance done by the fused kernel's first thread -- the kernel reads
 * neither word, the launches behind it on the stream see the new step: one launch less per train step. */
int rec_deepfm_fused3_main_direct_adv_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                                          int64_t B, const float* bias, const float* K0, const float* K0T, const float* b0,
                                          const float* K1, const float* b1, const float* K2, const float* b2,
                                          const float* label, float* gz, float* vals, float* prob, int* oob_flag,
                                          void* workspace, const int32_t* dloc, const int32_t* col_nu, float* g_embed_rows,
                                          int64_t* step_dev, const float* lr_table, int64_t n_table, float* lr_t_dev,
                                          void* stream);
/* rec_adam_dense_f32 on up to 16 parameters in ONE launch, step size read from device memory (host arrays of device
 * pointers, copied into the kernel arguments). */
int rec_adam_dense_multi_f32(int n_tensors, float* const* var, float* const* m, float* const* v, const float* const* g,
                             const int64_t* numel, const float* lr_t_dev, float b1, float b2, float eps, void* stream);
/* Keras sparse apply = DENSE SWEEP: m*=b1, v*=b2 on all V rows, m[ids]+=(1-b1)g, v[ids]+=(1-b2)g^2, then
 * var -= lr_t*m/(sqrt(v)+eps) on all V rows.  (uniq_ids, g_rows, n_uniq) as produced by the dedup above
 * (cap = allocated rows of uniq_ids/g_rows).  var has row stride ld; m, v are dense [V,E].
 * side: workspace of cap*3*E floats. */
int rec_adam_sparse_keras_f32(float* var, int64_t ld, float* m, float* v, int64_t V, int E,
                              const int64_t* uniq_ids, const float* g_rows, const int64_t* n_uniq, int64_t cap,
                              float* side, int64_t t, float lr, float b1, float b2, float eps, void* stream);
/* The same Keras sparse apply for BOTH tables of an FM-family layer that share the fused rows [embed(E) | w | pad]
 * (fused = row 0 of the [V,ld] array; w = fused + E): one sweep over the lines instead of two -- swept on its own, w
 * costs a full 128-byte line read + write per row.  Elementwise identical to two rec_adam_sparse_keras_f32 calls.
 * side_e [cap,3,E], side_w [cap,3,1]. */
int rec_adam_sparse_keras_pair_f32(float* fused, int64_t ld, float* m_e, float* v_e, float* m_w, float* v_w, int64_t V,
                                   int E, const int64_t* uniq_ids, const float* g_e_rows, const float* g_w_rows,
                                   const int64_t* n_uniq, int64_t cap, float* side_e, float* side_w, int64_t t,
                                   float lr, float b1, float b2, float eps, void* stream);

/* 'lazy' variant (NOT reference semantics): only the touched rows decay and move. */
int rec_adam_rows_f32(float* var, int64_t ld, float* m, float* v, int64_t V, int E, const int64_t* uniq_ids,
                      const float* g_rows, const int64_t* n_uniq, int64_t cap, int64_t t, float lr,
                      float b1, float b2, float eps, void* stream);

/* ---- L2 on the embedding rows a batch used (5.DIN/ModelManager.py:176-190):
 * loss = factor * l2_loss(table[uniq_ids[0..n_uniq)]) with l2_loss(x) = sum(x^2)/2; rows_out [n,E] = its gradient
 * factor * table[uniq_ids[u]] (zero rows on the padded tail u >= n_uniq).  uniq_ids / n_uniq: a rec_dedup_plan. */
size_t rec_l2_rows_workspace_bytes(int64_t n, int E);
int rec_l2_rows_f32(const float* table, int64_t ld, int64_t V, int E, const int64_t* uniq_ids, const int64_t* n_uniq,
                    int64_t n, float factor, float* rows_out, float* loss, float* workspace, void* stream);

/* ---- retrieval after the DSSM towers (2.FM/OfflineLoader.py:129-162, 2.FM/OnlineServer.py:53-75:
 * item vectors L2-normalised, sklearn BallTree(Euclidean).query(raw user vector, k) -- an exact search, i.e. the k
 * smallest ||u - i_hat||_2 ascending).  rec_l2_normalize_rows_f32: y[r,:] = x[r,:] / ||x[r,:]||_2.
 * rec_topk_l2_f32: out_idx [nq,k] (int64, item row numbers; equal distances: the lower index first), out_dist [nq,k]
 * ascending; k <= 64, d <= 64; if n < k the tail is (-1, +inf).  workspace: rec_topk_l2_workspace_bytes(nq, n, k). */
int rec_l2_normalize_rows_f32(const float* x, int64_t n, int d, int64_t ld_in, float* y, int64_t ld_out, void* stream);
size_t rec_topk_l2_workspace_bytes(int64_t nq, int64_t n, int k);
int rec_topk_l2_f32(const float* queries, int64_t nq, int d, int64_t ldq, const float* items, int64_t n, int64_t ldi,
                    int k, int64_t* out_idx, float* out_dist, void* workspace, size_t workspace_bytes, void* stream);

/* ---- (e) row-wise block sharding of a table (no reference counterpart).
 * owner = id / rows_per_shard.  perm[n]: positions grouped by owner, ascending inside a group;
 * send_counts[n_shard] (int64); local_ids[n] = ids[perm] - owner*rows_per_shard. */
size_t rec_shard_bucketize_workspace_bytes(int64_t n, int n_shard);
int rec_shard_bucketize_i64(const int64_t* ids, int64_t n, int64_t rows_per_shard, int n_shard,
                            int64_t* perm, int64_t* send_counts, int64_t* local_ids, int* oob_flag,
                            void* workspace, size_t workspace_bytes, void* stream);
/* De-duplicate-first exchange plan on top of rec_colsort_plan_i64 (columns own ascending id ranges => the batch's
 * unique ids, column after column, are globally ascending and grouped by owner): uidx [F,B] = compact index of every
 * lookup's id in that list (the fused kernel gathers the exchanged rows by it), uid_local [B*F] = id - owner *
 * rows_per_shard (first *n_uniq valid), send_counts [n_shard] (int64) = unique ids per owner. */
int rec_colsort_shard_map_i64(const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                              const int32_t* col_nu, int64_t B, int F, int64_t rows_per_shard, int n_shard,
                              int64_t* uid_local, int64_t* uidx, int64_t* send_counts, int64_t* n_uniq,
                              int* oob_flag, void* stream);
/* The same plan for FIXED-CAPACITY exchanges (constant split sizes: no count exchange, no host read; the whole sharded
 * step can be captured in a hipGraph).  Every owner has a slab of `cap` slots; cap >= the unique ids one batch can hold
 * for one owner = sum over the fields that intersect the owner's block of min(B, overlap) (else *oob_flag is set).
 * msg [n_shard, 2 + cap] int64: word 0 = unique ids for this owner, word 1 = 0, then the owner-local ids, ascending;
 * uidx [F,B] = owner * cap + j (row of the lookup in the [n_shard * cap, .] buffer the rows come back in);
 * slot_map [B*F] int32 (first *n_uniq valid): rank in the batch's ascending unique list -> owner * cap + j. */
int rec_colsort_shard_map_fixed_i64(const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                                    const int32_t* col_nu, int64_t B, int F, int64_t rows_per_shard, int n_shard,
                                    int64_t cap, int64_t* msg, int64_t* uidx, int32_t* slot_map, int64_t* n_uniq,
                                    int* oob_flag, void* stream);
/* The same kind of plan for a GENERIC lookup (any id list, any layer): on top of rec_dedup_plan_i64 of the flat ids
 * (uniq_ids ascending = grouped by owner under the block partition, seg_start, perm, *n_uniq on the device).
 * msg [n_shard, 2 + cap] as above (words beyond an owner's count are left alone: the caller keeps them zero);
 * slot [n] int64: row of every lookup in the [n_shard * cap, E] buffer the rows come back in.  *oob_flag is set when an
 * owner's unique ids exceed cap or an id lies outside [0, n_shard * rows_per_shard). */
int rec_shard_slab_map_i64(const int64_t* uniq_ids, const int64_t* n_uniq, const int32_t* seg_start, const int32_t* perm,
                           int64_t n, int64_t rows_per_shard, int n_shard, int64_t cap, int64_t* msg, int64_t* slot,
                           int* oob_flag, void* stream);
/* The same, and uslot [n] int64: the slot of every UNIQUE id (rank u of the plan -> owner * cap + j; ranks >= *n_uniq ->
 * n_shard * cap, one row past the buffer).  The plan of the ids is then also the plan of the slots (slot is monotone in the
 * id), which is what lets the backward of the lookup reuse the forward's de-duplication: segment sums in the plan's order,
 * scattered by uslot, are the dense gradient of the [n_shard * cap, E] rows buffer. */
int rec_shard_slab_map_uslot_i64(const int64_t* uniq_ids, const int64_t* n_uniq, const int32_t* seg_start,
                                 const int32_t* perm, int64_t n, int64_t rows_per_shard, int n_shard, int64_t cap,
                                 int64_t* msg, int64_t* slot, int64_t* uslot, int* oob_flag, void* stream);
/* Owner side of it.  Gather for n_lists received slabs (msg layout above): out [n_lists * cap, E], row (q, j) written
 * only for j < count_q = the first E floats of the table row (E a multiple of 4 up to 256, E <= ld; the sharded
 * DeepFM step sends E = 20 of the 32 floats of a fused row: [embed 16 | w | pad]); 16-byte aligned operands. */
int rec_emb_gather_lists_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* msg, int n_lists,
                             int64_t cap, float* out, int* oob_flag, void* stream);
/* Union of the n_lists ascending duplicate-free lists of such a message (rank merge, no sort) as a segment plan over
 * the [n_lists * cap, .] payload rows: uniq_ids [n_lists*cap], seg_start [n_lists*cap + 1], perm [n_lists*cap] (payload
 * row of every sorted position), *n_uniq; tails padded as rec_dedup_plan_i64 does (valid id, empty runs), so
 * rec_segment_sum_f32(n = n_lists*cap) follows without a host read.  workspace: rec_dedup_workspace_bytes(n_lists*cap). */
int rec_dedup_plan_sorted_slabs_i64(const int64_t* msg, int n_lists, int64_t cap, int64_t V, int64_t* uniq_ids,
                                    int32_t* seg_start, int32_t* perm, int64_t* n_uniq, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* Post launch of the fused step for that layout: dense-gradient reduction + per-unique-id sums of (vals, gz) written as
 * packed rows [embed 16 | w | 0 0 0] to g_rows [n_shard * cap, 20] at slot_map[rank]; unused slots are left alone. */
int rec_deepfm_fused_post_slots_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                                    float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                    void* workspace, const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                                    const int32_t* col_nu, const int32_t* slot_map, float* g_rows, void* stream);
/* out[perm[i], :] = in[i, :]   (inverse permutation of received rows) and its transpose */
int rec_permute_rows_f32(const float* in, const int64_t* perm, int64_t n, int E, int scatter, float* out,
                         void* stream);

/* ---- Fused DeepFM train step (2.FM/CustomLayers.py:279-308 under 2.FM/ModelManager.py:171-177) for the reference's
 * default head (embedding_dims 16, mlp_dims [32,8]) on the fused 128-byte row layout (ld = 32): index assembly from the
 * F feature columns, gather, FM, MLP (fp32 MFMA), sigmoid, Keras BCE and the whole backward in ONE kernel + one
 * fixed-order reduction.  Outputs: gz [B] = dL/dz, vals [B*F,16] = IndexedSlices values of `embed` (the values of `w`
 * are gz[b]), dense gradients, loss (device scalar, mean BCE), prob [B] (optional).  F <= 28.
 * workspace: rec_deepfm_fused_workspace_bytes(B, F). */
size_t rec_deepfm_fused_workspace_bytes(int64_t B, int F);
int rec_deepfm_fused_fwd_bwd_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                                 int64_t B, const float* bias, const float* K0, const float* b0, const float* K1,
                                 const float* b1, const float* K2, const float* b2, const float* label, float* gz,
                                 float* vals, float* prob, float* dK0, float* db0, float* dK1, float* db1, float* dK2,
                                 float* db2, float* dbias, float* loss, int* oob_flag, void* workspace, void* stream);
/* The same iteration when the de-duplication plan of the batch (rec_colsort_plan_i64) already exists: the fused kernel,
 * then ONE launch in which the reduction of the workgroup partials and the segment sums of rec_colseg_sum_f32 (packed:
 * of rec_colseg_sum_packed_f32, g_embed_rows [B*F,20], g_w_rows unused) run side by side. */
int rec_deepfm_fused_step_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                              int64_t B, const float* bias, const float* K0, const float* b0, const float* K1,
                              const float* b1, const float* K2, const float* b2, const float* label, float* gz,
                              float* vals, float* prob, float* dK0, float* db0, float* dK1, float* db1, float* dK2,
                              float* db2, float* dbias, float* loss, int* oob_flag, void* workspace,
                              const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                              const int32_t* col_nu, int64_t* uniq_ids, float* g_embed_rows, float* g_w_rows,
                              int64_t* n_uniq, int packed, void* stream);
/* The two halves of rec_deepfm_fused_step_f32 as separate calls (fused kernel | reduction + segment sums): a caller
 * that builds the plan on another stream puts its wait between them, so that only the second half depends on it.
 * These plan-after forms take any row stride ld >= 20 that is a multiple of 4 (rows [embed 16 | w | ...]: 32 for a
 * table, 20 for the rows a sharded step received); the direct-mode form below needs ld = 32. */
int rec_deepfm_fused_main_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                              int64_t B, const float* bias, const float* K0, const float* b0, const float* K1,
                              const float* b1, const float* K2, const float* b2, const float* label, float* gz,
                              float* vals, float* prob, int* oob_flag, void* workspace, void* stream);
int rec_deepfm_fused_post_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0, float* dK1,
                              float* db1, float* dK2, float* db2, float* dbias, float* loss, void* workspace,
                              const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                              const int32_t* col_nu, int64_t* uniq_ids, float* g_embed_rows, float* g_w_rows,
                              int64_t* n_uniq, int packed, void* stream);
/* De-duplication plan that uses the DataGenerator contract (2.FM/DataGenerator.py:76-88): column f only holds ids of
 * [col_lo[f], col_lo[f] + 2^key_bits) and columns are given in ascending range order, so duplicates occur only inside
 * a column and each column (B <= 16384 ids; max_key = largest id - col_lo over all columns, bits(max_key) +
 * ceil(log2 B) <= 32) is sorted on its own (chunk sort + rank merge + run detection).
 * perm [F,B], col_uid [F,B], col_seg [F,B+1], col_nu [F]; an id outside its column's range sets *bad_flag. */
size_t rec_colsort_workspace_bytes(int64_t B, int F);
int rec_colsort_plan_i64(const int64_t* const* cols_host, int F, int64_t B, int64_t V, const int64_t* col_lo,
                         int64_t max_key, int32_t* perm, int64_t* col_uid, int32_t* col_seg, int32_t* col_nu,
                         int* bad_flag, void* workspace, void* stream);
/* The same plan plus its inverse view: dloc [F,B] (int32) = for lookup (column f, example b) the index of its run of
 * equal ids inside the column (0 .. col_nu[f]-1), with the sign bit set unless the lookup is the FIRST member of the run
 * (smallest example index).  What the direct mode of the fused step needs to write value rows straight to their
 * de-duplicated slot (global slot = runs of the columns before + the run index). */
int rec_colsort_plan_dest_i64(const int64_t* const* cols_host, int F, int64_t B, int64_t V, const int64_t* col_lo,
                              int64_t max_key, int32_t* perm, int64_t* col_uid, int32_t* col_seg, int32_t* col_nu,
                              int32_t* dloc, int* bad_flag, void* workspace, void* stream);
/* Direct mode of the fused step: the plan of the batch (rec_colsort_plan_dest_i64) is complete before the launch.
 * rec_deepfm_fused_main_direct_f32 = rec_deepfm_fused_main_f32, except that the IndexedSlices value row of a lookup that
 * heads its run goes straight to g_embed_rows[slot] (only the other members of a run are written to vals);
 * rec_deepfm_fused_post_direct_f32 then adds the remaining members of runs longer than one in position order (sums
 * bit-identical to the plain path), writes uniq_ids / g_w_rows / n_uniq and the zero-padded tail, and reduces the dense
 * partials.  Replaces the 2.FM/ModelManager.py:176-179 IndexedSlices hand-over like the plain pair of calls does. */
int rec_deepfm_fused_main_direct_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                                     int64_t B, const float* bias, const float* K0, const float* b0, const float* K1,
                                     const float* b1, const float* K2, const float* b2, const float* label, float* gz,
                                     float* vals, float* prob, int* oob_flag, void* workspace, const int32_t* dloc,
                                     const int32_t* col_nu, float* g_embed_rows, void* stream);
int rec_deepfm_fused_post_direct_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                                     float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                     void* workspace, const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg,
                                     const int32_t* col_nu, int64_t* uniq_ids, float* g_embed_rows, float* g_w_rows,
                                     int64_t* n_uniq, void* stream);
/* Third form of the fused forward+backward kernel (csrc/deepfm_fused3.hip): same inputs, outputs and workspace layout as
 * rec_deepfm_fused_main_f32 / rec_deepfm_fused_main_direct_f32 -- the post launches above finish the step -- on a schedule
 * in which the two 16-example halves of a workgroup run one phase apart (the backward of one half on the matrix cores
 * while the rows of the other are still landing).  K0T [32, F*16] = the transpose of K0 [F*16, 32], kept by the caller
 * with rec_deepfm_k0t_f32 (layer 1 then reads its K0 operand as 16-byte pieces). */
int rec_deepfm_k0t_f32(const float* K0, int F, float* K0T, void* stream);
int rec_deepfm_fused3_main_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                               int64_t B, const float* bias, const float* K0, const float* K0T, const float* b0,
                               const float* K1, const float* b1, const float* K2, const float* b2, const float* label,
                               float* gz, float* vals, float* prob, int* oob_flag, void* workspace, void* stream);
int rec_deepfm_fused3_main_direct_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F,
                                      int64_t B, const float* bias, const float* K0, const float* K0T, const float* b0,
                                      const float* K1, const float* b1, const float* K2, const float* b2,
                                      const float* label, float* gz, float* vals, float* prob, int* oob_flag,
                                      void* workspace, const int32_t* dloc, const int32_t* col_nu, float* g_embed_rows,
                                      void* stream);
/* rec_deepfm_fused_post_direct_f32 + the lazy (touched-rows) Adam update of both tables applied to each row the moment its
 * gradient is final (optimizer in the backward; arithmetic of rec_adam_rows_f32; NOT Keras' dense-sweep
 * semantics of 2.FM/ModelManager.py:104,178-179 -- opt-in).  table: fused rows [V,32] = [embed 16 | w | pad] (ld = 32);
 * m_e, v_e [V,16]; m_w, v_w [V]; t: 1-based step. */
int rec_deepfm_fused_post_direct_adam_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                                          float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                          void* workspace, const int32_t* perm, const int64_t* col_uid,
                                          const int32_t* col_seg, const int32_t* col_nu, int64_t* uniq_ids,
                                          float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, float* table, int64_t ld,
                                          int64_t V, float* m_e, float* v_e, float* m_w, float* v_w, int64_t t, float lr,
                                          float b1, float b2, float eps, void* stream);
/* ... the same with the step size read from device memory (rec_adam_advance_f32 on the same stream) and explicit row
 * strides of the optimizer state: ld_state (floats) for m_e / v_e, ld_wstate for m_w / v_w -- 16 and 1 for dense arrays,
 * 32 and 32 for state packed beside the rows ([m 16 | v 16] as one 128-byte row, m_w / v_w in the padding of the fused
 * table row: a touched row then costs two line requests instead of five or six). */
int rec_deepfm_fused_post_direct_adam_dev_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                                              float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                              void* workspace, const int32_t* perm, const int64_t* col_uid,
                                              const int32_t* col_seg, const int32_t* col_nu, int64_t* uniq_ids,
                                              float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, float* table,
                                              int64_t ld, int64_t V, float* m_e, float* v_e, float* m_w, float* v_w,
                                              int64_t ld_state, int64_t ld_wstate, const float* lr_t_dev, float b1,
                                              float b2, float eps, int32_t* last, const int64_t* step_dev, void* stream);
/* Keras Adam evaluated lazily and exactly (2.FM/ModelManager.py:178-179: the sparse apply of IndexedSlices gradients is
 * a dense sweep over every row).  An untouched row's update at step j depends only on the row and lr_j, so rows may skip
 * the sweeps and replay them later with the sweep's own arithmetic: last [V] int32 = the step each row holds (zero at
 * the start).  rec_adam_keras_catchup_f32 brings the unique rows of a batch's plan (col_uid / col_nu of
 * rec_colsort_plan_dest_i64) up to step *step_dev BEFORE the batch reads them; the post launch above, given `last`,
 * applies the touched update of the step that follows and stamps the rows; rec_adam_keras_flush_f32 brings every row up
 * to date (before the parameters are read from outside).  Rounding is pinned in every Adam kernel of this library, so
 * the tables equal those of rec_adam_sparse_keras_pair_f32 bit for bit.  lr_table: rec_adam_lr_t_f32 of steps 1..n. */
int rec_adam_keras_catchup_f32(const int64_t* col_uid, const int32_t* col_nu, int64_t B, int F, float* table, int64_t ld,
                               int64_t V, float* m_e, float* v_e, int64_t ld_state, float* m_w, float* v_w,
                               int64_t ld_wstate, const int32_t* last, const int64_t* step_dev, const float* lr_table,
                               int64_t n_table, float b1, float b2, float eps, void* stream);
int rec_adam_keras_flush_f32(float* table, int64_t ld, int64_t V, float* m_e, float* v_e, int64_t ld_state, float* m_w,
                             float* v_w, int64_t ld_wstate, int32_t* last, const int64_t* step_dev, const float* lr_table,
                             int64_t n_table, float b1, float b2, float eps, void* stream);
/* segment sums of vals [B*F,16] (embed) and gz [B] (w) over that plan + compaction to the global ascending list:
 * uniq_ids [B*F], g_embed_rows [B*F,16], g_w_rows [B*F], n_uniq; the tail is padded like rec_dedup_plan_i64's. */
int rec_colseg_sum_f32(const float* vals, const float* gz, const int32_t* perm, const int64_t* col_uid,
                       const int32_t* col_seg, const int32_t* col_nu, int64_t B, int F, int64_t* uniq_ids,
                       float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, void* stream);

/* Same sums written as rows of 20 floats [embed 16 | w | 0 0 0] (g_rows [B*F,20]): ONE buffer to send to the shard
 * owners in the row-sharded step. */
int rec_colseg_sum_packed_f32(const float* vals, const float* gz, const int32_t* perm, const int64_t* col_uid,
                              const int32_t* col_seg, const int32_t* col_nu, int64_t B, int F, int64_t* uniq_ids,
                              float* g_rows, int64_t* n_uniq, void* stream);

/* ---- K8/K9  DIN ActivationUnit + masked sum pooling (5.DIN/CustomLayers.py:163-180, 256-282), factorised:
 *   pre[b,t,:] = c_b + k_t . Eff_b,  Eff_b = (W_k - W_d) + M_b,  M_b[i,o] = sum_j q_j W_o[i,j,o],
 *   c_b = q (W_q + W_d) + b1;  score = act(pre) . w2 + b2;  pooled[b,:] = sum_t mask[b,t] * score[b,t] * k_t.
 * D = E*C (C item features), H = hidden width (36 in the reference), W1 = the Dense(H) kernel [3D + D*D, H].
 * activation kinds for per-feature activations (alpha/mean/var are [H] device vectors; unused ones may be NULL): */
enum { REC_DACT_NONE = 0, REC_DACT_RELU = 1, REC_DACT_SIGMOID = 2, REC_DACT_TANH = 3,
       REC_DACT_DICE = 4,   /* Dice, BN(center=False, scale=False) with moving statistics (5.DIN/CustomLayers.py:183-196) */
       REC_DACT_PRELU = 5 };
/* W1, b1 -> Wcat [D, D*H + H] = [Wo_r | W_q + W_d], Wkd [D,H] = W_k - W_d, bext [D*H + H] = [0 | b1]; then
 * Mext = q . Wcat + bext is ONE rec_gemm_f32 per batch.  prepare_bwd maps (gWcat, gWkd) back onto gW1. */
int rec_din_prepare_f32(const float* W1, const float* b1, int D, int H, float* Wcat, float* Wkd, float* bext,
                        void* stream);
int rec_din_prepare_bwd_f32(const float* gWcat, const float* gWkd, int D, int H, float* gW1, void* stream);
/* series: int64 [B,T,C] (the tf.stack(axis=2) of the behaviour series, :258); key k_t = concat_r embed[series[b,t,r]].
 * mask: reference behaviour (mask_valid = 0) keeps PADDED positions (series[b,t,0] == padding_index, :256,277-278);
 * mask_valid = 1 is the intended form.  scores [B,T] are the raw (unmasked) scores; pooled [B,D]. */
int rec_din_attn_fwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series, int64_t B,
                         int T, const float* Mext, const float* Wkd, int H, int act, const float* alpha,
                         const float* mean, const float* var, const float* w2, const float* b2,
                         int64_t padding_index, int mask_valid, float* scores, float* pooled, int* oob_flag,
                         void* stream);
/* backward: gkeys [B,T,D] (IndexedSlices values of the series lookups), gMext [B, D*H+H], and per-example partials
 * gw2p [B,H], galphap [B,H], gb2p [B] (column sums of these are the parameter gradients). */
int rec_din_attn_bwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series, int64_t B,
                         int T, const float* Mext, const float* Wkd, int H, int act, const float* alpha,
                         const float* mean, const float* var, const float* w2, const float* b2,
                         int64_t padding_index, int mask_valid, const float* scores, const float* gpooled,
                         float* gkeys, float* gMext, float* gw2p, float* galphap, float* gb2p, void* stream);

/* ---- rows of DIN's final MLP (make_mlp_layer, 5.DIN/CustomLayers.py:142-160) */
/* y = act(x) on [M,N] with per-feature parameters; bwd also returns gy * dy/dalpha per element (column-sum it) */
int rec_feat_act_fwd_f32(int kind, const float* x, const float* alpha, const float* mean, const float* var, float* y,
                         int64_t M, int N, void* stream);
int rec_feat_act_bwd_f32(int kind, const float* x, const float* gy, const float* alpha, const float* mean,
                         const float* var, float* gx, float* ga_elem, int64_t M, int N, void* stream);
/* keras LayerNormalization (epsilon 1e-3): y = xhat*gamma + beta; saves xhat [M,N] and rstd [M] */
int rec_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, int64_t M, int N, float* y,
                          float* xhat, float* rstd, void* stream);
int rec_layernorm_bwd_f32(const float* gy, const float* xhat, const float* rstd, const float* gamma, int64_t M, int N,
                          float* gx, float* gg_elem, void* stream);
int rec_softmax_fwd_f32(const float* x, int64_t M, int N, float* y, void* stream);
int rec_softmax_bwd_f32(const float* y, const float* gy, int64_t M, int N, float* gx, void* stream);

/* ==== SURVEY.md section 8 row f4: sibling interaction layers that share the gather ======================= */

/* ---- PNN inner product: Embedding -> Flatten ++ IpnLayer (2.FM/CustomLayers.py:729-745, :755-792).
 * out[b, f*E + d] = table[X[b,f], d];  out[b, F*E + p(i,j)] = <e_i, e_j> for i < j in the row-major order of the
 * upper triangle (the order tf.boolean_mask keeps).  ld_out >= F*E + F*(F-1)/2: `out` IS the reference's
 * combined_vector.  F <= 255. */
int rec_emb_ipn_fwd_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* X, int64_t B, int F,
                        float* out, int64_t ld_out, int* oob_flag, void* stream);
/* IndexedSlices values of that lookup: vals[b*F+i, :] = g[b, i*E:(i+1)*E] + sum_{j != i} g[b, F*E + p(i,j)] * e_j,
 * with the rows e read back from the forward output `out`. */
int rec_emb_ipn_bwd_vals_f32(const float* out, int64_t ld_out, const float* g, int64_t ld_g, int64_t B, int F, int E,
                             float* vals, void* stream);

/* ---- NFM bi-interaction pooling (3.DCN/CustomLayers.py:499-501): out[b,d] = 0.5*((sum_f e_fd)^2 - sum_f e_fd^2),
 * written with row stride ld_out (so it can land in the leading columns of [second_order | X_cont]); sumvec [B,E]
 * = sum_f e_f is kept for the backward  vals[b*F+f, d] = g[b,d] * (sumvec[b,d] - e_fd). */
int rec_emb_bi_fwd_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* X, int64_t B, int F, float* out,
                       int64_t ld_out, float* sumvec, int* oob_flag, void* stream);
int rec_emb_bi_bwd_vals_f32(const float* table, int64_t V, int E, int64_t ld, const int64_t* X, int64_t B, int F,
                            const float* g, int64_t ld_g, const float* sumvec, float* vals, void* stream);

/* ---- SIM GSU inner-product attention + sum pooling (7.SIM/CustomLayers.py:88-96, 107-118).
 * series int64 [B,T,C]; key k_t = concat_r embed[series[b,t,r]] (D = C*E <= 256); valid[b,t] = series[b,t,0] !=
 * padding_index; scores[b,t] = valid * <q_b, k_t> (the masked scores); pooled[b,:] = sum_t scores[b,t] * k_t. */
int rec_ip_attn_fwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series, int64_t B, int T,
                        const float* q, int64_t ld_q, int64_t padding_index, float* scores, float* pooled,
                        int64_t ld_pooled, int* oob_flag, void* stream);
/* gkeys [B,T,D] = IndexedSlices values of the series lookups, gq [B,D] = gradient of the target vector. */
int rec_ip_attn_bwd_f32(const float* embed, int64_t ld, int64_t V, int E, int C, const int64_t* series, int64_t B, int T,
                        const float* q, int64_t ld_q, int64_t padding_index, const float* scores, const float* gpooled,
                        int64_t ld_gpooled, float* gkeys, float* gq, void* stream);

/* ---- FFM, field-aware second order (FieldAwareInteractionLayer 2.FM/CustomLayers.py:428-462; FFMRankingLayer.call
 * :398-425 computes the same numbers from F separate tables).  v [V, F, E] with row stride ld_v >= F*E floats:
 * v[id, c, :] is the vector id uses against field c (table c of the loop form, row id).
 *   z[b] = bias + sum_a w[X[b,a]] + sum_{a<c} < v[X[b,a], c, :], v[X[b,c], a, :] >,  prob = sigmoid(z). */
int rec_ffm_fwd_f32(const float* v, int64_t ld_v, const float* w, int64_t ld_w, const float* bias, int64_t V, int E,
                    const int64_t* X, int64_t B, int F, float* z, float* prob, int* oob_flag, void* stream);
/* de-duplicated gradient rows of v on the plan of rec_dedup_plan_i64 over X (n = B*F):
 *   g_rows[u, c, :] = sum over the lookups (b,a) of unique id u of gz[b] * v[X[b,c], a, :]  (c != a);  rows >= n_uniq
 * are zero.  g_rows [B*F, F*E]. */
int rec_ffm_bwd_rows_f32(const float* v, int64_t ld_v, int64_t V, int E, const int64_t* X, int64_t B, int F,
                         const float* gz, const int32_t* perm, const int32_t* seg_start, const int64_t* n_uniq,
                         float* g_rows, void* stream);

/* ---- tf.keras.layers.BatchNormalization on [B,N] (3.DCN/CustomLayers.py:466,504; 2.FM/CustomLayers.py:69,78-79).
 * training != 0: batch mean / biased batch variance, moving statistics updated in place with `momentum`;
 * training == 0: moving statistics.  xhat [B,N] and rstd [N] are saved for the backward (may be NULL at inference).
 * gamma / beta may be NULL (scale / center off).  workspace: rec_batchnorm_workspace_bytes(B, N) device bytes. */
size_t rec_batchnorm_workspace_bytes(int64_t B, int N);
int rec_batchnorm_fwd_f32(const float* x, int64_t ld_x, int64_t B, int N, const float* gamma, const float* beta,
                          float eps, float momentum, int training, float* moving_mean, float* moving_var, float* y,
                          float* xhat, float* rstd, void* workspace, void* stream);
int rec_batchnorm_bwd_f32(const float* g, const float* xhat, const float* rstd, int64_t B, int N, const float* gamma,
                          int training, float* gx, float* ggamma, float* gbeta, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355REC_H */
