// TORCH_LIBRARY registration of the hot operators ("each is mirrored as a TORCH_LIBRARY(mi355rec, ...) op").
//
// Host-only translation unit (g++): every operator here allocates its outputs with ATen, takes the current HIP stream
// of the tensors' device and calls the SAME C ABI entry point of libmi355rec.so that include/mi355rec.h declares and the
// ctypes binding (_lib.py) uses -- no kernel and no arithmetic lives in this file.  What it buys over ctypes is the call
// path: one dispatcher hop instead of marshalling ~20 Python objects per call, which is what the eager (non-graphed)
// steps of the Layer mirror were paying ~5-10 us per operator for.  The Keras `Layer.__call__` boundary of the
// reference (2.FM/ModelManager.py:87-96) stays in layers.py; autograd stays in functional.py (its Functions call these
// operators in their forward / backward).  Dispatch key CUDA = the ROCm device of PyTorch-ROCm.
#include <ATen/ATen.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "mi355rec.h"

namespace {

using at::Tensor;
using c10::optional;

inline void* stream_of(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

inline void check(int status, const char* what) {
  if (status == REC_OK) return;
  TORCH_CHECK_VALUE(status != REC_E_ARG, what, ": invalid argument");
  TORCH_CHECK_NOT_IMPLEMENTED(status != REC_E_UNSUPPORTED, what, ": unsupported configuration");
  TORCH_CHECK(status != REC_E_WORKSPACE, what, ": workspace too small");
  TORCH_CHECK(false, what, ": hipError_t ", status);
}

inline const float* fp(const Tensor& t) { return t.data_ptr<float>(); }
inline float* fpm(const Tensor& t) { return t.data_ptr<float>(); }
inline const float* ofp(const optional<Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
inline int* oip(const optional<Tensor>& t) { return t.has_value() ? t->data_ptr<int>() : nullptr; }

Tensor index_pack(at::TensorList cols, const optional<Tensor>& out_, int64_t col0) {
  const int F = (int)cols.size();
  TORCH_CHECK_VALUE(F > 0, "index_pack: no columns");
  const int64_t rows = cols[0].numel();
  std::vector<const int64_t*> ptrs(F);
  for (int f = 0; f < F; ++f) {
    TORCH_CHECK_VALUE(cols[f].numel() == rows, "index columns differ in length");
    ptrs[f] = cols[f].data_ptr<int64_t>();
  }
  Tensor out = out_.has_value() ? *out_ : at::empty({rows, F}, cols[0].options());
  check(rec_index_pack_i64(ptrs.data(), F, rows, out.data_ptr<int64_t>(), out.size(1), col0, stream_of(out)),
        "rec_index_pack_i64");
  return out;
}

Tensor emb_gather(const Tensor& table, const Tensor& idx, const optional<Tensor>& oob) {
  const int64_t V = table.size(0), E = table.size(1);
  auto shape = idx.sizes().vec();
  shape.push_back(E);
  Tensor out = at::empty(shape, table.options());
  check(rec_emb_gather_f32(fp(table), V, (int)E, table.stride(0), idx.data_ptr<int64_t>(), idx.numel(), fpm(out),
                           oip(oob), stream_of(table)),
        "rec_emb_gather_f32");
  return out;
}

std::tuple<Tensor, optional<Tensor>, optional<Tensor>, optional<Tensor>> emb_fm_fwd(
    const Tensor& embed, const Tensor& w, const Tensor& bias, const Tensor& X, bool want_prob, bool want_rows,
    bool want_sum, const optional<Tensor>& oob) {
  const int64_t V = embed.size(0), E = embed.size(1), B = X.size(0), F = X.size(1);
  Tensor z = at::empty({B}, embed.options());
  optional<Tensor> prob, rows, S;
  if (want_prob) prob = at::empty({B}, embed.options());
  if (want_rows) rows = at::empty({B, F, E}, embed.options());
  if (want_sum) S = at::empty({B, E}, embed.options());
  check(rec_emb_fm_fwd_f32(fp(embed), embed.stride(0), fp(w), w.stride(0), fp(bias), V, (int)E, X.data_ptr<int64_t>(), B,
                           (int)F, fpm(z), prob ? fpm(*prob) : nullptr, rows ? fpm(*rows) : nullptr,
                           S ? fpm(*S) : nullptr, oip(oob), stream_of(embed)),
        "rec_emb_fm_fwd_f32");
  return {z, prob, rows, S};
}

Tensor emb_fm_bwd_vals(const Tensor& embed, const Tensor& X, const Tensor& gz, const Tensor& sumvec,
                       const optional<Tensor>& rows, const optional<Tensor>& extra) {
  const int64_t V = embed.size(0), E = embed.size(1), B = X.size(0), F = X.size(1);
  Tensor out = at::empty({B * F, E}, embed.options());
  check(rec_emb_fm_bwd_vals_f32(fp(embed), embed.stride(0), V, (int)E, X.data_ptr<int64_t>(), B, (int)F, fp(gz),
                                fp(sumvec), ofp(rows), ofp(extra), fpm(out), stream_of(embed)),
        "rec_emb_fm_bwd_vals_f32");
  return out;
}

Tensor gemm(const Tensor& A, const Tensor& B, bool transA, bool transB, int64_t epi, const optional<Tensor>& bias,
            const optional<Tensor>& e0, const optional<Tensor>& e1, int64_t split_k, const optional<Tensor>& out_,
            const optional<Tensor>& aux) {
  const int64_t M = transA ? A.size(1) : A.size(0), K = transA ? A.size(0) : A.size(1);
  const int64_t K2 = transB ? B.size(1) : B.size(0), N = transB ? B.size(0) : B.size(1);
  TORCH_CHECK_VALUE(K == K2, "gemm inner dimensions differ: ", K, " vs ", K2);
  Tensor out = out_.has_value() ? *out_ : at::empty({M, N}, A.options());
  Tensor ws;
  if (split_k > 1) ws = at::empty({split_k, M, N}, A.options());
  check(rec_gemm_f32(transA, transB, M, N, K, fp(A), A.stride(0), fp(B), B.stride(0), fpm(out), out.stride(0), (int)epi,
                     ofp(bias), ofp(e0), e0.has_value() ? e0->stride(0) : 0, ofp(e1), e1.has_value() ? e1->stride(0) : 0,
                     (int)split_k, split_k > 1 ? fpm(ws) : nullptr, aux.has_value() ? fpm(*aux) : nullptr, stream_of(A)),
        "rec_gemm_f32");
  return out;
}

Tensor act_fwd(int64_t act, const Tensor& x, const optional<Tensor>& x2) {
  Tensor y = at::empty_like(x);
  check(rec_act_fwd_f32((int)act, fp(x), ofp(x2), fpm(y), x.numel(), stream_of(x)), "rec_act_fwd_f32");
  return y;
}

Tensor act_bwd(int64_t act, const Tensor& post, const Tensor& dpost) {
  Tensor out = at::empty_like(dpost);
  check(rec_act_bwd_f32((int)act, fp(post), fp(dpost), fpm(out), post.numel(), stream_of(post)), "rec_act_bwd_f32");
  return out;
}

// Arrival counters of rec_colsum_fused_f32: one zero-initialised int32 buffer per device, handed out as a ring (a call
// takes the next ceil(N/32) slots, so calls that overlap on different streams do not meet; every call leaves its slots
// zero).  Allocated by the first call outside a stream capture (its memory must not belong to a graph's pool); until
// then, and for very wide matrices, the two-launch form is used.
constexpr int64_t COLSUM_RING = 1 << 16;
static Tensor* g_colsum_ring[64] = {nullptr};
static int64_t g_colsum_pos[64] = {0};

Tensor colsum(const Tensor& X, const optional<Tensor>& out_) {
  const int64_t M = X.size(0), N = X.size(1);
  Tensor out = out_.has_value() ? *out_ : at::empty({N}, X.options());
  Tensor ws = at::empty({(int64_t)(rec_colsum_workspace_bytes(M, N) / 4)}, X.options());
  const int dev = X.device().index();
  const int64_t nblk = (N + 31) / 32;
  if (dev >= 0 && dev < 64 && !g_colsum_ring[dev] &&
      c10::hip::currentStreamCaptureStatusMayInitCtx() == c10::hip::CaptureStatus::None)
    g_colsum_ring[dev] = new Tensor(at::zeros({COLSUM_RING}, X.options().dtype(at::kInt)));   // never freed: outlives
  if (dev >= 0 && dev < 64 && g_colsum_ring[dev] && nblk <= COLSUM_RING / 4 && M > 1024) {     // the HIP context
    if (g_colsum_pos[dev] + nblk > COLSUM_RING) g_colsum_pos[dev] = 0;
    int* cnt = g_colsum_ring[dev]->data_ptr<int>() + g_colsum_pos[dev];
    g_colsum_pos[dev] += nblk;
    check(rec_colsum_fused_f32(fp(X), M, N, X.stride(0), fpm(out), fpm(ws), cnt, stream_of(X)), "rec_colsum_fused_f32");
    return out;
  }
  check(rec_colsum_f32(fp(X), M, N, X.stride(0), fpm(out), fpm(ws), stream_of(X)), "rec_colsum_f32");
  return out;
}

std::tuple<Tensor, optional<Tensor>, optional<Tensor>> bce_fwd_bwd(const Tensor& y, const Tensor& p, bool want_dp,
                                                                   bool want_dz) {
  const int64_t n = p.numel();
  Tensor loss = at::empty({1}, p.options());
  optional<Tensor> dp, dz;
  if (want_dp) dp = at::empty({n}, p.options());
  if (want_dz) dz = at::empty({n}, p.options());
  check(rec_bce_fwd_bwd_f32(fp(y), fp(p), n, fpm(loss), dp ? fpm(*dp) : nullptr, dz ? fpm(*dz) : nullptr, stream_of(p)),
        "rec_bce_fwd_bwd_f32");
  return {loss, dp, dz};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> dedup_plan(const Tensor& ids, int64_t V) {
  const int64_t n = ids.numel();
  auto i64 = ids.options();
  auto i32 = ids.options().dtype(at::kInt);
  Tensor uniq = at::empty({n > 0 ? n : 1}, i64), seg = at::empty({n + 1}, i32), perm = at::empty({n > 0 ? n : 1}, i32),
         nu = at::empty({1}, i64);
  const size_t nbytes = rec_dedup_workspace_bytes(n);
  Tensor ws = at::empty({(int64_t)nbytes}, ids.options().dtype(at::kByte));
  check(rec_dedup_plan_i64(ids.data_ptr<int64_t>(), n, V, uniq.data_ptr<int64_t>(), seg.data_ptr<int>(),
                           perm.data_ptr<int>(), nu.data_ptr<int64_t>(), ws.data_ptr(), nbytes, stream_of(ids)),
        "rec_dedup_plan_i64");
  return {uniq, seg, perm, nu};
}

Tensor segment_sum(const Tensor& vals, int64_t E, const Tensor& perm, const Tensor& seg, int64_t n, int64_t row_div) {
  Tensor out = at::empty({n > 0 ? n : 1, E}, vals.options());
  Tensor ws = at::empty({(int64_t)(rec_segment_sum_workspace_bytes(n, (int)E) / 4)}, vals.options());
  check(rec_segment_sum_f32(fp(vals), (int)E, perm.data_ptr<int>(), seg.data_ptr<int>(), n, (int)row_div, fpm(out),
                            fpm(ws), stream_of(vals)),
        "rec_segment_sum_f32");
  return out;
}

Tensor cosine_fwd(const Tensor& u, const Tensor& i) {
  Tensor out = at::empty({u.size(0)}, u.options());
  check(rec_cosine_fwd_f32(fp(u), fp(i), u.size(0), (int)u.size(1), fpm(out), stream_of(u)), "rec_cosine_fwd_f32");
  return out;
}

std::tuple<Tensor, Tensor> cosine_bwd(const Tensor& u, const Tensor& i, const Tensor& g) {
  Tensor gu = at::empty_like(u), gi = at::empty_like(i);
  check(rec_cosine_bwd_f32(fp(u), fp(i), u.size(0), (int)u.size(1), fp(g), fpm(gu), fpm(gi), stream_of(u)),
        "rec_cosine_bwd_f32");
  return {gu, gi};
}

Tensor crossnet_mat_bwd_elem(const Tensor& g, const Tensor& x0, const Tensor& u, Tensor gx0, bool accumulate) {
  Tensor h = at::empty_like(g);
  check(rec_crossnet_mat_bwd_elem_f32(fp(g), fp(x0), fp(u), fpm(h), fpm(gx0), accumulate, g.numel(), stream_of(g)),
        "rec_crossnet_mat_bwd_elem_f32");
  return h;
}

Tensor axpby(double a, const Tensor& x, double b, Tensor y) {
  check(rec_axpby_f32((float)a, fp(x), (float)b, fpm(y), x.numel(), stream_of(x)), "rec_axpby_f32");
  return y;
}

}  // namespace

TORCH_LIBRARY(mi355rec, m) {
  m.def("index_pack(Tensor[] cols, Tensor(a!)? out, int col0) -> Tensor");
  m.def("emb_gather(Tensor table, Tensor idx, Tensor(a!)? oob) -> Tensor");
  m.def("emb_fm_fwd(Tensor embed, Tensor w, Tensor bias, Tensor X, bool want_prob, bool want_rows, bool want_sum, "
        "Tensor(a!)? oob) -> (Tensor, Tensor?, Tensor?, Tensor?)");
  m.def("emb_fm_bwd_vals(Tensor embed, Tensor X, Tensor gz, Tensor sumvec, Tensor? rows, Tensor? extra) -> Tensor");
  m.def("gemm(Tensor A, Tensor B, bool transA, bool transB, int epi, Tensor? bias, Tensor? e0, Tensor? e1, int split_k, "
        "Tensor(a!)? out, Tensor(b!)? aux) -> Tensor");
  m.def("act_fwd(int act, Tensor x, Tensor? x2) -> Tensor");
  m.def("act_bwd(int act, Tensor post, Tensor dpost) -> Tensor");
  m.def("colsum(Tensor X, Tensor(a!)? out) -> Tensor");
  m.def("bce_fwd_bwd(Tensor y, Tensor p, bool want_dp, bool want_dz) -> (Tensor, Tensor?, Tensor?)");
  m.def("dedup_plan(Tensor ids, int V) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("segment_sum(Tensor vals, int E, Tensor perm, Tensor seg, int n, int row_div) -> Tensor");
  m.def("cosine_fwd(Tensor u, Tensor i) -> Tensor");
  m.def("cosine_bwd(Tensor u, Tensor i, Tensor g) -> (Tensor, Tensor)");
  m.def("crossnet_mat_bwd_elem(Tensor g, Tensor x0, Tensor u, Tensor(a!) gx0, bool accumulate) -> Tensor");
  m.def("axpby(float a, Tensor x, float b, Tensor(a!) y) -> Tensor(a!)");
}

TORCH_LIBRARY_IMPL(mi355rec, CUDA, m) {
  m.impl("index_pack", &index_pack);
  m.impl("emb_gather", &emb_gather);
  m.impl("emb_fm_fwd", &emb_fm_fwd);
  m.impl("emb_fm_bwd_vals", &emb_fm_bwd_vals);
  m.impl("gemm", &gemm);
  m.impl("act_fwd", &act_fwd);
  m.impl("act_bwd", &act_bwd);
  m.impl("colsum", &colsum);
  m.impl("bce_fwd_bwd", &bce_fwd_bwd);
  m.impl("dedup_plan", &dedup_plan);
  m.impl("segment_sum", &segment_sum);
  m.impl("cosine_fwd", &cosine_fwd);
  m.impl("cosine_bwd", &cosine_bwd);
  m.impl("crossnet_mat_bwd_elem", &crossnet_mat_bwd_elem);
  m.impl("axpby", &axpby);
}
