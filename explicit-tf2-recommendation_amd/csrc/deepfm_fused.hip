// Fused DeepFM train step (2.FM/CustomLayers.py:279-308 under 2.FM/ModelManager.py:171-177) for the reference's
// default head (embedding_dims = 16, mlp_dims = [32, 8]) on the fused [embed(16) | w | pad] 128-byte row layout.
//
// The generic path runs ~35 launch-bound kernels per step (profiles/r01_v1_*): at batch 8192 every small kernel costs
// ~5 us while the whole gather is ~4 us of HBM time.  Here a step is two launches on the main stream plus the sort of
// upcoming batches' ids on a second one:
//
//   deepfm_fwd_bwd_kernel   one 8-wave workgroup per 32 examples; wave w owns fields w, w+8, ... from the id load to
//                           the last store: gather of the 128-B rows, layer 1 (416->32) on v_mfma_f32_32x32x2_f32
//                           UNDER the gather, the 32->8->1 head + sigmoid + Keras BCE + the way back inside 16-lane
//                           groups (DPP reductions), dX = dpre1 . K0^T and the workgroup's dK0 partial on
//                           v_mfma_f32_16x16x4_f32 tiles per field.  Direct mode: the IndexedSlices value row of a
//                           lookup that heads its run of equal ids goes straight to the run's slot of the
//                           de-duplicated gradient (the plan exists before the launch).  Details above the kernel.
//   deepfm_post_direct_kernel  ONE launch, two jobs side by side: the fixed-order sum of the per-workgroup partials
//                           (dK0, dK1, biases, loss) over ~210 workgroups, and what is left of the segment sums --
//                           runs of more than one lookup, unique ids, first-order rows, zero-padded tail; optionally
//                           the lazy Adam update of every finished row.  (deepfm_post_kernel / deepfm_reduce_kernel /
//                           colseg_sum_kernel: the plain, non-direct forms used by the row-sharded step.)
//   colsort_onewg_kernel    de-duplication plan: the DataGenerator contract (2.FM/DataGenerator.py:76-88) gives every
//                           feature column its own contiguous id range, so duplicates only occur inside a column:
//                           one 1024-thread workgroup sorts a column (B <= 16384 ids) in LDS as 32-bit
//                           (key << PB | position) words and finds the runs.  Depends on ids only: the engine runs it
//                           ahead, on a second stream, for the batches of the next call (up to 4 per launch).
//
// Everything is deterministic (no float atomics): per-workgroup partials + fixed-order reductions, stable sort keys.
#include "common.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EX = 32;        // examples per workgroup
constexpr int E16 = 16;       // embedding dims
constexpr int LD = 32;        // fused row stride (floats)
constexpr int U1 = 32, U2 = 8;
constexpr int SMALL = 320;    // floats of small partials per workgroup
constexpr int NWV = 8;        // waves per workgroup (512 threads: two waves per SIMD, one workgroup per CU)
constexpr int MAXFW = 4;      // fields per wave (F <= 28 over 8 waves)
constexpr int HSP = 34;       // row stride of a wave's partial layer-1 tile (even: float2 reads)
constexpr int HS1 = 36;       // row stride of H1s / DP1 (16-byte aligned rows: ds_read_b128 operand fetches)

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
  // direct mode (the batch's plan exists before the launch): a lookup that heads its run of equal ids writes its value
  // row straight to the run's slot of the de-duplicated gradient; only the other members of a run go through `vals`
  const int32_t* dloc;        // [F,B] column-local run index of lookup (f,b); sign bit set = not the head of its run
  const int32_t* col_nu;      // [F]   runs per column
  float* g_embed;             // [B*F,16] de-duplicated row sums
  float* g_w;                 // [B*F]    ... of the first-order table
  int64_t* uniq_ids;          // [B*F]    the id of every slot
  int ld;                     // row stride of `table` in floats; anything but 32 only in the plan-after form (the rows
                              // a sharded step received: [embed 16 | w | pad 3] = 80 bytes on the wire, not 128)
#ifdef REC_FUSED_STAMPS
  unsigned long long* stamps; // diagnostic build only: [nwg][8 waves][12] s_memrealtime ticks (10 ns)
#endif
};

#ifdef REC_FUSED_STAMPS
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    if (lane == 0) a.stamps[((int64_t)blockIdx.x * NWV + wave) * 12 + (k)] = wall_clock64();       \
  } while (0)
// diagnostic only: drain the wave's memory queues first, so that the stamp reads "everything issued so far is back"
#define STAMP_DRAINED(k)                                              \
  do {                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       \
    STAMP(k);                                                         \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMP_DRAINED(k) do {} while (0)
#endif

// row stride of the gathered-rows tile: 16F + 4.  (16F + 4)/4 is odd, so the 16-byte operand fetches of 16 lanes with
// distinct rows (mod 16) fall on distinct bank quads, and rows 4 apart sit 16 banks apart (the 32-lane halves of the
// transposed reads in the backward phases are conflict-free)
__device__ __forceinline__ int xs_of(int F) { return F * E16 + 4; }

// LDS carve (floats); everything in ONE dynamic array
struct Carve {
  int XT, PT, WP, H1s, DP1, Ss, h2s, dp2s, dzs, lss, K1s, b0s, b1s, K2s, total;
};
__host__ __device__ inline Carve carve_of(int F) {
  Carve c;
  int o = 0;
  c.XT = o; o += EX * (F * E16 + 4);
  c.PT = o; o += NWV * EX * HSP;
  c.WP = o; o += NWV * EX;
  c.H1s = o; o += EX * HS1;
  c.DP1 = o; o += EX * HS1;
  c.Ss = o; o += EX * E16;
  c.h2s = o; o += EX * U2;
  c.dp2s = o; o += EX * U2;
  c.dzs = o; o += EX;
  c.lss = o; o += EX;
  c.K1s = o; o += U1 * U2;
  c.b0s = o; o += U1;
  c.b1s = o; o += U2;
  c.K2s = o; o += U2;
  c.total = o;
  return c;
}

// One workgroup = 32 examples, 8 waves.  Wave w OWNS fields w, w+8, w+16, w+24 from the first id load to the last store:
//
//   A  (no barrier inside) the wave loads its fields' ids, issues all its row loads (16-byte pieces, 8 lanes per 128-byte
//      fused row) and its slice of K0 (straight from L2 into MFMA B fragments), then field by field as the rows arrive:
//      rows -> LDS tile XT, first-order weights summed in registers, the A fragments of that field read back
//      (lane = example: two 16-byte LDS reads), FM sums, 8 x v_mfma_f32_32x32x2_f32 into the wave's partial layer-1 tile.
//      The layer-1 product and the FM sums therefore run UNDER the gather instead of after it.
//   B  the 8 partial tiles / FM partials meet in LDS (barrier), 512 threads finish layer 1, the 32->8->1 head, sigmoid,
//      Keras BCE and their backward (two barriers).
//   C  (no barrier inside) per owned field, on v_mfma_f32_16x16x4_f32 tiles (one field = 16 columns, so the 26 fields
//      spread 7/7/6/6 over the four SIMDs): dX = dpre1 . K0_f^T with the IndexedSlices values dz*(S - x) + dX stored as
//      64-byte rows, and the workgroup's partial of dK0_f = X_f^T . dpre1.  K0_f comes from L2 as two 16-byte loads per
//      lane, the operands made of dpre1 stay in registers for all fields.
// K0 never sits in LDS (the 53 KB copy per workgroup of round 1 is gone); LDS holds the rows once (54 KB) + ~60 KB of
// exchange buffers.
template <bool DIRECT>
__global__ __launch_bounds__(512) void deepfm_fwd_bwd_kernel(Cols cols, FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int F = a.F, D = F * E16, XS = xs_of(F);
  const Carve cv = carve_of(F);
  float* XT = lds + cv.XT;               // [EX][XS]
  float* PT = lds + cv.PT;               // [8][EX][HSP]   partial layer-1 tiles
  float* WP = lds + cv.WP;               // [8][EX]        partial first-order sums
  float* H1s = lds + cv.H1s;             // [EX][HS1]      relu(h1)
  float* DP1 = lds + cv.DP1;             // [EX][HS1]      d pre-activation of layer 1
  float* Ss = lds + cv.Ss;               // [EX][16]
  float* h2s = lds + cv.h2s;             // [EX][8]
  float* dp2s = lds + cv.dp2s;           // [EX][8]
  float* dzs = lds + cv.dzs;
  float* lss = lds + cv.lss;
  float* K1s = lds + cv.K1s;             // [32][8]
  float* b0s = lds + cv.b0s;
  float* b1s = lds + cv.b1s;
  float* K2s = lds + cv.K2s;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // scalar: field ownership is wave-uniform
  const int64_t ex0 = (int64_t)blockIdx.x * EX;
  const int n_ex = (a.B - ex0 < EX) ? (int)(a.B - ex0) : EX;
  const int lo = lane & 31, hi = lane >> 5;

  // scalars and labels of phase B: loaded now, their HBM latency hides behind phase A
  const float bias_r = a.bias[0], b2_r = a.b2[0];
  const float inv_B = 1.f / (float)a.B;
  const float label_r = ((tid >> 4) < n_ex) ? a.label[ex0 + (tid >> 4)] : 0.f;

  // ================================================ phase A ======================================================
  STAMP(0);
  {
    // ids first (they head the longest dependent chain of the kernel): lane = (piece c of the row, example e' of the
    // group of 8); slot g = examples 8g + e'
    const int c = lane & 7, ep = lane >> 3;
    // every id load is issued unconditionally (clamped address) and only then checked: a load inside a guarded
    // block would be waited for before the next one is issued
    int64_t idr[MAXFW][4];
#pragma unroll
    for (int i = 0; i < MAXFW; ++i) {
      const int f = wave + NWV * i;
      const int64_t* col = cols.p[f < F ? f : F - 1];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int e = 8 * g + ep;
        idr[i][g] = col[ex0 + (e < n_ex ? e : n_ex - 1)];
      }
    }
    STAMP_DRAINED(8);                           // (diagnostic builds) the ids are here
    // rows: every load of the wave is issued before anything is consumed; an invalid lookup reads row 0 and is
    // zeroed afterwards, lanes c > 4 repeat the address of lane 4 (same request)
    bool ok[MAXFW][4];
    bool bad = false;
    float4 v[MAXFW][4];
    const int cc = c < 4 ? c : 4;
#pragma unroll
    for (int i = 0; i < MAXFW; ++i) {
      const int f = wave + NWV * i;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bool live = f < F && 8 * g + ep < n_ex;
        const bool inr = (uint64_t)idr[i][g] < (uint64_t)a.V;
        bad |= live && !inr;
        ok[i][g] = live && inr;
        const int64_t row = ok[i][g] ? idr[i][g] : 0;
        v[i][g] = *reinterpret_cast<const float4*>(a.table + row * (DIRECT ? LD : a.ld) + 4 * cc);
      }
    }
    // K0 slice of the owned fields as B fragments of the 32x32x2 product: step s of field i multiplies the row's dims
    // 8*hi + s (the k slot of a lane is its half), so B[k slot hi][unit lo] = K0[f*16 + 8*hi + s][lo]: two 128-byte
    // segments per wave instruction, L2-resident
    float kb[MAXFW][8];
#pragma unroll
    for (int i = 0; i < MAXFW; ++i) {
      const int f = wave + NWV * i;
      if (f < F) {
#pragma unroll
        for (int s = 0; s < 8; ++s) kb[i][s] = a.K0[(f * E16 + 8 * hi + s) * U1 + lo];
      }
    }
    if (bad && a.oob) *a.oob = 1;
    STAMP(1);                                   // ids have arrived, every row load is issued
    // small dense operands of phase B (first read after barrier 1)
    if (tid < U1 * U2) K1s[tid] = a.K1[tid];
    if (tid < U1) b0s[tid] = a.b0[tid];
    if (tid < U2) { b1s[tid] = a.b1[tid]; K2s[tid] = a.K2[tid]; }
#pragma unroll
    for (int i = 0; i < MAXFW; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (!ok[i][g]) v[i][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float wacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXFW; ++i) {
      const int f = wave + NWV * i;
      if (f < F) {                                                    // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (c < 4) *reinterpret_cast<float4*>(XT + (8 * g + ep) * XS + f * E16 + 4 * c) = v[i][g];
          else if (c == 4) wacc[g] += v[i][g].x;
        }
        // the same wave reads the field back example-major: LDS operations of one wave complete in program order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4 x0 = *reinterpret_cast<const float4*>(XT + lo * XS + f * E16 + 8 * hi);
        const float4 x1 = *reinterpret_cast<const float4*>(XT + lo * XS + f * E16 + 8 * hi + 4);
        const float xa[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], kb[i][s], acc, 0, 0, 0);
      }
    }
    float* pt = PT + wave * (EX * HSP);
#pragma unroll
    for (int r = 0; r < 16; ++r) pt[((r & 3) + 8 * (r >> 2) + 4 * hi) * HSP + lo] = acc[r];
    if (c == 4) {
#pragma unroll
      for (int g = 0; g < 4; ++g) WP[wave * EX + 8 * g + ep] = wacc[g];
    }
  }
  STAMP(2);                                     // rows consumed, partial tile written
  __syncthreads();
  STAMP(3);

  // ================================================ phase B ======================================================
  // 16 lanes per example; lane g16 owns units 2*g16, 2*g16+1 of layer 1.  Everything between the end of layer 1 and
  // dpre1 stays inside the 16-lane group (xor butterflies: every lane ends up with bit-identical sums), so the whole
  // head -- 32->8->1, sigmoid, Keras BCE and the way back -- needs no barrier of its own.
  const int e16 = tid >> 4, g16 = tid & 15;
  // first field's K0 fragments of phase C: requested now, back long before barrier 2
  const int cj = lane & 15, cq = lane >> 4;
  float4 kt0, kt1;
  {
    const float* kp = a.K0 + ((wave < F ? wave : 0) * E16 + cj) * U1 + 8 * cq;
    kt0 = *reinterpret_cast<const float4*>(kp);
    kt1 = *reinterpret_cast<const float4*>(kp + 4);
  }
  // direct mode: slot of every (example, owned field) value row.  Column-local run index + runs in the columns before
  int dl[MAXFW][2];
  if (DIRECT) {
    int run = 0;                                             // scalar prefix over col_nu (F <= 28 scalar loads)
    int before[MAXFW] = {0, 0, 0, 0};
    for (int q = 0; q < F; ++q) {
#pragma unroll
      for (int i = 0; i < MAXFW; ++i)
        if (q == wave + NWV * i) before[i] = run;
      run += a.col_nu[q];
    }
#pragma unroll
    for (int i = 0; i < MAXFW; ++i) {
      const int f = wave + NWV * i;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int e = 16 * m + cj;
        const bool live = f < F && e < n_ex;
        int d = live ? a.dloc[(int64_t)f * a.B + ex0 + e] : -1;
        dl[i][m] = d >= 0 ? d + before[i] : d;               // heads: global slot; others keep the sign bit
      }
    }
  }
  {
    // layer 1: the 8 partial tiles are added in wave order (fixed), + bias, relu
    float2 h = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const float2 p = *reinterpret_cast<const float2*>(PT + w * (EX * HSP) + e16 * HSP + 2 * g16);
      h.x += p.x; h.y += p.y;
    }
    const float h1a = fmaxf(h.x + b0s[2 * g16], 0.f), h1b = fmaxf(h.y + b0s[2 * g16 + 1], 0.f);
    H1s[e16 * HS1 + 2 * g16] = h1a;                       // kept for the dK1 partial of phase C
    H1s[e16 * HS1 + 2 * g16 + 1] = h1b;
    // FM: lane = dim d of the example: S_d and the sum of squares of that dim over the fields, straight from the rows
    // in LDS (the per-wave partial sums of round 2's first version cost 18 KB of LDS: with them the 37-KB sort workgroup
    // of the second stream could not share the CU and delayed every fused kernel it overlapped by a whole round)
    float sd = 0.f, sqd = 0.f;
    {
      const float* xr = XT + e16 * XS + g16;
      for (int f = 0; f < F; ++f) {
        const float x = xr[f * E16];
        sd += x;
        sqd += x * x;
      }
    }
    Ss[e16 * E16 + g16] = sd;
    float t = sd * sd - sqd;
    float fo = g16 < NWV ? WP[g16 * EX + e16] : 0.f;
    // layer 2 (32 -> 8): this lane's two units times K1, then the same butterfly
    float h2[U2];
    {
      const float4 ka0 = *reinterpret_cast<const float4*>(K1s + (2 * g16) * U2);
      const float4 ka1 = *reinterpret_cast<const float4*>(K1s + (2 * g16) * U2 + 4);
      const float4 kb0 = *reinterpret_cast<const float4*>(K1s + (2 * g16 + 1) * U2);
      const float4 kb1 = *reinterpret_cast<const float4*>(K1s + (2 * g16 + 1) * U2 + 4);
      h2[0] = h1a * ka0.x + h1b * kb0.x; h2[1] = h1a * ka0.y + h1b * kb0.y;
      h2[2] = h1a * ka0.z + h1b * kb0.z; h2[3] = h1a * ka0.w + h1b * kb0.w;
      h2[4] = h1a * ka1.x + h1b * kb1.x; h2[5] = h1a * ka1.y + h1b * kb1.y;
      h2[6] = h1a * ka1.z + h1b * kb1.z; h2[7] = h1a * ka1.w + h1b * kb1.w;
      t = row16_allsum(t);
      fo = row16_allsum(fo);
#pragma unroll
      for (int u = 0; u < U2; ++u) h2[u] = row16_allsum(h2[u]);
      float dnn = 0.f;
#pragma unroll
      for (int u = 0; u < U2; ++u) {
        h2[u] = fmaxf(h2[u] + b1s[u], 0.f);
        dnn += h2[u] * K2s[u];
      }
      const bool valid = e16 < n_ex;
      const float z = (bias_r + fo + 0.5f * t) + dnn + b2_r;
      const float p = sigmoid_acc(z);
      const float y = valid ? label_r : 0.f;
      const float eps = 1e-7f;
      const float pc = fminf(fmaxf(p, eps), 1.f - eps);
      // v_log_f32 / v_rcp_f32 (1 ulp) instead of the library logf and IEEE divisions: the head runs on every wave of
      // the workgroup and is bound by instruction issue; the results stay ~1e-7 relative from the exact ones
      float le = -(y * __logf(pc + eps) + (1.f - y) * __logf(1.f - pc + eps));
      const float inside = (p >= eps && p <= 1.f - eps) ? 1.f : 0.f;
      float dz = -(y * __builtin_amdgcn_rcpf(pc + eps) - (1.f - y) * __builtin_amdgcn_rcpf(1.f - pc + eps)) * inside * p *
                 (1.f - p) * inv_B;
      if (!valid) { dz = 0.f; le = 0.f; }
      // way back: dpre2 (all 8 in every lane), then this lane's two units of dpre1
      float dha = 0.f, dhb = 0.f;
      const float kav[8] = {ka0.x, ka0.y, ka0.z, ka0.w, ka1.x, ka1.y, ka1.z, ka1.w};
      const float kbv[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
      float dp2[U2];
#pragma unroll
      for (int u = 0; u < U2; ++u) {
        dp2[u] = h2[u] > 0.f ? dz * K2s[u] : 0.f;
        dha += dp2[u] * kav[u];
        dhb += dp2[u] * kbv[u];
      }
      DP1[e16 * HS1 + 2 * g16] = h1a > 0.f ? dha : 0.f;
      DP1[e16 * HS1 + 2 * g16 + 1] = h1b > 0.f ? dhb : 0.f;
      if (g16 == 0) {                                        // every lane of the group holds the same values
        *reinterpret_cast<float4*>(h2s + e16 * U2) = make_float4(h2[0], h2[1], h2[2], h2[3]);
        *reinterpret_cast<float4*>(h2s + e16 * U2 + 4) = make_float4(h2[4], h2[5], h2[6], h2[7]);
        *reinterpret_cast<float4*>(dp2s + e16 * U2) = make_float4(dp2[0], dp2[1], dp2[2], dp2[3]);
        *reinterpret_cast<float4*>(dp2s + e16 * U2 + 4) = make_float4(dp2[4], dp2[5], dp2[6], dp2[7]);
        dzs[e16] = dz;
        lss[e16] = le;
        if (valid) {
          a.gz[ex0 + e16] = dz;
          if (a.prob) a.prob[ex0 + e16] = p;
        }
      }
    }
  }
  __syncthreads();
  STAMP(5);

  // ================================================ phase C ======================================================
  {
    const int j = cj, q = cq;
    // Both products are computed TRANSPOSED, so that an accumulator holds 4 consecutive floats of an output row
    // (one 16-byte store per lane and tile instead of four 4-byte ones):
    //   dX_f^T  [dim x example]  = K0_f [dim x unit] . dpre1^T [unit x example]     (M tile m = examples 16m..16m+15)
    //   dK0_f^T [unit x dim]     = dpre1^T [unit x example] . X_f [example x dim]   (M tile n = units 16n..16n+15)
    // operands made of dpre1, the same for every field (k slot q of step s <-> unit 8q+s resp. example ex(q,s)):
    //   aP[m][s]  dpre1[example 16m + j][unit 8q + s]                      B of dX^T
    //   bP[n][s]  dpre1[example ex(q,s)][unit 16n + j],  ex(q,s) = 4q + (s&3) + 16(s>>2)     A of dK0^T
    float aP[2][8], bP[2][8], dzr[2];
    float4 sr[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 p0 = *reinterpret_cast<const float4*>(DP1 + (16 * m + j) * HS1 + 8 * q);
      const float4 p1 = *reinterpret_cast<const float4*>(DP1 + (16 * m + j) * HS1 + 8 * q + 4);
      aP[m][0] = p0.x; aP[m][1] = p0.y; aP[m][2] = p0.z; aP[m][3] = p0.w;
      aP[m][4] = p1.x; aP[m][5] = p1.y; aP[m][6] = p1.z; aP[m][7] = p1.w;
      dzr[m] = dzs[16 * m + j];
      sr[m] = *reinterpret_cast<const float4*>(Ss + (16 * m + j) * E16 + 4 * q);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int s = 0; s < 8; ++s) bP[n][s] = DP1[(4 * q + (s & 3) + 16 * (s >> 2)) * HS1 + 16 * n + j];

    float* part = a.dK0part + (int64_t)blockIdx.x * D * U1;
    // K0_f fragments (A of dX^T): K0[f*16 + j][8q .. 8q+8), two 16-byte loads per lane out of L2 (the first field's were
    // requested before phase B); the next field's are in flight while this field is multiplied
#pragma unroll
    for (int i = 0; i < MAXFW; ++i) {
      const int f = wave + NWV * i;
      if (f >= F) break;                                              // wave-uniform
      const float kt[8] = {kt0.x, kt0.y, kt0.z, kt0.w, kt1.x, kt1.y, kt1.z, kt1.w};
      {
        const int fn = f + NWV;
        const float* kp = a.K0 + ((fn < F ? fn : f) * E16 + j) * U1 + 8 * q;
        kt0 = *reinterpret_cast<const float4*>(kp);
        kt1 = *reinterpret_cast<const float4*>(kp + 4);
      }
      // rows of this field: xk[s] = X[example ex(q,s)][f*16 + j]  (B of dK0^T; rows 4 apart are 16 banks apart) ...
      float xk[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xk[s] = XT[(4 * q + (s & 3) + 16 * (s >> 2)) * XS + f * E16 + j];
      // ... and in the layout of dX^T's accumulator: xo[m] = X[example 16m + j][f*16 + 4q .. 4q+4)
      float4 xo[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) xo[m] = *reinterpret_cast<const float4*>(XT + (16 * m + j) * XS + f * E16 + 4 * q);
      f32x4 dx[2], dk[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        dx[m][0] = 0.f; dx[m][1] = 0.f; dx[m][2] = 0.f; dx[m][3] = 0.f;
        dk[m][0] = 0.f; dk[m][1] = 0.f; dk[m][2] = 0.f; dk[m][3] = 0.f;
      }
      // four independent accumulators, interleaved: the 16x16x4 product has a 40-cycle dependent latency
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        dx[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[s], aP[0][s], dx[0], 0, 0, 0);
        dx[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[s], aP[1][s], dx[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[0][s], xk[s], dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[1][s], xk[s], dk[1], 0, 0, 0);
      }
      // IndexedSlices values: lane (example j of the tile, dims 4q..4q+3): 4 lanes cover the 64-byte (example, field) row
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int e = 16 * m + j;
        if (e < n_ex) {
          float4 o;
          o.x = dzr[m] * (sr[m].x - xo[m].x) + dx[m][0];
          o.y = dzr[m] * (sr[m].y - xo[m].y) + dx[m][1];
          o.z = dzr[m] * (sr[m].z - xo[m].z) + dx[m][2];
          o.w = dzr[m] * (sr[m].w - xo[m].w) + dx[m][3];
          if (DIRECT && dl[i][m] >= 0) {
            // head of its run: the row is final unless the run has more members (the post launch adds those)
            *reinterpret_cast<float4*>(a.g_embed + (int64_t)dl[i][m] * E16 + 4 * q) = o;
          } else
            *reinterpret_cast<float4*>(a.vals + ((ex0 + e) * F + f) * E16 + 4 * q) = o;
        }
      }
      // dK0 partial: row f*16 + j (this lane's dim), units 16n + 4q .. +4
#pragma unroll
      for (int n = 0; n < 2; ++n)
        *reinterpret_cast<float4*>(part + (f * E16 + j) * U1 + 16 * n + 4 * q) =
            make_float4(dk[n][0], dk[n][1], dk[n][2], dk[n][3]);
    }

    STAMP(6);                                   // this wave's fields are done
    // small per-workgroup partials (fixed order over the 32 examples), spread over the waves that own fewer fields
    float* sm = a.small + (int64_t)blockIdx.x * SMALL;
    if (wave >= 4) {
      const int t2 = tid - 256, k = t2 >> 3, u = t2 & 7;
      float s = 0.f;
#pragma unroll 8
      for (int e = 0; e < EX; ++e) s += H1s[e * HS1 + k] * dp2s[e * U2 + u];
      sm[t2] = s;                                              // dK1 [32][8]
    } else if (wave == 3) {
      if (lane < U1) {
        float t = 0.f;
#pragma unroll 8
        for (int e = 0; e < EX; ++e) t += DP1[e * HS1 + lane];
        sm[256 + lane] = t;                                    // db0
      }
    } else if (wave == 2) {
      if (lane < U2) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
        for (int e = 0; e < EX; ++e) { t1 += dp2s[e * U2 + lane]; t2 += h2s[e * U2 + lane] * dzs[e]; }
        sm[288 + lane] = t1;                                   // db1
        sm[296 + lane] = t2;                                   // dK2
      } else if (lane == 32) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
        for (int e = 0; e < EX; ++e) { t1 += dzs[e]; t2 += lss[e]; }
        sm[304] = t1;                                          // db2 = dbias
        sm[305] = t2;                                          // sum of per-example BCE terms
      }
    }
  }
  STAMP(7);
}

// fixed-order sum of the per-workgroup partials.  1024 threads = 16 slices x 64 lanes; lane owns 4 consecutive
// outputs (float4), slice q adds workgroups q, q+16, q+32, ... (independent 16-B loads), then the 16 slices are
// added in slice order through LDS.
struct ReduceArgs {
  const float* dK0part; const float* small; int nwg; int D; int64_t B;
  float* dK0; float* dK1; float* db0; float* db1; float* dK2; float* db2; float* dbias; float* loss;
};

// 1024 threads = 64 slices x 16 lanes; a lane owns 4 consecutive outputs (float4), slice q adds the partials of
// workgroups q, q+64, q+128, ... (all of them in flight before the first add), then the 64 slices meet in LDS and are
// added in a fixed two-level order.  16 float4 columns per workgroup -> 208 + 5 workgroups at F = 26: with 64 columns
// per workgroup (round 1) only 54 CUs shared the 13.6 MB of partials, 256 KB each, and the reduction took ~10 us.
constexpr int RC = 16;         // float4 columns per reduce workgroup
constexpr int RS = 64;         // slices
__host__ __device__ inline int reduce_blocks(int D) {
  return (int)(((int64_t)D * U1 / 4 + RC - 1) / RC) + (SMALL / 4 + RC - 1) / RC;
}

__device__ __forceinline__ void reduce_body(const ReduceArgs& r, int bidx) {
  const float* __restrict__ dK0part = r.dK0part;
  const float* __restrict__ small = r.small;
  const int nwg = r.nwg, D = r.D;
  const int64_t B = r.B;
  float* __restrict__ dK0 = r.dK0; float* __restrict__ dK1 = r.dK1; float* __restrict__ db0 = r.db0;
  float* __restrict__ db1 = r.db1; float* __restrict__ dK2 = r.dK2; float* __restrict__ db2 = r.db2;
  float* __restrict__ dbias = r.dbias; float* __restrict__ loss = r.loss;
  __shared__ float4 red[RS][RC];
  __shared__ float4 red2[8][RC];
  const int lane = threadIdx.x & (RC - 1), q = threadIdx.x / RC;
  const int64_t n0 = (int64_t)D * U1;                  // multiple of 4
  const int nb0 = (int)((n0 / 4 + RC - 1) / RC);       // blocks that cover dK0
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t e4 = 0;
  const bool is_small = bidx >= nb0;
  const float* src;
  int64_t stride;
  bool in = false;
  if (!is_small) {
    e4 = (int64_t)bidx * RC + lane;                    // float4 index into dK0
    in = e4 * 4 < n0;
    src = dK0part + e4 * 4;
    stride = n0;
  } else {
    e4 = (int64_t)(bidx - nb0) * RC + lane;            // float4 index into the SMALL block
    in = e4 * 4 < SMALL;
    src = small + e4 * 4;
    stride = SMALL;
  }
  if (in) {
    int w = q;
    for (; w + 3 * RS < nwg; w += 4 * RS) {
      float4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const float4*>(src + (int64_t)(w + RS * j) * stride);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc.x += x[j].x; acc.y += x[j].y; acc.z += x[j].z; acc.w += x[j].w; }
    }
    for (; w < nwg; w += RS) {
      float4 x = *reinterpret_cast<const float4*>(src + (int64_t)w * stride);
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
  }
  red[q][lane] = acc;
  __syncthreads();
  if (q < 8) {                                         // slices 8q .. 8q+7, in order
    float4 s = red[8 * q][lane];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      float4 x = red[8 * q + k][lane];
      s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    }
    red2[q][lane] = s;
  }
  __syncthreads();
  if (q != 0) return;
  float4 s = red2[0][lane];
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    float4 x = red2[k][lane];
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  if (!is_small) {
    if (e4 * 4 < n0) *reinterpret_cast<float4*>(dK0 + e4 * 4) = s;
    return;
  }
  if (e4 * 4 >= SMALL) return;
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

size_t fused_lds_bytes(int F) { return (size_t)carve_of(F).total * sizeof(float); }

// ------------------------------------------------------------------------------------------------
// per-column sort of the de-duplication plan (the DataGenerator contract gives every feature column its own contiguous
// id range, so duplicates only occur inside a column): 32-bit words (key << pos_bits | example), key = id - col_lo.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t PADW = 0xFFFFFFFFu;

struct ColSortArgs {
  int64_t B; int F; int64_t V; int key_bits; int pos_bits;
  int32_t* perm;        // [F][B]  sorted position -> example
  int64_t* col_uid;     // [F][B]  unique ids of the column, ascending (first col_nu[f] valid)
  int32_t* col_seg;     // [F][B+1] run starts in the column's sorted order (tail = B)
  int32_t* col_nu;      // [F]
  int* bad;
  int32_t* dloc;        // [F][B] or null: run index of lookup (f, example) inside its column, sign bit = not the run's head
};

// ------------------------------------------------------------------------------------------------
// ONE kernel: one 1024-thread workgroup per column sorts the column's <= 16384 words in LDS (stable LSD radix sort on
// the key bits, 7 bits per pass, in place: every key is in a register between the barrier that ends the reads and the
// one that starts the writes) and goes straight on to the run heads.  (Round 1 ran a chunk-sort / rank-merge / heads
// chain of three latency-bound launches on ~200 CUs: 65 us for the plans of four batches against 42 us here on
// 4 x F workgroups of 37 KB of LDS.)
//   ranking: element e = wave*64*KPT + round*64 + lane, so (wave, round, lane) order is array order; lanes of equal
//   digit are matched by 7 ballots, the lowest lane of a group bumps the wave's 16-bit counter of the digit (LDS
//   operations of one wave complete in program order); a key's new place = digit base + counts of earlier waves + its
//   rank in the wave.
// ------------------------------------------------------------------------------------------------
constexpr int OW_T = 1024, OW_W = OW_T / 64, OW_BINS = 128, OW_DB = 7;

#ifdef REC_SORT_STAMPS
__device__ unsigned long long g_sort_stamps[256 * 16];
#define SSTAMP(k) do { if (threadIdx.x == 0) g_sort_stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define SSTAMP(k) do {} while (0)
#endif

// up to 256 columns per launch (8 batches of 26..32 columns: ONE launch per 8 upcoming batches -- every sort launch holds
// its CUs for the duration of a latency-bound kernel, and a fused kernel that finds CUs taken runs a second round)
constexpr int SORT_MAX_COLS = 256;
struct SortCols {
  const int64_t* p[SORT_MAX_COLS];
};

template <int KPT>
__global__ __launch_bounds__(OW_T, 4) void colsort_onewg_kernel(SortCols cols, const int64_t* __restrict__ col_lo, ColSortArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t owl[];
  constexpr int NW = OW_T * KPT;                       // padded word count
  uint32_t* words = owl;                               // [NW]
  unsigned short* cnt = reinterpret_cast<unsigned short*>(owl + NW);      // [OW_W][OW_BINS]
  uint32_t* dbase = owl + NW + OW_W * OW_BINS / 2;     // [OW_BINS]
  uint32_t* wtot = dbase + OW_BINS;                    // [OW_W] scratch of the block scans
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = blockIdx.x;
  const int64_t B = a.B;
  const int64_t lo = col_lo[f];
  const int pb = a.pos_bits;
  const uint32_t pmask = (1u << pb) - 1u;
  // ---- load: word = (id - lo) << pos_bits | example; pad words sort last
  SSTAMP(0);
  bool bad = false;
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int e = r * OW_T + tid;
    uint32_t w = PADW;
    if (e < B) {
      const int64_t id = cols.p[f][e];
      int64_t key = id - lo;
      if (key < 0 || key >= (int64_t(1) << a.key_bits) || (uint64_t)id >= (uint64_t)a.V) {
        bad = true;
        key = key < 0 ? 0 : (int64_t(1) << a.key_bits) - 1;
      }
      w = ((uint32_t)key << pb) | (uint32_t)e;
    }
    words[e] = w;
  }
  if (bad && a.bad) *a.bad = 1;
  SSTAMP(1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  int pass_ = 0;
  // ---- radix passes over the key bits
  for (int shift = pb; shift < pb + a.key_bits; shift += OW_DB) {
    reinterpret_cast<uint32_t*>(cnt)[tid] = 0;         // OW_W*OW_BINS/2 = 1024 words
    __syncthreads();
    uint32_t w[KPT];
    unsigned short loc[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) w[r] = words[wave * (64 * KPT) + r * 64 + lane];
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
      const uint32_t d = (w[r] >> shift) & (OW_BINS - 1);
      unsigned long long m = ~0ull;
#pragma unroll
      for (int b = 0; b < OW_DB; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const unsigned short old = cnt[wave * OW_BINS + d];
      if ((m & lt) == 0) cnt[wave * OW_BINS + d] = (unsigned short)(old + __popcll(m));
      loc[r] = (unsigned short)(old + __popcll(m & lt));
    }
    __syncthreads();                                   // every word is in a register: the array may be overwritten
    if (tid < OW_BINS) {                               // per digit: counts -> exclusive prefix over the waves, total
      uint32_t run = 0;
#pragma unroll
      for (int q = 0; q < OW_W; ++q) {
        const uint32_t c = cnt[q * OW_BINS + tid];
        cnt[q * OW_BINS + tid] = (unsigned short)run;
        run += c;
      }
      // exclusive scan of the 128 totals (two waves)
      uint32_t incl = run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wtot[wave] = incl;
      dbase[tid] = incl - run;                         // within the wave; wave 1 adds wave 0's total below
    }
    __syncthreads();
    if (tid >= 64 && tid < OW_BINS) dbase[tid] += wtot[0];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
      const uint32_t d = (w[r] >> shift) & (OW_BINS - 1);
      words[dbase[d] + cnt[wave * OW_BINS + d] + loc[r]] = w[r];
    }
    __syncthreads();
    SSTAMP(2 + pass_);
    ++pass_;
  }
  // ---- run heads: thread t owns the KPT consecutive sorted positions from t*KPT
  const int s0 = tid * KPT;
  uint32_t v[KPT];
  bool hd[KPT];
  int heads = 0;
  uint32_t prev = s0 > 0 ? words[s0 - 1] : PADW;
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int sp = s0 + r;
    v[r] = words[sp];
    const uint32_t pk = (r == 0 ? prev : v[r - 1]) >> pb;
    hd[r] = sp < B && (sp == 0 || (v[r] >> pb) != pk);
    heads += hd[r] ? 1 : 0;
  }
  int incl = heads;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = (uint32_t)incl;
  __syncthreads();
  int woff = 0, all = 0;
  for (int q = 0; q < OW_W; ++q) {
    const int c = (int)wtot[q];
    if (q < wave) woff += c;
    all += c;
  }
  int rank = woff + incl - heads;
  // Outputs go through LDS and leave coalesced.  (Stored straight from the registers -- a lane owns 8 consecutive sorted
  // positions -- every wave instruction wrote 64 scattered 4- or 8-byte pieces: ~32 such instructions per wave on one
  // address path, and the last wave finished 12 us after the first, 36 us into a kernel whose sort is done at 19.)
  //   st_uq [rank]    key of the run (aliases `words`: every thread holds its words in registers behind the barrier above)
  //   st_sg [rank]    first sorted position of the run         (16 bit: B <= 16384)
  //   st_dl [example] run index | 0x8000 unless head of its run (16 bit)
  uint32_t* st_uq = words;
  unsigned short* st_sg = reinterpret_cast<unsigned short*>(wtot + OW_W);
  unsigned short* st_dl = st_sg + NW;
  int32_t* permf = a.perm + (int64_t)f * B;
  if ((B & 7) == 0 && KPT == 8) {
    // the thread's 8 consecutive perm entries as two 16-byte stores: a wave writes 2 KB of contiguous memory
    if (s0 < B) {
      int4 p0 = make_int4((int)(v[0] & pmask), (int)(v[1] & pmask), (int)(v[2] & pmask), (int)(v[3] & pmask));
      int4 p1 = make_int4((int)(v[4 % KPT] & pmask), (int)(v[5 % KPT] & pmask), (int)(v[6 % KPT] & pmask), (int)(v[7 % KPT] & pmask));
      *reinterpret_cast<int4*>(permf + s0) = p0;
      *reinterpret_cast<int4*>(permf + s0 + 4) = p1;
    }
  } else {
#pragma unroll
    for (int r = 0; r < KPT; ++r)
      if (s0 + r < B) permf[s0 + r] = (int32_t)(v[r] & pmask);
  }
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int sp = s0 + r;
    if (sp < B) {
      if (hd[r]) {
        st_uq[rank] = v[r] >> pb;
        st_sg[rank] = (unsigned short)sp;
        ++rank;
      }
      st_dl[v[r] & pmask] = (unsigned short)(hd[r] ? (rank - 1) : ((rank - 1) | 0x8000));
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int i = r * OW_T + tid;
    if (i < B) {
      if (a.dloc) {
        const uint32_t d = st_dl[i];
        a.dloc[(int64_t)f * B + i] = (int32_t)((d & 0x7FFFu) | ((d & 0x8000u) << 16));
      }
      if (i < all) {
        a.col_uid[(int64_t)f * B + i] = lo + (int64_t)st_uq[i];
        a.col_seg[(int64_t)f * (B + 1) + i] = (int32_t)st_sg[i];
      } else {
        a.col_seg[(int64_t)f * (B + 1) + i] = (int32_t)B;           // tail [all .. B] = B
      }
    }
  }
  if (tid == 0) a.col_seg[(int64_t)f * (B + 1) + B] = (int32_t)B;
  if (tid == 0) a.col_nu[f] = all;
  SSTAMP(8);
#ifdef REC_SORT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SSTAMP(9);
  if (threadIdx.x == 1023) g_sort_stamps[blockIdx.x * 16 + 10] = wall_clock64();
  if (threadIdx.x == 512) g_sort_stamps[blockIdx.x * 16 + 11] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------
// segment sums of both tables + global compaction.  One lane group (4 lanes x float4) per (column, local run).
// Long runs: the first 16 rows per group directly; what is left of a long run is summed by the whole wave.
// ------------------------------------------------------------------------------------------------
struct ColSegArgs {
  const float4* vals; const float* gz; const int32_t* perm; const int64_t* col_uid; const int32_t* col_seg;
  const int32_t* col_nu; int64_t B; int F; int64_t* uniq_ids; float4* g_embed; float* g_w; int64_t* n_uniq; int packed;
  // lazy (touched-rows) Adam applied to a row the moment its gradient is final (direct-mode post launch only; table ==
  // null: off).  table: fused rows [V, 32] = [embed 16 | w | pad]; m_e, v_e [V,16]; m_w, v_w [V]
  float* table; float* m_e; float* v_e; float* m_w; float* v_w; int64_t V; float lr_t, b1, b2, eps;
  // fixed-capacity exchange layout (sharded step): slot of every unique id (index = its rank in the batch's ascending
  // list) inside the [owners x capacity] send buffer; null = the compact list itself
  const int32_t* slot_map;
  // bias-corrected step size of the lazy Adam read from device memory (set by rec_adam_advance_f32): the train step then
  // holds no per-step host scalar and can be replayed from a hipGraph; null = lr_t above
  const float* lr_t_dev;
  // row strides (floats) of m_e / v_e and of m_w / v_w: 16 and 1 for dense state arrays; 32 and 32 when the state is
  // packed beside the rows ([m 16 | v 16] in one 128-byte row, m_w / v_w in the padding of the table row) so that a
  // touched row costs two line requests instead of five or six
  int64_t ldm, ldw;
  // exact lazy evaluation of Keras' dense sweep (rec_adam_keras_catchup_f32): last[row] = the step whose update the row
  // holds; a touched row holds step *step_dev afterwards.  null: plain touched-rows Adam
  int32_t* last; const int64_t* step_dev;
};

// m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; var <- var - lr_t m / (sqrt(v) + eps)   (rec_adam_rows_f32's formula)
__device__ __forceinline__ void adam_elem(float& var, float& m, float& v, float g, float lr_t, float b1, float b2,
                                          float eps) {
  adam_touch(var, m, v, g, lr_t, b1, b2, eps);       // common.h: rounding pinned, the same bits in every kernel
}
__device__ __forceinline__ void adam_chunk(const ColSegArgs& k, int64_t id, int c, const float4& g) {
  float4* vp = reinterpret_cast<float4*>(k.table + id * LD) + c;
  float4* mp = reinterpret_cast<float4*>(k.m_e + id * k.ldm) + c;
  float4* qp = reinterpret_cast<float4*>(k.v_e + id * k.ldm) + c;
  float4 x = *vp, m = *mp, v = *qp;
  adam_elem(x.x, m.x, v.x, g.x, k.lr_t, k.b1, k.b2, k.eps);
  adam_elem(x.y, m.y, v.y, g.y, k.lr_t, k.b1, k.b2, k.eps);
  adam_elem(x.z, m.z, v.z, g.z, k.lr_t, k.b1, k.b2, k.eps);
  adam_elem(x.w, m.w, v.w, g.w, k.lr_t, k.b1, k.b2, k.eps);
  *vp = x; *mp = m; *qp = v;
}
__device__ __forceinline__ void adam_w(const ColSegArgs& k, int64_t id, float g) {
  float x = k.table[id * LD + E16], m = k.m_w[id * k.ldw], v = k.v_w[id * k.ldw];
  adam_elem(x, m, v, g, k.lr_t, k.b1, k.b2, k.eps);
  k.table[id * LD + E16] = x; k.m_w[id * k.ldw] = m; k.v_w[id * k.ldw] = v;
}

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
    if (k.slot_map) dst = k.slot_map[dst];
  } else {
    if (k.slot_map) return;                      // the unused slots of a send buffer are never read by the owner
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
  if (c == 0 && uniq_ids) uniq_ids[dst] = idv;
  if (grp == 0 && c == 0 && n_uniq) *n_uniq = total;
}

__global__ __launch_bounds__(256) void colseg_sum_kernel(ColSegArgs k) { colseg_body(k, (int)blockIdx.x); }

// ---- direct mode: what is left for the launch after the fused kernel.  ONE LANE per run (slot) of the plan:
//   a run of one lookup (almost all of them with uniform ids): g_w = gz of that lookup, uniq_ids = its id -- coalesced
//     stores, the value row is already in place;
//   a pair: the lane adds the second member's value row to the head's row itself;
//   a run of 3..FIX_SHORT lookups: four lanes (one 16-byte chunk each), 16 runs of the wave at a time, members in order;
//   a run of up to FIX_HUGE: the whole wave adds it, 16 members x 4 float4 chunks per round, fixed butterfly over the rows;
//   longer ones (Zipf heads): the whole workgroup, partial rows of its 16 waves added in wave order;
//   slots beyond the column's runs: the zero-padded tail.
constexpr int FIX_T = 1024, FIX_HUGE = 128, FIX_SHORT = 17, FIX_SKEW = 256;
// slot t of a column (thread order) -> run u, twice.
// fix_deal: the mapping of the FAST path (ids, single lookups, pairs, the padded tail -- nearly everything, and all of it
// coalesced stores): a wave takes 4 consecutive runs from each sixteenth of the column, so that plan words arrive and ids
// leave as 16-byte pieces, and a workgroup owns whole lines of uniq_ids / g_w.  (Interleaving the waves of different
// workgroups here cost 2.3 us with uniform ids: every line of the outputs was then written in pieces by four CUs.)
// fix_spread: the mapping under which the runs of MORE THAN TWO members are looked at a second time and summed: lane l of
// the column's wave w takes run l * (B / 64) + w -- the hot ids of a field, neighbours at the head of the column with
// the synthetic Zipf draws, fall to different waves (a wave adds its long runs one after the other).  Read-only plan
// words, scattered row stores either way.
__device__ __forceinline__ int fix_deal(int t, int64_t B) {
  if ((B & 63) != 0) return t;
  return ((t & 63) >> 2) * (int)(B >> 4) + 4 * (t >> 6) + (t & 3);
}
__device__ __forceinline__ int fix_spread(int t, int64_t B) {
  if ((B & 63) != 0) return t;
  return (t & 63) * (int)(B >> 6) + (t >> 6);
}
// lanes 4g .. 4g+3 get the lane index of the g-th set bit of mask (-1: fewer than g+1 bits).  mask is wave-uniform: the
// scan runs on the scalar unit
__device__ __forceinline__ int fix_group_src(unsigned long long mask, int lane) {
  int src = -1;
  const int g = lane >> 2;
  for (int i = 0; i < 16 && mask; ++i) {
    const int sl = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    if (g == i) src = sl;
  }
  return src;
}
__device__ __forceinline__ void fixup_body(const ColSegArgs& k_in, int bidx) {
  ColSegArgs k = k_in;
  if (k.lr_t_dev) k.lr_t = *k.lr_t_dev;
  const float4* __restrict__ vals = k.vals;
  const float* __restrict__ gz = k.gz;
  const int64_t B = k.B;
  const int F = k.F;
  float4* __restrict__ g_embed = k.g_embed;
  const int tid = threadIdx.x, lane = tid & 63;
  const int per_col = (int)((B + FIX_T - 1) / FIX_T);           // workgroups per column
  const int f = bidx / per_col;
  const int t = (bidx - f * per_col) * FIX_T + tid;             // slot of the column before the deal
  const int u = fix_deal(t, B);
  __shared__ int nu_s[REC_MAX_COLS];
  // runs of more than FIX_HUGE members (Zipf heads: the hottest id of a column holds ~10 % of its lookups) are left to
  // the WHOLE workgroup: at most B / FIX_HUGE <= 128 of them exist in a column
  __shared__ int huge_n, huge_s0[128], huge_s1[128], huge_u[128];
  __shared__ float huge_gz[128];
  __shared__ float4 hpart[FIX_T / 64][4];
  __shared__ float hw_s[FIX_T / 64];
  if (tid < F) nu_s[tid] = k.col_nu[tid];
  if (tid == 0) huge_n = 0;
  __syncthreads();
  int64_t before = 0, total = 0;
  for (int q = 0; q < F; ++q) {
    const int nq = nu_s[q];
    if (q < f) before += nq;
    total += nq;
  }
  const int nu = nu_s[f];
  const bool in_col = t < B;
  const bool live = in_col && u < nu;
  const int32_t* pf = k.perm + (int64_t)f * B;
  int s0 = 0, s1 = 0;
  if (live) {
    s0 = k.col_seg[(int64_t)f * (B + 1) + u];
    s1 = k.col_seg[(int64_t)f * (B + 1) + u + 1];
  }
  const int len = s1 - s0;
  const int64_t dst = before + u;
  const bool adam = k.table != nullptr;
  float accw = 0.f;                                             // gz of the run's head
  if (live) {
    accw = gz[pf[s0]];
    const int64_t id = k.col_uid[(int64_t)f * B + u];
    if (len == 2) {                                             // a pair (nearly every multi-member run of a uniform batch):
      const int64_t b2 = pf[s0 + 1];                            // this lane alone, two dependent round trips
      const float4* r2 = vals + (b2 * F + f) * 4;
      float4 a0 = g_embed[dst * 4], a1 = g_embed[dst * 4 + 1], a2 = g_embed[dst * 4 + 2], a3 = g_embed[dst * 4 + 3];
      const float4 x0 = r2[0], x1 = r2[1], x2 = r2[2], x3 = r2[3];
      a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
      a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
      a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
      a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
      g_embed[dst * 4] = a0; g_embed[dst * 4 + 1] = a1; g_embed[dst * 4 + 2] = a2; g_embed[dst * 4 + 3] = a3;
      k.g_w[dst] = accw + gz[b2];
    }
    if (len == 1) k.g_w[dst] = accw;
    k.uniq_ids[dst] = id;
  } else if (in_col) {
    // padded tail: slot = total + rank among the column's unused slots; id = the smallest id of column 0, zero rows
    const int64_t d2 = total + ((int64_t)f * B - before) + (u - nu);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    g_embed[d2 * 4] = z; g_embed[d2 * 4 + 1] = z; g_embed[d2 * 4 + 2] = z; g_embed[d2 * 4 + 3] = z;
    k.g_w[d2] = 0.f;
    k.uniq_ids[d2] = k.col_uid[0];
  }
  // ---- runs of more than two members: the second look (fix_spread) -- for a column with more than FIX_SKEW lookups
  // beyond the first of their runs; a column without (uniform ids: ~90 pairs) has next to nothing to balance, and the
  // second look costs 1.3 us of scattered plan-word loads: its few longer runs are added where the fast path found them
  const bool skew = B - nu > FIX_SKEW;                           // uniform over the workgroup
  const int ub = skew ? fix_spread(t, B) : u;
  int s0b = s0, s1b = s1;
  if (skew) {
    s0b = 0; s1b = 0;
    if (in_col && ub < nu) {
      s0b = k.col_seg[(int64_t)f * (B + 1) + ub];
      s1b = k.col_seg[(int64_t)f * (B + 1) + ub + 1];
    }
  }
  const int lenb = s1b - s0b;
  float accwb = 0.f;                                            // gz of the run's head
  if (lenb > 2) accwb = skew ? gz[pf[s0b]] : accw;
  // (without skew the few runs of more than two members all take the whole-wave path below: no list, no barrier)
  if (skew && lenb > FIX_HUGE) {                                 // (which entry a run gets does not matter: every run is
    const int i = atomicAdd(&huge_n, 1);                         // summed on its own, in a fixed order)
    if (i < 128) { huge_s0[i] = s0b; huge_s1[i] = s1b; huge_u[i] = ub; huge_gz[i] = accwb; }
  }
  const int r = lane >> 2, c = lane & 3;
  // runs of 3..FIX_SHORT members: FOUR lanes per run (one 16-byte chunk of the row each), 16 runs per round; head + member
  // 2 + member 3 ... in member order, four members' loads in flight at a time.  (One lane per run walking its members
  // paid two dependent round trips per member: up to 14 in a row for a run of 8 -- with Zipf ids every wave has a few.)
  {
    unsigned long long shortm = __ballot(skew && lenb > 2 && lenb <= FIX_SHORT);
#ifdef ABL_NOSHORT
    shortm = 0;
#endif
    while (shortm) {                                            // wave-uniform
      const int src = fix_group_src(shortm, lane);
      const bool has = src >= 0;
      const int sl = has ? src : 0;
      const int gs0 = __shfl(s0b, sl, 64), gs1_ = __shfl(s1b, sl, 64), gu = __shfl(ub, sl, 64);
      float wsum = __shfl(accwb, sl, 64);
      const int gs1 = has ? gs1_ : 0;                            // (the shuffles themselves need every lane)
      const int64_t gdst = before + gu;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has) acc = g_embed[gdst * 4 + c];
      for (int sp = gs0 + 1; __any(sp < gs1); sp += 4) {
        if (sp < gs1) {
          int e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = pf[sp + j < gs1 ? sp + j : gs1 - 1];
          float4 xm[4];
          float gm[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            xm[j] = vals[((int64_t)e[j] * F + f) * 4 + c];
            gm[j] = gz[e[j]];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (sp + j < gs1) {
              acc.x += xm[j].x; acc.y += xm[j].y; acc.z += xm[j].z; acc.w += xm[j].w;
              wsum += gm[j];
            }
          }
        }
      }
      if (has) {
        g_embed[gdst * 4 + c] = acc;
        if (c == 0) k.g_w[gdst] = wsum;
      }
      for (int i = 0; i < 16 && shortm; ++i) shortm &= shortm - 1;
    }
  }
  // long runs, one at a time, by the whole wave: lane = (member slot r of 16, chunk c of 4)
  unsigned long long longm = __ballot(skew ? (lenb > FIX_SHORT && lenb <= FIX_HUGE) : lenb > 2);
#ifdef ABL_NOLONG
  longm = 0;
#endif
  while (longm) {
    const int src = __ffsll((long long)longm) - 1;
    longm &= longm - 1;
    const int rs0 = __shfl(s0b, src, 64), rs1 = __shfl(s1b, src, 64);
    const int ru = __shfl(ub, src, 64);
    const float rgz = __shfl(accwb, src, 64);
    const int64_t rdst = before + ru;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
    float pw = 0.f;
#pragma unroll 4
    for (int s = rs0 + 1 + r; s < rs1; s += 16) {              // independent loads: several rounds in flight
      const int64_t b = pf[s];
      const float4 x = vals[(b * F + f) * 4 + c];
      pa.x += x.x; pa.y += x.y; pa.z += x.z; pa.w += x.w;
      pw += gz[b];
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {                         // fixed butterfly over the 16 member slots
      pa.x += __shfl_xor(pa.x, o, 64); pa.y += __shfl_xor(pa.y, o, 64);
      pa.z += __shfl_xor(pa.z, o, 64); pa.w += __shfl_xor(pa.w, o, 64);
      pw += __shfl_xor(pw, o, 64);
    }
    if (r == 0) {
      float4 h = g_embed[rdst * 4 + c];
      h.x += pa.x; h.y += pa.y; h.z += pa.z; h.w += pa.w;
      g_embed[rdst * 4 + c] = h;
      const float wsum = rgz + pw;
      if (c == 0) k.g_w[rdst] = wsum;
    }
  }
  // huge runs, up to 16 at a time, by the whole workgroup: with n of them in a round, 16 / n waves share a run (wave of run
  // h, sub-index sw: member slots 16 sw + r, stride 16 * waves-per-run, four members per lane in flight); the waves'
  // partial rows meet in LDS and the run's first wave adds them in wave order.  (One wave walking the 848 members of a
  // Zipf head 64 at a time was a 13-round dependent chain; the whole workgroup taking the huge runs one after the other
  // still paid ~5 round trips per run: 16 us of the 34-us launch with Zipf ids.)
  if (skew) __syncthreads();                                     // (uniform over the workgroup)
#ifdef ABL_NOHUGE
  const int n_huge = 0;
#else
  const int n_huge = skew ? (huge_n < 128 ? huge_n : 128) : 0;
#endif
  const int wv = tid >> 6;
  constexpr int NWV_F = FIX_T / 64;
  for (int h0 = 0; h0 < n_huge; h0 += NWV_F) {
    const int nr = n_huge - h0 < NWV_F ? n_huge - h0 : NWV_F;    // runs of this round
    const int wpr = NWV_F / nr;                                  // waves per run
    const int hl = wv / wpr, sw = wv - hl * wpr;
    const bool mine = hl < nr;
    const int h = h0 + (mine ? hl : 0);
    const int rs0 = huge_s0[h], rs1 = mine ? huge_s1[h] : 0;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
    float pw = 0.f;
    const int stride = 16 * wpr;
    // four members per lane and round in flight (eight cost 25 more registers: at 81 the launch dropped to one workgroup
    // per CU and the uniform step lost 0.7 us)
    for (int base = rs0 + 1 + 16 * sw + r; base < rs1; base += 4 * stride) {
      int bm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sidx = base + j * stride;
        bm[j] = pf[sidx < rs1 ? sidx : rs1 - 1];
      }
      float4 xm[4];
      float gm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xm[j] = vals[((int64_t)bm[j] * F + f) * 4 + c];
        gm[j] = gz[bm[j]];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (base + j * stride < rs1) {
          pa.x += xm[j].x; pa.y += xm[j].y; pa.z += xm[j].z; pa.w += xm[j].w;
          pw += gm[j];
        }
      }
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
      pa.x += __shfl_xor(pa.x, o, 64); pa.y += __shfl_xor(pa.y, o, 64);
      pa.z += __shfl_xor(pa.z, o, 64); pa.w += __shfl_xor(pa.w, o, 64);
      pw += __shfl_xor(pw, o, 64);
    }
    float4 hsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t rdst = before + huge_u[h];
    const bool fin = mine && sw == 0 && r == 0;                  // the run's first wave, one lane per chunk
    if (fin) hsum = g_embed[rdst * 4 + c];                       // (in flight across the barrier)
    if (r == 0) {
      hpart[wv][c] = pa;
      if (c == 0) hw_s[wv] = pw;
    }
    __syncthreads();
    if (fin) {
      float wsum = huge_gz[h];
      for (int q = 0; q < wpr; ++q) {
        const float4 x = hpart[wv + q][c];
        hsum.x += x.x; hsum.y += x.y; hsum.z += x.z; hsum.w += x.w;
        wsum += hw_s[wv + q];
      }
      g_embed[rdst * 4 + c] = hsum;
      if (c == 0) k.g_w[rdst] = wsum;
    }
    __syncthreads();
  }
  if (bidx == 0 && tid == 0) *k.n_uniq = total;
  // The optimizer, once every row sum of this workgroup's slots is final: 8 lanes per row -- lanes 0..3 one 16-byte
  // chunk of the embedding row and of its moments, lane 4 the first-order weight -- so that a wave instruction covers 8
  // rows with one line request each.  (Applied by the lane that owns the run, 16 bytes at a time, every instruction
  // touched 64 different rows: 2.5 M line requests per launch for 0.4 M distinct lines, 75 us.)
  if (adam) {                                                   // uniform over the launch
    __syncthreads();
    const int piece = tid & 7, rr = tid >> 3;
    const int step_now = k.last ? (int32_t)*k.step_dev : 0;
    // a row is updated by the workgroup that made its sum final: runs of up to two members under the fast mapping, longer
    // ones under the spread mapping (another workgroup may still be adding those of this one's fast slots)
    const bool skew_a = B - nu > FIX_SKEW;
#pragma unroll 1
    for (int p = 0; p < (skew_a ? 2 : 1) * (FIX_T / 128); ++p) {
      const bool spread = p >= FIX_T / 128;
      const int tl = (bidx - f * per_col) * FIX_T + (spread ? p - FIX_T / 128 : p) * 128 + rr;
      const int uu = spread ? fix_spread(tl, B) : fix_deal(tl, B);
      if (tl < B && uu < nu && piece < 5) {
        const int ln = k.col_seg[(int64_t)f * (B + 1) + uu + 1] - k.col_seg[(int64_t)f * (B + 1) + uu];
        if (!skew_a || (ln > 2) == spread) {
          const int64_t d2 = before + uu;
          const int64_t id = k.col_uid[(int64_t)f * B + uu];
          if ((uint64_t)id < (uint64_t)k.V) {
            if (piece < 4) {
              adam_chunk(k, id, piece, g_embed[d2 * 4 + piece]);
            } else {
              adam_w(k, id, k.g_w[d2]);
              if (k.last) k.last[id] = step_now;
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(1024, 8) void deepfm_post_direct_kernel(ReduceArgs r, ColSegArgs k, int nb_reduce) {
  if ((int)blockIdx.x < nb_reduce) reduce_body(r, (int)blockIdx.x);
  else fixup_body(k, (int)blockIdx.x - nb_reduce);
}

// reduction of the workgroup partials and the segment sums only depend on the fused kernel, not on each other: one
// launch, the first nb_reduce workgroups (1024 threads) reduce, the others sum segments -- they run side by side
__global__ __launch_bounds__(1024) void deepfm_post_kernel(ReduceArgs r, ColSegArgs k, int nb_reduce) {
  if ((int)blockIdx.x < nb_reduce) reduce_body(r, (int)blockIdx.x);
  else colseg_body(k, (int)blockIdx.x - nb_reduce);
}


// ---- Keras Adam, evaluated lazily AND exactly.  Keras' sparse apply is a dense sweep: every row of the table decays its
// moments and moves every step (2.FM/ModelManager.py:178-179 on IndexedSlices gradients).  The update of an untouched row
// at step j depends on nothing but the row and lr_j, so a row may skip the sweeps and REPLAY them later: last[row] = the
// step the row holds, and before a batch reads its rows this kernel applies the steps last+1 .. *step_dev they missed,
// element by element with the sweep's own arithmetic (adam_decay) -- the same bits the sweep would have left.  The rows
// of a batch then receive the touched update of the current step in the post launch, which also sets last.
struct CatchArgs {
  const int64_t* col_uid; const int32_t* col_nu; int64_t B; int F;   // plan of the batch; null col_uid: all rows [0, V)
  float* table; int64_t V; float* m_e; float* v_e; int64_t ldm; float* m_w; float* v_w; int64_t ldw;
  const int32_t* last; const int64_t* step_dev; const float* lr_tab; int64_t n_tab; float b1, b2, eps;
};
// A workgroup (256 threads) takes CATCH_R = 64 consecutive slots (unique ids of the plan, or table rows for the flush),
// FOUR lanes per row: lane q owns elements 4q .. 4q+3 of the row (lane 0 also the first-order weight) -- four or five
// independent chains per lane, 512 rows in flight per CU.  The kernel is a chain of latencies, not arithmetic (at 1M rows
// the arithmetic is ~25 us of the chip, the first form -- 16 lanes per row, 128 rows in flight per CU, a global load of
// lr_t in every replayed step, three dependent round trips before the first of them -- took 78 us):
//   trip 1   plan word of the slot (col_nu of its column beside it)
//   trip 2   last[id] AND the row's x / m / v pieces (their addresses only need the id)
//   LDS      the rows change hands: how many steps each one replays is known now, and the rows are handed out again in
//            DESCENDING order of that count -- a wave of 16 rows runs as long as its longest row, and with the pending
//            counts of a batch spread geometrically (mean ~47 at 213k of 10M rows per step) unsorted waves ran twice the
//            mean
//   replay   lr_t of the last CATCH_LR steps comes from LDS (older ones from the table in memory)
constexpr int CATCH_R = 64, CATCH_T = 256, CATCH_LR = 1024;
__global__ __launch_bounds__(CATCH_T) void adam_keras_catchup_kernel(CatchArgs a) {
  __shared__ float4 x_s[CATCH_R][4], m_s[CATCH_R][4], v_s[CATCH_R][4];
  __shared__ float xw_s[CATCH_R], mw_s[CATCH_R], vw_s[CATCH_R];
  __shared__ int64_t id_s[CATCH_R];
  __shared__ int k_s[CATCH_R];
  __shared__ unsigned char ord[CATCH_R];
  __shared__ float lr_s[CATCH_LR];
  const int tid = threadIdx.x, r = tid >> 2, q = tid & 3;
  const int64_t nslot = a.col_uid ? a.B * a.F : a.V;
  const int64_t j1 = *a.step_dev;
#pragma unroll
  for (int i = tid; i < CATCH_LR; i += CATCH_T) {          // lr_s[i] = lr_t of step j1 - i
    const int64_t j = j1 - i;
    lr_s[i] = j >= 1 ? a.lr_tab[(j < a.n_tab ? j : a.n_tab) - 1] : 0.f;
  }
  const int64_t slot = (int64_t)blockIdx.x * CATCH_R + r;
  int64_t id = -1;
  if (a.col_uid) {
    const int64_t sc = slot < nslot ? slot : nslot - 1;    // unconditional loads, judged afterwards
    const int f = (int)(sc / a.B);
    const int64_t u = sc - (int64_t)f * a.B;
    const int nu = a.col_nu[f];
    const int64_t cand = a.col_uid[sc];
    if (slot < nslot && u < nu) id = cand;
  } else if (slot < nslot) {
    id = slot;
  }
  const bool ok = (uint64_t)id < (uint64_t)a.V;
  const int64_t idc = ok ? id : 0;
  const int j0 = a.last[idc];
  float4 x4 = *reinterpret_cast<const float4*>(a.table + idc * LD + 4 * q);
  float4 m4 = *reinterpret_cast<const float4*>(a.m_e + idc * a.ldm + 4 * q);
  float4 v4 = *reinterpret_cast<const float4*>(a.v_e + idc * a.ldm + 4 * q);
  x_s[r][q] = x4; m_s[r][q] = m4; v_s[r][q] = v4;
  if (q == 0) {
    xw_s[r] = a.table[idc * LD + E16];
    mw_s[r] = a.m_w[idc * a.ldw];
    vw_s[r] = a.v_w[idc * a.ldw];
    id_s[r] = id;
    k_s[r] = (ok && j1 > j0) ? (int)(j1 - j0) : 0;
  }
  __syncthreads();
  if (tid < CATCH_R) {
    const int k = k_s[tid];
    int rank = 0;
#pragma unroll 8
    for (int c = 0; c < CATCH_R; ++c) rank += (k_s[c] > k || (k_s[c] == k && c < tid)) ? 1 : 0;
    ord[rank] = (unsigned char)tid;
  }
  __syncthreads();
  const int rr = ord[r];                                   // the row this lane group replays
  const int k = k_s[rr];
  if (k <= 0) return;
  const int64_t rid = id_s[rr];
  x4 = x_s[rr][q]; m4 = m_s[rr][q]; v4 = v_s[rr][q];
  float xw = 0.f, mw = 0.f, vw = 0.f;
  if (q == 0) { xw = xw_s[rr]; mw = mw_s[rr]; vw = vw_s[rr]; }
  // (a row that was never touched -- m = v = 0 -- stays put: x - lr*0/(0+eps) = x; skipping it is only cheaper)
  const bool any_e = m4.x != 0.f || v4.x != 0.f || m4.y != 0.f || v4.y != 0.f || m4.z != 0.f || v4.z != 0.f ||
                     m4.w != 0.f || v4.w != 0.f;
  const bool any_w = q == 0 && (mw != 0.f || vw != 0.f);
  if (!any_e && !any_w) return;
  for (int i = k - 1; i >= 0; --i) {                       // step j1 - i
    float lr;
    if (i < CATCH_LR) {
      lr = lr_s[i];
    } else {
      const int64_t j = j1 - i;
      lr = a.lr_tab[(j < a.n_tab ? j : a.n_tab) - 1];
    }
    adam_decay(x4.x, m4.x, v4.x, lr, a.b1, a.b2, a.eps);
    adam_decay(x4.y, m4.y, v4.y, lr, a.b1, a.b2, a.eps);
    adam_decay(x4.z, m4.z, v4.z, lr, a.b1, a.b2, a.eps);
    adam_decay(x4.w, m4.w, v4.w, lr, a.b1, a.b2, a.eps);
    if (q == 0) adam_decay(xw, mw, vw, lr, a.b1, a.b2, a.eps);
  }
  if (any_e) {
    *reinterpret_cast<float4*>(a.table + rid * LD + 4 * q) = x4;
    *reinterpret_cast<float4*>(a.m_e + rid * a.ldm + 4 * q) = m4;
    *reinterpret_cast<float4*>(a.v_e + rid * a.ldm + 4 * q) = v4;
  }
  if (any_w) {
    a.table[rid * LD + E16] = xw;
    a.m_w[rid * a.ldw] = mw;
    a.v_w[rid * a.ldw] = vw;
  }
}
__global__ __launch_bounds__(256) void fill_last_kernel(int32_t* __restrict__ last, int64_t V,
                                                        const int64_t* __restrict__ step_dev) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < V) last[t] = (int32_t)*step_dev;
}

}  // namespace

extern "C" size_t rec_deepfm_fused_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  size_t nwg = (size_t)ceil_div64(B, EX);
  return sizeof(float) * nwg * ((size_t)F * E16 * U1 + SMALL) + 256;
}

#ifdef REC_SORT_STAMPS
extern "C" int rec_debug_sort_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sort_stamps), sizeof(unsigned long long) * 256 * 16);
}
#endif
#ifdef REC_FUSED_STAMPS
// diagnostic build only (scripts/exp/fused_stamps.py builds it into its own library): phase stamps of the last launch
static unsigned long long* g_fused_stamps = nullptr;
extern "C" int rec_debug_fused_stamps(unsigned long long* host_out, int nwg) {
  if (!g_fused_stamps) return REC_E_ARG;
  return (int)hipMemcpy(host_out, g_fused_stamps, sizeof(unsigned long long) * 12 * NWV * (size_t)nwg, hipMemcpyDeviceToHost);
}
#endif

struct DirectArgs {
  const int32_t* dloc; const int32_t* col_nu; float* g_embed; float* g_w; int64_t* uniq_ids;
};

static int launch_fused(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F, int64_t B,
                        const float* bias, const float* K0, const float* b0, const float* K1, const float* b1,
                        const float* K2, const float* b2, const float* label, float* gz, float* vals, float* prob,
                        float* dK0, float* db0, float* dK1, float* db1, float* dK2, float* db2, float* dbias,
                        float* loss, int* oob_flag, void* workspace, void* stream, const ColSegArgs* seg,
                        bool main_only = false, const DirectArgs* direct = nullptr) {
  if (B <= 0 || F <= 0 || V <= 0) return REC_E_ARG;
  if (F > 28 || F > REC_MAX_COLS || V >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;
  if (direct ? ld != LD : (ld < 20 || (ld & 3) != 0)) return REC_E_UNSUPPORTED;
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
#ifdef REC_FUSED_STAMPS
  static unsigned long long* stamps = nullptr;
  if (!stamps && hipMalloc(&stamps, sizeof(unsigned long long) * 12 * NWV * 65536) != hipSuccess) return REC_E_ARG;
  g_fused_stamps = stamps;
  FusedArgs a{table, V, bias, K0, b0, K1, b1, K2, b2, label, B, F, gz, vals, prob, dK0part, small, oob_flag,
              direct ? direct->dloc : nullptr, direct ? direct->col_nu : nullptr, direct ? direct->g_embed : nullptr,
              direct ? direct->g_w : nullptr, direct ? direct->uniq_ids : nullptr, (int)ld, stamps};
#else
  FusedArgs a{table, V, bias, K0, b0, K1, b1, K2, b2, label, B, F, gz, vals, prob, dK0part, small, oob_flag,
              direct ? direct->dloc : nullptr, direct ? direct->col_nu : nullptr, direct ? direct->g_embed : nullptr,
              direct ? direct->g_w : nullptr, direct ? direct->uniq_ids : nullptr, (int)ld};
#endif
  if (direct) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(deepfm_fwd_bwd_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(deepfm_fwd_bwd_kernel<true>, dim3(nwg), dim3(512), lds, st, cp, a);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(deepfm_fwd_bwd_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(deepfm_fwd_bwd_kernel<false>, dim3(nwg), dim3(512), lds, st, cp, a);
  }
  REC_LAUNCH_CHECK();
  if (main_only) return REC_OK;
  int D = F * E16;
  unsigned nb = (unsigned)reduce_blocks(D);
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
               packed ? (float*)nullptr : g_w_rows, n_uniq, packed ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f,
               0.f, 0.f, 0.f};
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

static int launch_post(bool direct, int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                       float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss, void* workspace,
                       const int32_t* perm, const int64_t* col_uid, const int32_t* col_seg, const int32_t* col_nu,
                       int64_t* uniq_ids, float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, int packed,
                       void* stream, const ColSegArgs* adam = nullptr) {
  if (B <= 0 || F <= 0 || F > 28) return REC_E_ARG;
  if (!gz || !vals || !dK0 || !db0 || !dK1 || !db1 || !dK2 || !db2 || !dbias || !loss || !workspace || !perm ||
      !col_uid || !col_seg || !col_nu || !g_embed_rows || (!packed && !g_w_rows))
    return REC_E_ARG;
  const bool slots = adam && adam->slot_map;
  if (!slots && (!uniq_ids || !n_uniq)) return REC_E_ARG;
  if (slots && (direct || !packed)) return REC_E_UNSUPPORTED;
  if (direct && packed) return REC_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(vals) & 15) != 0 || (reinterpret_cast<uintptr_t>(g_embed_rows) & 15) != 0)
    return REC_E_UNSUPPORTED;
  int nwg = (int)ceil_div64(B, EX);
  int D = F * E16;
  float* dK0part = (float*)workspace;
  float* small = dK0part + (size_t)nwg * F * E16 * U1;
  unsigned nb = (unsigned)reduce_blocks(D);
  unsigned nbs = (unsigned)ceil_div64(B * F * 4, 1024);
  ReduceArgs r{dK0part, small, nwg, D, B, dK0, dK1, db0, db1, dK2, db2, dbias, loss};
  ColSegArgs k{(const float4*)vals, gz, perm, col_uid, col_seg, col_nu, B, F, uniq_ids, (float4*)g_embed_rows,
               packed ? (float*)nullptr : g_w_rows, n_uniq, packed ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f,
               0.f, 0.f, 0.f, nullptr};
  if (slots) {
    k.slot_map = adam->slot_map;
  } else if (adam) {
    k.table = adam->table; k.m_e = adam->m_e; k.v_e = adam->v_e; k.m_w = adam->m_w; k.v_w = adam->v_w; k.V = adam->V;
    k.lr_t = adam->lr_t; k.b1 = adam->b1; k.b2 = adam->b2; k.eps = adam->eps; k.lr_t_dev = adam->lr_t_dev;
    k.ldm = adam->ldm ? adam->ldm : E16; k.ldw = adam->ldw ? adam->ldw : 1;
    k.last = adam->last; k.step_dev = adam->step_dev;
  }
  if (direct) {
    unsigned nbf = (unsigned)F * (unsigned)ceil_div64(B, FIX_T);
    hipLaunchKernelGGL(deepfm_post_direct_kernel, dim3(nb + nbf), dim3(1024), 0, as_stream(stream), r, k, (int)nb);
  } else {
    hipLaunchKernelGGL(deepfm_post_kernel, dim3(nb + nbs), dim3(1024), 0, as_stream(stream), r, k, (int)nb);
  }
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_deepfm_fused_post_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0, float* db0,
                                         float* dK1, float* db1, float* dK2, float* db2, float* dbias, float* loss,
                                         void* workspace, const int32_t* perm, const int64_t* col_uid,
                                         const int32_t* col_seg, const int32_t* col_nu, int64_t* uniq_ids,
                                         float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, int packed,
                                         void* stream) {
  return launch_post(false, F, B, gz, vals, dK0, db0, dK1, db1, dK2, db2, dbias, loss, workspace, perm, col_uid, col_seg,
                     col_nu, uniq_ids, g_embed_rows, g_w_rows, n_uniq, packed, stream);
}

// sharded step with fixed-capacity exchanges: the packed rows [embed 16 | w | 0 0 0] of the batch's unique ids go to the
// slots rec_colsort_shard_map_fixed_i64 gave them inside g_rows [owners * capacity, 20]; unused slots are not written
extern "C" int rec_deepfm_fused_post_slots_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0,
                                               float* db0, float* dK1, float* db1, float* dK2, float* db2, float* dbias,
                                               float* loss, void* workspace, const int32_t* perm, const int64_t* col_uid,
                                               const int32_t* col_seg, const int32_t* col_nu, const int32_t* slot_map,
                                               float* g_rows, void* stream) {
  if (!slot_map) return REC_E_ARG;
  ColSegArgs a{};
  a.slot_map = slot_map;
  return launch_post(false, F, B, gz, vals, dK0, db0, dK1, db1, dK2, db2, dbias, loss, workspace, perm, col_uid, col_seg,
                     col_nu, nullptr, g_rows, nullptr, nullptr, 1, stream, &a);
}

// direct mode: the plan (rec_colsort_plan_dest_i64) exists BEFORE the fused kernel runs; the kernel writes the value row
// of every run's first member straight into g_embed_rows, rec_deepfm_fused_post_direct_f32 finishes runs with more
// members, fills uniq_ids / g_w_rows / n_uniq and the padded tail, and reduces the dense partials
extern "C" int rec_deepfm_fused_main_direct_f32(const float* table, int64_t ld, int64_t V,
                                                const int64_t* const* cols_host, int F, int64_t B, const float* bias,
                                                const float* K0, const float* b0, const float* K1, const float* b1,
                                                const float* K2, const float* b2, const float* label, float* gz,
                                                float* vals, float* prob, int* oob_flag, void* workspace,
                                                const int32_t* dloc, const int32_t* col_nu, float* g_embed_rows,
                                                void* stream) {
  if (!dloc || !col_nu || !g_embed_rows) return REC_E_ARG;
  if ((reinterpret_cast<uintptr_t>(g_embed_rows) & 15) != 0) return REC_E_UNSUPPORTED;
  float dummy = 0.f;
  float* d = &dummy;
  DirectArgs da{dloc, col_nu, g_embed_rows, nullptr, nullptr};
  return launch_fused(table, ld, V, cols_host, F, B, bias, K0, b0, K1, b1, K2, b2, label, gz, vals, prob, d, d, d, d, d,
                      d, d, d, oob_flag, workspace, stream, nullptr, true, &da);
}

extern "C" int rec_deepfm_fused_post_direct_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0,
                                                float* db0, float* dK1, float* db1, float* dK2, float* db2, float* dbias,
                                                float* loss, void* workspace, const int32_t* perm, const int64_t* col_uid,
                                                const int32_t* col_seg, const int32_t* col_nu, int64_t* uniq_ids,
                                                float* g_embed_rows, float* g_w_rows, int64_t* n_uniq, void* stream) {
  return launch_post(true, F, B, gz, vals, dK0, db0, dK1, db1, dK2, db2, dbias, loss, workspace, perm, col_uid, col_seg,
                     col_nu, uniq_ids, g_embed_rows, g_w_rows, n_uniq, 0, stream);
}

// ... with the lazy (touched-rows) Adam update of both tables (rec_adam_rows_f32's arithmetic) applied to every row the
// moment its gradient is final: no second pass over g_embed_rows / g_w_rows, no extra launch.  table: the fused rows
// [V, 32] (embed 16 | w | pad); t: 1-based step for the bias correction.  Non-reference semantics (Keras' sparse apply is
// a dense sweep: rec_adam_sparse_keras_pair_f32), opt-in.
extern "C" int rec_deepfm_fused_post_direct_adam_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0,
                                                     float* db0, float* dK1, float* db1, float* dK2, float* db2,
                                                     float* dbias, float* loss, void* workspace, const int32_t* perm,
                                                     const int64_t* col_uid, const int32_t* col_seg,
                                                     const int32_t* col_nu, int64_t* uniq_ids, float* g_embed_rows,
                                                     float* g_w_rows, int64_t* n_uniq, float* table, int64_t ld,
                                                     int64_t V, float* m_e, float* v_e, float* m_w, float* v_w, int64_t t,
                                                     float lr, float b1, float b2, float eps, void* stream) {
  if (!table || !m_e || !v_e || !m_w || !v_w || V <= 0 || t < 1) return REC_E_ARG;
  if (ld != LD || (reinterpret_cast<uintptr_t>(table) & 15) != 0 || (reinterpret_cast<uintptr_t>(m_e) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(v_e) & 15) != 0)
    return REC_E_UNSUPPORTED;
  // float32 arithmetic as Keras does (tf.pow on float32 scalars), the same as rec_adam_rows_f32
  const float b1p = powf(b1, (float)t), b2p = powf(b2, (float)t);
  ColSegArgs a{};
  a.table = table; a.m_e = m_e; a.v_e = v_e; a.m_w = m_w; a.v_w = v_w; a.V = V;
  a.lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p); a.b1 = b1; a.b2 = b2; a.eps = eps;
  return launch_post(true, F, B, gz, vals, dK0, db0, dK1, db1, dK2, db2, dbias, loss, workspace, perm, col_uid, col_seg,
                     col_nu, uniq_ids, g_embed_rows, g_w_rows, n_uniq, 0, stream, &a);
}

// ... with the step size read from device memory (lr_t_dev, advanced by rec_adam_advance_f32 on the same stream): no
// per-step host scalar, so a whole train step -- this launch included -- replays from a hipGraph
extern "C" int rec_deepfm_fused_post_direct_adam_dev_f32(int F, int64_t B, const float* gz, const float* vals, float* dK0,
                                                         float* db0, float* dK1, float* db1, float* dK2, float* db2,
                                                         float* dbias, float* loss, void* workspace, const int32_t* perm,
                                                         const int64_t* col_uid, const int32_t* col_seg,
                                                         const int32_t* col_nu, int64_t* uniq_ids, float* g_embed_rows,
                                                         float* g_w_rows, int64_t* n_uniq, float* table, int64_t ld,
                                                         int64_t V, float* m_e, float* v_e, float* m_w, float* v_w,
                                                         int64_t ld_state, int64_t ld_wstate,
                                                         const float* lr_t_dev, float b1, float b2, float eps,
                                                         int32_t* last, const int64_t* step_dev, void* stream) {
  if (!table || !m_e || !v_e || !m_w || !v_w || !lr_t_dev || V <= 0 || ld_state < E16 || (ld_state & 3) != 0 ||
      ld_wstate < 1)
    return REC_E_ARG;
  if (ld != LD || (reinterpret_cast<uintptr_t>(table) & 15) != 0 || (reinterpret_cast<uintptr_t>(m_e) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(v_e) & 15) != 0)
    return REC_E_UNSUPPORTED;
  ColSegArgs a{};
  a.table = table; a.m_e = m_e; a.v_e = v_e; a.m_w = m_w; a.v_w = v_w; a.V = V;
  if (last && !step_dev) return REC_E_ARG;
  a.lr_t = 0.f; a.lr_t_dev = lr_t_dev; a.b1 = b1; a.b2 = b2; a.eps = eps; a.ldm = ld_state; a.ldw = ld_wstate;
  a.last = last; a.step_dev = step_dev;
  return launch_post(true, F, B, gz, vals, dK0, db0, dK1, db1, dK2, db2, dbias, loss, workspace, perm, col_uid, col_seg,
                     col_nu, uniq_ids, g_embed_rows, g_w_rows, n_uniq, 0, stream, &a);
}

static int catchup_args_ok(const float* table, int64_t ld, int64_t V, const float* m_e, const float* v_e, int64_t ld_state,
                           const float* m_w, const float* v_w, int64_t ld_wstate, const int32_t* last,
                           const int64_t* step_dev, const float* lr_table, int64_t n_table) {
  if (!table || !m_e || !v_e || !m_w || !v_w || !last || !step_dev || !lr_table || V <= 0 || n_table <= 0 ||
      ld_state < E16 || ld_wstate < 1)
    return REC_E_ARG;
  if (ld != LD) return REC_E_UNSUPPORTED;
  // the rows' x / m / v travel as 16-byte pieces
  if ((ld_state & 3) != 0 || (reinterpret_cast<uintptr_t>(table) & 15) != 0 || (reinterpret_cast<uintptr_t>(m_e) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(v_e) & 15) != 0)
    return REC_E_UNSUPPORTED;
  return REC_OK;
}

extern "C" int rec_adam_keras_catchup_f32(const int64_t* col_uid, const int32_t* col_nu, int64_t B, int F, float* table,
                                          int64_t ld, int64_t V, float* m_e, float* v_e, int64_t ld_state, float* m_w,
                                          float* v_w, int64_t ld_wstate, const int32_t* last, const int64_t* step_dev,
                                          const float* lr_table, int64_t n_table, float b1, float b2, float eps,
                                          void* stream) {
  if (!col_uid || !col_nu || B <= 0 || F <= 0) return REC_E_ARG;
  int rc = catchup_args_ok(table, ld, V, m_e, v_e, ld_state, m_w, v_w, ld_wstate, last, step_dev, lr_table, n_table);
  if (rc != REC_OK) return rc;
  CatchArgs a{col_uid, col_nu, B, F, table, V, m_e, v_e, ld_state, m_w, v_w, ld_wstate, last, step_dev, lr_table, n_table,
              b1, b2, eps};
  hipLaunchKernelGGL(adam_keras_catchup_kernel, dim3((unsigned)ceil_div64(B * F, CATCH_R)), dim3(CATCH_T), 0,
                     as_stream(stream), a);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

// every row of the table up to date (before the parameters are read from outside: evaluation, checkpoint), last = step
extern "C" int rec_adam_keras_flush_f32(float* table, int64_t ld, int64_t V, float* m_e, float* v_e, int64_t ld_state,
                                        float* m_w, float* v_w, int64_t ld_wstate, int32_t* last, const int64_t* step_dev,
                                        const float* lr_table, int64_t n_table, float b1, float b2, float eps,
                                        void* stream) {
  int rc = catchup_args_ok(table, ld, V, m_e, v_e, ld_state, m_w, v_w, ld_wstate, last, step_dev, lr_table, n_table);
  if (rc != REC_OK) return rc;
  CatchArgs a{nullptr, nullptr, 0, 0, table, V, m_e, v_e, ld_state, m_w, v_w, ld_wstate, last, step_dev, lr_table,
              n_table, b1, b2, eps};
  hipLaunchKernelGGL(adam_keras_catchup_kernel, dim3((unsigned)ceil_div64(V, CATCH_R)), dim3(CATCH_T), 0,
                     as_stream(stream), a);
  REC_LAUNCH_CHECK();
  hipLaunchKernelGGL(fill_last_kernel, dim3((unsigned)ceil_div64(V, 256)), dim3(256), 0, as_stream(stream), last, V,
                     step_dev);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" size_t rec_colsort_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  return 256;      // the sort runs in LDS; the argument is kept for callers written against the three-kernel version
}

static int colsort_plan(const int64_t* const* cols_host, int F, int64_t B, int64_t V, const int64_t* col_lo,
                        int64_t max_key, int32_t* perm, int64_t* col_uid, int32_t* col_seg, int32_t* col_nu,
                        int32_t* dloc, int* bad_flag, void* workspace, void* stream) {
  if (!cols_host || !col_lo || !perm || !col_uid || !col_seg || !col_nu || !workspace || F <= 0 || B <= 0 || V <= 0 ||
      max_key < 0)
    return REC_E_ARG;
  if (F > SORT_MAX_COLS || B > 16384) return REC_E_UNSUPPORTED;
  int pos_bits = 1, key_bits = 1;
  while ((int64_t(1) << pos_bits) < B) ++pos_bits;
  while ((int64_t(1) << key_bits) <= max_key) ++key_bits;
  if (key_bits + pos_bits > 32) return REC_E_UNSUPPORTED;
  // the pad word 0xFFFFFFFF must be larger than every real (key, position) word
  if ((((uint64_t)max_key << pos_bits) | (uint64_t)(B - 1)) >= 0xFFFFFFFFull) return REC_E_UNSUPPORTED;
  SortCols cp;
  for (int f = 0; f < F; ++f) {
    if (!cols_host[f]) return REC_E_ARG;
    cp.p[f] = cols_host[f];
  }
  ColSortArgs a{B, F, V, key_bits, pos_bits, perm, col_uid, col_seg, col_nu, bad_flag, dloc};
  hipStream_t st = as_stream(stream);
  // one workgroup per column (LDS radix sort + run heads in one launch); columns longer than 16 x 1024 do not occur
  // (B <= 16384)
  {
    const int kpt = B <= 8192 ? 8 : 16;
    // words + counters + the 16-bit staging arrays of the outputs (st_sg, st_dl)
    const size_t lds = sizeof(uint32_t) * ((size_t)OW_T * kpt + OW_W * OW_BINS / 2 + OW_BINS + OW_W) +
                       2 * sizeof(unsigned short) * (size_t)OW_T * kpt;
    hipError_t e;
    if (kpt == 8) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(colsort_onewg_kernel<8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(colsort_onewg_kernel<8>, dim3(F), dim3(OW_T), lds, st, cp, col_lo, a);
    } else {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(colsort_onewg_kernel<16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(colsort_onewg_kernel<16>, dim3(F), dim3(OW_T), lds, st, cp, col_lo, a);
    }
    REC_LAUNCH_CHECK();
    return REC_OK;
  }
}

extern "C" int rec_colsort_plan_i64(const int64_t* const* cols_host, int F, int64_t B, int64_t V, const int64_t* col_lo,
                                    int64_t max_key, int32_t* perm, int64_t* col_uid, int32_t* col_seg, int32_t* col_nu,
                                    int* bad_flag, void* workspace, void* stream) {
  return colsort_plan(cols_host, F, B, V, col_lo, max_key, perm, col_uid, col_seg, col_nu, nullptr, bad_flag, workspace,
                      stream);
}

extern "C" int rec_colsort_plan_dest_i64(const int64_t* const* cols_host, int F, int64_t B, int64_t V,
                                         const int64_t* col_lo, int64_t max_key, int32_t* perm, int64_t* col_uid,
                                         int32_t* col_seg, int32_t* col_nu, int32_t* dloc, int* bad_flag, void* workspace,
                                         void* stream) {
  if (!dloc) return REC_E_ARG;
  return colsort_plan(cols_host, F, B, V, col_lo, max_key, perm, col_uid, col_seg, col_nu, dloc, bad_flag, workspace,
                      stream);
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
               g_w_rows, n_uniq, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0.f, 0.f};
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
               (float*)nullptr, n_uniq, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0.f, 0.f};
  hipLaunchKernelGGL(colseg_sum_kernel, dim3((unsigned)ceil_div64(groups * 4, 256)), dim3(256), 0, as_stream(stream), k);
  REC_LAUNCH_CHECK();
  return REC_OK;
}
