// Fused DeepFM forward + backward kernel, third form (2.FM/CustomLayers.py:279-308 under 2.FM/ModelManager.py:171-177;
// embedding_dims 16, mlp_dims [32, 8], fused [embed 16 | w | pad] rows).  Same inputs, outputs and workspace layout as
// deepfm_fwd_bwd_kernel of deepfm_fused.hip (gz, IndexedSlices value rows, per-workgroup partials of the dense
// gradients) -- the post launches of that file finish the step -- but a different schedule.
//
// What the B sweep of round 3 showed (profiles/r03_b_sweep.json): the chip sustains ~50 G random 128-byte lines/s, one CU
// pulls ~25-30 GB/s from HBM (its 832 rows of a 32-example tile need ~4 us whatever the chip does), and the old kernel
// costs ~20 us PER TILE however many tiles a CU works through: ids -> rows -> layer 1 -> head -> backward is one
// dependent chain per workgroup, with the 0.65 GFLOP of fp32 MFMA (4.2 us per CU at the f32 rate) strictly behind the
// gather.  Here one workgroup still owns 32 examples and ONE set of dK0 accumulators, but works as two HALVES of 16
// examples (waves 0-3 = half A, waves 4-7 = half B, every SIMD hosts one wave of each) that run one phase apart:
//
//            half A (examples 0..15)                         half B (examples 16..31)
//   P0   ids -> rows (issued FIRST) -> layer 1 on MFMA        ids -> rows issued behind A's, nothing consumed
//   ---- barrier 1
//   P1   head (32->8->1, sigmoid, BCE, way back to dpre1)     layer 1 on MFMA as the rows land
//   ---- barrier 2
//   P2   dX(A) and dK0 += X_A^T dpre1_A on MFMA               head
//   ---- barrier 3
//   P3   dK0 += X_B^T dpre1_B, partials out                   dX(B); small partials
//
// so the matrix pipe works on A's backward while B's rows are still landing, B's head hides behind A's backward, and the
// dK0 accumulators of the whole 32-example tile stay in the registers of the A waves (as many partial bytes as before).
//
// Lane maps (j = lane & 15, q = lane >> 4), chosen so that a loaded 16-byte row piece IS an MFMA operand:
//   row loads      lane (j, q) reads floats 4q..4q+3 of the row of example j: one instruction = 16 rows x 64 B, all
//                  lanes useful (the old form read 8 rows x 128 B with 3 of 8 lanes idle); the first-order weight
//                  (float 16 of the same 128-byte line) comes with one extra dword load per four fields
//   layer 1        H1^T[unit][example] = K0_f^T . X_f^T on v_mfma_f32_16x16x4_f32: B operand of k-step s = component s
//                  of the loaded piece (k = 4q + s), A operand = K0T[unit][f*16 + 4q + s] -- one 16-byte load from a
//                  TRANSPOSED copy of K0 (rec_deepfm_k0t_f32; the natural layout needs 4-byte loads, 32 per lane)
//   dX^T           [dim][example] = K0_f . dpre1^T: the accumulator has the lane map of the loaded piece, so
//                  value row = dz (S - x) + dX needs no LDS
//   dK0^T          [unit][dim] = dpre1^T . X_f: the only product that needs X with examples on the k axis: rows are
//                  parked in LDS once (XT) and read back transposed, 4 bytes per lane and k-step, conflict-free
#include "common.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int EX = 32;        // examples per workgroup
constexpr int HEX = 16;       // examples per half
constexpr int E16 = 16;
constexpr int U1 = 32, U2 = 8;
constexpr int SMALL = 320;    // floats of small partials per workgroup (layout of deepfm_fused.hip)
constexpr int NWV = 8, HWV = 4;
constexpr int MAXF = 7;       // fields per wave (F <= 28 over the 4 waves of a half)
constexpr int HS1 = 36;       // row stride of H1s / DP1 / the partial layer-1 tiles (16-byte aligned rows)
constexpr int SPS = 20;       // row stride of the partial S tiles (conflict-free 16-byte stores)

struct Cols3 {
  const int64_t* p[REC_MAX_COLS];
};

struct F3Args {
  const float* table; int64_t V; int ld;
  const float* bias;
  const float* K0; const float* K0T; const float* b0;     // [F*16,32], [32,F*16], [32]
  const float* K1; const float* b1;                       // [32,8], [8]
  const float* K2; const float* b2;                       // [8,1], [1]
  const float* label;
  int64_t B; int F;
  float* gz; float* vals; float* prob;
  float* dK0part; float* small;
  int* oob;
  const int32_t* dloc; const int32_t* col_nu; float* g_embed;     // direct mode
  int64_t* step_dev; const float* lr_tab; int64_t n_tab; float* lr_t_dev;   // optional: the optimizer's step counter
#ifdef REC_FUSED_STAMPS
  unsigned long long* stamps;
#endif
};

#ifdef REC_FUSED_STAMPS
#define STAMP3(k)                                                                                   \
  do {                                                                                              \
    if (lane == 0) a.stamps[((int64_t)blockIdx.x * NWV + wave) * 12 + (k)] = wall_clock64();        \
  } while (0)
#else
#define STAMP3(k) do {} while (0)
#endif

constexpr int KS = 36;        // row stride of K0 in LDS: 16-byte rows, (KS/4) odd -> conflict-free 16-byte row reads
struct Carve3 {
  int K0s, XT, PT, SP, QP, WP, H1s, DP1, Ss, h2s, dp2s, dzs, lss, K1s, b0s, b1s, K2s, flags, total;
};
__host__ __device__ inline Carve3 carve3_of(int F, bool klds) {
  Carve3 c;
  int o = 0;
  c.K0s = o; o += klds ? F * E16 * KS : 0;     // [16F][36]     layer 1's kernel, staged once per workgroup
  c.XT = o; o += EX * (F * E16 + 4);          // [32][16F+4]   rows (example-major), read back transposed for dK0
  c.PT = o; o += NWV * HEX * HS1;             // [8][16][36]   partial layer-1 tiles, one per wave
  c.SP = o; o += NWV * HEX * SPS;             // [8][16][20]   partial S = sum of rows over the wave's fields
  c.QP = o; o += NWV * HEX;                   // [8][16]       partial sum of squares
  c.WP = o; o += NWV * HEX;                   // [8][16]       partial first-order sum
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
  c.flags = o; o += 8;
  c.total = o;
  return c;
}

// Workgroup barrier for LDS hand-overs ONLY: the wave's own LDS operations are drained (lgkmcnt), its global loads are
// NOT (a __syncthreads() / workgroup fence waits vmcnt(0): half B crosses barrier 1 with all its row loads in flight,
// which is the point of the schedule).  Global stores of the kernel are never read back inside it.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Sync of the FOUR waves of one half (gfx950 has one hardware barrier per workgroup, and it would couple the halves: a B
// wave stuck behind its row-load issue would hold back A's head).  Arrival counter in LDS, zeroed before barrier 0 and
// used once per launch: a wave drains its LDS writes, lane 0 adds 1, everyone polls until all four have arrived.  LDS
// operations of a wave complete in program order, so data written before the add is visible to whoever saw the count.
// The poll is bounded (a lost wave would otherwise hang the GPU): it cannot expire in a correct run.
__device__ __forceinline__ void half_sync(int* cnt, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll 1
  for (int it = 0; it < (1 << 22); ++it) {
    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= HWV) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// KLDS: layer 1's kernel K0 [16F, 32] is staged ONCE per workgroup in LDS (F <= 26: 60 KB beside the 98 KB of the rest) and
// every operand fragment made of it -- layer 1's A operand (k-major: four 4-byte reads per step group) and dX's A
// operand (row-major: two 16-byte reads) -- comes from there.  Measured (profiles/r03_fused3_stamps.txt): fetched as
// per-wave fragments from L2 (the !KLDS form: 14 + 14 16-byte loads per lane, 224 KB per workgroup against 106 KB of
// rows) those loads delayed the ids by 2 us and the whole memory phase by 4 us -- they share the CU's one address path
// and in-order return queue with the row gather.
// 16-byte store with the sc1 bit (write-through: the line does not stay dirty in the XCD's L2) through a buffer
// resource -- the builtin is modelled by the compiler (waits and MFMA -> store hazards are its business; an inline-asm
// store right behind an MFMA read garbage accumulators).  The 27 MB a launch writes (value rows + dK0 partials) otherwise
// sit dirty in L2 until the kernel boundary writes them back in one burst: 19.2 -> 17.5 us per launch.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(__amdgpu_buffer_rsrc_t rs, unsigned off_bytes, const float4& v) {
  u32x4 t;
  t[0] = __float_as_uint(v.x); t[1] = __float_as_uint(v.y); t[2] = __float_as_uint(v.z); t[3] = __float_as_uint(v.w);
#ifdef ABL_PLAIN_STORES
  __builtin_amdgcn_raw_buffer_store_b128(t, rs, off_bytes, 0, 0);
#else
  __builtin_amdgcn_raw_buffer_store_b128(t, rs, off_bytes, 0, 16);
#endif
}

template <bool DIRECT, bool KLDS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void deepfm3_kernel(Cols3 cols, F3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int F = a.F, D = F * E16, XS = F * E16 + 4;
  const Carve3 cv = carve3_of(F, KLDS);
  float* K0s = lds + cv.K0s;
  float* XT = lds + cv.XT;
  float* PT = lds + cv.PT;
  float* SP = lds + cv.SP;
  float* QP = lds + cv.QP;
  float* WP = lds + cv.WP;
  float* H1s = lds + cv.H1s;
  float* DP1 = lds + cv.DP1;
  float* Ss = lds + cv.Ss;
  float* h2s = lds + cv.h2s;
  float* dp2s = lds + cv.dp2s;
  float* dzs = lds + cv.dzs;
  float* lss = lds + cv.lss;
  float* K1s = lds + cv.K1s;
  float* b0s = lds + cv.b0s;
  float* b1s = lds + cv.b1s;
  float* K2s = lds + cv.K2s;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, hw = wave & 3;               // wave-uniform
  const int j = lane & 15, q = lane >> 4;
  const int64_t ex0 = (int64_t)blockIdx.x * EX;
  const int n_ex = (a.B - ex0 < EX) ? (int)(a.B - ex0) : EX;
  const int eh = HEX * half + j;                           // this lane's example inside the workgroup
  const bool ex_live = eh < n_ex;
  const int64_t e_glob = ex0 + (ex_live ? eh : n_ex - 1);
  const int64_t ldr = DIRECT ? 32 : a.ld;
  const int th_ = tid & 255;                               // thread index inside the half
  STAMP3(0);

  // the device-side step counter of the optimizer (rec_adam_advance_f32's job, without its launch): this kernel does not
  // read it; the catch-up kernel before it needs the old value, the post launch and the dense update after it the new one
  if (a.step_dev && blockIdx.x == 0 && tid == 0) {
    const int64_t s = *a.step_dev + 1;
    *a.step_dev = s;
    *a.lr_t_dev = a.lr_tab[(s < a.n_tab ? s : a.n_tab) - 1];
  }
  int* sync_cnt = reinterpret_cast<int*>(lds + cv.flags);   // arrival counters of the half-workgroup syncs
  if (tid < 8) sync_cnt[tid] = 0;                          // (first used behind barrier 0)

  // ---- KLDS: the B waves fetch K0 (53 KB, coalesced, 13 pieces per thread at F = 26) -- they have nothing else to do
  // until A's row loads are queued -- and park it in LDS; the A waves' path to their rows carries no K0 work at all
  // (thirteen named registers, not an array: an array written in one conditional region and read in another is kept in
  // scratch memory by hipcc, and scratch accesses count in vmcnt -- the wait for them would cover the ids in flight)
#define KST_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#define KST_DECL(k) float4 kst##k = make_float4(0.f, 0.f, 0.f, 0.f);
#define KST_LOAD(k)                                                                              \
  {                                                                                              \
    const int p_ = th_ + 256 * k;                                                                \
    kst##k = *reinterpret_cast<const float4*>(a.K0 + 4 * (p_ < n4_ ? p_ : n4_ - 1));              \
  }
#define KST_PARK(k)                                                                              \
  {                                                                                              \
    const int p0_ = th_ + 256 * k;                                                               \
    const int p_ = p0_ < n4_ ? p0_ : n4_ - 1;                                                    \
    *reinterpret_cast<float4*>(K0s + (p_ >> 3) * KS + 4 * (p_ & 7)) = kst##k;                     \
  }
  KST_LIST(KST_DECL)
  const int n4_ = D * (U1 / 4);
  if constexpr (KLDS) {
    if (half == 1) { KST_LIST(KST_LOAD) }
  }
  // ---- loads that depend on nothing: ids (they head the longest chain), plan slots, the head's scalars.  All
  // unconditional (clamped addresses), judged afterwards.
  int64_t idr[MAXF];
#pragma unroll
  for (int i = 0; i < MAXF; ++i) {
    const int f = hw + HWV * i;
    idr[i] = cols.p[f < F ? f : F - 1][e_glob];
  }
  int dlr[MAXF];
  if (DIRECT) {
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      dlr[i] = a.dloc[(int64_t)(f < F ? f : F - 1) * a.B + e_glob];
    }
  }
  // runs per column of the plan (direct mode): lane c holds col_nu[c]; the prefix over the columns is a wave scan
  int nu_l = 0;
  if (DIRECT) nu_l = a.col_nu[lane < F ? lane : 0];
  // !KLDS (F > 26: K0 does not fit in LDS beside the rows): A operand of layer 1 per wave from the transposed copy,
  // ka[i][n] component s = K0[f*16 + 4q + s][16n + j]
  float4 ka[MAXF][2];
  if constexpr (!KLDS) {
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      const int fc = f < F ? f : F - 1;
#pragma unroll
      for (int n = 0; n < 2; ++n)
        ka[i][n] = *reinterpret_cast<const float4*>(a.K0T + (int64_t)(16 * n + j) * D + fc * E16 + 4 * q);
    }
  }
  const int e16 = th_ >> 4, g16 = th_ & 15;                // head: 16 lanes per example
  const int e32 = HEX * half + e16;
  const float bias_r = a.bias[0], b2_r = a.b2[0];
  const float inv_B = 1.f / (float)a.B;
  const float label_r = a.label[ex0 + (e32 < n_ex ? e32 : n_ex - 1)];
  // small dense operands of the head: loaded unconditionally (a load inside a guarded block is waited for on the spot,
  // together with everything issued before it), parked in LDS behind the row loads
  const float k1_r = a.K1[tid & (U1 * U2 - 1)], b0_r = a.b0[tid & (U1 - 1)], b1_r = a.b1[tid & (U2 - 1)],
              k2_r = a.K2[tid & (U2 - 1)];

  // ---- barrier 0, B side: K0 is in LDS, and -- because the A waves arrive only after queueing their row loads -- B's
  // row loads go out behind A's (a CU pulls ~25-30 GB/s: whoever queues first is served first, so A's rows are complete
  // about 2 us before B's)
  if (half == 1) {
    if constexpr (KLDS) { KST_LIST(KST_PARK) }             // unconditional stores: pieces past the end repeat the last one
    wg_barrier();
  }
  bool ok[MAXF];
  bool bad = false;
  float4 v[MAXF];
#ifdef REC_FUSED_STAMPS
  if (idr[MAXF - 1] == -12345) a.stamps[0] = 0;            // (diagnostic builds) forces the wait for the ids here
  STAMP3(8);
#endif
#pragma unroll
  for (int i = 0; i < MAXF; ++i) {
    const int f = hw + HWV * i;
    const bool live = f < F && ex_live;
    const bool inr = (uint64_t)idr[i] < (uint64_t)a.V;
    bad |= live && !inr;
    ok[i] = live && inr;
    const int64_t row = ok[i] ? idr[i] : 0;
    v[i] = *reinterpret_cast<const float4*>(a.table + row * ldr + 4 * q);
  }
  // first-order weights: lane (j, q) takes field slots q and 4 + q of its example
  float wv0, wv1;
  {
    const int64_t i0 = q == 0 ? idr[0] : q == 1 ? idr[1] : q == 2 ? idr[2] : idr[3];
    const int64_t i1 = q == 0 ? idr[4] : q == 1 ? idr[5] : idr[6];
    const bool ok0 = (hw + HWV * q < F) && ex_live && (uint64_t)i0 < (uint64_t)a.V;
    const bool ok1 = q < 3 && (hw + HWV * (4 + q) < F) && ex_live && (uint64_t)i1 < (uint64_t)a.V;
    const float x0 = a.table[(ok0 ? i0 : 0) * ldr + E16];
    const float x1 = a.table[(ok1 ? i1 : 0) * ldr + E16];
    wv0 = ok0 ? x0 : 0.f;
    wv1 = ok1 ? x1 : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMP3(1);
  if (bad && a.oob) *a.oob = 1;
  if (tid < U1 * U2) K1s[tid] = k1_r;                      // tid < 256: the A waves, before their barrier 0
  if (tid < U1) b0s[tid] = b0_r;
  if (tid < U2) { b1s[tid] = b1_r; K2s[tid] = k2_r; }
  if (half == 0) wg_barrier();                             // barrier 0, A side: every row load of the A waves is queued
  STAMP3(9);
  // plan slots of the value rows (direct mode): run index inside the column + runs of the columns before
  int dl[MAXF];
  if (DIRECT) {
    int incl = lane < F ? nu_l : 0;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    const int excl = incl - (lane < F ? nu_l : 0);          // runs in the columns before column `lane`
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      const int before = __shfl(excl, f < F ? f : 0, 64);
      dl[i] = dlr[i] >= 0 ? dlr[i] + before : dlr[i];
    }
  }

  // ================================ layer 1 of this half (P0 for A, P1 for B) ================================
  {
    f32x4 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) { acc[n][0] = 0.f; acc[n][1] = 0.f; acc[n][2] = 0.f; acc[n][3] = 0.f; }
    float4 S4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      if (f < F) {                                          // wave-uniform
        if (!ok[i]) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 x = v[i];
        *reinterpret_cast<float4*>(XT + eh * XS + f * E16 + 4 * q) = x;
        S4.x += x.x; S4.y += x.y; S4.z += x.z; S4.w += x.w;
        sq += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float k0, k1;
          if constexpr (KLDS) {                            // lanes j: 16 consecutive floats; q: rows 4 apart = 16 banks apart
            k0 = K0s[(f * E16 + 4 * q + s) * KS + j];
            k1 = K0s[(f * E16 + 4 * q + s) * KS + 16 + j];
          } else {
            k0 = s == 0 ? ka[i][0].x : s == 1 ? ka[i][0].y : s == 2 ? ka[i][0].z : ka[i][0].w;
            k1 = s == 0 ? ka[i][1].x : s == 1 ? ka[i][1].y : s == 2 ? ka[i][1].z : ka[i][1].w;
          }
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0, xs[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, xs[s], acc[1], 0, 0, 0);
        }
      }
    }
    // accumulator lane map: lane (j, q), register r = H1pre[example j][unit 16n + 4q + r]
    float* pt = PT + wave * (HEX * HS1) + j * HS1 + 4 * q;
    *reinterpret_cast<float4*>(pt) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    *reinterpret_cast<float4*>(pt + 16) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
    *reinterpret_cast<float4*>(SP + wave * (HEX * SPS) + j * SPS + 4 * q) = S4;
    float wsum = wv0 + wv1;
    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
    wsum += __shfl_xor(wsum, 16, 64); wsum += __shfl_xor(wsum, 32, 64);
    if (q == 0) { QP[wave * HEX + j] = sq; WP[wave * HEX + j] = wsum; }
  }
  // A operand of dX^T for every owned field: K0[f*16 + j][8q .. 8q+8) (natural layout, two 16-byte loads), requested now:
  // they land during the head, and the backward phases then run on registers and LDS alone
  float4 kt[MAXF][2];
  if constexpr (!KLDS) {
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      const float* kp = a.K0 + ((f < F ? f : 0) * E16 + j) * U1 + 8 * q;
      kt[i][0] = *reinterpret_cast<const float4*>(kp);
      kt[i][1] = *reinterpret_cast<const float4*>(kp + 4);
    }
  }
  // KLDS: the same fragment straight from LDS (row f*16 + j, floats 8q .. 8q+7: two conflict-free 16-byte reads)
  auto kt_of = [&](int i, int f, float* kv) {
    float4 t0, t1;
    if constexpr (KLDS) {
      t0 = *reinterpret_cast<const float4*>(K0s + (f * E16 + j) * KS + 8 * q);
      t1 = *reinterpret_cast<const float4*>(K0s + (f * E16 + j) * KS + 8 * q + 4);
    } else {
      t0 = kt[i][0]; t1 = kt[i][1];
    }
    kv[0] = t0.x; kv[1] = t0.y; kv[2] = t0.z; kv[3] = t0.w; kv[4] = t1.x; kv[5] = t1.y; kv[6] = t1.z; kv[7] = t1.w;
  };
  STAMP3(2);
  half_sync(sync_cnt + 2 * half, lane);                    // the four waves of this half: partial tiles are in LDS
  STAMP3(3);

  // ================================ head of this half (P1 for A, P2 for B) ================================
  {
    const float* ptb = PT + (HWV * half) * (HEX * HS1) + e16 * HS1 + 2 * g16;
    float2 h = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < HWV; ++w) {
      const float2 p = *reinterpret_cast<const float2*>(ptb + w * (HEX * HS1));
      h.x += p.x; h.y += p.y;
    }
    const float h1a = fmaxf(h.x + b0s[2 * g16], 0.f), h1b = fmaxf(h.y + b0s[2 * g16 + 1], 0.f);
    H1s[e32 * HS1 + 2 * g16] = h1a;
    H1s[e32 * HS1 + 2 * g16 + 1] = h1b;
    float sd = 0.f, sqs = 0.f, fo = 0.f;
#pragma unroll
    for (int w = 0; w < HWV; ++w) {
      sd += SP[(HWV * half + w) * (HEX * SPS) + e16 * SPS + g16];
      sqs += QP[(HWV * half + w) * HEX + e16];
      fo += WP[(HWV * half + w) * HEX + e16];
    }
    Ss[e32 * E16 + g16] = sd;
    float t = sd * sd;
    float h2[U2];
    const float4 ka0 = *reinterpret_cast<const float4*>(K1s + (2 * g16) * U2);
    const float4 ka1 = *reinterpret_cast<const float4*>(K1s + (2 * g16) * U2 + 4);
    const float4 kb0 = *reinterpret_cast<const float4*>(K1s + (2 * g16 + 1) * U2);
    const float4 kb1 = *reinterpret_cast<const float4*>(K1s + (2 * g16 + 1) * U2 + 4);
    h2[0] = h1a * ka0.x + h1b * kb0.x; h2[1] = h1a * ka0.y + h1b * kb0.y;
    h2[2] = h1a * ka0.z + h1b * kb0.z; h2[3] = h1a * ka0.w + h1b * kb0.w;
    h2[4] = h1a * ka1.x + h1b * kb1.x; h2[5] = h1a * ka1.y + h1b * kb1.y;
    h2[6] = h1a * ka1.z + h1b * kb1.z; h2[7] = h1a * ka1.w + h1b * kb1.w;
    t = row16_allsum(t) - sqs;
#pragma unroll
    for (int u = 0; u < U2; ++u) h2[u] = row16_allsum(h2[u]);
    float dnn = 0.f;
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      h2[u] = fmaxf(h2[u] + b1s[u], 0.f);
      dnn += h2[u] * K2s[u];
    }
    const bool valid = e32 < n_ex;
    const float z = (bias_r + fo + 0.5f * t) + dnn + b2_r;
    const float p = sigmoid_acc(z);
    const float y = valid ? label_r : 0.f;
    const float eps = 1e-7f;
    const float pc = fminf(fmaxf(p, eps), 1.f - eps);
    float le = -(y * __logf(pc + eps) + (1.f - y) * __logf(1.f - pc + eps));
    const float inside = (p >= eps && p <= 1.f - eps) ? 1.f : 0.f;
    float dz = -(y * __builtin_amdgcn_rcpf(pc + eps) - (1.f - y) * __builtin_amdgcn_rcpf(1.f - pc + eps)) * inside * p *
               (1.f - p) * inv_B;
    if (!valid) { dz = 0.f; le = 0.f; }
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
    DP1[e32 * HS1 + 2 * g16] = h1a > 0.f ? dha : 0.f;
    DP1[e32 * HS1 + 2 * g16 + 1] = h1b > 0.f ? dhb : 0.f;
    if (g16 == 0) {
      *reinterpret_cast<float4*>(h2s + e32 * U2) = make_float4(h2[0], h2[1], h2[2], h2[3]);
      *reinterpret_cast<float4*>(h2s + e32 * U2 + 4) = make_float4(h2[4], h2[5], h2[6], h2[7]);
      *reinterpret_cast<float4*>(dp2s + e32 * U2) = make_float4(dp2[0], dp2[1], dp2[2], dp2[3]);
      *reinterpret_cast<float4*>(dp2s + e32 * U2 + 4) = make_float4(dp2[4], dp2[5], dp2[6], dp2[7]);
      dzs[e32] = dz;
      lss[e32] = le;
      if (valid) {
        a.gz[ex0 + e32] = dz;
        if (a.prob) a.prob[ex0 + e32] = p;
      }
    }
  }
  if (half == 0) half_sync(sync_cnt + 1, lane);            // A: dpre1 / S / dz of A's examples are in LDS
  else wg_barrier();                                       // B: barrier 3 (B side) -- the A waves take B's dpre1 from here
  STAMP3(5);

  // ================================ backward ================================
  // dX of this half by its own waves: dX^T [dim][example] = K0_f [dim][unit] . dpre1^T [unit][example]
  //   A  kt[s]  = K0[f*16 + j][8q + s]            (lane: dim j, k slot q; unit = 8q + s)
  //   B  aP[s]  = dpre1[example j][8q + s]        (the same for every field)
  //   D  lane (j = example, q): dims 4q .. 4q+3 -- the lane map of the loaded piece v[i]
  float aP[8];
  {
    const float4 p0 = *reinterpret_cast<const float4*>(DP1 + eh * HS1 + 8 * q);
    const float4 p1 = *reinterpret_cast<const float4*>(DP1 + eh * HS1 + 8 * q + 4);
    aP[0] = p0.x; aP[1] = p0.y; aP[2] = p0.z; aP[3] = p0.w; aP[4] = p1.x; aP[5] = p1.y; aP[6] = p1.z; aP[7] = p1.w;
  }
  const float dzr = dzs[eh];
  const float4 sr = *reinterpret_cast<const float4*>(Ss + eh * E16 + 4 * q);

  // buffer resources of the three outputs (wave-uniform bases; offsets are 32-bit: the launcher refuses larger arrays)
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(a.vals, 0, 0x7FFFFFFC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(DIRECT ? a.g_embed : a.vals, 0, 0x7FFFFFFC, 0x00020000);
  auto store_row = [&](int i, int f, const f32x4& dx) {
    if (ex_live) {
      float4 o;
      o.x = dzr * (sr.x - v[i].x) + dx[0];
      o.y = dzr * (sr.y - v[i].y) + dx[1];
      o.z = dzr * (sr.z - v[i].z) + dx[2];
      o.w = dzr * (sr.w - v[i].w) + dx[3];
#ifdef ABL_NOVALS
      asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#else
      if (DIRECT && dl[i] >= 0)
        store16_wt(rs_g, (unsigned)(dl[i] * E16 + 4 * q) * 4u, o);
      else
        store16_wt(rs_v, (unsigned)(((ex0 + eh) * F + f) * E16 + 4 * q) * 4u, o);
#endif
    }
  };

  if (half == 0) {
    // ---- P2: dX(A) and dK0 of A's examples; P3: dK0 of B's examples.  dK0^T [unit][dim] = dpre1^T . X_f:
    //   A  bP[n][s] = dpre1[example 4q + s][16n + j]      B  xk[s] = X[example 4q + s][f*16 + j]
    //   D  lane (j = dim, q): units 16n + 4q .. +3 of row f*16 + j of dK0
    f32x4 dk[MAXF][2];
#pragma unroll
    for (int i = 0; i < MAXF; ++i)
#pragma unroll
      for (int n = 0; n < 2; ++n) { dk[i][n][0] = 0.f; dk[i][n][1] = 0.f; dk[i][n][2] = 0.f; dk[i][n][3] = 0.f; }
    float bP[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int s = 0; s < 4; ++s) bP[n][s] = DP1[(4 * q + s) * HS1 + 16 * n + j];
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      if (f < F) {                                          // wave-uniform
        float kv[8];
        kt_of(i, f, kv);
        float xk[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xk[s] = XT[(4 * q + s) * XS + f * E16 + j];
        f32x4 dx;
        dx[0] = 0.f; dx[1] = 0.f; dx[2] = 0.f; dx[3] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                       // three independent chains, interleaved
          dx = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[2 * s], aP[2 * s], dx, 0, 0, 0);
          dk[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[0][s], xk[s], dk[i][0], 0, 0, 0);
          dx = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[2 * s + 1], aP[2 * s + 1], dx, 0, 0, 0);
          dk[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[1][s], xk[s], dk[i][1], 0, 0, 0);
        }
        store_row(i, f, dx);
      }
    }
    STAMP3(6);
    wg_barrier();                                          // barrier 3 (A side)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int s = 0; s < 4; ++s) bP[n][s] = DP1[(HEX + 4 * q + s) * HS1 + 16 * n + j];
    float* part = a.dK0part + (int64_t)blockIdx.x * D * U1;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(part, 0, (unsigned)(D * U1 * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < MAXF; ++i) {
      const int f = hw + HWV * i;
      if (f < F) {
        float xk[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xk[s] = XT[(HEX + 4 * q + s) * XS + f * E16 + j];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          dk[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[0][s], xk[s], dk[i][0], 0, 0, 0);
          dk[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bP[1][s], xk[s], dk[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#ifdef ABL_NOPART
          asm volatile("" :: "v"(dk[i][n][0]), "v"(dk[i][n][1]), "v"(dk[i][n][2]), "v"(dk[i][n][3]));
          (void)part;
#else
          store16_wt(rs_p, (unsigned)((f * E16 + j) * U1 + 16 * n + 4 * q) * 4u,
                     make_float4(dk[i][n][0], dk[i][n][1], dk[i][n][2], dk[i][n][3]));
#endif
        }
      }
    }
  } else {
    // ---- P3: dX(B), two fields at a time (two independent accumulator chains)
#pragma unroll
    for (int i = 0; i < MAXF; i += 2) {
      const int f = hw + HWV * i, f2 = f + HWV;
      if (f < F) {                                          // wave-uniform
        const int i2 = (i + 1 < MAXF) ? i + 1 : i;
        const bool two = (i + 1 < MAXF) && f2 < F;
        float kv[8], ku[8];
        kt_of(i, f, kv);
        kt_of(i2, two ? f2 : f, ku);
        f32x4 dxa, dxb;
        dxa[0] = 0.f; dxa[1] = 0.f; dxa[2] = 0.f; dxa[3] = 0.f;
        dxb[0] = 0.f; dxb[1] = 0.f; dxb[2] = 0.f; dxb[3] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          dxa = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[s], aP[s], dxa, 0, 0, 0);
          dxb = __builtin_amdgcn_mfma_f32_16x16x4f32(ku[s], aP[s], dxb, 0, 0, 0);
        }
        store_row(i, f, dxa);
        if (two) store_row(i2, f2, dxb);
      }
    }
    STAMP3(6);
  }

  // ---- small per-workgroup partials over the 32 examples (fixed order), as in deepfm_fused.hip: after barrier 3
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
  STAMP3(7);
}

// K0T [32][F*16] = K0^T: 32 x 32 tiles through LDS (coalesced both ways); 53 KB at F = 26
__global__ __launch_bounds__(256) void k0_transpose_kernel(const float* __restrict__ K0, int D, float* __restrict__ K0T) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + ty + 8 * p;
    tile[ty + 8 * p][tx] = r < D ? K0[(int64_t)r * U1 + tx] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int u = ty + 8 * p, r = r0 + tx;
    if (r < D) K0T[(int64_t)u * D + r] = tile[tx][u];
  }
}

}  // namespace

#ifdef REC_FUSED_STAMPS
static unsigned long long* g_fused3_stamps = nullptr;
extern "C" int rec_debug_fused3_stamps(unsigned long long* host_out, int nwg) {
  if (!g_fused3_stamps) return REC_E_ARG;
  return (int)hipMemcpy(host_out, g_fused3_stamps, sizeof(unsigned long long) * 12 * NWV * (size_t)nwg, hipMemcpyDeviceToHost);
}
#endif

extern "C" int rec_deepfm_k0t_f32(const float* K0, int F, float* K0T, void* stream) {
  if (!K0 || !K0T || F <= 0) return REC_E_ARG;
  const int D = F * E16;
  hipLaunchKernelGGL(k0_transpose_kernel, dim3((unsigned)ceil_div64(D, 32)), dim3(256), 0, as_stream(stream), K0, D, K0T);
  REC_LAUNCH_CHECK();
  return REC_OK;
}

static int launch_fused3(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host, int F, int64_t B,
                         const float* bias, const float* K0, const float* K0T, const float* b0, const float* K1,
                         const float* b1, const float* K2, const float* b2, const float* label, float* gz, float* vals,
                         float* prob, int* oob_flag, void* workspace, const int32_t* dloc, const int32_t* col_nu,
                         float* g_embed, bool direct, void* stream, int64_t* step_dev = nullptr,
                         const float* lr_tab = nullptr, int64_t n_tab = 0, float* lr_t_dev = nullptr) {
  if (B <= 0 || F <= 0 || V <= 0) return REC_E_ARG;
  if (step_dev && (!lr_tab || !lr_t_dev || n_tab <= 0)) return REC_E_ARG;
  if (F > 28 || F > REC_MAX_COLS || V >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;
  if (direct ? ld != 32 : (ld < 20 || (ld & 3) != 0)) return REC_E_UNSUPPORTED;
  if (!table || !cols_host || !bias || !K0 || !K0T || !b0 || !K1 || !b1 || !K2 || !b2 || !label || !gz || !vals ||
      !workspace)
    return REC_E_ARG;
  if (direct && (!dloc || !col_nu || !g_embed)) return REC_E_ARG;
  if (B * F * E16 * 4 >= (int64_t(1) << 31)) return REC_E_UNSUPPORTED;    // 32-bit byte offsets into vals / g_embed_rows
  if ((reinterpret_cast<uintptr_t>(table) & 15) != 0 || (reinterpret_cast<uintptr_t>(K0) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(K0T) & 15) != 0 || (reinterpret_cast<uintptr_t>(vals) & 15) != 0 ||
      (direct && (reinterpret_cast<uintptr_t>(g_embed) & 15) != 0))
    return REC_E_UNSUPPORTED;
  // K0 in LDS whenever it fits beside the rest (F <= 26); else its fragments come from L2 (K0 and K0T)
  bool klds = F <= 26;                                   // the B waves stage 13 x 256 16-byte pieces: 8 * 16 F <= 3328
  size_t lds = (size_t)carve3_of(F, true).total * sizeof(float);
  if (!klds || lds > 160 * 1024) {
    klds = false;
    lds = (size_t)carve3_of(F, false).total * sizeof(float);
  }
  if (lds > 160 * 1024) return REC_E_UNSUPPORTED;
  Cols3 cp;
  for (int f = 0; f < F; ++f) {
    if (!cols_host[f]) return REC_E_ARG;
    cp.p[f] = cols_host[f];
  }
  const int nwg = (int)ceil_div64(B, EX);
  float* dK0part = (float*)workspace;
  float* small = dK0part + (size_t)nwg * F * E16 * U1;
#ifdef REC_FUSED_STAMPS
  static unsigned long long* stamps = nullptr;
  if (!stamps && hipMalloc(&stamps, sizeof(unsigned long long) * 12 * NWV * 65536) != hipSuccess) return REC_E_ARG;
  g_fused3_stamps = stamps;
  F3Args a{table, V, (int)ld, bias, K0, K0T, b0, K1, b1, K2, b2, label, B, F, gz, vals, prob, dK0part, small, oob_flag,
           dloc, col_nu, g_embed, step_dev, lr_tab, n_tab, lr_t_dev, stamps};
#else
  F3Args a{table, V, (int)ld, bias, K0, K0T, b0, K1, b1, K2, b2, label, B, F, gz, vals, prob, dK0part, small, oob_flag,
           dloc, col_nu, g_embed, step_dev, lr_tab, n_tab, lr_t_dev};
#endif
  hipStream_t st = as_stream(stream);
#define LAUNCH3(DIR, KL)                                                                                         \
  do {                                                                                                          \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(deepfm3_kernel<DIR, KL>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    hipLaunchKernelGGL((deepfm3_kernel<DIR, KL>), dim3(nwg), dim3(512), lds, st, cp, a);                        \
  } while (0)
  if (direct) {
    if (klds) LAUNCH3(true, true); else LAUNCH3(true, false);
  } else {
    if (klds) LAUNCH3(false, true); else LAUNCH3(false, false);
  }
#undef LAUNCH3
  REC_LAUNCH_CHECK();
  return REC_OK;
}

extern "C" int rec_deepfm_fused3_main_f32(const float* table, int64_t ld, int64_t V, const int64_t* const* cols_host,
                                          int F, int64_t B, const float* bias, const float* K0, const float* K0T,
                                          const float* b0, const float* K1, const float* b1, const float* K2,
                                          const float* b2, const float* label, float* gz, float* vals, float* prob,
                                          int* oob_flag, void* workspace, void* stream) {
  return launch_fused3(table, ld, V, cols_host, F, B, bias, K0, K0T, b0, K1, b1, K2, b2, label, gz, vals, prob, oob_flag,
                       workspace, nullptr, nullptr, nullptr, false, stream);
}

extern "C" int rec_deepfm_fused3_main_direct_f32(const float* table, int64_t ld, int64_t V,
                                                 const int64_t* const* cols_host, int F, int64_t B, const float* bias,
                                                 const float* K0, const float* K0T, const float* b0, const float* K1,
                                                 const float* b1, const float* K2, const float* b2, const float* label,
                                                 float* gz, float* vals, float* prob, int* oob_flag, void* workspace,
                                                 const int32_t* dloc, const int32_t* col_nu, float* g_embed_rows,
                                                 void* stream) {
  return launch_fused3(table, ld, V, cols_host, F, B, bias, K0, K0T, b0, K1, b1, K2, b2, label, gz, vals, prob, oob_flag,
                       workspace, dloc, col_nu, g_embed_rows, true, stream);
}

// rec_deepfm_fused3_main_direct_f32 + rec_adam_advance_f32 in one launch: *step_dev += 1 and *lr_t_dev =
// lr_table[min(*step_dev, n_table) - 1] are done by the fused kernel's first thread (the kernel itself reads neither), so the
// post launch and the dense update behind it see the new step, the catch-up kernel before it saw the old one.
extern "C" int rec_deepfm_fused3_main_direct_adv_f32(const float* table, int64_t ld, int64_t V,
                                                     const int64_t* const* cols_host, int F, int64_t B, const float* bias,
                                                     const float* K0, const float* K0T, const float* b0, const float* K1,
                                                     const float* b1, const float* K2, const float* b2, const float* label,
                                                     float* gz, float* vals, float* prob, int* oob_flag, void* workspace,
                                                     const int32_t* dloc, const int32_t* col_nu, float* g_embed_rows,
                                                     int64_t* step_dev, const float* lr_table, int64_t n_table,
                                                     float* lr_t_dev, void* stream) {
  if (!step_dev || !lr_table || !lr_t_dev || n_table <= 0) return REC_E_ARG;
  return launch_fused3(table, ld, V, cols_host, F, B, bias, K0, K0T, b0, K1, b1, K2, b2, label, gz, vals, prob, oob_flag,
                       workspace, dloc, col_nu, g_embed_rows, true, stream, step_dev, lr_table, n_table, lr_t_dev);
}
