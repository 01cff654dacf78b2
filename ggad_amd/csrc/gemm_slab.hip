// Tall product against a small second operand, round 5: C[M x N] = A[M x K] op(B) (+ bias, ReLU) for the projections of the full-graph
// path (reference `model.py:27` -- nn.Linear inside every GCN layer -- and its data gradient; M = 7.5 K ... 39 K rows, N = K = 300).
//
// What three rounds of tiled kernels measured (profiles/r04_gemm_mfma_pmc.csv): 0.32-0.52 of the FP32 matrix peak whatever the
// staging, because (a) 64 x 64 workgroup tiles quantise the grid (860 tiles on 256 CUs = 3.36 -> 4 rounds, and N = 300 pads to 320),
// (b) the compiler turned the accumulator chains of the resident-slab kernel (k_gemm_bres, two accumulators) into runs of DEPENDENT
// v_mfma_f32_16x16x4_f32 (40 cycles each instead of 32) with s_nop hazards around renamed accumulators, (c) every workgroup's
// prologue / epilogue is exposed.  This kernel changes the decomposition, not the staging:
//
//   * ONE persistent workgroup per CU (8 waves = two per SIMD) keeps a column SLAB of op(B) -- 80 or 64 columns x all of K, K-fast rows of
//     KP floats -- in LDS for its whole life: N = 300 is 19 column tiles of 16 (304, not 320), dealt as slabs of 5 + 5 + 5 + 4 tiles,
//     and the CUs of every XCD are dealt to the slabs in proportion to their tiles.
//   * the unit of work is one 16-row block of A against the slab (TC tiles x K / 4 MFMAs = 375 at K = 300, 12 K cycles): the row
//     blocks of an XCD's eighth are dealt round-robin to the SIMDs of the slab's CUs, and the two waves of a SIMD alternate, so
//     every SIMD gets floor or ceil of an equal share (Reddit: 2.7 -> 3 units of 5 us; the 64 x 64 tiles: 3.4 -> 4 of 4 us plus padding).
//   * a wave owns its units outright: A comes straight from memory into the operand registers (one float4 per lane per 16 k, both
//     halves of a row block in flight around the MFMAs of the other), B from the slab by one ds_read_b128 per tile and 16 k, TC
//     INDEPENDENT accumulators issued round-robin (a chain re-issues every TC-th MFMA: never dependent back to back), operand
//     registers ping-ponged over fully unrolled steps so nothing is renamed or copied.  No barrier after the fill.
//   * all four slabs of an XCD walk the same row blocks in the same order: A is fetched from HBM once per XCD and served to the other
//     three slabs by that XCD's L2.
// Exact f32 (v_mfma_f32_16x16x4_f32 = fmaf chain); the order of the k-sum inside an output differs from the tiled kernels (lane
// group ks of step s supplies k = 16 s + 4 ks + m to MFMA m), results agree to round-off (tests/test_fullgraph_gpu.py::test_gemm_f32_all_layouts).
#include <mutex>
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

typedef float sl_f4 __attribute__((ext_vector_type(4)));

#ifndef GGAD_SLAB_WAVES
#define GGAD_SLAB_WAVES 8
#endif
constexpr int SL_WAVES = GGAD_SLAB_WAVES;      // waves per workgroup: two per SIMD
constexpr int SL_THREADS = SL_WAVES * 64;
constexpr int SL_MAXTC = 5;                    // column tiles of a slab (80 columns)
constexpr int SL_MAXSLABS = 16;                // N <= 1280
#ifndef GGAD_SLAB_RING
#define GGAD_SLAB_RING 0
#endif
// How a wave keeps its row block of A in registers.  false: two halves, each requested while the other is multiplied (half a row
// block ahead).  true: every float4 is requested again for the NEXT row block right after its step (one load per step, a whole row
// block ahead) -- measured slower (T-Finance 78 against 72 us, profiles/r05_gemm_slab_probe.log): 240 VGPRs and a partner wave that starves
constexpr bool SL_RING = GGAD_SLAB_RING != 0;

__device__ __forceinline__ void sl_dma16(const float *src, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(src) : "m0", "memory");
}

// smallest row stride >= kpad that is 8 or 56 mod 64 floats: the 16 lanes of a ds_read_b128 service group (8 rows at lane group ks,
// 8 at ks + 1) then touch 64 different banks
static int sl_kp(int kpad) {
  int kp = kpad;
  while ((kp & 63) != 8 && (kp & 63) != 56) kp += 4;
  return kp;
}

struct SlabArgs {
  const float *A, *B, *bias;
  const float *prelu_a;                       // with C2: C2 = PReLU(C) stored beside C (the first GCN layer on a cached aggregate, model.py:27-35)
  float *C2;
  int64_t ldc2;
  float *C;
  int M, N, K, KP, relu, vec_store, bias_off; // bias_off: floats from the start of LDS to the slab's bias values;           // vec_store: 16-byte stores of C are aligned and never straddle N
  int64_t lda, sbk, sbn, ldc;
  int n_tiles, n_slabs, n_slabs5;             // slabs 0 .. n_slabs5 - 1 hold 5 tiles, the rest 4
  int cu_end[SL_MAXSLABS];                    // workgroups (per XCD) of slabs 0 .. s: slab s owns [cu_end[s - 1], cu_end[s])
  uint32_t spr_magic;                         // ceil(2^32 / (KP / 4)): q / (KP / 4) = umulhi(q, spr_magic) for the slot numbers of a slab
#ifdef GGAD_SLAB_PROF
  unsigned long long *prof;                   // scripts/gemm_slab_probe.hip: 8 wall clocks (100 MHz) per wave
#endif
};
#ifdef GGAD_SLAB_PROF
#define SL_STAMP(slot) do { if (P.prof && lane == 0) P.prof[(size_t)(blockIdx.x * SL_WAVES + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define SL_STAMP(slot) do { } while (0)
#endif

// the row blocks rb0, rb0 + stride, ... < rb_end of one wave against the slab `bt` ([16 TC columns][KP], K-fast, zero beyond K)
// step s of a row block: this lane's float4 of A (k = 16 s + 4 ks ..+3; beyond K: zero, the address clamped into the row)
template <int KS>
__device__ __forceinline__ sl_f4 slab_load_a(const SlabArgs &P, const float *ar, int s, int ks) {
  const int koff = 16 * s + 4 * ks;
#ifdef GGAD_SLAB_NO_A        // (probe: the operand registers are never loaded)
  return sl_f4{(float)koff, 1.f, 2.f, 3.f};
#endif
  // (the last step's float4 may lie beyond K: the address is clamped into the row here and the value is replaced by zero WHERE IT IS USED --
  //  a select placed here is scheduled right behind the load and parks the wave for a memory round trip at the top of every row block:
  //  s_waitcnt vmcnt(4) in front of the first MFMA, 15 % of a T-Finance launch, profiles/r05_gemm_slab_probe.log)
  return *reinterpret_cast<const sl_f4 *>(ar + (s < KS - 1 ? koff : min(koff, P.K - 4)));
}

template <int KS, int TC>
__device__ __forceinline__ void slab_rows(const SlabArgs &P, const float *__restrict__ bt, int col0, int rb0, int rb_end, int stride, int lane,
                                          sl_f4 (&a)[KS]) {
  const int li = lane & 15, ks = lane >> 4;
  if (rb0 >= rb_end) return;
  constexpr int H0 = (KS + 1) / 2;
  const int M = P.M;
#ifdef GGAD_SLAB_SAME_A      // (probe: every wave reads row block 0 -- the same instructions, always L1 / L2 hits)
  auto row_ptr = [&](int rb) { return P.A + (int64_t)min((rb & 0) * 16 + li, M - 1) * P.lda; };
#else
  auto row_ptr = [&](int rb) { return P.A + (int64_t)min(rb * 16 + li, M - 1) * P.lda; };
#endif
  auto load_a = [&](const float *ar, int s) { return slab_load_a<KS>(P, ar, s, ks); };
  const float *bl = bt + li * P.KP + 4 * ks;
  // The slab fragment is passed as the FIRST MFMA operand and A's as the second: the tile comes out transposed, lane (li, ks) holding
  // C[row li][columns 16 t + 4 ks .. + 3] -- one 16-byte store per tile instead of four 4-byte ones (the stores of a row block cost 8 %
  // of a T-Finance launch as scalars, profiles/r05_gemm_slab_probe.log)
  const float *biasl = bt + P.bias_off + 4 * ks;          // the slab's 80 bias values (zeros without a bias) sit behind it in LDS
  const bool vec_store = P.vec_store;
  for (int rb = rb0; rb < rb_end; rb += stride) {
    // a[s] holds this lane's float4 of step s of the CURRENT row block; it is requested again -- for the NEXT row block of this wave -- as
    // soon as step s has issued: one load per step, a whole row block (19 steps, >= 5 us) ahead of its use
    const float *arn = row_ptr(min(rb + stride, rb_end - 1));          // (the last block: itself again, unused)
    if (!SL_RING) {                    // halves: steps H0 .. KS - 1 of THIS block now, steps 0 .. H0 - 1 of the next one at step H0
      const float *ar = row_ptr(rb);
#pragma unroll
      for (int s = H0; s < KS; ++s) a[s] = load_a(ar, s);
    }
    sl_f4 acc[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[t] = sl_f4{0.f, 0.f, 0.f, 0.f};
    sl_f4 b[2][TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) b[0][t] = *reinterpret_cast<const sl_f4 *>(bl + 16 * t * P.KP);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sl_f4 av = a[s];
      if (s == KS - 1 && 16 * (KS - 1) + 4 * ks >= P.K) av = sl_f4{0.f, 0.f, 0.f, 0.f};
      // m = 0 first, then the slab reads of step s + 1 and the reload of the operand of step s - 1 (their destinations were last read a
      // whole MFMA group ago: no hazard padding), then m = 1 .. 3: the reads land under 15 MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s & 1][t].x, av.x, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < KS) {
#pragma unroll
        for (int t = 0; t < TC; ++t) {
#ifdef GGAD_SLAB_NO_B      // (probe: no slab reads)
          b[(s + 1) & 1][t] = b[s & 1][t];
#else
          b[(s + 1) & 1][t] = *reinterpret_cast<const sl_f4 *>(bl + 16 * t * P.KP + 16 * (s + 1));
#endif
        }
      }
      if (SL_RING) {
        if (s >= 1) a[s - 1] = load_a(arn, s - 1);
      } else if (s == H0) {
#pragma unroll
        for (int q = 0; q < H0; ++q) a[q] = load_a(arn, q);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s & 1][t].y, av.y, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s & 1][t].z, av.z, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s & 1][t].w, av.w, acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (SL_RING) a[KS - 1] = load_a(arn, KS - 1);
    __builtin_amdgcn_sched_barrier(0);
#ifdef GGAD_SLAB_SAME_C      // (probe: every wave stores to the rows of its first row block)
    const int row = rb0 * 16 + li;
#else
    const int row = rb * 16 + li;
#endif
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      const int col = col0 + 16 * t + 4 * ks;
      sl_f4 o = acc[t] + *reinterpret_cast<const sl_f4 *>(biasl + 16 * t);
      if (P.relu) o = sl_f4{fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
      float *dst = P.C + (int64_t)row * P.ldc + col;
      if (P.C2) {                                                  // (vec_store holds for both: checked by the host)
        const float pa = *P.prelu_a;
        const sl_f4 o2 = sl_f4{o.x > 0.f ? o.x : pa * o.x, o.y > 0.f ? o.y : pa * o.y, o.z > 0.f ? o.z : pa * o.z, o.w > 0.f ? o.w : pa * o.w};
        if (row < M && col < P.N) *reinterpret_cast<sl_f4 *>(P.C2 + (int64_t)row * P.ldc2 + col) = o2;
      }
#ifdef GGAD_SLAB_NO_STORE    // (probe: results are dropped unless they hit an impossible value)
      if (o.x == 123456.789f) *reinterpret_cast<sl_f4 *>(dst) = o;
#else
      if (vec_store) {                                            // N % 4 == 0: a float4 is inside or outside as a whole
        if (row < M && col < P.N) *reinterpret_cast<sl_f4 *>(dst) = o;
      } else if (row < M) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (col + v < P.N) dst[v] = o[v];
      }
#endif
    }
    if (rb == rb0) SL_STAMP(3);
  }
}

template <int KS>
__global__ void __launch_bounds__(SL_THREADS) k_gemm_slab(SlabArgs P) {
  extern __shared__ __attribute__((aligned(16))) float bt[];      // [16 TC][KP]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroup b runs on XCD b % 8 (dispatcher rotation; a different placement costs L2 hits, never correctness).  An XCD owns an eighth
  // of the row blocks and has workgroups on EVERY slab, dealt in proportion to the slabs' column tiles
  const int GX = gridDim.x >> 3, x = blockIdx.x & 7, bi = blockIdx.x >> 3;
  int slab = 0, beg = 0, end = GX, tiles_before = 0;
  for (int s2 = 0; s2 < P.n_slabs; ++s2) {
    const int tc2 = s2 < P.n_slabs5 ? 5 : 4;
    const int e2 = P.cu_end[s2];                                   // (host: the split that minimises the busiest SIMD's MFMA count)
    if (bi >= e2) { slab = s2 + 1; beg = e2; tiles_before += tc2; }
    else { end = e2; break; }
  }
  const int tc = slab < P.n_slabs5 ? 5 : 4;
  const int col0 = 16 * tiles_before;
  const int n_cu = end - beg, g = bi - beg;
  const int RB = (P.M + 15) >> 4;
  const int rb_lo = (int)((int64_t)RB * x >> 3), rb_hi = (int)((int64_t)RB * (x + 1) >> 3);
  const int K = P.K, N = P.N, KP = P.KP, kpad = KS * 16;
  SL_STAMP(0);
#ifdef GGAD_SLAB_PROF
  if (P.prof && lane == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    P.prof[(size_t)(blockIdx.x * SL_WAVES + (threadIdx.x >> 6)) * 8 + 5] = hwid;
  }
#endif
  // ---- the row blocks of this XCD, dealt to the 4 n_cu SIMDs of the slab's workgroups; waves w and w + 4 share a SIMD and alternate.
  // The first half of the first row block is requested before the fill: both wait for the same first touch of memory
  const int J = 4 * n_cu, j = 4 * g + (wid & 3), rank = wid >> 2;
  const int rb0 = rb_lo + j + J * rank, stride = (SL_WAVES / 4) * J;
  sl_f4 a0[KS];
#define SLAB_LOAD_A0                                                                                                       \
  {                                                                                                                        \
    const float *ar = P.A + (int64_t)min(min(rb0, rb_hi - 1) * 16 + (lane & 15), P.M - 1) * P.lda;                        \
    _Pragma("unroll") for (int s = 0; s < (SL_RING ? KS : (KS + 1) / 2); ++s) a0[s] = slab_load_a<KS>(P, ar, s, lane >> 4);                 \
  }
  // ---- fill: bt[n][k] = op(B)[k][col0 + n], zero beyond N / K
  if (P.sbk == 1) {
    // K-fast in memory (x W^T: the rows of W): LDS-DMA, 16 bytes per lane, slot q of the slab = (column q / (KP / 4), float4 q % (KP / 4));
    // slots beyond K (and columns beyond N) fetch a clamped address and are overwritten with zeros by the wave that requested them
    const int spr = KP >> 2, slots = 16 * tc * spr, k4n = K >> 2;
    const int pieces = (slots + 63) >> 6;                          // (the allocation is rounded up to whole 1-KB pieces)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)bt;
    for (int p = wid; p < pieces; p += SL_WAVES) {
      const int q = min(64 * p + lane, slots - 1);
      const int n = (int)__umulhi((unsigned)q, P.spr_magic), c = q - n * spr;
      const float *src = P.B + (int64_t)min(col0 + n, N - 1) * P.sbn + 4 * min(c, k4n - 1);
      sl_dma16(src, lds0 + (uint32_t)(64 * p * 16));
    }
    SLAB_LOAD_A0;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SL_RING ? KS : (KS + 1) / 2) : "memory");      // the slab pieces (older) have landed; A stays in flight
    for (int p = wid; p < pieces; p += SL_WAVES) {
      const int q = 64 * p + lane;
      const int n = (int)__umulhi((unsigned)q, P.spr_magic), c = q - n * spr;
      if (q < slots && (c >= k4n || col0 + n >= N)) *reinterpret_cast<sl_f4 *>(bt + 4 * q) = sl_f4{0.f, 0.f, 0.f, 0.f};
    }
  } else {
    // N-fast in memory (dz W): transposed on the way in.  A wave takes 64 consecutive k (lane = k) and walks the float4 of their rows:
    // the scalar stores of a wave then fall on consecutive banks; the lines stay in the L1 between the passes
    SLAB_LOAD_A0;
    for (int kb = 64 * wid; kb < kpad; kb += 64 * SL_WAVES) {
      const int k = kb + lane;
      const float *src = P.B + (int64_t)min(k, K - 1) * P.sbk;
      sl_f4 v[4 * SL_MAXTC];
#pragma unroll
      for (int q = 0; q < 4 * SL_MAXTC; ++q) v[q] = *reinterpret_cast<const sl_f4 *>(src + min(col0 + 4 * q, N - 4));      // (N % 4 == 0)
#pragma unroll
      for (int q = 0; q < 4 * SL_MAXTC; ++q)
        if (k >= K || col0 + 4 * q >= N) v[q] = sl_f4{0.f, 0.f, 0.f, 0.f};
      if (k < kpad) {
#pragma unroll
        for (int q = 0; q < 4 * SL_MAXTC; ++q) {
          if (q < 4 * tc) {
            float *d = bt + (4 * q) * KP + k;
            d[0] = v[q].x; d[KP] = v[q].y; d[2 * KP] = v[q].z; d[3 * KP] = v[q].w;
          }
        }
      }
    }
  }
  if (threadIdx.x < 16 * SL_MAXTC)
    bt[P.bias_off + threadIdx.x] = (P.bias && col0 + (int)threadIdx.x < N) ? P.bias[col0 + threadIdx.x] : 0.0f;
#undef SLAB_LOAD_A0
  SL_STAMP(1);
  __syncthreads();
  SL_STAMP(2);
#ifdef GGAD_SLAB_STAGGER
  // The two waves of a SIMD run the same code on units of the same length: left alone they stay in phase, both in their load bursts and
  // both in their epilogues at once, and the MFMA pipe has nobody to take the slots.  The second wave starts a fraction of a unit late.
  if (rank == 1) {
    for (int i = 0; i < GGAD_SLAB_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);       // 16 x 64 clocks each
  }
#endif
  if (tc == 5) slab_rows<KS, 5>(P, bt, col0, rb0, rb_hi, stride, lane, a0);
  else slab_rows<KS, 4>(P, bt, col0, rb0, rb_hi, stride, lane, a0);
  SL_STAMP(4);
}

}  // namespace

#ifdef GGAD_SLAB_PROF
static unsigned long long *g_slab_prof = nullptr;
#endif
constexpr int SL_MAXDEV = 64;
static std::mutex g_slab_mu;
static int g_slab_cus[SL_MAXDEV] = {};

// C++ linkage, called by ggad_gemm_f32 (gemm.hip).  Returns 1 when the product was launched here, 0 when the shape is not this kernel's
// (the caller goes on to the tiled kernels), < 0 on a launch error.
int ggad_int_gemm_slab(const float *A, const float *B, float *C, int M, int N, int K, int64_t lda, int64_t sbk, int64_t sbn, int64_t ldc,
                       const float *bias, int relu, hipStream_t st, const float *prelu_a, float *C2, int64_t ldc2) {
  static const int enabled = [] { const char *e = getenv("GGAD_GEMM_SLAB"); return e ? atoi(e) : 1; }();
  static const int min_m = [] { const char *e = getenv("GGAD_GEMM_SLAB_MIN_M"); return e ? atoi(e) : 4096; }();
  if (!enabled || M < min_m || K % 4 != 0) return 0;
  const int KSn = (K + 15) / 16;
  if (!(KSn == 2 || KSn == 4 || (KSn >= 16 && KSn <= 20))) return 0;       // K = 17 .. 32 / 49 .. 64 (narrow first layers) or 241 .. 320
  const int n_tiles = (N + 15) / 16, n_slabs = (n_tiles + SL_MAXTC - 1) / SL_MAXTC, n4 = SL_MAXTC * n_slabs - n_tiles;
  if (n_tiles < 8 || n4 > n_slabs) return 0;                           // (5 a + 4 b = n_tiles has no solution with a + b = n_slabs)
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0 || lda % 4 != 0) return 0;
  if (!((sbk == 1 && sbn % 4 == 0) || (sbn == 1 && sbk % 4 == 0 && N % 4 == 0))) return 0;
  // per DEVICE, under a mutex: the CU count that sizes the grid and the slab split, and -- below -- the opt-in for
  // more than 64 KB of dynamic LDS, which hipFuncSetAttribute records for the current device only
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SL_MAXDEV) return 0;
  int n_cus;
  {
    std::lock_guard<std::mutex> lock(g_slab_mu);
    if (g_slab_cus[dev] == 0) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
      g_slab_cus[dev] = v;
    }
    n_cus = g_slab_cus[dev];
  }
  const int G = n_cus / 8 * 8, GX = G / 8;
  if (GX < n_slabs || n_slabs > SL_MAXSLABS) return 0;
  SlabArgs P;
  P.A = A; P.B = B; P.bias = bias; P.C = C; P.M = M; P.N = N; P.K = K; P.KP = sl_kp(KSn * 16); P.relu = relu;
  P.vec_store = (N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0) ? 1 : 0;
  P.prelu_a = prelu_a; P.C2 = C2; P.ldc2 = ldc2;
  if (C2 && !(prelu_a && P.vec_store && ldc2 % 4 == 0 && ldc2 >= N && ((uintptr_t)C2 & 15) == 0)) return 0;
  P.spr_magic = (uint32_t)(((1ull << 32) + (uint64_t)(P.KP / 4) - 1) / (uint64_t)(P.KP / 4));
  P.lda = lda; P.sbk = sbk; P.sbn = sbn; P.ldc = ldc; P.n_tiles = n_tiles; P.n_slabs = n_slabs; P.n_slabs5 = n_slabs - n4;
  // The CUs of an XCD are dealt to the slabs so that the busiest SIMD issues as few MFMAs as possible: slab s with n CUs gives each of its
  // 4 n SIMDs ceil(row blocks / 4 n) units of TC_s tiles.  (In proportion to the tiles -- 8 / 8 / 9 / 7 for 5 + 5 + 5 + 4 -- the 7-CU slab
  // of Reddit's 86 row blocks per XCD runs 4 units x 4 tiles against 3 x 5 everywhere else.)  Smallest feasible bound first.
  {
    const int rbx = ((M + 15) / 16 + 7) / 8;
    int need[SL_MAXSLABS], best = 0;
    for (int L = 1;; ++L) {                                          // L: tile-units of the busiest SIMD
      int tot = 0;
      for (int s2 = 0; s2 < n_slabs; ++s2) {
        const int tc2 = s2 < P.n_slabs5 ? 5 : 4, units = L / tc2;
        need[s2] = units == 0 ? GX + 1 : (rbx + 4 * units - 1) / (4 * units);
        tot += need[s2];
      }
      if (tot <= GX) { best = tot; break; }
    }
    // spare CUs go, one at a time, to the slab whose SIMDs are busiest
    for (int spare = GX - best; spare > 0; --spare) {
      int arg = 0, worst = -1;
      for (int s2 = 0; s2 < n_slabs; ++s2) {
        const int tc2 = s2 < P.n_slabs5 ? 5 : 4, load = (rbx + 4 * need[s2] - 1) / (4 * need[s2]) * tc2;
        if (load > worst) { worst = load; arg = s2; }
      }
      need[arg]++;
    }
    int acc = 0;
    for (int s2 = 0; s2 < SL_MAXSLABS; ++s2) { if (s2 < n_slabs) acc += need[s2]; P.cu_end[s2] = acc; }
  }
#ifdef GGAD_SLAB_PROF
  P.prof = g_slab_prof;
#endif
  P.bias_off = (int)(((size_t)16 * SL_MAXTC * P.KP * sizeof(float) + 1023) / 1024 * 1024 / sizeof(float));      // behind the slab's whole 1-KB pieces
  const size_t lds = ((size_t)P.bias_off + 16 * SL_MAXTC) * sizeof(float);
#define GGAD_SLAB(KSV)                                                                                                                   \
  case KSV: {                                                                                                                            \
    static int state[SL_MAXDEV] = {};                /* 0 unknown, 1 ready, -1 this device refuses the opt-in: shape not taken */         \
    {                                                                                                                                    \
      std::lock_guard<std::mutex> lock(g_slab_mu);                                                                                       \
      if (state[dev] == 0) {                                                                                                             \
        const bool ok = hipFuncSetAttribute((const void *)k_gemm_slab<KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess; \
        (void)hipGetLastError();                                                                                                         \
        state[dev] = ok ? 1 : -1;                                                                                                        \
      }                                                                                                                                  \
      if (state[dev] != 1) return 0;                                                                                                     \
    }                                                                                                                                    \
    k_gemm_slab<KSV><<<dim3(G), dim3(SL_THREADS), lds, st>>>(P);                                                                          \
  } break
  switch (KSn) {
    GGAD_SLAB(2); GGAD_SLAB(4); GGAD_SLAB(16); GGAD_SLAB(17); GGAD_SLAB(18); GGAD_SLAB(19); GGAD_SLAB(20);
    default: return 0;
  }
#undef GGAD_SLAB
  if (hipGetLastError() != hipSuccess) return -1;
  return 1;
}
