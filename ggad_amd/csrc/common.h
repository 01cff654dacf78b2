// Shared helpers for the gfx950 kernels of libggad_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ggad_hip.h"

#define GGAD_WAVE 64
#define GGAD_MAX_D 64    // embedding width handled by one wave (lane = channel)
#define GGAD_MAX_F 1024  // feature width limit of the gather kernels

void ggad_set_error(hipError_t e, const char *where);

#define GGAD_CHECK_LAUNCH(where)                 \
  do {                                           \
    hipError_t _e = hipGetLastError();           \
    if (_e != hipSuccess) {                      \
      ggad_set_error(_e, where);                 \
      return GGAD_E_LAUNCH;                      \
    }                                            \
  } while (0)

#define GGAD_REQUIRE(cond) \
  do {                     \
    if (!(cond)) return GGAD_E_INVALID; \
  } while (0)

static inline hipStream_t as_stream(ggad_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (GGAD_WAVE - 1); }

// Sum over the 64 lanes of a wave; every lane gets the result.  Fixed butterfly order,
// so the value is deterministic for given inputs.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, GGAD_WAVE);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, GGAD_WAVE);
  return v;
}

// first index in sorted a[lo,hi) with a[idx] >= key
__device__ __forceinline__ int lower_bound_i32(const int32_t *__restrict__ a, int lo, int hi, int key) {
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---- plan launches that leave one XCD alone.  The dispatcher deals the workgroups of a launch to the eight XCDs in strict rotation
// (block b runs on XCD (first + b) % 8, `first` a constant of the stream: scripts/xcd_block_map.hip).  While the XCD-resident chunk
// kernel (step_xcd.hip) owns an XCD, the plan kernels of the next chunk run on the other seven: launched with ggad_skip_grid(n)
// workgroups, every eighth one -- blockIdx.x % 8 == skip, the ones the dispatcher would put on that XCD -- returns at once and the
// others number themselves 0 .. n' - 1 (n' >= n: every kernel bounds-checks what it derives from the number).  Which blocks are
// skipped is decided by INDEX, so the results never depend on the placement; only the speed does.  skip < 0: plain launch.
__device__ __forceinline__ bool ggad_vblock(int skip, unsigned &vb, unsigned &vg) {
  const unsigned b = blockIdx.x, n = gridDim.x;
  if (skip < 0) { vb = b; vg = n; return true; }
  const unsigned sk = (unsigned)skip;
  if ((b & 7u) == sk) return false;
  vb = b - ((b >> 3) + ((b & 7u) > sk ? 1u : 0u));
  vg = n - ((n >> 3) + ((n & 7u) > sk ? 1u : 0u));
  return true;
}
static inline unsigned ggad_skip_grid(unsigned want, int skip) { return skip < 0 ? want : (unsigned)(((uint64_t)(want + 1) * 8 + 6) / 7); }

// ---- internal (C++ linkage) pieces of ggad_mb_plan_build, shared between plan_build.cpp, plan.hip and hop2_ldsw.hip
// counters[] of a plan: every counter on a 64-byte line of its own.  A device-scope atomic on ONE address (or line) completes every
// ~7 ns whoever issues it (measured: 43 K wave-level atomics = 313 us of k_build_groups), so the plan kernels reserve storage
// once per WORKGROUP (wg_reserve below), not once per wave, and the work-item cursor of the 2-hop gather is bumped by big grabs.
constexpr int GGAD_CTR_STRIDE = 16;
constexpr int GGAD_CTR_GROUPS = 0 * GGAD_CTR_STRIDE, GGAD_CTR_ITEMS = 1 * GGAD_CTR_STRIDE, GGAD_CTR_PART = 2 * GGAD_CTR_STRIDE,
              GGAD_CTR_CURSOR = 3 * GGAD_CTR_STRIDE, GGAD_CTR_PC = 4 * GGAD_CTR_STRIDE,
              GGAD_CTR_BIG = 5 * GGAD_CTR_STRIDE,          // groups of range-partitioned owners (hop2_ldsw.hip)
              GGAD_CTR_RANGE0 = 6 * GGAD_CTR_STRIDE,       // .. 13: the work cursor of each of the eight id ranges
              GGAD_CTR_PROF = 14 * GGAD_CTR_STRIDE;        // 14, 15: phase clocks of a -DGGAD_G2_PROF build (16 x 64 bits)
constexpr int GGAD_RANGES = 8;            // id ranges of the 2-hop gather: one per XCD, so that each L2 keeps an eighth of the hot rows
constexpr int GGAD_RANGE_DEG = 256;       // owners with more neighbours than this are gathered range by range
constexpr int GGAD_PLAN_COUNTERS = 16 * GGAD_CTR_STRIDE;      // ints of ggad_mb_plan::counters

struct ggad_plan_view {      // device views into the staging block of ONE build + its exact sizes (known on the host)
  const int32_t *batch_ptr, *batch_ent_ptr, *nodes, *row_slot, *ent_ptr, *row_ck_ptr, *ck_rc, *ck_e0;
  int32_t n_batches, n_rows, n_ents, n_chunks;
  int32_t seg_stride;        // entries per tile row of seg_t (n_ents rounded up to 64)
  int64_t pair_bound;        // host bound on the (owner, neighbour) pairs of THIS build (the capacities may be far larger)
};
int ggad_int_hop1(const ggad_mb_plan *P, const ggad_plan_view &V, int ldsw, int reset_now, hipStream_t st);
int ggad_int_global_hop2(const ggad_mb_plan *P, const ggad_plan_view &V, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
int ggad_int_ldsw_hop2(const ggad_mb_plan *P, const ggad_plan_view &V, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
int ggad_int_range_deg();      // owners above this degree are gathered by id range (INT32_MAX: the partition is off)

// ---- gemm_slab.hip: the slab-resident tall product; 1 = launched, 0 = not this kernel's shape, < 0 = launch error
int ggad_int_gemm_slab(const float *A, const float *B, float *C, int M, int N, int K, int64_t lda, int64_t sbk, int64_t sbn, int64_t ldc,
                       const float *bias, int relu, hipStream_t st, const float *prelu_a = nullptr, float *C2 = nullptr, int64_t ldc2 = 0);

// ---- mlp.hip: D (m x n) = P^T Q over R rows as row-range partials + ordered reduction (the weight gradients); 1 = launched, 0 = not taken
int64_t ggad_int_wgrad_tn_ws(int R, int m, int n);
int ggad_int_wgrad_tn(const float *P, int64_t ldp, const float *Q, int64_t ldq, int R, int m, int n, float *D, float *ws, hipStream_t st);

// ---- one-shot gradient exchange (exchange.cpp owns the handle, step.hip the kernel)
constexpr int GGAD_XCHG_MAX_WORLD = 16;
struct ggad_xchg_view {                          // what the kernel needs, passed by value
  float *peer[GGAD_XCHG_MAX_WORLD];             // peer[q] = rank q's buffer as mapped into THIS process (peer[rank] = own)
  int32_t rank, world;
  int64_t n;                                     // floats per slot
  int32_t *err;                                  // device word: set when a wait timed out
  unsigned long long timeout_ticks;              // bound of a wait in 100 MHz wall-clock ticks (GGAD_XCHG_TIMEOUT_S, default 20 s)
};
struct ggad_xchg {
  ggad_xchg_view view;
  void *local;                                   // own buffer (fine-grained device memory)
  size_t bytes;
  uint32_t step;                                 // exchanges done so far (all ranks call in lockstep)
  bool opened[GGAD_XCHG_MAX_WORLD];
};
static inline size_t ggad_xchg_granules(int world, int64_t n) { return (size_t)2 * world * (size_t)n; }   // 8 bytes each: {value, step}
