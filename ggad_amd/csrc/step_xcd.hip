// One launch per CHUNK, resident on ONE XCD: the optimiser steps of all batches of a chunk (the per-batch loop of
// src/model_handler.py:330-364 -- GCNEncoder.forward + GCN.loss (src/graphsage.py:395-454, 171-258), backward, Adam
// (model_handler.py:363-364)) inside one persistent kernel whose workgroups all sit behind the same L2.
//
// Why one XCD.  The launch chain of step.hip is 5 dependent launches per 200-row step: 33-35 us for < 30 MFLOP and < 2 MB,
// because every dependent level of loads misses the L2 (written a launch earlier on other XCDs) and every boundary costs
// ~1.5-1.9 us.  A persistent kernel spread over the chip (round 1) paid the same at every phase: a chip-wide barrier needs the
// agent-scope release / acquire (L2 write-back, write-through stores) because the eight L2s are not coherent with each other.
// Inside ONE XCD none of that is needed: plain stores stay in the shared L2, sc1 loads bypass the reader's L1 and are served by
// that L2 (119 ns), and a barrier is a tagged-slot all-gather -- every workgroup stores the round number into its own 4-byte slot
// and one wave polls the 32 slots with one L2-served load: 0.43 us (scripts/xcd_barrier_bench.hip: 1.16 us for a round of
// 2 KB hand-offs + barrier + 3 remote reads; zero stale words under skewed arrivals and HBM noise; a device-scope counter
// barrier over the chip: 3-7 us).  The other seven XCDs stay free for the plan kernels of the next chunk.
//
// Placement.  HIP cannot place workgroups, but the dispatcher deals the workgroups of a launch to the eight XCDs in strict
// rotation (exactly grid / 8 each, whatever the stream's CU mask: scripts/xcd_block_map.hip), so the launch has 8 nv workgroups:
// every workgroup reads HW_REG_XCC_ID; the ones that are not on the host-chosen XCD leave at once; the nv on it take ranks in
// registration order and wait (bounded) until all nv are there.  If the rotation ever were not exact the launch would not hang
// or compute garbage: the wait times out, the error word is set, every workgroup leaves, and the host raises at its next
// check (ggad_mb_xcd_status).  nv <= 32 is a launch parameter (the trainer leaves a few CUs of the XCD to the plan stream).
// A barrier wait is bounded the same way.
//
// Phases of one step (4 barriers), lane = embedding channel d, F = 17:
//   A   pieces (<= 16 consecutive entries of ONE row, tables of the plan): partial sums of relu(W x2[own(e)]) -> chunk_part;
//       rows: h1 = relu(W x1), 16 rows per wave.  Both on the matrix cores (one piece = the M of a 16 x 16 x 4 tile); the
//       operands of a wave's pieces stay in registers for phase C.
//   R   positions q of combined_all: nbar = (1/r) sum of the row's piece partials (piece order), gen = relu(fc nbar) of the
//       source row of a generated column, score, BCE term, cosine affinity, norms, recon norm (graphsage.py:174,197,234,246)
//   C   loss scalars (same reduction tree as k_loss_rows), per piece: the row's backward coefficients recomputed by the wave
//       that owns the piece (no extra barrier), relu mask recomputed (same instructions as in A: same bits), dW += C^T X on the
//       matrix cores, dW partial per virtual workgroup (fixed LDS combine)
//   E   gradient reduction over the 32 partials / the label-1 rows / the rows (fixed order), [data-parallel exchange,] Adam,
//       transposed weight copies; next step's weights are re-read through the L2 after the barrier
// Latency.  What bounds a step on 32 compute units is neither flops nor bytes but the number of DEPENDENT memory round trips
// (~0.7 us each under load, L2 hits included): the first version walked ~20 per step (piece -> first entry -> owner -> feature
// row; row -> label / position -> that position's label; ...) and ran at the launch chain's 35 us.  Now (1) k_xcd_prep, one
// whole-chip launch per chunk, flattens the plan's tables into one 32-byte record per piece and per position and completes x2
// per entry, so a piece's rows are consecutive; (2) the records and matrix-core operands of step b + 1 are requested during
// phase C of step b, BEHIND that phase's own loads (loads return in order), and wait in registers; (3) every phase issues all
// its loads first: one round trip per phase.
//
// Same piece order per row and the same loss reduction tree as the launch chain; the projections run as f32 MFMA k-steps of 4
// instead of 17 sequential fma, and the dW partial sums are grouped by piece instead of by flat entry stripes, so losses and
// gradients agree with the chain to fp32 round-off (tests: 2e-6).  Everything is deterministic (no float atomics), for any
// number of surviving workgroups.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include <cstddef>
#include "step_common.h"

namespace {

constexpr int GGAD_MAX_DEVICES = 16;
constexpr int XW = 8;                 // waves per workgroup
constexpr int XT = XW * GGAD_WAVE;    // threads per workgroup
constexpr int XMAXWG = 32;            // workgroups of a launch that stay (CUs of an XCD)
constexpr int XMINWG = 18;            // 17 columns of W + w need one owner each in phase E
constexpr int XFT = 17;               // feature width (DGraph-Fin)

constexpr int XPC = 2;                // pieces per virtual wave whose records / operands are fetched a step ahead and kept in registers
constexpr int XHUB = 32;              // pieces from which a row's partial sums are added by the whole workgroup (phase R)
constexpr int XRA = 512;              // label-1 rows of a batch listed in LDS for phase E
constexpr int XBT = 256;              // batches whose offsets are staged in LDS at launch (later ones are read per step)
constexpr int FCS = GGAD_MAX_D + 1;   // row stride of fc^T in LDS (conflict-free for both access patterns)

struct XcdCtrl {
  unsigned reg[8][16];                // [xcc][0]: workgroups registered on that XCD (one 64-byte line each)
  unsigned slot[64];                  // barrier: last round number published by rank r
  unsigned err, survivors, xcc, placed;  // placed: launch id, written by rank 0 once xcc is valid (read by the helper kernel)
  unsigned long long prof[16];        // wall clocks (100 MHz) of rank 0 per phase (0..7) and sub-phase (8..15), accumulated over the launch
  unsigned done, helpers, pad[2];     // done: the chunk kernel has left; helpers: registration counter of the helper kernel
};
static_assert(sizeof(XcdCtrl) <= 1020 && sizeof(XcdCtrl) % 16 == 0, "control block is 256 floats of the workspace");
// The control block is cleared at every launch; the LAST word of its 256-float region is not: it keeps the first error code of ANY
// launch since the host last cleared it (ggad_mb_xcd_clear_error), so a time-out of chunk k is still visible after chunk k + 1 has
// reset `err` (the host reads the status once per run, not once per chunk).
constexpr int XCD_STICKY_WORD = 255;
__device__ __forceinline__ void xcd_fail(XcdCtrl *C, unsigned code) {
  C->err = code;
  atomicCAS(reinterpret_cast<unsigned *>(C) + XCD_STICKY_WORD, 0u, code);
}

struct XcdArgs {
  ggad_mb_step s;
  const int32_t *batch_ptr;           // device, n_batches + 1 row offsets
  int n_batches, log_base, ld_o;      // ld_o: positions per row of pos_o (multiple of 4)
  float *loss_log;
  XcdCtrl *ctrl;
  float *pos_scal, *pos_o, *gw_row, *dw_part;
  const int32_t *ck_rec, *pos_rec;    // records of k_xcd_prep (8 ints per piece / per position)
  const int32_t *batch_n0;            // label-0 positions per batch (k_xcd_prep)
  const float *adam_sc;               // Adam's step_size / sqrt(bias_correction2) per batch of the chunk (k_xcd_prep: double pow)
  ggad_xchg_view X;
  uint32_t xstep0;
  float grad_scale;
  unsigned long long timeout_ticks;
  unsigned launch_id;
  int dbg;
  int nv, want_xcd;                    // workgroups that stay, the XCD they stay on
};

__device__ __forceinline__ float cld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct f2 { float x, y; };
__device__ __forceinline__ f2 cld2(const float *p) {          // 8-byte aligned pair, one L1-bypassing load
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f2{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))};
}
__device__ __forceinline__ unsigned cldu(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ int ri(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// All workgroups of the launch sit on one XCD: plain stores are in the shared L2 once the wave's vmcnt has drained; the round
// number in the own slot publishes them; one wave polls all slots with one L1-bypassing load.
__device__ __forceinline__ bool xcd_barrier(XcdCtrl *C, int rank, int G, unsigned round, unsigned long long timeout) {
  __shared__ int s_dead;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < GGAD_WAVE) {
    if (threadIdx.x == 0) { C->slot[rank] = round; s_dead = 0; }
    const int l = (int)threadIdx.x < G ? (int)threadIdx.x : 0;
    const unsigned long long t0 = wall_clock64();
    int spins = 0;
    while (true) {
      const unsigned v = cldu(&C->slot[l]);
      if (__all((int)(v - round) >= 0)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023) == 0 && (wall_clock64() - t0 > timeout || cldu(&C->err) != 0)) {
        if (threadIdx.x == 0) { xcd_fail(C, 1); C->done = 1; s_dead = 1; }
        break;
      }
    }
  }
  __syncthreads();
  return s_dead == 0;
}

// ---- the per-piece products on the matrix cores.  v_mfma_f32_16x16x4_f32: D[16 x 16] += A[16 x 4] B[4 x 16]; lane l = (a, g) =
// (l % 16, l / 16) holds A[a][g], B[g][a] and D[4 g + v][a] (v = 0..3).  A piece is <= 16 entries of one row = the M of one tile.
//   forward / relu mask:  H[e][ch] = sum_f X[e][f] W[ch][f]     A = X (entry a, features 4 j + g), B = W^T tile t (channels 16 t + a),
//                         5 k-steps (17 features padded to 20) x 4 channel tiles = 20 instructions for 16 x 64 outputs;
//   weight gradient:      dW[ch][f] += sum_e C[e][ch] X[e][f]   A = the masked coefficients IN THE LAYOUT THE FORWARD PRODUCT
//                         LEAVES THEM (lane (a, g), register v = entry 4 g + v, channel 16 t + a: k-step v covers the entries
//                         {v, 4 + v, 8 + v, 12 + v}), B = X (entry 4 g + v, feature a < 16): 16 instructions; feature 16 is a
//                         separate column accumulated on the VALU (4 fma per tile).
// (The readlane-broadcast VALU form of the same arithmetic, lane = channel, costs 34 / 53 VALU instructions per entry.)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int XKS = (XFT + 3) / 4;       // k-steps of the forward product (5)
constexpr int XNT = GGAD_MAX_D / 16;     // channel tiles (4)
constexpr int XLS = 68;                  // LDS row stride of a dW partial: position of (f, ch) = 68 f + 16 (ch % 4) + ch / 4 is
                                         // conflict-free for the tile layout (bank 4 a + g) and for lane = channel
__device__ __forceinline__ int dw_pos(int f, int ch) { return XLS * f + 16 * (ch & 3) + (ch >> 2); }

struct PieceX {
  float a[XKS];      // forward operand: X[entry a][4 j + g]
  float b[4];        // gradient operand: X[entry 4 g + v][feature a]
};
// n consecutive rows of x (stride XFT floats) from row `base` on; rows beyond n read as zero.  Every load is unconditional
// (clamped address, select afterwards): 9 loads in flight per piece.
__device__ __forceinline__ void load_piece_x(const float *__restrict__ x, int base, int n, int lane, PieceX &P) {
  const int a = lane & 15, g = lane >> 4;
  const unsigned ra = base + min(a, n - 1);
#pragma unroll
  for (int j = 0; j < XKS; ++j) {
    const int f = 4 * j + g;
    const float xv = x[ra * XFT + (f < XFT ? f : 0)];
    P.a[j] = (a < n && f < XFT) ? xv : 0.0f;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int e = 4 * g + v;
    const float xv = x[(unsigned)(base + min(e, n - 1)) * XFT + a];
    P.b[v] = e < n ? xv : 0.0f;
  }
}
__device__ __forceinline__ void piece_fwd(const PieceX &P, const float (&WB)[XKS][XNT], f4 (&h)[XNT]) {
#pragma unroll
  for (int t = 0; t < XNT; ++t) h[t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < XKS; ++j)
#pragma unroll
    for (int t = 0; t < XNT; ++t) h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(P.a[j], WB[j][t], h[t], 0, 0, 0);
}

// ---- sum_k m[k * stride] * v[lane k], k = 0..63 in order (one FMA chain: the order of the launch chain's kernels).  Sixteen LDS
// reads in flight per group: left to itself the compiler waits for every ds_read before the next (64 x ~100 clocks on the
// critical path of every generated position and label-1 piece).
__device__ __forceinline__ float lds_matvec(const float *m, int stride, float v) {
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < GGAD_MAX_D; c += 16) {
    float fv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) fv[k] = m[(c + k) * stride];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) a = fmaf(fv[k], rl(v, c + k), a);
    __builtin_amdgcn_sched_barrier(0);
  }
  return a;
}

// ---- records (k_xcd_prep, once per chunk): everything a step needs to know about a piece / a position in ONE 32-byte line,
// so that no phase walks a chain of dependent table look-ups (row -> label / position -> position's label, source row -> its pieces)
//   piece c:        [0] (row << 6) | entries   [1] first entry   [2] label | first piece of its row << 1 | label of position q1 << 2
//                   [3] entries of the row     [4] q1 = position whose column this row is     [5] pos_meta of position (row - row0)
//   position q:     [0] pos_meta   [1] first piece of row q   [2] its pieces   [3] entries of row q
//                   [4] first piece of the source row (generated columns)   [5] its pieces   [6] its entries
constexpr int REC = 8;
struct RowIn { float H1, NB, Gl, pp, nbq, c2, xr; };      // pp: lanes 0..7 = scalars of position q1, lanes 8..15 = of position i

// Adam on one parameter whose state is already in registers (torch.optim.Adam's single-tensor update op by op, step_common.h)
__device__ __forceinline__ void adam_apply(float *__restrict__ params, float *__restrict__ m, float *__restrict__ v, ParamLayout L,
                                           int i, float p, float mi, float vi, float g, float wd, float step_size, float bc2s) {
  g = fmaf(wd, p, g);                                       // grad.add(param, alpha=weight_decay)
  mi = fmaf(g - mi, 0.1f, mi);                              // exp_avg.lerp_(grad, 1 - beta1)
  vi = fmaf(0.001f * g, g, vi * 0.999f);                    // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(vi) / bc2s + 1e-8f;
  p = p - step_size * (mi / denom);                         // addcdiv_(exp_avg, denom, -step_size)
  params[i] = p; m[i] = mi; v[i] = vi;
  const int D = L.D, F = L.F;
  if (i >= L.o_W() && i < L.o_fc()) {
    const int u = i - L.o_W(); const int d = u / F, f = u - d * F;
    params[L.o_Wt() + f * D + d] = p;
  } else if (i >= L.o_fc()) {
    const int u = i - L.o_fc(); const int d = u / D, d2 = u - d * D;
    params[L.o_fcT() + d2 * D + d] = p;
  }
}

template <int MODE, int DT>      // MODE 1: Adam; 2: one-shot data-parallel exchange + Adam.  DT: embedding width at compile time (0 = run time)
__global__ void __launch_bounds__(XT) k_train_chunk_xcd(XcdArgs A) {
  __shared__ int s_rank;
  __shared__ float wt_lds[XFT * GGAD_MAX_D];            // W^T of this step
  __shared__ float w_lds[GGAD_MAX_D];
  __shared__ float fct[GGAD_MAX_D * FCS];               // fc^T of this step, rows padded
  __shared__ float accw[XW][XFT * XLS];                 // dW combine (phase C), sub-reducer partials (phase E)
  __shared__ int ra_lds[XRA];                           // label-1 rows of the batch (sources of the generated columns), phase E
  __shared__ float t_lds[8];
  __shared__ float sc[2];
  __shared__ int hub_first[2 * XW], hub_n[2 * XW], hub_any;   // phase R: rows summed by the whole workgroup
  __shared__ float hub_part[XW][GGAD_WAVE];
  __shared__ int bt_row0[XBT + 1], bt_ck0[XBT + 1], bt_n0[XBT + 1];   // row / piece offsets, label-0 positions of the first XBT batches
  XcdCtrl *C = A.ctrl;
  // ---------------------------------------------------------------- placement
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    int r = -1;
    if ((int)xcc == A.want_xcd) {
      r = (int)__hip_atomic_fetch_add(&C->reg[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      while (r < A.nv && cldu(&C->reg[xcc][0]) < (unsigned)A.nv) {
        __builtin_amdgcn_s_sleep(2);
        if (cldu(&C->err) != 0 || wall_clock64() - t0 > A.timeout_ticks) { xcd_fail(C, 2); C->done = 1; r = -1; break; }
      }
      if (r >= A.nv) { xcd_fail(C, 3); r = -1; }                // more than grid / 8 workgroups on one XCD: not the dealing we rely on
      if (r == 0) {
        C->survivors = (unsigned)A.nv; C->xcc = xcc;
        __hip_atomic_store(&C->placed, A.launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    s_rank = r;
  }
  __syncthreads();
  const int rank = s_rank;
  if (rank < 0) return;
  const int nv = A.nv, NVW = nv * XW, G = nv;

  const ggad_mb_step &S = A.s;
  const int D = DT ? DT : S.D;
  const ParamLayout L{D, XFT};
  const int lane = lane_id(), wid = threadIdx.x / GGAD_WAVE;
  const int la = lane & 15, lg = lane >> 4;             // (a, g) of the matrix-core layouts
  const int l8 = lane & 7;
  const bool on = lane < D;
  const int d = on ? lane : D - 1;
  const int fl = lane < XFT ? lane : XFT - 1;
  // the records and operands of step b + 1 are fetched into registers during step b
  constexpr bool piped = true;
  float *params = S.params;
  const int step0 = *S.step_counter;
  for (int i = threadIdx.x; i < GGAD_MAX_D * FCS; i += XT) fct[i] = 0.0f;      // rows / columns beyond D stay zero
  if (threadIdx.x == 0) hub_any = 0;
  for (int i = threadIdx.x; i <= min(A.n_batches, XBT); i += XT) {
    const int r0 = A.batch_ptr[i];
    bt_row0[i] = r0;
    bt_ck0[i] = S.row_ck_ptr[r0];
    bt_n0[i] = i < A.n_batches ? A.batch_n0[i] : 0;
  }
  __syncthreads();
  auto batch_info = [&](int bb, int &row0, int &B, int &ck0, int &nck) {
    if (bb < XBT) {
      row0 = bt_row0[bb]; B = bt_row0[bb + 1] - row0; ck0 = bt_ck0[bb]; nck = bt_ck0[bb + 1] - ck0;
    } else {
      row0 = A.batch_ptr[bb]; B = A.batch_ptr[bb + 1] - row0; ck0 = S.row_ck_ptr[row0]; nck = S.row_ck_ptr[row0 + B] - ck0;
    }
  };
  // virtual wave of this wave in its first virtual workgroup: pieces / positions v0, v0 + NVW, ...  Consecutive pieces (and
  // positions) go to consecutive WORKGROUPS, so a batch of 358 pieces puts 11-12 on every compute unit
  const int v0 = rank + nv * wid;
  auto piece_rec = [&](int c) { return A.ck_rec[(unsigned)c * REC + l8]; };
  auto pos_rec = [&](int row) { return A.pos_rec[(unsigned)row * REC + l8]; };
  auto issue_recs = [&](int bb, int (&rv)[XPC], int &pv) {          // records of this wave's pieces / position in step bb
    int r0, Bn, c0, nc;
    batch_info(bb, r0, Bn, c0, nc);
#pragma unroll
    for (int j = 0; j < XPC; ++j) rv[j] = piece_rec(c0 + min(v0 + j * NVW, max(nc - 1, 0)));
    pv = pos_rec(r0 + min(v0, max(Bn - 1, 0)));
  };
  auto issue_ops = [&](int rv, PieceX &P) { load_piece_x(S.x2, ri(rv, 1), max(ri(rv, 0) & 63, 1), lane, P); };
  int recv[XPC], posv = 0, recn[XPC], posn = 0;
  PieceX xc[XPC], xn[XPC];
#pragma unroll
  for (int j = 0; j < XPC; ++j) { recv[j] = 0; recn[j] = 0; }
  if (piped) {
    issue_recs(0, recv, posv);
#pragma unroll
    for (int j = 0; j < XPC; ++j) issue_ops(recv[j], xc[j]);
  }
  unsigned round = 0;
  unsigned long long t_prev = wall_clock64();
  const int prof_rank = A.dbg >> 8;                      // GGAD_XCD_DEBUG = 4 + 16 * wave + 256 * rank: whose clocks
  const int prof_thread = ((A.dbg >> 4) & 7) * GGAD_WAVE;
  const bool prof_on = (A.dbg & 4) != 0;                 // phase clocks: accumulated in LDS by one thread, written out at the end
  __shared__ unsigned long long prof_lds[16];
  if (threadIdx.x < 16) prof_lds[threadIdx.x] = 0;
#define XCD_TICK(slot)                                                         \
  if (prof_on && rank == prof_rank && threadIdx.x == prof_thread) {                           \
    const unsigned long long t_now = wall_clock64();                           \
    prof_lds[slot] += t_now - t_prev;                                          \
    t_prev = t_now;                                                            \
  }
#define XCD_BARRIER()                                                          \
  if (!xcd_barrier(C, rank, G, ++round, A.timeout_ticks)) return;

  for (int b = 0; b < A.n_batches; ++b) {
    int row0, B, ck0, nck;
    batch_info(b, row0, B, ck0, nck);
    float *log8 = A.loss_log + (int64_t)8 * (A.log_base + b);
    // ---------------------------------------------------------------- weights of this step: L2 -> LDS (written by phase E of the previous step)
    // ALL the loads of this block first, then the LDS stores: as loops of (load, store) the compiler put an s_waitcnt vmcnt(0)
    // behind every load -- nine dependent L2 round trips (W^T 2, fc^T 4, w, the batch's n0, its label-1 sources): 1.3 us per step
    const int n0s = b < XBT ? bt_n0[b] : A.batch_n0[b];
    const int n_ra = (A.dbg & 2) ? 0 : min(B - n0s, XRA);
    const int ra_v = S.pos_meta[row0 + min(n0s + (int)threadIdx.x, B - 1)];      // sources of the generated columns (label-1 rows), for phase E
                                                                                   // (unconditional, clamped: in flight with the weights)
    const float w_v = (threadIdx.x >= XT - GGAD_WAVE && on) ? cld(params + lane) : 0.0f;
    if ((D & 1) == 0) {                                   // all three blocks start on even offsets: 8-byte loads
      constexpr int NWT = (XFT * GGAD_MAX_D / 2 + XT - 1) / XT, NFC = (GGAD_MAX_D * GGAD_MAX_D / 2 + XT - 1) / XT;
      f2 vwt[NWT], vfc[NFC];
#pragma unroll
      for (int k = 0; k < NWT; ++k) {
        const int i = 2 * ((int)threadIdx.x + k * XT);
        vwt[k] = cld2(params + L.o_Wt() + min(i, XFT * D - 2));
      }
#pragma unroll
      for (int k = 0; k < NFC; ++k) {
        const int i = 2 * ((int)threadIdx.x + k * XT);
        vfc[k] = cld2(params + L.o_fcT() + min(i, D * D - 2));
      }
#pragma unroll
      for (int k = 0; k < NWT; ++k) {
        const int i = 2 * ((int)threadIdx.x + k * XT);
        if (i < XFT * D) { wt_lds[i] = vwt[k].x; wt_lds[i + 1] = vwt[k].y; }
      }
#pragma unroll
      for (int k = 0; k < NFC; ++k) {
        const int i = 2 * ((int)threadIdx.x + k * XT);
        if (i < D * D) {
          const int r2 = i / D, c2 = i - r2 * D;
          fct[r2 * FCS + c2] = vfc[k].x; fct[r2 * FCS + c2 + 1] = vfc[k].y;
        }
      }
    } else {
      for (int i = threadIdx.x; i < XFT * D; i += XT) wt_lds[i] = cld(params + L.o_Wt() + i);
      for (int i = threadIdx.x; i < D * D; i += XT) {
        const int r2 = i / D, c2 = i - r2 * D;
        fct[r2 * FCS + c2] = cld(params + L.o_fcT() + i);
      }
    }
    if (threadIdx.x >= XT - GGAD_WAVE) w_lds[lane] = w_v;
    if ((int)threadIdx.x < n_ra) ra_lds[threadIdx.x] = ra_v >> 2;
    __syncthreads();
    float WB[XKS][XNT];                                   // W^T in the B-operand layout: W[16 t + a][4 j + g]
#pragma unroll
    for (int j = 0; j < XKS; ++j)
#pragma unroll
      for (int t = 0; t < XNT; ++t) {
        const int f = 4 * j + lg, ch = 16 * t + la;
        WB[j][t] = (f < XFT && ch < D) ? wt_lds[f * D + ch] : 0.0f;
      }
    const float wd_r = w_lds[d];
    XCD_TICK(8)

    // ================================================================ A: piece partials of relu(W x2), rows' h1
    auto fwd_piece = [&](int c, const PieceX &P) -> unsigned {
      f4 h[XNT];
      piece_fwd(P, WB, h);
      float out = 0.0f;
      unsigned pos = 0;                                    // [h2 > 0] of this lane's 16 outputs: phase C's relu mask
#pragma unroll
      for (int t = 0; t < XNT; ++t) {
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) pos |= (h[t][vv] > 0.0f ? 1u : 0u) << (4 * t + vv);                      // relu(W x2[u])   graphsage.py:419 ; entries beyond cnt are zero rows
        float st = (fmaxf(h[t][0], 0.0f) + fmaxf(h[t][1], 0.0f)) + (fmaxf(h[t][2], 0.0f) + fmaxf(h[t][3], 0.0f));
        st += __shfl_xor(st, 16, GGAD_WAVE);
        st += __shfl_xor(st, 32, GGAD_WAVE);
        out = lg == t ? st : out;                           // lane l = channel 16 (l / 16) + l % 16
      }
      S.chunk_part[(unsigned)c * 64 + lane] = out;
      return pos;
    };
    unsigned hpos[XPC];
#pragma unroll
    for (int j = 0; j < XPC; ++j) hpos[j] = 0;
    {
      const int v = v0;
#pragma unroll
      for (int j = 0; j < XPC; ++j) {
        const int w = v + j * NVW;
        if (w < nck) {
          if (!piped) { recv[j] = piece_rec(ck0 + w); issue_ops(recv[j], xc[j]); }
          hpos[j] = fwd_piece(ck0 + w, xc[j]);
        }
      }
      for (int w = v + XPC * NVW; w < nck; w += NVW) {     // more than XPC pieces per wave (a batch of > 8,000 entries)
        PieceX P;
        issue_ops(piece_rec(ck0 + w), P);
        fwd_piece(ck0 + w, P);
      }
      XCD_TICK(9)
      for (int p = NVW - 1 - v; 16 * p < B; p += NVW) {     // h1 = relu(W x1[row]), 16 rows per wave     graphsage.py:412
        PieceX P;
        load_piece_x(S.x1, row0 + 16 * p, min(16, B - 16 * p), lane, P);
        f4 h[XNT];
        piece_fwd(P, WB, h);
#pragma unroll
        for (int t = 0; t < XNT; ++t)
#pragma unroll
          for (int vv = 0; vv < 4; ++vv) {
            const int i = 16 * p + 4 * lg + vv, ch = 16 * t + la;
            if (i < B && ch < D) S.h1[(unsigned)(row0 + i) * D + ch] = fmaxf(h[t][vv], 0.0f);
          }
      }
    }
    XCD_TICK(0)
    XCD_BARRIER()
    XCD_TICK(1)

    // ================================================================ R: positions of combined_all (k_loss_pos_ck of step.hip)
    // piece order; groups of 4 loads behind wave-uniform guards (a row has 2-3 pieces), up to 32 in flight
    auto sum_row = [&](int first, int n, float &tot) {
      if (n <= 0) return;
      if (n <= 4) {                                                 // the usual row: no loop, no guards
        float pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = cld(S.chunk_part + (unsigned)(first + min(k, n - 1)) * 64 + lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) tot += (k < n) ? pv[k] : 0.0f;
        return;
      }
      constexpr int PF = 32;
      for (int c0 = 0; c0 < n; c0 += PF) {
        float pv[PF];
#pragma unroll
        for (int g4 = 0; g4 < PF / 4; ++g4) {
          if (c0 + 4 * g4 < n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[4 * g4 + k] = cld(S.chunk_part + (unsigned)(first + min(c0 + 4 * g4 + k, n - 1)) * 64 + lane);
          }
        }
#pragma unroll
        for (int g4 = 0; g4 < PF / 4; ++g4) {
          if (c0 + 4 * g4 < n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tot += (c0 + 4 * g4 + k < n) ? pv[4 * g4 + k] : 0.0f;
          }
        }
      }
    };
    const int xhub = (A.dbg & 1) ? (1 << 30) : XHUB;
    {
      const int vw = rank;
      for (int q0 = vw; q0 < B; q0 += NVW) {                        // (workgroup-uniform trip count: the hub pass below synchronises)
        const int q = q0 + nv * wid;
        const bool act = q < B;
        const int pr = (piped && q == v0) ? posv : pos_rec(row0 + (act ? q : 0));
        const int meta = ri(pr, 0);
        const int src = meta >> 2, y = meta & 1;
        const bool from_gen = (meta & 2) != 0;
        const int row = row0 + q;
        const int qa = ri(pr, 1), nq = act ? ri(pr, 2) : 0, rq = ri(pr, 3), sa = ri(pr, 4), ns = (act && from_gen) ? ri(pr, 5) : 0, rs = ri(pr, 6);
        const float hs_r = cld(S.h1 + (unsigned)src * D + d);
        // a hub row (thousands of entries = hundreds of pieces) is summed by ALL waves of the workgroup, an eighth each, the eight
        // partial sums added in order; every other row by its own wave
        const bool my_hub = nq > xhub || ns > xhub;
        if (lane == 0) {
          hub_first[2 * wid] = qa; hub_n[2 * wid] = nq > xhub ? nq : 0;
          hub_first[2 * wid + 1] = sa; hub_n[2 * wid + 1] = ns > xhub ? ns : 0;
          if (my_hub) hub_any = 1;
        }
        float totq = 0.0f, tots = 0.0f;
        if (nq <= xhub) sum_row(qa, nq, totq);
        if (ns <= xhub) sum_row(sa, ns, tots);
        __syncthreads();
        const int any_hub = hub_any;                                 // (most batches: no hub row in this workgroup, nothing to walk)
        for (int sidx = 0; any_hub && sidx < 2 * XW; ++sidx) {
          const int n = hub_n[sidx];
          if (n == 0) continue;
          const int first = hub_first[sidx];
          const int seg = ((n + XW - 1) / XW + 3) & ~3;
          const int lo = min(wid * seg, n), hi = min(lo + seg, n);
          float part = 0.0f;
          sum_row(first + lo, hi - lo, part);
          hub_part[wid][lane] = part;
          __syncthreads();
          if (wid == (sidx >> 1)) {
            float tt = 0.0f;
#pragma unroll
            for (int u = 0; u < XW; ++u) tt += hub_part[u][lane];
            if (sidx & 1) tots = tt; else totq = tt;
          }
          __syncthreads();
        }
        if (any_hub) {                                               // clear for the next use (all waves have read it)
          __syncthreads();
          if (threadIdx.x == 0) hub_any = 0;
        }
        XCD_TICK(15)
        if (!act) continue;
        const float nb_r = (1.0f / (float)rq) * totq;                                            // mask_row = mask / rowsum  graphsage.py:317
        if (on) S.nbar[(unsigned)row * D + lane] = nb_r;                                          // to_feats_neigh[q, :]
        float c_r = hs_r;                                                                        // combined_all[:, q] = h1[src] ...
        if (from_gen) {                                                                          // ... or gen[src] = relu(fc nbar[src])  :428-430
          const float nbm = on ? (1.0f / (float)rs) * tots : 0.0f;
          float a = 0.0f;
          a = lds_matvec(fct + d, FCS, nbm);
          c_r = fmaxf(a, 0.0f);
          if (on) S.gen[(unsigned)src * D + lane] = c_r;
        }
        XCD_TICK(13)
        const float wd = on ? wd_r : 0.0f, c = on ? c_r : 0.0f, nb = on ? nb_r : 0.0f;
        const float hs = (on && from_gen) ? hs_r : 0.0f;
        const PosVals pv = eval_position(wd, c, nb);
        float recn_ = 0.0f;
        if (from_gen) { const float dl2 = hs - c; recn_ = sqrtf(wave_sum_fast(dl2 * dl2)); }     // recon2   graphsage.py:197-198
        XCD_TICK(14)
        const float o0 = (1.0f - (float)y) * pv.s - log_sigmoid(pv.s);                           // BCEWithLogits, pos_weight 1 :246
        const float sv = lane == 0 ? pv.s : lane == 1 ? pv.aff : lane == 2 ? pv.na : lane == 3 ? pv.nbn : recn_;
        if (lane < 5) A.pos_scal[(unsigned)q * 8 + lane] = sv;
        const float ov = lane == 0 ? o0 : lane == 1 ? (y == 0 ? pv.aff : 0.0f) : lane == 2 ? (y == 1 ? pv.aff : 0.0f)
                       : lane == 3 ? recn_ : lane == 4 ? (y == 0 ? 1.0f : 0.0f) : (y == 1 ? 1.0f : 0.0f);
        if (lane < 6) A.pos_o[(unsigned)lane * A.ld_o + q] = ov;
      }
    }
    XCD_TICK(2)
    XCD_BARRIER()
    XCD_TICK(3)

    // ================================================================ C: loss scalars, row coefficients per piece, dW partial
    auto row_load = [&](int rv) {                         // what the coefficients of a piece's row read (written in phases A / R)
      RowIn in;
      const int row = ri(rv, 0) >> 6, fl2 = ri(rv, 2), q1 = ri(rv, 4), m2 = ri(rv, 5);
      const int i = row - row0;
      const unsigned off = (unsigned)row * D + d;
      in.H1 = cld(S.h1 + off);
      in.NB = cld(S.nbar + off);
      in.Gl = (fl2 & 1) ? cld(S.gen + off) : 0.0f;
      in.pp = cld(A.pos_scal + (unsigned)(lane < 8 ? q1 : i) * 8 + l8);
      in.nbq = cld(S.nbar + (unsigned)(row0 + q1) * D + d);
      in.c2 = cld(((m2 & 2) ? S.gen : S.h1) + (unsigned)(m2 >> 2) * D + d);
      in.xr = (fl2 & 2) ? S.x1[(unsigned)row * XFT + fl] : 0.0f;      // first piece of its row: the row's own item
      return in;
    };
    RowIn inc[XPC];
    if (piped) {
#pragma unroll
      for (int j = 0; j < XPC; ++j) inc[j] = row_load(v0 + j * NVW < nck ? recv[j] : recv[0]);   // in flight across the reduction below
    }
    if (wid < 6) {                                          // the reduction tree of k_loss_pos_ck (groups of 4) + k_loss_rows, one
      const int nwg = loss_nwg(B);                         // scalar per wave: two loads and one wave sum on the critical path
      const int k = wid;
      float v = 0.0f;
      for (int g = lane; g < nwg; g += GGAD_WAVE) {
        const float *po = A.pos_o + (unsigned)k * A.ld_o + 4 * g;
        const f2 oa = cld2(po), ob = cld2(po + 2);
        const float o1 = 4 * g + 1 < B ? oa.y : 0.0f, o2 = 4 * g + 2 < B ? ob.x : 0.0f, o3 = 4 * g + 3 < B ? ob.y : 0.0f;
        v += (oa.x + o1) + (o2 + o3);
      }
      const float tk = wave_sum_fast(v);
      if (lane == 0) t_lds[k] = tk;
    }
    const bool more = piped && b + 1 < A.n_batches;
    if (more) issue_recs(b + 1, recn, posn);                // next step's records: behind this phase's own loads
    if (threadIdx.x == XT - 1) {                            // Adam scalars of this step: double pow(), evaluated by k_xcd_prep for every
      sc[0] = A.adam_sc[2 * b];                            // batch of the chunk (300 f64 instructions in one wave per step stalled its
      sc[1] = A.adam_sc[2 * b + 1];                        // whole workgroup at the barrier below for > 1 us)
    }
    __syncthreads();
    XCD_TICK(10)
    float t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = t_lds[k];
    const float fB = (float)B;
    const float cls = t[0] / fB;
    const float an = t[1] / t[4], ab = t[2] / t[5];
    const float mg = 1.0f - (an - ab);                                     // confidence_margin = 1      graphsage.py:236-240
    const float active = (mg >= 0.0f) ? 1.0f : 0.0f;                       // clamp_min backward: pass where x >= min
    const float rec_coef = 0.1f / t[5];
    const int n0 = (int)t[4], n1 = (int)t[5];
    if (rank == 0 && threadIdx.x == 0) {
      const float margin = fmaxf(mg, 0.0f), rec = t[3] / t[5];
      log8[0] = cls + margin + 0.1f * rec;                                 // graphsage.py:258
      log8[1] = cls; log8[2] = margin; log8[3] = rec;
      log8[4] = rec_coef; log8[5] = active; log8[6] = t[4]; log8[7] = t[5];
      *S.step_counter = step0 + b + 1;
    }
    // backward coefficients of a piece's row (k_loss_rows of step.hip); first piece of the row: its wave also owns the row's x1
    // item, dz and the d w term
    // (the coefficients use hardware reciprocals, 1 ulp, where the launch chain divides: ~1e-7 relative, inside the tolerance the
    //  two paths are compared with; the loss values above are computed with true divisions)
    const float inv_fB = 1.0f / fB, inv_t4 = 1.0f / t[4], inv_t5 = 1.0f / t[5];
    auto qd = [](float a, float b2) { return a * __frcp_rn(b2); };
    auto row_coefs = [&](int rv, const RowIn &in, float &cg, float &ca) {
      const int row = ri(rv, 0) >> 6, fl2 = ri(rv, 2), r = ri(rv, 3);
      const int y = fl2 & 1, y1 = (fl2 >> 2) & 1;
      const bool first = (fl2 & 2) != 0;
      const int i = row - row0;
      const unsigned off = (unsigned)row * D + d;
      const float H1 = in.H1, NB = in.NB, Gl = in.Gl, pp = in.pp, nbq = in.nbq, c2 = in.c2;
      const float s1 = rl(pp, 0), aff1 = rl(pp, 1), na1 = rl(pp, 2), nbn1 = rl(pp, 3), recn_ = rl(pp, 4);
      const float aff2 = rl(pp, 9), na2 = rl(pp, 10), nbn2 = rl(pp, 11);
      const float Gv = (y == 1) ? Gl : 0.0f;
      const float Cc = (y == 1) ? Gv : H1;                                 // this row's column of combined_all
      const float wd = on ? wd_r : 0.0f;
      const float nac1 = fmaxf(na1, 1e-8f), nbc1 = fmaxf(nbn1, 1e-8f);
      const float ds = (qd(1.0f, 1.0f + __expf(-s1)) - (float)y1) * inv_fB;
      const float gq1 = active * (y1 == 0 ? -inv_t4 : inv_t5);
      const float cA = na1 > 0.0f ? qd(Cc, na1) : 0.0f;
      const float inac1 = __frcp_rn(nac1);
      const float dC = ds * wd + gq1 * ((nbq * __frcp_rn(nbc1)) * inac1 - (aff1 * inac1) * cA);
      float gH = dC, gG = 0.0f;
      if (y == 1) {                                                        // recon term 0.1 * mean_i |h1_i - gen_i|  graphsage.py:258
        const float tt = rec_coef * qd(H1 - Gv, recn_);
        gH = tt; gG = dC - tt;
      }
      const float nac2 = fmaxf(na2, 1e-8f), nbc2 = fmaxf(nbn2, 1e-8f);
      const float gq2 = active * (y == 0 ? -inv_t4 : inv_t5);
      const float cb = nbn2 > 0.0f ? qd(NB, nbn2) : 0.0f;
      const float inbc2 = __frcp_rn(nbc2);
      float dNb = gq2 * ((c2 * __frcp_rn(nac2)) * inbc2 - (aff2 * inbc2) * cb);
      if (y == 1) {
        const float dZ = (Gv > 0.0f) ? gG : 0.0f;                          // relu(fc(.))
        if (first && on) S.dz[off] = dZ;
        const float dZm = on ? dZ : 0.0f;
        float a = 0.0f;
        a = lds_matvec(fct + d * FCS, 1, dZm);                                                  // fc^T dZ: fc[dd][d] = fct[d][dd]
        dNb += a;
      }
      ca = (on && H1 > 0.0f) ? gH : 0.0f;
      cg = on ? dNb * __frcp_rn((float)r) : 0.0f;
      if (first) A.gw_row[(unsigned)i * GGAD_WAVE + lane] = on ? ds * Cc : 0.0f;       // d w = sum_q ds_q * combined_all[:, q]
    };
    {
      const int vw = rank, v = v0;
      f4 dacc[XNT];                                         // dW[16 t + 4 g + v'][f = a], f < 16
      float d16[XNT];                                       // dW[16 t + a][16] (partial over the lane group)
#pragma unroll
      for (int t4 = 0; t4 < XNT; ++t4) { dacc[t4] = f4{0.0f, 0.0f, 0.0f, 0.0f}; d16[t4] = 0.0f; }
      auto bwd_piece = [&](int rv, const RowIn &in, const PieceX &P, unsigned pos) {
        float cg, ca;
        row_coefs(rv, in, cg, ca);
        //XCD_TICK(13)
        float x16[4];
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) x16[vv] = __shfl(P.a[XKS - 1], 4 * lg + vv, GGAD_WAVE);   // X[4 g + v][16] (lanes 0..15 of k-step 4)
#pragma unroll
        for (int t4 = 0; t4 < XNT; ++t4) {
          const float cgt = __shfl(cg, 16 * t4 + la, GGAD_WAVE);
#pragma unroll
          for (int vv = 0; vv < 4; ++vv) {
            const float cf = ((pos >> (4 * t4 + vv)) & 1u) ? cgt : 0.0f;                     // [h2 > 0] * coef_g[row][ch]
            dacc[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, P.b[vv], dacc[t4], 0, 0, 0);
            d16[t4] = fmaf(cf, x16[vv], d16[t4]);
          }
        }
        if (ri(rv, 2) & 2) {                                // the row's own item coef_a (x) x1[row]: one more k-step, entry slot g = 0
          const float xb = lg == 0 ? in.xr : 0.0f;           // lanes 0..15 of in.xr = features 0..15, lane 16 = feature 16
          const float x16r = rl(in.xr, 16);
#pragma unroll
          for (int t4 = 0; t4 < XNT; ++t4) {
            const float cat = __shfl(ca, 16 * t4 + la, GGAD_WAVE);
            const float af = lg == 0 ? cat : 0.0f;
            dacc[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, xb, dacc[t4], 0, 0, 0);
            d16[t4] = fmaf(af, x16r, d16[t4]);
          }
        }
        //XCD_TICK(14)
      };
#pragma unroll
      for (int j = 0; j < XPC; ++j) {
        const int w = v + j * NVW;
        if (w < nck) {
          if (!piped) {
            recv[j] = piece_rec(ck0 + w);
            issue_ops(recv[j], xc[j]);
            inc[j] = row_load(recv[j]);
          }
          bwd_piece(recv[j], inc[j], xc[j], hpos[j]);
        }
      }
      for (int w = v + XPC * NVW; w < nck; w += NVW) {
        const int rv = piece_rec(ck0 + w);
        PieceX P;
        issue_ops(rv, P);
        const RowIn in = row_load(rv);
        f4 h[XNT];
        piece_fwd(P, WB, h);                                // beyond the XPC masks kept from phase A: h2 again, exactly as there
        unsigned pos = 0;
#pragma unroll
        for (int t4 = 0; t4 < XNT; ++t4)
#pragma unroll
          for (int vv = 0; vv < 4; ++vv) pos |= (h[t4][vv] > 0.0f ? 1u : 0u) << (4 * t4 + vv);
        bwd_piece(rv, in, P, pos);
      }
      if (more) {                                           // next step's operands: behind every load of this phase, they land
#pragma unroll
        for (int jj = 0; jj < XPC; ++jj) issue_ops(recn[jj], xn[jj]);     // during the combine, the barrier and phase E
      }
      // dW partial of the virtual workgroup: its 8 waves in a fixed tree
      XCD_TICK(11)
      __syncthreads();
      XCD_TICK(12)
      float *mine = accw[wid];
#pragma unroll
      for (int t4 = 0; t4 < XNT; ++t4) {
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) mine[dw_pos(la, 16 * t4 + 4 * lg + vv)] = dacc[t4][vv];
        float s16 = d16[t4];
        s16 += __shfl_xor(s16, 16, GGAD_WAVE);
        s16 += __shfl_xor(s16, 32, GGAD_WAVE);
        if (lg == 0) mine[dw_pos(16, 16 * t4 + la)] = s16;
      }
      __syncthreads();
      float *out = A.dw_part + (unsigned)vw * XFT * GGAD_WAVE;
      for (int i = 4 * threadIdx.x; i < XFT * GGAD_WAVE; i += 4 * XT) {     // [f][ch], four channels per thread
        f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = dw_pos(i >> 6, (i & 63) + k);
          o[k] = ((accw[0][p] + accw[1][p]) + (accw[2][p] + accw[3][p])) + ((accw[4][p] + accw[5][p]) + (accw[6][p] + accw[7][p]));
        }
        *reinterpret_cast<f4 *>(out + i) = o;
      }
    }
    XCD_TICK(4)
    XCD_BARRIER()
    XCD_TICK(5)

    // ================================================================ E: gradient reduction, [exchange,] Adam
    // workgroup vw owns  W[:, f = vw] (vw < 17),  w (vw == 17),  fc[vw], fc[vw + nv], fc[vw + 2 nv];  wave = sub-reducer
    {
      const int vw = rank;
      float gW = 0.0f, gw = 0.0f, gf[3] = {0.0f, 0.0f, 0.0f};
      int ddk[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) ddk[k] = vw + k * nv;      // (nv >= 22: three rows cover D <= 64)
      int pidx = -1, sel = 0;
      if (wid == 0 && vw < XFT && on) { pidx = L.o_W() + lane * XFT + vw; sel = 0; }
      if (wid == 1 && vw == XFT && on) { pidx = lane; sel = 64; }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (wid == 2 + k && ddk[k] < D && on) { pidx = L.o_fc() + ddk[k] * D + lane; sel = 128 + 64 * k; }
      // every load of the wave first (one round trip), then the arithmetic: label-1 rows wid, wid + 8, ... in blocks of EU;
      // the optimiser state of the parameter this thread owns comes with them (through the L2: the 128-byte lines of params /
      // exp_avg / exp_avg_sq hold words owned by other compute units -- W is strided by F)
      constexpr int EU = 8, PWN = XMAXWG / XW;
      float pw[PWN];
      if (vw < XFT) {
#pragma unroll
        for (int k = 0; k < PWN; ++k) pw[k] = cld(A.dw_part + (unsigned)min(wid * PWN + k, nv - 1) * XFT * GGAD_WAVE + vw * GGAD_WAVE + lane);
      }
      float ap = 0.0f, am = 0.0f, av = 0.0f;
      if (wid < 5) {
        const int pl = pidx >= 0 ? pidx : 0;
        ap = cld(params + pl); am = cld(S.exp_avg + pl); av = cld(S.exp_avg_sq + pl);
      }
      {
        for (int j0 = wid; j0 < n1; j0 += XW * EU) {         // label-1 rows = sources of the last n1 columns, in order
          float dzr[EU], nbr[EU];
#pragma unroll
          for (int u = 0; u < EU; ++u) {
            const int j = min(j0 + u * XW, n1 - 1);
            const int ra = (j < XRA && !(A.dbg & 2)) ? ra_lds[j] : (S.pos_meta[row0 + n0 + j] >> 2);
            dzr[u] = cld(S.dz + (unsigned)ra * D + d);
            nbr[u] = cld(S.nbar + (unsigned)ra * D + d);
          }
#pragma unroll
          for (int u = 0; u < EU; ++u) {
            if (j0 + u * XW < n1) {
              const float dzm = on ? dzr[u] : 0.0f;
#pragma unroll
              for (int k = 0; k < 3; ++k) gf[k] = fmaf(rl(dzm, min(ddk[k], GGAD_MAX_D - 1)), nbr[u], gf[k]);   // d fc[dd][d2] = sum_i dZ_i[dd] * nbar_i[d2]
            }
          }
        }
      }
      if (vw < XFT) {
#pragma unroll
        for (int k = 0; k < PWN; ++k) gW += (wid * PWN + k < nv) ? pw[k] : 0.0f;
      } else if (vw == XFT) {
        for (int i0 = wid; i0 < B; i0 += XW * EU) {
          float gr[EU];
#pragma unroll
          for (int u = 0; u < EU; ++u) gr[u] = cld(A.gw_row + (unsigned)min(i0 + u * XW, B - 1) * GGAD_WAVE + lane);
#pragma unroll
          for (int u = 0; u < EU; ++u) gw += (i0 + u * XW < B) ? gr[u] : 0.0f;
        }
      }
      __syncthreads();
      accw[wid][lane] = gW; accw[wid][64 + lane] = gw;
#pragma unroll
      for (int k = 0; k < 3; ++k) accw[wid][128 + 64 * k + lane] = gf[k];
      __syncthreads();
      if (pidx >= 0) {
        float g = 0.0f;
#pragma unroll
        for (int k = 0; k < XW; ++k) g += accw[k][sel + lane];              // fixed order
        S.grads[pidx] = g;
        if (MODE == 2) g = xchg_sum(A.X, A.xstep0 + (uint32_t)b + 1u, pidx, g) * A.grad_scale;
        adam_apply(params, S.exp_avg, S.exp_avg_sq, L, pidx, ap, am, av, g, S.weight_decay, sc[0], sc[1]);
      }
    }
    XCD_TICK(6)
    XCD_BARRIER()
    XCD_TICK(7)
    if (more) {                                             // step b + 1's records and operands become the current ones
#pragma unroll
      for (int j = 0; j < XPC; ++j) { recv[j] = recn[j]; xc[j] = xn[j]; }
      posv = posn;
    }
  }
  if (prof_on && rank == prof_rank && threadIdx.x == prof_thread)
    for (int k = 0; k < 16; ++k) C->prof[k] = prof_lds[k];
  if (rank == 0 && threadIdx.x == 0) C->done = 1;
}

// ------------------------------------------------------------------ records + entry-complete x2 (once per chunk, whole chip)
struct XcdPrepArgs {
  const int32_t *batch_ptr, *ent_ptr, *row_ck_ptr, *ck_rc, *ck_e0, *ent_own, *labels, *pos_meta, *row_pos;
  float *x2;
  int32_t *ck_rec, *pos_rec, *batch_n0;
  float *adam_sc;
  const int32_t *step_counter;
  float lr;
  int n_batches, n_rows, n_pieces, n_ents;
  unsigned *ctrl_clear;                 // the control block of the launch that follows, or null: zeroed here (no memset launch)
  int ctrl_words;
};
__global__ void __launch_bounds__(256) k_xcd_prep(XcdPrepArgs P) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (P.ctrl_clear && tid < P.ctrl_words) P.ctrl_clear[tid] = 0u;
  auto batch_row0 = [&](int row) {                       // first row of the batch that holds `row`
    int lo = 0, hi = P.n_batches;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.batch_ptr[mid] <= row) lo = mid; else hi = mid; }
    return P.batch_ptr[lo];
  };
  if (P.adam_sc && tid < P.n_batches) {                   // adam_scalars of step.hip for optimiser step step0 + tid + 1
    const double ts = (double)(*P.step_counter + tid + 1);
    const double bc1 = 1.0 - pow(0.9, ts), bc2 = 1.0 - pow(0.999, ts);
    P.adam_sc[2 * tid] = (float)((double)P.lr / bc1);       // step_size
    P.adam_sc[2 * tid + 1] = (float)sqrt(bc2);              // bias_correction2_sqrt
  }
  if (tid < P.n_pieces) {
    const int c = tid;
    const int rc = P.ck_rc[c], row = rc >> 6;
    const int row0 = batch_row0(row);
    const int q1 = P.row_pos[row];
    int32_t *o = P.ck_rec + (int64_t)c * REC;
    o[0] = rc;
    o[1] = P.ck_e0[c];
    o[2] = (P.labels[row] & 1) | ((c == P.row_ck_ptr[row]) ? 2 : 0) | ((P.pos_meta[row0 + q1] & 1) << 2);
    o[3] = P.ent_ptr[row + 1] - P.ent_ptr[row];
    o[4] = q1;
    o[5] = P.pos_meta[row];
    o[6] = 0; o[7] = 0;
  }
  if (tid < P.n_rows) {
    const int row = tid;
    const int meta = P.pos_meta[row];
    int32_t *o = P.pos_rec + (int64_t)row * REC;
    o[0] = meta;
    o[1] = P.row_ck_ptr[row];
    o[2] = P.row_ck_ptr[row + 1] - P.row_ck_ptr[row];
    o[3] = P.ent_ptr[row + 1] - P.ent_ptr[row];
    int sa = 0, ns = 0, rs = 1;
    if (meta & 2) {
      const int src = meta >> 2;
      sa = P.row_ck_ptr[src]; ns = P.row_ck_ptr[src + 1] - sa; rs = P.ent_ptr[src + 1] - P.ent_ptr[src];
    }
    o[4] = sa; o[5] = ns; o[6] = rs; o[7] = 0;
    // the columns of combined_all are the label-0 rows, then the label-1 rows (graphsage.py:450; bit 1 = the source row's label;
    // bit 0 is the label of ROW q in original order, the reference's quirk): n0 = the first column with a label-1 source
    int lo = 0, hi = P.n_batches;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.batch_ptr[mid] <= row) lo = mid; else hi = mid; }
    const int r0 = P.batch_ptr[lo], r1 = P.batch_ptr[lo + 1];
    if ((meta & 2) && (row == r0 || !(P.pos_meta[row - 1] & 2))) P.batch_n0[lo] = row - r0;
    if (!(meta & 2) && row == r1 - 1) P.batch_n0[lo] = r1 - r0;
  }
  // x2 is stored at owner entries (the reference's deduplicated unique_nodes_list, graphsage.py:306); the other entries of a
  // (batch, column) get a copy of their owner's row, so that the rows of a piece are the consecutive entries [e0, e0 + cnt)
  // (half a wave per entry, lane = feature: one read of the owner per entry, not one per float)
  // four entries per trip, their owner ids first, then the four rows (every load unconditional: an owner re-reads its own row),
  // then the stores: one entry per trip was a chain of two dependent round trips per entry, ten entries per thread
  const int f = min((int)(threadIdx.x & 31), XFT - 1);
  const bool fl = (threadIdx.x & 31) < XFT;
  const int64_t stride = ((int64_t)gridDim.x * 256) >> 5;
  for (int64_t e0 = tid >> 5; e0 < P.n_ents; e0 += 4 * stride) {
    int64_t e[4];
    int o[4];
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { e[q] = min(e0 + q * stride, (int64_t)P.n_ents - 1); o[q] = P.ent_own[e[q]]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = P.x2[(int64_t)o[q] * XFT + f];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (fl && e0 + q * stride < P.n_ents && o[q] != (int)e[q]) P.x2[e[q] * XFT + f] = v[q];
  }
}

// ------------------------------------------------------------------ L2 warmer
// The plan's outputs (x2, ent_own, piece / row tables) were written by kernels on the other XCDs: the chunk kernel's first touch
// of every line is an HBM / Infinity-Cache round trip (1-2 us under the plan's traffic) on a chain of 2-3 DEPENDENT loads per
// phase.  A few single-wave workgroups of this kernel -- the ones that land on the chunk kernel's XCD -- stay `ahead` steps in
// front of it and touch every line a step will read (one lane per 128-byte line), so the chunk kernel's loads are served by
// the shared L2.  It only reads; it never delays the chunk kernel (own waves, own vmcnt) and leaves when that kernel does.
struct XcdWarmArgs {
  XcdCtrl *ctrl;
  const int32_t *batch_ptr, *ent_ptr, *row_ck_ptr, *ck_rc, *ck_e0, *ent_own, *labels, *pos_meta, *row_pos;
  const float *x1, *x2;
  int n_batches, ahead;
  unsigned launch_id;
  unsigned long long timeout_ticks;
};
constexpr int XWARM = 8;              // helper workgroups that stay (one wave each)

__device__ __forceinline__ void warm_range(const void *base, int64_t byte0, int64_t byte1, int r, int lane) {
  const char *p = reinterpret_cast<const char *>(base);
  const int64_t l0 = byte0 >> 7, l1 = (byte1 + 127) >> 7;             // 128-byte lines [l0, l1)
  for (int64_t l = l0 + r + (int64_t)XWARM * lane; l < l1; l += (int64_t)XWARM * GGAD_WAVE) {
    const int v = *reinterpret_cast<const volatile int *>(p + (l << 7));
    asm volatile("" ::"v"(v));
  }
}

__global__ void __launch_bounds__(GGAD_WAVE) k_xcd_warm(XcdWarmArgs P) {
  XcdCtrl *C = P.ctrl;
  const int lane = lane_id();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(&C->placed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != P.launch_id) {
    __builtin_amdgcn_s_sleep(8);
    if (cldu(&C->done) != 0 || cldu(&C->err) != 0 || wall_clock64() - t0 > P.timeout_ticks) return;
  }
  if (cldu(&C->xcc) != xcc) return;
  int r = 0;
  if (lane == 0) r = (int)__hip_atomic_fetch_add(&C->helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r = __builtin_amdgcn_readfirstlane(r);
  if (r >= XWARM) return;
  for (int s = 0; s < P.n_batches; ++s) {
    while ((int)(cldu(&C->slot[0]) >> 2) < s - P.ahead) {              // 4 barrier rounds per step
      __builtin_amdgcn_s_sleep(4);
      if (cldu(&C->done) != 0 || wall_clock64() - t0 > P.timeout_ticks) return;
    }
    const int row0 = P.batch_ptr[s], row1 = P.batch_ptr[s + 1];
    const int64_t e0 = P.ent_ptr[row0], e1 = P.ent_ptr[row1];
    const int64_t c0 = P.row_ck_ptr[row0], c1 = P.row_ck_ptr[row1];
    warm_range(P.x2, e0 * XFT * 4, e1 * XFT * 4, r, lane);
    warm_range(P.ent_own, e0 * 4, e1 * 4, r, lane);
    warm_range(P.ck_rc, c0 * 4, c1 * 4, r, lane);
    warm_range(P.ck_e0, c0 * 4, c1 * 4, r, lane);
    warm_range(P.x1, (int64_t)row0 * XFT * 4, (int64_t)row1 * XFT * 4, r, lane);
    warm_range(P.labels, (int64_t)row0 * 4, (int64_t)row1 * 4, r, lane);
    warm_range(P.pos_meta, (int64_t)row0 * 4, (int64_t)row1 * 4, r, lane);
    warm_range(P.row_pos, (int64_t)row0 * 4, (int64_t)row1 * 4, r, lane);
    warm_range(P.ent_ptr, (int64_t)row0 * 4, (int64_t)row1 * 4 + 4, r, lane);
    warm_range(P.row_ck_ptr, (int64_t)row0 * 4, (int64_t)row1 * 4 + 4, r, lane);
  }
}

}  // namespace

__global__ void k_xcd_probe(int32_t *out) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = (int32_t)(xcc & 7u);
  }
}

extern "C" {

int32_t ggad_mb_xcd_grid(void) { return 8 * XMAXWG; }

int ggad_xcd_first_of_stream(int32_t *first_host, ggad_stream_t stream) {
  GGAD_REQUIRE(first_host);
  int32_t *d = nullptr;
  if (hipMalloc(&d, 16 * sizeof(int32_t)) != hipSuccess) return GGAD_E_LAUNCH;
  hipStream_t st = as_stream(stream);
  int32_t h[16];
  int rc = GGAD_OK;
  k_xcd_probe<<<dim3(16), dim3(GGAD_WAVE), 0, st>>>(d);
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    rc = GGAD_E_LAUNCH;
  (void)hipFree(d);
  if (rc != GGAD_OK) return rc;
  for (int b = 0; b < 16; ++b)
    if (h[b] != ((h[0] + b) & 7)) return GGAD_E_INVALID;      // not the rotation the skipping launches rely on
  *first_host = h[0];
  return GGAD_OK;
}

int64_t ggad_mb_xcd_workspace_elems(int32_t max_rows, int32_t D, int32_t F, int64_t rows_cap, int64_t pieces_cap) {
  if (max_rows < 1 || D < 1 || F < 1 || rows_cap < 0 || pieces_cap < 0) return 0;
  // (the per-batch table at the end holds at most rows_cap entries: a batch has at least one row)
  const int64_t ld = ((int64_t)max_rows + 3) / 4 * 4;
  return 256 + (int64_t)max_rows * 8 + 8 * ld + (int64_t)max_rows * GGAD_WAVE + (int64_t)XMAXWG * F * GGAD_WAVE +
         (rows_cap + pieces_cap + 2) * REC + 3 * (rows_cap + 1);
}

int64_t ggad_mb_xcd_record_elems(int64_t rows_cap, int64_t pieces_cap) {
  if (rows_cap < 0 || pieces_cap < 0) return 0;
  return (rows_cap + pieces_cap + 2) * REC + (rows_cap + 1);
}

namespace {
// layout of a record block: position records, piece records, first generated column per batch
inline void record_views(int32_t *recs, int64_t rows_cap, int64_t pieces_cap, int32_t *&pos_rec, int32_t *&ck_rec, int32_t *&batch_n0) {
  pos_rec = recs;
  ck_rec = recs + (rows_cap + 1) * REC;
  batch_n0 = recs + (rows_cap + pieces_cap + 2) * REC;
}
int launch_records(const ggad_mb_step &s, int32_t n_batches, const int32_t *batch_ptr_dev, int32_t n_rows, int32_t n_pieces, int32_t n_ents,
                   int32_t *pos_rec, int32_t *ck_rec, int32_t *batch_n0, float *adam_sc, hipStream_t st, void *ctrl_clear = nullptr) {
  XcdPrepArgs Q;
  Q.ctrl_clear = static_cast<unsigned *>(ctrl_clear);
  Q.ctrl_words = (int)(sizeof(XcdCtrl) / sizeof(unsigned));
  Q.batch_ptr = batch_ptr_dev; Q.ent_ptr = s.ent_ptr; Q.row_ck_ptr = s.row_ck_ptr; Q.ck_rc = s.ck_rc; Q.ck_e0 = s.ck_e0;
  Q.ent_own = s.ent_own; Q.labels = s.labels; Q.pos_meta = s.pos_meta; Q.row_pos = s.row_pos;
  Q.x2 = const_cast<float *>(s.x2);
  Q.ck_rec = ck_rec; Q.pos_rec = pos_rec; Q.batch_n0 = batch_n0;
  Q.adam_sc = adam_sc; Q.step_counter = s.step_counter; Q.lr = s.lr;
  Q.n_batches = n_batches; Q.n_rows = n_rows; Q.n_pieces = n_pieces; Q.n_ents = n_ents;
  const int64_t work = std::max<int64_t>(std::max<int64_t>(std::max<int64_t>(std::max<int64_t>(n_rows, n_pieces), (int64_t)n_ents * 8), n_batches),
                                         ctrl_clear ? Q.ctrl_words : 0);
  const unsigned blocks = (unsigned)std::min<int64_t>((work + 255) / 256, 8192);
  k_xcd_prep<<<dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st>>>(Q);
  GGAD_CHECK_LAUNCH("mb_train_chunk_xcd (records)");
  return GGAD_OK;
}
}  // namespace

/* The records of a chunk (and x2 completed per entry) on a stream of the caller's choice -- the PLAN's: in the overlapped trainer
 * the chunk kernel's stream owns 28 compute units of one XCD, where this whole-chip pass took 0.45 ms per chunk on its critical
 * path.  `records`: int32[ggad_mb_xcd_record_elems(rows_cap, pieces_cap)], owned by the chunk (double-buffered with it); hand the
 * same block to ggad_mb_train_chunk_xcd. */
int ggad_mb_xcd_prepare(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr_dev, int32_t n_rows, int32_t n_pieces,
                        int32_t n_ents, int64_t rows_cap, int64_t pieces_cap, int32_t *records, ggad_stream_t stream) {
  GGAD_REQUIRE(tmpl && batch_ptr_dev && records && n_batches >= 0 && n_rows >= 0 && n_pieces >= 0 && n_ents >= 0);
  GGAD_REQUIRE(n_rows <= rows_cap && n_pieces <= pieces_cap && (int64_t)n_ents * XFT < ((int64_t)1 << 31));
  const ggad_mb_step &s = *tmpl;
  GGAD_REQUIRE(s.x2 && s.ent_ptr && s.ent_own && s.labels && s.pos_meta && s.row_pos && s.row_ck_ptr && s.ck_rc && s.ck_e0 && s.F == XFT);
  if (n_batches == 0) return GGAD_OK;
  int32_t *pos_rec, *ck_rec, *batch_n0;
  record_views(records, rows_cap, pieces_cap, pos_rec, ck_rec, batch_n0);
  return launch_records(s, n_batches, batch_ptr_dev, n_rows, n_pieces, n_ents, pos_rec, ck_rec, batch_n0, nullptr, as_stream(stream));
}

/* The dense steps of a whole chunk as ONE launch resident on one XCD (see the header of this file).  tmpl as for
 * ggad_mb_train_chunk (row_ck_ptr / ck_rc / ck_e0 / chunk_part REQUIRED, F == 17); batch_ptr: DEVICE int32[n_batches + 1];
 * workspace: float[ggad_mb_xcd_workspace_elems(largest batch, D, F)], 16-byte aligned; xchg NULL = single GPU. */
int ggad_mb_train_chunk_xcd(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr_dev, int32_t max_rows,
                            int32_t n_rows, int32_t n_pieces, int32_t n_ents, int64_t rows_cap, int64_t pieces_cap, float *loss_log,
                            int32_t log_base, float *workspace, float grad_scale, ggad_xchg *xchg, int32_t n_wg, const int32_t *records,
                            ggad_stream_t stream) {
  GGAD_REQUIRE(tmpl && batch_ptr_dev && loss_log && workspace && n_batches >= 0 && log_base >= 0 && max_rows >= 1);
  GGAD_REQUIRE(n_rows >= 0 && n_pieces >= 0 && n_ents >= 0 && n_rows <= rows_cap && n_pieces <= pieces_cap);
  GGAD_REQUIRE((int64_t)n_ents * XFT < ((int64_t)1 << 31) && (int64_t)n_pieces * 64 < ((int64_t)1 << 31));      // 32-bit element offsets
  const ggad_mb_step &s = *tmpl;
  GGAD_REQUIRE(s.params && s.exp_avg && s.exp_avg_sq && s.grads && s.step_counter && s.x1 && s.x2 && s.ent_ptr && s.ent_own &&
               s.labels && s.pos_meta && s.row_pos && s.h1 && s.nbar && s.gen && s.dz && s.row_ck_ptr && s.ck_rc && s.ck_e0 &&
               s.chunk_part);
  GGAD_REQUIRE(s.F == XFT && s.D >= 1 && s.D <= GGAD_MAX_D && ((uintptr_t)workspace & 15) == 0);
  if (n_batches == 0) return GGAD_OK;
  XcdArgs A;
  A.s = s;
  A.batch_ptr = batch_ptr_dev;
  A.n_batches = n_batches; A.log_base = log_base;
  A.ld_o = (max_rows + 3) / 4 * 4;
  A.loss_log = loss_log;
  A.ctrl = reinterpret_cast<XcdCtrl *>(workspace);
  A.pos_scal = workspace + 256;
  A.pos_o = A.pos_scal + (int64_t)max_rows * 8;
  A.gw_row = A.pos_o + (int64_t)8 * A.ld_o;
  A.dw_part = A.gw_row + (int64_t)max_rows * GGAD_WAVE;
  int32_t *recs = reinterpret_cast<int32_t *>(A.dw_part + (int64_t)XMAXWG * XFT * GGAD_WAVE);
  int32_t *w_pos, *w_ck, *w_n0;
  record_views(recs, rows_cap, pieces_cap, w_pos, w_ck, w_n0);
  A.adam_sc = reinterpret_cast<const float *>(w_n0 + rows_cap + 1);
  if (records) {                                            // prepared by ggad_mb_xcd_prepare (on the plan's stream)
    int32_t *p, *c, *n0;
    record_views(const_cast<int32_t *>(records), rows_cap, pieces_cap, p, c, n0);
    A.pos_rec = p; A.ck_rec = c; A.batch_n0 = n0;
  } else {
    A.pos_rec = w_pos; A.ck_rec = w_ck; A.batch_n0 = w_n0;
  }
  A.grad_scale = grad_scale;
  static const unsigned long long timeout = [] {            // barrier time-out in seconds (wall clock, 100 MHz ticks)
    const char *e = getenv("GGAD_XCD_TIMEOUT_S");
    const double sec = e ? atof(e) : 10.0;
    return (unsigned long long)((sec > 0.001 ? sec : 0.001) * 1e8);
  }();
  A.timeout_ticks = timeout;
  static const int dbg = [] { const char *e = getenv("GGAD_XCD_DEBUG"); return e ? atoi(e) : 0; }();
  A.dbg = dbg;
  // workgroups that stay (= compute units of the chosen XCD the stream may use) and the XCD: GGAD_XCD_WGS / GGAD_XCD_ID
  static const int env_nv = [] { const char *e = getenv("GGAD_XCD_WGS"); return e ? atoi(e) : 0; }();
  static const int env_xcd = [] { const char *e = getenv("GGAD_XCD_ID"); return e ? atoi(e) : 0; }();
  A.nv = n_wg > 0 ? n_wg : (env_nv > 0 ? env_nv : XMAXWG);
  A.want_xcd = env_xcd & 7;
  GGAD_REQUIRE(A.nv >= XMINWG + 4 && A.nv <= XMAXWG && A.nv % 4 == 0);     // 24 / 28 / 32: three fc rows per workgroup cover 64 channels; the
                                                                            // XCD's four shader engines get the same number of workgroups
  static unsigned launch_seq = 0;
  A.launch_id = ++launch_seq ? launch_seq : ++launch_seq;      // never 0 (the cleared control block)
  hipStream_t st = as_stream(stream);
  // registration counters, barrier slots and the error word start from zero on every launch (profile clocks too): cleared by the
  // prep launch below (a memset is a launch of its own: 4 us + a gap in front of every chunk)
  {  // records + entry-complete x2: one whole-chip launch per chunk (the plan's tables are read-only for everybody else) -- or,
     // when the caller prepared them, only the Adam scalars of the chunk's steps (they need the step counter as it is NOW)
    const int rc = records ? launch_records(s, n_batches, batch_ptr_dev, 0, 0, 0, nullptr, nullptr, nullptr, const_cast<float *>(A.adam_sc), st, workspace)
                           : launch_records(s, n_batches, batch_ptr_dev, n_rows, n_pieces, n_ents, const_cast<int32_t *>(A.pos_rec),
                                            const_cast<int32_t *>(A.ck_rec), const_cast<int32_t *>(A.batch_n0),
                                            const_cast<float *>(A.adam_sc), st, workspace);
    if (rc) return rc;
  }
  // the L2 warmer runs BESIDE the chunk kernel on a stream of its own: it starts once the control block is cleared (ev0) and the
  // caller's stream continues only after it has left (ev1: it reads the plan's buffers)
  static const int warm_ahead = [] { const char *e = getenv("GGAD_XCD_WARM"); return e ? atoi(e) : 0; }();   // off: measured without effect
  struct Side { hipStream_t st; hipEvent_t ev0, ev1; };
  static Side side[GGAD_MAX_DEVICES] = {};
  Side *sd = nullptr;
  if (warm_ahead > 0) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < GGAD_MAX_DEVICES) {
      sd = &side[dev];
      if (!sd->st) {
        if (hipStreamCreateWithFlags(&sd->st, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sd->ev0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd->ev1, hipEventDisableTiming) != hipSuccess) {
          sd->st = nullptr;
          sd = nullptr;
        }
      }
    }
    if (sd && (hipEventRecord(sd->ev0, st) != hipSuccess || hipStreamWaitEvent(sd->st, sd->ev0, 0) != hipSuccess)) sd = nullptr;
    (void)hipGetLastError();
  }
  if (xchg) {
    GGAD_REQUIRE(xchg->view.n >= s.D + s.D * s.F + s.D * s.D);
    for (int q = 0; q < xchg->view.world; ++q) GGAD_REQUIRE(xchg->view.peer[q] != nullptr);
    A.X = xchg->view;
    A.xstep0 = xchg->step;
    xchg->step += (uint32_t)n_batches;
    if (s.D == 64) k_train_chunk_xcd<2, 64><<<dim3(8 * A.nv), dim3(XT), 0, st>>>(A);
    else k_train_chunk_xcd<2, 0><<<dim3(8 * A.nv), dim3(XT), 0, st>>>(A);
  } else {
    A.X = ggad_xchg_view{};
    A.xstep0 = 0;
    if (s.D == 64) k_train_chunk_xcd<1, 64><<<dim3(8 * A.nv), dim3(XT), 0, st>>>(A);
    else k_train_chunk_xcd<1, 0><<<dim3(8 * A.nv), dim3(XT), 0, st>>>(A);
  }
  GGAD_CHECK_LAUNCH("mb_train_chunk_xcd");
  if (sd) {
    XcdWarmArgs W;
    W.ctrl = A.ctrl;
    W.batch_ptr = batch_ptr_dev; W.ent_ptr = s.ent_ptr; W.row_ck_ptr = s.row_ck_ptr; W.ck_rc = s.ck_rc; W.ck_e0 = s.ck_e0;
    W.ent_own = s.ent_own; W.labels = s.labels; W.pos_meta = s.pos_meta; W.row_pos = s.row_pos;
    W.x1 = s.x1; W.x2 = s.x2;
    W.n_batches = n_batches; W.ahead = warm_ahead; W.launch_id = A.launch_id; W.timeout_ticks = timeout;
    k_xcd_warm<<<dim3(8 * 2 * XWARM), dim3(GGAD_WAVE), 0, sd->st>>>(W);
    if (hipGetLastError() == hipSuccess && hipEventRecord(sd->ev1, sd->st) == hipSuccess) (void)hipStreamWaitEvent(st, sd->ev1, 0);
    (void)hipGetLastError();
  }
  return GGAD_OK;
}

/* Control words of the LAST launch that used `workspace` (device -> host copy on `stream`, synchronises it):
 * out[0] error (0 ok, 1 barrier time-out, 2 registration time-out, 3 placement) -- of the last launch, or, when that one was clean,
 * the first error of any launch since ggad_mb_xcd_clear_error (sticky); out[1] workgroups that stayed, out[2] their XCD,
 * out[3..10] phase clocks of rank 0 in 10 ns ticks (A, barrier, R, barrier, C, barrier, E, barrier). */
int ggad_mb_xcd_status(const float *workspace, int64_t *out19, ggad_stream_t stream) {
  GGAD_REQUIRE(workspace && out19);
  int64_t *out11 = out19;
  union { XcdCtrl c; unsigned w[256]; } h;
  hipStream_t st = as_stream(stream);
  if (hipMemcpyAsync(&h, workspace, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) return GGAD_E_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return GGAD_E_LAUNCH;
  out11[0] = h.c.err ? h.c.err : h.w[XCD_STICKY_WORD]; out11[1] = h.c.survivors; out11[2] = h.c.xcc;
  for (int k = 0; k < 16; ++k) out11[3 + k] = (int64_t)h.c.prof[k];
  return GGAD_OK;
}

/* Clears the error words of `workspace` (the sticky one and the last launch's; after the host has dealt with the error), or --
 * code != 0 -- sets both as a launch that timed out would (tests of the host's recovery path). */
int ggad_mb_xcd_clear_error(float *workspace, int32_t code, ggad_stream_t stream) {
  GGAD_REQUIRE(workspace && code >= 0);
  const unsigned v = (unsigned)code;
  hipStream_t st = as_stream(stream);
  if (hipMemcpyAsync(reinterpret_cast<unsigned *>(workspace) + XCD_STICKY_WORD, &v, sizeof(v), hipMemcpyHostToDevice, st) != hipSuccess) return GGAD_E_LAUNCH;
  // ... and the error word of the LAST launch with it (a launch that times out writes both): when the failed launch was the last
  // resident one (the host fell back to the launch chain) nothing else would ever reset it and ggad_mb_xcd_status would keep reporting it
  if (hipMemcpyAsync(reinterpret_cast<char *>(workspace) + offsetof(XcdCtrl, err), &v, sizeof(v), hipMemcpyHostToDevice, st) != hipSuccess)
    return GGAD_E_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return GGAD_E_LAUNCH;
  return GGAD_OK;
}

}  // extern "C"
