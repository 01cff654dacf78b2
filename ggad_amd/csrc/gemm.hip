// FP32 GEMM on the gfx950 matrix cores for the dense projections of the full-graph GGAD path.
//
// Replaces the reference's nn.Linear / torch.mm call sites on the hot path (model.py:27 gcn fc, :156 fc4,
// :176-180 scorer MLP, and their autograd dgrad / wgrad).  Inputs are fp32 and the result must match the
// reference's fp32 CPU arithmetic to ~1e-6, so the kernel uses the exact-f32 MFMA
// v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain, no reduced-precision path exists on gfx950).
//
//   C[m][n] = epilogue( sum_k A(m,k) * B(k,n) )        A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
//
// so NN / NT / TN are the same kernel with different strides.  Workgroup = 4 waves = a 128 x 64 tile of C
// (each wave two 32 x 32 MFMA accumulators), K stepped by 32 through LDS with register prefetch of the next
// K-tile; both LDS tiles are k-major
// (As[k][m], Bs[k][n]) because the MFMA fragment of lane l is A[m = l & 31][k = l >> 5] /
// B[k = l >> 5][n = l & 31]: 32 consecutive lanes read 32 consecutive words -> conflict-free ds_read_b32.
// Long-K / few-tile shapes (weight gradients: K = number of nodes) are split along K over gridDim.z into
// a workspace and summed by a second kernel in a fixed order (deterministic, no float atomics).
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32;      // BM: the large row tile; BM / 2 is used when the grid would not fill the chip
constexpr int LDB_S = BN + 1;                       // +1 pad: the transposing LDS store is conflict-free

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Workgroup = 4 waves = a 128 x 64 tile of C; wave (wr, wc) owns rows [64 wr, 64 wr + 64) x cols [32 wc, 32 wc + 32):
// two 32x32 MFMA accumulators.  K is stepped by 32; the next K-tile is fetched from global memory into registers
// while the current one is consumed from LDS (global latency hidden behind 32 MFMAs = 2048 cycles per wave).
// Register diet (the first version needed 186 VGPRs = 2 waves per SIMD and left the MFMA pipe idle 48 % of the time,
// rocprofv3 SQ_WAIT_INST_ANY): the operand layouts are template parameters, every thread keeps ONE 32-bit offset per
// element relative to a wave-uniform tile pointer (SGPR base + VGPR offset addressing), the row / column bounds are
// folded into those offsets once (out-of-range elements point at element 0 and are zeroed by a mask bit), and only
// the last K-tile checks k.
template <bool A_KFAST, bool B_NFAST, int TM, bool VEC, bool PART>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) k_gemm_f32(const float *__restrict__ A, const float *__restrict__ B,
                                                  float *__restrict__ C, int M, int N, int K, int sam, int sak,
                                                  int sbk, int sbn, int64_t ldc, const float *__restrict__ bias,
                                                  int relu, int k_per_split, int64_t c_split_stride) {
  constexpr int NACC = TM / 64;                // 32x32 accumulators per wave (rows 32 * NACC)
  __shared__ float As[BK][TM + 1];
  __shared__ float Bs[BK][LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  floatx16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  floatx16 acc1 = acc0;
  // Thread -> tile elements.  Scalar mode: element p = one float, unit stride across threads along the operand's fastest
  // index.  VEC mode (extents along the fastest index are multiples of 4, checked by the launcher): element p = one
  // float4 along the fastest index -- a quarter of the load instructions and of the address arithmetic; the K = 300
  // projections are bound by exactly that (per K-tile the MFMA phase is 0.4 us, the 24 dword loads per thread were not).
  constexpr int VW = VEC ? 4 : 1;
  constexpr int NA = TM * BK / 256 / VW, NB = BN * BK / 256 / VW;      // elements (floats or float4s) per thread
  int oa[NA], ob[NB];                    // offsets relative to the tile pointers (0 for out-of-range rows / columns)
  unsigned amask = 0, bmask = 0;         // bit p: row (column) inside the matrix
  // A element p -> (m, k) of its first float; B element p -> (n, k)
  auto a_mk = [&](int p, int &m, int &k) {
    if constexpr (VEC) {
      if (A_KFAST) { k = (tid & 7) * 4; m = (tid >> 3) + 32 * p; }
      else { m = (tid % (TM / 4)) * 4; k = tid / (TM / 4) + (1024 / TM) * p; }
    } else {
      if (A_KFAST) { k = tid & 31; m = (tid >> 5) + 8 * p; }
      else { m = tid % TM; k = tid / TM + (256 / TM) * p; }
    }
  };
  auto b_nk = [&](int p, int &n, int &k) {
    if constexpr (VEC) {
      if (B_NFAST) { n = (tid & 15) * 4; k = (tid >> 4) + 16 * p; }
      else { k = (tid & 7) * 4; n = (tid >> 3) + 32 * p; }
    } else {
      if (B_NFAST) { n = tid & 63; k = (tid >> 6) + 4 * p; }
      else { k = tid & 31; n = (tid >> 5) + 8 * p; }
    }
  };
  // static validity per float of every element along the non-k index (bit VW * p + j); a float4 along m (n) may straddle the
  // edge of the matrix when M (N) is not a multiple of 4
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    int m, k; a_mk(p, m, k);
    const bool in = m0 + m < M;
    oa[p] = in ? m * sam + k * sak : 0;
#pragma unroll
    for (int j = 0; j < VW; ++j) amask |= ((m0 + m + ((VEC && !A_KFAST) ? j : 0) < M) ? 1u : 0u) << (VW * p + j);
  }
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    int n, k; b_nk(p, n, k);
    const bool in = n0 + n < N;
    ob[p] = in ? k * sbk + n * sbn : 0;
#pragma unroll
    for (int j = 0; j < VW; ++j) bmask |= ((n0 + n + ((VEC && B_NFAST) ? j : 0) < N) ? 1u : 0u) << (VW * p + j);
  }
  // can a float4 of this workgroup straddle an edge along m / n?  (uniform; along k it depends on the K-tile, see fetch)
  const bool a_edge = VEC && PART && !A_KFAST && (m0 + TM > M) && (M & 3);
  const bool b_edge = VEC && PART && B_NFAST && (n0 + BN > N) && (N & 3);
  const float *At = A + (int64_t)m0 * sam + (int64_t)k_begin * sak;     // wave-uniform tile pointers
  const float *Bt = B + (int64_t)n0 * sbn + (int64_t)k_begin * sbk;
  typedef float fvec __attribute__((ext_vector_type(4)));
  float ra[NA * VW], rb[NB * VW];
  unsigned oka = 0, okb = 0;             // validity of the elements fetched last
  auto fetch = [&](int k0) {
    const int klim = k_end - k0;         // k >= klim lies past the end of this split (last K-tile only)
    oka = 0; okb = 0;
    // VEC: float4 loads unless one of them could straddle an edge in THIS tile (uniform test) -- then the same elements are
    // fetched float by float with per-float masks (last K-tile when the split length is not a multiple of 4, edge row /
    // column tiles when M / N are not)
    // (PART = false is instantiated for shapes whose extents are all multiples of 4: no second code path at all)
    const bool partial = VEC && PART && (a_edge || b_edge || ((klim < BK) && (klim & 3) && (A_KFAST || !B_NFAST)));
    if (!partial) {
#pragma unroll
      for (int p = 0; p < NA; ++p) {     // branch-free: every load is issued before the first one is consumed
        int m, k; a_mk(p, m, k);
        const unsigned ok = ((amask >> (VW * p)) & 1u) & (unsigned)(k < klim);
        oka |= (ok ? ((1u << VW) - 1u) : 0u) << (VW * p);
        const float *src = At + (ok ? oa[p] : 0);
        if constexpr (VEC) {
          const fvec v = *reinterpret_cast<const fvec __attribute__((aligned(4))) *>(src);
          ra[4 * p] = v.x; ra[4 * p + 1] = v.y; ra[4 * p + 2] = v.z; ra[4 * p + 3] = v.w;
        } else {
          ra[p] = *src;
        }
      }
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        int n, k; b_nk(p, n, k);
        const unsigned ok = ((bmask >> (VW * p)) & 1u) & (unsigned)(k < klim);
        okb |= (ok ? ((1u << VW) - 1u) : 0u) << (VW * p);
        const float *src = Bt + (ok ? ob[p] : 0);
        if constexpr (VEC) {
          const fvec v = *reinterpret_cast<const fvec __attribute__((aligned(4))) *>(src);
          rb[4 * p] = v.x; rb[4 * p + 1] = v.y; rb[4 * p + 2] = v.z; rb[4 * p + 3] = v.w;
        } else {
          rb[p] = *src;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NA; ++p) {
        int m, k; a_mk(p, m, k);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          const unsigned ok = ((amask >> (VW * p + j)) & 1u) & (unsigned)(k + (A_KFAST ? j : 0) < klim);
          oka |= ok << (VW * p + j);
          ra[VW * p + j] = At[ok ? oa[p] + j : 0];            // the fastest index has unit stride
        }
      }
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        int n, k; b_nk(p, n, k);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          const unsigned ok = ((bmask >> (VW * p + j)) & 1u) & (unsigned)(k + (B_NFAST ? 0 : j) < klim);
          okb |= ok << (VW * p + j);
          rb[VW * p + j] = Bt[ok ? ob[p] + j : 0];
        }
      }
    }
    At += (int64_t)BK * sak;             // (the masks are applied in stash(): nothing here waits for the loads)
    Bt += (int64_t)BK * sbk;
  };
  auto stash = [&]() {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      int m, k; a_mk(p, m, k);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const float v = ((oka >> (VW * p + j)) & 1u) ? ra[VW * p + j] : 0.0f;
        if (A_KFAST) As[k + j][m] = v; else As[k][m + j] = v;
      }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      int n, k; b_nk(p, n, k);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const float v = ((okb >> (VW * p + j)) & 1u) ? rb[VW * p + j] : 0.0f;
        if (B_NFAST) Bs[k][n + j] = v; else Bs[k + j][n] = v;
      }
    }
  };
  if (k_begin < k_end) {
    fetch(k_begin);
    stash();
    __syncthreads();
  }
  const int i = lane & 31, kk = lane >> 5;
  // (a second LDS buffer with one barrier per K-tile was measured: no gain -- 66 vs 63-72 TF on 39357 x 300 x 300, 106 vs 116 TF on
  // 4096^3; rocprofv3: the MFMA pipe is busy 45 % / 68 % of the kernel's cycles on those shapes, profiles/r02_gemm_mfma_pmc.csv)
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const bool more = (k0 + BK) < k_end;
    if (more) fetch(k0 + BK);
    auto mma = [&](int ks) {
      const float b = Bs[ks + kk][wc * 32 + i];
      const float a0 = As[ks + kk][wr * 32 * NACC + i];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      if constexpr (NACC == 2) {
        const float a1 = As[ks + kk][wr * 64 + 32 + i];
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
      }
    };
    if (more) {
#pragma unroll 4
      for (int ks = 0; ks < BK; ks += 2) mma(ks);
    } else {
      // the last K-tile of K = 300 holds 12 valid k: its zero-padded tail is not multiplied (6 % of the MFMAs of such a product)
      const int ks_end = min(BK, (k_end - k0 + 1) & ~1);
      for (int ks = 0; ks < ks_end; ks += 2) mma(ks);
    }
    __syncthreads();
    if (more) {
      stash();
      __syncthreads();
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float *Cout = C + (int64_t)blockIdx.z * c_split_stride;
  const int col = n0 + wc * 32 + (lane & 31);
  if (col < N) {
    const float bv = (bias != nullptr) ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 32 * NACC + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float v = acc0[r] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        Cout[(int64_t)row * ldc + col] = v;
      }
      if (NACC == 2 && row + 32 < M) {
        float v = acc1[r] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        Cout[(int64_t)(row + 32) * ldc + col] = v;
      }
    }
  }
}

// ---- round 4: the same 64 x 64 x 32 tiling with the operands staged by LDS-DMA ---------------------------------------------------
// k_gemm_f32 above brings every K-tile through registers (global -> VGPR -> masked ds_write): 156 VGPRs = 3 workgroups per CU, two
// barriers and 16 LDS stores per thread per K-tile; on the K = 300 products of the path (10 K-tiles) the matrix pipe is busy a third
// of the time.  Here (shapes whose 16-byte chunks are aligned: every N x 300 x 300 product and weight gradient of the path)
//  - both operand tiles are written by `global_load_lds_dwordx4` (no staging registers, no LDS stores, no masks: rows / columns
//    outside the matrix read a clamped row -- their outputs are never stored --, k outside the split reads a zero line), double
//    buffered, ONE barrier per K-tile; 60 VGPRs and 32 KB of LDS: 5 workgroups per CU, so a wave's barrier waits and the prologue /
//    epilogue of one workgroup hide behind the MFMAs of four others;
//  - an operand whose k index is the fast one (x and W of x W^T) keeps that layout in LDS: [row][32 k], its 16-byte chunks
//    XOR-swizzled by (row >> 1) & 7, and is read with ONE ds_read_b128 per 4 MFMA k-steps, conflict-free for the 16-lane service
//    groups of that instruction (cdna_hip_programming.md T2).  The MFMA sums over the 2 k its lane halves hold; which k only has to
//    agree between the two fragments: half h of step (j, c) takes k = 8 j + 4 h + c for both;
//  - an operand whose m / n index is the fast one keeps [32 k][64] rows and ds_read_b32 as before.
__device__ __attribute__((aligned(16))) float g_gemm_zero_line[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void gemm_dma16(const float *src, uint32_t lds_off) {
  // (inline asm: hipcc then does not know of an LDS-DMA in flight and puts no vmcnt(0) in front of the LDS reads of the OTHER buffer)
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_off), "v"(src) : "m0", "memory");
}

template <bool A_KFAST, bool B_NFAST, int NBUF>
__global__ void __launch_bounds__(256) k_gemm_dma(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
                                                  int N, int K, int sam, int sak, int sbk, int sbn, int64_t ldc,
                                                  const float *__restrict__ bias, int relu, int k_per_split, int64_t c_split_stride) {
  __shared__ __attribute__((aligned(16))) float lds[NBUF][2][2048];    // [buffer][A / B][64 x 32 floats]; NBUF - 1 K-tiles in flight
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int k_begin = blockIdx.z * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // this lane's two 16-byte pieces of each operand tile (piece p = 2 wid + u lands at LDS bytes p * 1024 + lane * 16)
  const float *srcA[2], *srcB[2];
  int kA[2], kB[2];                                                   // k offset of the piece inside the K-tile
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = 2 * wid + u;
    if (A_KFAST) {
      const int r = 8 * p + (lane >> 3), lc = (lane & 7) ^ ((r >> 1) & 7);
      kA[u] = 4 * lc;
      srcA[u] = A + (int64_t)min(m0 + r, M - 1) * sam + kA[u];
    } else {                                                          // [k][m], m fast
      const int kk = 4 * p + (lane >> 4), mm = m0 + 4 * (lane & 15);
      kA[u] = kk;
      srcA[u] = A + (int64_t)kk * sak + (mm < M ? mm : 0);
    }
    if (!B_NFAST) {                                                   // [n][k], k fast
      const int r = 8 * p + (lane >> 3), lc = (lane & 7) ^ ((r >> 1) & 7);
      kB[u] = 4 * lc;
      srcB[u] = B + (int64_t)min(n0 + r, N - 1) * sbn + kB[u];
    } else {
      const int kk = 4 * p + (lane >> 4), nn = n0 + 4 * (lane & 15);
      kB[u] = kk;
      srcB[u] = B + (int64_t)kk * sbk + (nn < N ? nn : 0);
    }
  }
  const int64_t stepA = A_KFAST ? 32 : (int64_t)32 * sak, stepB = B_NFAST ? (int64_t)32 * sbk : 32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)&lds[0][0][0];
  auto issue = [&](int k0, int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t off = lds0 + (uint32_t)buf * 16384u + (uint32_t)(2 * wid + u) * 1024u;
      gemm_dma16(k0 + kA[u] < k_end ? srcA[u] + (int64_t)((k0 - k_begin) / 32) * stepA + (A_KFAST ? k_begin : (int64_t)k_begin * sak) : g_gemm_zero_line, off);
      gemm_dma16(k0 + kB[u] < k_end ? srcB[u] + (int64_t)((k0 - k_begin) / 32) * stepB + (B_NFAST ? (int64_t)k_begin * sbk : k_begin) : g_gemm_zero_line,
                 off + 8192u);
    }
  };
  const int i = lane & 31, h = lane >> 5;
  const int rowA = wr * 32 + i, rowB = wc * 32 + i;
  // K-tiles t .. t + NBUF - 2 are in flight while tile t is multiplied; a wave issues 4 loads per tile and loads return in order,
  // so vmcnt(4 (NBUF - 2)) at the end of tile t means ITS pieces of tile t + 1 have landed (tiles past the end are issued as
  // zero lines into buffers nobody reads: the count stays uniform)
  if (k_begin < k_end) {
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) issue(k_begin + 32 * t, t);
    if (NBUF == 2) __builtin_amdgcn_s_waitcnt(0x0f70); else __builtin_amdgcn_s_waitcnt(0x0f74);      // vmcnt(0) / vmcnt(4)
    __syncthreads();
  }
  int buf = 0, nxt = NBUF - 1;
  for (int k0 = k_begin; k0 < k_end; k0 += 32) {
    issue(k0 + 32 * (NBUF - 1), nxt);                                 // (the barrier that ended the previous tile freed that buffer)
    const float *At = &lds[buf][0][0], *Bt = &lds[buf][1][0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f4 a, b;
      if (A_KFAST) a = *reinterpret_cast<const f4 *>(At + rowA * 32 + 4 * ((2 * j + h) ^ ((rowA >> 1) & 7)));
      else { a.x = At[(8 * j + 4 * h) * 64 + rowA]; a.y = At[(8 * j + 4 * h + 1) * 64 + rowA]; a.z = At[(8 * j + 4 * h + 2) * 64 + rowA]; a.w = At[(8 * j + 4 * h + 3) * 64 + rowA]; }
      if (!B_NFAST) b = *reinterpret_cast<const f4 *>(Bt + rowB * 32 + 4 * ((2 * j + h) ^ ((rowB >> 1) & 7)));
      else { b.x = Bt[(8 * j + 4 * h) * 64 + rowB]; b.y = Bt[(8 * j + 4 * h + 1) * 64 + rowB]; b.z = Bt[(8 * j + 4 * h + 2) * 64 + rowB]; b.w = Bt[(8 * j + 4 * h + 3) * 64 + rowB]; }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    if (NBUF == 2) __builtin_amdgcn_s_waitcnt(0x0f70); else __builtin_amdgcn_s_waitcnt(0x0f74);      // the next tile has landed (this wave's pieces) ...
    __syncthreads();                                                  // ... everybody's; and everybody is done reading this one
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float *Cout = C + (int64_t)blockIdx.z * c_split_stride;
  const int col = n0 + wc * 32 + (lane & 31);
  if (col < N) {
    const float bv = (bias != nullptr) ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float v = acc[r] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        Cout[(int64_t)row * ldc + col] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_splitk_reduce(const float *__restrict__ ws, int splits, int64_t split_stride,
                                                       float *__restrict__ C, int M, int N, int64_t ldc,
                                                       const float *__restrict__ bias, int relu) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
  float s = 0.0f;
  const float *__restrict__ p = ws + (int64_t)m * N + n;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {                  // eight partials in flight, added in split order (one dependent load per
    float v[8];                                      // split made this 26 us at 128 splits of a 300 x 10 gradient)
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * split_stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < splits; ++k) s += p[(int64_t)k * split_stride];
  if (bias != nullptr) s += bias[n];
  if (relu) s = fmaxf(s, 0.0f);
  C[(int64_t)m * ldc + n] = s;
}

template <int TM, bool VEC, bool PART, typename... Args>
static void launch_gemm2(bool a_kfast, bool b_nfast, dim3 grid, hipStream_t st, Args... args) {
  if (a_kfast && b_nfast) k_gemm_f32<true, true, TM, VEC, PART><<<grid, dim3(256), 0, st>>>(args...);
  else if (a_kfast) k_gemm_f32<true, false, TM, VEC, PART><<<grid, dim3(256), 0, st>>>(args...);
  else if (b_nfast) k_gemm_f32<false, true, TM, VEC, PART><<<grid, dim3(256), 0, st>>>(args...);
  else k_gemm_f32<false, false, TM, VEC, PART><<<grid, dim3(256), 0, st>>>(args...);
}
// mode 0: per-float loads; 1: float4 loads, every extent along a fastest index is a multiple of 4; 2: float4 loads with the
// per-float path for tiles where a float4 could straddle an edge
template <int TM, typename... Args>
static void launch_gemm(int mode, bool a_kfast, bool b_nfast, dim3 grid, hipStream_t st, Args... args) {
  if (mode == 1) launch_gemm2<TM, true, false>(a_kfast, b_nfast, grid, st, args...);
  else if (mode == 2) launch_gemm2<TM, true, true>(a_kfast, b_nfast, grid, st, args...);
  else launch_gemm2<TM, false, false>(a_kfast, b_nfast, grid, st, args...);
}

// ---- tall products with a SMALL second operand: op(B) resident in LDS (k_gemm_bres) -------------------------------------------------
// The projections of the full-graph path are (N nodes x 300) (300 x 300): the second operand is 360 KB, the first is streamed once.
// The 64 x 64 x 32 tiles above re-stage a 64 x 32 piece of B per K-tile and workgroup behind a barrier and run 860 workgroups of ten
// K-tiles each at Reddit size: prologue, epilogue and the per-tile hand-off are never hidden (MFMA pipe busy 0.39-0.51,
// profiles/r04_gemm_mfma_pmc.csv).  Here a workgroup keeps a SLAB of op(B) -- 32 columns x all of K, K-fast rows of KP floats, 40 KB
// -- in LDS for its whole life (3-4 workgroups per CU, a persistent grid) and its four waves walk 16-row blocks of A on their own: no
// barrier after the fill.  v_mfma_f32_16x16x4_f32 (exact f32): lane (i = lane % 16, ks = lane / 16) loads ONE float4 of A per 16 k --
// A[row i][16 s + 4 ks ..+3], straight from memory into the operand registers -- and one ds_read_b128 of the slab per column tile --
// Bt[col j][16 s + 4 ks ..+3] --: component m of both feeds MFMA m, i.e. lane group ks supplies k = 16 s + 4 ks + m (any assignment
// of k to the lane-group slots is a valid one as long as A and B agree).  8 MFMA per 16-byte load of A; the row block's K range is
// fetched in two halves, each in flight while the other is multiplied (the second half of a block with the first half of the next).
// KP = 8 or 56 mod 64: the 16 lanes of a ds_read_b128 service group then touch 64 different banks.
// Summation order per output: k ascending inside a lane group's chain is NOT the order of the tiles above (one accumulator, k-slots
// interleaved 16 apart) -- fixed, deterministic, and exact-f32 products / adds like theirs; results agree to round-off.
constexpr int BR_COLS = 32;                  // columns of a slab (two MFMA column tiles)
typedef float br_f4 __attribute__((ext_vector_type(4)));

// the row blocks rb0, rb0 + stride, ... < rb_end of one wave against the slab in LDS; TC = column tiles of the slab (2, or 1 for the
// last one).  A block's K range is fetched in two halves, each in flight while the other is multiplied (the second half of a block
// with the first half of the next).
template <int KS, int TC>
__device__ __forceinline__ void bres_rows(const float *__restrict__ A, float *__restrict__ C, const float *__restrict__ bt, int M, int N,
                                          int K, int64_t lda, int64_t ldc, const float *__restrict__ bias, int relu, int KP, int col0,
                                          int rb0, int rb_end, int stride, int lane) {
  const int li = lane & 15, ks = lane >> 4;
  const int koff = 4 * ks;
  if (rb0 >= rb_end) return;
  // the float4 of step s: k = 16 s + 4 ks; beyond K: the last valid one again (it meets the slab's zeros)
  auto load_part = [&](int rb, int s0, int cnt, float4 *dst) {
    const int row = min(rb * 16 + li, M - 1);
    const float *ar = A + (int64_t)row * lda;
#pragma unroll
    for (int q = 0; q < cnt; ++q) dst[q] = *reinterpret_cast<const float4 *>(ar + min(16 * (s0 + q) + koff, K - 4));
  };
  const float *bl = bt + li * KP + koff;
  float bv[TC];
#pragma unroll
  for (int t = 0; t < TC; ++t) bv[t] = bias ? bias[min(col0 + 16 * t + li, N - 1)] : 0.0f;
  // (the slab reads of step q + 1 are requested before the MFMAs of step q and nothing moves across a step: left to itself the
  // compiler hoists every ds_read of a half in front of its MFMAs -- 250 VGPRs and spills)
  auto mul_part = [&](int s0, int cnt, const float4 *a, br_f4 (&acc)[TC]) {
    float4 bq[TC], bn[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) bq[t] = *reinterpret_cast<const float4 *>(bl + 16 * t * KP + 16 * s0);
#pragma unroll
    for (int q = 0; q < cnt; ++q) {
      const float4 av = a[q];
      const int qn = q + 1 < cnt ? q + 1 : q;
#pragma unroll
      for (int t = 0; t < TC; ++t) bn[t] = *reinterpret_cast<const float4 *>(bl + 16 * t * KP + 16 * (s0 + qn));
      __builtin_amdgcn_sched_barrier(0);                               // (the reads stay in front of the step's MFMAs: scheduled freely
#pragma unroll                                                         //  they sat behind the seventh, their latency exposed every step)
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq[t].w, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TC; ++t) bq[t] = bn[t];
    }
  };
  auto store_block = [&](int rb, const br_f4 (&acc)[TC]) {
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      const int col = col0 + 16 * t + li;
      if (col < N) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = rb * 16 + 4 * ks + v;
          float o = acc[t][v] + bv[t];
          if (relu) o = fmaxf(o, 0.0f);
          if (row < M) C[(int64_t)row * ldc + col] = o;
        }
      }
    }
  };
  constexpr int H0 = (KS + 1) / 2, H1 = KS - H0;
  float4 a0[H0], a1[H1 > 0 ? H1 : 1];
  int rb = rb0;
  load_part(rb, 0, H0, a0);
  for (; rb < rb_end; rb += stride) {
    load_part(rb, H0, H1, a1);
    br_f4 acc[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) acc[t] = br_f4{0.f, 0.f, 0.f, 0.f};
    mul_part(0, H0, a0, acc);
    load_part(min(rb + stride, rb_end - 1), 0, H0, a0);       // the next block's first half (the last block: itself again, unused)
    mul_part(H0, H1, a1, acc);
    store_block(rb, acc);
  }
}

template <int KS>                            // K-steps of 16 (K <= 16 KS)
__global__ void __launch_bounds__(256, 3) k_gemm_bres(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                   int M, int N, int K, int64_t lda, int64_t sbk, int64_t sbn, int64_t ldc,
                                                   const float *__restrict__ bias, int relu, int KP) {
  extern __shared__ __attribute__((aligned(16))) float bt[];      // [BR_COLS][KP]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_tiles = (N + 15) >> 4, n_slabs = (n_tiles + 1) >> 1;
  // workgroup b runs on XCD b % 8 (dispatcher rotation).  An XCD owns an eighth of the row blocks and has workgroups on EVERY slab,
  // dealt in proportion to the slabs' column tiles (the last slab may hold one), all walking the XCD's row blocks in the same order:
  // the ten slabs read a row block of A at about the same time, so its second to tenth reads are served by that XCD's L2 (slabs
  // dealt chip-wide re-read A from the fabric once per slab: 470 MB for T-Finance's 47 MB, 30 of 115 us).
  const int GX = gridDim.x >> 3, x = blockIdx.x & 7, bi = blockIdx.x >> 3;
  int slab = 0, beg = 0, end = 0;
  for (int s2 = 0; s2 < n_slabs; ++s2) {
    const int e2 = (int)((int64_t)GX * min(2 * s2 + 2, n_tiles) / n_tiles);
    if (bi >= e2) { slab = s2 + 1; beg = e2; }
    else { end = e2; break; }
  }
  const int col0 = slab * BR_COLS, tcount = min(2, n_tiles - 2 * slab);
  const int gs = end - beg, g = bi - beg;
  const int RB = (M + 15) >> 4;
  const int rb_lo = (int)((int64_t)RB * x >> 3), rb_hi = (int)((int64_t)RB * (x + 1) >> 3);
  // ---- fill: Bt[n][k] = op(B)[k][col0 + n], zero beyond N / K
  const int kpad = KS * 16;
  if (sbk == 1) {                                                  // K-fast in memory: rows copied (float4)
    for (int i = threadIdx.x; i < BR_COLS * (kpad >> 2); i += 256) {
      const int n = i / (kpad >> 2), k4 = (i - n * (kpad >> 2)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col0 + n < N && k4 < K) v = *reinterpret_cast<const float4 *>(B + (int64_t)(col0 + n) * sbn + k4);      // (K % 4 == 0)
      *reinterpret_cast<float4 *>(bt + n * KP + k4) = v;
    }
  } else {                                                         // N-fast in memory: transposed on the way in.  A wave takes 64
    // consecutive k (lane = k) and walks the eight float4 of their 128-byte lines: the scalar stores of a lane group then fall on
    // consecutive banks (lanes over the columns instead -- coalesced reads -- put eight lanes on every bank: 4 KP = 0 mod 32; the
    // conflict counter showed a third of the LDS cycles of the launch there); the lines stay in the L1 between the eight passes
    for (int kb = 64 * wid; kb < kpad; kb += 256) {
      const int k = kb + lane;
#pragma unroll
      for (int n4 = 0; n4 < BR_COLS; n4 += 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && col0 + n4 < N) v = *reinterpret_cast<const float4 *>(B + (int64_t)k * sbk + col0 + n4);      // (N % 4 == 0)
        if (k < kpad) { bt[(n4 + 0) * KP + k] = v.x; bt[(n4 + 1) * KP + k] = v.y; bt[(n4 + 2) * KP + k] = v.z; bt[(n4 + 3) * KP + k] = v.w; }
      }
    }
  }
  __syncthreads();
  if (tcount == 2) bres_rows<KS, 2>(A, C, bt, M, N, K, lda, ldc, bias, relu, KP, col0, rb_lo + g * 4 + wid, rb_hi, gs * 4, lane);
  else bres_rows<KS, 1>(A, C, bt, M, N, K, lda, ldc, bias, relu, KP, col0, rb_lo + g * 4 + wid, rb_hi, gs * 4, lane);
}


// ---- small products (round 5): the 239 ... 843-row products of the outlier head (fc4 forward, its two gradients: model.py:151-156) ran as
// 20 workgroups walking ten 32-deep K tiles behind ten barriers -- 12-15 us each for 0.04 GFLOP, three per epoch.  Here a workgroup takes a
// 32 x 32 tile, stages BOTH operand panels for the WHOLE K in LDS at once (one memory round trip instead of ten), and its four waves run
// one 16 x 16 tile each: two interleaved accumulator chains over the k-steps, operands by ds_read_b128 (k-slot assignment as in the
// slab kernel), bias / ReLU in the epilogue.  Any layout (strides), any M, N; K <= 512.
constexpr int SM_T = 32;
__host__ __device__ inline int sm_kp(int k16) {             // row stride (floats) >= k16 that is 8 or 56 mod 64: conflict-free b128 fragment reads
  int kp = k16;
  while ((kp & 63) != 8 && (kp & 63) != 56) kp += 4;
  return kp;
}
__global__ void __launch_bounds__(256) k_gemm_small(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N,
                                                    int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc,
                                                    const float *__restrict__ bias, int relu, int KP) {
  extern __shared__ __attribute__((aligned(16))) float sm_lds[];      // As[32][KP], Bs[32][KP]  (row = m / n, k fastest)
  float *As = sm_lds, *Bs = sm_lds + SM_T * KP;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m0 = blockIdx.y * SM_T, n0 = blockIdx.x * SM_T;
  const int k16 = (K + 15) & ~15;
  // fill: along whichever index is fastest in memory, a batch of loads in flight before the first store (one at a time the 75 loads of
  // a thread were 75 round trips: 21 us); rows / k beyond the matrix read a clamped element and are stored as zero
  typedef float f4 __attribute__((ext_vector_type(4)));
  auto fill = [&](float *dst, const float *src, int r0, int R, int64_t sr, int64_t sk) {
    if (sk == 1 && (K & 3) == 0 && (sr & 3) == 0 && ((uintptr_t)src & 15) == 0) {
      // k fastest, 16-byte rows: thread = (row t / 8, float4 t % 8 + 8 q): eight threads read 128 contiguous bytes of a row
      const int r = tid >> 3, c0 = tid & 7, k4 = K >> 2, n4 = k16 >> 2;
      const float *rp = src + (int64_t)min(r0 + r, R - 1) * sr;
      constexpr int NB = 8;
      for (int q0 = 0; q0 * 8 < n4; q0 += NB) {
        f4 v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = *reinterpret_cast<const f4 *>(rp + 4 * min(c0 + 8 * (q0 + q), k4 - 1));
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int c = c0 + 8 * (q0 + q);
          if (c < n4) *reinterpret_cast<f4 *>(dst + r * KP + 4 * c) = (r0 + r < R && c < k4) ? v[q] : f4{0.f, 0.f, 0.f, 0.f};
        }
      }
    } else if (sk == 1) {                                     // k fastest, unaligned: thread = (row t / 8, k t % 8 + 8 q)
      const int r = tid >> 3, c0 = tid & 7;
      const float *rp = src + (int64_t)min(r0 + r, R - 1) * sr;
      constexpr int NB = 16;
      for (int q0 = 0; q0 * 8 < k16; q0 += NB) {
        float v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = rp[min(c0 + 8 * (q0 + q), K - 1)];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int k = c0 + 8 * (q0 + q);
          if (k < k16) dst[r * KP + k] = (r0 + r < R && k < K) ? v[q] : 0.0f;
        }
      }
    } else {                                                  // row index fastest: thread = (k t / 32 + 8 q, row t % 32): a wave reads two k of 32 rows
      const int r = tid & 31, kq = tid >> 5;
      const float *rp = src + (int64_t)min(r0 + r, R - 1) * sr;
      constexpr int NB = 16;
      for (int q0 = 0; q0 * 8 < k16; q0 += NB) {
        float v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = rp[(int64_t)min(kq + 8 * (q0 + q), K - 1) * sk];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int k = kq + 8 * (q0 + q);
          if (k < k16) dst[r * KP + k] = (r0 + r < R && k < K) ? v[q] : 0.0f;
        }
      }
    }
  };
  fill(As, A, m0, M, sam, sak);
  fill(Bs, B, n0, N, sbn, sbk);
  __syncthreads();
  const int li = lane & 15, ks = lane >> 4;
  const int wr = wid >> 1, wc = wid & 1;
  const float *ap = As + (wr * 16 + li) * KP + 4 * ks, *bp = Bs + (wc * 16 + li) * KP + 4 * ks;
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // the B fragment goes first: the tile comes out with lane (li, ks) holding C[row li][columns 4 ks .. + 3] (16-byte rows of the store)
  for (int s = 0; s < k16; s += 16) {
    const f4 a = *reinterpret_cast<const f4 *>(ap + s), b = *reinterpret_cast<const f4 *>(bp + s);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a.w, acc1, 0, 0, 0);
  }
  const f4 acc = acc0 + acc1;
  const int row = m0 + wr * 16 + li, col = n0 + wc * 16 + 4 * ks;
  if (row < M) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (col + v < N) {
        float o = acc[v] + (bias ? bias[col + v] : 0.0f);
        if (relu) o = fmaxf(o, 0.0f);
        C[(int64_t)row * ldc + col + v] = o;
      }
    }
  }
}

static int bres_kp(int kpad) {               // smallest row stride >= kpad that is 8 or 56 mod 64
  int kp = kpad;
  while ((kp & 63) != 8 && (kp & 63) != 56) kp += 4;
  return kp;
}

// Row tile: 64 x 64 workgroup tiles everywhere.  The 128-row tile (two accumulators per wave) was the default for grids of
// >= 3 workgroups per CU; measured again in round 2 it loses on every shape of the path (39357 x 300 x 300: 105-107 vs 98 us,
// 4096^3: 1269 vs 1239 us, Photo's K = 745: 79 vs 60 us).  GGAD_GEMM_TM=128 still selects it for experiments.
static int pick_tm(int M, int N) {
  static const int forced = [] { const char *e = getenv("GGAD_GEMM_TM"); return e ? atoi(e) : 0; }();
  (void)M; (void)N;
  return forced == 128 ? 128 : 64;
}
}  // namespace

extern "C" {

// split-K factor of the tiled kernels for an M x N x K product (1 = no split); the ONE place that decides it -- the workspace query sizes
// for it, the launches below use it (they used to divide the workspace size, which is the max() with the row-range kernel's: ADVICE round 5)
static int gemm_splits(int32_t M, int32_t N, int32_t K) {
  const int tm = pick_tm(M, N);
  const int64_t tiles = (int64_t)((M + tm - 1) / tm) * ((N + BN - 1) / BN);
  // (the bar was K >= 2048 with >= 256 of K per split: the weight gradients of the scorer MLP on Reddit / Photo / Amazon -- K = the
  // 1,200-1,900 rows the loss reads, 1-15 tiles -- then ran as a handful of workgroups walking 58 K-tiles each: 92-99 us per
  // call, three calls per epoch, a quarter of a 1 ms epoch)
  if (K < 512 || tiles >= 256) return 1;
  // tiles x splits <= 512 workgroups = two per CU, all resident at once.  (Rounded UP -- 25 tiles x 21 splits = 525 -- the last 13
  // workgroups of a 300 x 300 weight gradient ran as a round of their own: 127 -> 106 us at K = 39,357, 46 -> 42 at 11,944;
  // scripts/gemm_split_sweep.sh.  GGAD_GEMM_SPLIT_FLOOR=0 restores the old rounding.)
  static const int split_target = [] { const char *e = getenv("GGAD_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 512; }();
  static const int split_floor = [] { const char *e = getenv("GGAD_GEMM_SPLIT_FLOOR"); return e ? atoi(e) : 1; }();
  int splits = split_floor ? (int)std::max<int64_t>(split_target / tiles, 1) : (int)((split_target + tiles - 1) / tiles);
  const int max_splits = K >= 2048 ? (K + 255) / 256 : (K + 127) / 128;
  if (splits > max_splits) splits = max_splits;
  if (splits > 128) splits = 128;
  return splits < 1 ? 1 : splits;
}

int64_t ggad_gemm_workspace_elems(int32_t M, int32_t N, int32_t K) {
  const int splits = gemm_splits(M, N, K);
  // (a "TN" product with a long K may take the row-range kernel of mlp.hip instead: its partials fit too)
  return splits <= 1 ? 0 : std::max<int64_t>((int64_t)splits * M * N, ggad_int_wgrad_tn_ws(K, M, N));
}

int ggad_gemm_f32(const float *A, const float *B, float *C, int32_t M, int32_t N, int32_t K, int64_t sam, int64_t sak,
                  int64_t sbk, int64_t sbn, int64_t ldc, const float *bias, int32_t relu, float *workspace,
                  ggad_stream_t stream) {
  GGAD_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 0 && ldc >= N);
  // 32-bit element offsets inside a tile row range (BM rows x K) / (K x BN): every stride times its extent must fit
  GGAD_REQUIRE((sam == 1 || sak == 1) && (sbk == 1 || sbn == 1));
  GGAD_REQUIRE((int64_t)BM * sam + (int64_t)K * sak < (1LL << 31) && (int64_t)BN * sbn + (int64_t)K * sbk < (1LL << 31));
  if (M == 0 || N == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  const int tm = pick_tm(M, N);
  const int gx = (N + BN - 1) / BN, gy = (M + tm - 1) / tm;
  const bool a_kfast = (sak == 1), b_nfast = (sbn == 1);
  // float4 loads along each operand's fastest index; tiles in which a float4 could straddle the end of a row / column / split
  // fall back to per-float loads inside the kernel, so nothing is read outside the tensors for any M, N, K
  static const bool force_scalar = [] { const char *e = getenv("GGAD_GEMM_SCALAR"); return e && e[0] == '1'; }();
  const bool aligned = ((a_kfast ? K : M) % 4 == 0) && ((b_nfast ? N : K) % 4 == 0);
  const int vec = force_scalar ? 0 : (aligned ? 1 : 2);
  const int64_t ws_elems = workspace ? ggad_gemm_workspace_elems(M, N, K) : 0;
  // tall product, small second operand: op(B) resident in LDS
  // round 5: 80-column slabs, one persistent workgroup per CU, row blocks dealt to SIMDs (gemm_slab.hip); what it declines goes on below
  if (a_kfast && ws_elems == 0) {
    const int r = ggad_int_gemm_slab(A, B, C, (int)M, (int)N, (int)K, sam, sbk, sbn, ldc, bias, (int)relu, st);
    if (r < 0) { ggad_set_error(hipErrorLaunchFailure, "gemm_f32 (slab)"); return GGAD_E_LAUNCH; }
    if (r > 0) return GGAD_OK;
  }
  static const int bres = [] { const char *e = getenv("GGAD_GEMM_BRES"); return e ? atoi(e) : 1; }();
  // (measured, scripts/gemm_bres_ab.py: 39,357 rows 105-108 -> 92-97 us; at 7,500-12,000 rows -7 ... +5 %, inside the box-to-box noise --
  //  three waves per SIMD walking 2.1-2.3 row blocks each leave a third round that is 15 % full -- so those keep the tiled kernel)
  static const int bres_min_m = [] { const char *e = getenv("GGAD_GEMM_BRES_MIN_M"); return e ? atoi(e) : 16384; }();
  static const int bres_wgs = [] { const char *e = getenv("GGAD_GEMM_BRES_WGS"); return e ? atoi(e) : 3; }();       // workgroups per CU (146 VGPRs, 40 KB of LDS)
  if (bres && a_kfast && ws_elems == 0 && M >= bres_min_m && K >= 128 && K <= 320 && K % 4 == 0 && N > 64 && N <= 1024 &&
      (((uintptr_t)A & 15) == 0) && sam % 4 == 0 && (((uintptr_t)B & 15) == 0) &&
      ((sbk == 1 && sbn % 4 == 0) || (sbn == 1 && sbk % 4 == 0 && N % 4 == 0))) {
    const int KSn = (K + 15) / 16, KP = bres_kp(KSn * 16);
    const size_t lds = (size_t)BR_COLS * KP * sizeof(float);
    const int G = 256 * std::max(1, std::min(bres_wgs, (int)(160 * 1024 / (lds + 512))));      // (a multiple of 8; per XCD >= 32 >= slabs)
#define GGAD_BRES(KSV) k_gemm_bres<KSV><<<dim3(G), dim3(256), lds, st>>>(A, B, C, (int)M, (int)N, (int)K, sam, sbk, sbn, ldc, bias, (int)relu, KP)
    switch (KSn) {
      case 8: GGAD_BRES(8); break;   case 9: GGAD_BRES(9); break;   case 10: GGAD_BRES(10); break; case 11: GGAD_BRES(11); break;
      case 12: GGAD_BRES(12); break; case 13: GGAD_BRES(13); break; case 14: GGAD_BRES(14); break; case 15: GGAD_BRES(15); break;
      case 16: GGAD_BRES(16); break; case 17: GGAD_BRES(17); break; case 18: GGAD_BRES(18); break; case 19: GGAD_BRES(19); break;
      default: GGAD_BRES(20); break;
    }
#undef GGAD_BRES
    GGAD_CHECK_LAUNCH("gemm_f32 (resident operand)");
    return GGAD_OK;
  }
  // weight gradients (op(A) = A^T with A: K x M row-major, B: K x N row-major, K long): row ranges + ordered reduction (mlp.hip)
  if (ws_elems > 0 && sam == 1 && sbn == 1 && ldc == N && !bias && !relu && ws_elems >= ggad_int_wgrad_tn_ws(K, M, N) && ggad_int_wgrad_tn_ws(K, M, N) > 0) {
    const int r = ggad_int_wgrad_tn(A, sak, B, sbk, (int)K, (int)M, (int)N, C, workspace, st);
    if (r < 0) { ggad_set_error(hipErrorLaunchFailure, "gemm_f32 (wgrad tn)"); return GGAD_E_LAUNCH; }
    if (r > 0) return GGAD_OK;
  }
  // small products: both operand panels of a 32 x 32 tile in LDS for the whole K (one round trip), GGAD_GEMM_SMALL=0 turns it off
  static const bool no_small = [] { const char *e = getenv("GGAD_GEMM_SMALL"); return e && e[0] == '0'; }();
  // (gate = the measured regime, scripts/gemm_small_time.py: the 127-843-row products of the outlier head; a tall M with K = 512 would be
  //  hundreds of 133-KB workgroups staging serially in front of the pipelined kernels -- ADVICE round 5)
  if (!no_small && ws_elems == 0 && K >= 1 && K <= 512 && M <= 1024 && (int64_t)((M + SM_T - 1) / SM_T) * ((N + SM_T - 1) / SM_T) <= 1024 &&
      (int64_t)M * N <= (int64_t)1024 * 512) {
    const int KP = sm_kp((K + 15) & ~15);
    const size_t lds = (size_t)2 * SM_T * KP * sizeof(float);
    // the opt-in for > 64 KB of dynamic LDS is recorded per DEVICE (0 unknown, 1 ready, -1 refused: the tiled kernels take the product)
    static std::mutex mu;
    static int state[64] = {};
    int dev = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (ok) {
      std::lock_guard<std::mutex> lock(mu);
      if (state[dev] == 0) {
        state[dev] = hipFuncSetAttribute((const void *)k_gemm_small, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SM_T * 520 * 4 + 1024) == hipSuccess ? 1 : -1;
        (void)hipGetLastError();
      }
      ok = state[dev] == 1;
    }
    if (ok) {
      k_gemm_small<<<dim3((N + SM_T - 1) / SM_T, (M + SM_T - 1) / SM_T), dim3(256), lds, st>>>(A, B, C, (int)M, (int)N, (int)K, sam, sak, sbk, sbn, ldc,
                                                                                                  bias, (int)relu, KP);
      GGAD_CHECK_LAUNCH("gemm_f32 (small)");
      return GGAD_OK;
    }
    (void)hipGetLastError();
  }
  // LDS-DMA kernel: every 16-byte chunk an operand tile is made of must be 16-byte aligned in memory and lie inside one row
  static const bool no_dma = [] { const char *e = getenv("GGAD_GEMM_DMA"); return e && e[0] == '0'; }();
  const bool a_ok = (((uintptr_t)A & 15) == 0) && (a_kfast ? (sam % 4 == 0 && K % 4 == 0) : (sak % 4 == 0 && M % 4 == 0));
  const bool b_ok = (((uintptr_t)B & 15) == 0) && (b_nfast ? (sbk % 4 == 0 && N % 4 == 0) : (sbn % 4 == 0 && K % 4 == 0));
  if (!no_dma && tm == 64 && a_ok && b_ok && K >= 32) {
    const int splits = ws_elems ? gemm_splits(M, N, K) : 1;
    int kps = K;
    if (splits > 1) { kps = (K + splits - 1) / splits; kps = (kps + BK - 1) / BK * BK; }
    float *out = splits > 1 ? workspace : C;
    const dim3 grid(gx, gy, splits);
#define GGAD_DMA_ARGS A, B, out, (int)M, (int)N, (int)K, (int)sam, (int)sak, (int)sbk, (int)sbn, splits > 1 ? (int64_t)N : ldc, \
                      splits > 1 ? (const float *)nullptr : bias, splits > 1 ? 0 : (int)relu, kps, (int64_t)M * N
    static const int nbuf = [] { const char *e = getenv("GGAD_GEMM_DMA_BUFS"); return e ? atoi(e) : 2; }();
    if (nbuf == 2) {
      if (a_kfast && b_nfast) k_gemm_dma<true, true, 2><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
      else if (a_kfast) k_gemm_dma<true, false, 2><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
      else if (b_nfast) k_gemm_dma<false, true, 2><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
      else k_gemm_dma<false, false, 2><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
    } else {
      if (a_kfast && b_nfast) k_gemm_dma<true, true, 3><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
      else if (a_kfast) k_gemm_dma<true, false, 3><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
      else if (b_nfast) k_gemm_dma<false, true, 3><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
      else k_gemm_dma<false, false, 3><<<grid, dim3(256), 0, st>>>(GGAD_DMA_ARGS);
    }
#undef GGAD_DMA_ARGS
    if (splits > 1) {
      const int64_t tot = (int64_t)M * N;
      k_splitk_reduce<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(workspace, splits, (int64_t)M * N, C, M, N, ldc, bias, relu);
    }
    GGAD_CHECK_LAUNCH("gemm_f32 (dma)");
    return GGAD_OK;
  }
  if (ws_elems == 0) {
    if (tm == 128)
      launch_gemm<128>(vec, a_kfast, b_nfast, dim3(gx, gy, 1), st, A, B, C, (int)M, (int)N, (int)K, (int)sam, (int)sak, (int)sbk, (int)sbn,
                       ldc, bias, (int)relu, K > 0 ? (int)K : 1, (int64_t)0);
    else
      launch_gemm<64>(vec, a_kfast, b_nfast, dim3(gx, gy, 1), st, A, B, C, (int)M, (int)N, (int)K, (int)sam, (int)sak, (int)sbk, (int)sbn,
                      ldc, bias, (int)relu, K > 0 ? (int)K : 1, (int64_t)0);
  } else {
    const int splits = gemm_splits(M, N, K);
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    if (tm == 128)
      launch_gemm<128>(vec, a_kfast, b_nfast, dim3(gx, gy, splits), st, A, B, workspace, (int)M, (int)N, (int)K, (int)sam, (int)sak,
                       (int)sbk, (int)sbn, (int64_t)N, (const float *)nullptr, 0, kps, (int64_t)M * N);
    else
      launch_gemm<64>(vec, a_kfast, b_nfast, dim3(gx, gy, splits), st, A, B, workspace, (int)M, (int)N, (int)K, (int)sam, (int)sak,
                      (int)sbk, (int)sbn, (int64_t)N, (const float *)nullptr, 0, kps, (int64_t)M * N);
    const int64_t tot = (int64_t)M * N;
    k_splitk_reduce<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(workspace, splits, (int64_t)M * N, C, M, N, ldc,
                                                                             bias, relu);
  }
  GGAD_CHECK_LAUNCH("gemm_f32");
  return GGAD_OK;
}


/* z = A W^T + bias and out = PReLU(z) in ONE launch (the first GCN layer on a cached aggregate, reference model.py:27-35), when the slab
 * kernel takes the shape (M >= 4096, K % 4 == 0 in 17 .. 32 / 49 .. 64 / 241 .. 320, N % 4 == 0, 16-byte aligned rows): returns GGAD_OK, or
 * GGAD_E_UNSUPPORTED without launching anything -- the caller then runs ggad_gemm_f32 + ggad_prelu_fwd_f32.  A: M x K (rows lda apart),
 * W: N x K (rows ldw apart), z / out: M x N (rows ldz / ldo apart). */
int ggad_linear_prelu_f32(const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, const float *prelu_a, int32_t M,
                          int32_t N, int32_t K, float *z, int64_t ldz, float *out, int64_t ldo, ggad_stream_t stream) {
  GGAD_REQUIRE(A && W && prelu_a && z && out && M >= 0 && N >= 1 && K >= 1 && lda >= K && ldw >= K && ldz >= N && ldo >= N);
  if (M == 0) return GGAD_OK;
  const int r = ggad_int_gemm_slab(A, W, z, (int)M, (int)N, (int)K, lda, 1, ldw, ldz, bias, 0, as_stream(stream), prelu_a, out, ldo);
  if (r < 0) { ggad_set_error(hipErrorLaunchFailure, "linear_prelu_f32 (slab)"); return GGAD_E_LAUNCH; }
  return r > 0 ? GGAD_OK : GGAD_E_UNSUPPORTED;
}

}  // extern "C"
