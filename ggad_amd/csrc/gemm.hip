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
#include "common.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32;      // BM: the large row tile; BM / 2 is used when the grid would not fill the chip
constexpr int LDA_S = BM + 1, LDB_S = BN + 1;       // +1 pad: the transposing LDS store is conflict-free
constexpr int A_PER_T = BM * BK / 256;              // 16 elements of the A tile per thread
constexpr int B_PER_T = BN * BK / 256;              // 8 elements of the B tile per thread

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Workgroup = 4 waves = a 128 x 64 tile of C; wave (wr, wc) owns rows [64 wr, 64 wr + 64) x cols [32 wc, 32 wc + 32):
// two 32x32 MFMA accumulators.  K is stepped by 32; the next K-tile is fetched from global memory into registers
// while the current one is consumed from LDS (global latency hidden behind 32 MFMAs = 2048 cycles per wave).
// Register diet (the first version needed 186 VGPRs = 2 waves per SIMD and left the MFMA pipe idle 48 % of the time,
// rocprofv3 SQ_WAIT_INST_ANY): the operand layouts are template parameters, every thread keeps ONE 32-bit offset per
// element relative to a wave-uniform tile pointer (SGPR base + VGPR offset addressing), the row / column bounds are
// folded into those offsets once (out-of-range elements point at element 0 and are zeroed by a mask bit), and only
// the last K-tile checks k.
template <bool A_KFAST, bool B_NFAST, int TM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) k_gemm_f32(const float *__restrict__ A, const float *__restrict__ B,
                                                  float *__restrict__ C, int M, int N, int K, int sam, int sak,
                                                  int sbk, int sbn, int64_t ldc, const float *__restrict__ bias,
                                                  int relu, int k_per_split, int64_t c_split_stride) {
  constexpr int TA_PER_T = TM * BK / 256;      // A elements per thread
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
  // element p of this thread inside a tile: A (m, k), B (n, k); unit stride along the fastest index of each operand
  int oa[TA_PER_T], ob[B_PER_T];          // offsets relative to the tile pointers
  unsigned amask = 0, bmask = 0;         // bit p: row (column) inside the matrix
  int ka0, kb0;                          // k of element 0 of this thread; element p adds a constant step
  constexpr int KA_STEP = A_KFAST ? 0 : 256 / TM, KB_STEP = B_NFAST ? 4 : 0;
  if (A_KFAST) ka0 = tid & 31; else ka0 = tid / TM;
  if (B_NFAST) kb0 = tid >> 6; else kb0 = tid & 31;
#pragma unroll
  for (int p = 0; p < TA_PER_T; ++p) {
    const int m = A_KFAST ? (tid >> 5) + 8 * p : (tid % TM);
    const int k = ka0 + KA_STEP * p;
    const bool in = m0 + m < M;
    oa[p] = in ? m * sam + k * sak : 0;
    amask |= (in ? 1u : 0u) << p;
  }
#pragma unroll
  for (int p = 0; p < B_PER_T; ++p) {
    const int n = B_NFAST ? (tid & 63) : (tid >> 5) + 8 * p;
    const int k = kb0 + KB_STEP * p;
    const bool in = n0 + n < N;
    ob[p] = in ? k * sbk + n * sbn : 0;
    bmask |= (in ? 1u : 0u) << p;
  }
  const float *At = A + (int64_t)m0 * sam + (int64_t)k_begin * sak;     // wave-uniform tile pointers
  const float *Bt = B + (int64_t)n0 * sbn + (int64_t)k_begin * sbk;
  float ra[TA_PER_T], rb[B_PER_T];
  unsigned oka = 0, okb = 0;             // validity of the elements fetched last
  auto fetch = [&](int k0) {
    const int klim = k_end - k0;         // elements with k >= klim lie past the end of this split (last K-tile only)
    oka = 0; okb = 0;
#pragma unroll
    for (int p = 0; p < TA_PER_T; ++p) {  // branch-free: every load is issued before the first one is consumed
      const unsigned ok = ((amask >> p) & 1u) & (unsigned)(ka0 + KA_STEP * p < klim);
      oka |= ok << p;
      ra[p] = At[ok ? oa[p] : 0];
    }
#pragma unroll
    for (int p = 0; p < B_PER_T; ++p) {
      const unsigned ok = ((bmask >> p) & 1u) & (unsigned)(kb0 + KB_STEP * p < klim);
      okb |= ok << p;
      rb[p] = Bt[ok ? ob[p] : 0];
    }
    At += (int64_t)BK * sak;             // (the masks are applied in stash(): nothing here waits for the loads)
    Bt += (int64_t)BK * sbk;
  };
  auto stash = [&]() {
#pragma unroll
    for (int p = 0; p < TA_PER_T; ++p) {
      const int m = A_KFAST ? (tid >> 5) + 8 * p : (tid % TM);
      As[ka0 + KA_STEP * p][m] = ((oka >> p) & 1u) ? ra[p] : 0.0f;
    }
#pragma unroll
    for (int p = 0; p < B_PER_T; ++p) {
      const int n = B_NFAST ? (tid & 63) : (tid >> 5) + 8 * p;
      Bs[kb0 + KB_STEP * p][n] = ((okb >> p) & 1u) ? rb[p] : 0.0f;
    }
  };
  if (k_begin < k_end) {
    fetch(k_begin);
    stash();
    __syncthreads();
  }
  const int i = lane & 31, kk = lane >> 5;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const bool more = (k0 + BK) < k_end;
    if (more) fetch(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      const float b = Bs[ks + kk][wc * 32 + i];
      const float a0 = As[ks + kk][wr * 32 * NACC + i];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      if constexpr (NACC == 2) {
        const float a1 = As[ks + kk][wr * 64 + 32 + i];
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
      }
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

__global__ void __launch_bounds__(256) k_splitk_reduce(const float *__restrict__ ws, int splits, int64_t split_stride,
                                                       float *__restrict__ C, int M, int N, int64_t ldc,
                                                       const float *__restrict__ bias, int relu) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
  float s = 0.0f;
  for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * split_stride + (int64_t)m * N + n];   // fixed order
  if (bias != nullptr) s += bias[n];
  if (relu) s = fmaxf(s, 0.0f);
  C[(int64_t)m * ldc + n] = s;
}

template <int TM, typename... Args>
static void launch_gemm(bool a_kfast, bool b_nfast, dim3 grid, hipStream_t st, Args... args) {
  if (a_kfast && b_nfast) k_gemm_f32<true, true, TM><<<grid, dim3(256), 0, st>>>(args...);
  else if (a_kfast) k_gemm_f32<true, false, TM><<<grid, dim3(256), 0, st>>>(args...);
  else if (b_nfast) k_gemm_f32<false, true, TM><<<grid, dim3(256), 0, st>>>(args...);
  else k_gemm_f32<false, false, TM><<<grid, dim3(256), 0, st>>>(args...);
}

// Row tile: 128 when that still gives >= 3 workgroups per CU (they hide each other's global-load latency: the kernel has one
// K-tile of look-ahead), else 64.
static int pick_tm(int M, int N) {
  const int64_t tiles128 = (int64_t)((M + 127) / 128) * ((N + BN - 1) / BN);
  return tiles128 >= 3 * 256 ? 128 : 64;
}
}  // namespace

extern "C" {

int64_t ggad_gemm_workspace_elems(int32_t M, int32_t N, int32_t K) {
  const int tm = pick_tm(M, N);
  const int64_t tiles = (int64_t)((M + tm - 1) / tm) * ((N + BN - 1) / BN);
  if (K < 2048 || tiles >= 256) return 0;
  int splits = (int)((512 + tiles - 1) / tiles);
  const int max_splits = (K + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits > 64) splits = 64;
  return splits <= 1 ? 0 : (int64_t)splits * M * N;
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
  const int64_t ws_elems = workspace ? ggad_gemm_workspace_elems(M, N, K) : 0;
  if (ws_elems == 0) {
    if (tm == 128)
      launch_gemm<128>(a_kfast, b_nfast, dim3(gx, gy, 1), st, A, B, C, (int)M, (int)N, (int)K, (int)sam, (int)sak, (int)sbk, (int)sbn,
                       ldc, bias, (int)relu, K > 0 ? (int)K : 1, (int64_t)0);
    else
      launch_gemm<64>(a_kfast, b_nfast, dim3(gx, gy, 1), st, A, B, C, (int)M, (int)N, (int)K, (int)sam, (int)sak, (int)sbk, (int)sbn,
                      ldc, bias, (int)relu, K > 0 ? (int)K : 1, (int64_t)0);
  } else {
    const int splits = (int)(ws_elems / ((int64_t)M * N));
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    if (tm == 128)
      launch_gemm<128>(a_kfast, b_nfast, dim3(gx, gy, splits), st, A, B, workspace, (int)M, (int)N, (int)K, (int)sam, (int)sak,
                       (int)sbk, (int)sbn, (int64_t)N, (const float *)nullptr, 0, kps, (int64_t)M * N);
    else
      launch_gemm<64>(a_kfast, b_nfast, dim3(gx, gy, splits), st, A, B, workspace, (int)M, (int)N, (int)K, (int)sam, (int)sak,
                      (int)sbk, (int)sbn, (int64_t)N, (const float *)nullptr, 0, kps, (int64_t)M * N);
    const int64_t tot = (int64_t)M * N;
    k_splitk_reduce<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(workspace, splits, (int64_t)M * N, C, M, N, ldc,
                                                                             bias, relu);
  }
  GGAD_CHECK_LAUNCH("gemm_f32");
  return GGAD_OK;
}

}  // extern "C"
