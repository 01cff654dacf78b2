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

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int LDA_S = BM + 1, LDB_S = BN + 1;       // +1 pad: the transposing LDS store is conflict-free
constexpr int A_PER_T = BM * BK / 256;              // 16 elements of the A tile per thread
constexpr int B_PER_T = BN * BK / 256;              // 8 elements of the B tile per thread

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Workgroup = 4 waves = a 128 x 64 tile of C; wave (wr, wc) owns rows [64 wr, 64 wr + 64) x cols [32 wc, 32 wc + 32):
// two 32x32 MFMA accumulators.  K is stepped by 32; the next K-tile is fetched from global memory into registers
// while the current one is consumed from LDS (global latency hidden behind 32 MFMAs = 2048 cycles per wave).
__global__ void __launch_bounds__(256) k_gemm_f32(const float *__restrict__ A, const float *__restrict__ B,
                                                  float *__restrict__ C, int M, int N, int K, int64_t sam, int64_t sak,
                                                  int64_t sbk, int64_t sbn, int64_t ldc, const float *__restrict__ bias,
                                                  int relu, int k_per_split, int64_t c_split_stride) {
  __shared__ float As[BK][LDA_S];
  __shared__ float Bs[BK][LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  floatx16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  floatx16 acc1 = acc0;
  // thread -> (row, k) mapping of the global loads: unit stride along the fastest index of each operand
  const bool a_kfast = (sak == 1);
  const bool b_nfast = (sbn == 1);
  float ra[A_PER_T], rb[B_PER_T];
  auto a_pos = [&](int p, int &m, int &k) {
    if (a_kfast) { k = tid & 31; m = (tid >> 5) + 8 * p; } else { m = tid & 127; k = (tid >> 7) + 2 * p; }
  };
  auto b_pos = [&](int p, int &n, int &k) {
    if (b_nfast) { n = tid & 63; k = (tid >> 6) + 4 * p; } else { k = tid & 31; n = (tid >> 5) + 8 * p; }
  };
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < A_PER_T; ++p) {
      int m, k; a_pos(p, m, k);
      const int gm = m0 + m, gk = k0 + k;
      ra[p] = (gm < M && gk < k_end) ? A[(int64_t)gm * sam + (int64_t)gk * sak] : 0.0f;
    }
#pragma unroll
    for (int p = 0; p < B_PER_T; ++p) {
      int n, k; b_pos(p, n, k);
      const int gn = n0 + n, gk = k0 + k;
      rb[p] = (gn < N && gk < k_end) ? B[(int64_t)gk * sbk + (int64_t)gn * sbn] : 0.0f;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int p = 0; p < A_PER_T; ++p) { int m, k; a_pos(p, m, k); As[k][m] = ra[p]; }
#pragma unroll
    for (int p = 0; p < B_PER_T; ++p) { int n, k; b_pos(p, n, k); Bs[k][n] = rb[p]; }
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
      const float a0 = As[ks + kk][wr * 64 + i];
      const float a1 = As[ks + kk][wr * 64 + 32 + i];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
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
      const int row = m0 + wr * 64 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float v = acc0[r] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        Cout[(int64_t)row * ldc + col] = v;
      }
      if (row + 32 < M) {
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

}  // namespace

extern "C" {

int64_t ggad_gemm_workspace_elems(int32_t M, int32_t N, int32_t K) {
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
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
  if (M == 0 || N == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
  const int64_t ws_elems = workspace ? ggad_gemm_workspace_elems(M, N, K) : 0;
  if (ws_elems == 0) {
    k_gemm_f32<<<dim3(gx, gy, 1), dim3(256), 0, st>>>(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, bias, relu, K > 0 ? K : 1, 0);
  } else {
    const int splits = (int)(ws_elems / ((int64_t)M * N));
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    k_gemm_f32<<<dim3(gx, gy, splits), dim3(256), 0, st>>>(A, B, workspace, M, N, K, sam, sak, sbk, sbn, N, nullptr, 0, kps,
                                                          (int64_t)M * N);
    const int64_t tot = (int64_t)M * N;
    k_splitk_reduce<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(workspace, splits, (int64_t)M * N, C, M, N, ldc,
                                                                             bias, relu);
  }
  GGAD_CHECK_LAUNCH("gemm_f32");
  return GGAD_OK;
}

}  // extern "C"
