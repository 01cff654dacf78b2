// Micro-benchmark (round 5): random gather of a node's 17 features when the table is [n][32] floats (one 128-byte line per row, what the
// 2-hop gather reads today: 474 MB at DGraph-Fin size) against [n][16] floats (ONE 64-byte sector per row: 237 MB, below the 256 MiB of
// the Infinity Cache) + the 17th feature in an array of its own (15 MB), gathered one neighbour per lane.
// hipcc --offload-arch=gfx950 -O3 scripts/gather_row64_bench.hip -o /tmp/gr64 && /tmp/gr64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// one wave handles 64 consecutive pairs.  V 0: stride 32, 32 lanes per row (2 rows per load instruction);
// V 1: stride 16, 16 lanes per row (4 rows per instruction) + f17[k] per lane;  V 2: stride 16 without the 17th feature
template <int V>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, long npairs, const float* __restrict__ tab,
                                         const float* __restrict__ f17, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long base = wave * 64;
  if (base >= npairs) return;
  constexpr int STR = V == 0 ? 32 : 16;
  constexpr int RPI = 64 / STR;
  const int g = lane / STR, f = lane - g * STR;
  const int k = idx[base + lane];
  float acc = 0.f, a17 = 0.f;
  if (V == 1) a17 = f17[k];
  float x[64 / RPI];
#pragma unroll
  for (int t = 0; t < 64 / RPI; ++t) {
    const int kk = __shfl(k, t * RPI + g);
    x[t] = (V != 0 || f < 17) ? tab[(long)kk * STR + f] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 64 / RPI; ++t) acc += x[t];
  out[wave * 64 + lane] = acc + a17;
}

int main(int argc, char** argv) {
  const long N = 3700550, P = 64L * 1000 * 1000;   // 64M pairs
  const int skew = argc > 1 ? atoi(argv[1]) : 0;   // 1: ids biased to a hot set (a tenth of the nodes takes half of the pairs)
  std::vector<int> h(P);
  std::mt19937_64 r(1);
  for (long i = 0; i < P; ++i) { const unsigned long long v = r(); h[i] = (skew && (v & 1)) ? (int)((v >> 1) % (N / 10)) * 10 % N : (int)((v >> 1) % N); }
  int *idx; float *t16, *t32, *f17, *out;
  CK(hipMalloc(&idx, P * 4)); CK(hipMalloc(&f17, N * 4)); CK(hipMalloc(&t16, N * 16 * 4)); CK(hipMalloc(&t32, N * 32 * 4));
  CK(hipMalloc(&out, P * 4));
  CK(hipMemcpy(idx, h.data(), P * 4, hipMemcpyHostToDevice));
  CK(hipMemset(f17, 0, N * 4)); CK(hipMemset(t16, 0, N * 16 * 4)); CK(hipMemset(t32, 0, N * 32 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long pairs : {4L * 1000 * 1000, 64L * 1000 * 1000}) {
    const int blocks = (int)((pairs / 64 + 3) / 4);
    for (int v = 0; v < 3; ++v) {
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (v == 0) k<0><<<blocks, 256>>>(idx, pairs, t32, f17, out);
        if (v == 1) k<1><<<blocks, 256>>>(idx, pairs, t16, f17, out);
        if (v == 2) k<2><<<blocks, 256>>>(idx, pairs, t16, f17, out);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 3) printf("%s ids, %3ld M rows, %s: %8.3f ms  %6.1f G rows/s\n", skew ? "skewed " : "uniform", pairs / 1000000,
                             v == 0 ? "[n][32] 128-byte rows           " : (v == 1 ? "[n][16] 64-byte rows + f17[n]   " : "[n][16] 64-byte rows alone      "),
                             ms, pairs / ms / 1e6);
      }
    }
  }
  return 0;
}
