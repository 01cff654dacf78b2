// Would a source-range-partitioned 2-hop gather pay?    Upper bound on this memory system:
// 64 M neighbour ids drawn in proportion to a power-law degree sequence (alpha 2.1, max degree 2,000, ids unrelated to
// degree -- the bench graph), 68-byte rows of a 3.7 M x 17 table, every wave gathers 64 rows per step like k_gather2_items.
//   (a) every workgroup draws from the whole id space                       (today: each XCD's L2 sees all hot rows)
//   (b) workgroup b only draws ids of range b % 8 (eighths of the id space)  (each L2 keeps 1/8 of the hot rows)
//   (c) as (b) but only 8 ids of every 64 are taken per load step           (what a low-degree owner split 8 ways does)
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_range_bench.hip -o /tmp/grb && /tmp/grb
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template <int PER>    // ids used of every 64 (64 = full block, 8 = an eighth)
__global__ void __launch_bounds__(256) k(const int *__restrict__ idx, const long *__restrict__ rng_ptr, const float *__restrict__ tab,
                                         float *__restrict__ out, int ranged) {
  extern __shared__ int occupancy_limiter[];      // dynamic LDS only bounds the workgroups per CU
  const int lane = threadIdx.x & 63;
  const int r = ranged ? blockIdx.x % 8 : 0;
  const long lo = rng_ptr[ranged ? r : 0], hi = rng_ptr[ranged ? r + 1 : 8];
  const long wv = ((long)(ranged ? blockIdx.x / 8 : blockIdx.x) * 4 + (threadIdx.x >> 6));
  const long nw = (long)(ranged ? gridDim.x / 8 : gridDim.x) * 4;
  const int g = lane / 17, f = lane - g * 17;
  const bool act = g < 3;
  float acc = 0.f;
  for (long base = lo + wv * 64; base + 64 <= hi; base += nw * 64) {
    const int kid = idx[base + lane];
    float x[22];
#pragma unroll
    for (int t = 0; t < (PER + 2) / 3; ++t) {
      const int kk = __shfl(kid, (t * 3 + (act ? g : 2)) & 63);
      x[t] = tab[(long)kk * 17 + f];
    }
#pragma unroll
    for (int t = 0; t < (PER + 2) / 3; ++t) acc += act ? x[t] : 0.f;
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const long N = 3700550, P = 64L * 1000 * 1000;
  std::mt19937_64 r(1);
  // power-law degrees, alias-free sampling through a cumulative table
  std::vector<double> cum(N);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  double s = 0;
  for (long i = 0; i < N; ++i) { double d = std::min(2000.0, std::pow(1.0 - U(r), -1.0 / 1.1)); s += d; cum[i] = s; }
  std::vector<int> h(P);
  for (long i = 0; i < P; ++i) h[i] = (int)(std::lower_bound(cum.begin(), cum.end(), U(r) * s) - cum.begin());
  std::vector<int> hs(h);
  std::vector<long> ptr(9, 0);
  { // bucket by eighth of the id space, random order inside a bucket
    std::vector<std::vector<int>> b(8);
    for (long i = 0; i < P; ++i) b[std::min<long>(7, (long)h[i] * 8 / N)].push_back(h[i]);
    long o = 0;
    for (int q = 0; q < 8; ++q) { ptr[q] = o; std::copy(b[q].begin(), b[q].end(), hs.begin() + o); o += (long)b[q].size(); }
    ptr[8] = o;
  }
  int *idx, *idxs; long *dptr; float *tab, *out;
  CK(hipMalloc(&idx, P * 4)); CK(hipMalloc(&idxs, P * 4)); CK(hipMalloc(&dptr, 9 * 8)); CK(hipMalloc(&tab, N * 17 * 4));
  const int blocks = 256 * 8 * 4;
  CK(hipMalloc(&out, (long)blocks * 256 * 4));
  CK(hipMemcpy(idx, h.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(idxs, hs.data(), P * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dptr, ptr.data(), 72, hipMemcpyHostToDevice)); CK(hipMemset(tab, 0, N * 17 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"(a) whole id space per workgroup, 64 rows per step", "(b) one eighth of the id space per XCD, 64 rows per step",
                         "(a8) whole id space, 8 rows per step", "(c) one eighth per XCD, 8 rows per step"};
  for (int wpc : {32, 16, 12, 8}) {                 // waves per CU (k_gather2_items: 165 VGPRs = 12)
  const int lds = wpc == 32 ? 0 : (160 * 1024) / (wpc / 4) - 512;
  CK(hipFuncSetAttribute((const void *)k<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("-- at most %d waves per CU\n", wpc);
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (v == 0) k<64><<<blocks, 256, lds>>>(idx, dptr, tab, out, 0);
      if (v == 1) k<64><<<blocks, 256, lds>>>(idxs, dptr, tab, out, 1);
      if (v == 2) k<8><<<blocks, 256, lds>>>(idx, dptr, tab, out, 0);
      if (v == 3) k<8><<<blocks, 256, lds>>>(idxs, dptr, tab, out, 1);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) best = std::min(best, ms);
    }
    const double rows = (v < 2) ? (double)P : (double)P / 64 * 9;      // (8 + 2) / 3 = 3 load steps x 3 rows
    printf("%-62s %.3f ms  %.1f G rows/s  %.0f GB/s at 72 B/row\n", names[v], best, rows / best / 1e6, 72.0 * rows / best / 1e6);
  }
  }
  return 0;
}
