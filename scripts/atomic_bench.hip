// Micro-benchmark: device-scope atomic-add / gather rates vs footprint and clustering (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// each thread: M atomics at idx[t]*stride + j  (j < M consecutive words)
template <int M>
__global__ void __launch_bounds__(256) k_atomic(const unsigned* __restrict__ idx, long n, int* __restrict__ cnt, long stride) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long base = (long)idx[t] * stride;
#pragma unroll
  for (int j = 0; j < M; ++j) atomicAdd(&cnt[base + j], 1);
}
template <int M>
__global__ void __launch_bounds__(256) k_read(const unsigned* __restrict__ idx, long n, const int* __restrict__ cnt, long stride, int* out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long base = (long)idx[t] * stride;
  int s = 0;
#pragma unroll
  for (int j = 0; j < M; ++j) s += cnt[base + j];
  if (s == 0x7fffffff) out[0] = s;
}

int main() {
  const long P = 1L << 26;   // 67M threads
  std::vector<unsigned> h(P);
  std::mt19937_64 r(1);
  unsigned *idx; int *cnt, *out;
  const long CAP = 600L * 1000 * 1000;   // 2.4 GB of counters
  CK(hipMalloc(&idx, P * 4)); CK(hipMalloc(&cnt, CAP * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemset(cnt, 0, CAP * 4));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct Cfg { const char* name; long range; long stride; int m; };
  // range = number of distinct "rows"; stride words per row
  Cfg cfgs[] = {{"random 1 atomic, 2.2 GB footprint", 550000000L, 1, 1}, {"random 1 atomic, 256 MB", 64000000L, 1, 1},
                {"random 1 atomic, 32 MB", 8000000L, 1, 1}, {"random 1 atomic, 3 MB", 750000L, 1, 1},
                {"clustered 8 atomics in a 300 B row, 1.1 GB", 3700000L, 75, 8}, {"clustered 16 in a 300 B row", 3700000L, 75, 16}};
  for (auto& c : cfgs) {
    for (long i = 0; i < P; ++i) h[i] = (unsigned)(r() % c.range);
    CK(hipMemcpy(idx, h.data(), P * 4, hipMemcpyHostToDevice));
    const long n = P / c.m;
    const int blocks = (int)((n + 255) / 256);
    float msa = 0, msr = 0;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (c.m == 1) k_atomic<1><<<blocks, 256>>>(idx, n, cnt, c.stride);
      if (c.m == 8) k_atomic<8><<<blocks, 256>>>(idx, n, cnt, c.stride);
      if (c.m == 16) k_atomic<16><<<blocks, 256>>>(idx, n, cnt, c.stride);
      (void)hipEventRecord(e1); CK(hipEventSynchronize(e1)); (void)hipEventElapsedTime(&msa, e0, e1);
      (void)hipEventRecord(e0);
      if (c.m == 1) k_read<1><<<blocks, 256>>>(idx, n, cnt, c.stride, out);
      if (c.m == 8) k_read<8><<<blocks, 256>>>(idx, n, cnt, c.stride, out);
      if (c.m == 16) k_read<16><<<blocks, 256>>>(idx, n, cnt, c.stride, out);
      (void)hipEventRecord(e1); CK(hipEventSynchronize(e1)); (void)hipEventElapsedTime(&msr, e0, e1);
    }
    printf("%-46s atomics %7.1f G/s   reads %7.1f G/s\n", c.name, (double)n * c.m / msa / 1e6, (double)n * c.m / msr / 1e6);
  }
  return 0;
}
