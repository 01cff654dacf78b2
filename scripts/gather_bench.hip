// Micro-benchmark: random gather of 17-float rows, layouts/counter placement variants (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// one wave handles 64 consecutive pairs; V: 0 = stride17 + separate counter, 1 = stride17 no counter,
// 2 = stride32 counter in row, 3 = stride32 no counter
template <int V>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, long npairs, const float* __restrict__ tab,
                                         const int* __restrict__ cnt, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long base = wave * 64;
  if (base >= npairs) return;
  constexpr int STR = (V >= 2) ? 32 : 17;
  constexpr int LPR = (V == 2) ? 18 : 17;       // lanes per row
  constexpr int RPI = 64 / LPR;                 // rows per instruction (3)
  const int g = lane / LPR, f = lane - g * LPR;
  const bool act = g < RPI;
  const int k = idx[base + lane];
  float w = 1.0f;
  if (V == 0) w = 1.0f / sqrtf((float)(cnt[k] + 1));
  float acc = 0.f;
  for (int t = 0; t < (64 + RPI - 1) / RPI; t += 1) {
    const int src = t * RPI + g;
    const int kk = __shfl(k, src & 63);
    float ws = __shfl(w, src & 63);
    float x = 0.f;
    if (act && src < 64) x = tab[(long)kk * STR + f];
    if (V == 2) {   // lane 17 of the group holds the counter word
      const float c = __shfl(x, g * LPR + 17);
      ws = 1.0f / sqrtf(__float_as_int(c) + 1.0f);
    }
    acc = fmaf((src < 64) ? ws : 0.f, x, acc);
  }
  if (act && f < 17) out[wave * 64 + lane] = acc;
}

int main() {
  const long N = 3700550, P = 64L * 1000 * 1000;   // 64M pairs
  std::vector<int> h(P);
  std::mt19937_64 r(1);
  for (long i = 0; i < P; ++i) h[i] = (int)(r() % N);
  int *idx, *cnt; float *t17, *t32, *out;
  CK(hipMalloc(&idx, P * 4)); CK(hipMalloc(&cnt, N * 4)); CK(hipMalloc(&t17, N * 17 * 4)); CK(hipMalloc(&t32, N * 32 * 4));
  CK(hipMalloc(&out, P * 4));
  CK(hipMemcpy(idx, h.data(), P * 4, hipMemcpyHostToDevice));
  CK(hipMemset(cnt, 0, N * 4)); CK(hipMemset(t17, 0, N * 17 * 4)); CK(hipMemset(t32, 0, N * 32 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = (int)((P / 64 + 3) / 4);
  for (int v = 0; v < 4; ++v) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (v == 0) k<0><<<blocks, 256>>>(idx, P, t17, cnt, out);
      if (v == 1) k<1><<<blocks, 256>>>(idx, P, t17, cnt, out);
      if (v == 2) k<2><<<blocks, 256>>>(idx, P, t32, cnt, out);
      if (v == 3) k<3><<<blocks, 256>>>(idx, P, t32, cnt, out);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("variant %d: %.3f ms  %.1f Gpairs/s  alg %.0f GB/s (72 B/pair)\n", v, ms, P / ms / 1e6, 72.0 * P / ms / 1e6);
    }
  }
  return 0;
}
