// What the f32 matrix cores of this chip sustain when nothing but MFMA is issued: every SIMD of every CU runs independent
// accumulator chains of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 for ~1 ms (clocks settle to the sustained value).
// The GEMM kernels of csrc/gemm.hip are priced against this figure beside the nominal 157 TF (256 CUs x 256 flop/clk x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/mfma_peak_probe.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k16(float *out, int iters, float a, float b) {
  f4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k32(float *out, int iters, float a, float b) {
  f16v acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[c][k] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][5];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <class F>
static double run(F launch, double flop_per_iter_per_wave, int waves, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return flop_per_iter_per_wave * waves * iters / (ms * 1e-3) / 1e12;
}

int main() {
  float *out;
  hipMalloc(&out, 1024);
  for (int wg_per_cu : {1, 2}) {
    const int blocks = 256 * wg_per_cu, waves = blocks * 4;
    const int iters = 20000 / wg_per_cu;
    printf("%d workgroups of 4 waves per CU:\n", wg_per_cu);
    printf("  16x16x4, 1 chain : %6.1f TF\n", run([&](int it) { k16<1><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 8.0 * 2048, waves, iters));
    printf("  16x16x4, 2 chains: %6.1f TF\n", run([&](int it) { k16<2><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 16.0 * 2048, waves, iters));
    printf("  16x16x4, 4 chains: %6.1f TF\n", run([&](int it) { k16<4><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 32.0 * 2048, waves, iters));
    printf("  32x32x2, 1 chain : %6.1f TF\n", run([&](int it) { k32<1><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 4096, waves, iters));
    printf("  32x32x2, 2 chains: %6.1f TF\n", run([&](int it) { k32<2><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 8.0 * 4096, waves, iters));
  }
  return 0;
}
