// What the f32 matrix cores sustain with REAL operand data: scripts/mfma_peak_probe.hip multiplies the constants 1 and 2 (nothing toggles,
// the chip holds its peak clock); here every lane holds different random operands, rotated every MFMA, like the projections of the
// full-graph path.  Prints TF/s and the shader clock the run implies (s_memtime cycles of wave 0 / wall time of the launch).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rand scripts/mfma_peak_random_probe.hip && /tmp/mfma_rand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool RANDOM>
__global__ void __launch_bounds__(256) k16(const float *__restrict__ in, float *out, unsigned long long *clk, int iters) {
  f4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    a[r] = RANDOM ? in[(blockIdx.x * 256 + threadIdx.x) * 16 + r] : 1.0f;
    b[r] = RANDOM ? in[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + r] : 2.0f;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + c) & 7], acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.f) out[threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
  const int blocks = 512, waves = blocks * 4;
  float *in, *out;
  unsigned long long *clk;
  std::vector<float> h((size_t)blocks * 256 * 16);
  srand(1);
  for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, 1024); hipMalloc(&clk, 8);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char *tag, int chains, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(in, out, clk, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(in, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double tf = 8.0 * chains * 2048.0 * waves * iters / (ms * 1e-3) / 1e12;
    printf("  %-28s %6.1f TF   %.3f ms   s_memtime ticks of wave 0 / wall = %.0f MHz\n", tag, tf, ms, (double)c / (ms * 1e3));
  };
  printf("2 workgroups of 4 waves per CU, v_mfma_f32_16x16x4_f32, 4 chains per wave:\n");
  for (int rep = 0; rep < 2; ++rep) {
    run(k16<4, false>, "constant operands (1, 2)", 4, 10000);
    run(k16<4, true>, "random operands per lane", 4, 10000);
  }
  printf("short launches (~20 us, the size of a Reddit projection):\n");
  run(k16<4, false>, "constant, 120 iterations", 4, 120);
  run(k16<4, true>, "random, 120 iterations", 4, 120);
  return 0;
}
