// Probe of the VGPR index mode (s_set_gpr_idx_on) on gfx950: does it apply to VOP3P (v_pk_add_f32), and what do the indexed additions
// and the mode toggles cost?  hipcc --offload-arch=gfx950 -O3 scripts/gpr_idx_probe.hip -o scripts/gpr_idx_probe && scripts/gpr_idx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v8f __attribute__((ext_vector_type(8)));

// out[0..7] = accumulators v[64:71] after adding (1,2) to the pair selected by idx (expected: pair idx / 2 holds (1,2))
__global__ void k_func(float *out, int idx, int mode) {
  v8f a;
  int i = __builtin_amdgcn_readfirstlane(idx);
  asm volatile(
      "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"
      "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 2.0\n"
      "s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n"
      "s_set_gpr_idx_on %1, 0xa\n v_pk_add_f32 v[64:65], v[20:21], v[64:65]\n s_set_gpr_idx_off\n s_branch 2f\n"
      "1:\n s_set_gpr_idx_on %1, 0x9\n v_pk_add_f32 v[64:65], v[64:65], v[20:21]\n s_set_gpr_idx_off\n"
      "2:\n"
      : "={v[64:71]}"(a) : "s"(i), "s"(mode) : "v20", "v21", "scc");
  if (threadIdx.x == 0) for (int j = 0; j < 8; ++j) out[j] = a[j];
}

// VARIANT 0: 16 plain v_add_f32 (no index mode); 1: on + 16 indexed v_add_f32 + off; 2: on + 8 indexed v_pk_add_f32 + off;
// 3: 8 plain v_pk_add_f32; 4: on/off only; 5: 16 indexed v_add_f32 with the mode left on (toggled once outside the loop)
template <int VARIANT>
__global__ void __launch_bounds__(1024) k_time(long long *cycles, float *sink, int iters) {
  v8f a;
  long long t0 = clock64();
  int n = __builtin_amdgcn_readfirstlane(iters);
  asm volatile(
      "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"
      "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 2.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 2.0\n s_mov_b32 s40, 4\n"
      ".if %c2 == 5\n s_set_gpr_idx_on s40, 0xa\n .endif\n"
      "1:\n"
      ".if %c2 == 0\n"
      ".rept 4\n v_add_f32_e32 v64, v20, v64\n v_add_f32_e32 v65, v21, v65\n v_add_f32_e32 v66, v22, v66\n v_add_f32_e32 v67, v23, v67\n .endr\n"
      ".endif\n"
      ".if %c2 == 1\n s_set_gpr_idx_on s40, 0xa\n"
      ".rept 4\n v_add_f32_e32 v64, v20, v64\n v_add_f32_e32 v65, v21, v65\n v_add_f32_e32 v66, v22, v66\n v_add_f32_e32 v67, v23, v67\n .endr\n"
      "s_set_gpr_idx_off\n .endif\n"
      ".if %c2 == 2\n s_set_gpr_idx_on s40, 0xa\n"
      ".rept 4\n v_pk_add_f32 v[64:65], v[20:21], v[64:65]\n v_pk_add_f32 v[66:67], v[22:23], v[66:67]\n .endr\n"
      "s_set_gpr_idx_off\n .endif\n"
      ".if %c2 == 3\n"
      ".rept 4\n v_pk_add_f32 v[64:65], v[20:21], v[64:65]\n v_pk_add_f32 v[66:67], v[22:23], v[66:67]\n .endr\n"
      ".endif\n"
      ".if %c2 == 4\n s_set_gpr_idx_on s40, 0xa\n s_set_gpr_idx_off\n .endif\n"
      ".if %c2 == 5\n"
      ".rept 4\n v_add_f32_e32 v64, v20, v64\n v_add_f32_e32 v65, v21, v65\n v_add_f32_e32 v66, v22, v66\n v_add_f32_e32 v67, v23, v67\n .endr\n"
      ".endif\n"
      "s_sub_u32 %1, %1, 1\n s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1b\n"
      ".if %c2 == 5\n s_set_gpr_idx_off\n .endif\n"
      : "={v[64:71]}"(a), "+s"(n) : "n"(VARIANT) : "v20", "v21", "v22", "v23", "s40", "scc");
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a[0] == -1.f) sink[0] = a[1] + a[4];
}

template <int V> void run(const char *what, int threads) {
  long long *d; float *s;
  hipMalloc(&d, 16 * sizeof(long long)); hipMalloc(&s, 16);
  const int iters = 20000;
  k_time<V><<<1, threads>>>(d, s, iters);
  k_time<V><<<1, threads>>>(d, s, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(16);
  hipMemcpy(h.data(), d, 16 * sizeof(long long), hipMemcpyDeviceToHost);
  printf("%-44s %2d waves: %.1f clock64 ticks per iteration (wave 0)\n", what, threads / 64, (double)h[0] / iters);
  hipFree(d); hipFree(s);
}

int main() {
  float *o; hipMalloc(&o, 64);
  for (int mode = 0; mode < 2; ++mode)
    for (int idx = 0; idx <= 6; idx += 2) {
      k_func<<<1, 64>>>(o, idx, mode);
      float h[8]; hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
      printf("pk_add idx %d (%s): %g %g | %g %g | %g %g | %g %g\n", idx, mode ? "src1+dst 0xa" : "src0+dst 0x9", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
  for (int th : {64, 256, 1024}) {
    run<0>("16 plain v_add_f32", th);
    run<1>("on + 16 indexed v_add_f32 + off", th);
    run<5>("16 indexed v_add_f32 (mode left on)", th);
    run<2>("on + 8 indexed v_pk_add_f32 + off", th);
    run<3>("8 plain v_pk_add_f32", th);
    run<4>("on + off only", th);
  }
  return 0;
}
