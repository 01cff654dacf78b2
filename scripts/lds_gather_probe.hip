// What does a ds_read_b128 cost when its 8-lane groups read 8 different 128-byte LDS rows (the access pattern of k_spmm_panel /
// k_spmm_ring), against the linear pattern the 256 B/clk figure is quoted for?  One workgroup of `waves` waves per CU, every wave keeps
// 8 reads in flight; time per wave-instruction per CU from HIP events.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_gather_probe.hip -o scripts/lds_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int LDS_BYTES = 158976;

// addr[lane][8]: byte addresses of the 8 reads of an iteration (read from global once)
__global__ void __launch_bounds__(1024) k_probe(const uint32_t *__restrict__ addr, float *sink, int iters, int with_adds) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  for (int i = threadIdx.x; i < LDS_BYTES / 16; i += blockDim.x) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const uint32_t *a = addr + ((size_t)(threadIdx.x >> 6) * 64 + (threadIdx.x & 63)) * 8;
  uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7];
  float acc = 0.f;
  int n = __builtin_amdgcn_readfirstlane(iters);
  if (with_adds) {
    asm volatile(
        "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"
        "ds_read_b128 v[16:19], %1\n ds_read_b128 v[20:23], %2\n ds_read_b128 v[24:27], %3\n ds_read_b128 v[28:31], %4\n"
        "1:\n"
        "ds_read_b128 v[32:35], %5\n ds_read_b128 v[36:39], %6\n ds_read_b128 v[40:43], %7\n ds_read_b128 v[44:47], %8\n"
        "s_waitcnt lgkmcnt(4)\n"
        "v_pk_add_f32 v[64:65], v[16:17], v[64:65]\n v_pk_add_f32 v[66:67], v[18:19], v[66:67]\n v_pk_add_f32 v[64:65], v[20:21], v[64:65]\n v_pk_add_f32 v[66:67], v[22:23], v[66:67]\n"
        "v_pk_add_f32 v[64:65], v[24:25], v[64:65]\n v_pk_add_f32 v[66:67], v[26:27], v[66:67]\n v_pk_add_f32 v[64:65], v[28:29], v[64:65]\n v_pk_add_f32 v[66:67], v[30:31], v[66:67]\n"
        "ds_read_b128 v[16:19], %1\n ds_read_b128 v[20:23], %2\n ds_read_b128 v[24:27], %3\n ds_read_b128 v[28:31], %4\n"
        "s_waitcnt lgkmcnt(4)\n"
        "v_pk_add_f32 v[64:65], v[32:33], v[64:65]\n v_pk_add_f32 v[66:67], v[34:35], v[66:67]\n v_pk_add_f32 v[64:65], v[36:37], v[64:65]\n v_pk_add_f32 v[66:67], v[38:39], v[66:67]\n"
        "v_pk_add_f32 v[64:65], v[40:41], v[64:65]\n v_pk_add_f32 v[66:67], v[42:43], v[66:67]\n v_pk_add_f32 v[64:65], v[44:45], v[64:65]\n v_pk_add_f32 v[66:67], v[46:47], v[66:67]\n"
        "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n v_add_f32 %9, v64, v66\n"
        : "+s"(n) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(acc)
        : "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
          "v40","v41","v42","v43","v44","v45","v46","v47","v64","v65","v66","v67","scc","memory");
  } else {
    asm volatile(
        "1:\n"
        "ds_read_b128 v[16:19], %1\n ds_read_b128 v[20:23], %2\n ds_read_b128 v[24:27], %3\n ds_read_b128 v[28:31], %4\n"
        "s_waitcnt lgkmcnt(4)\n"
        "ds_read_b128 v[32:35], %5\n ds_read_b128 v[36:39], %6\n ds_read_b128 v[40:43], %7\n ds_read_b128 v[44:47], %8\n"
        "s_waitcnt lgkmcnt(4)\n"
        "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"
        : "+s"(n) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
        : "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
          "v40","v41","v42","v43","v44","v45","v46","v47","scc","memory");
  }
  if (acc == -1.f) sink[0] = acc;
}

int main() {
  hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int iters = 4000, rows = 1240;
  uint32_t *d; float *s;
  hipMalloc(&d, 16 * 64 * 8 * 4); hipMalloc(&s, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"linear (lane * 16 + i * 1024)", "8 rows per instruction, partners of opposite parity", "8 rows per instruction, random parity",
                         "8 rows per instruction, all the same parity", "ONE row per instruction (all groups the same row)"};
  for (int pat = 0; pat < 5; ++pat) {
    std::vector<uint32_t> h(16 * 64 * 8);
    srand(7);
    for (int w = 0; w < 16; ++w)
      for (int i = 0; i < 8; ++i) {
        int row_of_group[8];
        for (int g = 0; g < 8; ++g) row_of_group[g] = rand() % rows;
        if (pat == 1) { const int partner[8] = {3, 2, 1, 0, 7, 6, 5, 4}; for (int g = 0; g < 8; ++g) if (g < partner[g]) { row_of_group[g] &= ~1; row_of_group[partner[g]] |= 1; } }
        if (pat == 3) for (int g = 0; g < 8; ++g) row_of_group[g] &= ~1;
        if (pat == 4) for (int g = 1; g < 8; ++g) row_of_group[g] = row_of_group[0];
        for (int l = 0; l < 64; ++l)
          h[((size_t)w * 64 + l) * 8 + i] = pat == 0 ? (uint32_t)(l * 16 + ((i + 8 * w) % 150) * 1024) : (uint32_t)(row_of_group[l >> 3] * 128 + (l & 7) * 16);
      }
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int waves : {4, 8, 15, 16})
      for (int adds = 0; adds < 2; ++adds) {
        k_probe<<<256, waves * 64, LDS_BYTES>>>(d, s, 10, adds);
        hipEventRecord(e0);
        k_probe<<<256, waves * 64, LDS_BYTES>>>(d, s, iters, adds);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 8 * waves;       // per CU
        printf("%-52s %2d waves %s: %.3f ms -> %.2f ns per ds_read_b128 per CU (= %.2f cycles at 2.1 GHz; the LDS fill costs ~10 us)\n", names[pat], waves,
               adds ? "with 2 v_pk_add_f32 per read" : "reads only                  ", ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.1);
      }
  }
  return 0;
}
