// The inner loop of k_spmm_ring in isolation (conflict-free gathered addresses from registers, no stream, no barriers), one feature
// at a time: what keeps the real kernel at 3.3 ns per ds_read_b128 per CU when the bare reads run at 1.8?
//   hipcc --offload-arch=gfx950 -O3 scripts/ring_loop_probe.hip -o scripts/ring_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int LDS_BYTES = 158976;

// FEAT bits: 1 = v_mad_u32_u16 address per read (from the row index words), 2 = index mode on/off around the additions,
// 4 = s_bfe of the accumulator offset per pair, 8 = accumulator changes every pair (s42 cycles through 16 offsets)
template <int FEAT>
__global__ void __launch_bounds__(1024) k_loop(const uint32_t *__restrict__ words, float *sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  for (int i = threadIdx.x; i < LDS_BYTES / 16; i += blockDim.x) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const uint32_t *a = words + ((size_t)(threadIdx.x >> 6) * 64 + (threadIdx.x & 63)) * 4;
  uint32_t w0 = a[0], w1 = a[1], w2 = a[2], w3 = a[3];        // 8 row indices (16 bit each)
  uint32_t lane_off = (threadIdx.x & 7) * 16;
  float acc = 0.f;
  int n = __builtin_amdgcn_readfirstlane(iters);
#define PAIR(SLOT_LO, SLOT_A, SLOT_B, SLOT_C, SLOT_D, R0, R1, W)                                                      \
  ".if %c7 & 4\n s_bfe_u32 s42, s41, 0x60008\n .endif\n"                                                              \
  ".if %c7 & 8\n s_add_u32 s42, s42, 4\n s_and_b32 s42, s42, 60\n .endif\n"                                            \
  "s_waitcnt lgkmcnt(6)\n"                                                                                            \
  ".if %c7 & 2\n s_set_gpr_idx_on s42, 0xa\n .endif\n"                                                                \
  "v_pk_add_f32 v[64:65], " SLOT_A ", v[64:65]\n v_pk_add_f32 v[66:67], " SLOT_B ", v[66:67]\n"                       \
  "v_pk_add_f32 v[64:65], " SLOT_C ", v[64:65]\n v_pk_add_f32 v[66:67], " SLOT_D ", v[66:67]\n"                       \
  ".if %c7 & 2\n s_set_gpr_idx_off\n .endif\n"                                                                        \
  ".if %c7 & 1\n v_mad_u32_u16 v8, " W ", s43, %5\n v_mad_u32_u16 v9, " W ", s43, %5 op_sel:[1,0,0,0]\n .endif\n"      \
  "ds_read_b128 " R0 ", v8\n ds_read_b128 " R1 ", v9\n"
  asm volatile(
      "s_movk_i32 s43, 0x80\n s_mov_b32 s42, 0\n s_mov_b32 s41, 0x0c080400\n"
      "v_mad_u32_u16 v8, %1, s43, %5\n v_mad_u32_u16 v9, %1, s43, %5 op_sel:[1,0,0,0]\n"
      ".irp r, 64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127\n v_mov_b32 v\\r, 0\n .endr\n"
      "ds_read_b128 v[16:19], v8\n ds_read_b128 v[20:23], v9\n ds_read_b128 v[24:27], v8\n ds_read_b128 v[28:31], v9\n"
      "ds_read_b128 v[32:35], v8\n ds_read_b128 v[36:39], v9\n ds_read_b128 v[40:43], v8\n ds_read_b128 v[44:47], v9\n"
      "1:\n"
      PAIR(0, "v[16:17]", "v[18:19]", "v[20:21]", "v[22:23]", "v[16:19]", "v[20:23]", "%1")
      PAIR(1, "v[24:25]", "v[26:27]", "v[28:29]", "v[30:31]", "v[24:27]", "v[28:31]", "%2")
      PAIR(2, "v[32:33]", "v[34:35]", "v[36:37]", "v[38:39]", "v[32:35]", "v[36:39]", "%3")
      PAIR(3, "v[40:41]", "v[42:43]", "v[44:45]", "v[46:47]", "v[40:43]", "v[44:47]", "%4")
      "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
      "s_waitcnt lgkmcnt(0)\n v_add_f32 %6, v64, v66\n"
      : "+s"(n) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(lane_off), "v"(acc), "n"(FEAT)
      : "v8","v9","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
        "v40","v41","v42","v43","v44","v45","v46","v47","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83",
        "v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109",
        "v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","s41","s42","s43","scc","memory");
  if (acc == -1.f) sink[0] = acc;
}

template <int FEAT> void run(const uint32_t *d, float *s, const char *what) {
  hipFuncSetAttribute((const void *)k_loop<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int waves : {15}) {
    k_loop<FEAT><<<256, waves * 64, LDS_BYTES>>>(d, s, 10);
    hipEventRecord(e0);
    k_loop<FEAT><<<256, waves * 64, LDS_BYTES>>>(d, s, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s %2d waves: %.2f ns per ds_read_b128 per CU\n", what, waves, ms * 1e6 / ((double)iters * 8 * waves));
  }
}

int main() {
  uint32_t *d; float *s;
  hipMalloc(&d, 16 * 64 * 4 * 4); hipMalloc(&s, 16);
  std::vector<uint32_t> h(16 * 64 * 4);
  srand(7);
  const int partner[8] = {3, 2, 1, 0, 7, 6, 5, 4};
  for (int w = 0; w < 16; ++w)
    for (int i = 0; i < 8; ++i) {
      int row[8];
      for (int g = 0; g < 8; ++g) row[g] = rand() % 1240;
      for (int g = 0; g < 8; ++g) if (g < partner[g]) { row[g] &= ~1; row[partner[g]] |= 1; }
      for (int l = 0; l < 64; ++l) {
        uint32_t &word = h[((size_t)w * 64 + l) * 4 + i / 2];
        word = (i & 1) ? (word & 0xffffu) | ((uint32_t)row[l >> 3] << 16) : (word & 0xffff0000u) | (uint32_t)row[l >> 3];
      }
    }
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<1>(d, s, "reads + adds + address mad");
  run<3>(d, s, "+ index mode on / off per pair");
  run<7>(d, s, "+ s_bfe per pair");
  run<11>(d, s, "+ index mode, accumulator changes every pair");
  run<15>(d, s, "all");
  run<0>(d, s, "reads + adds only (same two addresses)");
  return 0;
}
