// Is "block b runs on XCD b % 8" stable across consecutive launches, odd grid sizes, block sizes and CU-masked streams?
//   hipcc --offload-arch=gfx950 -O2 scripts/xcd_block_map.hip -o scripts/xcd_block_map && scripts/xcd_block_map
// For every launch: the number of blocks whose XCC_ID differs from blockIdx.x % 8.  (The plan kernels skip the blocks with
// blockIdx.x % 8 == 0 when the dense chunk kernel owns XCD 0: they must then really be the blocks XCD 0 would have run.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k_map(int *mismatch, int *per_xcc, int spin) {
  extern __shared__ int dyn[];
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if ((int)xcc != (int)(blockIdx.x % 8)) atomicAdd(mismatch, 1);
    atomicAdd(&per_xcc[xcc], 1);
    dyn[0] = (int)xcc;
  }
  for (volatile int i = 0; i < spin; ++i) {}
}

static hipStream_t masked(const std::vector<int> &bits) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b : bits) mask[b / 32] |= 1u << (b % 32);
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("mask stream failed\n"); return nullptr; }
  return st;
}

int main() {
  int *mm, *px;
  hipMalloc(&mm, 4); hipMalloc(&px, 32);
  // bit i of the mask = CU i / 8 of XCD i % 8
  std::vector<int> plan_bits, dense_bits;
  for (int i = 0; i < 256; ++i) {
    const int xcd = i % 8, cu = i / 8;
    if (xcd != 0 || cu >= 28) plan_bits.push_back(i);
    if (xcd == 0 && cu < 28) dense_bits.push_back(i);
    if (xcd != 0 && cu == 0) dense_bits.push_back(i);
  }
  hipStream_t plain, pm = masked(plan_bits), dm = masked(dense_bits);
  hipStreamCreate(&plain);
  struct Case { const char *name; hipStream_t st; };
  Case cases[] = {{"plain stream", plain}, {"plan mask (XCD0: CUs 28..31, others all)", pm}, {"dense mask (XCD0: CUs 0..27, others CU 0)", dm}};
  const int grids[] = {8, 9, 13, 224, 256, 257, 1000, 1003, 17101, 2500, 7, 65536};
  for (auto &c : cases) {
    if (!c.st) continue;
    printf("%s\n", c.name);
    for (int threads : {64, 256, 1024}) {
      for (int g : grids) {
        for (int lds : {0, 100 * 1024}) {
          if (lds && threads != 1024) continue;
          hipMemsetAsync(mm, 0, 4, c.st); hipMemsetAsync(px, 0, 32, c.st);
          if (lds) hipFuncSetAttribute((const void *)k_map, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          k_map<<<g, threads, lds ? lds : 4, c.st>>>(mm, px, g < 3000 ? 2000 : 50);
          hipStreamSynchronize(c.st);
          int m, p[8];
          hipMemcpy(&m, mm, 4, hipMemcpyDeviceToHost); hipMemcpy(p, px, 32, hipMemcpyDeviceToHost);
          printf("  threads %4d lds %6d grid %6d: mismatches %6d   per XCC:", threads, lds, g, m);
          for (int i = 0; i < 8; ++i) printf(" %d", p[i]);
          printf("\n");
        }
      }
    }
  }
  return 0;
}
