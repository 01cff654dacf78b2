// Which XCDs do the bits of a HIP CU mask select?  hipcc --offload-arch=gfx950 -O2 scripts/cu_mask_xcc.hip -o /tmp/cu_mask_xcc && /tmp/cu_mask_xcc
// For a few masks: the XCC_ID histogram of 4096 workgroups launched on a stream confined to that mask.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k_where(int *hist, int *cu_seen) {
  if (threadIdx.x == 0) {
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    atomicAdd(&hist[xcc & 15], 1);
    const unsigned cu = (hwid >> 8) & 15, sh = (hwid >> 12) & 1, se = (hwid >> 13) & 7;     // cu_id, sh_id, se_id
    cu_seen[(xcc & 7) * 256 + (se * 2 + sh) * 16 + cu] = 1;
  }
  for (volatile int i = 0; i < 2000; ++i) {}
}

static void run(const char *name, const std::vector<int> &bits) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b : bits) mask[b / 32] |= 1u << (b % 32);
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  int *hist, *seen;
  hipMalloc(&hist, 16 * sizeof(int));
  hipMalloc(&seen, 2048 * sizeof(int));
  hipMemset(hist, 0, 16 * sizeof(int));
  hipMemset(seen, 0, 2048 * sizeof(int));
  k_where<<<65536, 64, 0, st>>>(hist, seen);
  hipStreamSynchronize(st);
  int h[16];
  hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s XCC histogram:", name);
  for (int i = 0; i < 8; ++i) printf(" %5d", h[i]);
  std::vector<int> sv(2048);
  hipMemcpy(sv.data(), seen, 2048 * sizeof(int), hipMemcpyDeviceToHost);
  printf("   distinct CUs per XCC:");
  for (int x = 0; x < 8; ++x) { int c = 0; for (int i = 0; i < 256; ++i) c += sv[x * 256 + i]; printf(" %3d", c); }
  printf("\n");
  hipStreamDestroy(st);
  hipFree(hist);
  hipFree(seen);
}

int main() {
  std::vector<int> a, b, c, d;
  for (int i = 0; i < 32; ++i) a.push_back(i);
  for (int i = 0; i < 256; i += 8) b.push_back(i);
  for (int i = 0; i < 64; ++i) c.push_back(i);
  for (int i = 0; i < 256; ++i) if (i % 8 < 2) d.push_back(i);
  run("bits 0..31", a);
  run("every 8th bit (0, 8, ...)", b);
  run("bits 0..63", c);
  run("bits with i % 8 < 2", d);
  return 0;
}
