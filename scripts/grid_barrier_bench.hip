// What would a persistent dense-step kernel pay per phase?  G workgroups x 256 threads loop over K phases; between phases a
// grid barrier (one relaxed agent-scope atomic add per workgroup + a polling loop on the same counter), and a hand-off: every
// workgroup writes a value the NEXT workgroup reads in the following phase (write-through store / L1-bypassing load), checked.
// Variants: all workgroups on one XCD (CU mask) or spread over the chip.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/grid_barrier_bench scripts/grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// mode 0: barrier only; mode 1: + hand-off of 256 floats per workgroup between phases
__global__ void __launch_bounds__(256) k_persistent(unsigned *counter, float *buf, int K, int mode, unsigned *errors) {
  const int G = gridDim.x, b = blockIdx.x;
  unsigned bad = 0;
  for (int k = 0; k < K; ++k) {
    if (mode) {
      // write-through, agent scope: visible to the other XCDs' L2s without a release fence
      __hip_atomic_store(&buf[(size_t)(k & 1) * G * 256 + b * 256 + threadIdx.x], (float)(k * 1000 + b), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(counter, (unsigned)(k + 1) * G);
    if (mode) {
      const int src = (b + 1) % G;
      const float v = __hip_atomic_load(&buf[(size_t)(k & 1) * G * 256 + src * 256 + threadIdx.x], __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
      bad += (v != (float)(k * 1000 + src));
    }
  }
  if (bad) atomicAdd(errors, bad);
}

int main() {
  unsigned *counter, *errors; float *buf;
  hipMalloc(&counter, 4); hipMalloc(&errors, 4); hipMalloc(&buf, 2 * 256 * 256 * 4);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
  const int K = 2000;
  for (int variant = 0; variant < 3; ++variant) {
    // 0: plain stream; 1: CU mask = XCD 0 only (bits 0, 8, 16, ...); 2: CU mask = 64 CUs on XCDs 0 and 1
    hipStream_t st;
    if (variant == 0) hipStreamCreate(&st);
    else {
      std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
      for (int c = 0; c < ncu; ++c) if ((c % 8) < variant) mask[c / 32] |= 1u << (c % 32);
      if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("cu mask failed\n"); continue; }
    }
    for (int G : {8, 32, 64}) {
      if (variant == 1 && G > 32) continue;       // must be co-resident: one workgroup per CU at most here
      for (int mode = 0; mode < 2; ++mode) {
        hipMemsetAsync(counter, 0, 4, st); hipMemsetAsync(errors, 0, 4, st);
        hipEventRecord(a, st);
        k_persistent<<<G, 256, 0, st>>>(counter, buf, K, mode, errors);
        hipEventRecord(e, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, e);
        unsigned herr; hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
        printf("%-22s G=%2d %-9s %.2f us per phase, stale reads %u\n",
               variant == 0 ? "whole chip" : (variant == 1 ? "one XCD (32 CUs)" : "two XCDs (64 CUs)"), G,
               mode ? "+hand-off" : "barrier", ms * 1e3 / K, herr);
      }
    }
    hipStreamDestroy(st);
  }
  return 0;
}
