// XCD-resident synchronisation primitives on gfx950: what does a barrier + data hand-off cost when every participating workgroup
// sits on ONE XCD (one L2)?   hipcc --offload-arch=gfx950 -O2 scripts/xcd_barrier_bench.hip -o /tmp/xcd_barrier_bench
//
// A grid of NWG workgroups is launched; every workgroup reads HW_REG_XCC_ID, registers itself in a per-XCD counter and waits
// until the whole grid has registered; the workgroups of XCD `pick` stay (rank = registration order inside the XCD), the others
// exit.  The survivors run `iters` rounds of  write payload -> barrier -> read the payload of other workgroups (checked word by
// word) .  Hand-off forms:
//   mode 0  tagged slots: plain payload stores, per-wave s_waitcnt vmcnt(0), __syncthreads, ONE plain 4-byte store of the round number
//           into slot[rank]; wave 0 polls all slots with ONE sc1 (L1-bypassing, L2-served) load per lane
//   mode 1  counter: the same drain, then a relaxed AGENT-scope atomic add on one counter, sc1 poll of the counter
//   mode 2  counter with a WORKGROUP-scope atomic add (executes in the XCD's L2, no sc1), sc1 poll
// payload reads are sc1 loads (relaxed agent-scope atomic loads): they bypass the reader's L1 and are served by the shared L2.
// `skew` > 0: workgroup r sleeps (r * 37 + round * 11) % skew  x 64 clocks before arriving (uneven load).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Ctrl {
  unsigned reg[8];       // workgroups registered per XCD
  unsigned total;        // workgroups registered
  unsigned pad[7];
  unsigned counter;      // modes 1 / 2
  unsigned pad2[15];
  unsigned slot[64];     // mode 0: round number per rank
  unsigned long long clocks;   // wall clocks (100 MHz) of rank 0 over the loop
  unsigned errors, survivors, wave_l2_lat;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1f(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__device__ __forceinline__ void xcd_barrier(Ctrl *C, int rank, int G, unsigned round) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    if (MODE == 0) {
      if (threadIdx.x == 0) C->slot[rank] = round;
      const int l = threadIdx.x < G ? threadIdx.x : 0;
      while (true) {
        const unsigned v = ld_sc1(&C->slot[l]);
        if (__all((int)(v - round) >= 0)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
      if (threadIdx.x == 0) {
        if (MODE == 1) __hip_atomic_fetch_add(&C->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&C->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      while (ld_sc1(&C->counter) < round * (unsigned)G) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(512) k_bench(Ctrl *C, float *payload, int iters, int words, int skew, int pick) {
  __shared__ int s_rank, s_G;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const unsigned r = __hip_atomic_fetch_add(&C->reg[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&C->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (ld_sc1(&C->total) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    s_rank = (int)xcc == pick ? (int)r : -1;
    s_G = (int)ld_sc1(&C->reg[pick]);
  }
  __syncthreads();
  const int rank = s_rank, G = s_G;
  if (rank < 0) return;
  if (rank == 0 && threadIdx.x == 0) C->survivors = G;
  unsigned errors = 0;
  const unsigned long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    float *mine = payload + ((size_t)(it & 1) * 64 + rank) * words;
    for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = (float)(it * 64 + rank) + (float)i * 0.5f;
    if (skew > 0 && threadIdx.x == 0) {
      const int n = (rank * 37 + it * 11) % skew;
      for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1);
    }
    xcd_barrier<MODE>(C, rank, G, (unsigned)it);
    for (int hop = 1; hop <= 3; ++hop) {
      const int src = (rank + hop * 5 + it) % G;
      const float *theirs = payload + ((size_t)(it & 1) * 64 + src) * words;
      for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const float v = ld_sc1f(theirs + i);
        if (v != (float)(it * 64 + src) + (float)i * 0.5f) ++errors;
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (errors) atomicAdd(&C->errors, errors);
  if (rank == 0 && threadIdx.x == 0) C->clocks = t1 - t0;
}

// latency of a dependent chain of sc1 loads served by the L2 (pointer chase inside 64 KB), one lane
__global__ void k_chase(const unsigned *chain, int n, unsigned *out, unsigned long long *clk) {
  unsigned p = 0;
  for (int i = 0; i < 64; ++i) p = ld_sc1(chain + p);      // warm
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) p = ld_sc1(chain + p);
  const unsigned long long t1 = wall_clock64();
  *out = p; *clk = t1 - t0;
}

__global__ void k_noise(float4 *buf, size_t n, int rounds) {     // streaming HBM traffic on the other XCDs (and this one)
  float4 acc = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const float4 v = buf[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  if (acc.x == 12345.678f) buf[0] = acc;
}

template <int MODE>
static void run(const char *name, int nwg, int iters, int words, int skew, bool noise, float4 *nbuf, size_t nn) {
  Ctrl *C; float *payload;
  CK(hipMalloc(&C, sizeof(Ctrl)));
  CK(hipMalloc(&payload, (size_t)2 * 64 * words * sizeof(float)));
  CK(hipMemset(C, 0, sizeof(Ctrl)));
  CK(hipMemset(payload, 0, (size_t)2 * 64 * words * sizeof(float)));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  if (noise) k_noise<<<2048, 256, 0, s2>>>(nbuf, nn, 40);
  k_bench<MODE><<<nwg, 512, 0, s1>>>(C, payload, iters, words, skew, 0);
  CK(hipStreamSynchronize(s1));
  CK(hipDeviceSynchronize());
  Ctrl h;
  CK(hipMemcpy(&h, C, sizeof(Ctrl), hipMemcpyDeviceToHost));
  printf("%-34s nwg %4d survivors %2u words %5d skew %3d noise %d : %7.3f us per round, errors %u  (reg:", name, nwg, h.survivors, words,
         skew, (int)noise, (double)h.clocks / 100.0 / iters, h.errors);
  for (int i = 0; i < 8; ++i) printf(" %u", h.reg[i]);
  printf(")\n");
  CK(hipFree(C)); CK(hipFree(payload));
  CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
}

int main() {
  const size_t nn = (size_t)1 << 26;     // 1 GiB of float4
  float4 *nbuf;
  CK(hipMalloc(&nbuf, nn * sizeof(float4)));
  CK(hipMemset(nbuf, 0, nn * sizeof(float4)));
  {  // L2 latency of an sc1 load
    const int n = 16384;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned)((i * 4099u + 977u) % n);
    unsigned *chain, *out; unsigned long long *clk;
    CK(hipMalloc(&chain, n * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&clk, 8));
    CK(hipMemcpy(chain, h.data(), n * 4, hipMemcpyHostToDevice));
    k_chase<<<1, 64>>>(chain, 4096, out, clk);
    unsigned long long c;
    CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    printf("dependent sc1 load (64 KB footprint): %.1f ns each\n", (double)c * 10.0 / 4096);
  }
  const int iters = 2000;
  for (int words : {512, 4096}) {
    run<0>("mode 0 tagged slots", 256, iters, words, 0, false, nbuf, nn);
    run<1>("mode 1 agent atomic counter", 256, iters, words, 0, false, nbuf, nn);
    run<2>("mode 2 workgroup atomic counter", 256, iters, words, 0, false, nbuf, nn);
  }
  run<0>("mode 0 tagged slots, skew", 256, iters, 512, 40, false, nbuf, nn);
  run<2>("mode 2 wg atomic, skew", 256, iters, 512, 40, false, nbuf, nn);
  run<0>("mode 0 tagged slots, 128 wgs", 128, iters, 512, 0, false, nbuf, nn);
  run<0>("mode 0 tagged slots, 512 wgs", 512, iters, 512, 0, false, nbuf, nn);
  run<0>("mode 0 tagged slots + HBM noise", 256, iters, 512, 0, true, nbuf, nn);
  run<1>("mode 1 agent atomic + HBM noise", 256, iters, 512, 0, true, nbuf, nn);
  run<2>("mode 2 wg atomic + HBM noise", 256, iters, 512, 0, true, nbuf, nn);
  run<0>("mode 0 + noise + skew", 256, iters, 4096, 40, true, nbuf, nn);
  // barrier alone (payload of 0 words)
  run<0>("mode 0 barrier only", 256, iters, 0, 0, false, nbuf, nn);
  run<1>("mode 1 barrier only", 256, iters, 0, 0, false, nbuf, nn);
  run<2>("mode 2 barrier only", 256, iters, 0, 0, false, nbuf, nn);
  return 0;
}
