// Cost model of a chain of tiny dependent launches on MI355X: empty kernel, 1 / 2 / 3 dependent global loads,
// grid of 1 / 64 / 256 workgroups.   hipcc --offload-arch=gfx950 -O3 -o scripts/launch_bench scripts/launch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(int *p) { if (p == nullptr) return; }
template <int DEP>
__global__ void k_chain(const int *__restrict__ idx, int *out, int n) {
  int v = (blockIdx.x * blockDim.x + threadIdx.x) % n;
#pragma unroll
  for (int d = 0; d < DEP; ++d) v = idx[v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  const int n = 1 << 22;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 2654435761LL + 12345) % n);
  int *idx, *out;
  hipMalloc(&idx, n * 4); hipMalloc(&out, n * 4);
  hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 3000;
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 200; ++i) launch();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.2f us per launch\n", name, ms * 1e3 / reps);
  };
  for (int g : {1, 64, 256, 1024}) {
    char nm[64];
    snprintf(nm, 64, "empty, %d workgroups x 256", g);
    run(nm, [&] { k_empty<<<g, 256, 0, st>>>(out); });
    snprintf(nm, 64, "1 dependent load, %d workgroups", g);
    run(nm, [&] { k_chain<1><<<g, 256, 0, st>>>(idx, out, n); });
    snprintf(nm, 64, "2 dependent loads, %d workgroups", g);
    run(nm, [&] { k_chain<2><<<g, 256, 0, st>>>(idx, out, n); });
    snprintf(nm, 64, "4 dependent loads, %d workgroups", g);
    run(nm, [&] { k_chain<4><<<g, 256, 0, st>>>(idx, out, n); });
  }
  run("empty, 200 workgroups x 1024", [&] { k_empty<<<200, 1024, 0, st>>>(out); });
  return 0;
}
