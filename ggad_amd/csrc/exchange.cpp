// One-shot gradient exchange between the ranks of one node (host side): buffers in fine-grained device memory, exported /
// imported as HIP IPC handles, written directly by the peers' kernels over xGMI (k_xchg_adam, step.hip).
//
// The reference has no data parallelism at all (src/main.py:9 pins one device); SURVEY.md section 8e adds it: W ranks take W
// consecutive batches of the reference's stream, the 5,248 gradients are summed and the identical Adam step runs everywhere.
// At 21 KB the exchange is pure latency: an RCCL all-reduce costs a launch plus two stream hand-offs per 45 us step.  Here every
// thread of the gradient-reduce launch stores its parameter's gradient as an 8-byte {value, step} granule straight into all
// ranks' buffers, reads the W granules of its own buffer until they carry this step's number, sums them in rank order
// (bit-identical on every rank) and applies Adam -- no extra launch, no flag, no fence, no host involvement, no collective
// library call inside the step loop.
#include <cstdlib>
#include <cstring>

#include "common.h"

extern "C" {

int ggad_xchg_create(int32_t rank, int32_t world, int64_t n_floats, ggad_xchg **out) {
  GGAD_REQUIRE(out && world >= 1 && world <= GGAD_XCHG_MAX_WORLD && rank >= 0 && rank < world && n_floats >= 1);
  ggad_xchg *x = new ggad_xchg();
  std::memset(x, 0, sizeof(*x));
  x->bytes = ggad_xchg_granules(world, n_floats) * sizeof(uint64_t) + 64;
  // fine-grained (uncached, device-coherent across agents) memory: peers' stores become visible without a kernel boundary
  hipError_t e = hipExtMallocWithFlags(&x->local, x->bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { ggad_set_error(e, "xchg_create (hipExtMallocWithFlags)"); delete x; return GGAD_E_LAUNCH; }
  e = hipMemset(x->local, 0, x->bytes);
  if (e != hipSuccess) { ggad_set_error(e, "xchg_create (memset)"); (void)hipFree(x->local); delete x; return GGAD_E_LAUNCH; }
  x->view.rank = rank;
  x->view.world = world;
  x->view.n = n_floats;
  x->view.peer[rank] = static_cast<float *>(x->local);
  x->view.err = reinterpret_cast<int32_t *>(static_cast<char *>(x->local) + x->bytes - 64);
  {                                                         // a slow peer (checkpoint, first touch) is not a lost peer: seconds, not spins
    const char *ev = getenv("GGAD_XCHG_TIMEOUT_S");
    const double sec = ev ? atof(ev) : 20.0;
    x->view.timeout_ticks = (unsigned long long)((sec > 0.001 ? sec : 0.001) * 1e8);
  }
  *out = x;
  return GGAD_OK;
}

int32_t ggad_xchg_handle_bytes(void) { return (int32_t)sizeof(hipIpcMemHandle_t); }

int ggad_xchg_handle(ggad_xchg *x, void *handle_out) {
  GGAD_REQUIRE(x && handle_out);
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, x->local);
  if (e != hipSuccess) { ggad_set_error(e, "xchg_handle (hipIpcGetMemHandle)"); return GGAD_E_LAUNCH; }
  std::memcpy(handle_out, &h, sizeof(h));
  return GGAD_OK;
}

/* handles: world x ggad_xchg_handle_bytes() bytes, rank order (the own entry is ignored) */
int ggad_xchg_connect(ggad_xchg *x, const void *handles) {
  GGAD_REQUIRE(x && handles);
  const char *hb = static_cast<const char *>(handles);
  for (int q = 0; q < x->view.world; ++q) {
    if (q == x->view.rank || x->opened[q]) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, hb + (size_t)q * sizeof(h), sizeof(h));
    void *p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { ggad_set_error(e, "xchg_connect (hipIpcOpenMemHandle)"); return GGAD_E_LAUNCH; }
    x->view.peer[q] = static_cast<float *>(p);
    x->opened[q] = true;
  }
  return GGAD_OK;
}

int ggad_xchg_error(ggad_xchg *x, int32_t *err_host) {
  GGAD_REQUIRE(x && err_host);
  const hipError_t e = hipMemcpy(err_host, x->view.err, sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { ggad_set_error(e, "xchg_error"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}

int ggad_xchg_destroy(ggad_xchg *x) {
  if (!x) return GGAD_OK;
  for (int q = 0; q < x->view.world; ++q)
    if (x->opened[q]) (void)hipIpcCloseMemHandle(x->view.peer[q]);
  if (x->local) (void)hipFree(x->local);
  delete x;
  return GGAD_OK;
}

}  // extern "C"
