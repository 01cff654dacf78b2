// Host-side runtime pieces of libggad_hip.so: error reporting, CU-masked streams (the batch sampler lives in sampler.cpp).
#include <string>

#include "common.h"

static thread_local std::string g_last_error;


void ggad_set_error(hipError_t e, const char *where) {
  g_last_error = std::string(where) + ": " + hipGetErrorString(e);
}

// ---- CU-partitioned streams.  The plan of chunk c+1 (two launches that fill the chip for milliseconds) and the dense
// steps of chunk c (hundreds of dependent launches of a few microseconds) only overlap if they do not compete for
// the same compute units: a tiny launch queued behind 600k gather waves waits for a free CU.  Two streams with
// disjoint CU masks give the dense chain its own CUs.
extern "C" int ggad_stream_create_cu_mask(const uint32_t *mask, int32_t n_words, ggad_stream_t *out) {
  if (!mask || n_words <= 0 || !out) return GGAD_E_INVALID;
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask);
  if (e != hipSuccess) { ggad_set_error(e, "stream_create_cu_mask"); return GGAD_E_LAUNCH; }
  *out = reinterpret_cast<ggad_stream_t>(st);
  return GGAD_OK;
}

extern "C" int ggad_stream_destroy(ggad_stream_t stream) {
  if (!stream) return GGAD_OK;
  hipError_t e = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { ggad_set_error(e, "stream_destroy"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}

extern "C" int ggad_device_cu_count(int32_t device, int32_t *out) {
  if (!out) return GGAD_E_INVALID;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { ggad_set_error(e, "device_cu_count"); return GGAD_E_LAUNCH; }
  *out = prop.multiProcessorCount;
  return GGAD_OK;
}

extern "C" {

int ggad_abi_version(void) { return 10; }      // 10: ggad_full_loss_fused_*, ggad_rownorm_bwd_add_f32, ggad_prelu_bwd_one_f32; 9: ggad_linear_prelu_f32, GGAD_E_UNSUPPORTED; 8: ggad_mb_tile_major, ggad_mb_plan.tile_start / col_t; 7: ggad_mlp_score_wgrad_*; 6: ggad_spmm_rowline_*, ggad_prelu_bwd_ld_f32, ggad_mb_plan.node_pack_host  (5: ggad_spmm_ring_*; exports ggad_* only)
const char *ggad_last_error(void) { return g_last_error.c_str(); }

}  // extern "C"
