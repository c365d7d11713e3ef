// Error reporting + ABI bookkeeping for libnacf_hip.
#include "common.hpp"

static thread_local char g_err[512] = "";

void nacf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ void rng_advance_kernel(uint64_t* st) { st[1] += 1; }

extern "C" {
const char* nacf_last_error(void) { return g_err; }
int nacf_version(void) { return 1; }
int nacf_abi_count(void) { return NACF_ABI_COUNT; }
int nacf_rng_advance(uint64_t* rng_state, nacf_stream_t stream) {
  NACF_CHECK(rng_state, NACF_EINVAL, "nacf_rng_advance: null pointer");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, as_hip(stream), rng_state);
  NACF_LAUNCH_CHECK("nacf_rng_advance");
  return NACF_OK;
}
}
