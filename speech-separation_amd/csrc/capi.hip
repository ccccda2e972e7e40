// capi.hip -- version, error reporting and device queries of libsepkern's C ABI.
#include "sk_common.h"

static thread_local char g_err[512] = "";

char* sk_errbuf() { return g_err; }

int sk_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int sk_version(void) { return SK_VERSION; }

extern "C" const char* sk_last_error(void) { return g_err; }

extern "C" unsigned sk_build_flags(void) { return sk_gemm_build_flags() | sk_lstm_build_flags(); }

extern "C" int sk_device_info(int* num_cu, int* lds_bytes) {
  int dev = 0;
  hipDeviceProp_t p;
  SK_CHECK_HIP(hipGetDevice(&dev));
  SK_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  if (num_cu) *num_cu = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  return SK_OK;
}
