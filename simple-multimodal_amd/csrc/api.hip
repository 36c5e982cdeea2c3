// ABI plumbing: version, thread-local error string, device query.
#include "mmf_internal.h"

static thread_local char g_err[512] = "";

void mmf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mmf_version(void) { return MMF_ABI_VERSION; }
extern "C" const char* mmf_last_error(void) { return g_err; }
extern "C" int mmf_device_cu_count(void) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    mmf_set_error("mmf_device_cu_count: no HIP device");
    return MMF_E_LAUNCH;
  }
  return prop.multiProcessorCount;
}
