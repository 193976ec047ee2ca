// capi.cpp — host-side plumbing of the C ABI: version, error string, device query.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mn_version(void) { return MN_VERSION; }
extern "C" const char* mn_last_error(void) { return g_err; }

extern "C" int mn_num_cus(void) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;  // MI355X
  }
  return cus;
}
