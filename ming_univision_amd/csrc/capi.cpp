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

// ---- tensor-parallel communicator memory (host-side setup, mingnative.h §6) ----------------------------------------------
// Fine-grained allocations are coherent at system scope: peer GPUs' posted writes and this GPU's reads need no cache maintenance
// beyond the release / acquire the kernels already do.
extern "C" int mn_tp_alloc(size_t bytes, void** dptr) {
  MN_CHECK_ARG(dptr && bytes > 0, "mn_tp_alloc: bad args");
  hipError_t e = hipExtMallocWithFlags(dptr, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { mn_set_error("mn_tp_alloc(%zu): %s", bytes, hipGetErrorString(e)); return MN_ELAUNCH; }
  e = hipMemset(*dptr, 0, bytes);
  if (e != hipSuccess) { mn_set_error("mn_tp_alloc memset: %s", hipGetErrorString(e)); return MN_ELAUNCH; }
  return MN_OK;
}
extern "C" int mn_tp_free(void* dptr) { return hipFree(dptr) == hipSuccess ? MN_OK : MN_ELAUNCH; }
extern "C" int mn_tp_ipc_handle(void* dptr, void* handle_out_64) {
  MN_CHECK_ARG(dptr && handle_out_64, "mn_tp_ipc_handle: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, dptr);
  if (e != hipSuccess) { mn_set_error("hipIpcGetMemHandle: %s", hipGetErrorString(e)); return MN_ELAUNCH; }
  memcpy(handle_out_64, &h, 64);
  return MN_OK;
}
extern "C" int mn_tp_ipc_open(const void* handle_64, void** dptr) {
  MN_CHECK_ARG(handle_64 && dptr, "mn_tp_ipc_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle_64, 64);
  const hipError_t e = hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { mn_set_error("hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return MN_ELAUNCH; }
  return MN_OK;
}
extern "C" int mn_tp_ipc_close(void* dptr) { return hipIpcCloseMemHandle(dptr) == hipSuccess ? MN_OK : MN_ELAUNCH; }
