// C-ABI plumbing shared by all kernels: thread-local error string, version, device query.
#include "common.h"
#include "../../include/pixart_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

void pxa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* pxa_last_error(void) { return g_err; }
extern "C" int pxa_abi_version(void) { return PXA_ABI_VERSION; }
extern "C" int pxa_operand_dtype(void) { return PXA_OPERAND_DTYPE_ID; }
extern "C" int pxa_device_info(int* cu_count, int* is_gfx950) {
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { pxa_set_error("no HIP device"); return -1; }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (is_gfx950) *is_gfx950 = (strncmp(p.gcnArchName, "gfx950", 6) == 0);
  return 0;
}
