// lib.hip -- library-level entry points of the C ABI (include/vvcgpu.h).
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vvcgpu_set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" {
int vvcgpu_version(void) { return 1; }
const char* vvcgpu_last_error(void) { return g_err; }
int vvcgpu_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int vvcgpu_set_device(int device)
{
  VVC_HIP(hipSetDevice(device));
  return VVCGPU_OK;
}
}
