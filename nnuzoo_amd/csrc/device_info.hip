// Library probe entry points (no kernels).
#include "common.hpp"
#include <string.h>

extern "C" int nnz_version(void) { return 100; }

extern "C" int nnz_device_info(char* arch, int arch_len, int* num_cu, long* hbm_bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (long)prop.totalGlobalMem;
  return NNZ_OK;
}
