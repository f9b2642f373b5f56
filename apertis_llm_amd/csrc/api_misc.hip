// Version / error-string entry points of libapertis_hip.so.
#include "common.h"

extern "C" int apertis_abi_version(void) { return APERTIS_ABI_VERSION; }
extern "C" const char *apertis_arch(void) { return "gfx950"; }
extern "C" const char *apertis_strerror(int code) {
  switch (code) {
    case APERTIS_OK: return "ok";
    case APERTIS_ERR_ARG: return "invalid argument (null pointer, negative size or bad enum)";
    case APERTIS_ERR_UNSUPPORTED: return "shape or dtype combination not supported by the gfx950 kernels";
    case APERTIS_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error";
  }
}
