// capi.hip — library-level entry points of the C ABI (include/pyg_amd.h).
#include "common.h"

namespace pygamd {
thread_local int g_last_hip_error = 0;
}

extern "C" {

int pygamd_abi_version(void) { return PYGAMD_ABI_VERSION; }

const char* pygamd_status_string(int status) {
  switch (status) {
    case PYGAMD_OK:
      return "ok";
    case PYGAMD_ERR_INVALID_ARG:
      return "invalid argument";
    case PYGAMD_ERR_UNSUPPORTED:
      return "unsupported combination";
    case PYGAMD_ERR_WORKSPACE:
      return "workspace too small";
    case PYGAMD_ERR_HIP:
      return "HIP runtime error";
    default:
      return "unknown status";
  }
}

int pygamd_last_hip_error(void) { return pygamd::g_last_hip_error; }

const char* pygamd_build_arch(void) { return "gfx950"; }

}  // extern "C"
