// Version / error strings of the C ABI.
#include "common.h"

extern "C" int jg_version(void) { return 100; }

extern "C" const char* jg_strerror(int code) {
  switch (code) {
    case JG_OK: return "ok";
    case JG_ERR_BAD_ARG: return "bad argument (null pointer, misaligned leading dimension or unsupported shape)";
    case JG_ERR_UNSUPPORTED: return "unsupported configuration";
    case JG_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}
