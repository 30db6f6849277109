// ABI bookkeeping for libfmgan_hip.so.
#include "common.h"

extern "C" int fmgan_abi_version(void) { return FMGAN_ABI_VERSION; }
extern "C" int fmgan_refresh_entry_bytes(void) { return (int)sizeof(fmgan_refresh_entry); }

extern "C" const char* fmgan_status_string(int status) {
  switch (status) {
    case FMGAN_OK: return "ok";
    case FMGAN_EINVAL: return "invalid argument (null pointer, non-positive or inconsistent dimension)";
    case FMGAN_EUNSUPPORTED: return "unsupported dtype / act / mode / path";
    case FMGAN_ELAUNCH: return "HIP kernel launch failed";
    case FMGAN_EOVERFLOW: return "element count exceeds the kernel's index range";
    default: return "unknown status";
  }
}
