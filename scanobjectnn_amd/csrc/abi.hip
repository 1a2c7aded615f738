// abi.hip -- status strings and ABI version of libpcops.
#include "common.h"

extern "C" const char *pcops_strerror(int status) {
    switch (status) {
        case PCOPS_OK: return "ok";
        case PCOPS_ERR_NULL_POINTER: return "null pointer argument";
        case PCOPS_ERR_BAD_SHAPE: return "invalid tensor extents";
        case PCOPS_ERR_BAD_ARGUMENT: return "invalid attribute (radius/nsample/npoint/k)";
        case PCOPS_ERR_UNSUPPORTED: return "shape outside the range the gfx950 kernels are built for";
        case PCOPS_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown pcops status";
    }
}

extern "C" int pcops_abi_version(void) { return 1; }
