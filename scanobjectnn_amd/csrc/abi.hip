// abi.hip -- status strings, ABI version and the process-wide switches of libpcops.
#include <atomic>
#include <cstdlib>

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

extern "C" int pcops_abi_version(void) { return 2; }   // 2: stat_pivot (shifted BN moments) on the forward-statistics producers

// bit-reproducible backward passes (reference hazard tf_grouping_g.cu:61-78, tf_sampling_g.cu:183-192:
// float atomics).  Off: scatter-adds may use atomics / unordered lists.  On: every sum is taken by one owner in
// ascending row order; entry points that only have an atomic form return PCOPS_ERR_UNSUPPORTED.
static std::atomic<int> g_deterministic{[] {
    const char *e = getenv("PCOPS_DETERMINISTIC");
    return (e && e[0] == '1') ? 1 : 0;
}()};
extern "C" void pcops_set_deterministic(int on) { g_deterministic.store(on ? 1 : 0); }
extern "C" int pcops_get_deterministic(void) { return g_deterministic.load(); }
