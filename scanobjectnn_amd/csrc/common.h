// common.h -- shared host/device helpers of libpcops (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/pcops.h"

#define PCOPS_REQUIRE_PTR(p) \
    do { if ((p) == nullptr) return PCOPS_ERR_NULL_POINTER; } while (0)
#define PCOPS_REQUIRE_SHAPE(cond) \
    do { if (!(cond)) return PCOPS_ERR_BAD_SHAPE; } while (0)
#define PCOPS_REQUIRE_ARG(cond) \
    do { if (!(cond)) return PCOPS_ERR_BAD_ARGUMENT; } while (0)

static inline int pcops_launch_status() {
    return hipGetLastError() == hipSuccess ? PCOPS_OK : PCOPS_ERR_LAUNCH;
}

static inline hipStream_t as_stream(pcops_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

extern "C" int pcops_get_deterministic(void);     // abi.hip: bit-reproducible backward passes requested

constexpr int kWave = 64;  // CDNA wavefront

// unsigned max across the 64 lanes of a wave, wave-uniform result.  DPP row shifts + the two gfx9 row broadcasts
// (a max-"scan" whose last lane holds the total): 6 VALU instructions and one v_readlane, no LDS round trips.
// (The bpermute butterfly this replaces cost 12 ds_bpermute with ~100 cycles of latency each per 64-bit key --
// most of an FPS round.)  Lanes a step does not reach see 0, the identity of an unsigned max.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    auto step = [&](auto ctrl, auto rmask) {
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, decltype(rmask)::value,
                                                                 0xf, true);
        v = o > v ? o : v;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 -> rows 1, 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// 64-bit max across the 64 lanes of a wave (all lanes get the result): the high words first, then the low words
// of the lanes that hold the winning high word.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)(v & 0xffffffffu);
    const unsigned mh = wave_max_u32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}
