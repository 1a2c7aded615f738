// common.h -- shared host/device helpers of libpcops (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

constexpr int kWave = 64;  // CDNA wavefront

// 64-bit max across the 64 lanes of a wave (all lanes get the result).
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), off, kWave);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, kWave);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}
