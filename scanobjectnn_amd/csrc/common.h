// common.h -- shared host/device helpers of libpcops (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/pcops.h"

#define PCOPS_REQUIRE_PTR(p) \
    do { if ((p) == nullptr) return PCOPS_ERR_NULL_POINTER; } while (0)
#define PCOPS_REQUIRE_SHAPE(cond) \
    do { if (!(cond)) return PCOPS_ERR_BAD_SHAPE; } while (0)
#define PCOPS_REQUIRE_ARG(cond) \
    do { if (!(cond)) return PCOPS_ERR_BAD_ARGUMENT; } while (0)

static inline int pcops_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PCOPS_OK;
    static const bool verbose = getenv("PCOPS_DEBUG_LAUNCH") != nullptr;      // what HIP said, on stderr
    if (verbose) fprintf(stderr, "libpcops: launch failed: %s (%s)\n", hipGetErrorString(e), hipGetErrorName(e));
    return PCOPS_ERR_LAUNCH;
}

static inline hipStream_t as_stream(pcops_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

extern "C" int pcops_get_deterministic(void);     // abi.hip: bit-reproducible backward passes requested
extern "C" int pcops_get_option(int option);      // abi.hip: the arithmetic options of pcops.h (read at every call)
void pcops_note_pipe(int pipe);                   // abi.hip: what pcops_last_launch_pipe() reports (0 fp32, 1 split bf16, 2 both)

constexpr int kWave = 64;  // CDNA wavefront

// Asymmetric STATIC wave priority (round 4; MI355X_MICROARCH.md "Two waves per SIMD").  Two co-resident waves that run
// the same [MFMA chain | VALU phase] loop at equal priority settle in LOCKSTEP: both contend for the matrix pipe (each
// chain takes twice as long), then both contend for the VALU issue slots -- time = MFMA + VALU, nothing overlaps,
// exactly the "matrix-pipe busy + 4.5 x other instructions" the round-2/3 counters showed.  With the odd wave slot of
// every SIMD at priority 1 its chain runs unimpeded while the partner waits, and from then on one wave's VALU phase
// sits under the other's MFMA chain.  PCOPS_WAVE_PRIO (compile time, A/B builds through PCOPS_LIB): bit 0 = the
// single-role kernels (gemm_ws, knn_mfma) by hardware wave slot, bit 1 = the consumer (MFMA) waves of the producer /
// consumer kernels.
#ifndef PCOPS_WAVE_PRIO
#define PCOPS_WAVE_PRIO 0
#endif
__device__ __forceinline__ void wave_prio_stagger() {
#if PCOPS_WAVE_PRIO & 1
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID.wave_id: slot on the SIMD
    if (slot & 1) __builtin_amdgcn_s_setprio(1);
#endif
}
__device__ __forceinline__ void wave_prio_consumer(bool consumer) {
#if PCOPS_WAVE_PRIO & 2
    if (consumer) __builtin_amdgcn_s_setprio(1);
#else
    (void)consumer;
#endif
}

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

// ---- XCD-aware numbering of workgroups (round 5).  Workgroup `bid` (linear dispatch order) of `nwg` runs on XCD bid % 8
// (observed, MI355X_MICROARCH.md "Workgroup dispatch" -- a speed assumption only, nothing may depend on it): renumber so
// that every XCD owns a CONTIGUOUS range of the virtual ids (bijective for any nwg).  Kernels that deal several
// workgroups to one cloud use it to keep a cloud's rows in ONE L2 instead of fetching them into all eight.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
// a (parts-per-cloud, clouds) grid: the (part, cloud) this workgroup takes, all parts of a cloud on one XCD
struct CloudPart { int part, cloud; };
__device__ __forceinline__ CloudPart xcd_cloud_part() {
    const unsigned vb = xcd_contiguous(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const unsigned cloud = vb / gridDim.x;
    return CloudPart{(int)(vb - cloud * gridDim.x), (int)cloud};
}
