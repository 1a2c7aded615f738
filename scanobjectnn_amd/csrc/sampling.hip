// sampling.hip -- farthest point sampling, gather_point and its gradient, prob_sample for gfx950.
//
// Replaces sampling/tf_sampling_g.cu:105-192 + the launchers at :203-211 of the
// reference (behaviour only; the design is CDNA4-first):
//   * FPS: one workgroup per cloud, the cloud's xyz AND the running min-distance
//     live in VGPRs for the whole kernel (P points per lane), the picked point is
//     broadcast from an LDS copy, the argmax is a 64-bit packed-key wave reduction
//     (DPP/bpermute) + one LDS slot per wave, ONE barrier per round (double-buffered
//     slots).  No global scratch (`temp` of the reference is unused).
//   * tie rule of the reference's 512-thread tree (smaller k mod 512, then smaller k)
//     is encoded in the low word of the key, so it holds for any thread count.
// Distances are uncontracted fp32 (file is compiled with -ffp-contract=off).
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ unsigned fps_tiebreak(unsigned k) {
    // larger value == preferred: smaller (k mod 512) first, then smaller k
    return 0xFFFFFFFFu - (((k & 511u) << 22) | (k >> 9));
}
__device__ __forceinline__ int fps_decode(unsigned long long key) {
    const unsigned tb = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu);
    return (int)(((tb & 0x3FFFFFu) << 9) | (tb >> 22));
}

// T threads per cloud, P points per thread (n <= T*P).  LDSPTS: keep a float4 copy of
// the cloud in LDS for the per-round broadcast of the picked point.
template <int T, int P, bool LDSPTS>
__global__ __launch_bounds__(T) void fps_kernel(int n, int m, const float *__restrict__ inp,
                                                int *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NW = T / kWave;
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_raw);  // [2][NW]
    float4 *pts = reinterpret_cast<float4 *>(smem_raw + ((2 * NW * 8 + 15) / 16) * 16);

    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const float *p = inp + (size_t)b * n * 3;
    int *o = out + (size_t)b * m;

    // Instruction count is what a round costs (one wave per SIMD, ~150 instructions per round at 4.5 cycles each next to
    // the barrier): the coordinates sit pair-wise in adjacent registers so that the distance arithmetic is packed fp32
    // (two points per instruction), min(d, md) is one v_med3, and the lane's best key is found as max of the distance
    // words followed by max of the tie-break words of the points that attain it (u32 maxima instead of 64-bit compares
    // with selects).  A slot beyond the cloud keeps md = 0 and tie-break 0: it loses against every real point.
    float px[P], py[P], pz[P], md[P];
    unsigned tb[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t + i * T;
        if (k < n) {
            px[i] = p[k * 3 + 0];
            py[i] = p[k * 3 + 1];
            pz[i] = p[k * 3 + 2];
            if (LDSPTS) pts[k] = make_float4(px[i], py[i], pz[i], 0.f);
            md[i] = 1e38f;
            tb[i] = fps_tiebreak((unsigned)k);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            md[i] = 0.f;
            tb[i] = 0u;
        }
    }
    if (t == 0) o[0] = 0;
    if (LDSPTS) __syncthreads();

    int old = 0;
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if (LDSPTS) {
            const float4 q = pts[old];
            x1 = q.x; y1 = q.y; z1 = q.z;
        } else {
            x1 = p[old * 3 + 0]; y1 = p[old * 3 + 1]; z1 = p[old * 3 + 2];
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        unsigned hi = 0u;
        if (P >= 2) {
            const f32x2 qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
#pragma unroll
            for (int i = 0; i + 1 < P; i += 2) {
                const f32x2 vx = {px[i], px[i + 1]}, vy = {py[i], py[i + 1]}, vz = {pz[i], pz[i + 1]};
                const f32x2 dx = vx - qx, dy = vy - qy, dz = vz - qz;
                const f32x2 d = dx * dx + dy * dy + dz * dz;         // per component ((dx dx + dy dy) + dz dz), uncontracted
                md[i] = __builtin_amdgcn_fmed3f(d.x, md[i], -INFINITY);          // = min(d, md): one instruction
                md[i + 1] = __builtin_amdgcn_fmed3f(d.y, md[i + 1], -INFINITY);
            }
        } else {
            const float dx = px[0] - x1, dy = py[0] - y1, dz = pz[0] - z1;
            md[0] = __builtin_amdgcn_fmed3f(dx * dx + dy * dy + dz * dz, md[0], -INFINITY);
        }
#pragma unroll
        for (int i = 0; i < P; ++i) hi = max(hi, __float_as_uint(md[i]));     // distances are >= 0: bit order = value order
        unsigned lo = 0u;
#pragma unroll
        for (int i = 0; i < P; ++i) lo = max(lo, __float_as_uint(md[i]) == hi ? tb[i] : 0u);
        unsigned long long best = ((unsigned long long)hi << 32) | lo;
        best = wave_max_u64(best);
        if (NW > 1) {
            unsigned long long *s = slots + (j & 1) * NW;
            if ((t & (kWave - 1)) == 0) s[t / kWave] = best;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const unsigned long long v = s[w];
                best = v > best ? v : best;
            }
        }
        old = fps_decode(best);
        if (t == 0) o[j] = old;
    }
}

// Any n: the running min-distance lives in the caller's `temp` scratch (b x n floats) and the cloud is streamed from
// L2 / HBM every round, like the reference does beyond its 3072 LDS-resident points (tf_sampling_g.cu:133-141).
// Same keys, same tie rule; one 1024-thread workgroup per cloud.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, const float *__restrict__ inp,
                                                          float *__restrict__ temp, int *__restrict__ out) {
    __shared__ unsigned long long slots[2][16];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *p = inp + (size_t)b * n * 3;
    float *md = temp + (size_t)b * n;
    int *o = out + (size_t)b * m;
    for (int k = t; k < n; k += 1024) md[k] = 1e38f;
    if (t == 0) o[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = p[(size_t)old * 3 + 0], y1 = p[(size_t)old * 3 + 1], z1 = p[(size_t)old * 3 + 2];
        unsigned long long best = 0ull;
        for (int k = t; k < n; k += 1024) {
            const float dx = p[(size_t)k * 3 + 0] - x1, dy = p[(size_t)k * 3 + 1] - y1, dz = p[(size_t)k * 3 + 2] - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            const float d2 = fminf(d, md[k]);
            md[k] = d2;
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(d2) << 32) | fps_tiebreak((unsigned)k);
            best = key > best ? key : best;
        }
        best = wave_max_u64(best);
        unsigned long long *s = slots[j & 1];
        if ((t & (kWave - 1)) == 0) s[t / kWave] = best;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned long long v = s[w];
            best = v > best ? v : best;
        }
        old = fps_decode(best);
        if (t == 0) o[j] = old;
    }
}

template <int T, int P>
int launch_fps(int b, int n, int m, const float *inp, int *out, hipStream_t st) {
    constexpr int NW = T / kWave;
    const size_t slot_bytes = ((2 * NW * 8 + 15) / 16) * 16;
    const size_t pts_bytes = (size_t)n * 16;
    if (slot_bytes + pts_bytes <= 140 * 1024) {
        if (slot_bytes + pts_bytes > 48 * 1024) {
            // >64 KiB of dynamic LDS needs an explicit opt-in (idempotent, per kernel)
            static const hipError_t once = hipFuncSetAttribute(
                reinterpret_cast<const void *>(&fps_kernel<T, P, true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
            (void)once;
        }
        hipLaunchKernelGGL((fps_kernel<T, P, true>), dim3(b), dim3(T), slot_bytes + pts_bytes, st,
                           n, m, inp, out);
    } else {
        hipLaunchKernelGGL((fps_kernel<T, P, false>), dim3(b), dim3(T), slot_bytes, st, n, m, inp,
                           out);
    }
    return pcops_launch_status();
}

__global__ __launch_bounds__(256) void gather_point_kernel(long long total, int n, int m,
                                                           const float *__restrict__ inp,
                                                           const int *__restrict__ idx,
                                                           float *__restrict__ out) {
    // one thread per output float: total = b*m*3
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long row = e / 3;
        const int l = (int)(e - row * 3);
        const long long bi = row / m;
        const int a = idx[row];
        out[e] = inp[(bi * n + a) * 3 + l];
    }
}

__global__ __launch_bounds__(256) void gather_point_grad_kernel(long long total, int n, int m,
                                                                const float *__restrict__ out_g,
                                                                const int *__restrict__ idx,
                                                                float *__restrict__ inp_g) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long row = e / 3;
        const int l = (int)(e - row * 3);
        const long long bi = row / m;
        const int a = idx[row];
        atomicAdd(&inp_g[(bi * n + a) * 3 + l], out_g[e]);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// prob_sample = row cumsum + lower-bound search (sampling/tf_sampling_g.cu:7-103, launcher :197-200).
// The integer output is read off fp32 partial sums, so the ASSOCIATION of the reference's scan is the contract (stated
// in oracle/pcops_oracle.c, oracle_cumsum): groups of four summed serially, the group totals scanned by an up-sweep /
// down-sweep pair, S[g-1] added to group g, chunks of 8192 offset by a compensated running sum.  What is free is where
// the values live: one workgroup per ROW (the reference grid-strides 32 blocks over the rows), every thread keeps its
// eight groups' four partial sums in registers (the reference's 32 KB `buffer4` does not exist), only the 2048 group
// totals go through LDS (bank-skewed by one slot per 32), rows are read and written as float4 where they are aligned.
constexpr int kScanThreads = 256;
constexpr int kScanChunk = 8192;                         // elements per chunk (fixed by the reference: BlockSize * 4)
constexpr int kScanGroups = kScanChunk / 4;              // group totals per chunk
constexpr int kScanPerThread = kScanGroups / kScanThreads;

__device__ __forceinline__ int scan_slot(int i) { return i + (i >> 5); }

__global__ __launch_bounds__(kScanThreads) void cumsum_kernel(int n, const float *__restrict__ inp,
                                                              float *__restrict__ out) {
    __shared__ float tot[kScanGroups + (kScanGroups >> 5)];
    const int t = threadIdx.x;
    const float *x = inp + (size_t)blockIdx.x * n;
    float *y = out + (size_t)blockIdx.x * n;
    const bool vec = ((n & 3) == 0) && (((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    float runningsum = 0.f, runningsum2 = 0.f;
    for (int j = 0; j < n; j += kScanChunk) {
        const int len = min(n - j, kScanChunk);
        const int n2 = (len + 3) >> 2;
        float v[kScanPerThread][4];
#pragma unroll
        for (int i = 0; i < kScanPerThread; ++i) {
            const int g = t + i * kScanThreads;          // group: elements j + 4g .. j + 4g + 3
            const int k = g * 4;
            if (k + 3 < len) {
                float v1, v2, v3, v4;
                if (vec) {
                    const float4 q = *reinterpret_cast<const float4 *>(x + j + k);
                    v1 = q.x; v2 = q.y; v3 = q.z; v4 = q.w;
                } else {
                    v1 = x[j + k]; v2 = x[j + k + 1]; v3 = x[j + k + 2]; v4 = x[j + k + 3];
                }
                v2 += v1;
                v4 += v3;
                v3 += v2;
                v4 += v2;
                v[i][0] = v1; v[i][1] = v2; v[i][2] = v3; v[i][3] = v4;
                tot[scan_slot(g)] = v4;
            } else if (k < len) {                        // the ragged last group: serial sum of what exists, replicated
                float a = 0.f;
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    if (k + l < len) a += x[j + k + l];
                    v[i][l] = a;
                }
                tot[scan_slot(g)] = a;
            }
        }
        int u = 0;
        for (; (2 << u) <= n2; ++u) {                    // up-sweep: aligned 2^(u+1) blocks, right half += left half
            __syncthreads();
            for (int k = t; k < (n2 >> (u + 1)); k += kScanThreads)
                tot[scan_slot((((k << 1) + 2) << u) - 1)] += tot[scan_slot((((k << 1) + 1) << u) - 1)];
        }
        for (--u; u >= 0; --u) {                         // down-sweep: S[p] = T(2^u block ending at p) + S[p - 2^u]
            __syncthreads();
            for (int k = t; k < ((n2 - (1 << u)) >> (u + 1)); k += kScanThreads)
                tot[scan_slot((((k << 1) + 3) << u) - 1)] += tot[scan_slot((((k << 1) + 2) << u) - 1)];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kScanPerThread; ++i) {
            const int g = t + i * kScanThreads;
            const int k = g * 4;
            if (k >= len) continue;
            float o[4];
            const float p = g ? tot[scan_slot(g - 1)] : 0.f;
#pragma unroll
            for (int l = 0; l < 4; ++l) o[l] = (g ? v[i][l] + p : v[i][l]) + runningsum;
            if (vec) {
                *reinterpret_cast<float4 *>(y + j + k) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    if (k + l < len) y[j + k + l] = o[l];
            }
        }
        const float tt = tot[scan_slot(n2 - 1)] + runningsum2;     // compensated carry into the next chunk (:79-83)
        const float r2 = runningsum + tt;
        runningsum2 = tt - (r2 - runningsum);
        runningsum = r2;
        __syncthreads();
    }
}

// q = r * cumsum[n-1]; descending power-of-two walk to the smallest index whose cumulative value is >= q (:83-103).  The
// row of partial sums was just written by cumsum_kernel and is L2-resident; rows up to kSearchLds floats are staged in
// LDS once per workgroup so that the log2(n) dependent probes of a lane are LDS reads.
constexpr int kSearchLds = 16384;
__global__ __launch_bounds__(256) void binary_search_kernel(int n, int m, int base, const float *__restrict__ dataset,
                                                            const float *__restrict__ query, int *__restrict__ result) {
    __shared__ float row[kSearchLds];
    const float *d = dataset + (size_t)blockIdx.x * n;
    const bool staged = n <= kSearchLds;
    if (staged) {
        for (int k = threadIdx.x; k < n; k += 256) row[k] = d[k];
        __syncthreads();
    }
    const float last = staged ? row[n - 1] : d[n - 1];
    for (int j = blockIdx.y * 256 + threadIdx.x; j < m; j += gridDim.y * 256) {
        const float q = query[(size_t)blockIdx.x * m + j] * last;
        int r = n - 1;
        for (int k = base; k >= 1; k >>= 1)
            if (r >= k && (staged ? row[r - k] : d[r - k]) >= q) r -= k;
        result[(size_t)blockIdx.x * m + j] = r;
    }
}

}  // namespace

constexpr int kFpsRegisterMax = 16384;       // clouds up to this size keep xyz and the min-distance in registers

extern "C" unsigned long long pcops_farthest_point_sample_workspace_bytes(int b, int n) {
    return n > kFpsRegisterMax ? sizeof(float) * (unsigned long long)b * n : 0ull;
}

extern "C" int pcops_farthest_point_sample(int b, int n, int m, const float *inp, float *temp,
                                           int *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    PCOPS_REQUIRE_ARG(m > 0);  // tf_sampling.cpp:99 npoint>0
    if (b == 0) return PCOPS_OK;
    PCOPS_REQUIRE_SHAPE(n >= 1);
    PCOPS_REQUIRE_PTR(inp);
    PCOPS_REQUIRE_PTR(out);
    hipStream_t st = as_stream(stream);
    if (n <= 64) return launch_fps<64, 1>(b, n, m, inp, out, st);
    if (n <= 128) return launch_fps<64, 2>(b, n, m, inp, out, st);
    if (n <= 256) return launch_fps<64, 4>(b, n, m, inp, out, st);
    // (512 / 1024 threads per 2048-point cloud were measured too: 294 / 450 us against 267 us for 256 x 8)
    if (n <= 512) return launch_fps<64, 8>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps<256, 4>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps<256, 8>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps<512, 8>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps<1024, 8>(b, n, m, inp, out, st);
    if (n <= kFpsRegisterMax) return launch_fps<1024, 16>(b, n, m, inp, out, st);
    // any larger cloud: the streamed form (fps_tiebreak packs any non-negative int k: (k mod 512) << 22 | k >> 9)
    PCOPS_REQUIRE_PTR(temp);      // pcops_farthest_point_sample_workspace_bytes(b, n) bytes
    hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(1024), 0, st, n, m, inp, temp, out);
    return pcops_launch_status();
}

extern "C" int pcops_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out,
                                  pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    const long long total = (long long)b * m * 3;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(inp);
    PCOPS_REQUIRE_PTR(idx);
    PCOPS_REQUIRE_PTR(out);
    const unsigned grid = cdiv(total, 256) < 4096u ? cdiv(total, 256) : 4096u;
    hipLaunchKernelGGL(gather_point_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total, n, m,
                       inp, idx, out);
    return pcops_launch_status();
}

extern "C" int pcops_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx,
                                       float *inp_g, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    if ((long long)b * n == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(inp_g);
    if (pcops_get_deterministic()) return PCOPS_ERR_UNSUPPORTED;   // float atomics: pcops_scatter_rows_sorted instead
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, st) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    const long long total = (long long)b * m * 3;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(out_g);
    PCOPS_REQUIRE_PTR(idx);
    const unsigned grid = cdiv(total, 256) < 4096u ? cdiv(total, 256) : 4096u;
    hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid), dim3(256), 0, st, total, n, m, out_g,
                       idx, inp_g);
    return pcops_launch_status();
}

// probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)   sampling/tf_sampling.cpp:65, tf_sampling_g.cu:197-200.
// inp_p (b,n) weights, inp_r (b,m) uniform numbers, temp (b,n) caller scratch that receives the row cumsum (the
// reference's allocate_temp, tf_sampling.cpp:86) -> out (b,m) int32.
extern "C" int pcops_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                                 pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    if (b == 0) return PCOPS_OK;
    PCOPS_REQUIRE_SHAPE(n >= 1);                 // the reference reads dataset[n-1]
    if (b > 65535 * 32) return PCOPS_ERR_UNSUPPORTED;
    PCOPS_REQUIRE_PTR(inp_p);
    PCOPS_REQUIRE_PTR(temp);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(cumsum_kernel, dim3(b), dim3(kScanThreads), 0, st, n, inp_p, temp);
    int rc = pcops_launch_status();
    if (rc != PCOPS_OK || m == 0) return rc;
    PCOPS_REQUIRE_PTR(inp_r);
    PCOPS_REQUIRE_PTR(out);
    int base = 1;
    while (base < n) base <<= 1;
    unsigned gy = cdiv(m, 256);
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(binary_search_kernel, dim3(b, gy), dim3(256), 0, st, n, m, base, temp, inp_r, out);
    return pcops_launch_status();
}
