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
// prob_sample = row cumsum + lower-bound search (reference: sampling/tf_sampling_g.cu:7-103, launcher :197-200).
// The integer output is read off fp32 partial sums, so the ASSOCIATION of the reference's scan is the contract; it is
// stated independently in tests/test_prob_sample_cpu.py and in oracle/pcops_oracle.c.  In this file's own words:
//   * a chunk is 8 192 elements = 2 048 QUADS; inside a quad  c0 = x0, c1 = x1 + x0, c2 = x2 + c1, c3 = (x3 + x2) + c1;
//   * the quad totals t[0 .. nq) form a FENWICK TREE: node(e) = sum of the aligned block of lowbit(e + 1) totals ending at e,
//     built bottom-up as  node = (its upper half's node) + (its lower half's node)  -- only blocks that lie inside nq exist;
//   * the inclusive prefix of the totals is the tree's query, taken from the LARGEST block down:
//         pre(e) = node(e) + pre(e - lowbit(e + 1)),   pre(-1) absent (no addition);
//   * quad g > 0 adds pre(g - 1) to its four partial sums, then every element adds the carry of the chunks before;
//   * the carry is kept as a (sum, compensation) pair across chunks.
// Where the values live is this kernel's own: one workgroup per row, a lane owns EIGHT CONSECUTIVE quads, so the three
// lowest tree levels are register arithmetic, the next six are wave shuffles of the lanes' top nodes, the last two combine
// the four waves' tops; the query reads at most eight nodes of other lanes from a 256-entry LDS table.  No 32 KB element
// buffer, no workgroup barrier per tree level.
constexpr int kScanThreads = 256;
constexpr int kQuadsPerLane = 8;
constexpr int kScanChunk = kScanThreads * kQuadsPerLane * 4;      // 8 192 elements (fixed by the contract)

// prefix of the lanes' top nodes in front of lane `lane_ix` (1 .. 256): Fenwick query over node[0 .. 255], largest block first
__device__ __forceinline__ float tops_before(const float *node, int lane_ix, bool &any) {
    float acc = 0.f;
    any = false;
    for (int bit = 8; bit >= 0; --bit) {
        if (!((lane_ix >> bit) & 1)) continue;
        const int at = (lane_ix & ~((1 << bit) - 1)) - 1;          // the node that closes this block
        acc = any ? node[at] + acc : node[at];
        any = true;
    }
    return acc;
}

__global__ __launch_bounds__(kScanThreads) void cumsum_kernel(int n, const float *__restrict__ inp,
                                                              float *__restrict__ out) {
    __shared__ float lane_node[kScanThreads];            // Fenwick nodes over the lanes' top nodes
    __shared__ float wave_top[kScanThreads / 64];
    __shared__ float chunk_total;
    const int lane_ix = threadIdx.x, wlane = lane_ix & 63, wv = lane_ix >> 6;
    const float *src = inp + (size_t)blockIdx.x * n;
    float *dst = out + (size_t)blockIdx.x * n;
    const bool vec = ((n & 3) == 0) && (((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    float carry = 0.f, carry_lost = 0.f;                 // running sum of the chunks so far and what its rounding dropped
    for (int c0 = 0; c0 < n; c0 += kScanChunk) {
        const int len = min(n - c0, kScanChunk);
        const int nq = (len + 3) >> 2;                   // quads of this chunk
        const int q0 = lane_ix * kQuadsPerLane;          // this lane's first quad
        float part[kQuadsPerLane][4];                    // partial sums inside each quad
        float nd[kQuadsPerLane];                         // tree nodes of the lane's quads (level <= 3)
#pragma unroll
        for (int i = 0; i < kQuadsPerLane; ++i) {
            const int e0 = (q0 + i) * 4;                 // first element of the quad inside the chunk
            nd[i] = 0.f;
            if (e0 + 3 < len) {
                float4 x;
                if (vec) x = *reinterpret_cast<const float4 *>(src + c0 + e0);
                else x = make_float4(src[c0 + e0], src[c0 + e0 + 1], src[c0 + e0 + 2], src[c0 + e0 + 3]);
                const float lo = x.y + x.x, hi = x.w + x.z;
                part[i][0] = x.x; part[i][1] = lo; part[i][2] = x.z + lo; part[i][3] = hi + lo;
                nd[i] = part[i][3];
            } else if (e0 < len) {                       // the ragged last quad: running sum of what exists, then held
                float run = 0.f;
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    if (e0 + l < len) run += src[c0 + e0 + l];
                    part[i][l] = run;
                }
                nd[i] = run;
            }
        }
        // tree levels 1..3 in registers: a block exists only if it ends inside nq
        const int have = nq - q0;                        // quads of this lane that exist
        if (have > 1) nd[1] += nd[0];
        if (have > 3) nd[3] += nd[2];
        if (have > 5) nd[5] += nd[4];
        if (have > 7) nd[7] += nd[6];
        if (have > 3) nd[3] += nd[1];
        if (have > 7) nd[7] += nd[5];
        if (have > 7) nd[7] += nd[3];
        // levels 4..9: the lanes' top nodes inside a wave (a lane whose block is incomplete never feeds a complete one)
        float top = nd[7];
        const bool whole = have > 7;
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) {
            const float below = __shfl_up(top, 1 << lv, 64);
            if (((wlane + 1) & ((2 << lv) - 1)) == 0 && whole) top += below;
        }
        // levels 10, 11: the four waves' tops -- W1 = w1 + w0, W3 = (w3 + w2) + W1
        if (wlane == 63) wave_top[wv] = top;
        __syncthreads();
        if (wlane == 63 && whole) {
            if (wv == 1) top += wave_top[0];
            if (wv == 3) top = (top + wave_top[2]) + (wave_top[1] + wave_top[0]);
        }
        lane_node[lane_ix] = top;
        __syncthreads();
        // query: prefix of the totals in front of this lane's quads, then through them
        bool any;
        float before = tops_before(lane_node, lane_ix, any);
        // pre[i] = inclusive prefix at quad q0 + i (i < 7 suffices: quad i + 1 needs it); largest block first
        float pre[kQuadsPerLane];
        pre[0] = any ? nd[0] + before : nd[0];
        pre[1] = any ? nd[1] + before : nd[1];
        pre[2] = nd[2] + pre[1];
        pre[3] = any ? nd[3] + before : nd[3];
        pre[4] = nd[4] + pre[3];
        pre[5] = nd[5] + pre[3];
        pre[6] = nd[6] + pre[5];
        pre[7] = 0.f;                                    // (needed only as the chunk total: below)
        if (lane_ix == (nq - 1) / kQuadsPerLane) {       // the lane that owns the chunk's last quad publishes the total
            const int li = (nq - 1) % kQuadsPerLane;
            float tot;
            if (li == 7) {
                bool any2;
                tot = tops_before(lane_node, lane_ix + 1, any2);
            } else {
                tot = pre[0];
#pragma unroll
                for (int i = 1; i < 7; ++i)
                    if (li == i) tot = pre[i];
            }
            chunk_total = tot;
        }
#pragma unroll
        for (int i = 0; i < kQuadsPerLane; ++i) {
            const int e0 = (q0 + i) * 4;
            if (e0 >= len) continue;
            const bool first = (q0 + i) == 0;            // the chunk's first quad adds no prefix
            const float add = i == 0 ? before : pre[i - 1];
            float o[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) o[l] = (first ? part[i][l] : part[i][l] + add) + carry;
            if (vec) {
                *reinterpret_cast<float4 *>(dst + c0 + e0) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    if (e0 + l < len) dst[c0 + e0 + l] = o[l];
            }
        }
        __syncthreads();
        // carry into the next chunk: the total plus what the previous update's rounding dropped, re-split
        const float inc = chunk_total + carry_lost;
        const float grown = carry + inc;
        carry_lost = inc - (grown - carry);
        carry = grown;
        __syncthreads();
    }
}

// sample index = the smallest position whose cumulative weight reaches  u * (row total)  (reference :83-103: a descent over
// powers of two from the row's end).  The row of partial sums was just written by cumsum_kernel and is L2-resident; rows up
// to kSearchLds floats are staged in LDS once per workgroup so that the log2(n) dependent probes of a lane are LDS reads.
constexpr int kSearchLds = 16384;
__global__ __launch_bounds__(256) void binary_search_kernel(int n, int m, int top_step, const float *__restrict__ cum,
                                                            const float *__restrict__ uniform, int *__restrict__ picked) {
    __shared__ float staged_row[kSearchLds];
    const float *grow = cum + (size_t)blockIdx.x * n;
    const bool in_lds = n <= kSearchLds;
    if (in_lds) {
        for (int e = threadIdx.x; e < n; e += 256) staged_row[e] = grow[e];
        __syncthreads();
    }
    const float *row = in_lds ? staged_row : grow;
    const float total = row[n - 1];
    for (int s = blockIdx.y * 256 + threadIdx.x; s < m; s += gridDim.y * 256) {
        const float target = uniform[(size_t)blockIdx.x * m + s] * total;
        int pos = n - 1;                                 // always a position whose cumulative weight reaches the target
        for (int step = top_step; step > 0; step >>= 1) {
            const int cand = pos - step;
            if (cand >= 0 && row[cand] >= target) pos = cand;
        }
        picked[(size_t)blockIdx.x * m + s] = pos;
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
