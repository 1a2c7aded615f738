// grouping.hip -- ball query, group_point(+grad), selection-sort top-k for gfx950.
//
// Replaces grouping/tf_grouping_g.cu:3-141 of the reference (behaviour only).
//
// Ball query design (CDNA4-first, not the reference's one-block-per-cloud thread-per-query scan).
// Measured on MI355X: this loop is bound by INSTRUCTION ISSUE -- every VALU or SALU wave-instruction costs a SIMD
// about four cycles, whatever the lane count (rocprofv3 SQ_INSTS_VALU + SQ_INSTS_SALU of three earlier formulations:
// 118 M instructions x 4 cycles / 1024 SIMDs = the kernel time).  So the design minimises instructions per pair test:
//   * the DATASET lives in registers, lane l owning the PL CONSECUTIVE points PL*l .. PL*l + PL-1 (the cloud is
//     staged once per workgroup through LDS -- coalesced global reads, conflict-free 16-byte LDS reads with a
//     3 PL + 4 dword lane stride -- then never touched again);
//   * the QUERY is wave-uniform (read out of a lane with v_readlane): one VALU instruction tests it against 64
//     dataset points (compilers pair two of a lane's points into v_pk_* instructions: 128 per instruction), no
//     divergence, no LDS, no load in the scan;
//   * a hit is ONE BIT: lane l, bit c <=> point PL*l + c.  The reference's "first nsample hits in ascending
//     dataset order" is lane-major, bit-minor order of that bitmap, so the slot of every hit is
//     (exclusive wave scan of the per-lane popcounts, 6 DPP adds) + (rank of the bit inside its lane): the ordering
//     work is ~40 instructions per QUERY instead of ~15 per 64 pair tests;
//   * hits leave as a few scattered stores into ONE idx row (3-5 store instructions per query; the reference wrote
//     64 different rows per store); the padding with the first hit is one coalesced store of the row's tail;
//   * `sqrtf(d2) < r` is replaced by the exactly equivalent `d2 <= T`, T = the largest fp32 with
//     sqrtf(T) < r, computed on the host (sqrtf is monotone and correctly rounded).  The naive
//     d2 < r*r is NOT equivalent;
//   * the multi-radius (MSG) form keeps one bitmap per radius against one d2 in the same pass;
//   * distances are uncontracted fp32: ((dx*dx + dy*dy) + dz*dz), dx = query - dataset
//     (file is compiled with -ffp-contract=off).
// Bound: instruction issue / fp32 VALU, not HBM: the algorithmic bytes are 98 KB per cloud against 1.05 M pair
// tests.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kQbpMaxScales = 4;

struct QbpScale {
    float thresh;  // d2 <= thresh  <=>  max(sqrtf(d2),1e-20f) < radius
    int nsample;
    int *idx;
    int *cnt;
};
template <int NS>
struct QbpArgs {
    QbpScale s[NS];
};

// largest t with max(sqrtf(t), 1e-20f) < radius, or -1 if no t >= 0 qualifies
float qbp_threshold(float radius) {
    if (!(radius > 1e-20f)) return -1.f;  // also NaN
    if (isinf(radius)) return 3.402823466e38f;
    float t = (float)((double)radius * (double)radius);
    if (isinf(t)) t = 3.402823466e38f;
    while (t > 0.f && !(sqrtf(t) < radius)) t = nextafterf(t, -INFINITY);
    while (t < 3.402823466e38f && sqrtf(nextafterf(t, INFINITY)) < radius)
        t = nextafterf(t, INFINITY);
    return sqrtf(t) < radius ? t : -1.f;
}

__device__ __forceinline__ float rl_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside the 16-lane DPP rows, then the two
// row-broadcast steps of gfx9 (the sequence LLVM's own wave scan uses); lanes a step does not reach add 0
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2 and 3
    return v;
}

// NS radii; PL dataset points per lane (a pass covers 64 PL consecutive points); 4 waves per workgroup, all on one
// cloud, every wave owning `qpw` consecutive queries (qpw <= 64: their coordinates and running state sit one per lane)
template <int NS, int PL>
__global__ __launch_bounds__(256) void qbp_kernel(int n, int m, int qpw, QbpArgs<NS> a,
                                                  const float *__restrict__ xyz1,
                                                  const float *__restrict__ xyz2) {
    constexpr int STRIDE = 3 * PL + 4;          // dwords per lane in LDS: 16-byte aligned, conflict-free b128 reads
    constexpr int W = PL > 32 ? 2 : 1;          // 32-bit bitmap words per lane and radius
    __shared__ __attribute__((aligned(16))) float stage[64 * STRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const CloudPart cp = xcd_cloud_part();      // a cloud's workgroups on ONE XCD: the cloud is fetched into one L2 (round 5;
    const int b = cp.cloud;                     // round-robin placement had every XCD stage every cloud: 2.75x the bytes)
    const int q0 = (cp.part * 4 + wave) * qpw;
    const int nq = q0 < m ? min(qpw, m - q0) : 0;      // a wave without queries still helps staging the cloud
    const float *p1 = xyz1 + (size_t)b * n * 3;
    const size_t qbase = (size_t)b * m + q0;

    float qxv = 0.f, qyv = 0.f, qzv = 0.f;     // lane j: query q0 + j
    if (lane < nq) {
        const float *qp = xyz2 + (qbase + lane) * 3;
        qxv = qp[0]; qyv = qp[1]; qzv = qp[2];
    }
    int cntv[NS], firstv[NS];                  // lane j: hits so far / first hit of query q0 + j
#pragma unroll
    for (int s = 0; s < NS; ++s) cntv[s] = firstv[s] = 0;

    for (int t0 = 0; t0 < n; t0 += 64 * PL) {
        // ---- stage 64 PL points: AoS run of 3 PL dwords per lane, coalesced reads
        if (t0) __syncthreads();
        const int tn = min(64 * PL, n - t0);
        for (int e = tid; e < 64 * PL * 3; e += 256) {
            const int ln = e / (3 * PL), r = e - ln * (3 * PL);
            // a lane's run is stored PAIR-WISE, (x0 x1 y0 y1 z0 z1)(x2 x3 ...): two consecutive points then sit in
            // adjacent registers per coordinate, which is what the packed fp32 instructions (v_pk_add / v_pk_mul: two
            // points per instruction) take as their 64-bit operands
            const int c = r / 3, d = r - 3 * c;
            // beyond the cloud: a point whose distance to anything overflows to +inf (inf <= T is false)
            stage[ln * STRIDE + 6 * (c >> 1) + 2 * d + (c & 1)] = e < tn * 3 ? p1[(size_t)t0 * 3 + e] : 3.0e38f;
        }
        __syncthreads();
        float p[3 * PL];                        // pairs of this lane's PL points: x0 x1 y0 y1 z0 z1 | x2 x3 ...
#pragma unroll
        for (int i = 0; i < 3 * PL / 4; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(&stage[lane * STRIDE + 4 * i]);
            p[4 * i + 0] = v.x; p[4 * i + 1] = v.y; p[4 * i + 2] = v.z; p[4 * i + 3] = v.w;
        }
        const bool last_pass = t0 + 64 * PL >= n;
        const int kbase = t0 + PL * lane;       // index of this lane's first point
        for (int j = 0; j < nq; ++j) {
            const float qx = rl_f(qxv, j), qy = rl_f(qyv, j), qz = rl_f(qzv, j);
            int cnt0[NS], first[NS];
            bool full = true;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                cnt0[s] = __builtin_amdgcn_readlane(cntv[s], j);
                first[s] = __builtin_amdgcn_readlane(firstv[s], j);
                full = full && cnt0[s] >= a.s[s].nsample;
            }
            // hit bitmap, filled by SHIFTING the compare result in (v_cmp writes VCC, v_addc_co computes hb + hb + VCC:
            // two instructions per point and radius instead of compare + select + or): after the 32 points of a word,
            // point kbase + 32 w + c sits at bit 31 - c of word w -- ascending index = descending bit, read with clz
            unsigned hb[NS][W];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int w = 0; w < W; ++w) hb[s][w] = 0u;
            if (!full) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
#pragma unroll
                for (int i = 0; i < PL / 2; ++i) {
                    const f32x2 px = {p[6 * i], p[6 * i + 1]}, py = {p[6 * i + 2], p[6 * i + 3]},
                                pz = {p[6 * i + 4], p[6 * i + 5]};
                    const f32x2 dx = qx2 - px, dy = qy2 - py, dz = qz2 - pz;
                    const f32x2 d2 = dx * dx + dy * dy + dz * dz;       // per component ((dx dx + dy dy) + dz dz), uncontracted
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        asm("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                            : "+v"(hb[s][(2 * i) / 32]) : "v"(d2.x), "s"(a.s[s].thresh) : "vcc");
                        asm("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                            : "+v"(hb[s][(2 * i) / 32]) : "v"(d2.y), "s"(a.s[s].thresh) : "vcc");
                    }
                }
                if (PL < 32) {                  // a partly filled word: move point 0 to bit 31
#pragma unroll
                    for (int s = 0; s < NS; ++s) hb[s][0] <<= (32 - (PL < 32 ? PL : 0)) & 31;
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ns = a.s[s].nsample;
                int *row = a.s[s].idx + (qbase + j) * ns;
                int cnt = cnt0[s];
                if (cnt0[s] < ns) {                                       // wave-uniform
                    const int mine = __popc(hb[s][0]) + (W > 1 ? __popc(hb[s][W - 1]) : 0);
                    const int incl = wave_incl_scan(mine);
                    const int total = __builtin_amdgcn_readlane(incl, 63);
                    if (total > 0) {                                      // wave-uniform
                        if (cnt0[s] == 0) {       // first hit of the query: lowest lane with a bit, its lowest bit
                            const int fl = (int)__builtin_ctzll(__ballot(mine > 0));
                            const int myfirst = kbase + (hb[s][0] ? __builtin_clz(hb[s][0])
                                                                  : 32 + __builtin_clz(hb[s][W - 1] | 1u));
                            first[s] = __builtin_amdgcn_readlane(myfirst, fl);
                        }
                        int pos = cnt0[s] + incl - mine;                  // slot of this lane's first hit
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            unsigned bits = hb[s][w];
                            while (bits != 0u && pos < ns) {              // per lane: 0-3 hits, rarely more
                                const int c = __builtin_clz(bits);
                                row[pos] = kbase + 32 * w + c;
                                bits &= ~(0x80000000u >> c);
                                ++pos;
                            }
                        }
                        cnt = cnt0[s] + total;
                    }
                    cntv[s] = lane == j ? cnt : cntv[s];
                    firstv[s] = lane == j ? first[s] : firstv[s];
                }
                if (last_pass) {
                    // pad the row with the first hit (rows with no hit at all: 0, the build's definition of what the
                    // reference leaves unwritten) and publish the count
                    const int c = cnt < ns ? cnt : ns;
                    for (int l = c + lane; l < ns; l += 64) row[l] = first[s];
                    if (a.s[s].cnt && lane == 0) a.s[s].cnt[qbase + j] = c;
                }
            }
        }
    }
    if (n <= 0) {   // empty dataset: every row is "no hit"
        for (int j = 0; j < nq; ++j) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                int *row = a.s[s].idx + (qbase + j) * a.s[s].nsample;
                for (int l = lane; l < a.s[s].nsample; l += 64) row[l] = 0;
                if (a.s[s].cnt && lane == 0) a.s[s].cnt[qbase + j] = 0;
            }
        }
    }
}

template <int NS>
int launch_qbp(int b, int n, int m, const QbpArgs<NS> &a, const float *xyz1, const float *xyz2,
               hipStream_t st) {
    // queries per wave: enough waves to fill the chip several times over, but not so few queries per wave that
    // staging the cloud and loading it into registers dominates
    long long qpw = ((long long)b * m + 8191) / 8192;
    qpw = qpw < 8 ? 8 : (qpw > 64 ? 64 : qpw);
    if (qpw > m) qpw = m;
    const dim3 grid(cdiv(m, qpw * 4), b);
#define PCOPS_QBP_P(P_) hipLaunchKernelGGL((qbp_kernel<NS, P_>), grid, dim3(256), 0, st, n, m, (int)qpw, a, xyz1, xyz2)
    if (n <= 512) PCOPS_QBP_P(8);              // points per lane: the smallest that covers the cloud in one pass
    else if (n <= 1024) PCOPS_QBP_P(16);
    else if (n <= 2048) PCOPS_QBP_P(32);
    else PCOPS_QBP_P(64);                      // n > 4096: several passes, per-query state carried one per lane
#undef PCOPS_QBP_P
    return pcops_launch_status();
}

// ---------------------------------------------------------------------------------
// group_point: out[row, :] = points[b(row), idx[row], :], row = (b, j, s) flattened.
// VEC = floats per thread access (4 when c % 4 == 0 -> dwordx4 both sides).
template <int VEC>
__global__ __launch_bounds__(256) void group_point_kernel(long long total, int n, int cv, int rows_per_b,
                                                          const float *__restrict__ points,
                                                          const int *__restrict__ idx,
                                                          float *__restrict__ out) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long row = e / cv;
        const int col = (int)(e - row * cv);
        const long long bi = row / rows_per_b;
        const int ii = idx[row];
        const vec_t v = *reinterpret_cast<const vec_t *>(points + ((bi * n + ii) * (long long)cv + col) * VEC);
        *reinterpret_cast<vec_t *>(out + e * VEC) = v;
    }
}

__global__ __launch_bounds__(256) void group_point_grad_kernel(long long total, int n, int c,
                                                               int rows_per_b,
                                                               const float *__restrict__ grad_out,
                                                               const int *__restrict__ idx,
                                                               float *__restrict__ grad_points) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long row = e / c;
        const int col = (int)(e - row * c);
        const long long bi = row / rows_per_b;
        const int ii = idx[row];
        atomicAdd(&grad_points[(bi * n + ii) * (long long)c + col], grad_out[e]);
    }
}

// ---------------------------------------------------------------------------------
// selection_sort: literal restatement of the unstable per-row selection sort (the knn
// branch of sample_and_group; no in-scope model takes it).  One lane per row, rows of the
// (b*m, n) matrix processed in place in the output buffers.
__global__ __launch_bounds__(64) void selection_sort_kernel(long long rows, int n, int k,
                                                            const float *__restrict__ dist,
                                                            int *__restrict__ outi,
                                                            float *__restrict__ out) {
    const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    const float *src = dist + r * n;
    float *v = out + r * n;
    int *ix = outi + r * n;
    for (int s = 0; s < n; ++s) {
        v[s] = src[s];
        ix[s] = s;
    }
    for (int s = 0; s < k && s < n; ++s) {
        int mn = s;
        float mv = v[s];
        for (int t = s + 1; t < n; ++t) {
            const float x = v[t];
            if (x < mv) {
                mv = x;
                mn = t;
            }
        }
        if (mn != s) {
            v[mn] = v[s];
            v[s] = mv;
            const int ti = ix[mn];
            ix[mn] = ix[s];
            ix[s] = ti;
        }
    }
}

// The same sort with one WAVE per row and the row in LDS (n <= 8192): every step's minimum search is n / 64 LDS reads per
// lane + a 6-step butterfly on (value, index) instead of n global loads by one lane.  The reference's scan keeps the
// FIRST minimum that is strictly smaller than v[s] (`x < mv`, selection_sort.cpp:29-36): lanes keep their first minimum
// (ascending t, strict <), the butterfly prefers the smaller index on equal values, and the winner only replaces
// position s if it is strictly smaller than v[s] -- NaNs never win and a NaN at s never moves, as in the scalar loop.
//
// FUSED (knn_point, tf_grouping.py:49-74): the row is not read but COMPUTED -- dist[t] = sum_l (xyz1[b,t,l] - xyz2[b,j,l])^2,
// l ascending, uncontracted (this file is built with -ffp-contract=off; the order TF's reduce_sum leaves open is fixed as in
// oracle_knn_point) -- and only the first k columns leave, as (val, idx) of shape (b, m, k): neither the tiled
// (b, m, n, c) difference tensor nor the (b, m, n) distance matrix of the reference graph ever exists.
template <bool FUSED>
__global__ __launch_bounds__(256) void selection_sort_wave_kernel(long long rows, int n, int k, int waves,
                                                                  const float *__restrict__ dist,
                                                                  int *__restrict__ outi, float *__restrict__ out,
                                                                  int c, int m, const float *__restrict__ xyz1,
                                                                  const float *__restrict__ xyz2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * waves + wave;
    if (wave >= waves || r >= rows) return;
    float *v = sm + (size_t)wave * 2 * n;
    int *ix = reinterpret_cast<int *>(v + n);
    if (FUSED) {
        const float *q = xyz2 + r * c;                               // query row (b, j) = r
        const float *p = xyz1 + (r / m) * (long long)n * c;          // its cloud
        for (int t = lane; t < n; t += 64) {
            float acc = 0.f;
            for (int l = 0; l < c; ++l) {
                const float d = p[(long long)t * c + l] - q[l];
                acc = acc + d * d;
            }
            v[t] = acc;
            ix[t] = t;
        }
    } else {
        const float *src = dist + r * n;
        for (int t = lane; t < n; t += 64) {
            v[t] = src[t];
            ix[t] = t;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int kk = k < n ? k : n;
    for (int s = 0; s < kk; ++s) {
        float mv = INFINITY;
        int mn = 0x7fffffff;
        for (int t = s + 1 + lane; t < n; t += 64) {
            const float x = v[t];
            if (x < mv) { mv = x; mn = t; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(mv, off, 64);
            const int on = __shfl_xor(mn, off, 64);
            if (ov < mv || (ov == mv && on < mn)) { mv = ov; mn = on; }
        }
        const float vs = v[s];
        if (mn != 0x7fffffff && mv < vs) {          // wave-uniform
            if (lane == 0) {
                v[mn] = vs;
                v[s] = mv;
                const int ti = ix[mn];
                ix[mn] = ix[s];
                ix[s] = ti;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (FUSED) {
        for (int t = lane; t < kk; t += 64) {
            out[r * k + t] = v[t];
            outi[r * k + t] = ix[t];
        }
        return;
    }
    float *vo = out + r * n;
    int *io = outi + r * n;
    for (int t = lane; t < n; t += 64) {
        vo[t] = v[t];
        io[t] = ix[t];
    }
}

// knn_point's distance matrix on its own (clouds too large for the fused kernel's LDS row): dist (b, m, n), same
// arithmetic order as above; thread = one (query, candidate) pair, candidates fastest (coalesced stores)
__global__ __launch_bounds__(256) void knn_point_dist_kernel(long long total, int n, int c, int m,
                                                             const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, float *__restrict__ dist) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / n;
        const int t = (int)(e - r * n);
        const float *q = xyz2 + r * c;
        const float *p = xyz1 + ((r / m) * n + t) * (long long)c;
        float acc = 0.f;
        for (int l = 0; l < c; ++l) {
            const float d = p[l] - q[l];
            acc = acc + d * d;
        }
        dist[e] = acc;
    }
}

}  // namespace

extern "C" int pcops_query_ball_point_multi(int b, int n, int m, int nscale, const float *radius,
                                            const int *nsample, const float *xyz1,
                                            const float *xyz2, int *const *idx, int *const *pts_cnt,
                                            pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    PCOPS_REQUIRE_ARG(nscale >= 1 && nscale <= kQbpMaxScales);
    PCOPS_REQUIRE_PTR(radius);
    PCOPS_REQUIRE_PTR(nsample);
    PCOPS_REQUIRE_PTR(idx);
    for (int s = 0; s < nscale; ++s) {
        PCOPS_REQUIRE_ARG(radius[s] > 0.f);  // tf_grouping.cpp:71
        PCOPS_REQUIRE_ARG(nsample[s] > 0);   // tf_grouping.cpp:74
    }
    if ((long long)b * m == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(xyz2);
    if (n > 0) PCOPS_REQUIRE_PTR(xyz1);
    for (int s = 0; s < nscale; ++s) PCOPS_REQUIRE_PTR(idx[s]);
    hipStream_t st = as_stream(stream);
#define PCOPS_QBP_CASE(NS)                                                         \
    case NS: {                                                                     \
        QbpArgs<NS> a;                                                             \
        for (int s = 0; s < NS; ++s)                                               \
            a.s[s] = QbpScale{qbp_threshold(radius[s]), nsample[s], idx[s],       \
                              pts_cnt ? pts_cnt[s] : nullptr};                     \
        return launch_qbp<NS>(b, n, m, a, xyz1, xyz2, st);                         \
    }
    switch (nscale) {
        PCOPS_QBP_CASE(1)
        PCOPS_QBP_CASE(2)
        PCOPS_QBP_CASE(3)
        PCOPS_QBP_CASE(4)
    }
#undef PCOPS_QBP_CASE
    return PCOPS_ERR_BAD_ARGUMENT;
}

extern "C" int pcops_query_ball_point(int b, int n, int m, float radius, int nsample,
                                      const float *xyz1, const float *xyz2, int *idx, int *pts_cnt,
                                      pcops_stream_t stream) {
    int *idxs[1] = {idx};
    int *cnts[1] = {pts_cnt};
    return pcops_query_ball_point_multi(b, n, m, 1, &radius, &nsample, xyz1, xyz2, idxs, cnts,
                                        stream);
}

extern "C" int pcops_group_point(int b, int n, int c, int m, int nsample, const float *points,
                                 const int *idx, float *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0);
    const long long rows = (long long)b * m * nsample;
    if (rows * c == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(points);
    PCOPS_REQUIRE_PTR(idx);
    PCOPS_REQUIRE_PTR(out);
    hipStream_t st = as_stream(stream);
    const bool v4 = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int cv = v4 ? c / 4 : c;
    const long long total = rows * cv;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    if (v4)
        hipLaunchKernelGGL((group_point_kernel<4>), dim3(grid), dim3(256), 0, st, total, n, cv,
                           m * nsample, points, idx, out);
    else
        hipLaunchKernelGGL((group_point_kernel<1>), dim3(grid), dim3(256), 0, st, total, n, cv,
                           m * nsample, points, idx, out);
    return pcops_launch_status();
}

extern "C" int pcops_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                                      const int *idx, float *grad_points, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0);
    if ((long long)b * n * c == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_points);
    if (pcops_get_deterministic()) return PCOPS_ERR_UNSUPPORTED;   // float atomics: pcops_scatter_rows_sorted instead
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    const long long total = (long long)b * m * nsample * c;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_out);
    PCOPS_REQUIRE_PTR(idx);
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid), dim3(256), 0, st, total, n, c,
                       m * nsample, grad_out, idx, grad_points);
    return pcops_launch_status();
}

extern "C" int pcops_selection_sort(int b, int n, int m, int k, const float *dist, int *outi,
                                    float *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    PCOPS_REQUIRE_ARG(k > 0);  // tf_grouping.cpp:113
    const long long rows = (long long)b * m;
    if (rows * n == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(dist);
    PCOPS_REQUIRE_PTR(outi);
    PCOPS_REQUIRE_PTR(out);
    if (n <= 8192) {        // a wave per row, the row in LDS
        int waves = (int)(65536 / ((size_t)8 * n));
        waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
        const size_t lds = (size_t)waves * 8 * n;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(selection_sort_wave_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        hipLaunchKernelGGL(selection_sort_wave_kernel<false>, dim3(cdiv(rows, waves)), dim3(256), lds, as_stream(stream),
                           rows, n, k, waves, dist, outi, out, 0, 1, (const float *)nullptr, (const float *)nullptr);
        return pcops_launch_status();
    }
    hipLaunchKernelGGL(selection_sort_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, as_stream(stream),
                       rows, n, k, dist, outi, out);
    return pcops_launch_status();
}

// knn_point (pointnet2/tf_ops/grouping/tf_grouping.py:49-74): squared distances + the literal selection sort + the first
// k columns, in one kernel (a wave per query, its distance row in LDS).  n <= 8192 (PCOPS_ERR_UNSUPPORTED beyond: build
// the matrix with pcops_knn_point_dist and sort it with pcops_selection_sort).
extern "C" int pcops_knn_point_supported(int n) { return n >= 1 && n <= 8192 ? 1 : 0; }

extern "C" int pcops_knn_point(int b, int n, int c, int m, int k, const float *xyz1, const float *xyz2, float *val,
                               int *idx, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0 && c >= 1);
    PCOPS_REQUIRE_ARG(k > 0);  // tf_grouping.cpp:113
    PCOPS_REQUIRE_SHAPE(k <= n || (long long)b * m == 0);
    const long long rows = (long long)b * m;
    if (rows * n == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(xyz1); PCOPS_REQUIRE_PTR(xyz2); PCOPS_REQUIRE_PTR(val); PCOPS_REQUIRE_PTR(idx);
    if (!pcops_knn_point_supported(n)) return PCOPS_ERR_UNSUPPORTED;
    int waves = (int)(65536 / ((size_t)8 * n));
    waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
    const size_t lds = (size_t)waves * 8 * n;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(selection_sort_wave_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    hipLaunchKernelGGL(selection_sort_wave_kernel<true>, dim3(cdiv(rows, waves)), dim3(256), lds, as_stream(stream), rows, n,
                       k, waves, (const float *)nullptr, idx, val, c, m, xyz1, xyz2);
    return pcops_launch_status();
}

extern "C" int pcops_knn_point_dist(int b, int n, int c, int m, const float *xyz1, const float *xyz2, float *dist,
                                    pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0 && c >= 1);
    const long long total = (long long)b * m * n;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(xyz1); PCOPS_REQUIRE_PTR(xyz2); PCOPS_REQUIRE_PTR(dist);
    const unsigned grid = cdiv(total, 256) < 65536u ? cdiv(total, 256) : 65536u;
    hipLaunchKernelGGL(knn_point_dist_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total, n, c, m, xyz1, xyz2, dist);
    return pcops_launch_status();
}
