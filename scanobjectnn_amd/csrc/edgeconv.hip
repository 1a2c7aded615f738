// edgeconv.hip -- round 5: the gather / scatter family of the DGCNN path rebuilt around three measured facts
// (profiles/r04_pmc_insts_dgcnn.txt, profiles/r05_pmc_traffic_detail_dgcnn_before.json):
//   * the round-2..4 kernels (gather.hip: edge_pool_fwd, edge_pool_bwd_dense, sa_gather_fwd, sa_scatter_*) were bound by
//     INSTRUCTION ISSUE, not by HBM: 13 VALU instructions per gathered element where the arithmetic needs 4.5 -- 64-bit
//     address products per neighbour, an unsigned division per list entry and lane, every lane of a group re-loading the
//     group's indices, both directions of the extremum evaluated and selected per element;
//   * they moved 4.1x (forward) and 4x (dense backward) their algorithmic bytes: consecutive workgroups of one cloud were
//     dealt round-robin to the 8 XCDs, so every XCD's L2 re-fetched every cloud's Q / Ctr rows;
//   * the T-Net's scatter read Y1 (2.7 GB) only to form  sum q.Y1  -- which is  q (cnt_i Q[i] + sum Ctr[g])  exactly as
//     in the EdgeConv layer, i.e. a walk over L2-resident rows.
// What is here:
//   ec_fwd_kernel<MODE, UP>   y[g,s,:] = Q[idx[g,s],:] + Ctr[g,:]  in 64-channel slices, a 16-lane set per group:
//                             MODE 0 the pooled EdgeConv layer (group sums, extremum + first arg, shifted moments),
//                             MODE 1 the stored first layer of a gather stack (rows of Y + shifted moments).
//                             The group's byte offsets are staged once per workgroup in LDS (pre-multiplied), a gather is
//                             one 32-bit VALU add + one bounds-checked buffer load, the statistics are packed fp32.
//   ec_csr_build_kernel       inverse index per cloud: packed (group << 8 | slot) entries sorted by source point.
//   ec_walk_kernel<HAS_G>     owner walk of the inverse index: sum of the Ctr rows (and, T-Net, of the G rows) that
//                             reference a point; entries beyond a list read out of bounds (zeros), no per-entry branch.
//   ec_tnet_ctr_kernel        per-group output of the T-Net's scatter:  dCtr = p sum_s G + q (sum_s Q[idx] + k Ctr) + k t.
// Work is numbered (cloud, slice, chunk) and workgroups are mapped to it XCD-contiguously: a cloud's rows are fetched by
// ONE L2.  Dispatch placement is a speed assumption only (MI355X_MICROARCH.md "Workgroup dispatch"); nothing is shared
// between workgroups.
#include "common.h"
#include "edgeconv.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int ec_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ec_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ec_load4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    const ec_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ unsigned ec_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
}


constexpr int kGB = 64;          // groups (forward) / points (walk) per workgroup
constexpr int kSets = 16;        // 16-lane sets per 256-thread workgroup: one group / point each, four rounds

// ldq / ldc (and lddq / lddc below): row strides in floats of Q / Ctr (dQ / dCtr) -- C for dense tensors, 2 C when the
// two are the column halves of ONE (b, n, 2 C) product [Q | Ctr] (the *_ld entry points: one GEMM, one gradient)
struct FwdArgs {
    int b, n, m, S, C;
    int ldq, ldc;
    const float *Q, *Ctr;
    const int *idx;
    const float *gamma;          // MODE 0: direction of the extremum
    float *SQ, *qsel;            // MODE 0
    unsigned char *arg;          // MODE 0
    float *Y;                    // MODE 1
    float *stats;                // [b * m / 64][2][C] or NULL
    const float *pivot;          // shifted moments around this vector (or NULL: 0)
    int nt;                      // MODE 1: Y leaves with non-temporal stores (256 MB and more)
};

// ---- forward ------------------------------------------------------------------------------------------------------
// MODE 0, per (group, channel):  sq = sum_s (q - qz), sq2 = sum_s (q - qz)^2 around the sample row qz = Q[0,0,:] (the
// formulas of gather.hip edge_pool_fwd_kernel, kept so that the statistics are the same numbers), ex = max_s sg.q and the
// FIRST s attaining it (sg = -1 where gamma < 0: BN + ReLU is then decreasing and the pooled row is the minimum).
// MODE 1, per row: y = Ctr + q stored, d = y - pivot, s1 += d, s2 += d^2 (gather.hip sa_gather_fwd_kernel).
template <int MODE, bool UP>
__device__ __forceinline__ void ec_fwd_groups(const FwdArgs &a, const unsigned *offs, long long g0, int set, int quad,
                                              unsigned tq, __amdgpu_buffer_rsrc_t rq, int ch, f2 (&s1)[2], f2 (&s2)[2]) {
    const int S = a.S, C = a.C;
    const float kf = (float)S;
    f2 sg[2] = {{1.f, 1.f}, {1.f, 1.f}};
    f2 qz[2] = {{0.f, 0.f}, {0.f, 0.f}}, cz[2] = {{0.f, 0.f}, {0.f, 0.f}}, pv[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (MODE == 0) {
        if (!UP) {
            const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + ch);
            sg[0] = f2{ga.x < 0.f ? -1.f : 1.f, ga.y < 0.f ? -1.f : 1.f};
            sg[1] = f2{ga.z < 0.f ? -1.f : 1.f, ga.w < 0.f ? -1.f : 1.f};
        }
        if (a.stats) {
            const float4 q0 = *reinterpret_cast<const float4 *>(a.Q + ch);
            float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.pivot) p4 = *reinterpret_cast<const float4 *>(a.pivot + ch);
            qz[0] = f2{q0.x, q0.y}; qz[1] = f2{q0.z, q0.w};
            cz[0] = f2{q0.x - p4.x, q0.y - p4.y}; cz[1] = f2{q0.z - p4.z, q0.w - p4.w};
        }
    } else if (a.stats && a.pivot) {
        const float4 p4 = *reinterpret_cast<const float4 *>(a.pivot + ch);
        pv[0] = f2{p4.x, p4.y}; pv[1] = f2{p4.z, p4.w};
    }
#pragma unroll 1
    for (int r = 0; r < kGB / kSets; ++r) {
        const int gl = set + kSets * r;
        const long long g = g0 + gl;
        const float4 c4 = *reinterpret_cast<const float4 *>(a.Ctr + g * a.ldc + ch);
        const f2 ct[2] = {{c4.x, c4.y}, {c4.z, c4.w}};
        f2 sq[2] = {{0.f, 0.f}, {0.f, 0.f}}, sq2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        float ex[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int ea[4] = {0, 0, 0, 0};
        float *yrow = MODE == 1 ? a.Y + g * S * (long long)C + ch : nullptr;
        const bool ynt = MODE == 1 && a.nt != 0;
        const unsigned *og = offs + gl * S;
        auto take = [&](const float4 &q, int s) {
            const f2 qa = f2{q.x, q.y}, qb = f2{q.z, q.w};
            if (MODE == 0) {
                const f2 da = qa - qz[0], db = qb - qz[1];
                sq[0] += da; sq[1] += db;
                sq2[0] = __builtin_elementwise_fma(da, da, sq2[0]);
                sq2[1] = __builtin_elementwise_fma(db, db, sq2[1]);
                const f2 va = UP ? qa : qa * sg[0], vb = UP ? qb : qb * sg[1];
                const float v[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool better = v[e] > ex[e];             // strict: the first extremum keeps the slot
                    ea[e] = better ? s : ea[e];                   // (two selects on one compare: fmaxf would add a
                    ex[e] = better ? v[e] : ex[e];                //  canonicalising multiply per loaded register pair)
                }
            } else {
                const f2 ya = ct[0] + qa, yb = ct[1] + qb;
                // (non-temporal for the big streams -- the T-Net's 2.7 GB: 748 -> 695 us; nothing reads Y back before it
                // has left every cache anyway)
                typedef float ec_f4 __attribute__((ext_vector_type(4)));
                if (ynt) __builtin_nontemporal_store(ec_f4{ya.x, ya.y, yb.x, yb.y}, reinterpret_cast<ec_f4 *>(yrow + (long long)s * C));
                else *reinterpret_cast<float4 *>(yrow + (long long)s * C) = make_float4(ya.x, ya.y, yb.x, yb.y);
                const f2 da = ya - pv[0], db = yb - pv[1];
                s1[0] += da; s1[1] += db;
                s2[0] = __builtin_elementwise_fma(da, da, s2[0]);
                s2[1] = __builtin_elementwise_fma(db, db, s2[1]);
            }
        };
        int s = 0;
        if ((S & 3) == 0) {
            for (; s + 8 <= S; s += 8) {
                const uint4 o0 = *reinterpret_cast<const uint4 *>(og + s), o1 = *reinterpret_cast<const uint4 *>(og + s + 4);
                const float4 q0 = ec_load4(rq, o0.x + tq), q1 = ec_load4(rq, o0.y + tq), q2 = ec_load4(rq, o0.z + tq),
                             q3 = ec_load4(rq, o0.w + tq), q4 = ec_load4(rq, o1.x + tq), q5 = ec_load4(rq, o1.y + tq),
                             q6 = ec_load4(rq, o1.z + tq), q7 = ec_load4(rq, o1.w + tq);
                take(q0, s); take(q1, s + 1); take(q2, s + 2); take(q3, s + 3);
                take(q4, s + 4); take(q5, s + 5); take(q6, s + 6); take(q7, s + 7);
            }
            for (; s + 4 <= S; s += 4) {
                const uint4 o0 = *reinterpret_cast<const uint4 *>(og + s);
                const float4 q0 = ec_load4(rq, o0.x + tq), q1 = ec_load4(rq, o0.y + tq), q2 = ec_load4(rq, o0.z + tq),
                             q3 = ec_load4(rq, o0.w + tq);
                take(q0, s); take(q1, s + 1); take(q2, s + 2); take(q3, s + 3);
            }
        }
        for (; s < S; ++s) take(ec_load4(rq, og[s] + tq), s);
        if (MODE == 0) {
            *reinterpret_cast<float4 *>(a.SQ + g * C + ch) =
                make_float4(fmaf(kf, qz[0].x, sq[0].x), fmaf(kf, qz[0].y, sq[0].y), fmaf(kf, qz[1].x, sq[1].x),
                            fmaf(kf, qz[1].y, sq[1].y));
            *reinterpret_cast<float4 *>(a.qsel + g * C + ch) =
                UP ? make_float4(ex[0], ex[1], ex[2], ex[3])
                   : make_float4(ex[0] * sg[0].x, ex[1] * sg[0].y, ex[2] * sg[1].x, ex[3] * sg[1].y);
            uchar4 a4;
            a4.x = (unsigned char)ea[0]; a4.y = (unsigned char)ea[1]; a4.z = (unsigned char)ea[2]; a4.w = (unsigned char)ea[3];
            *reinterpret_cast<uchar4 *>(a.arg + g * C + ch) = a4;
            if (a.stats) {
                const f2 k2 = f2{kf, kf};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f2 cv = ct[h] + cz[h];
                    s1[h] += __builtin_elementwise_fma(k2, cv, sq[h]);
                    s2[h] += __builtin_elementwise_fma(cv, __builtin_elementwise_fma(k2, cv, sq[h] + sq[h]), sq2[h]);
                }
            }
        }
    }
    (void)quad;
}

template <int MODE>
__global__ __launch_bounds__(256) void ec_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned ec_sm[];      // offsets [64][S] | statistics [16][2][64]
    const int S = a.S, C = a.C;
    const int H = C >> 6, P = a.m / kGB;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)(H * P));
    const int rem = (int)(vb - (unsigned)b * (unsigned)(H * P));
    const int h = rem / P, chunk = rem - h * P;
    const long long g0 = (long long)b * a.m + (long long)chunk * kGB;
    const int tid = threadIdx.x, set = tid >> 4, quad = tid & 15;
    const int ch = h * 64 + quad * 4;
    // byte offsets of the gathered rows, relative to Q: ((b n + idx) C + h 64) * 4 -- one LDS word per (group, slot)
    const int *ig = a.idx + g0 * S;
    const unsigned base = ((unsigned)b * (unsigned)a.n * (unsigned)a.ldq + (unsigned)h * 64u) * 4u;
    const unsigned c4 = (unsigned)a.ldq * 4u;
    for (int e = tid; e < kGB * S; e += 256) ec_sm[e] = (unsigned)ig[e] * c4 + base;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rq = ec_rsrc(a.Q, (unsigned)((long long)a.b * a.n * a.ldq * 4));
    const unsigned tq = (unsigned)quad * 16u;
    f2 s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    bool up = true;
    if (MODE == 0) {
        const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + ch);
        up = !(ga.x < 0.f) && !(ga.y < 0.f) && !(ga.z < 0.f) && !(ga.w < 0.f);
        up = __all(up) != 0;                                               // wave-uniform: one loop body per wave
    }
    if (up) ec_fwd_groups<MODE, true>(a, ec_sm, g0, set, quad, tq, rq, ch, s1, s2);
    else ec_fwd_groups<MODE, false>(a, ec_sm, g0, set, quad, tq, rq, ch, s1, s2);
    if (a.stats == nullptr) return;
    float *sm = reinterpret_cast<float *>(ec_sm + kGB * S);
    {
        float *d = sm + set * 128 + quad * 4;
        *reinterpret_cast<float4 *>(d) = make_float4(s1[0].x, s1[0].y, s1[1].x, s1[1].y);
        *reinterpret_cast<float4 *>(d + 64) = make_float4(s2[0].x, s2[0].y, s2[1].x, s2[1].y);
    }
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < kSets; ++l) t += sm[l * 128 + tid];
        const int which = tid >> 6, c = tid & 63;
        const long long row = (long long)b * P + chunk;
        a.stats[(row * 2 + which) * C + h * 64 + c] = t;
    }
}

// ---- inverse index ------------------------------------------------------------------------------------------------
// order [b][m S] u32: (group << 8 | slot) of every grouped row, sorted by the source point it references (within a list
// in the order the slots were handed out: no particular order);  start [b][n + 1]: list boundaries.
// One workgroup per cloud; the rows are read 16 bytes per lane, the group of a row comes from ONE division per four rows.
// STAGE: the sorted list is assembled in LDS as 16-bit (group << sbits | slot) codes and leaves with coalesced stores --
// scattering 4-byte entries straight to HBM cost 8.5x the list's bytes in partial-sector writes (357 MB counted for the
// 42 MB list of the cfg3 graph) and was most of the kernel's 130 us.
// perm [b][n]: the cloud's points in order of DESCENDING list length (ties in no particular order).  The walk deals
// positions of this order to its lane sets, so the 16 points a wave works on at a time have lists of (almost) one length
// -- a wave otherwise runs to the longest of 16 lists, about twice the mean on a kNN graph.
template <bool STAGE>
__global__ __launch_bounds__(1024) void ec_csr_build_kernel(int n, int m, int S, int sbits, const int *__restrict__ idx,
                                                            unsigned *__restrict__ order, int *__restrict__ start,
                                                            int *__restrict__ perm, unsigned short *__restrict__ codes_out) {
    extern __shared__ int ec_si[];              // cnt / cursor [n] | scan scratch [1024] | length bins [1024] | STAGE: codes [m S] u16
    int *cnt = ec_si, *sc = ec_si + n, *lh = sc + 1024;
    unsigned short *codes = reinterpret_cast<unsigned short *>(lh + 1024);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int mS = m * S;
    const int *ib = idx + (long long)b * mS;
    for (int i = tid; i < n; i += 1024) cnt[i] = 0;
    __syncthreads();
    const int n4 = mS >> 2;
    for (int e4 = tid; e4 < n4; e4 += 1024) {
        const int4 v = *reinterpret_cast<const int4 *>(ib + 4 * e4);
        atomicAdd(&cnt[v.x], 1); atomicAdd(&cnt[v.y], 1); atomicAdd(&cnt[v.z], 1); atomicAdd(&cnt[v.w], 1);
    }
    for (int e = 4 * n4 + tid; e < mS; e += 1024) atomicAdd(&cnt[ib[e]], 1);
    __syncthreads();
    {   // counting sort of the points by list length (bins 0 .. 1023, longer lists share the last bin), longest first
        lh[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) atomicAdd(&lh[1023 - min(cnt[i], 1023)], 1);
        __syncthreads();
        const int mine = lh[tid];
        sc[tid] = mine;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = tid >= off ? sc[tid - off] : 0;
            __syncthreads();
            sc[tid] += v;
            __syncthreads();
        }
        lh[tid] = sc[tid] - mine;                // exclusive: first position of the bin
        __syncthreads();
        int *pb = perm + (long long)b * n;
        for (int i = tid; i < n; i += 1024) pb[atomicAdd(&lh[1023 - min(cnt[i], 1023)], 1)] = i;
        __syncthreads();
    }
    const int per = (n + 1023) / 1024;
    const int i0 = tid * per, i1 = min(n, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; ++i) local += cnt[i];
    sc[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? sc[tid - off] : 0;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    int run = sc[tid] - local;
    int *sb = start + (long long)b * (n + 1);
    for (int i = i0; i < i1; ++i) {
        const int c = cnt[i];
        cnt[i] = run;                            // the bin becomes the list's cursor
        sb[i] = run;
        run += c;
    }
    if (tid == 0) sb[n] = mS;
    __syncthreads();
    unsigned *ob = order + (long long)b * mS;
    auto put = [&](int i, int g, int s) {
        const int pos = atomicAdd(&cnt[i], 1);
        if (STAGE) codes[pos] = (unsigned short)((g << sbits) | s);
        else ob[pos] = ((unsigned)g << 8) | (unsigned)s;
    };
    for (int e4 = tid; e4 < n4; e4 += 1024) {
        const int4 v = *reinterpret_cast<const int4 *>(ib + 4 * e4);
        const int e = 4 * e4;
        int g = e / S, s = e - g * S;
        const int vi[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            put(vi[u], g, s);
            if (++s == S) { s = 0; ++g; }
        }
    }
    for (int e = 4 * n4 + tid; e < mS; e += 1024) {
        const int g = e / S;
        put(ib[e], g, e - g * S);
    }
    if (!STAGE) return;
    __syncthreads();
    const unsigned smask = (1u << sbits) - 1u;
    unsigned short *cb = codes_out + (long long)b * mS;       // the 16-bit form too: the LDS-resident walk stages it
    for (int k = tid; k < mS; k += 1024) {
        const unsigned c = codes[k];
        ob[k] = ((c >> sbits) << 8) | (c & smask);
        cb[k] = (unsigned short)c;
    }
}

// ---- owner walk ---------------------------------------------------------------------------------------------------
//   HAS_G = false (EdgeConv, after the arg-row kernel initialised dQ):  dQ[b,i,:] += q (cnt Q[i] + sum Ctr[g]) + cnt t
//   HAS_G = true  (first layer of a gather stack):  dQ[b,i,:] = p sum G[g,s] + q (cnt Q[i] + sum Ctr[g]) + cnt t
// A 16-lane set per source point and 64-channel slice, four list entries in flight; an entry beyond the list is the
// sentinel (m << 8), whose rows lie outside the cloud-sized buffer resources and read as zeros.
struct WalkArgs {
    int b, n, m, S, C;
    int ldq, ldc, lddq;
    const float *Q, *Ctr, *G;
    const float *p, *q, *t;
    const unsigned *order;
    const int *start;
    float *dQ;
};

template <bool HAS_G>
__global__ __launch_bounds__(256) void ec_walk_kernel(WalkArgs a) {
    const int C = a.C, S = a.S;
    const int H = C >> 6, P = (a.n + kGB - 1) / kGB;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)(H * P));
    const int rem = (int)(vb - (unsigned)b * (unsigned)(H * P));
    const int h = rem / P, chunk = rem - h * P;
    const int tid = threadIdx.x, set = tid >> 4, quad = tid & 15;
    const int ch = h * 64 + quad * 4;
    const unsigned c4 = (unsigned)C * 4u, cc4 = (unsigned)a.ldc * 4u, tq = ((unsigned)h * 64u + (unsigned)quad * 4u) * 4u;
    const int mS = a.m * S;
    // (the Ctr resource ends with the cloud's last row: a column-slice view shares its rows with Q, never with cloud b + 1)
    const __amdgpu_buffer_rsrc_t rc = ec_rsrc(a.Ctr + (long long)b * a.m * a.ldc, (unsigned)(a.m - 1) * cc4 + c4);
    const __amdgpu_buffer_rsrc_t rg = ec_rsrc(HAS_G ? a.G + (long long)b * mS * C : a.Ctr, HAS_G ? (unsigned)mS * c4 : 0u);
    const __amdgpu_buffer_rsrc_t ro = ec_rsrc(a.order + (long long)b * mS, (unsigned)mS * 4u);
    const int *sb = a.start + (long long)b * (a.n + 1);
    const float4 cq = *reinterpret_cast<const float4 *>(a.q + ch);
    const float4 ct = *reinterpret_cast<const float4 *>(a.t + ch);
    float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_G) cp = *reinterpret_cast<const float4 *>(a.p + ch);
    const unsigned sentinel = (unsigned)a.m << 8;
#pragma unroll 1
    for (int r = 0; r < kGB / kSets; ++r) {
        const int i = chunk * kGB + set + kSets * r;
        if (i >= a.n) break;
        const int k0 = sb[i], k1 = sb[i + 1];
        f2 ac[2] = {{0.f, 0.f}, {0.f, 0.f}}, ag[2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int k = k0; k < k1; k += 4) {
            unsigned w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = ec_load1(ro, (unsigned)(k + u) * 4u);     // beyond the cloud's list: 0 ...
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = k + u < k1 ? w[u] : sentinel;             // ... beyond this list: sentinel
            float4 cc[4], gg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned g = w[u] >> 8;
                cc[u] = ec_load4(rc, g * cc4 + tq);
                if (HAS_G) gg[u] = ec_load4(rg, (g * (unsigned)S + (w[u] & 255u)) * c4 + tq);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ac[0] += f2{cc[u].x, cc[u].y}; ac[1] += f2{cc[u].z, cc[u].w};
                if (HAS_G) { ag[0] += f2{gg[u].x, gg[u].y}; ag[1] += f2{gg[u].z, gg[u].w}; }
            }
        }
        const long long pt = (long long)b * a.n + i;
        const float kf = (float)(k1 - k0);
        const float4 qi = *reinterpret_cast<const float4 *>(a.Q + pt * a.ldq + ch);
        float4 *dst = reinterpret_cast<float4 *>(a.dQ + pt * a.lddq + ch);
        float4 d;
        d.x = fmaf(cq.x, fmaf(kf, qi.x, ac[0].x), kf * ct.x);
        d.y = fmaf(cq.y, fmaf(kf, qi.y, ac[0].y), kf * ct.y);
        d.z = fmaf(cq.z, fmaf(kf, qi.z, ac[1].x), kf * ct.z);
        d.w = fmaf(cq.w, fmaf(kf, qi.w, ac[1].y), kf * ct.w);
        if (HAS_G) {
            d.x = fmaf(cp.x, ag[0].x, d.x); d.y = fmaf(cp.y, ag[0].y, d.y);
            d.z = fmaf(cp.z, ag[1].x, d.z); d.w = fmaf(cp.w, ag[1].y, d.w);
        } else {
            const float4 o = *dst;
            d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w;
        }
        *dst = d;
    }
}

// ---- per-group output of the T-Net's scatter ------------------------------------------------------------------------
//   dCtr[g,:] = sum_s dY[g,s,:],  dY = p G + q (Q[idx] + Ctr[g]) + t   =>   p sum_s G[g,s] + q (sum_s Q[idx[g,s]] + k Ctr[g]) + k t
// the G rows of a group are one contiguous run (streamed), the Q rows are L2 gathers exactly as in the forward.
struct CtrArgs {
    int b, n, m, S, C;
    int ldq, ldc, lddc;
    const float *Q, *Ctr, *G;
    const int *idx;
    const float *p, *q, *t;
    float *dCtr;
};

__global__ __launch_bounds__(256) void ec_tnet_ctr_kernel(CtrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned ec_sm[];      // offsets [64][S]
    const int S = a.S, C = a.C;
    const int H = C >> 6, P = a.m / kGB;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)(H * P));
    const int rem = (int)(vb - (unsigned)b * (unsigned)(H * P));
    const int h = rem / P, chunk = rem - h * P;
    const long long g0 = (long long)b * a.m + (long long)chunk * kGB;
    const int tid = threadIdx.x, set = tid >> 4, quad = tid & 15;
    const int ch = h * 64 + quad * 4;
    const int *ig = a.idx + g0 * S;
    const unsigned base = ((unsigned)b * (unsigned)a.n * (unsigned)a.ldq + (unsigned)h * 64u) * 4u;
    const unsigned c4 = (unsigned)a.ldq * 4u;
    for (int e = tid; e < kGB * S; e += 256) ec_sm[e] = (unsigned)ig[e] * c4 + base;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rq = ec_rsrc(a.Q, (unsigned)((long long)a.b * a.n * a.ldq * 4));
    const unsigned tq = (unsigned)quad * 16u;
    const float4 cp = *reinterpret_cast<const float4 *>(a.p + ch);
    const float4 cq = *reinterpret_cast<const float4 *>(a.q + ch);
    const float4 ct = *reinterpret_cast<const float4 *>(a.t + ch);
    const float kf = (float)S;
#pragma unroll 1
    for (int r = 0; r < kGB / kSets; ++r) {
        const int gl = set + kSets * r;
        const long long g = g0 + gl;
        const unsigned *og = ec_sm + gl * S;
        const float *grow = a.G + g * S * (long long)C + ch;
        f2 sq[2] = {{0.f, 0.f}, {0.f, 0.f}}, sgm[2] = {{0.f, 0.f}, {0.f, 0.f}};
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            float4 qv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                qv[u] = ec_load4(rq, og[s + u] + tq);
                gv[u] = *reinterpret_cast<const float4 *>(grow + (long long)(s + u) * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sq[0] += f2{qv[u].x, qv[u].y}; sq[1] += f2{qv[u].z, qv[u].w};
                sgm[0] += f2{gv[u].x, gv[u].y}; sgm[1] += f2{gv[u].z, gv[u].w};
            }
        }
        for (; s < S; ++s) {
            const float4 qv = ec_load4(rq, og[s] + tq);
            const float4 gv = *reinterpret_cast<const float4 *>(grow + (long long)s * C);
            sq[0] += f2{qv.x, qv.y}; sq[1] += f2{qv.z, qv.w};
            sgm[0] += f2{gv.x, gv.y}; sgm[1] += f2{gv.z, gv.w};
        }
        const float4 c4v = *reinterpret_cast<const float4 *>(a.Ctr + g * a.ldc + ch);
        float4 d;
        d.x = fmaf(cp.x, sgm[0].x, fmaf(cq.x, fmaf(kf, c4v.x, sq[0].x), kf * ct.x));
        d.y = fmaf(cp.y, sgm[0].y, fmaf(cq.y, fmaf(kf, c4v.y, sq[0].y), kf * ct.y));
        d.z = fmaf(cp.z, sgm[1].x, fmaf(cq.z, fmaf(kf, c4v.z, sq[1].x), kf * ct.z));
        d.w = fmaf(cp.w, sgm[1].y, fmaf(cq.w, fmaf(kf, c4v.w, sq[1].y), kf * ct.w));
        *reinterpret_cast<float4 *>(a.dCtr + g * a.lddc + ch) = d;
    }
}

// ---- LDS-resident slices (clouds of up to ~2500 points) --------------------------------------------------------------
// The L2-gather kernels above top out at ~10.7 TB/s of gathered rows (measured: 2.7 GB of 256-byte row pieces in 255 us
// at C = 64, 5.4 GB in 492 us at C = 128 -- a third of the L2's streaming rate, and traffic from HBM already at the
// algorithmic 0.62 GB).  A cloud's rows are gathered k = 20 times each, so the table belongs in LDS: one workgroup of
// 1024 lanes per (cloud, 16-channel slice) stages the slice once ([n][16] floats = 128 KB at n = 2048) and every gather
// is a ds_read_b128.  Four lanes per group / point (a float4 of channels each), 256 groups in flight.
constexpr int kSliceCh = 16;

template <bool UP, int NI>         // NI int4 index registers per group: S <= 4 NI in the prefetched form
__device__ __forceinline__ void ec_fwd_lds_groups(const FwdArgs &a, const float4 *qs, int b, int quad, int cl, int ch,
                                                  f2 (&s1)[2], f2 (&s2)[2]) {
    const int S = a.S, C = a.C, m = a.m;
    const float kf = (float)S;
    f2 sg[2] = {{1.f, 1.f}, {1.f, 1.f}};
    f2 qz[2] = {{0.f, 0.f}, {0.f, 0.f}}, cz[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (!UP) {
        const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + ch);
        sg[0] = f2{ga.x < 0.f ? -1.f : 1.f, ga.y < 0.f ? -1.f : 1.f};
        sg[1] = f2{ga.z < 0.f ? -1.f : 1.f, ga.w < 0.f ? -1.f : 1.f};
    }
    if (a.stats) {
        const float4 q0 = *reinterpret_cast<const float4 *>(a.Q + ch);
        float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.pivot) p4 = *reinterpret_cast<const float4 *>(a.pivot + ch);
        qz[0] = f2{q0.x, q0.y}; qz[1] = f2{q0.z, q0.w};
        cz[0] = f2{q0.x - p4.x, q0.y - p4.y}; cz[1] = f2{q0.z - p4.z, q0.w - p4.w};
    }
    // the group's indices (and its Ctr quad) are loaded ONE GROUP AHEAD: counters of the first version showed the waves
    // parked at s_waitcnt 68 % of their cycles, a global-load latency per four neighbours (profiles/r05_ec_counters.txt)
    const bool fast = (S & 3) == 0 && S <= 4 * NI;
    const int ni = S >> 2;
    int4 cur[NI];
    float4 curc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quad < m) {
        const long long g = (long long)b * m + quad;
        if (fast) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (j < ni) cur[j] = *reinterpret_cast<const int4 *>(a.idx + g * S + 4 * j);
        }
        curc = *reinterpret_cast<const float4 *>(a.Ctr + g * a.ldc + ch);
    }
#pragma unroll 1
    for (int gl = quad; gl < m; gl += 256) {
        const long long g = (long long)b * m + gl;
        int4 nxt[NI];
        float4 nxtc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gl + 256 < m) {
            if (fast) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (j < ni) nxt[j] = *reinterpret_cast<const int4 *>(a.idx + (g + 256) * S + 4 * j);
            }
            nxtc = *reinterpret_cast<const float4 *>(a.Ctr + (g + 256) * a.ldc + ch);
        }
        const f2 ct[2] = {{curc.x, curc.y}, {curc.z, curc.w}};
        f2 sq[2] = {{0.f, 0.f}, {0.f, 0.f}}, sq2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        float ex[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int ea[4] = {0, 0, 0, 0};
        auto take = [&](const float4 &q, int s) {
            const f2 qa = f2{q.x, q.y}, qb = f2{q.z, q.w};
            const f2 da = qa - qz[0], db = qb - qz[1];
            sq[0] += da; sq[1] += db;
            sq2[0] = __builtin_elementwise_fma(da, da, sq2[0]);
            sq2[1] = __builtin_elementwise_fma(db, db, sq2[1]);
            const f2 va = UP ? qa : qa * sg[0], vb = UP ? qb : qb * sg[1];
            const float v[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool better = v[e] > ex[e];                 // strict: the first extremum keeps the slot
                ea[e] = better ? s : ea[e];
                ex[e] = better ? v[e] : ex[e];
            }
        };
        if (fast) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (j < ni) {
                    const float4 q0 = qs[cur[j].x * 4 + cl], q1 = qs[cur[j].y * 4 + cl], q2 = qs[cur[j].z * 4 + cl],
                                 q3 = qs[cur[j].w * 4 + cl];
                    take(q0, 4 * j); take(q1, 4 * j + 1); take(q2, 4 * j + 2); take(q3, 4 * j + 3);
                }
        } else {
            const int *ig = a.idx + g * S;
            for (int s = 0; s < S; ++s) take(qs[ig[s] * 4 + cl], s);
        }
        *reinterpret_cast<float4 *>(a.SQ + g * C + ch) =
            make_float4(fmaf(kf, qz[0].x, sq[0].x), fmaf(kf, qz[0].y, sq[0].y), fmaf(kf, qz[1].x, sq[1].x),
                        fmaf(kf, qz[1].y, sq[1].y));
        *reinterpret_cast<float4 *>(a.qsel + g * C + ch) =
            UP ? make_float4(ex[0], ex[1], ex[2], ex[3])
               : make_float4(ex[0] * sg[0].x, ex[1] * sg[0].y, ex[2] * sg[1].x, ex[3] * sg[1].y);
        uchar4 a4;
        a4.x = (unsigned char)ea[0]; a4.y = (unsigned char)ea[1]; a4.z = (unsigned char)ea[2]; a4.w = (unsigned char)ea[3];
        *reinterpret_cast<uchar4 *>(a.arg + g * C + ch) = a4;
        if (a.stats) {
            const f2 k2 = f2{kf, kf};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 cv = ct[h] + cz[h];
                s1[h] += __builtin_elementwise_fma(k2, cv, sq[h]);
                s2[h] += __builtin_elementwise_fma(cv, __builtin_elementwise_fma(k2, cv, sq[h] + sq[h]), sq2[h]);
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) cur[j] = nxt[j];
        curc = nxtc;
    }
}

// statistics: ONE partial row per cloud, stats [b][2][C]
__global__ __launch_bounds__(1024) void ec_fwd_lds_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 ec_qs[];      // [n][4] | wave sums [16][2][16] floats
    const int C = a.C, n = a.n, NS = C / kSliceCh;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)NS), sl = (int)(vb - (unsigned)b * (unsigned)NS);
    const int tid = threadIdx.x, quad = tid >> 2, cl = tid & 3;
    const int ch = sl * kSliceCh + cl * 4;
    {
        const float *src = a.Q + (long long)b * n * a.ldq + ch;
        for (int i = quad; i < n; i += 256) ec_qs[i * 4 + cl] = *reinterpret_cast<const float4 *>(src + (long long)i * a.ldq);
    }
    __syncthreads();
    f2 s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + ch);
    const bool up = __all(!(ga.x < 0.f) && !(ga.y < 0.f) && !(ga.z < 0.f) && !(ga.w < 0.f)) != 0;
    // (five index registers: k <= 20 neighbours take the prefetched form, larger groups read their indices in line;
    //  eight registers per buffer spilled at the 128 VGPRs a 1024-lane workgroup leaves per lane)
    if (up) ec_fwd_lds_groups<true, 5>(a, ec_qs, b, quad, cl, ch, s1, s2);
    else ec_fwd_lds_groups<false, 5>(a, ec_qs, b, quad, cl, ch, s1, s2);
    if (a.stats == nullptr) return;
    float v[8] = {s1[0].x, s1[0].y, s1[1].x, s1[1].y, s2[0].x, s2[0].y, s2[1].x, s2[1].y};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int d = 4; d < 64; d <<= 1) v[e] += __shfl_xor(v[e], d, 64);     // the 16 quads of a wave, per channel lane
    }
    float *ws = reinterpret_cast<float *>(ec_qs + (size_t)n * 4);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ws[(wave * 2 + 0) * kSliceCh + lane * 4 + e] = v[e];
            ws[(wave * 2 + 1) * kSliceCh + lane * 4 + e] = v[4 + e];
        }
    }
    __syncthreads();
    if (tid < 2 * kSliceCh) {
        const int which = tid / kSliceCh, c = tid % kSliceCh;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += ws[(w * 2 + which) * kSliceCh + c];
        a.stats[((long long)b * 2 + which) * C + sl * kSliceCh + c] = t;
    }
}

// owner walk with the cloud's Ctr slice in LDS (+ one zero row at index m for the entries beyond a list), the list
// boundaries and the length-sorted order of the points next to it (a wave's 64 points have lists of one length).
// ONE LANE per source point and 8-channel slice, and EVERYTHING the inner loop touches in LDS: the cloud's Ctr slice
// ([m + 1][8] floats, row m = zeros), the list boundaries and the inverse index itself as 16-bit (group << sbits | slot)
// codes (m S of them: 80 KB for the cfg3 graph).  History of this kernel, all measured on the cfg3 graph at C = 64
// (profiles/r05_ec_counters.txt): four lanes per point with the lists read from global memory four entries at a time
// parked the waves at s_waitcnt 81 % of their cycles (a global-load latency per batch); batches of eight with the next
// batch in flight cut the waiting but paid the entry decode once per float4 (54 k VALU per SIMD); one lane per point with
// per-lane list loads turned the address coalescer into the bottleneck (64 distinct lines per load instruction).  With the
// list in LDS there is no global access inside a list at all.  The two float4 of a row are taken in the rotated order
// (j + lane) mod 2 and rows are 32 bytes apart, so the 16 lanes of an LDS conflict group spread over all 16 bank quads.
constexpr int kWalkCh = 8;

struct WalkLdsArgs {
    WalkArgs w;
    const int *perm;
    const unsigned short *codes;
    int sbits;
};

__global__ __launch_bounds__(1024) void ec_walk_lds_kernel(WalkLdsArgs wa) {
    const WalkArgs &a = wa.w;
    extern __shared__ __attribute__((aligned(16))) float4 ec_cs[];      // Ctr [m + 1][2] | start [n + 1] | codes [m S] u16
    const int C = a.C, n = a.n, m = a.m, NS = C / kWalkCh;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)NS), sl = (int)(vb - (unsigned)b * (unsigned)NS);
    const int tid = threadIdx.x;
    const int mS = m * a.S;
    int *ss = reinterpret_cast<int *>(ec_cs + ((size_t)m + 1) * 2);
    // (the offset is taken on the LDS pointer: rounding it up through uintptr_t made it a GENERIC pointer and every read of
    // the codes a flat load through the LDS aperture)
    unsigned short *cs16 = reinterpret_cast<unsigned short *>(ec_cs + ((size_t)m + 1) * 2 + ((size_t)n + 1 + 3) / 4);
    {
        const int pair = tid >> 1, cl = tid & 1;
        const float *src = a.Ctr + (long long)b * m * a.ldc + sl * kWalkCh + cl * 4;
        for (int j = pair; j < m; j += 512) ec_cs[j * 2 + cl] = *reinterpret_cast<const float4 *>(src + (long long)j * a.ldc);
        if (pair == 0) ec_cs[m * 2 + cl] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int *sb = a.start + (long long)b * (n + 1);
        for (int i = tid; i <= n; i += 1024) ss[i] = sb[i];
        const unsigned short *cg = wa.codes + (long long)b * mS;
        if ((mS & 7) == 0 && (reinterpret_cast<uintptr_t>(cg) & 15) == 0) {
            const uint4 *c4 = reinterpret_cast<const uint4 *>(cg);
            uint4 *d4 = reinterpret_cast<uint4 *>(cs16);
            for (int k = tid; k < (mS >> 3); k += 1024) d4[k] = c4[k];
        } else {
            for (int k = tid; k < mS; k += 1024) cs16[k] = cg[k];
        }
    }
    __syncthreads();
    const int sbits = wa.sbits;
    const int c0 = tid & 1, c1 = c0 ^ 1;         // this lane's float4 at read 0 / read 1
    const float4 cq0 = *reinterpret_cast<const float4 *>(a.q + sl * kWalkCh + c0 * 4);
    const float4 cq1 = *reinterpret_cast<const float4 *>(a.q + sl * kWalkCh + c1 * 4);
    const float4 ct0 = *reinterpret_cast<const float4 *>(a.t + sl * kWalkCh + c0 * 4);
    const float4 ct1 = *reinterpret_cast<const float4 *>(a.t + sl * kWalkCh + c1 * 4);
    const float *Qb = a.Q + (long long)b * n * a.ldq + sl * kWalkCh;
    float *dQb = a.dQ + (long long)b * n * a.lddq + sl * kWalkCh;
    const int ldq = a.ldq, lddq = a.lddq;
    const int *pb = wa.perm + (long long)b * n;
#pragma unroll 1
    for (int p = tid; p < n; p += 1024) {
        const int i = pb[p];
        const int k0 = ss[i], k1 = ss[i + 1];
        const float4 q0 = *reinterpret_cast<const float4 *>(Qb + (long long)i * ldq + c0 * 4);
        const float4 q1 = *reinterpret_cast<const float4 *>(Qb + (long long)i * ldq + c1 * 4);
        const float4 o0 = *reinterpret_cast<const float4 *>(dQb + (long long)i * lddq + c0 * 4);
        const float4 o1 = *reinterpret_cast<const float4 *>(dQb + (long long)i * lddq + c1 * 4);
        f2 a0[2] = {{0.f, 0.f}, {0.f, 0.f}}, a1[2] = {{0.f, 0.f}, {0.f, 0.f}};
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            unsigned g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = (unsigned)cs16[k + u] >> sbits;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 x0 = ec_cs[g[u] * 2 + c0], x1 = ec_cs[g[u] * 2 + c1];
                a0[0] += f2{x0.x, x0.y}; a0[1] += f2{x0.z, x0.w};
                a1[0] += f2{x1.x, x1.y}; a1[1] += f2{x1.z, x1.w};
            }
        }
        for (; k < k1; ++k) {
            const unsigned g = (unsigned)cs16[k] >> sbits;
            const float4 x0 = ec_cs[g * 2 + c0], x1 = ec_cs[g * 2 + c1];
            a0[0] += f2{x0.x, x0.y}; a0[1] += f2{x0.z, x0.w};
            a1[0] += f2{x1.x, x1.y}; a1[1] += f2{x1.z, x1.w};
        }
        const float kf = (float)(k1 - k0);
        float4 d0, d1;
        d0.x = o0.x + fmaf(cq0.x, fmaf(kf, q0.x, a0[0].x), kf * ct0.x);
        d0.y = o0.y + fmaf(cq0.y, fmaf(kf, q0.y, a0[0].y), kf * ct0.y);
        d0.z = o0.z + fmaf(cq0.z, fmaf(kf, q0.z, a0[1].x), kf * ct0.z);
        d0.w = o0.w + fmaf(cq0.w, fmaf(kf, q0.w, a0[1].y), kf * ct0.w);
        d1.x = o1.x + fmaf(cq1.x, fmaf(kf, q1.x, a1[0].x), kf * ct1.x);
        d1.y = o1.y + fmaf(cq1.y, fmaf(kf, q1.y, a1[0].y), kf * ct1.y);
        d1.z = o1.z + fmaf(cq1.z, fmaf(kf, q1.z, a1[1].x), kf * ct1.z);
        d1.w = o1.w + fmaf(cq1.w, fmaf(kf, q1.w, a1[1].y), kf * ct1.w);
        *reinterpret_cast<float4 *>(dQb + (long long)i * lddq + c0 * 4) = d0;
        *reinterpret_cast<float4 *>(dQb + (long long)i * lddq + c1 * 4) = d1;
    }
}

constexpr size_t kLdsMax = 160 * 1024;
bool ec_lds_fwd_ok(int n) { return (size_t)n * 64 + 16 * 2 * kSliceCh * sizeof(float) <= kLdsMax; }
size_t ec_lds_walk_bytes(int n, int m, int s) { return ((size_t)m + 1) * 32 + ((size_t)n + 1) * 4 + (((size_t)m * s * 2 + 15) & ~(size_t)15) + 16; }
int ec_sbits(int s) { int sb = 0; while ((1 << sb) < s) ++sb; return sb; }
bool ec_codes16_ok(int m, int s) { return ((long long)m << ec_sbits(s)) <= 65536; }
bool ec_lds_walk_ok(int n, int m, int s) { return ec_codes16_ok(m, s) && ec_lds_walk_bytes(n, m, s) <= kLdsMax; }
bool ec_lds_on() {
    static const bool on = [] { const char *e = getenv("PCOPS_EDGECONV_LDS"); return !(e && e[0] == '0'); }();
    return on;
}

// ---- the arg-row term of the EdgeConv backward + dCtr (gather.hip edge_pool_bwd_sparse_kernel with row strides) -------
//   a[g,c] = p[c] gpool[g,c] [relu(bn(ysel[g,c])) > 0]  goes to row idx[g, arg[g,c]] of dQ;  dCtr[g] = q (SQ + k Ctr) + k t + a.
// One workgroup per (cloud, 16-channel slice): the slice of dQ (n x 16 floats) is accumulated in LDS with LDS atomics --
// one per (group, channel), 20x fewer than the dense term -- and leaves with plain stores (this initialises dQ).
constexpr int kSparseSlice = 16;
struct SparseArgs {
    int n, m, S, C, ldc, lddq, lddc;
    const float *gpool, *ysel, *SQ, *Ctr;
    const unsigned char *arg;
    const int *idx;
    const float *scale, *shift, *p, *q, *t;
    float *dCtr, *dQ;
};

__global__ __launch_bounds__(1024) void ec_sparse_kernel(SparseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ec_acc[];          // [n][16]
    constexpr int NT = 1024, GL = NT / kSparseSlice, U = 4;
    const int n = a.n, m = a.m, S = a.S, C = a.C;
    const int nsl = C / kSparseSlice;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)nsl), c0 = (int)(vb - (unsigned)b * (unsigned)nsl) * kSparseSlice;
    const int tid = threadIdx.x, cl = tid % kSparseSlice, gl = tid / kSparseSlice;
    for (int e = tid; e < n * kSparseSlice; e += NT) ec_acc[e] = 0.f;
    __syncthreads();
    const int c = c0 + cl;
    const float sc = a.scale[c], sh = a.shift[c], pc = a.p[c], qc = a.q[c], tc = a.t[c];
    const float kf = (float)S;
    for (int j0 = gl; j0 < m; j0 += GL * U) {
        float ys[U], gp[U], ce[U], sq[U];
        int ar[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * GL < m ? j0 + u * GL : j0;
            const long long g = (long long)b * m + j;
            const long long e = g * C + c;
            ys[u] = a.ysel[e]; gp[u] = a.gpool[e]; ce[u] = a.Ctr[g * a.ldc + c]; sq[u] = a.SQ[e]; ar[u] = a.arg[e];
        }
        int di[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * GL < m ? j0 + u * GL : j0;
            di[u] = a.idx[((long long)b * m + j) * S + ar[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * GL;
            if (j >= m) continue;
            const long long g = (long long)b * m + j;
            const float av = fmaf(ys[u], sc, sh) > 0.f ? pc * gp[u] : 0.f;
            a.dCtr[g * a.lddc + c] = fmaf(qc, fmaf(kf, ce[u], sq[u]), fmaf(kf, tc, av));
            if (av != 0.f) atomicAdd(&ec_acc[di[u] * kSparseSlice + cl], av);
        }
    }
    __syncthreads();
    float *dst = a.dQ + (long long)b * n * a.lddq + c0;
    for (int e = tid; e < n * (kSparseSlice / 4); e += NT) {
        const int i = e / (kSparseSlice / 4), quad = (e % (kSparseSlice / 4)) * 4;
        *reinterpret_cast<float4 *>(dst + (long long)i * a.lddq + quad) = *reinterpret_cast<const float4 *>(&ec_acc[i * kSparseSlice + quad]);
    }
}

// ---- both terms of the EdgeConv backward in ONE owner walk (round 5, second half) -----------------------------------
//   dQ[i,c] = q (cnt Q[i] + sum_{(g,s) -> i} Ctr[g]) + cnt t  +  sum_{(g,s) -> i, arg[g,c] == s} a[g,c],
//   a[g,c] = p gpool[g,c] [relu(bn(ysel[g,c])) > 0],     dCtr[g] = q (SQ + k Ctr) + k t + a.
// The arg-row term used to be its own kernel that added a[g,c] into an LDS copy of dQ with ds_add_f32 -- 0.33 lane-ops per
// clock and CU (tools/ubench/lds_atomic.hip): 165 of its 195 us were that one instruction -- and the dense term a second
// walk that read dQ back.  Here a point's list is walked ONCE: next to the Ctr rows the LDS holds a[g] and the four arg
// bytes of the slice, and an entry adds a[g,c] where its slot IS the arg -- a compare and a select, no atomic; dQ leaves
// with plain stores and is never read, dCtr comes out of the staging phase.
// One workgroup per (cloud, 8-channel slice): the slice is loaded 32 contiguous bytes per group row (two lanes), parked in
// registers, and walked in two passes of 4 channels -- LDS: Ctr | a [m][8] floats, arg [m] u32, codes [m S] u16 (154 KB at
// m = 2048, S = 20), the list boundaries come from L2.  A lane owns TWO points per pass: position p and position n - 1 - p
// of the descending-length order (perm), so every wave gets long lists and short ones.
constexpr int kBwdCh = 8, kBwdGP = 4;        // channels per workgroup; groups a lane pair stages (m <= 512 kBwdGP)

struct BwdLdsArgs {
    int n, m, S, C, sbits;
    int ldq, ldc, lddq, lddc;
    const float *Q, *Ctr, *gpool, *ysel, *SQ;
    const unsigned char *arg;
    const float *scale, *shift, *p, *q, *t;
    const int *start, *perm;
    const unsigned short *codes;
    float *dQ, *dCtr;
};

#ifdef PCOPS_EC_PROF
__device__ unsigned long long g_ec_prof[8];
#define EC_T(i_) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_ec_prof[i_], n_ - ec_t); ec_t = n_; } } while (0)
#else
#define EC_T(i_) do {} while (0)
#endif
__global__ __launch_bounds__(1024) void ec_bwd_lds_kernel(BwdLdsArgs a) {
#ifdef PCOPS_EC_PROF
    unsigned long long ec_t = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) float4 ec_ca[];      // Ctr rows [m + 1] | a rows [m + 1]
    const int n = a.n, m = a.m, C = a.C, NS = C / kBwdCh, mS = m * a.S;
    const unsigned vb = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(vb / (unsigned)NS), sl = (int)(vb - (unsigned)b * (unsigned)NS);
    const int tid = threadIdx.x;
    unsigned *ar = reinterpret_cast<unsigned *>(ec_ca + ((size_t)m + 1) * 2);                   // [m + 1]
    // (offsets on the LDS pointer itself: a round trip through uintptr_t makes it a generic pointer and the reads flat loads)
    unsigned short *cs16 = reinterpret_cast<unsigned short *>(ec_ca + ((size_t)m + 1) * 2 + ((size_t)m + 1 + 3) / 4);
    // ---- staging: this lane's half (4 channels) of up to kBwdGP group rows, dCtr on the way
    const int h = tid & 1, cbase = sl * kBwdCh + 4 * h;
    const float4 cq = *reinterpret_cast<const float4 *>(a.q + cbase);
    const float4 ct = *reinterpret_cast<const float4 *>(a.t + cbase);
    float4 sctr[kBwdGP], sa[kBwdGP];
    unsigned sarg[kBwdGP];
    {
        const float4 cs = *reinterpret_cast<const float4 *>(a.scale + cbase);
        const float4 ch = *reinterpret_cast<const float4 *>(a.shift + cbase);
        const float4 cp = *reinterpret_cast<const float4 *>(a.p + cbase);
        const float kf = (float)a.S;
        float4 ys[kBwdGP], gp[kBwdGP], sq[kBwdGP];
#pragma unroll
        for (int r = 0; r < kBwdGP; ++r) {
            const int j = (tid >> 1) + 512 * r;
            const long long g = (long long)b * m + (j < m ? j : 0);
            sctr[r] = *reinterpret_cast<const float4 *>(a.Ctr + g * a.ldc + cbase);
            ys[r] = *reinterpret_cast<const float4 *>(a.ysel + g * C + cbase);
            gp[r] = *reinterpret_cast<const float4 *>(a.gpool + g * C + cbase);
            sq[r] = *reinterpret_cast<const float4 *>(a.SQ + g * C + cbase);
            sarg[r] = *reinterpret_cast<const unsigned *>(a.arg + g * C + cbase);
        }
#pragma unroll
        for (int r = 0; r < kBwdGP; ++r) {
            const int j = (tid >> 1) + 512 * r;
            float4 av;
            av.x = fmaf(ys[r].x, cs.x, ch.x) > 0.f ? cp.x * gp[r].x : 0.f;
            av.y = fmaf(ys[r].y, cs.y, ch.y) > 0.f ? cp.y * gp[r].y : 0.f;
            av.z = fmaf(ys[r].z, cs.z, ch.z) > 0.f ? cp.z * gp[r].z : 0.f;
            av.w = fmaf(ys[r].w, cs.w, ch.w) > 0.f ? cp.w * gp[r].w : 0.f;
            sa[r] = av;
            if (j < m) {
                float4 d;
                d.x = fmaf(cq.x, fmaf(kf, sctr[r].x, sq[r].x), fmaf(kf, ct.x, av.x));
                d.y = fmaf(cq.y, fmaf(kf, sctr[r].y, sq[r].y), fmaf(kf, ct.y, av.y));
                d.z = fmaf(cq.z, fmaf(kf, sctr[r].z, sq[r].z), fmaf(kf, ct.z, av.z));
                d.w = fmaf(cq.w, fmaf(kf, sctr[r].w, sq[r].w), fmaf(kf, ct.w, av.w));
                *reinterpret_cast<float4 *>(a.dCtr + ((long long)b * m + j) * a.lddc + cbase) = d;
            }
        }
        const unsigned short *cg = a.codes + (long long)b * mS;
        if ((mS & 7) == 0 && (reinterpret_cast<uintptr_t>(cg) & 15) == 0) {
            const uint4 *c4 = reinterpret_cast<const uint4 *>(cg);
            uint4 *d4 = reinterpret_cast<uint4 *>(cs16);
            for (int k = tid; k < (mS >> 3); k += 1024) d4[k] = c4[k];
        } else {
            for (int k = tid; k < mS; k += 1024) cs16[k] = cg[k];
        }
    }
    EC_T(0);                                                             // staging loads consumed, dCtr and codes issued
    const int sbits = a.sbits;
    const unsigned smask = (1u << sbits) - 1u;
    const int *sb = a.start + (long long)b * (n + 1);
    const int *pb = a.perm + (long long)b * n;
    // this lane's two points (positions tid and n - 1 - tid ... of the descending-length order: a long list and a short
    // one) and their list boundaries: loaded ONCE, next to the staging loads -- a chain of three dependent L2 round trips
    // (perm -> start -> Q row) at the top of every point was most of the walk's time
    int pi[2], pk0[2], pk1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int base = u * 1024, len = min(1024, n - base);
        const bool in = tid < len;
        const int pos = in ? (u ? base + len - 1 - tid : base + tid) : 0;
        pi[u] = in ? pb[pos] : -1;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        pk0[u] = pi[u] >= 0 ? sb[pi[u]] : 0;
        pk1[u] = pi[u] >= 0 ? sb[pi[u] + 1] : 0;
    }
    const float *Qb = a.Q + (long long)b * n * a.ldq + sl * kBwdCh;
    float *dQb = a.dQ + (long long)b * n * a.lddq + sl * kBwdCh;
    float4 qn[2];                                                        // Q rows of the pass to come
#pragma unroll
    for (int u = 0; u < 2; ++u) qn[u] = *reinterpret_cast<const float4 *>(Qb + (long long)(pi[u] >= 0 ? pi[u] : 0) * a.ldq);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                                 // the previous pass has left the LDS slice
        if (h == pass) {
#pragma unroll
            for (int r = 0; r < kBwdGP; ++r) {
                const int j = (tid >> 1) + 512 * r;
                if (j < m) {
                    ec_ca[j] = sctr[r];
                    ec_ca[m + 1 + j] = sa[r];
                    ar[j] = sarg[r];
                }
            }
        }
        const float4 qc[2] = {qn[0], qn[1]};
        if (pass == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                qn[u] = *reinterpret_cast<const float4 *>(Qb + (long long)(pi[u] >= 0 ? pi[u] : 0) * a.ldq + 4);
        }
        __syncthreads();
        EC_T(1 + 2 * pass);                                              // barriers + LDS fill
        const int c0 = sl * kBwdCh + 4 * pass;
        const float4 wq = *reinterpret_cast<const float4 *>(a.q + c0);
        const float4 wt = *reinterpret_cast<const float4 *>(a.t + c0);
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int i = u ? pi[1] : pi[0];
            if (i < 0) continue;
            const int k0 = u ? pk0[1] : pk0[0], k1 = u ? pk1[1] : pk1[0];
            const float4 qi = u ? qc[1] : qc[0];
            f2 sc0 = {0.f, 0.f}, sc1 = {0.f, 0.f};                       // sums of Ctr, channels (0 1) (2 3)
            float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);                  // arg-row sums
            // an entry reads its group's Ctr row and arg word; the a row ONLY when one of the four arg bytes is its slot
            // (18 % of the lanes: the LDS moves bytes for active lanes only, and random-row reads are what bounds the walk)
            auto entry = [&](unsigned code) {
                const unsigned g = code >> sbits, s = code & smask;
                const float4 x = ec_ca[g];
                const unsigned z = ar[g] ^ (s * 0x01010101u);             // a zero byte where arg == slot
                sc0 += f2{x.x, x.y};
                sc1 += f2{x.z, x.w};
                if ((z - 0x01010101u) & ~z & 0x80808080u) {
                    const float4 av = ec_ca[m + 1 + g];
                    sv.x += (z & 0x000000ffu) == 0u ? av.x : 0.f;
                    sv.y += (z & 0x0000ff00u) == 0u ? av.y : 0.f;
                    sv.z += (z & 0x00ff0000u) == 0u ? av.z : 0.f;
                    sv.w += (z & 0xff000000u) == 0u ? av.w : 0.f;
                }
            };
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                unsigned cd[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) cd[v] = cs16[k + v];
#pragma unroll
                for (int v = 0; v < 4; ++v) entry(cd[v]);
            }
            for (; k < k1; ++k) entry(cs16[k]);
            const float kf = (float)(k1 - k0);
            float4 d;
            d.x = fmaf(wq.x, fmaf(kf, qi.x, sc0.x), fmaf(kf, wt.x, sv.x));
            d.y = fmaf(wq.y, fmaf(kf, qi.y, sc0.y), fmaf(kf, wt.y, sv.y));
            d.z = fmaf(wq.z, fmaf(kf, qi.z, sc1.x), fmaf(kf, wt.z, sv.z));
            d.w = fmaf(wq.w, fmaf(kf, qi.w, sc1.y), fmaf(kf, wt.w, sv.w));
            *reinterpret_cast<float4 *>(dQb + (long long)i * a.lddq + 4 * pass) = d;
        }
        EC_T(2 + 2 * pass);                                              // wave 0's walk of the pass
    }
#ifdef PCOPS_EC_PROF
    __syncthreads();
    EC_T(5);                                                             // the other waves' tail
    if (threadIdx.x == 0) atomicAdd(&g_ec_prof[7], 1ull);
#endif
}

size_t ec_bwd_lds_bytes(int m, int s) {
    return ((size_t)m + 1) * 32 + (((size_t)m + 1) * 4 + 15) / 16 * 16 + (((size_t)m * s * 2 + 15) & ~(size_t)15) + 16;
}
// ---- first EdgeConv layer of a stack whose INPUT needs no gradient (DGCNN's T-Net: the raw cloud) -------------------
// The layer is linear in the six edge-feature channels e = [x_g | x_j - x_g] (j = idx[g, s]):  Y1 = e W + b, so its weight
// gradient is  dW = p (E^T Gm) + q (E^T E W + E^T 1 b) + t E^T 1  with dY1 = p Gm + q Y1 + t  (gather.hip
// xyz_first_layer_grads_kernel states the 3-input form): ONE streaming pass over the masked gradient Gm for E^T Gm and a
// tiny pass over the edges for the 27 moments, instead of the scatter to per-point dQ / dCtr (two passes over Gm, one of
// them a gather through an inverse index) and the GEMM backward behind it.
constexpr int kEdgeMom = 27;        // 21 second moments (upper triangle, row-major) + 6 first moments

__global__ __launch_bounds__(256) void edge_moments_kernel(int n, int m, int S, long long rows, const float *__restrict__ x,
                                                           const int *__restrict__ idx, float *__restrict__ part,
                                                           float *__restrict__ e8) {
    __shared__ float red[4][kEdgeMom];
    float a[kEdgeMom];
#pragma unroll
    for (int i = 0; i < kEdgeMom; ++i) a[i] = 0.f;
    const long long mS = (long long)m * S;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
        const long long b = r / mS;
        const int g = (int)((r - b * mS) / S);
        const float *xg = x + ((long long)b * n + g) * 3, *xj = x + ((long long)b * n + idx[r]) * 3;
        float e[6];
        e[0] = xg[0]; e[1] = xg[1]; e[2] = xg[2];
        e[3] = xj[0] - e[0]; e[4] = xj[1] - e[1]; e[5] = xj[2] - e[2];
        if (e8) {        // the rows themselves, 32 bytes each, for the one-pass backward of the layer above (mlp.hip SIDE)
            *reinterpret_cast<float4 *>(e8 + r * 8) = make_float4(e[0], e[1], e[2], e[3]);
            *reinterpret_cast<float4 *>(e8 + r * 8 + 4) = make_float4(e[4], e[5], 0.f, 0.f);
        }
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { a[k] = fmaf(e[i], e[j], a[k]); ++k; }
#pragma unroll
        for (int i = 0; i < 6; ++i) a[21 + i] += e[i];
    }
#pragma unroll
    for (int i = 0; i < kEdgeMom; ++i) {
        float v = a[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kEdgeMom)
        part[(long long)blockIdx.x * kEdgeMom + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// E^T Gm, partial per workgroup: [gridDim][6][C].  A lane set of C / 4 lanes per row (float4 of Gm each), 256 / (C / 4) rows
// per step, four steps in flight; the six edge values of a row are computed by every lane of its set (broadcast loads).
template <int LPR>     // lanes per row = C / 4 (16 or 32)
__global__ __launch_bounds__(256) void edge_first_wgrad_kernel(int n, int m, int S, long long rows, const float *__restrict__ G,
                                                               const float *__restrict__ x, const int *__restrict__ idx,
                                                               float *__restrict__ part) {
    constexpr int RW = 256 / LPR, C = 4 * LPR, U = 4;
    extern __shared__ float ew_red[];                   // [RW][6][C]
    const int tid = threadIdx.x, rl = tid / LPR, cq = (tid % LPR) * 4;
    float acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    const long long mS = (long long)m * S;
    const long long step = (long long)gridDim.x * RW * U;
    for (long long r0 = (long long)blockIdx.x * RW * U + rl; r0 < rows; r0 += step) {
        float4 gv[U];
        int jj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = r0 + (long long)u * RW;
            const bool in = r < rows;
            gv[u] = in ? *reinterpret_cast<const float4 *>(G + r * C + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
            jj[u] = in ? idx[r] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = r0 + (long long)u * RW;
            if (r >= rows) continue;
            const long long b = r / mS;
            const int g = (int)((r - b * mS) / S);
            const float *xg = x + ((long long)b * n + g) * 3, *xj = x + ((long long)b * n + jj[u]) * 3;
            float e[6];
            e[0] = xg[0]; e[1] = xg[1]; e[2] = xg[2];
            e[3] = xj[0] - e[0]; e[4] = xj[1] - e[1]; e[5] = xj[2] - e[2];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc[i][0] = fmaf(e[i], gv[u].x, acc[i][0]); acc[i][1] = fmaf(e[i], gv[u].y, acc[i][1]);
                acc[i][2] = fmaf(e[i], gv[u].z, acc[i][2]); acc[i][3] = fmaf(e[i], gv[u].w, acc[i][3]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        *reinterpret_cast<float4 *>(&ew_red[(rl * 6 + i) * C + cq]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    __syncthreads();
    for (int e = tid; e < 6 * C; e += 256) {
        float v = 0.f;
        for (int r = 0; r < RW; ++r) v += ew_red[r * 6 * C + e];
        part[(long long)blockIdx.x * 6 * C + e] = v;
    }
}

//   dW[i][c] = p A[i][c] + q B[i][c] + t S[i],  B = M W + S b,  db[c] = p sumG + q mean rows + t rows      (all sums in double)
__global__ __launch_bounds__(1024) void edge_first_grads_kernel(int P1, const float *__restrict__ wpart, int P2,
                                                                const float *__restrict__ mpart, int C,
                                                                const float *__restrict__ W, const float *__restrict__ bias,
                                                                const float *__restrict__ p, const float *__restrict__ q,
                                                                const float *__restrict__ t, const float *__restrict__ sumG,
                                                                const float *__restrict__ mean, double rows,
                                                                float *__restrict__ dW, float *__restrict__ dbias) {
    __shared__ double smA[6][32][32];
    __shared__ double smM[kEdgeMom][32];
    __shared__ double mom[kEdgeMom];
    const int tid = threadIdx.x, cl = tid & 31, g = tid >> 5;
    const int c = blockIdx.x * 32 + cl;
    if (tid < kEdgeMom * 32) {
        const int l = tid / kEdgeMom, k = tid % kEdgeMom;
        double a = 0.0;
        for (int r = l; r < P2; r += 32) a += (double)mpart[(long long)r * kEdgeMom + k];
        smM[k][l] = a;
    }
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < C)
        for (int r = g; r < P1; r += 32)
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] += (double)wpart[((long long)r * 6 + i) * C + c];
#pragma unroll
    for (int i = 0; i < 6; ++i) smA[i][g][cl] = a[i];
    __syncthreads();
    if (tid < kEdgeMom) {
        double v = 0.0;
        for (int l = 0; l < 32; ++l) v += smM[tid][l];
        mom[tid] = v;
    }
    __syncthreads();
    if (g != 0 || c >= C) return;
    double A[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int l = 0; l < 32; ++l)
#pragma unroll
        for (int i = 0; i < 6; ++i) A[i] += smA[i][l][cl];
    const double b = bias ? (double)bias[c] : 0.0;
    const double pc = p[c], qc = q[c], tc = t[c];
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        double B = mom[21 + i] * b;
#pragma unroll 1
        for (int j = 0; j < 6; ++j) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            B += mom[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)] * (double)W[j * C + c];     // upper triangle, row-major
        }
        dW[i * C + c] = (float)(pc * A[i] + qc * B + tc * mom[21 + i]);
    }
    if (dbias) dbias[c] = (float)(pc * (double)sumG[c] + qc * ((double)mean[c] * rows) + tc * rows);
}

bool ec_shape_ok(int b, int n, int m, int s, int c) {
    // 64-channel slices, whole 64-group chunks, 8-bit slots, 32-bit byte offsets into Q (row stride up to 2 c) / the cloud's G rows
    return c >= 64 && c % 64 == 0 && m >= kGB && m % kGB == 0 && s >= 1 && s <= 128 && n >= 1 &&
           (long long)b * n * c * 8 < (1ll << 32) && ((long long)m * s + 256) * c * 4 < (1ll << 32) && m < (1 << 23) &&
           (long long)b * (c / 64) * ((n + kGB - 1) / kGB) < (1ll << 31) && (long long)b * (c / 64) * (m / kGB) < (1ll << 31);
}

}  // namespace

bool ec_bwd_fused_ok(int n, int m, int s, int c) {
    // OFF by default: measured 523 us against 478 us for the two kernels at (256, 2048, k = 20, C = 64).  The phase split
    // (tools/r5_ecprof.py, profiles/r05_ec_bwd_fused_phases.txt): 75 000 of a workgroup's 138 000 cycles are the STAGING of its
    // 8-channel slice -- 32 bytes out of every 256-byte row of four matrices, ~10 000 L1 misses per workgroup at ~0.13 lines
    // per cycle and CU -- and two 4-channel walks take 27 000 each (VALU issue: 27 instructions per entry).  The walk itself
    // is what was hoped for (no atomics: 183 us per layer for both terms); row-major operands sliced this thin are not.
    static const bool on = [] { const char *e = getenv("PCOPS_EDGECONV_BWD_FUSED"); return e && e[0] == '1'; }();
    return on && c % kBwdCh == 0 && m <= 512 * kBwdGP && s <= 255 && ec_codes16_ok(m, s) && ec_bwd_lds_bytes(m, s) <= kLdsMax &&
           n >= 1 && n <= 2048;
}

int ec_bwd_fused(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const float *gpool,
                 const float *ysel, const float *SQ, const unsigned char *arg, const float *scale, const float *shift,
                 const float *p, const float *q, const float *t, const void *workspace, float *dQ, int lddq, float *dCtr,
                 int lddc, hipStream_t st) {
    const unsigned *order = static_cast<const unsigned *>(workspace);
    const int *start = reinterpret_cast<const int *>(order + (size_t)b * m * s);
    const int *perm = start + (size_t)b * (n + 1);
    const unsigned short *codes = reinterpret_cast<const unsigned short *>(perm + (size_t)b * n);
    BwdLdsArgs a = {n, m, s, c, ec_sbits(s), ldq, ldc, lddq, lddc, Q, Ctr, gpool, ysel, SQ, arg, scale, shift, p, q, t,
                    start, perm, codes, dQ, dCtr};
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(ec_bwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    hipLaunchKernelGGL(ec_bwd_lds_kernel, dim3((unsigned)b * (c / kBwdCh)), dim3(1024), ec_bwd_lds_bytes(m, s), st, a);
    return pcops_launch_status();
}


constexpr int kEdgeFirstGrid = 2048;     // partial rows of the two passes below
int ec_edge_first_rows() { return kEdgeFirstGrid; }
bool ec_edge_first_supported(int b, int n, int m, int s, int c) {
    // the kernels take the centre of group g as x[cloud][g]: an EdgeConv graph, one group per source point (m == n)
    return b >= 1 && n >= 1 && m == n && s >= 1 && (c == 64 || c == 128);
}
int ec_edge_first_moments(int b, int n, int m, int s, const float *x, const int *idx, float *part, float *e8, hipStream_t st) {
    const long long rows = (long long)b * m * s;
    hipLaunchKernelGGL(edge_moments_kernel, dim3(kEdgeFirstGrid), dim3(256), 0, st, n, m, s, rows, x, idx, part, e8);
    return pcops_launch_status();
}
int ec_edge_first_wgrad(int b, int n, int m, int s, int c, const float *G, const float *x, const int *idx, float *part,
                        hipStream_t st) {
    const long long rows = (long long)b * m * s;
    if (c == 64)
        hipLaunchKernelGGL(edge_first_wgrad_kernel<16>, dim3(kEdgeFirstGrid), dim3(256), (size_t)16 * 6 * 64 * 4, st, n, m, s, rows, G, x,
                           idx, part);
    else if (c == 128)
        hipLaunchKernelGGL(edge_first_wgrad_kernel<32>, dim3(kEdgeFirstGrid), dim3(256), (size_t)8 * 6 * 128 * 4, st, n, m, s, rows, G, x,
                           idx, part);
    else
        return PCOPS_ERR_UNSUPPORTED;
    return pcops_launch_status();
}
int ec_edge_first_grads(int P1, const float *wpart, int P2, const float *mpart, int c, const float *W, const float *bias,
                        const float *p, const float *q, const float *t, const float *sumG, const float *mean, long long rows,
                        float *dW, float *dbias, hipStream_t st) {
    hipLaunchKernelGGL(edge_first_grads_kernel, dim3((c + 31) / 32), dim3(1024), 0, st, P1, wpart, P2, mpart, c, W, bias, p, q, t,
                       sumG, mean, (double)rows, dW, dbias);
    return pcops_launch_status();
}

#ifdef PCOPS_EC_PROF
// diagnostics build (tools/build_variant.sh ecprof "-DPCOPS_EC_PROF=1"): phase cycle sums of ec_bwd_lds_kernel's lane 0
extern "C" int pcops_ec_debug_prof(unsigned long long *out8) {
    if (hipDeviceSynchronize() != hipSuccess) return PCOPS_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ec_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return PCOPS_ERR_LAUNCH;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ec_prof), z, sizeof(z)) != hipSuccess) return PCOPS_ERR_LAUNCH;
    return PCOPS_OK;
}
#endif

bool ec_enabled() {
    static const bool on = [] { const char *e = getenv("PCOPS_EDGECONV_R5"); return !(e && e[0] == '0'); }();
    return on;
}

bool ec_fwd_supported(int b, int n, int m, int s, int c) { return ec_enabled() && ec_shape_ok(b, n, m, s, c); }

int ec_stats_rows(long long G) { return (int)((G + kGB - 1) / kGB); }

// rows of partial statistics pcops_edge_pool_fwd writes on this path: one per cloud (LDS slices) or one per 64 groups
int ec_edge_pool_stats_rows(int b, int n, int m) {
    return (ec_lds_on() && ec_lds_fwd_ok(n)) ? b : ec_stats_rows((long long)b * m);
}

int ec_edge_pool_fwd(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const int *idx,
                     const float *gamma, float *SQ, float *qsel, unsigned char *arg, float *stats, const float *pivot,
                     hipStream_t st) {
    FwdArgs a = {b, n, m, s, c, ldq, ldc, Q, Ctr, idx, gamma, SQ, qsel, arg, nullptr, stats, pivot};
    if (ec_lds_on() && ec_lds_fwd_ok(n)) {
        const size_t lds = (size_t)n * 64 + 16 * 2 * kSliceCh * sizeof(float);
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ec_fwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kLdsMax) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        hipLaunchKernelGGL(ec_fwd_lds_kernel, dim3((unsigned)b * (c / kSliceCh)), dim3(1024), lds, st, a);
        return pcops_launch_status();
    }
    const unsigned grid = (unsigned)((long long)b * (c / 64) * (m / kGB));
    const size_t lds = ((size_t)kGB * s + kSets * 128) * sizeof(float);
    hipLaunchKernelGGL(ec_fwd_kernel<0>, dim3(grid), dim3(256), lds, st, a);
    return pcops_launch_status();
}

int ec_gather_fwd(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const int *idx,
                  float *Y, float *stats, const float *pivot, hipStream_t st) {
    static const bool nt_on = [] { const char *e = getenv("PCOPS_NT_STORE"); return !(e && e[0] == '0'); }();   // kernel A/B only
    FwdArgs a = {b, n, m, s, c, ldq, ldc, Q, Ctr, idx, nullptr, nullptr, nullptr, nullptr, Y, stats, pivot,
                 (nt_on && (long long)b * m * s * c * 4 >= (256ll << 20)) ? 1 : 0};
    const unsigned grid = (unsigned)((long long)b * (c / 64) * (m / kGB));
    const size_t lds = ((size_t)kGB * s + kSets * 128) * sizeof(float);
    hipLaunchKernelGGL(ec_fwd_kernel<1>, dim3(grid), dim3(256), lds, st, a);
    return pcops_launch_status();
}

bool ec_bwd_supported(int b, int n, int m, int s, int c) {
    // (the inverse index keeps order | start | perm inside the caller's workspace of 4 b m s + b (n + 1) + 2 ints)
    return ec_enabled() && ec_shape_ok(b, n, m, s, c) && n <= 16384 && s <= 256 && 2ll * m * s >= n;
}

// workspace: order (b m s u32) | start (b (n + 1) int32) | perm (b n int32) | codes (b m s u16) -- inside what pcops_sa_scatter_workspace_bytes asks for
int ec_csr_build(int b, int n, int m, int s, const int *idx, void *workspace, hipStream_t st) {
    unsigned *order = static_cast<unsigned *>(workspace);
    int *start = reinterpret_cast<int *>(order + (size_t)b * m * s);
    int *perm = start + (size_t)b * (n + 1);
    unsigned short *codes = reinterpret_cast<unsigned short *>(perm + (size_t)b * n);
    const size_t lds = ((size_t)n + 2048) * sizeof(int);
    const int sbits = ec_sbits(s);
    const bool stage = ec_codes16_ok(m, s) && lds + (size_t)m * s * 2 <= 160 * 1024;
    if (stage) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ec_csr_build_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        hipLaunchKernelGGL(ec_csr_build_kernel<true>, dim3(b), dim3(1024), lds + (size_t)m * s * 2, st, n, m, s, sbits, idx, order,
                           start, perm, codes);
    } else {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ec_csr_build_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        hipLaunchKernelGGL(ec_csr_build_kernel<false>, dim3(b), dim3(1024), lds, st, n, m, s, sbits, idx, order, start, perm,
                           codes);
    }
    return pcops_launch_status();
}

int ec_walk(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const float *G,
            const float *p, const float *q, const float *t, const void *workspace, float *dQ, int lddq, hipStream_t st) {
    const unsigned *order = static_cast<const unsigned *>(workspace);
    const int *start = reinterpret_cast<const int *>(order + (size_t)b * m * s);
    WalkArgs a = {b, n, m, s, c, ldq, ldc, lddq, Q, Ctr, G, p, q, t, order, start, dQ};
    if (!G && ec_lds_on() && ec_lds_walk_ok(n, m, s) && c % kWalkCh == 0) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ec_walk_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kLdsMax) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        const int *perm = start + (size_t)b * (n + 1);
        const unsigned short *codes = reinterpret_cast<const unsigned short *>(perm + (size_t)b * n);
        WalkLdsArgs wa = {a, perm, codes, ec_sbits(s)};
        hipLaunchKernelGGL(ec_walk_lds_kernel, dim3((unsigned)b * (c / kWalkCh)), dim3(1024), ec_lds_walk_bytes(n, m, s), st, wa);
        return pcops_launch_status();
    }
    const unsigned grid = (unsigned)((long long)b * (c / 64) * ((n + kGB - 1) / kGB));
    if (G) hipLaunchKernelGGL(ec_walk_kernel<true>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(ec_walk_kernel<false>, dim3(grid), dim3(256), 0, st, a);
    return pcops_launch_status();
}

int ec_tnet_ctr(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const float *G,
                const int *idx, const float *p, const float *q, const float *t, float *dCtr, int lddc, hipStream_t st) {
    CtrArgs a = {b, n, m, s, c, ldq, ldc, lddc, Q, Ctr, G, idx, p, q, t, dCtr};
    const unsigned grid = (unsigned)((long long)b * (c / 64) * (m / kGB));
    hipLaunchKernelGGL(ec_tnet_ctr_kernel, dim3(grid), dim3(256), (size_t)kGB * s * sizeof(unsigned), st, a);
    return pcops_launch_status();
}

bool ec_sparse_ok(int n) { return (size_t)n * kSparseSlice * sizeof(float) <= kLdsMax; }

int ec_sparse(int b, int n, int m, int s, int c, const float *gpool, const float *ysel, const float *SQ, const float *Ctr,
              int ldc, const unsigned char *arg, const int *idx, const float *scale, const float *shift, const float *p,
              const float *q, const float *t, float *dCtr, int lddc, float *dQ, int lddq, hipStream_t st) {
    SparseArgs a = {n, m, s, c, ldc, lddq, lddc, gpool, ysel, SQ, Ctr, arg, idx, scale, shift, p, q, t, dCtr, dQ};
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(ec_sparse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    hipLaunchKernelGGL(ec_sparse_kernel, dim3((unsigned)b * (c / kSparseSlice)), dim3(1024), (size_t)n * kSparseSlice * sizeof(float),
                       st, a);
    return pcops_launch_status();
}

// ---- the concatenated EdgeConv weight ---------------------------------------------------------------------------------
// W1 (2 c, cp) = [W_a ; W_b] (rows 0..c-1 multiply x_i, rows c..2c-1 multiply x_j - x_i)  ->  Wcat (kp, 2 cp) =
// [W_b | W_a - W_b] with rows c..kp-1 zero (the input is zero-padded to kp channels), bcat (2 cp) = [0 | b1].
// Backward: dW_a = dWcat[:, cp:], dW_b = dWcat[:, :cp] - dWcat[:, cp:], db1 = dbcat[cp:].  One launch each instead of the
// slice / subtract / pad / concatenate chain (and, backward, the zero-fill + copy + add of every slice).
namespace {
__global__ __launch_bounds__(256) void edge_weights_fwd_kernel(int c, int cp, int kp, const float *__restrict__ W1,
                                                               const float *__restrict__ b1, float *__restrict__ Wcat,
                                                               float *__restrict__ bcat) {
    const int total = kp * 2 * cp;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total + 2 * cp; e += gridDim.x * 256) {
        if (e >= total) {
            const int j = e - total;
            bcat[j] = j < cp ? 0.f : (b1 ? b1[j - cp] : 0.f);
            continue;
        }
        const int r = e / (2 * cp), j = e - r * 2 * cp;
        float v = 0.f;
        if (r < c) v = j < cp ? W1[(c + r) * cp + j] : W1[r * cp + (j - cp)] - W1[(c + r) * cp + (j - cp)];
        Wcat[e] = v;
    }
}
__global__ __launch_bounds__(256) void edge_weights_bwd_kernel(int c, int cp, const float *__restrict__ dWcat,
                                                               const float *__restrict__ dbcat, float *__restrict__ dW1,
                                                               float *__restrict__ db1) {
    const int total = 2 * c * cp;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total + cp; e += gridDim.x * 256) {
        if (e >= total) {
            if (db1) db1[e - total] = dbcat[cp + (e - total)];
            continue;
        }
        const int r = e / cp, j = e - r * cp;
        dW1[e] = r < c ? dWcat[r * 2 * cp + cp + j] : dWcat[(r - c) * 2 * cp + j] - dWcat[(r - c) * 2 * cp + cp + j];
    }
}
}  // namespace

extern "C" int pcops_edge_weights_fwd(int c, int cp, int kp, const float *W1, const float *b1, float *Wcat, float *bcat,
                                      pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(c >= 1 && cp >= 1 && kp >= c);
    PCOPS_REQUIRE_PTR(W1); PCOPS_REQUIRE_PTR(Wcat); PCOPS_REQUIRE_PTR(bcat);
    const int total = kp * 2 * cp + 2 * cp;
    hipLaunchKernelGGL(edge_weights_fwd_kernel, dim3(cdiv(total, 256) < 256u ? cdiv(total, 256) : 256u), dim3(256), 0,
                       as_stream(stream), c, cp, kp, W1, b1, Wcat, bcat);
    return pcops_launch_status();
}

extern "C" int pcops_edge_weights_bwd(int c, int cp, const float *dWcat, const float *dbcat, float *dW1, float *db1,
                                      pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(c >= 1 && cp >= 1);
    PCOPS_REQUIRE_PTR(dWcat); PCOPS_REQUIRE_PTR(dW1);
    if (db1) PCOPS_REQUIRE_PTR(dbcat);
    const int total = 2 * c * cp + cp;
    hipLaunchKernelGGL(edge_weights_bwd_kernel, dim3(cdiv(total, 256) < 256u ? cdiv(total, 256) : 256u), dim3(256), 0,
                       as_stream(stream), c, cp, dWcat, dbcat, dW1, db1);
    return pcops_launch_status();
}
