// knn.hip -- DGCNN dynamic-graph ops for gfx950: pairwise_distance, knn (top-k), the fused
// knn_graph that never materialises the (n,n) adjacency, and get_edge_feature(+grad).
//
// The reference has no native code here: dgcnn/utils/tf_util.py:638-706 builds the graph in
// TF-Python (matmul + top_k + gather) and writes a (B,N,N) fp32 matrix five times per forward.
// Arithmetic contract (shared with oracle/pcops_oracle.c, "parity unpinned" w.r.t. TensorFlow):
//   inner_ij = fmaf chain over c ascending from +0, s_i likewise, D_ij = (s_i + (-2*inner)) + s_j;
//   top-k ascending in D, ties -> lower j.  Zero-padding the channel axis is exact
//   (fmaf(0,0,acc) == acc), which lets the fused kernel keep the query row in VGPRs at a
//   compile-time width.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------ fused kNN graph
// grid (query tiles, clouds); one lane per query row, the row's channels in VGPRs; candidate
// rows staged TJ at a time in LDS and read wave-uniformly (ds_read_b128 broadcast); each lane's
// sorted top-k list lives in LDS as [slot][lane] (conflict free).
template <int C, int TJ>
__global__ __launch_bounds__(256) void knn_graph_kernel(int n, int c, int k,
                                                        const float *__restrict__ x,
                                                        int *__restrict__ nn_idx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nthr = blockDim.x;
    float *cand = lds;                    // [TJ][C]
    float *sj = cand + TJ * C;            // [TJ]
    float *lv = sj + TJ;                  // [k][nthr]
    int *li = reinterpret_cast<int *>(lv + k * nthr);

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const float *xb = x + (size_t)b * n * c;
    const int i = blockIdx.x * nthr + tid;
    const bool valid = i < n;

    float xi[C];
#pragma unroll
    for (int l = 0; l < C; ++l) xi[l] = (valid && l < c) ? xb[(size_t)i * c + l] : 0.f;
    float si = 0.f;
#pragma unroll
    for (int l = 0; l < C; ++l) si = fmaf(xi[l], xi[l], si);

    for (int s = 0; s < k; ++s) {
        lv[s * nthr + tid] = INFINITY;
        li[s * nthr + tid] = 0;
    }
    float worst = INFINITY;

    for (int j0 = 0; j0 < n; j0 += TJ) {
        const int tn = min(TJ, n - j0);
        __syncthreads();
        for (int e = tid; e < tn * C; e += nthr) {
            const int r = e / C, l = e - r * C;
            cand[e] = l < c ? xb[(size_t)(j0 + r) * c + l] : 0.f;
        }
        __syncthreads();
        for (int r = tid; r < tn; r += nthr) {
            float s = 0.f;
            for (int l = 0; l < C; ++l) s = fmaf(cand[r * C + l], cand[r * C + l], s);
            sj[r] = s;
        }
        __syncthreads();
        for (int jj = 0; jj < tn; ++jj) {
            const float *cr = cand + jj * C;
            float inner = 0.f;
#pragma unroll
            for (int l = 0; l < C; l += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(cr + l);
                inner = fmaf(xi[l + 0], v.x, inner);
                inner = fmaf(xi[l + 1], v.y, inner);
                inner = fmaf(xi[l + 2], v.z, inner);
                inner = fmaf(xi[l + 3], v.w, inner);
            }
            const float d = (si + (-2.f * inner)) + sj[jj];
            if (valid && d < worst) {
                int pos = k - 1;
                while (pos > 0) {
                    const float pv = lv[(pos - 1) * nthr + tid];
                    if (!(d < pv)) break;
                    lv[pos * nthr + tid] = pv;
                    li[pos * nthr + tid] = li[(pos - 1) * nthr + tid];
                    --pos;
                }
                lv[pos * nthr + tid] = d;
                li[pos * nthr + tid] = j0 + jj;
                worst = lv[(k - 1) * nthr + tid];
            }
        }
    }
    if (valid) {
        int *o = nn_idx + ((size_t)b * n + i) * k;
        for (int s = 0; s < k; ++s) o[s] = li[s * nthr + tid];
    }
}

template <int C>
int launch_knn_graph(int b, int n, int c, int k, const float *x, int *nn_idx, hipStream_t st) {
    constexpr int TJ = C <= 32 ? 128 : 64;
    int threads = n >= 256 ? 256 : ((n + kWave - 1) / kWave) * kWave;
    while (threads > 64 && (size_t)(TJ * C + TJ + 2 * k * threads) * 4 > 64 * 1024) threads -= 64;
    const size_t lds = (size_t)(TJ * C + TJ + 2 * k * threads) * 4;
    if (lds > 64 * 1024) return PCOPS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((knn_graph_kernel<C, TJ>), dim3(cdiv(n, threads), b), dim3(threads), lds, st,
                       n, c, k, x, nn_idx);
    return pcops_launch_status();
}

// ------------------------------------------------------------------ fused kNN graph on the matrix cores
// D = -2 X X^T + s_i + s_j on v_mfma_f32_32x32x2_f32: gfx950's f32 MFMA is bit-for-bit a k-ordered fmaf chain,
// i.e. exactly the arithmetic contract above (inner = fmaf chain over c ascending from +0), so the indices stay
// bit-exact w.r.t. the oracle while the VALU is left free for the top-k bookkeeping.
//   * workgroup = 4 waves = 128 query rows of one cloud; a wave owns 32 queries whose channels sit in VGPRs as
//     MFMA B fragments (query = lane & 31, channel parity = lane >> 5);
//   * candidates stream through LDS in chunks of 128 rows (row stride CP+1 -> conflict-free column reads);
//     a 32-candidate x 32-query tile costs CP/2 MFMAs; candidates are the tile ROWS, so each lane sees 16
//     candidates of ONE query per tile, in ascending index order -> strict '<' keeps the lower index on ties;
//   * each lane keeps a sorted top-KL list in registers: insertion is one v_med3_f32 + one compare + two selects
//     per slot; the two half-waves of a query (rows r and r+4 interleaved) are merged at the end with a
//     (distance, index) lexicographic insertion.
typedef float f32x16k __attribute__((ext_vector_type(16)));

template <int KL>
struct TopK {
    float v[KL];
    int ix[KL];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < KL; ++s) { v[s] = INFINITY; ix[s] = 0; }
    }
    // candidates arrive in ascending index order: strict '<' on the value alone keeps the lower index first
    __device__ __forceinline__ void insert_ascending(float d, int j) {
        bool c_hi = d < v[KL - 1];
#pragma unroll
        for (int s = KL - 1; s >= 1; --s) {
            const bool c_lo = d < v[s - 1];
            const float nv = __builtin_amdgcn_fmed3f(v[s - 1], d, v[s]);
            ix[s] = c_lo ? ix[s - 1] : (c_hi ? j : ix[s]);
            v[s] = nv;
            c_hi = c_lo;
        }
        if (c_hi) { v[0] = d; ix[0] = j; }
    }
    // general insertion: (distance, index) lexicographic
    __device__ __forceinline__ void insert_lex(float d, int j) {
        bool c_hi = d < v[KL - 1] || (d == v[KL - 1] && j < ix[KL - 1]);
#pragma unroll
        for (int s = KL - 1; s >= 1; --s) {
            const bool c_lo = d < v[s - 1] || (d == v[s - 1] && j < ix[s - 1]);
            const float nv = c_lo ? v[s - 1] : (c_hi ? d : v[s]);
            ix[s] = c_lo ? ix[s - 1] : (c_hi ? j : ix[s]);
            v[s] = nv;
            c_hi = c_lo;
        }
        if (c_hi) { v[0] = d; ix[0] = j; }
    }
};

// Round 5: the sorted list as ONE 64-bit key per slot, ordered as an IEEE double, so that a sorted insertion is
// v_max_f64 + v_min_f64 per slot (2 instructions, fp64 min / max run at the fp32 rate on this chip) instead of
// compare + med3 + two selects (4), and the (distance, index) tie rule is part of the order itself:
//   key = bits((double)d)  [exact; the low 29 mantissa bits of a converted float are zero]
//         | index in those 29 bits  (complemented when d < 0: a negative double grows DOWNWARDS with its magnitude)
//         + 512 in the exponent field  (a monotone shift that keeps d = +0 -- every query's distance to itself -- and
//           fp32 denormals away from fp64 denormals: nothing depends on how min / max treat those).
// Keys of distinct candidates are distinct, so  L[s] <- min(L[s], max(L[s-1], x))  is an exact sorted insertion whatever
// order the candidates arrive in.  Non-finite distances never enter (the old lists' strict '<' against +inf): they and
// the idle lanes of a flush round insert the sentinel, which is above every key; an unfilled slot reads back as
// (+inf, index 0) like TopK's.
__device__ __forceinline__ double key_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // (fmin() would canonicalise both operands first: 2 more)
    return r;
}
__device__ __forceinline__ double key_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int KL>
struct TopKey {
    static constexpr int kSentHi = 0x7FE00000, kBias = 0x20000000, kIdxMask = 0x1FFFFFFF;
    double key[KL];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < KL; ++s) key[s] = __hiloint2double(kSentHi, 0);
    }
    // accept: the lane has a candidate and its distance is finite (d < +inf)
    static __device__ __forceinline__ double make(float d, int j, bool accept) {
        const double kd = (double)d;
        int hi = __double2hiint(kd);
        const int t = hi >> 31;
        const int lo = __double2loint(kd) | ((j ^ t) & kIdxMask);
        hi = accept ? hi + kBias : kSentHi;
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ void insert(double x) {
#pragma unroll
        for (int s = KL - 1; s >= 1; --s) key[s] = key_min(key[s], key_max(key[s - 1], x));
        key[0] = key_min(key[0], x);
    }
    __device__ __forceinline__ float value(int s) const {
        const int hi = __double2hiint(key[s]);
        const float v = (float)__hiloint2double(hi - kBias, __double2loint(key[s]) & ~kIdxMask);
        return hi == kSentHi ? INFINITY : v;
    }
    __device__ __forceinline__ int index(int s) const {
        const int hi = __double2hiint(key[s]);
        return hi == kSentHi ? 0 : ((__double2loint(key[s]) ^ (hi >> 31)) & kIdxMask);
    }
};

#ifdef PCOPS_KNN_LIST32
// A/B build (tools/build_variant.sh list32 "-DPCOPS_KNN_LIST32=1"): the (value, index) register lists of rounds 2-4 behind
// TopKey's interface
template <int KL>
struct TopList {
    struct Entry { float d; int j; };
    TopK<KL> t;
    __device__ __forceinline__ void init() { t.init(); }
    static __device__ __forceinline__ Entry make(float d, int j, bool accept) { return Entry{accept ? d : INFINITY, j}; }
    __device__ __forceinline__ void insert(Entry e) { t.insert_ascending(e.d, e.j); }
    __device__ __forceinline__ void merge(Entry e) { t.insert_lex(e.d, e.j); }
    __device__ __forceinline__ Entry get(int s) const { return Entry{t.v[s], t.ix[s]}; }
    static __device__ __forceinline__ Entry from_lane(Entry e, int src) {
        return Entry{__shfl(e.d, src, 64), __shfl(e.j, src, 64)};
    }
    __device__ __forceinline__ float value(int s) const { return t.v[s]; }
    __device__ __forceinline__ int index(int s) const { return t.ix[s]; }
};
#else
template <int KL>
struct TopList : TopKey<KL> {
    typedef double Entry;
    __device__ __forceinline__ void merge(double x) { this->insert(x); }
    __device__ __forceinline__ double get(int s) const { return this->key[s]; }
    static __device__ __forceinline__ double from_lane(double e, int src) {
        return __hiloint2double(__shfl(__double2hiint(e), src, 64), __shfl(__double2loint(e), src, 64));
    }
};
#endif

// (Round 2 also tried pipelining the tiles INSIDE a wave -- the MFMA chain of tile t+1 issued two at a time between
// the selection instructions of tile t, fragments preloaded a group ahead, branch-free pushes: 2482 vs 2505 us on 64
// channels, 1090 vs 937 us on 3.  Per SIMD the time is close to (VALU + SALU + LDS instructions) x ~4.5 cycles PLUS the
// MFMA time -- 765 instructions per 32 x 32 tile -- so fewer instructions per distance is the lever, not more overlap.
// Two tiles per iteration with interleaved, independent MFMA chains: 2478 vs 2505 us (64 channels), 1107 vs 937 us (3).
// Round 3: one compare per distance shifted into a per-lane bitmap by its carry, the distances staged in LDS, the pushes in
// a loop over the set bits instead of sixteen exec-masked blocks: 3627 vs 2399 us -- the loop's LDS round trips are serial.)
// Candidates that beat a lane's current k-th distance are QUEUED in LDS (slot-major, conflict free) instead of being
// inserted at once: the per-candidate work is then one compare (+ two LDS writes for the ~5 % that pass), and the
// sorted insertion -- 4 VALU per list slot, the expensive part -- runs for whole batches when some lane's queue
// is nearly full.  The queue keeps arrival (= ascending index) order and the insertion re-checks against the
// up-to-date k-th value, so the result is that of inserting every candidate in order.
constexpr int kKnnQD = 12;             // queue slots per lane; a flush is due when any lane holds more than QD - 4

template <int CP, int KL>
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(int n, int c, int k, const float *__restrict__ x,
                                                          int *__restrict__ nn_idx, const int *__restrict__ seed) {
    constexpr int CH = 128;            // candidate rows per LDS chunk
    constexpr int LD = CP + 1;         // odd row stride: lanes 0..31 read 32 rows at one column without conflicts
    constexpr int NKK = CP / 2;        // MFMAs per tile
    constexpr int QD = kKnnQD;
    wave_prio_stagger();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cs = lds;                   // [CH][LD]
    float *sc = cs + CH * LD;          // [CH]   (16-byte aligned: CH * LD is a multiple of 4)
    float *qd = sc + CH;               // [QD][256] queued distances
    int *qj = reinterpret_cast<int *>(qd + QD * 256);   // [QD][256] and their indices
    const CloudPart cp = xcd_cloud_part();      // the query blocks of a cloud share its candidates: one XCD, one L2
    const int b = cp.cloud;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    const float *xb = x + (size_t)b * n * c;
    const int q = cp.part * 128 + wave * 32 + li;          // this lane's query row
    const bool qin = q < n;

    // query fragments: bq[kk] = x[q][2 kk + half]; squared norm by the contract's fmaf chain
    float bq[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int ch = 2 * kk + half;
        bq[kk] = (qin && ch < c) ? xb[(size_t)q * c + ch] : 0.f;
    }
    float sq = 0.f;
    if (qin)
        for (int l = 0; l < c; ++l) sq = fmaf(xb[(size_t)q * c + l], xb[(size_t)q * c + l], sq);

    // SEEDED threshold (pcops_knn_graph_seeded): `seed` names k DISTINCT points per query -- in DGCNN the neighbours of the
    // previous layer's graph.  The k-th smallest distance of the scan cannot exceed the largest distance to those k
    // points, so every candidate beyond  tau = max_s |x_q - x_seed(s)|^2  (+ a margin for the two summation orders) is
    // rejected by ONE compare before it reaches the queue: the sorted insertion (117 vector instructions per queued
    // candidate) is what the selection costs, and a query that starts from a threshold near its final one queues a
    // fraction of the ~k ln(n / k) candidates an unseeded scan does.  The result is the same list: the filter is an upper
    // bound, selection and tie rule are untouched.
    float tau = INFINITY;
    if (seed != nullptr) {
        // the two half-waves hold the same 32 queries: each takes every other seed, 16-byte loads when the rows allow
        const int *sd = seed + ((size_t)b * n + (qin ? q : 0)) * k;
        const bool v4 = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0);
        float worst = 0.f, smax = 0.f;
        // PRECONDITION (pcops.h): k distinct in-range indices per row.  It is CHECKED: a row with an index outside [0, n)
        // or with a repeated index does not bound the k-th distance (fewer than k distinct points), so it gives no bound
        // at all (tau stays +inf) instead of a clamped / too small one that would silently drop true neighbours
        bool seed_ok = true;
        int mine[(KL + 1) / 2];
#pragma unroll
        for (int u = 0; u < (KL + 1) / 2; ++u) {
            const int s = half + 2 * u;
            mine[u] = (s < k && qin) ? sd[s] : -1 - u - 64 * half;          // distinct negative fillers beyond k
            if (s < k && qin && (mine[u] < 0 || mine[u] >= n)) seed_ok = false;
        }
#pragma unroll
        for (int u = 0; u < (KL + 1) / 2; ++u) {
#pragma unroll
            for (int w = u + 1; w < (KL + 1) / 2; ++w) seed_ok = seed_ok && (mine[u] != mine[w]);
            const int theirs = __shfl_xor(mine[u], 32, 64);                 // the partner half-wave's seeds of this query
#pragma unroll
            for (int w = 0; w < (KL + 1) / 2; ++w) seed_ok = seed_ok && (theirs != mine[w]);
        }
        seed_ok = seed_ok && (__shfl_xor((int)seed_ok, 32, 64) != 0);
        for (int s = half; s < k && qin; s += 2) {
            int j = sd[s];
            j = j < 0 ? 0 : (j >= n ? n - 1 : j);                           // (address safety only; such a row is rejected above)
            const float *pq = xb + (size_t)q * c, *pj = xb + (size_t)j * c;
            float d = 0.f, sj = 0.f;
            if (v4) {
                for (int l = 0; l < c; l += 4) {
                    const float4 a = *reinterpret_cast<const float4 *>(pq + l);
                    const float4 bb = *reinterpret_cast<const float4 *>(pj + l);
                    d = fmaf(a.x - bb.x, a.x - bb.x, d); d = fmaf(a.y - bb.y, a.y - bb.y, d);
                    d = fmaf(a.z - bb.z, a.z - bb.z, d); d = fmaf(a.w - bb.w, a.w - bb.w, d);
                    sj = fmaf(bb.x, bb.x, sj); sj = fmaf(bb.y, bb.y, sj);
                    sj = fmaf(bb.z, bb.z, sj); sj = fmaf(bb.w, bb.w, sj);
                }
            } else {
                for (int l = 0; l < c; ++l) {
                    const float a = pq[l], bb = pj[l];
                    d = fmaf(a - bb, a - bb, d);
                    sj = fmaf(bb, bb, sj);
                }
            }
            worst = fmaxf(worst, d);
            smax = fmaxf(smax, sj);
        }
        worst = fmaxf(worst, __shfl_xor(worst, 32, 64));
        smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
        // the scan evaluates (s_q - 2 <x_q, x_j>) + s_j with c-term fmaf chains: within ~4 c eps max(s_q, s_j) of the
        // direct form above (c <= 128: 6e-5); the margin is an order of magnitude wider
        if (qin && seed_ok) tau = worst + 1e-3f * (sq + smax) + 1e-30f;
    }
    TopList<KL> top;
    top.init();
    int nq = 0;                        // entries in this lane's queue
    // SHARED threshold of the two half-waves (round 3): lanes l and l ^ 32 scan the two halves of the SAME query's
    // candidates, each with its own sorted list.  The ceil(KL/2) best of one half and the ceil(KL/2) best of the other
    // are >= KL distinct candidates, all <= tau2 = max of the two lists' ceil(KL/2)-th values: the query's KL-th smallest
    // distance cannot exceed tau2, and a candidate beyond it need not enter either list.  A half on its own only knows
    // its OWN KL-th value -- with it the two lists queue ~2 KL ln(n / 2 KL) candidates per query, with tau2 about half.
    // (<= keeps ties with the bound; the lists and the final merge are unchanged.)
    float tau2 = INFINITY;
    float thr = tau;                   // d < kth && d <= tau2 && d < tau as ONE compare (the list only changes in flush)
    auto flush = [&]() {
        const int mx = (int)wave_max_u32((unsigned)nq);
        for (int u = 0; u < mx; ++u)
            top.insert(TopList<KL>::make(qd[u * 256 + tid], qj[u * 256 + tid], u < nq));   // (queued distances are finite)
        nq = 0;
        const float hv = top.value((KL + 1) / 2 - 1);
        tau2 = fmaxf(hv, __shfl_xor(hv, 32, 64));
        thr = fminf(fminf(top.value(KL - 1), nextafterf(tau2, INFINITY)), tau);
    };

    // the candidate chunks are software pipelined: chunk i+1 travels global -> registers (16-byte loads when the rows
    // are 16-byte multiples) while the MFMAs and the top-k bookkeeping of chunk i run; the registers go to LDS at the
    // top of the next iteration.  (Loading a chunk right before it is needed left the matrix pipe idle 73 % of the time.)
    constexpr int NQ4 = CH * (CP / 4);                    // float4 per chunk
    constexpr int NV = (NQ4 + 255) / 256;                 // float4 per thread per chunk
    const bool vec = (c == CP) && (CP % 4 == 0) && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0);
    float4 pre[NV];
    auto fetch = [&](int j0n) {
        const int tnn = min(CH, n - j0n);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e4 = tid + 256 * u;                  // float4 index inside the chunk: row e4 / (CP/4)
            const int r = e4 / (CP / 4), l = (e4 - r * (CP / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e4 < NQ4 && r < tnn) {
                const float *src = xb + (size_t)(j0n + r) * c;
                if (vec) {
                    v = *reinterpret_cast<const float4 *>(src + l);
                } else {
                    v.x = l + 0 < c ? src[l + 0] : 0.f; v.y = l + 1 < c ? src[l + 1] : 0.f;
                    v.z = l + 2 < c ? src[l + 2] : 0.f; v.w = l + 3 < c ? src[l + 3] : 0.f;
                }
            }
            pre[u] = v;
        }
    };
    if (n > 0) fetch(0);
    for (int j0 = 0; j0 < n; j0 += CH) {
        const int tn = min(CH, n - j0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e4 = tid + 256 * u;
            const int r = e4 / (CP / 4), l = (e4 - r * (CP / 4)) * 4;
            if (e4 < NQ4) {
                float *dst = cs + r * LD + l;              // odd row stride: scalar stores
                dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
            }
        }
        __syncthreads();
        if (j0 + CH < n) fetch(j0 + CH);                   // in flight during this chunk's tiles
        if (tid < CH) {
            float s = 0.f;
#pragma unroll 16
            for (int l = 0; l < CP; ++l) s = fmaf(cs[tid * LD + l], cs[tid * LD + l], s);
            sc[tid] = tid < tn ? s : INFINITY;             // rows beyond the cloud: distance +inf, never a neighbour
        }
        __syncthreads();
        for (int t = 0; t < (tn + 31) / 32; ++t) {
            f32x16k acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            const float *arow = cs + (32 * t + li) * LD + half;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * kk], bq[kk], acc, 0, 0, 0);
            // acc[v]: candidate row 32 t + (v&3) + 8 (v>>2) + 4 half, query li
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int row0 = 32 * t + 8 * g4 + 4 * half;
                const float4 s4 = *reinterpret_cast<const float4 *>(sc + row0);
                const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = (sq + (-2.f * acc[4 * g4 + e])) + sv[e];
                    if (d < thr) {
                        qd[nq * 256 + tid] = d;
                        qj[nq * 256 + tid] = j0 + row0 + e;
                        ++nq;
                    }
                }
                if (__any(nq > QD - 4)) flush();
            }
        }
    }
    if (__any(nq > 0)) flush();

    // merge the half-wave lists of each query: lanes 0..31 absorb their partner's (already sorted) entries
    {
        typename TopList<KL>::Entry theirs[KL];
#pragma unroll
        for (int s = 0; s < KL; ++s) theirs[s] = TopList<KL>::from_lane(top.get(s), li + 32);
#pragma unroll
        for (int s = 0; s < KL; ++s)
            if (half == 0) top.merge(theirs[s]);
    }
    if (half == 0 && qin) {
        int *o = nn_idx + ((size_t)b * n + q) * k;
#pragma unroll
        for (int s = 0; s < KL; ++s)
            if (s < k) o[s] = top.index(s);
    }
}

template <int CP, int KL>
int launch_knn_mfma(int b, int n, int c, int k, const float *x, int *nn_idx, const int *seed, hipStream_t st) {
    const size_t lds = (size_t)(128 * (CP + 1) + 128 + 2 * kKnnQD * 256) * sizeof(float);
    auto kern = knn_mfma_kernel<CP, KL>;
    if (lds > 48 * 1024) {
        static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)once;
    }
    hipLaunchKernelGGL(kern, dim3(cdiv(n, 128), b), dim3(256), lds, st, n, c, k, x, nn_idx, seed);
    return pcops_launch_status();
}

// ------------------------------------------------------------------ fused kNN graph, fp16 pre-filter (round 4)
// Measured on this chip (tools/ubench/mfma_valu_coexec.hip, profiles/r04_ubench_mfma_valu_coexec.txt): MFMA time and VALU
// time ADD on a SIMD -- for v_mfma_f32_32x32x2_f32 and for the 16-bit MFMAs alike, with one wave per SIMD or two, clustered
// or interleaved, and a static priority split between the two waves of a SIMD changes nothing (profiles/
// r04_wave_prio_ab.txt).  So knn_mfma_kernel's 64-channel tile costs its 2048 matrix cycles PLUS ~2700 cycles of selection,
// and the matrix part cannot hide.  What can shrink is the matrix part itself: the exact fp32 distance is only needed for
// the few candidates that can still enter a top-k list.
//   * every 32 x 32 tile is first evaluated in FP16 on v_mfma_f32_32x32x16_f16 -- 4 instructions of 32 cycles instead of 32
//     of 64 for 64 channels (1/16 of the matrix time).  fp16 rounding of both operands moves a product by <= 2^-10 |x y|,
//     the distance by <= 2^-10 (s_i + s_j) (Cauchy-Schwarz + AM-GM); with the fp32 accumulation of the MFMA, the rounding of
//     the test itself and fp16 underflow:  |d_exact - d_fp16| <= A (s_i + s_j) + B,  A = 1.1e-3, B = 1e-6.
//   * a candidate is REJECTED iff  d_fp16 - A (s_i + s_j) - B > thr  (thr = the lane's current bound on its k-th distance,
//     as in knn_mfma_kernel) -- then d_exact > thr as well, and knn_mfma_kernel would not have queued it either.  The test is
//     one fma and one compare per pair:  fma(-2, acc, (1 - A) s_j)  >  thr - (1 - A) s_i + B.  Anything non-finite (a
//     feature beyond the fp16 range) fails the '>' and is kept.
//   * survivors (j only) wait in a per-lane LDS list; when a list fills or the candidate chunk is about to leave LDS, each
//     lane evaluates the EXACT distance of its survivors -- the contract's fmaf chain over c on the VALU, query row in
//     registers, candidate row from the fp32 copy of the chunk -- and inserts it in its sorted list exactly as
//     knn_mfma_kernel's flush does (ascending index order, strict '<').  Same lists, same merge, same indices: bit-exact
//     w.r.t. the oracle (tests/test_knn_gpu.py, test_bench_size_gpu.py run through this kernel).
// Workgroup: 8 waves (two per SIMD) = 256 queries on one candidate chunk; 32 queries per wave, two half-waves per
// query as in knn_mfma_kernel (shared bound tau2).  c == 64 (DGCNN's feature
// graphs), k <= 20; everything else takes knn_mfma_kernel.
typedef _Float16 f16x8k __attribute__((ext_vector_type(8)));
constexpr int kKnnPD = 32;             // pending survivors per lane; checked once per 32-row tile (<= 16 new entries each)
// (A, B assume GRADUAL underflow in v_cvt_f16_f32 and in the fp16 MFMA -- a channel below 6.1e-5 keeps its subnormal value.  It
// holds on gfx950: tests/test_knn_gpu.py "subnormal_mix" builds clouds whose graphs would lose neighbours by five times the
// bound if subnormals were flushed, and they are bit-exact.)
constexpr float kKnnA = 1.1e-3f, kKnnB = 1e-6f;

#ifdef PCOPS_KNN_STATS
__device__ unsigned long long g_knn_stats[4];    // pairs tested, survivors, exact distances below the bound, process rounds
#endif

template <int KL>
__global__ __launch_bounds__(512, 2) void knn_f16_kernel(int n, int k, const float *__restrict__ x,
                                                         int *__restrict__ nn_idx, const int *__restrict__ seed) {
    constexpr int CP = 64, CH = 128, LD = CP + 4;      // fp32 chunk row stride: 16-byte aligned rows
    constexpr int LH = CP + 8;                         // fp16 chunk row stride in halves (144 B: conflict-free b128 reads)
    constexpr int PD = kKnnPD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cs = lds;                                   // [CH][LD]   fp32 candidates (exact distances, squared norms)
    float *sc = cs + CH * LD;                          // [CH]       s_j
    float *su = sc + CH;                               // [CH]       (1 - A) s_j
    int *pj = reinterpret_cast<int *>(su + CH);        // [PD][512]  pending survivors
    _Float16 *chh = reinterpret_cast<_Float16 *>(pj + PD * 512);   // [CH][LH]  fp16 candidates (the filter's A operand)
    const CloudPart cp = xcd_cloud_part();      // the query blocks of a cloud share its candidates: one XCD, one L2
    const int b = cp.cloud;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    const float *xb = x + (size_t)b * n * CP;
    const int q = cp.part * 256 + wave * 32 + li;         // 8 waves = 256 queries share one candidate chunk
    const bool qin = q < n;

    // the query row: fp32 in registers (exact distances) and fp16 B fragments (channels 16 kk + 8 half .. + 7)
    float xq[CP];
    {
        const float4 *src = reinterpret_cast<const float4 *>(xb + (size_t)(qin ? q : 0) * CP);
#pragma unroll
        for (int l4 = 0; l4 < CP / 4; ++l4) {
            const float4 v = qin ? src[l4] : make_float4(0.f, 0.f, 0.f, 0.f);
            xq[4 * l4] = v.x; xq[4 * l4 + 1] = v.y; xq[4 * l4 + 2] = v.z; xq[4 * l4 + 3] = v.w;
        }
    }
    float sq = 0.f;
#pragma unroll
    for (int l = 0; l < CP; ++l) sq = fmaf(xq[l], xq[l], sq);
    f16x8k bh[CP / 16];
#pragma unroll
    for (int kk = 0; kk < CP / 16; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // (both half-waves hold the whole row; the fragment of a lane is the half-wave's 8 channels of every 16)
            const float lo = xq[16 * kk + e], hi = xq[16 * kk + 8 + e];
            bh[kk][e] = (_Float16)(half ? hi : lo);
        }
    const float sqa = (1.f - kKnnA) * sq;
    constexpr float kF16Safe = 6.0e4f;                 // |x| <= 6e4 converts to a finite fp16 with relative error 2^-11
    int big = 0;
    bool filter_off = false;
#pragma unroll
    for (int l = 0; l < CP; ++l) big |= (int)!(fabsf(xq[l]) <= kF16Safe);

    // SEEDED start (pcops_knn_graph_seeded, as in knn_mfma_kernel): k distinct points per query -- DGCNN: the previous
    // layer's neighbours -- bound the k-th distance from above by the largest exact distance to them.  Here a survivor
    // costs an exact 64-channel distance AND an insertion, so a query that starts from a bound near its final one saves
    // far more than the k / 2 exact distances per half-wave the bound costs.  Invalid rows (out of range / repeated
    // index) give no bound.
    float tau = INFINITY;
    if (seed != nullptr) {
        const int *sd = seed + ((size_t)b * n + (qin ? q : 0)) * k;
        bool seed_ok = true;
        int mine[(KL + 1) / 2];
#pragma unroll
        for (int u = 0; u < (KL + 1) / 2; ++u) {
            const int s_ = half + 2 * u;
            mine[u] = (s_ < k && qin) ? sd[s_] : -1 - u - 64 * half;
            if (s_ < k && qin && (mine[u] < 0 || mine[u] >= n)) seed_ok = false;
        }
#pragma unroll
        for (int u = 0; u < (KL + 1) / 2; ++u) {
#pragma unroll
            for (int w = u + 1; w < (KL + 1) / 2; ++w) seed_ok = seed_ok && (mine[u] != mine[w]);
            const int theirs = __shfl_xor(mine[u], 32, 64);
#pragma unroll
            for (int w = 0; w < (KL + 1) / 2; ++w) seed_ok = seed_ok && (theirs != mine[w]);
        }
        seed_ok = seed_ok && (__shfl_xor((int)seed_ok, 32, 64) != 0);
        float worst = 0.f, smax = 0.f;
#pragma unroll 1
        for (int u = 0; u < (KL + 1) / 2; ++u) {
            const int s_ = half + 2 * u;
            if (!(s_ < k && qin)) continue;
            int j = mine[u];
            j = j < 0 ? 0 : (j >= n ? n - 1 : j);
            const float4 *pj4 = reinterpret_cast<const float4 *>(xb + (size_t)j * CP);
            float d = 0.f, sj = 0.f;
#pragma unroll
            for (int l4 = 0; l4 < CP / 4; ++l4) {
                const float4 v = pj4[l4];
                const float e0 = xq[4 * l4] - v.x, e1 = xq[4 * l4 + 1] - v.y, e2 = xq[4 * l4 + 2] - v.z, e3 = xq[4 * l4 + 3] - v.w;
                d = fmaf(e0, e0, d); d = fmaf(e1, e1, d); d = fmaf(e2, e2, d); d = fmaf(e3, e3, d);
                sj = fmaf(v.x, v.x, sj); sj = fmaf(v.y, v.y, sj); sj = fmaf(v.z, v.z, sj); sj = fmaf(v.w, v.w, sj);
            }
            worst = fmaxf(worst, d);
            smax = fmaxf(smax, sj);
        }
        worst = fmaxf(worst, __shfl_xor(worst, 32, 64));
        smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
        // the scan's distance (s_q - 2 <x_q, x_j>) + s_j is within ~4 c eps max(s_q, s_j) of the direct form above; the margin
        // is an order of magnitude wider (as in knn_mfma_kernel)
        if (qin && seed_ok) tau = worst + 1e-3f * (sq + smax) + 1e-30f;
    }
    TopList<KL> top;
    top.init();
    int nq = 0;
    float tau2 = INFINITY, thr = tau, tp = (tau - sqa) + kKnnB;
#ifdef PCOPS_KNN_STATS
    unsigned st_surv = 0, st_acc = 0, st_rounds = 0;
#endif
    int j0 = 0;
    constexpr int NQ4 = CH * (CP / 4);                  // float4 per chunk
    constexpr int NV = NQ4 / 512;                       // float4 per thread per chunk (4)
    float4 pre[NV];
    auto fetch = [&](int j0n) {
        const int tnn = min(CH, n - j0n);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e4 = tid + 512 * u;
            const int r = e4 / (CP / 4), l = (e4 - r * (CP / 4)) * 4;
            pre[u] = r < tnn ? *reinterpret_cast<const float4 *>(xb + (size_t)(j0n + r) * CP + l)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (n > 0) fetch(0);
    for (j0 = 0; j0 < n; j0 += CH) {
        const int tn = min(CH, n - j0);
        __syncthreads();                                // every wave has processed its survivors of the previous chunk
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e4 = tid + 512 * u;
            const int r = e4 / (CP / 4), l = (e4 - r * (CP / 4)) * 4;
            *reinterpret_cast<float4 *>(cs + r * LD + l) = pre[u];
            typedef _Float16 f16x4k __attribute__((ext_vector_type(4)));
            f16x4k h;
            h[0] = (_Float16)pre[u].x; h[1] = (_Float16)pre[u].y; h[2] = (_Float16)pre[u].z; h[3] = (_Float16)pre[u].w;
            *reinterpret_cast<f16x4k *>(chh + r * LH + l) = h;
            big |= (int)!(fabsf(pre[u].x) <= kF16Safe) | (int)!(fabsf(pre[u].y) <= kF16Safe) |
                   (int)!(fabsf(pre[u].z) <= kF16Safe) | (int)!(fabsf(pre[u].w) <= kF16Safe);
        }
        // a feature beyond the fp16 range (or a NaN) anywhere in this chunk or in this workgroup's queries: the error bound
        // of the filter does not hold -- from here on nothing is rejected (tp = +inf), every pair takes the exact path
        if (__syncthreads_or(big)) { filter_off = true; tp = INFINITY; }
        big = 0;
        if (j0 + CH < n) fetch(j0 + CH);
        if (tid < CH) {
            float s = 0.f;
#pragma unroll 16
            for (int l = 0; l < CP; ++l) s = fmaf(cs[tid * LD + l], cs[tid * LD + l], s);
            s = tid < tn ? s : INFINITY;                // rows beyond the cloud: never a neighbour
            sc[tid] = s;
            su[tid] = (1.f - kKnnA) * s;
        }
        __syncthreads();
        const int ntile = (tn + 31) / 32;
        for (int t = 0; t < ntile; ++t) {
            f32x16k acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            const _Float16 *arow = chh + (32 * t + li) * LH + 8 * half;
#pragma unroll
            for (int kk = 0; kk < CP / 16; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8k *>(arow + 16 * kk), bh[kk], acc, 0, 0, 0);
            // acc[v]: candidate row 32 t + (v&3) + 8 (v>>2) + 4 half, query li
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int row0 = 32 * t + 8 * g4 + 4 * half;
                const float4 u4 = *reinterpret_cast<const float4 *>(su + row0);
                const float uv[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float val = fmaf(-2.f, acc[4 * g4 + e], uv[e]);
                    if (!(val > tp)) {
                        pj[nq * 512 + tid] = j0 + row0 + e;
                        ++nq;
                    }
                }
            }
            // a list could overflow in the next tile, or these were the chunk's last rows (they leave LDS at the next
            // barrier).  Checked once per tile, after the accumulators are dead: inside the row groups the exact chain's
            // registers came on top of them and spilled
            const bool last = t == ntile - 1;
            if (__any(nq > PD - 16) || (last && __any(nq > 0))) {
#ifdef PCOPS_KNN_STATS
                st_surv += (unsigned)nq;
#endif
                {
                // the two half-waves of a query POOL their survivors (round 5): both lanes hold the query row and either list may
                // take any candidate now that the lists are keys (the merge takes the best of the union, tau2's argument holds for
                // any split), so the lane with fewer survivors takes the partner's LAST ones -- a round costs 133 instructions for
                // the whole wave, and the rounds of a flush are the fullest lane's count: max over 32 pairs of half the pair's
                // sum instead of max over 64 lanes
                // (the partner's pending entries were written by another lane of this wave: order the LDS writes before the
                // pooled reads below -- the hardware executes a wave's LDS instructions in order, the memory model does not say so)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int nb = __shfl_xor(nq, 32, 64);                 // the partner's count
                const int mine = (nq + nb + (half == 0 ? 1 : 0)) >> 1; // survivors this lane processes
                const int mx = (int)wave_max_u32((unsigned)mine);
                for (int u = 0; u < mx; ++u) {
                    const bool live = u < mine;
                    const bool own = u < nq;                           // (nq >= mine: all of them its own)
                    const int slot = own ? u * 512 + tid : (nb - 1 - (u - nq)) * 512 + (tid ^ 32);
                    const int j = live ? pj[slot] : j0;
                    const int rr = j - j0;
                    const float *row = cs + rr * LD;
                    float inner = 0.f;
#pragma unroll
                    for (int l16 = 0; l16 < CP / 16; ++l16) {          // four 16-byte reads in flight, then their 16 chain steps
                        float4 v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(row + 16 * l16 + 4 * i);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            inner = fmaf(xq[16 * l16 + 4 * i], v[i].x, inner);
                            inner = fmaf(xq[16 * l16 + 4 * i + 1], v[i].y, inner);
                            inner = fmaf(xq[16 * l16 + 4 * i + 2], v[i].z, inner);
                            inner = fmaf(xq[16 * l16 + 4 * i + 3], v[i].w, inner);
                        }
                        __builtin_amdgcn_sched_barrier(0);             // (all sixteen reads hoisted to the top cost 21 spilled registers)
                    }
                    const float d = (sq + (-2.f * inner)) + sc[rr];
#ifdef PCOPS_KNN_STATS
                    st_acc += (live && d < thr) ? 1u : 0u;
#endif
                    top.insert(TopList<KL>::make(d, j, live && d < INFINITY));
                }
#ifdef PCOPS_KNN_STATS
                st_rounds += (lane == 0) ? (unsigned)mx : 0u;
#endif
                nq = 0;
                const float hv = top.value((KL + 1) / 2 - 1);
                tau2 = fmaxf(hv, __shfl_xor(hv, 32, 64));
                thr = fminf(fminf(top.value(KL - 1), nextafterf(tau2, INFINITY)), tau);
                tp = filter_off ? INFINITY : (thr - sqa) + kKnnB;   // reject iff  fma(-2, acc, su_j) > tp
                }
            }
        }
    }
#ifdef PCOPS_KNN_STATS
    atomicAdd(&g_knn_stats[1], (unsigned long long)st_surv);
    atomicAdd(&g_knn_stats[2], (unsigned long long)st_acc);
    atomicAdd(&g_knn_stats[3], (unsigned long long)st_rounds);
    if (tid == 0) atomicAdd(&g_knn_stats[0], (unsigned long long)256 * n);
#endif

    // merge the half-wave lists of each query: lanes 0..31 absorb their partner's (already sorted) entries
    {
        typename TopList<KL>::Entry theirs[KL];
#pragma unroll
        for (int s = 0; s < KL; ++s) theirs[s] = TopList<KL>::from_lane(top.get(s), li + 32);
#pragma unroll
        for (int s = 0; s < KL; ++s)
            if (half == 0) top.merge(theirs[s]);
    }
    if (half == 0 && qin) {
        int *o = nn_idx + ((size_t)b * n + q) * k;
#pragma unroll
        for (int s = 0; s < KL; ++s)
            if (s < k) o[s] = top.index(s);
    }
}

static bool knn_f16_enabled() { return pcops_get_option(PCOPS_OPT_KNN_F16_PREFILTER) != 0; }

int launch_knn_f16(int b, int n, int k, const float *x, int *nn_idx, const int *seed, hipStream_t st) {
    constexpr int CP = 64, CH = 128;
    const size_t lds = (size_t)(CH * (CP + 4) + 2 * CH + kKnnPD * 512) * sizeof(float) + (size_t)CH * (CP + 8) * 2;
    auto kern = knn_f16_kernel<20>;
    // (the kernel also has 256 bytes of STATIC LDS -- __syncthreads_or -- so asking for the full 160 KB is refused)
    static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)once;
    hipLaunchKernelGGL(kern, dim3(cdiv(n, 256), b), dim3(512), lds, st, n, k, x, nn_idx, seed);
    return pcops_launch_status();
}

// ------------------------------------------------------------------ materialised path
// pairwise_distance: 64x64 output tile per workgroup, 4x4 outputs per lane, channel chunks of
// CK staged in LDS; the fmaf chains run c-ascending across chunks.
constexpr int kPdT = 64, kPdCK = 32;

__global__ __launch_bounds__(256) void pairwise_distance_kernel(int n, int c,
                                                                const float *__restrict__ x,
                                                                float *__restrict__ adj) {
    __shared__ float xi[kPdT][kPdCK + 1];
    __shared__ float xj[kPdT][kPdCK + 1];
    __shared__ float ssi[kPdT], ssj[kPdT];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * kPdT, j0 = blockIdx.x * kPdT;
    const int tid = threadIdx.x;
    const int ti = tid / 16, tj = tid % 16;
    const float *xb = x + (size_t)b * n * c;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[a][d] = 0.f;
    float srow = 0.f;  // lanes 0..63: s of i-rows, lanes 64..127: s of j-rows

    for (int c0 = 0; c0 < c; c0 += kPdCK) {
        const int cn = min(kPdCK, c - c0);
        __syncthreads();
        for (int e = tid; e < kPdT * kPdCK; e += 256) {
            const int r = e / kPdCK, l = e - r * kPdCK;
            xi[r][l] = (i0 + r < n && l < cn) ? xb[(size_t)(i0 + r) * c + c0 + l] : 0.f;
            xj[r][l] = (j0 + r < n && l < cn) ? xb[(size_t)(j0 + r) * c + c0 + l] : 0.f;
        }
        __syncthreads();
        if (tid < 64) {
            for (int l = 0; l < cn; ++l) srow = fmaf(xi[tid][l], xi[tid][l], srow);
        } else if (tid < 128) {
            for (int l = 0; l < cn; ++l) srow = fmaf(xj[tid - 64][l], xj[tid - 64][l], srow);
        }
        for (int l = 0; l < cn; ++l) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = xi[ti * 4 + a][l];
#pragma unroll
            for (int d = 0; d < 4; ++d) bv[d] = xj[tj * 4 + d][l];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][d] = fmaf(av[a], bv[d], acc[a][d]);
        }
    }
    if (tid < 64) ssi[tid] = srow;
    else if (tid < 128) ssj[tid - 64] = srow;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ti * 4 + a;
        if (i >= n) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int j = j0 + tj * 4 + d;
            if (j < n)
                adj[((size_t)b * n + i) * n + j] = (ssi[ti * 4 + a] + (-2.f * acc[a][d])) + ssj[tj * 4 + d];
        }
    }
}

// top_k(-adj, k) on a materialised (rows, n) matrix.  64 rows per workgroup (one lane each);
// 64x64 tiles are loaded coalesced and transposed through LDS; sorted lists in LDS.
__global__ __launch_bounds__(64) void knn_topk_kernel(long long rows, int n, int k,
                                                      const float *__restrict__ adj,
                                                      int *__restrict__ nn_idx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tile = lds;             // [64][65]
    float *lv = tile + 64 * 65;    // [k][64]
    int *li = reinterpret_cast<int *>(lv + k * 64);
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 64;
    const bool valid = r0 + tid < rows;
    for (int s = 0; s < k; ++s) {
        lv[s * 64 + tid] = INFINITY;
        li[s * 64 + tid] = 0;
    }
    float worst = INFINITY;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int tn = min(64, n - j0);
        __syncthreads();
        for (int r = 0; r < 64; ++r)
            if (r0 + r < rows && tid < tn) tile[r * 65 + tid] = adj[(r0 + r) * n + j0 + tid];
        __syncthreads();
        for (int jj = 0; jj < tn; ++jj) {
            const float d = tile[tid * 65 + jj];
            if (valid && d < worst) {
                int pos = k - 1;
                while (pos > 0) {
                    const float pv = lv[(pos - 1) * 64 + tid];
                    if (!(d < pv)) break;
                    lv[pos * 64 + tid] = pv;
                    li[pos * 64 + tid] = li[(pos - 1) * 64 + tid];
                    --pos;
                }
                lv[pos * 64 + tid] = d;
                li[pos * 64 + tid] = j0 + jj;
                worst = lv[(k - 1) * 64 + tid];
            }
        }
    }
    if (valid)
        for (int s = 0; s < k; ++s) nn_idx[(r0 + tid) * k + s] = li[s * 64 + tid];
}

// ------------------------------------------------------------------ edge features
// out[b,i,s,:] = [x_i | x_j - x_i], j = nn_idx[b,i,s]
template <int VEC>
__global__ __launch_bounds__(256) void edge_feature_kernel(long long total, int n, int cv, int k,
                                                           const float *__restrict__ x,
                                                           const int *__restrict__ nn_idx,
                                                           float *__restrict__ out) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const vec_t *xv = reinterpret_cast<const vec_t *>(x);
    vec_t *ov = reinterpret_cast<vec_t *>(out);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long edge = e / cv;  // (b, i, s)
        const int col = (int)(e - edge * cv);
        const long long node = edge / k;  // (b, i)
        const long long bi = node / n;
        const int j = nn_idx[edge];
        const vec_t ci = xv[node * cv + col];
        const vec_t cj = xv[(bi * n + j) * cv + col];
        ov[edge * 2 * cv + col] = ci;
        ov[edge * 2 * cv + cv + col] = cj - ci;
    }
}

// grad_x[b,i,:] += sum_s (ga - gb)[b,i,s,:] ; grad_x[b,nn[b,i,s],:] += gb[b,i,s,:]
template <bool CENTRAL>
__global__ __launch_bounds__(256) void edge_feature_grad_kernel(long long total, int n, int c, int k,
                                                                const float *__restrict__ grad_out,
                                                                const int *__restrict__ nn_idx,
                                                                float *__restrict__ grad_x) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long node = e / c;  // (b, i)
        const int col = (int)(e - node * c);
        const long long bi = node / n;
        float central = 0.f;
        for (int s = 0; s < k; ++s) {
            const long long edge = node * k + s;
            const float ga = grad_out[edge * 2 * c + col];
            const float gb = grad_out[edge * 2 * c + c + col];
            central += ga - gb;
            if (!CENTRAL) atomicAdd(&grad_x[(bi * n + nn_idx[edge]) * (long long)c + col], gb);
        }
        if (CENTRAL) grad_x[e] = central;       // the neighbour term is the caller's sorted scatter (deterministic mode)
        else atomicAdd(&grad_x[e], central);
    }
}

}  // namespace

extern "C" int pcops_knn_graph_seeded(int b, int n, int c, int k, const float *x, const int *seed, int *nn_idx,
                                      pcops_stream_t stream);
extern "C" int pcops_knn_graph(int b, int n, int c, int k, const float *x, int *nn_idx,
                               pcops_stream_t stream) {
    return pcops_knn_graph_seeded(b, n, c, k, x, nullptr, nn_idx, stream);
}

static bool knn_use_mfma() {
    static const bool on = [] { const char *e = getenv("PCOPS_KNN_MFMA"); return !(e && e[0] == '0'); }();   // kernel A/B only
    return on;
}
static bool knn_takes_f16(int n, int c, int k, const float *x) {
    return knn_use_mfma() && knn_f16_enabled() && c == 64 && k <= 20 && n >= 256 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}
extern "C" int pcops_knn_graph_path(int b, int n, int c, int k, const float *x) {
    (void)b;
    if (knn_takes_f16(n, c, k, x)) return 3 | 16;
    if (knn_use_mfma() && k <= 32 && c <= 128) return 2 | 16;
    return c <= 128 ? 1 : 0;
}

extern "C" int pcops_knn_graph_seeded(int b, int n, int c, int k, const float *x, const int *seed, int *nn_idx,
                                      pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 1);
    PCOPS_REQUIRE_ARG(k > 0 && k <= n);  // tf.nn.top_k: k must not exceed the last dimension
    if (b == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(x);
    PCOPS_REQUIRE_PTR(nn_idx);
    PCOPS_REQUIRE_SHAPE(b <= 65535);
    hipStream_t st = as_stream(stream);
    const bool use_mfma = knn_use_mfma();
    // 64-channel graphs (DGCNN's feature graphs): fp16 pre-filter on the 16-bit matrix pipe + exact distances of the
    // survivors (knn_f16_kernel above); rows must be 16-byte aligned
    if (knn_takes_f16(n, c, k, x)) return launch_knn_f16(b, n, k, x, nn_idx, seed, st);
    if (use_mfma && k <= 32 && c <= 128) {
#define PCOPS_KNN_CASE(CP_)                                                      \
    do {                                                                         \
        if (k <= 20) return launch_knn_mfma<CP_, 20>(b, n, c, k, x, nn_idx, seed, st); \
        return launch_knn_mfma<CP_, 32>(b, n, c, k, x, nn_idx, seed, st);              \
    } while (0)
        if (c <= 4) PCOPS_KNN_CASE(4);
        if (c <= 16) PCOPS_KNN_CASE(16);
        if (c <= 64) PCOPS_KNN_CASE(64);
        PCOPS_KNN_CASE(128);
#undef PCOPS_KNN_CASE
    }
    if (c <= 4) return launch_knn_graph<4>(b, n, c, k, x, nn_idx, st);
    if (c <= 8) return launch_knn_graph<8>(b, n, c, k, x, nn_idx, st);
    if (c <= 16) return launch_knn_graph<16>(b, n, c, k, x, nn_idx, st);
    if (c <= 32) return launch_knn_graph<32>(b, n, c, k, x, nn_idx, st);
    if (c <= 64) return launch_knn_graph<64>(b, n, c, k, x, nn_idx, st);
    if (c <= 128) return launch_knn_graph<128>(b, n, c, k, x, nn_idx, st);
    return PCOPS_ERR_UNSUPPORTED;
}

#ifdef PCOPS_KNN_STATS
// diagnostics build only (tools/build_variant.sh knnstats "-DPCOPS_KNN_STATS=1"): counters of knn_f16_kernel since the
// last call -- pairs tested, survivors of the fp16 filter, exact distances below the lane's bound, processing rounds
extern "C" int pcops_knn_debug_stats(unsigned long long *out4) {
    if (hipDeviceSynchronize() != hipSuccess) return PCOPS_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_knn_stats), 4 * sizeof(unsigned long long)) != hipSuccess) return PCOPS_ERR_LAUNCH;
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_stats), z, sizeof(z)) != hipSuccess) return PCOPS_ERR_LAUNCH;
    return PCOPS_OK;
}
#endif

extern "C" int pcops_pairwise_distance(int b, int n, int c, const float *x, float *adj,
                                       pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 1);
    if ((long long)b * n == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(x);
    PCOPS_REQUIRE_PTR(adj);
    PCOPS_REQUIRE_SHAPE(b <= 65535);
    hipLaunchKernelGGL(pairwise_distance_kernel, dim3(cdiv(n, kPdT), cdiv(n, kPdT), b), dim3(256), 0,
                       as_stream(stream), n, c, x, adj);
    return pcops_launch_status();
}

extern "C" int pcops_knn_topk(int rows, int n, int k, const float *adj, int *nn_idx,
                              pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(rows >= 0 && n >= 0);
    PCOPS_REQUIRE_ARG(k > 0 && k <= n);
    if (rows == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(adj);
    PCOPS_REQUIRE_PTR(nn_idx);
    const size_t lds = (size_t)(64 * 65 + 2 * k * 64) * 4;
    if (lds > 64 * 1024) return PCOPS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(knn_topk_kernel, dim3(cdiv(rows, 64)), dim3(64), lds, as_stream(stream),
                       (long long)rows, n, k, adj, nn_idx);
    return pcops_launch_status();
}

extern "C" int pcops_edge_feature(int b, int n, int c, int k, const float *x, const int *nn_idx,
                                  float *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 0 && k >= 0);
    if ((long long)b * n * k * c == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(x);
    PCOPS_REQUIRE_PTR(nn_idx);
    PCOPS_REQUIRE_PTR(out);
    const bool v4 = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int cv = v4 ? c / 4 : c;
    const long long total = (long long)b * n * k * cv;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    if (v4)
        hipLaunchKernelGGL((edge_feature_kernel<4>), dim3(grid), dim3(256), 0, as_stream(stream), total,
                           n, cv, k, x, nn_idx, out);
    else
        hipLaunchKernelGGL((edge_feature_kernel<1>), dim3(grid), dim3(256), 0, as_stream(stream), total,
                           n, cv, k, x, nn_idx, out);
    return pcops_launch_status();
}

extern "C" int pcops_edge_feature_grad(int b, int n, int c, int k, const float *grad_out,
                                       const int *nn_idx, float *grad_x, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 0 && k >= 0);
    const long long total = (long long)b * n * c;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_x);
    if (pcops_get_deterministic()) return PCOPS_ERR_UNSUPPORTED;   // atomics: use _central + pcops_scatter_rows_sorted
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)total, st) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    if (k == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_out);
    PCOPS_REQUIRE_PTR(nn_idx);
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(edge_feature_grad_kernel<false>, dim3(grid), dim3(256), 0, st, total, n, c, k,
                       grad_out, nn_idx, grad_x);
    return pcops_launch_status();
}

extern "C" int pcops_edge_feature_grad_central(int b, int n, int c, int k, const float *grad_out, float *grad_x,
                                               pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && c >= 0 && k >= 0);
    const long long total = (long long)b * n * c;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_x);
    if (k > 0) PCOPS_REQUIRE_PTR(grad_out);
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(edge_feature_grad_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), total, n, c, k,
                       grad_out, (const int *)nullptr, grad_x);
    return pcops_launch_status();
}
