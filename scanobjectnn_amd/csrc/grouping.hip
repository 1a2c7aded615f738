// grouping.hip -- ball query, group_point(+grad), selection-sort top-k for gfx950.
//
// Replaces grouping/tf_grouping_g.cu:3-141 of the reference (behaviour only).
//
// Ball query design (CDNA4-first, not the reference's one-block-per-cloud scan):
//   * grid = (query tiles, clouds); every lane owns ONE query for the whole scan, so the
//     reference's "first nsample hits in ascending dataset order" falls out of the scan
//     order with no cross-lane bookkeeping;
//   * the dataset cloud is staged ONCE per workgroup into LDS as SoA x[],y[],z[] by
//     coalesced dword loads of the (n,3) AoS tensor; the inner loop reads it with
//     wave-uniform ds_read_b128 (4 points per instruction, broadcast, conflict free);
//   * `sqrtf(d2) < r` is replaced by the exactly equivalent `d2 <= T`, T = the largest
//     fp32 with sqrtf(T) < r, computed on the host (sqrtf is monotone and correctly
//     rounded).  The naive d2 < r*r is NOT equivalent;
//   * the multi-radius (MSG) form tests up to 4 radii against one d2 in the same pass;
//   * distances are uncontracted fp32: ((dx*dx + dy*dy) + dz*dz), dx = query - dataset
//     (file is compiled with -ffp-contract=off).
#include <math.h>

#include "common.h"

namespace {

constexpr int kQbpMaxScales = 4;
constexpr int kQbpTile = 4096;  // dataset points staged per LDS pass (48 KiB)

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

template <int NS>
__global__ __launch_bounds__(256) void qbp_kernel(int n, int m, QbpArgs<NS> a,
                                                  const float *__restrict__ xyz1,
                                                  const float *__restrict__ xyz2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const float *p1 = xyz1 + (size_t)b * n * 3;
    const int j = blockIdx.x * nthr + tid;
    const bool valid = j < m;
    const size_t q = (size_t)b * m + (valid ? j : 0);

    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
        qx = xyz2[q * 3 + 0];
        qy = xyz2[q * 3 + 1];
        qz = xyz2[q * 3 + 2];
    }
    int cnt[NS], first[NS];
    int *row[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        cnt[s] = 0;
        first[s] = 0;
        row[s] = a.s[s].idx + q * a.s[s].nsample;
    }
    bool done = !valid;

    for (int t0 = 0; t0 < n; t0 += kQbpTile) {
        const int tn = min(kQbpTile, n - t0);
        const int tp = (tn + 3) & ~3;  // padded to the b128 read width
        float *xs = lds, *ys = lds + tp, *zs = lds + 2 * tp;
        if (t0) __syncthreads();
        for (int e = tid; e < tn * 3; e += nthr) {  // coalesced AoS read -> SoA LDS
            const float v = p1[(size_t)t0 * 3 + e];
            const int k = e / 3;
            lds[(e - k * 3) * tp + k] = v;
        }
        if (tid < tp - tn) {  // pad so that d2 = +inf for the tail
            xs[tn + tid] = 3.0e38f;
            ys[tn + tid] = 3.0e38f;
            zs[tn + tid] = 3.0e38f;
        }
        __syncthreads();
        if (__all(done)) continue;  // wave-uniform; barriers above stay matched

        for (int k0 = 0; k0 < tp; k0 += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(xs + k0);
            const float4 Y = *reinterpret_cast<const float4 *>(ys + k0);
            const float4 Z = *reinterpret_cast<const float4 *>(zs + k0);
            const float px[4] = {X.x, X.y, X.z, X.w};
            const float py[4] = {Y.x, Y.y, Y.z, Y.w};
            const float pz[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = qx - px[u], dy = qy - py[u], dz = qz - pz[u];
                const float d2 = dx * dx + dy * dy + dz * dz;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (!done && d2 <= a.s[s].thresh && cnt[s] < a.s[s].nsample) {
                        const int k = t0 + k0 + u;
                        if (cnt[s] == 0) first[s] = k;
                        row[s][cnt[s]] = k;
                        ++cnt[s];
                    }
                }
            }
            bool full = true;
#pragma unroll
            for (int s = 0; s < NS; ++s) full = full && (cnt[s] >= a.s[s].nsample);
            done = done || full;
            if (__all(done)) break;
        }
    }

    if (valid) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            for (int l = cnt[s]; l < a.s[s].nsample; ++l) row[s][l] = first[s];
            if (a.s[s].cnt) a.s[s].cnt[q] = cnt[s];
        }
    }
}

template <int NS>
int launch_qbp(int b, int n, int m, const QbpArgs<NS> &a, const float *xyz1, const float *xyz2,
               hipStream_t st) {
    const int threads = m >= 256 ? 256 : ((m + kWave - 1) / kWave) * kWave;
    const int tile = n < kQbpTile ? n : kQbpTile;
    const size_t lds = (size_t)3 * ((tile + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL((qbp_kernel<NS>), dim3(cdiv(m, threads), b), dim3(threads), lds, st, n, m, a,
                       xyz1, xyz2);
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
    hipLaunchKernelGGL(selection_sort_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, as_stream(stream),
                       rows, n, k, dist, outi, out);
    return pcops_launch_status();
}
