// interpolate.hip -- three_nn, three_interpolate and its gradient for gfx950.
//
// The reference only has single-threaded CPU kernels for these ops
// (3d_interpolation/tf_interpolate.cpp:60-153, registered DEVICE_CPU at :187,222,262), which
// forces a GPU->CPU->GPU bounce inside every feature-propagation module.  These are the
// device versions with identical results:
//   * three_nn: one lane per unknown point, the known cloud staged in LDS as SoA and read
//     with wave-uniform ds_read_b128; strict-< insertion cascade, +inf sentinels (the
//     reference's `double best=1e40` holds a widened FLOAT distance and becomes +inf on the
//     float store, so pure fp32 with +inf is bit-identical); squared distances.
//   * three_interpolate: out = (p1*w1 + p2*w2) + p3*w3, uncontracted (-ffp-contract=off).
#include <math.h>

#include "common.h"

namespace {

constexpr int kNnTile = 4096;

__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2,
                                                       float *__restrict__ dist,
                                                       int *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const float *p2 = xyz2 + (size_t)b * m * 3;
    const int j = blockIdx.x * nthr + tid;
    const bool valid = j < n;
    const size_t q = (size_t)b * n + (valid ? j : 0);
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (valid) {
        x1 = xyz1[q * 3 + 0];
        y1 = xyz1[q * 3 + 1];
        z1 = xyz1[q * 3 + 2];
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;

    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int tn = min(kNnTile, m - t0);
        const int tp = (tn + 3) & ~3;
        float *xs = lds, *ys = lds + tp, *zs = lds + 2 * tp;
        if (t0) __syncthreads();
        for (int e = tid; e < tn * 3; e += nthr) {
            const float v = p2[(size_t)t0 * 3 + e];
            const int k = e / 3;
            lds[(e - k * 3) * tp + k] = v;
        }
        __syncthreads();
        for (int k0 = 0; k0 < tn; k0 += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(xs + k0);
            const float4 Y = *reinterpret_cast<const float4 *>(ys + k0);
            const float4 Z = *reinterpret_cast<const float4 *>(zs + k0);
            const float px[4] = {X.x, X.y, X.z, X.w};
            const float py[4] = {Y.x, Y.y, Y.z, Y.w};
            const float pz[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u < tn) {  // wave-uniform tail guard (pad lanes hold garbage)
                    const int k = t0 + k0 + u;
                    const float dx = px[u] - x1, dy = py[u] - y1, dz = pz[u] - z1;
                    const float d = dx * dx + dy * dy + dz * dz;
                    if (d < b1) {
                        b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
                    } else if (d < b2) {
                        b3 = b2; i3 = i2; b2 = d; i2 = k;
                    } else if (d < b3) {
                        b3 = d; i3 = k;
                    }
                }
            }
        }
    }
    if (valid) {
        dist[q * 3 + 0] = b1; dist[q * 3 + 1] = b2; dist[q * 3 + 2] = b3;
        idx[q * 3 + 0] = i1; idx[q * 3 + 1] = i2; idx[q * 3 + 2] = i3;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void three_interpolate_kernel(long long total, int m, int cv, int n,
                                                                const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long row = e / cv;  // (b, j)
        const int col = (int)(e - row * cv);
        const long long bi = row / n;
        const int i1 = idx[row * 3 + 0], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
        const float w1 = weight[row * 3 + 0], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
        const vec_t *base = reinterpret_cast<const vec_t *>(points) + bi * m * (long long)cv + col;
        const vec_t a = base[(long long)i1 * cv], bq = base[(long long)i2 * cv], c3 = base[(long long)i3 * cv];
        reinterpret_cast<vec_t *>(out)[e] = a * w1 + bq * w2 + c3 * w3;
    }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(long long total, int n, int c, int m,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ grad_points) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const long long row = e / c;
        const int col = (int)(e - row * c);
        const long long bi = row / n;
        const float g = grad_out[e];
        float *base = grad_points + bi * m * (long long)c + col;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            atomicAdd(base + (long long)idx[row * 3 + t] * c, g * weight[row * 3 + t]);
    }
}

}  // namespace

extern "C" int pcops_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                              int *idx, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0);
    if ((long long)b * n == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(xyz1);
    if (m > 0) PCOPS_REQUIRE_PTR(xyz2);
    PCOPS_REQUIRE_PTR(dist);
    PCOPS_REQUIRE_PTR(idx);
    const int threads = n >= 256 ? 256 : ((n + kWave - 1) / kWave) * kWave;
    const int tile = m < kNnTile ? m : kNnTile;
    const size_t lds = (size_t)3 * ((tile + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, threads), b), dim3(threads), lds,
                       as_stream(stream), n, m, xyz1, xyz2, dist, idx);
    return pcops_launch_status();
}

extern "C" int pcops_three_interpolate(int b, int m, int c, int n, const float *points,
                                       const int *idx, const float *weight, float *out,
                                       pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0 && c >= 0);
    if ((long long)b * n * c == 0) return PCOPS_OK;
    PCOPS_REQUIRE_SHAPE(m >= 1);
    PCOPS_REQUIRE_PTR(points);
    PCOPS_REQUIRE_PTR(idx);
    PCOPS_REQUIRE_PTR(weight);
    PCOPS_REQUIRE_PTR(out);
    const bool v4 = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int cv = v4 ? c / 4 : c;
    const long long total = (long long)b * n * cv;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    if (v4)
        hipLaunchKernelGGL((three_interpolate_kernel<4>), dim3(grid), dim3(256), 0, as_stream(stream),
                           total, m, cv, n, points, idx, weight, out);
    else
        hipLaunchKernelGGL((three_interpolate_kernel<1>), dim3(grid), dim3(256), 0, as_stream(stream),
                           total, m, cv, n, points, idx, weight, out);
    return pcops_launch_status();
}

extern "C" int pcops_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out,
                                            const int *idx, const float *weight, float *grad_points,
                                            pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 0 && m >= 0 && c >= 0);
    if ((long long)b * m * c == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_points);
    if (pcops_get_deterministic()) return PCOPS_ERR_UNSUPPORTED;   // float atomics: pcops_scatter_rows_sorted instead
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, st) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    const long long total = (long long)b * n * c;
    if (total == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(grad_out);
    PCOPS_REQUIRE_PTR(idx);
    PCOPS_REQUIRE_PTR(weight);
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid), dim3(256), 0, st, total, n, c, m,
                       grad_out, idx, weight, grad_points);
    return pcops_launch_status();
}
