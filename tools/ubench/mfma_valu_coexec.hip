// Does VALU work hide under MFMAs on gfx950 -- and does it depend on the MFMA's type?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coexec mfma_valu_coexec.hip && ./mfma_valu_coexec
// One wave per SIMD (256 threads per workgroup, one workgroup per CU) and two.  Per iteration a wave issues M MFMAs on
// independent accumulators (no dependency stalls) and V independent v_fma_f32; three kernels per MFMA type:
//   mfma only (V = 0) | valu only (M = 0) | both, interleaved in program order.
// If the two pipes co-execute, t(both) ~ max(t(mfma), t(valu)); if they share the datapath, t(both) ~ the sum.
// Types: v_mfma_f32_32x32x2_f32 (64 cycles) and v_mfma_f32_32x32x16_bf16 (32 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// interleaved placement: MFMA, V/M fillers, MFMA, V/M fillers ... pinned with sched_barrier
template <int TYPE, int M, int V>
__global__ __launch_bounds__(512) void ki(const float *in, float *out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = in[128 + ((threadIdx.x + i) & 63)];
    const float m = in[200], c = in[201];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (TYPE == 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
                else acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < V / M; ++j) f[j & 7] = __builtin_fmaf(f[j & 7], m, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TYPE, int M, int V>
__global__ __launch_bounds__(512) void k(const float *in, float *out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = in[128 + ((threadIdx.x + i) & 63)];
    const float m = in[200], c = in[201];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (M) {
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    if (TYPE == 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
                    else acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i & 3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) f[j & 7] = __builtin_fmaf(f[j & 7], m, c);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TYPE, int M, int V, bool IL = false>
double run(int wps, const float *in, float *out) {
    const int iters = 2048, threads = 256 * wps, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = IL ? ki<TYPE, (M ? M : 1), V> : k<TYPE, M, V>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1e3 / (iters * 4.0);        // microseconds per inner block (M MFMAs + V VALU per wave)
}

template <int TYPE, int M, int V>
void trio(const char *name, const float *in, float *out) {
    for (int wps = 1; wps <= 2; ++wps) {
        const double tm = run<TYPE, M, 0>(wps, in, out), tv = run<TYPE, 0, V>(wps, in, out), tb = run<TYPE, M, V>(wps, in, out);
        const double ti = run<TYPE, M, V, true>(wps, in, out);
        printf("%-28s waves/SIMD %d  M=%d V=%-3d : mfma %.4f  valu %.4f  both %.4f  interleaved %.4f us/block   both/(mfma+valu) %.2f  "
               "interleaved/(mfma+valu) %.2f  interleaved/max %.2f\n",
               name, wps, M, V, tm, tv, tb, ti, tb / (tm + tv), ti / (tm + tv), ti / (tm > tv ? tm : tv));
    }
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    h[200] = 0.999f; h[201] = 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    trio<0, 4, 16>("f32 32x32x2 (4 x 64 cyc)", in, out);
    trio<0, 4, 64>("f32 32x32x2 (4 x 64 cyc)", in, out);
    trio<1, 4, 16>("bf16 32x32x16 (4 x 32 cyc)", in, out);
    trio<1, 4, 32>("bf16 32x32x16 (4 x 32 cyc)", in, out);
    trio<0, 4, 32>("f32 32x32x2 (4 x 64 cyc)", in, out);
    return 0;
}
