// Practical ceiling of v_mfma_f32_32x32x2_f32 on this chip: W waves per SIMD, A independent accumulators per wave,
// operands from registers (random data: DVFS depends on toggling).   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int A>
__global__ void k(const float *in, float *out, int iters) {
    f32x16 acc[A];
    for (int i = 0; i < A; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (i + 4)]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < A; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r + i) & 3], b[(r * 3 + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < A; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int A>
void run(int wps, const float *in, float *out) {
    const int iters = 4096;
    const int threads = 64 * 4 * wps;      // waves per SIMD * 4 SIMDs
    const int blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<A>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<A>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = (double)blocks * (threads / 64) * iters * 4.0 * A * 4096.0;
    printf("waves/SIMD %d  acc %d : %.3f ms  %.1f TF/s\n", wps, A, ms, flops / ms / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 4 * 512 * 4);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
        run<2>(wps, in, out);
        run<4>(wps, in, out);
        run<8>(wps, in, out);
    }
    // zero data: the DVFS give-back
    hipMemset(in, 0, 4096 * 4);
    run<8>(1, in, out);
    return 0;
}
