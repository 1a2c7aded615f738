// What clock does a dense fp32-MFMA kernel really run at?  s_memtime counts shader-clock cycles, s_memrealtime a
// constant 100 MHz: their ratio over a ~2 ms MFMA loop is the shader clock the wave saw.
//   hipcc --offload-arch=gfx950 -O3 -o sclk_probe sclk_probe.hip && ./sclk_probe [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long *out, float *sink) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    unsigned long long *out, h[2];
    float *sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4);
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, 8000, out, sink);
        if (l % (launches / 10 > 0 ? launches / 10 : 1) == 0 || l == launches - 1) {
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("launch %4d: %llu shader cycles in %.1f us -> %.0f MHz\n", l, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
        }
    }
    return 0;
}
