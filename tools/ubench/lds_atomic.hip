// lds_atomic.hip -- rate of ds_add_f32 (no return) against ds_read + v_add + ds_write on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, const int *__restrict__ rows, float *out) {
    __shared__ float tile[8][32 * 68];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *t = tile[wave];
    for (int i = lane; i < 32 * 68; i += 64) t[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    float v = 1.f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = rows[(it * 16 + u) & 1023];          // wave-uniform row
            if (MODE == 0) atomicAdd(&t[r * 68 + lane], v);
            else if (MODE == 1) t[r * 68 + lane] += v;
            else if (MODE == 3) {                                 // integer atomics, same address pattern as MODE 0 (round 5)
                atomicAdd(reinterpret_cast<unsigned *>(&t[r * 68 + lane]), (unsigned)lane + 1u);
            } else if (MODE == 4) {                               // 64-bit integer atomics: 32 lanes' worth of 8-byte slots per row
                atomicAdd(reinterpret_cast<unsigned long long *>(&t[(r & 15) * 136]) + lane, (unsigned long long)lane + 1ull);
            } else if (MODE == 2) {                                 // 16 lanes x 4 components per row, 4 rows per instruction
                const int rr = rows[((it * 16 + u) * 4 + (lane >> 4)) & 1023];
                float *q = &t[rr * 68 + (lane & 15) * 4];
                atomicAdd(q, v); atomicAdd(q + 1, v); atomicAdd(q + 2, v); atomicAdd(q + 3, v);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
    for (int i = lane; i < 32 * 68; i += 64) s += t[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    int *rows; float *out;
    std::vector<int> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (i * 7 + (i >> 3)) & 31;
    hipMalloc(&rows, 4096); hipMemcpy(rows, h.data(), 4096, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](auto kern, const char *name, double lane_ops_per_u) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, 10, rows, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, rows, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 16 * 8;             // per CU (8 waves)
        printf("%s: %.3f ms, %.2f ns per wave-op per CU, %.2f lane-ops/clk/CU at 2.4 GHz\n", name, ms,
               ms * 1e6 / instr, instr * lane_ops_per_u / (ms * 1e-3 * 2.4e9));
    };
    run(k<0>, "ds_add_f32 row-uniform ", 64);
    run(k<1>, "read+add+write         ", 64);
    run(k<2>, "ds_add_f32 x4, 4 rows  ", 256);
    run(k<3>, "ds_add_u32 row-uniform ", 64);
    run(k<4>, "ds_add_u64 row-uniform ", 64);
    return 0;
}
