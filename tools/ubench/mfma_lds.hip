// What keeps an LDS-fed v_mfma_f32_32x32x2_f32 loop below the register-only ceiling?  One consumer wave per SIMD
// (TK x TN accumulators), fragments from LDS, optional barrier per 16 row pairs, optional co-resident VALU waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TK = 2, TN = 4, LD = 384, RS = 32;

template <int MODE>   // bit0: LDS fragment reads, bit1: barrier per stripe, bit2: 4 extra waves doing VALU + LDS writes
__global__ __launch_bounds__(512, 1) void k(const float *in, float *out, int stripes, const float *bigbuf) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * RS * LD; e += blockDim.x) lds[e] = in[e % 4096];
    __syncthreads();
    if (wave >= 4) {
        if (!(MODE & 4)) {
            if (MODE & 2) for (int s = 0; s < stripes; ++s) __syncthreads();
            return;
        }
        float4 x[12];
        for (int j = 0; j < 12; ++j) x[j] = reinterpret_cast<const float4 *>(in)[(tid + 64 * j) % 1024];
        const int pt = tid - 256;
        const float4 *big = reinterpret_cast<const float4 *>(bigbuf) + (size_t)blockIdx.x * stripes * 3072;
        for (int s = 0; s < stripes; ++s) {
            float *dst = lds + (s & 1) * RS * LD;
            if (MODE & 16) {
#pragma unroll
                for (int j = 0; j < 12; ++j) x[j] = big[(size_t)s * 3072 + j * 256 + pt];
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                float4 v = x[j];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    v.x = fmaf(v.x, 1.0001f, 0.001f); v.y = fmaf(v.y, 0.9999f, 0.002f);
                    v.z = fmaf(v.z, 1.0002f, -0.001f); v.w = fmaf(v.w, 0.9998f, 0.003f);
                }
                x[j] = v;
                if (MODE & 8) *reinterpret_cast<float4 *>(&dst[(pt / 32 + j * 2 + (j >= 4 ? 0 : 0)) % RS * LD + (pt % 32) * 4 + (j % 3) * 128]) = v;
            }
            if (MODE & 2) __syncthreads();
        }
        out[tid] = x[0].x + x[11].w;
        return;
    }
    const int half = lane >> 5, li = lane & 31;
    const int ck = wave >> 1, cn = wave & 1;
    f32x16 acc[TK][TN];
    for (int i = 0; i < TK; ++i)
        for (int j = 0; j < TN; ++j)
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const int aoff = half * LD + ck * TK * 32 + li, doff = half * LD + 128 + cn * TN * 32 + li;
    float av_n[TK], dv_n[TN];
    for (int x = 0; x < TK; ++x) av_n[x] = lds[aoff + 32 * x];
    for (int y = 0; y < TN; ++y) dv_n[y] = lds[doff + 32 * y];
    for (int s = 0; s < stripes; ++s) {
        const float *sb = lds + (s & 1) * RS * LD;
#pragma unroll
        for (int it = 0; it < RS / 2; ++it) {
            float av[TK], dv[TN];
            for (int x = 0; x < TK; ++x) av[x] = av_n[x];
            for (int y = 0; y < TN; ++y) dv[y] = dv_n[y];
            if (MODE & 1) {
                const int nx = (it + 1) % (RS / 2);
                for (int x = 0; x < TK; ++x) av_n[x] = sb[aoff + 2 * nx * LD + 32 * x];
                for (int y = 0; y < TN; ++y) dv_n[y] = sb[doff + 2 * nx * LD + 32 * y];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < TK; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], dv[y], acc[x][y], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 2) __syncthreads();
    }
    float sum = 0.f;
    for (int i = 0; i < TK; ++i)
        for (int j = 0; j < TN; ++j)
            for (int v = 0; v < 16; ++v) sum += acc[i][j][v];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE>
void run(const float *in, float *out, const float *big) {
    const int stripes = 256;
    auto kern = k<MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = 2 * RS * LD * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, in, out, stripes, big);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, in, out, stripes, big);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = 256.0 * 4 * stripes * 16 * TK * TN * 4096.0;
    printf("mode %2d (lds %d barrier %d valu-waves %d lds-writes %d hbm-stream %d): %.3f ms  %.1f TF/s  (%.0f GB/s streamed)\n", MODE, MODE & 1, (MODE >> 1) & 1,
           (MODE >> 2) & 1, (MODE >> 3) & 1, (MODE >> 4) & 1, ms, flops / ms / 1e9, (MODE & 16) ? 256.0 * stripes * 49152 / ms / 1e6 : 0.0);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    float *big;
    hipMalloc(&big, (size_t)256 * 256 * 3072 * 16);   // 3.2 GB: 48 KB per stripe per workgroup
    hipMemset(big, 0x3c, (size_t)256 * 256 * 3072 * 16);
    run<0>(in, out, big); run<3>(in, out, big); run<7>(in, out, big); run<15>(in, out, big); run<23>(in, out, big); run<31>(in, out, big);
    run<20>(in, out, big);
    return 0;
}
