// Proof of concept for DESIGN.md section 10-0: Y[M][N] = A[M][K] W[K][N] (fp32 in, fp32 out) with the operands SPLIT into
// three bf16 pieces on the fly and six of the nine partial products on v_mfma_f32_32x32x16_bf16, next to the same loop on
// v_mfma_f32_32x32x2_f32.        hipcc --offload-arch=gfx950 -O3 -o gemm_bf16x3 gemm_bf16x3.hip && ./gemm_bf16x3
// One workgroup = 4 waves on a 128-column block of W (pieces resident in LDS in fragment order); a wave walks 32-row tiles,
// A straight from global memory in the MFMA's A layout (lane: row lane%32, eight consecutive k), split in registers.
// Checked against float64 on a sample of rows; timed with events.  Not the product path: no BN / ReLU prologue, no
// statistics, no pooling.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)x[e];
        const float r1 = x[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
    }
}

// W pieces in LDS, fragment order: Wf[piece][ks][nt][lane] = 8 bf16 (k = 16 ks + 8 (lane / 32) + e, column 32 nt + lane % 32)
// MODE 0: one accumulator per column block, the six products of a 16-k step back to back (smallest first)
//      1: TWO accumulators per block -- the h.h products alone in one, the five small ones in the other, added at the end
//      2: one accumulator, the h.h product FIRST in every step
//      3: two accumulators, all NINE products (the three extra ones with the small ones)
template <int K, int MODE>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3(long long M, int N, const float *__restrict__ A, const float *__restrict__ W,
                                                   float *__restrict__ Y) {
    constexpr int KS = K / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8 *Wf = reinterpret_cast<bf16x8 *>(smem);             // [3][KS][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * 128;
    for (int f = tid; f < KS * 4 * 64; f += 512) {
        const int ln = f & 63, nt = (f >> 6) & 3, ks = f >> 8;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = W[(long long)(16 * ks + 8 * (ln >> 5) + e) * N + n0 + 32 * nt + (ln & 31)];
        bf16x8 h, m, l;
        split3(x, h, m, l);
        Wf[(0 * KS + ks) * 256 + nt * 64 + ln] = h;
        Wf[(1 * KS + ks) * 256 + nt * 64 + ln] = m;
        Wf[(2 * KS + ks) * 256 + nt * 64 + ln] = l;
    }
    __syncthreads();
    const long long ntiles = (M + 31) / 32;
    for (long long t = (long long)blockIdx.x * 8 + wave; t < ntiles; t += (long long)gridDim.x * 8) {
        const long long row = t * 32 + (lane & 31);
        const float *ar = A + (row < M ? row : M - 1) * K + 8 * (lane >> 5);
        f32x16 acc[4], sm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = sm[i][v] = 0.f;
        float4 p0 = *reinterpret_cast<const float4 *>(ar), p1 = *reinterpret_cast<const float4 *>(ar + 4);
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            const float x[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            const int kn = ks + 1 < KS ? ks + 1 : ks;
            p0 = *reinterpret_cast<const float4 *>(ar + 16 * kn);
            p1 = *reinterpret_cast<const float4 *>(ar + 16 * kn + 4);
            bf16x8 ah, am, al;
            split3(x, ah, am, al);
            bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bh[nt] = Wf[(0 * KS + ks) * 256 + nt * 64 + lane];
                bm[nt] = Wf[(1 * KS + ks) * 256 + nt * 64 + lane];
                bl[nt] = Wf[(2 * KS + ks) * 256 + nt * 64 + lane];
            }
#define MM(A_, B_, C_) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) C_[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_[nt], C_[nt], 0, 0, 0)
            if (MODE == 0) {
                MM(al, bh, acc); MM(ah, bl, acc); MM(am, bm, acc); MM(am, bh, acc); MM(ah, bm, acc); MM(ah, bh, acc);
            } else if (MODE == 1) {
                MM(al, bh, sm); MM(ah, bl, sm); MM(am, bm, sm); MM(am, bh, sm); MM(ah, bm, sm); MM(ah, bh, acc);
            } else if (MODE == 2) {
                MM(ah, bh, acc); MM(ah, bm, acc); MM(am, bh, acc); MM(am, bm, acc); MM(ah, bl, acc); MM(al, bh, acc);
            } else {
                MM(al, bl, sm); MM(al, bm, sm); MM(am, bl, sm);
                MM(al, bh, sm); MM(ah, bl, sm); MM(am, bm, sm); MM(am, bh, sm); MM(ah, bm, sm); MM(ah, bh, acc);
            }
#undef MM
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = t * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                if (r < M) Y[r * N + n0 + 32 * nt + (lane & 31)] = (MODE == 1 || MODE == 3) ? acc[nt][v] + sm[nt][v] : acc[nt][v];
            }
    }
}

// the same loop on the fp32 matrix pipe: W resident in LDS as [k][128] fp32, A element per MFMA from global (k pair per lane)
template <int K>
__global__ __launch_bounds__(256) void gemm_f32(long long M, int N, const float *__restrict__ A, const float *__restrict__ W,
                                                float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Ws = reinterpret_cast<float *>(smem);               // [K][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * 128;
    for (int e = tid; e < K * 128; e += 256) Ws[e] = W[(long long)(e >> 7) * N + n0 + (e & 127)];
    __syncthreads();
    const long long ntiles = (M + 31) / 32;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += (long long)gridDim.x * 4) {
        const long long row = t * 32 + (lane & 31);
        const float *ar = A + (row < M ? row : M - 1) * K + (lane >> 5);
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
#pragma unroll 8
        for (int kk = 0; kk < K / 2; ++kk) {
            const float a = ar[2 * kk];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Ws[(2 * kk + (lane >> 5)) * 128 + 32 * nt + (lane & 31)], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = t * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                if (r < M) Y[r * N + n0 + 32 * nt + (lane & 31)] = acc[nt][v];
            }
    }
}

template <int K>
void run(long long M, int N) {
    float *A, *W, *Y;
    hipMalloc(&A, M * K * 4); hipMalloc(&W, (size_t)K * N * 4); hipMalloc(&Y, M * N * 4);
    std::vector<float> hA((size_t)4096 * K), hW((size_t)K * N);
    srand(7);
    for (auto &v : hA) { float r = (float)rand() / RAND_MAX * 2.f - 1.f; v = r > 0 ? r * 1.7f : 0.f; }      // post-ReLU-like
    for (auto &v : hW) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * sqrtf(6.f / (K + N));
    for (long long r = 0; r < M; r += 4096) hipMemcpy(A + r * K, hA.data(), (size_t)((M - r < 4096 ? M - r : 4096)) * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    const dim3 grid(512, N / 128), blk(256);
    const size_t lds3 = (size_t)3 * (K / 16) * 256 * 16, ldsf = (size_t)K * 128 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3<K, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3<K, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3<K, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3<K, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f32<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<float> y3((size_t)64 * N), yf((size_t)64 * N);
    static const char *names[5] = {"bf16x3, 6 products", "bf16x3, 6, two accum.", "bf16x3, 6, h.h first", "bf16x3, 9, two accum.", "fp32 MFMA"};
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 5;
    const int only = getenv("ONLY") ? atoi(getenv("ONLY")) : -1;      // one variant, many repetitions: clock / power polling
    for (int which = 0; which < 5; ++which) {
        if (only >= 0 && which != only) continue;
        auto launch = [&]() {
            if (which == 0) hipLaunchKernelGGL((gemm_bf16x3<K, 0>), dim3(256, N / 128), dim3(512), lds3, 0, M, N, A, W, Y);
            else if (which == 1) hipLaunchKernelGGL((gemm_bf16x3<K, 1>), dim3(256, N / 128), dim3(512), lds3, 0, M, N, A, W, Y);
            else if (which == 2) hipLaunchKernelGGL((gemm_bf16x3<K, 2>), dim3(256, N / 128), dim3(512), lds3, 0, M, N, A, W, Y);
            else if (which == 3) hipLaunchKernelGGL((gemm_bf16x3<K, 3>), dim3(256, N / 128), dim3(512), lds3, 0, M, N, A, W, Y);
            else hipLaunchKernelGGL(gemm_f32<K>, grid, blk, ldsf, 0, M, N, A, W, Y);
        };
        launch();
        hipDeviceSynchronize();
        hipMemcpy(y3.data(), Y, (size_t)64 * N * 4, hipMemcpyDeviceToHost);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        double num = 0, den = 0;
        const std::vector<float> &y = y3;
        for (int r = 0; r < 64; ++r)
            for (int n = 0; n < N; ++n) {
                double ex = 0;
                for (int k = 0; k < K; ++k) ex += (double)hA[(size_t)r * K + k] * hW[(size_t)k * N + n];
                num += (y[(size_t)r * N + n] - ex) * (y[(size_t)r * N + n] - ex);
                den += ex * ex;
            }
        const double flop = 2.0 * M * K * N, bytes = 4.0 * M * (K + N);
        printf("M=%lld K=%d N=%d  %-22s %8.1f us  %6.1f TF/s-equivalent (%.2f of the 157.3 fp32 MFMA peak)  %5.2f TB/s  rel. RMS error %.2e\n",
               M, K, N, names[which], ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3,
               bytes / ms / 1e9, sqrt(num / den));
    }
    hipFree(A); hipFree(W); hipFree(Y);
}

int main() {
    run<64>(4194304, 128);      // SA1 layer of the SSG config
    run<128>(2097152, 128);     // SA2
    run<128>(2097152, 256);
    return 0;
}
