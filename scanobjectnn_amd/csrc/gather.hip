// gather.hip -- the FIRST layer of a grouped shared MLP, moved in front of the grouping.
//
// The reference materialises the grouped tensor and runs the first 1x1 conv on every (point, sample) row:
//   set abstraction   new_points = concat(xyz[idx] - new_xyz, points[idx])  -> conv2d     (pointnet2/utils/
//                     pointnet_util.py:44-54,117-122; [feats | xyz] order in the MSG module :180-189)
//   EdgeConv          edge = concat(x_i, x_j - x_i) -> conv2d                             (dgcnn/utils/tf_util.py:
//                     699-705, dgcnn/models/dgcnn.py:39-44)
// A 1x1 conv is linear, so  concat(a[idx] - c, f[idx]) W  =  (a W_a + f W_f)[idx] - c W_a : the contraction runs
// ONCE per source point (B*N rows, a small library GEMM on the host side) and the big (B, M, S, C) tensor is
// produced by a gather + add:      Y[b, j, s, :] = Q[b, idx[b, j, s], :] + Ctr[b, j, :]
// with Q = a W_a + f W_f + bias and Ctr = -new_xyz W_a (SA) or x (W_i - W_d) (EdgeConv, Q = x W_d + bias).
// Same math up to fp32 rounding; 16x (S=16) to 128x fewer multiply-adds for that layer, no 3+C wide concat
// tensor, and the backward is one scatter-add of dY (what group_point_grad already had to do).
//
//   sa_gather_fwd : Y = Q[idx] + Ctr, plus the per-channel (sum, sum of squares) partials batch-norm needs
//   sa_scatter_bwd: dY = p.G + q.Y + t rebuilt on the fly (same contract as mlp.hip), dQ += scatter(dY)
//                   (fp32 atomics, L2 resident), dCtr[b, j, :] = sum_s dY (in-block, no atomics)
#include "common.h"

namespace {

// block = 256 threads = RL row lanes x C4 column quads; one workgroup per `groups_per_block` groups of S rows
__global__ __launch_bounds__(256) void sa_gather_fwd_kernel(long long G, int n, int m, int S, int C,
                                                            const float *__restrict__ Q,
                                                            const float *__restrict__ Ctr,
                                                            const int *__restrict__ idx, float *__restrict__ Y,
                                                            float *__restrict__ stats, int groups_per_block) {
    extern __shared__ float sm[];  // [RL][2][C]
    const int c4n = C / 4;
    const int RL = 256 / c4n;
    const int cq = (threadIdx.x % c4n) * 4, rl = threadIdx.x / c4n;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
        for (long long g = g0; g < g1; ++g) {
            const long long b = g / m;
            const float4 ctr = *reinterpret_cast<const float4 *>(Ctr + g * C + cq);
            const float *qb = Q + b * n * (long long)C + cq;
            for (int s = rl; s < S; s += RL) {
                const long long r = g * S + s;
                const int i = idx[r];
                const float4 q = *reinterpret_cast<const float4 *>(qb + (long long)i * C);
                float4 y;
                y.x = q.x + ctr.x; y.y = q.y + ctr.y; y.z = q.z + ctr.z; y.w = q.w + ctr.w;
                *reinterpret_cast<float4 *>(Y + r * C + cq) = y;
                s1[0] += y.x; s1[1] += y.y; s1[2] += y.z; s1[3] += y.w;
                s2[0] = fmaf(y.x, y.x, s2[0]); s2[1] = fmaf(y.y, y.y, s2[1]);
                s2[2] = fmaf(y.z, y.z, s2[2]); s2[3] = fmaf(y.w, y.w, s2[3]);
            }
        }
    }
    if (stats == nullptr) return;
    if (rl < RL)
        for (int e = 0; e < 4; ++e) {
            sm[(rl * 2 + 0) * C + cq + e] = s1[e];
            sm[(rl * 2 + 1) * C + cq + e] = s2[e];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        float t = 0.f;
        for (int l = 0; l < RL; ++l) t += sm[(l * 2 + which) * C + c];
        stats[((long long)blockIdx.x * 2 + which) * C + c] = t;
    }
}

// one lane per CHANNEL (a wave covers 64 consecutive floats of one row, so every atomic instruction is two full
// cache lines); 256/C... rows in flight per workgroup.  Rows that repeat the group's first index -- the padding
// ball query appends when fewer than S points are in range -- are summed in registers and leave as ONE atomic
// per channel, which removes the worst same-address contention.
template <bool POOLED>
__global__ __launch_bounds__(256) void sa_scatter_bwd_kernel(long long G, int n, int m, int S, int C,
                                                             const float *__restrict__ Gm,
                                                             const float *__restrict__ Y,
                                                             const float *__restrict__ p,
                                                             const float *__restrict__ q,
                                                             const float *__restrict__ t,
                                                             const float *__restrict__ gpool,
                                                             const unsigned char *__restrict__ argmax,
                                                             const float *__restrict__ psc,
                                                             const float *__restrict__ psh,
                                                             const int *__restrict__ idx,
                                                             float *__restrict__ dQ, float *__restrict__ dCtr,
                                                             int groups_per_block) {
    extern __shared__ float sm[];  // [RL][2][C]: per row lane (sum of all dY, sum of the first-index dY)
    const int RL = C >= 256 ? 1 : 256 / C;          // row lanes
    const int rl = C >= 256 ? 0 : threadIdx.x / C;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    for (long long g = g0; g < g1; ++g) {
        const long long b = g / m;
        const int first = idx[g * S];
        for (int c = threadIdx.x % (C >= 256 ? 256 : C); c < C; c += 256) {
            const float cp = p[c], cq = q[c], ct = t[c];
            float cs = 0.f, ch = 0.f, gp = 0.f;
            int am = -1;
            if (POOLED) {
                cs = psc[c]; ch = psh[c];
                gp = gpool[g * C + c];
                am = argmax[g * C + c];
            }
            float *dqb = dQ + b * n * (long long)C + c;
            float all = 0.f, dup = 0.f;
            for (int s = rl; s < S; s += RL) {
                const long long r = g * S + s;
                const float y = Y[r * C + c];
                float gm;
                if (POOLED) gm = (am == s && fmaf(y, cs, ch) > 0.f) ? gp : 0.f;
                else gm = Gm[r * C + c];
                const float d = fmaf(cp, gm, fmaf(cq, y, ct));
                all += d;
                const int i = idx[r];
                if (i == first) dup += d;
                else atomicAdd(dqb + (long long)i * C, d);
            }
            sm[(rl * 2 + 0) * C + c] = all;
            sm[(rl * 2 + 1) * C + c] = dup;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f, d = 0.f;
            for (int l = 0; l < RL; ++l) {
                a += sm[(l * 2 + 0) * C + c];
                d += sm[(l * 2 + 1) * C + c];
            }
            dCtr[g * C + c] = a;
            atomicAdd(dQ + (b * n + first) * (long long)C + c, d);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int pcops_sa_gather_stats_rows(long long G) { return (int)((G + 7) / 8); }

/* Y[b,j,s,:] = Q[b, idx[b,j,s], :] + Ctr[b,j,:];  Q (b,n,c), Ctr (b,m,c), idx (b,m,s) -> Y (b,m,s,c).
 * stats_partial (may be NULL): float [pcops_sa_gather_stats_rows(b*m)][2][c] partial (sum Y, sum Y*Y). */
int pcops_sa_gather_fwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const int *idx,
                        float *Y, float *stats_partial, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && c >= 4 && c % 4 == 0);
    PCOPS_REQUIRE_SHAPE(c <= 1024 && 256 % (c / 4) == 0);
    const long long G = (long long)b * m;
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(Y);
    const int rl = 256 / (c / 4);
    hipLaunchKernelGGL(sa_gather_fwd_kernel, dim3(pcops_sa_gather_stats_rows(G)), dim3(256),
                       (size_t)rl * 2 * c * sizeof(float), as_stream(stream), G, n, m, s, c, Q, Ctr, idx, Y,
                       stats_partial, 8);
    return pcops_launch_status();
}

/* backward of the above through the following BN+ReLU: dY = p.G + q.Y + t (or the pooled form when
 * gpool != NULL, as in pcops_mlp_gemm_dgrad); dQ (b,n,c) = scatter-add over idx (zeroed here), dCtr (b,m,c)
 * = sum over s. */
int pcops_sa_scatter_bwd(int b, int n, int m, int s, int c, const float *G, const float *Y, const float *p,
                         const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                         const float *pool_scale, const float *pool_shift, const int *idx, float *dQ,
                         float *dCtr, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && c >= 4 && c % 4 == 0);
    PCOPS_REQUIRE_SHAPE(c <= 1024 && 256 % (c / 4) == 0);
    const long long Gn = (long long)b * m;
    PCOPS_REQUIRE_PTR(dQ);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess) return PCOPS_ERR_LAUNCH;
    if (Gn == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(dCtr);
    PCOPS_REQUIRE_SHAPE(c >= 256 || 256 % c == 0);
    const int rl = c >= 256 ? 1 : 256 / c;
    const size_t lds = (size_t)rl * 2 * c * sizeof(float);
    const int gpb = 4;
    const unsigned grid = cdiv(Gn, gpb);
    if (gpool) {
        PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale); PCOPS_REQUIRE_PTR(pool_shift);
        PCOPS_REQUIRE_SHAPE(s <= 256);
        hipLaunchKernelGGL((sa_scatter_bwd_kernel<true>), dim3(grid), dim3(256), lds, st, Gn, n, m, s, c, G, Y, p, q,
                           t, gpool, argmax, pool_scale, pool_shift, idx, dQ, dCtr, gpb);
    } else {
        PCOPS_REQUIRE_PTR(G);
        hipLaunchKernelGGL((sa_scatter_bwd_kernel<false>), dim3(grid), dim3(256), lds, st, Gn, n, m, s, c, G, Y, p, q,
                           t, gpool, argmax, pool_scale, pool_shift, idx, dQ, dCtr, gpb);
    }
    return pcops_launch_status();
}

}  // extern "C"
