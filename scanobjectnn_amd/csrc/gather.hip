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

// Y[b,j,s,:] = Q[b,idx,:] + Ctr[b,j,:] + (xyz[b,idx,:] - new_xyz[b,j,:]) Wxyz + bias      (every term optional)
// The coordinate term is evaluated INLINE on the centred offsets -- exactly the reference's arithmetic for those
// three channels -- because pushing it through Q/Ctr would subtract two O(1) numbers to get an O(radius) one.
// block = 256 threads = RL row lanes x C4 column quads; one workgroup per `groups_per_block` groups of S rows
__global__ __launch_bounds__(256) void sa_gather_fwd_kernel(long long G, int n, int m, int S, int C,
                                                            const float *__restrict__ Q,
                                                            const float *__restrict__ Ctr,
                                                            const float *__restrict__ xyz,
                                                            const float *__restrict__ new_xyz,
                                                            const float *__restrict__ Wxyz,
                                                            const float *__restrict__ bias,
                                                            const int *__restrict__ idx, float *__restrict__ Y,
                                                            float *__restrict__ stats, int groups_per_block) {
    extern __shared__ float sm[];  // [RL][2][C]
    const int c4n = C / 4;
    const int RL = 256 / c4n;
    const int cq = (threadIdx.x % c4n) * 4, rl = threadIdx.x / c4n;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0, bb = w0;
    if (Wxyz) {
        w0 = *reinterpret_cast<const float4 *>(Wxyz + 0 * C + cq);
        w1 = *reinterpret_cast<const float4 *>(Wxyz + 1 * C + cq);
        w2 = *reinterpret_cast<const float4 *>(Wxyz + 2 * C + cq);
    }
    if (bias) bb = *reinterpret_cast<const float4 *>(bias + cq);
    if (rl < RL) {
        for (long long g = g0; g < g1; ++g) {
            const long long b = g / m;
            float4 ctr = bb;
            if (Ctr) {
                const float4 c4 = *reinterpret_cast<const float4 *>(Ctr + g * C + cq);
                ctr.x += c4.x; ctr.y += c4.y; ctr.z += c4.z; ctr.w += c4.w;
            }
            float cx = 0.f, cy = 0.f, cz = 0.f;
            if (Wxyz) { cx = new_xyz[g * 3 + 0]; cy = new_xyz[g * 3 + 1]; cz = new_xyz[g * 3 + 2]; }
            for (int s = rl; s < S; s += RL) {
                const long long r = g * S + s;
                const int i = idx[r];
                float4 y = ctr;
                if (Q) {
                    const float4 q = *reinterpret_cast<const float4 *>(Q + (b * n + i) * (long long)C + cq);
                    y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
                }
                if (Wxyz) {
                    const float *px = xyz + (b * n + i) * 3;
                    const float dx = px[0] - cx, dy = px[1] - cy, dz = px[2] - cz;
                    y.x = fmaf(dz, w2.x, fmaf(dy, w1.x, fmaf(dx, w0.x, y.x)));
                    y.y = fmaf(dz, w2.y, fmaf(dy, w1.y, fmaf(dx, w0.y, y.y)));
                    y.z = fmaf(dz, w2.z, fmaf(dy, w1.z, fmaf(dx, w0.z, y.z)));
                    y.w = fmaf(dz, w2.w, fmaf(dy, w1.w, fmaf(dx, w0.w, y.w)));
                }
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

// backward.  One lane per CHANNEL (a wave covers 64 consecutive floats of one row, so every atomic instruction is
// two full cache lines).  Rows that repeat the group's first index -- the padding ball query appends when fewer
// than S points are in range -- are summed in registers and leave as ONE atomic per channel, which removes the
// worst same-address contention.  wpart [gridDim.x][4][C]: partial (dWxyz rows 0..2, dbias).
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
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             float *__restrict__ dQ, float *__restrict__ dCtr,
                                                             float *__restrict__ wpart, int groups_per_block) {
    extern __shared__ float sm[];  // [RL][2][C] group scratch  +  [RL][4][C] weight-gradient scratch
    const int RL = C >= 256 ? 1 : 256 / C;          // row lanes
    const int rl = C >= 256 ? 0 : threadIdx.x / C;
    float *smw = sm + RL * 2 * C;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    const int cstart = threadIdx.x % (C >= 256 ? 256 : C);
    constexpr int MAXCI = 4;                          // C <= 1024
    float aw[MAXCI][4];
#pragma unroll
    for (int ci = 0; ci < MAXCI; ++ci) aw[ci][0] = aw[ci][1] = aw[ci][2] = aw[ci][3] = 0.f;
    for (long long g = g0; g < g1; ++g) {
        const long long b = g / m;
        const int first = idx[g * S];
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (xyz) { cx = new_xyz[g * 3 + 0]; cy = new_xyz[g * 3 + 1]; cz = new_xyz[g * 3 + 2]; }
#pragma unroll
        for (int ci = 0; ci < MAXCI; ++ci) {
            const int c = cstart + ci * 256;
            if (c >= C) break;
            const float cp = p[c], cq = q[c], ct = t[c];
            float cs = 0.f, ch = 0.f, gp = 0.f;
            int am = -1;
            if (POOLED) {
                cs = psc[c]; ch = psh[c];
                gp = gpool[g * C + c];
                am = argmax[g * C + c];
            }
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
                if (xyz) {
                    const float *px = xyz + (b * n + i) * 3;
                    aw[ci][0] = fmaf(px[0] - cx, d, aw[ci][0]);
                    aw[ci][1] = fmaf(px[1] - cy, d, aw[ci][1]);
                    aw[ci][2] = fmaf(px[2] - cz, d, aw[ci][2]);
                }
                if (dQ) {
                    if (i == first) dup += d;
                    else atomicAdd(dQ + (b * n + i) * (long long)C + c, d);
                }
            }
            aw[ci][3] += all;
            sm[(rl * 2 + 0) * C + c] = all;
            sm[(rl * 2 + 1) * C + c] = dup;
        }
        if (dQ || dCtr) {
            __syncthreads();
            for (int c = threadIdx.x; c < C; c += 256) {
                float a = 0.f, d = 0.f;
                for (int l = 0; l < RL; ++l) {
                    a += sm[(l * 2 + 0) * C + c];
                    d += sm[(l * 2 + 1) * C + c];
                }
                if (dCtr) dCtr[g * C + c] = a;
                if (dQ) atomicAdd(dQ + (b * n + first) * (long long)C + c, d);
            }
            __syncthreads();
        }
    }
    if (wpart) {
#pragma unroll
        for (int ci = 0; ci < MAXCI; ++ci) {
            const int c = cstart + ci * 256;
            if (c >= C) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) smw[(rl * 4 + e) * C + c] = aw[ci][e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * C; i += 256) {
            const int e = i / C, c = i % C;
            float v = 0.f;
            for (int l = 0; l < RL; ++l) v += smw[(l * 4 + e) * C + c];
            wpart[((long long)blockIdx.x * 4 + e) * C + c] = v;
        }
    }
}

// out[L] = sum_p part[p][L] in double (deterministic)
__global__ __launch_bounds__(256) void sum_rows_kernel(int P, int L, const float *__restrict__ part,
                                                       float *__restrict__ out) {
    __shared__ double sm[256];
    const int i = blockIdx.x;      // one workgroup per output element
    double s = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) s += (double)part[(long long)p * L + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = (float)sm[0];
}

}  // namespace

extern "C" {

int pcops_sa_gather_stats_rows(long long G) { return (int)((G + 7) / 8); }
int pcops_sa_scatter_rows(long long G) { return (int)((G + 15) / 16); }

int pcops_sa_gather_fwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const float *xyz,
                        const float *new_xyz, const float *Wxyz, const float *bias, const int *idx, float *Y,
                        float *stats_partial, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && c >= 4 && c % 4 == 0);
    PCOPS_REQUIRE_SHAPE(c <= 1024 && 256 % (c / 4) == 0);
    const long long G = (long long)b * m;
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_ARG(Q != nullptr || Wxyz != nullptr);
    if (Wxyz) { PCOPS_REQUIRE_PTR(xyz); PCOPS_REQUIRE_PTR(new_xyz); }
    const int rl = 256 / (c / 4);
    hipLaunchKernelGGL(sa_gather_fwd_kernel, dim3(pcops_sa_gather_stats_rows(G)), dim3(256),
                       (size_t)rl * 2 * c * sizeof(float), as_stream(stream), G, n, m, s, c, Q, Ctr, xyz, new_xyz,
                       Wxyz, bias, idx, Y, stats_partial, 8);
    return pcops_launch_status();
}

int pcops_sa_scatter_bwd(int b, int n, int m, int s, int c, const float *G, const float *Y, const float *p,
                         const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                         const float *pool_scale, const float *pool_shift, const int *idx, const float *xyz,
                         const float *new_xyz, float *dQ, float *dCtr, float *wpartial, float *dWxyz,
                         float *dbias, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && c >= 4 && c % 4 == 0);
    PCOPS_REQUIRE_SHAPE(c <= 1024 && (c >= 256 || 256 % c == 0));
    const long long Gn = (long long)b * m;
    hipStream_t st = as_stream(stream);
    if (dQ && hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess) return PCOPS_ERR_LAUNCH;
    if (Gn == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(idx);
    if (xyz) { PCOPS_REQUIRE_PTR(new_xyz); }
    if (dWxyz || dbias) PCOPS_REQUIRE_PTR(wpartial);
    const int rl = c >= 256 ? 1 : 256 / c;
    const size_t lds = (size_t)rl * 6 * c * sizeof(float);
    const int gpb = 16;
    const unsigned grid = pcops_sa_scatter_rows(Gn);
    float *wp = (dWxyz || dbias) ? wpartial : nullptr;
    if (gpool) {
        PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale); PCOPS_REQUIRE_PTR(pool_shift);
        PCOPS_REQUIRE_SHAPE(s <= 256);
        hipLaunchKernelGGL((sa_scatter_bwd_kernel<true>), dim3(grid), dim3(256), lds, st, Gn, n, m, s, c, G, Y, p, q,
                           t, gpool, argmax, pool_scale, pool_shift, idx, xyz, new_xyz, dQ, dCtr, wp, gpb);
    } else {
        PCOPS_REQUIRE_PTR(G);
        hipLaunchKernelGGL((sa_scatter_bwd_kernel<false>), dim3(grid), dim3(256), lds, st, Gn, n, m, s, c, G, Y, p, q,
                           t, gpool, argmax, pool_scale, pool_shift, idx, xyz, new_xyz, dQ, dCtr, wp, gpb);
    }
    int rc = pcops_launch_status();
    if (rc) return rc;
    if (wp) {
        // wpartial rows are [block][4][c]: element e*c + col -> strided view: sum over blocks
        if (dWxyz) hipLaunchKernelGGL(sum_rows_kernel, dim3(3 * c), dim3(256), 0, st, (int)grid, 4 * c, wp, dWxyz);
        if (dbias) hipLaunchKernelGGL(sum_rows_kernel, dim3(c), dim3(256), 0, st, (int)grid, 4 * c, wp + 3 * c, dbias);
        rc = pcops_launch_status();
    }
    return rc;
}

}  // extern "C"
