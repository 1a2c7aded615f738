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
#include <stdlib.h>

#include "common.h"
#include "edgeconv.h"

namespace {

// ---- compacted rows (pcops_rows_t, see pcops.h "compacted rows") -------------------------------------------------
constexpr int kBlk = 16;
struct RowBlock { int g, s0; float w; int pad; };

__device__ __forceinline__ int rows_blocks_of(int cnt, int S) {
    const int c = cnt < 1 ? 1 : cnt;                       // a neighbourhood without any hit is S copies of index 0
    const int nb = (c + kBlk - 1) / kBlk;
    return nb < S / kBlk ? nb : S / kBlk;
}

// block_start = exclusive scan of the blocks per group, rows = 16 * total; one workgroup, contiguous runs per thread
__global__ __launch_bounds__(1024) void rows_plan_scan_kernel(int G, int S, const int *__restrict__ cnt,
                                                              int *__restrict__ bstart, int *__restrict__ rows) {
    // one workgroup, every thread a contiguous run of `per` groups.  The run is read ONCE, 16 bytes at a time, and kept in
    // registers for the second pass (round 3: the scalar form read every count twice with a 128-byte stride between
    // neighbouring threads -- 53 us for 32 768 groups, most of it memory latency)
    __shared__ int sc[1024];
    constexpr int MAXV = 16;                         // up to 64 groups per thread in registers (G <= 65 536)
    const int tid = threadIdx.x;
    const int per = (G + 1023) / 1024;
    const int g0 = tid * per, g1 = min(G, g0 + per);
    const bool vec = per % 4 == 0 && per <= 4 * MAXV && ((reinterpret_cast<uintptr_t>(cnt) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(bstart) & 15) == 0);
    int4 nb[MAXV];
    int local = 0;
    if (vec) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int4 c = make_int4(0, 0, 0, 0);
            const int g = g0 + 4 * i;
            if (4 * i < per && g < G) {              // (G is a multiple of 4 here only if the last run is whole: checked per lane)
                if (g + 3 < G) c = *reinterpret_cast<const int4 *>(cnt + g);
                else { c.x = cnt[g]; c.y = g + 1 < G ? cnt[g + 1] : 0; c.z = g + 2 < G ? cnt[g + 2] : 0; }
                c.x = rows_blocks_of(c.x, S);
                c.y = g + 1 < G ? rows_blocks_of(c.y, S) : 0;
                c.z = g + 2 < G ? rows_blocks_of(c.z, S) : 0;
                c.w = g + 3 < G ? rows_blocks_of(c.w, S) : 0;
            }
            nb[i] = c;
            local += (c.x + c.y) + (c.z + c.w);
        }
    } else {
        for (int g = g0; g < g1; ++g) local += rows_blocks_of(cnt[g], S);
    }
    sc[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? sc[tid - off] : 0;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    int run = sc[tid] - local;
    if (vec) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int g = g0 + 4 * i;
            if (4 * i < per && g < G) {
                const int4 c = nb[i];
                const int4 o = make_int4(run, run + c.x, run + c.x + c.y, run + c.x + c.y + c.z);
                if (g + 3 < G) *reinterpret_cast<int4 *>(bstart + g) = o;
                else { bstart[g] = o.x; if (g + 1 < G) bstart[g + 1] = o.y; if (g + 2 < G) bstart[g + 2] = o.z; }
                run += (c.x + c.y) + (c.z + c.w);
            }
        }
    } else {
        for (int g = g0; g < g1; ++g) {
            bstart[g] = run;
            run += rows_blocks_of(cnt[g], S);
        }
    }
    if (tid == 1023) {
        bstart[G] = sc[1023];
        rows[0] = sc[1023] * kBlk;
    }
}

__global__ __launch_bounds__(256) void rows_plan_fill_kernel(long long total, int S, const int *__restrict__ cnt,
                                                             const int *__restrict__ bstart,
                                                             RowBlock *__restrict__ blocks) {
    const int bpg = S / kBlk;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int g = (int)(e / bpg), k = (int)(e - (long long)g * bpg);
        const int nb = rows_blocks_of(cnt[g], S);
        if (k < nb) {
            RowBlock rb;
            rb.g = g;
            rb.s0 = k * kBlk;
            rb.w = k == 0 ? (float)(S - nb * kBlk + 1) : 1.f;      // row 0 also stands for the copies that were dropped
            rb.pad = 0;
            blocks[bstart[g] + k] = rb;
        }
    }
}

// one quad of Y[b,j,s,:] in the forward's own operation order -- the backward kernels REBUILD the first layer's
// output with it instead of reading it back (bit-identical, and one (b,m,s,c) tensor less to stream from HBM):
//   y = (bias + Ctr) + Q;  y = fma(dz, w2, fma(dy, w1, fma(dx, w0, y)))
__device__ __forceinline__ float4 first_layer_quad(float4 ctr, bool has_q, float4 q, bool has_w, float dx, float dy,
                                                   float dz, float4 w0, float4 w1, float4 w2) {
    float4 y = ctr;
    if (has_q) { y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w; }
    if (has_w) {
        y.x = fmaf(dz, w2.x, fmaf(dy, w1.x, fmaf(dx, w0.x, y.x)));
        y.y = fmaf(dz, w2.y, fmaf(dy, w1.y, fmaf(dx, w0.y, y.y)));
        y.z = fmaf(dz, w2.z, fmaf(dy, w1.z, fmaf(dx, w0.z, y.z)));
        y.w = fmaf(dz, w2.w, fmaf(dy, w1.w, fmaf(dx, w0.w, y.w)));
    }
    return y;
}

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
                                                            float *__restrict__ off4, float *__restrict__ stats,
                                                            const float *__restrict__ pivot,
                                                            float *__restrict__ moments, int groups_per_block,
                                                            const RowBlock *__restrict__ blocks,
                                                            const int *__restrict__ bstart, int nt) {
    // nt: Y leaves with non-temporal stores (a stream of 256 MB and more that nothing reads back soon, mlp.hip buf_store4)
    // blocks != NULL: compacted rows -- group g writes its first 16 (bstart[g+1] - bstart[g]) rows to rows
    // 16 bstart[g] ..., and its row 0 enters the statistics / moments with the weight of the copies left out
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [RL][2][C] statistics scratch | staged rows
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
    // statistics relative to a per-channel pivot (shifted moments, pcops.h pcops_mlp_gemm_fwd): sums of y - pv
    const float4 pv = (stats && pivot) ? *reinterpret_cast<const float4 *>(pivot + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    // rows are staged first, ONE thread per grouped row: index + centred offsets go to LDS (and to off4), so the
    // C/4 lanes that then share a row read them as an LDS broadcast instead of each issuing its own global loads
    float4 *st4 = reinterpret_cast<float4 *>(sm + RL * 2 * C);        // (dx, dy, dz, index bits) per staged row
    const int gch = S >= 1024 ? 1 : 1024 / S;                         // groups per staging chunk
    float mo[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // offset moments: xx xy xz yy yz zz | x y z
    for (long long gb = g0; gb < g1; gb += gch) {
        const long long ge = min(g1, gb + gch);
        const int nrows = (int)(ge - gb) * S;
        for (int t = threadIdx.x; t < nrows; t += 256) {
            const long long r = gb * S + t;
            const long long g = gb + t / S;
            const long long b = g / m;
            const int i = idx[r];
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (Wxyz) {
                const float *px = xyz + (b * n + i) * 3;
                dx = px[0] - new_xyz[g * 3 + 0]; dy = px[1] - new_xyz[g * 3 + 1]; dz = px[2] - new_xyz[g * 3 + 2];
            }
            st4[t] = make_float4(dx, dy, dz, __int_as_float(i));
            long long ro = r;                 // output row
            float wt = 1.f;                   // weight of the row in the sums
            bool keep = true;
            if (blocks) {
                const int sg = t % S, b0 = bstart[g];
                keep = sg < (bstart[g + 1] - b0) * kBlk;
                ro = (long long)b0 * kBlk + sg;
                if (sg == 0) wt = blocks[b0].w;
            }
            if (!keep) continue;
            if (off4) *reinterpret_cast<float4 *>(off4 + ro * 4) = make_float4(dx, dy, dz, 0.f);
            if (moments) {
                const float wx = wt * dx, wy = wt * dy, wz = wt * dz;
                mo[0] = fmaf(wx, dx, mo[0]); mo[1] = fmaf(wx, dy, mo[1]); mo[2] = fmaf(wx, dz, mo[2]);
                mo[3] = fmaf(wy, dy, mo[3]); mo[4] = fmaf(wy, dz, mo[4]); mo[5] = fmaf(wz, dz, mo[5]);
                mo[6] += wx; mo[7] += wy; mo[8] += wz;
            }
        }
        __syncthreads();
        if (rl < RL) {
            for (long long g = gb; g < ge; ++g) {
                const long long b = g / m;
                float4 ctr = bb;
                if (Ctr) {
                    const float4 c4 = *reinterpret_cast<const float4 *>(Ctr + g * C + cq);
                    ctr.x += c4.x; ctr.y += c4.y; ctr.z += c4.z; ctr.w += c4.w;
                }
                const float4 *sg = st4 + (g - gb) * S;
                int Sg = S;
                long long rbase = g * S;
                float w0row = 1.f;
                if (blocks) {
                    const int b0 = bstart[g];
                    Sg = (bstart[g + 1] - b0) * kBlk;
                    rbase = (long long)b0 * kBlk;
                    w0row = blocks[b0].w;
                }
                for (int s = rl; s < Sg; s += RL) {
                    const long long r = rbase + s;
                    const float4 e = sg[s];
                    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (Q) q = *reinterpret_cast<const float4 *>(Q + (b * n + __float_as_int(e.w)) * (long long)C + cq);
                    const float4 y = first_layer_quad(ctr, Q != nullptr, q, Wxyz != nullptr, e.x, e.y, e.z, w0, w1, w2);
                    if (Y) {                                                      // NULL: statistics only
                        typedef float g_f4 __attribute__((ext_vector_type(4)));
                        if (nt) __builtin_nontemporal_store(g_f4{y.x, y.y, y.z, y.w}, reinterpret_cast<g_f4 *>(Y + r * C + cq));
                        else *reinterpret_cast<float4 *>(Y + r * C + cq) = y;
                    }
                    const float wt = s == 0 ? w0row : 1.f;
                    const float4 d = make_float4(y.x - pv.x, y.y - pv.y, y.z - pv.z, y.w - pv.w);
                    const float4 wy = make_float4(wt * d.x, wt * d.y, wt * d.z, wt * d.w);
                    s1[0] += wy.x; s1[1] += wy.y; s1[2] += wy.z; s1[3] += wy.w;
                    s2[0] = fmaf(wy.x, d.x, s2[0]); s2[1] = fmaf(wy.y, d.y, s2[1]);
                    s2[2] = fmaf(wy.z, d.z, s2[2]); s2[3] = fmaf(wy.w, d.w, s2[3]);
                }
            }
        }
        __syncthreads();
    }
    if (moments) {          // block sums of the nine offset moments (the staging area is free again)
        float *mm = reinterpret_cast<float *>(st4);
#pragma unroll
        for (int k = 0; k < 9; ++k) mm[k * 256 + threadIdx.x] = mo[k];
        __syncthreads();
        if (threadIdx.x < 9) {
            float t = 0.f;
            for (int j = 0; j < 256; ++j) t += mm[threadIdx.x * 256 + j];
            moments[(long long)blockIdx.x * 9 + threadIdx.x] = t;
        }
        __syncthreads();
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

// backward, LDS-accumulating variant (the default).  One workgroup owns ONE cloud b and a slice of CS channels:
// the cloud's slice of dQ (n x CS floats) lives in LDS, every row's dY is added there with LDS atomics (no
// same-address traffic to L2 at all), and the slice leaves once, with plain coalesced stores.  A wave walks whole
// groups (so dCtr is a register sum + lane reduction), lanes = (row, column quad) with 16-byte loads, U row
// chunks in flight.  The cloud's xyz sits in LDS too (the offsets of the inline coordinate term).
// Workgroups of the same cloud are spaced 8 apart in the launch order: same XCD, so the 128-byte lines two
// slices share are fetched from HBM once.   wpart [b][4][C]: per-cloud partial (dWxyz rows 0..2, dbias).
struct ScatterArgs {
    int b, n, m, S, C, nsl, gsplit;     // gsplit > 1 (streaming use only, dQ == NULL): groups of a cloud dealt to
                                        // several workgroups; wpart rows are then [b * gsplit]
    const float *Gm, *Y, *p, *q, *t, *gpool;
    const unsigned char *argmax;
    const float *psc, *psh;
    const int *idx;
    const float *xyz, *new_xyz;
    float *dQ, *dCtr, *wpart;
    float *dQarg;                       // pooled form: global dQ receiving p.gpool at the arg-max rows (atomics)
    const float *fQ, *fCtr, *fW, *fbias;    // RC kernels: the forward's sources, Y is rebuilt instead of read
};

template <bool POOLED, int CS, bool RC>
__global__ __launch_bounds__(1024) void sa_scatter_lds_kernel(ScatterArgs a) {
    constexpr int LPR = CS / 4;        // lanes per row (one float4 each)
    constexpr int RW = 64 / LPR;       // rows per wave instruction
    constexpr int U = 4;               // row chunks in flight
    constexpr int NW = 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // launch order -> (cloud, slice): ids 8 apart share a cloud
    const int L = blockIdx.x / a.gsplit, part = blockIdx.x % a.gsplit;
    const int blo = L & 7, rest = L >> 3;
    const int sl = rest % a.nsl, b = (rest / a.nsl) * 8 + blo;
    if (b >= a.b) return;
    const int n = a.n, m = a.m, S = a.S, C = a.C;
    float *acc = sm;                                            // [n][CS]   (dQ slice)  | reduction scratch
    const int accn = a.dQ ? n * CS : NW * 4 * CS;
    float *sx = sm + (accn > NW * 4 * CS ? accn : NW * 4 * CS);  // [n][3]
    if (a.dQ)
        for (int e = tid; e < n * CS / 4; e += 1024)
            reinterpret_cast<float4 *>(acc)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.xyz)
        for (int e = tid; e < n * 3; e += 1024) sx[e] = a.xyz[(long long)b * n * 3 + e];
    __syncthreads();

    const int rsub = lane / LPR, quad = lane % LPR;
    const int c0 = sl * CS + quad * 4;
    const float4 cp = *reinterpret_cast<const float4 *>(a.p + c0);
    const float4 cq = *reinterpret_cast<const float4 *>(a.q + c0);
    const float4 ct = *reinterpret_cast<const float4 *>(a.t + c0);
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f), ch4 = cs4;
    if (POOLED) {
        cs4 = *reinterpret_cast<const float4 *>(a.psc + c0);
        ch4 = *reinterpret_cast<const float4 *>(a.psh + c0);
    }
    float aw[4][4];                    // [x, y, z, 1][channel of the quad]
#pragma unroll
    for (int i = 0; i < 4; ++i) aw[i][0] = aw[i][1] = aw[i][2] = aw[i][3] = 0.f;
    float4 fw0 = make_float4(0.f, 0.f, 0.f, 0.f), fw1 = fw0, fw2 = fw0, fbb = fw0;
    if (RC) {
        if (a.fW) {
            fw0 = *reinterpret_cast<const float4 *>(a.fW + 0 * C + c0);
            fw1 = *reinterpret_cast<const float4 *>(a.fW + 1 * C + c0);
            fw2 = *reinterpret_cast<const float4 *>(a.fW + 2 * C + c0);
        }
        if (a.fbias) fbb = *reinterpret_cast<const float4 *>(a.fbias + c0);
    }

    const int jper = (m + a.gsplit - 1) / a.gsplit;
    const int jbeg = part * jper, jend = min(m, jbeg + jper);
    for (int j = jbeg + wave; j < jend; j += NW) {
        const long long g = (long long)b * m + j;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (a.xyz) { cx = a.new_xyz[g * 3 + 0]; cy = a.new_xyz[g * 3 + 1]; cz = a.new_xyz[g * 3 + 2]; }
        float4 gp = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned am = 0;
        if (POOLED) {
            gp = *reinterpret_cast<const float4 *>(a.gpool + g * C + c0);
            am = *reinterpret_cast<const unsigned *>(a.argmax + g * C + c0);
        }
        float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 fctr = fbb;
        if (RC && a.fCtr) {
            const float4 c4 = *reinterpret_cast<const float4 *>(a.fCtr + g * C + c0);
            fctr.x += c4.x; fctr.y += c4.y; fctr.z += c4.z; fctr.w += c4.w;
        }
        for (int s0 = 0; s0 < S; s0 += RW * U) {
            int ii[U];
            float4 yy[U], gg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u * RW + rsub;
                const long long r = g * S + (s < S ? s : S - 1);
                ii[u] = a.idx[r];
                if (!RC) yy[u] = *reinterpret_cast<const float4 *>(a.Y + r * C + c0);
                if (!POOLED) gg[u] = *reinterpret_cast<const float4 *>(a.Gm + r * C + c0);
            }
            if (RC && a.fQ) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    yy[u] = *reinterpret_cast<const float4 *>(a.fQ + ((long long)b * n + ii[u]) * C + c0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u * RW + rsub;
                float4 y = yy[u];
                if (RC) {
                    float dx = 0.f, dy = 0.f, dz = 0.f;
                    if (a.fW) { dx = sx[ii[u] * 3 + 0] - cx; dy = sx[ii[u] * 3 + 1] - cy; dz = sx[ii[u] * 3 + 2] - cz; }
                    y = first_layer_quad(fctr, a.fQ != nullptr, y, a.fW != nullptr, dx, dy, dz, fw0, fw1, fw2);
                }
                float4 gm;
                if (POOLED) {
                    const unsigned us = (unsigned)s;
                    gm.x = ((am & 0xffu) == us && fmaf(y.x, cs4.x, ch4.x) > 0.f) ? gp.x : 0.f;
                    gm.y = (((am >> 8) & 0xffu) == us && fmaf(y.y, cs4.y, ch4.y) > 0.f) ? gp.y : 0.f;
                    gm.z = (((am >> 16) & 0xffu) == us && fmaf(y.z, cs4.z, ch4.z) > 0.f) ? gp.z : 0.f;
                    gm.w = ((am >> 24) == us && fmaf(y.w, cs4.w, ch4.w) > 0.f) ? gp.w : 0.f;
                    if (a.dQarg && s < S) {      // one atomic per (group, channel): the sparse part of the pooled dY
                        float *dst = a.dQarg + ((long long)b * n + ii[u]) * C + c0;
                        if (gm.x != 0.f) atomicAdd(dst + 0, cp.x * gm.x);
                        if (gm.y != 0.f) atomicAdd(dst + 1, cp.y * gm.y);
                        if (gm.z != 0.f) atomicAdd(dst + 2, cp.z * gm.z);
                        if (gm.w != 0.f) atomicAdd(dst + 3, cp.w * gm.w);
                    }
                } else {
                    gm = gg[u];
                }
                float d[4];
                d[0] = fmaf(cp.x, gm.x, fmaf(cq.x, y.x, ct.x));
                d[1] = fmaf(cp.y, gm.y, fmaf(cq.y, y.y, ct.y));
                d[2] = fmaf(cp.z, gm.z, fmaf(cq.z, y.z, ct.z));
                d[3] = fmaf(cp.w, gm.w, fmaf(cq.w, y.w, ct.w));
                if (s >= S) d[0] = d[1] = d[2] = d[3] = 0.f;
                dsum.x += d[0]; dsum.y += d[1]; dsum.z += d[2]; dsum.w += d[3];
                const int i = ii[u];
                if (a.xyz) {
                    const float ox = sx[i * 3 + 0] - cx, oy = sx[i * 3 + 1] - cy, oz = sx[i * 3 + 2] - cz;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        aw[0][e] = fmaf(ox, d[e], aw[0][e]);
                        aw[1][e] = fmaf(oy, d[e], aw[1][e]);
                        aw[2][e] = fmaf(oz, d[e], aw[2][e]);
                    }
                }
                if (a.dQ && s < S) {
                    float *dst = acc + i * CS + quad * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, d[e]);
                }
            }
        }
        aw[3][0] += dsum.x; aw[3][1] += dsum.y; aw[3][2] += dsum.z; aw[3][3] += dsum.w;
        if (a.dCtr) {
#pragma unroll
            for (int off = 32; off >= LPR; off >>= 1) {
                dsum.x += __shfl_xor(dsum.x, off, 64); dsum.y += __shfl_xor(dsum.y, off, 64);
                dsum.z += __shfl_xor(dsum.z, off, 64); dsum.w += __shfl_xor(dsum.w, off, 64);
            }
            if (rsub == 0) *reinterpret_cast<float4 *>(a.dCtr + g * C + c0) = dsum;
        }
    }
    __syncthreads();
    if (a.dQ) {
        for (int e = tid; e < n * LPR; e += 1024) {
            const int i = e / LPR, qd = e % LPR;
            *reinterpret_cast<float4 *>(a.dQ + ((long long)b * n + i) * C + sl * CS + qd * 4) =
                *reinterpret_cast<const float4 *>(acc + i * CS + qd * 4);
        }
        __syncthreads();
    }
    if (a.wpart) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = aw[i][e];
#pragma unroll
                for (int off = 32; off >= LPR; off >>= 1) v += __shfl_xor(v, off, 64);
                if (rsub == 0) acc[(wave * 4 + i) * CS + quad * 4 + e] = v;
            }
        __syncthreads();
        for (int e = tid; e < 4 * CS; e += 1024) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += acc[w * 4 * CS + e];
            a.wpart[(((long long)b * a.gsplit + part) * 4 + e / CS) * C + sl * CS + e % CS] = v;
        }
    }
}

template <bool POOLED, int CS>
static int launch_scatter_lds(const ScatterArgs &a, size_t lds, hipStream_t st) {
    // rebuilding Y pays when it costs arithmetic only (coordinate term + bias); a Q / Ctr row gathered per grouped
    // row is slower than streaming the stored Y
    const bool rc = (a.fW || a.fbias) && !a.fQ && !a.fCtr;
    auto kern = rc ? sa_scatter_lds_kernel<POOLED, CS, true> : sa_scatter_lds_kernel<POOLED, CS, false>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    const unsigned grid = (unsigned)((a.b + 7) / 8 * 8 * a.nsl * a.gsplit);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, st, a);
    return pcops_launch_status();
}

// slice width: the widest of 64/32/16/8 that divides C and whose LDS footprint fits; 0 = none (fallback kernel)
static int scatter_lds_slice(int n, int C, bool has_dq, bool has_xyz, size_t *lds) {
    for (int cs = 64; cs >= 8; cs >>= 1) {
        if (C % cs) continue;
        const size_t red = (size_t)16 * 4 * cs;
        size_t fl = has_dq ? (size_t)n * cs : 0;
        if (fl < red) fl = red;
        if (has_xyz) fl += (size_t)n * 3;
        if (fl * sizeof(float) <= 156 * 1024) { *lds = fl * sizeof(float); return cs; }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// backward as a GATHER: the scatter-add turned inside out.  A counting sort per cloud (LDS histogram of idx, scan,
// slot assignment) gives for every source point the list of (group, sample) rows that reference it; then one
// wave per source point sums those rows of dY in registers and writes dQ[b, i, :] once -- no float atomics, no
// memset, every row of G / Y read once with 16-byte loads, dWxyz / dbias partial sums in the same pass.
//   order [b][m*S] int2 : (row-in-cloud j*S + s, idx) sorted by idx      start [b][n+1] int32 : list boundaries
// compacted rows (blocks != NULL): the list covers the cloud's rows 16 bstart[b m] .. 16 bstart[(b+1) m] - 1 and
// order.x is the row relative to the cloud's first row
__global__ __launch_bounds__(1024) void sa_csr_build_kernel(int n, int mS, const int *__restrict__ idx,
                                                            int2 *__restrict__ order, int *__restrict__ start,
                                                            int m, int S, const RowBlock *__restrict__ blocks,
                                                            const int *__restrict__ bstart,
                                                            int2 *__restrict__ sorted = nullptr) {
    extern __shared__ int si[];                 // cnt[n] | cursor[n] | scan scratch [1024]
    int *cnt = si, *cursor = si + n, *sc = si + 2 * n;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int *ib = idx + (long long)b * mS;
    long long r0 = (long long)b * mS;
    int nrows = mS;
    if (blocks) {
        r0 = (long long)bstart[(long long)b * m] * kBlk;
        nrows = (int)((long long)bstart[(long long)(b + 1) * m] * kBlk - r0);
    }
    auto index_of = [&](int e) {                // source point of the cloud's row e
        if (!blocks) return ib[e];
        const long long r = r0 + e;
        const RowBlock rb = blocks[r / kBlk];
        return idx[(long long)rb.g * S + rb.s0 + (int)(r % kBlk)];
    };
    for (int i = tid; i < n; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int e = tid; e < nrows; e += 1024) atomicAdd(&cnt[index_of(e)], 1);
    __syncthreads();
    // exclusive scan: each thread owns a contiguous run of bins
    const int per = (n + 1023) / 1024;
    const int i0 = tid * per, i1 = min(n, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; ++i) local += cnt[i];
    sc[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? sc[tid - off] : 0;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    int run = sc[tid] - local;
    int *sb = start + (long long)b * (n + 1);
    for (int i = i0; i < i1; ++i) {
        cursor[i] = run;
        sb[i] = run;
        run += cnt[i];
    }
    if (tid == 0) sb[n] = nrows;
    __syncthreads();
    int2 *ob = order + r0;
    for (int e = tid; e < nrows; e += 1024) {
        const int i = index_of(e);
        ob[atomicAdd(&cursor[i], 1)] = make_int2(e, i);
    }
    if (!sorted) return;
    // deterministic mode: the slots above were handed out by atomics, i.e. in no particular order.  A rank sort per
    // point (a wave each; the rows of a list are distinct) rewrites every list in ascending row order, so the owner
    // that walks it adds in the same order every run.  O(L^2 / 64) per list -- the price of the switch.
    __syncthreads();
    int2 *ob2 = sorted + r0;
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < n; i += 16) {
        const int L = cnt[i], s0 = cursor[i] - L;
        for (int a0 = 0; a0 < L; a0 += 64) {
            const int ka = a0 + lane < L ? ob[s0 + a0 + lane].x : 0x7fffffff;
            int rank = 0;
            for (int b0 = 0; b0 < L; b0 += 64) {
                const int kb = b0 + lane < L ? ob[s0 + b0 + lane].x : 0x7fffffff;
                const int nb = min(64, L - b0);
                for (int t = 0; t < nb; ++t) rank += __builtin_amdgcn_readlane(kb, t) < ka ? 1 : 0;
            }
            if (a0 + lane < L) ob2[s0 + rank] = make_int2(ka, i);
        }
    }
}

struct CsrArgs {
    int b, n, m, S, C;
    const float *Gm, *Y, *p, *q, *t;
    const float *xyz, *new_xyz;
    const int2 *order;
    float *dQ, *wpart;
    const RowBlock *blocks;      // compacted rows: see sa_csr_build_kernel; the first row of a block carries a weight
    const int *bstart;
    const int *start;            // [b][n+1] list boundaries (owner kernel)
};

// YONLY: the pooled form -- dY = q.Y + t everywhere plus p.gpool at the arg-max rows, which the streaming
// kernel adds with one atomic per (group, channel) (G is never materialised there)
template <int LPR, bool YONLY, int CH = 64>   // lanes per row: C = 4 LPR for LPR < 64; LPR == 64 walks C in blocks of 256
__global__ __launch_bounds__(256) void sa_scatter_csr_kernel(CsrArgs a) {
    // Work is dealt out in CHUNKS of the sorted row list, not per source point: ball query pads short
    // neighbourhoods with their first index and prefers low indices, so a few points own very long lists.  A wave
    // walks its chunk in sorted order, keeps the running sum of the current point in registers and flushes it
    // with ONE global atomic per channel when the point changes -- (points + chunks) x C atomics instead of
    // rows x C, and dQ is zeroed by the launcher.
    constexpr int RW = 64 / LPR, U = (CH / RW) < 4 ? (CH / RW) : 4;   // CH: sorted rows per chunk (16 for small problems)
    __shared__ float red[4][4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rsub = lane / LPR, quad = lane % LPR;
    const int n = a.n, m = a.m, S = a.S, C = a.C, mS = m * S;
    const int nch = (mS + CH - 1) / CH;
    const long long nchunks = (long long)a.b * nch;
    const long long wstride = (long long)gridDim.x * 4;
    for (int cb = 0; cb < C; cb += 4 * LPR) {
        const int c0 = cb + quad * 4;
        const float4 cp = *reinterpret_cast<const float4 *>(a.p + c0);
        const float4 cq = *reinterpret_cast<const float4 *>(a.q + c0);
        const float4 ct = *reinterpret_cast<const float4 *>(a.t + c0);
        float aw[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[i][0] = aw[i][1] = aw[i][2] = aw[i][3] = 0.f;
        for (long long ch = (long long)blockIdx.x * 4 + wave; ch < nchunks; ch += wstride) {
            const int b = (int)(ch / nch);
            long long rowoff = (long long)b * mS;
            int nrows = mS;
            if (a.blocks) {
                rowoff = (long long)a.bstart[(long long)b * m] * kBlk;
                nrows = (int)((long long)a.bstart[(long long)(b + 1) * m] * kBlk - rowoff);
            }
            const int kb = (int)(ch - (long long)b * nch) * CH, ke = min(nrows, kb + CH);
            if (kb >= nrows) continue;                   // wave-uniform
            const int2 *ob = a.order + rowoff;
            int cur = -1;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            auto flush = [&]() {
                float *dst = a.dQ + ((long long)b * n + cur) * C + c0;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[e]);
            };
            // each row-lane set owns a CONTIGUOUS run of CH / RW sorted rows (interleaving would cut every
            // point's list into RW pieces and multiply the flushes)
            const int ks = kb + rsub * (CH / RW), kse = min(ke, ks + CH / RW);
            for (int k0 = 0; k0 < CH / RW; k0 += U) {
                if (kb + k0 >= ke) break;               // wave-uniform: nothing left for any lane set
                int ee[U], ii[U];
                float4 yy[U], gg[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = ks + k0 + u;
                    const int2 oe = ob[k < kse ? k : kb];
                    ee[u] = oe.x;
                    ii[u] = oe.y;
                    const long long r = rowoff + ee[u];
                    yy[u] = *reinterpret_cast<const float4 *>(a.Y + r * C + c0);
                    if (!YONLY) gg[u] = *reinterpret_cast<const float4 *>(a.Gm + r * C + c0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = ks + k0 + u;
                    if (k >= kse) continue;
                    const float4 y = yy[u];
                    const float4 gm = YONLY ? make_float4(0.f, 0.f, 0.f, 0.f) : gg[u];
                    float d[4];
                    float wt = 1.f;
                    int j = (int)((unsigned)ee[u] / (unsigned)S);          // group inside the cloud
                    if (a.blocks) {
                        const long long r = rowoff + ee[u];
                        const RowBlock rb = a.blocks[r / kBlk];
                        wt = (r % kBlk) == 0 ? rb.w : 1.f;
                        j = rb.g - b * m;
                    }
                    d[0] = fmaf(cp.x, gm.x, wt * fmaf(cq.x, y.x, ct.x));
                    d[1] = fmaf(cp.y, gm.y, wt * fmaf(cq.y, y.y, ct.y));
                    d[2] = fmaf(cp.z, gm.z, wt * fmaf(cq.z, y.z, ct.z));
                    d[3] = fmaf(cp.w, gm.w, wt * fmaf(cq.w, y.w, ct.w));
                    const int i = ii[u];
                    if (i != cur) {
                        if (cur >= 0) flush();
                        cur = i;
                        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += d[e];
                    if (a.wpart) {
                        if (a.xyz) {
                            const float *px = a.xyz + ((long long)b * n + i) * 3;
                            const float *cx = a.new_xyz + ((long long)b * m + j) * 3;
                            const float ox = px[0] - cx[0], oy = px[1] - cx[1], oz = px[2] - cx[2];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                aw[0][e] = fmaf(ox, d[e], aw[0][e]);
                                aw[1][e] = fmaf(oy, d[e], aw[1][e]);
                                aw[2][e] = fmaf(oz, d[e], aw[2][e]);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) aw[3][e] += d[e];
                    }
                }
            }
            if (cur >= 0) flush();
        }
        if (a.wpart) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = aw[i][e];
#pragma unroll
                    for (int off = 32; off >= LPR; off >>= 1) v += __shfl_xor(v, off, 64);
                    if (rsub == 0) red[wave][i][quad * 4 + e] = v;
                }
            __syncthreads();
            for (int e = tid; e < 4 * 4 * LPR; e += 256) {
                const int i = e / (4 * LPR), c = e % (4 * LPR);
                a.wpart[((long long)blockIdx.x * 4 + i) * C + cb + c] =
                    red[0][i][c] + red[1][i][c] + red[2][i][c] + red[3][i][c];
            }
            __syncthreads();
        }
    }
}

// OWNER form of the same pass: a lane set per SOURCE POINT walks the point's list start[i] .. start[i+1] and writes
// dQ[b, i, :] once with a plain store -- no atomic, no memset, and with the lists in ascending row order (sorted build)
// the sum is taken in the same order every run.  Long lists (ball query favours low indices) unbalance the waves a
// little; the atomics of the chunked form cost more than that.
template <int LPR>
__global__ __launch_bounds__(256) void sa_scatter_owner_kernel(CsrArgs a) {
    constexpr int PW = 64 / LPR, U = 4;          // points per wave, list entries in flight per point
    __shared__ float red[4][4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int psub = lane / LPR, quad = lane % LPR;
    const int n = a.n, m = a.m, S = a.S, C = a.C, mS = m * S;
    const long long npts = (long long)a.b * n;
    const long long pstride = (long long)gridDim.x * 4 * PW;
    for (int cb = 0; cb < C; cb += 4 * LPR) {
        const int c0 = cb + quad * 4;
        const float4 cp = *reinterpret_cast<const float4 *>(a.p + c0);
        const float4 cq = *reinterpret_cast<const float4 *>(a.q + c0);
        const float4 ct = *reinterpret_cast<const float4 *>(a.t + c0);
        float aw[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[i][0] = aw[i][1] = aw[i][2] = aw[i][3] = 0.f;
        // the longest lists belong to the lowest indices of every cloud (ball query pads with a group's first index):
        // with a plain stride those points, n apart, would all land on the same few waves -- each pass rotates the
        // wave -> point map by an odd step instead
        const long long nwv = (long long)gridDim.x * 4;
        const long long wv = (long long)blockIdx.x * 4 + wave;
        for (long long pass = 0; pass * pstride < npts; ++pass) {
            const long long pt0 = pass * pstride + ((wv + pass * 1031) % nwv) * PW;
            const long long pt = pt0 + psub;
            const bool pin = pt < npts;
            const int b = (int)((pin ? pt : 0) / n), i = (int)((pin ? pt : 0) - (long long)b * n);
            long long rowoff = (long long)b * mS;
            if (a.blocks) rowoff = (long long)a.bstart[(long long)b * m] * kBlk;
            const int *sb = a.start + (long long)b * (n + 1);
            const int k0 = pin ? sb[i] : 0, k1 = pin ? sb[i + 1] : 0;
            const int2 *ob = a.order + rowoff;
            float px = 0.f, py = 0.f, pz = 0.f;
            if (a.wpart && a.xyz) {
                const float *pp = a.xyz + ((long long)b * n + i) * 3;
                px = pp[0]; py = pp[1]; pz = pp[2];
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = k0; k < k1; k += U) {
                float4 yy[U], gg[U];
                float wt[U];
                int jj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u < k1 ? k + u : k0;
                    const int e = ob[kk].x;
                    const long long r = rowoff + e;
                    yy[u] = *reinterpret_cast<const float4 *>(a.Y + r * C + c0);
                    gg[u] = *reinterpret_cast<const float4 *>(a.Gm + r * C + c0);
                    wt[u] = 1.f;
                    jj[u] = (int)((unsigned)e / (unsigned)S);
                    if (a.blocks) {
                        const RowBlock rb = a.blocks[r / kBlk];
                        wt[u] = (r % kBlk) == 0 ? rb.w : 1.f;
                        jj[u] = rb.g - b * m;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u >= k1) continue;
                    const float4 y = yy[u], gm = gg[u];
                    float d[4];
                    d[0] = fmaf(cp.x, gm.x, wt[u] * fmaf(cq.x, y.x, ct.x));
                    d[1] = fmaf(cp.y, gm.y, wt[u] * fmaf(cq.y, y.y, ct.y));
                    d[2] = fmaf(cp.z, gm.z, wt[u] * fmaf(cq.z, y.z, ct.z));
                    d[3] = fmaf(cp.w, gm.w, wt[u] * fmaf(cq.w, y.w, ct.w));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += d[e];
                    if (a.wpart) {
                        if (a.xyz) {
                            const float *cx = a.new_xyz + ((long long)b * m + jj[u]) * 3;
                            const float ox = px - cx[0], oy = py - cx[1], oz = pz - cx[2];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                aw[0][e] = fmaf(ox, d[e], aw[0][e]);
                                aw[1][e] = fmaf(oy, d[e], aw[1][e]);
                                aw[2][e] = fmaf(oz, d[e], aw[2][e]);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) aw[3][e] += d[e];
                    }
                }
            }
            if (pin) *reinterpret_cast<float4 *>(a.dQ + pt * C + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        if (a.wpart) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = aw[i][e];
#pragma unroll
                    for (int off = 32; off >= LPR; off >>= 1) v += __shfl_xor(v, off, 64);
                    if (psub == 0) red[wave][i][quad * 4 + e] = v;
                }
            __syncthreads();
            for (int e = tid; e < 4 * 4 * LPR; e += 256) {
                const int i = e / (4 * LPR), c = e % (4 * LPR);
                a.wpart[((long long)blockIdx.x * 4 + i) * C + cb + c] =
                    red[0][i][c] + red[1][i][c] + red[2][i][c] + red[3][i][c];
            }
            __syncthreads();
        }
    }
}

// ---- first layer of a stack whose groups are whole clouds in their own order (identity index, round 5) ---------------------
//   Y[r, :] = Q[r, :] + Ctr[r / rpg, :]   (+ shifted moments),      backward:  dQ = p G + q Y + t,  dCtr[g] = sum_{r in g} dQ[r]
// The general gather kernels deal ONE workgroup per group -- 128 workgroups for dgcnn_bga's segmentation head (2048 x 512 per
// cloud), 659 us for a 1 GB stream; its backward was two elementwise torch launches plus a column-sum pass over the same rows.
constexpr int kCloudRows = 256;         // rows per workgroup (a whole number of them per group)
__global__ __launch_bounds__(256) void cloud_bias_fwd_kernel(long long rows, int rpg, int C, const float *__restrict__ Q,
                                                             const float *__restrict__ Ctr, float *__restrict__ Y,
                                                             float *__restrict__ stats, const float *__restrict__ pivot, int nt) {
    extern __shared__ float cb_red[];               // [RL][2][C]
    const int c4n = C / 4, RL = 256 / c4n;
    const int cq = (threadIdx.x % c4n) * 4, rl = threadIdx.x / c4n;
    const long long r0 = (long long)blockIdx.x * kCloudRows;
    const long long g = r0 / rpg;
    const float4 ct = *reinterpret_cast<const float4 *>(Ctr + g * C + cq);
    const float4 pv = pivot ? *reinterpret_cast<const float4 *>(pivot + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    typedef float cb_f4 __attribute__((ext_vector_type(4)));
    for (int i = rl; i < kCloudRows; i += 4 * RL) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = r0 + i + u * RL;
            q[u] = (i + u * RL < kCloudRows && r < rows) ? *reinterpret_cast<const float4 *>(Q + r * C + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = r0 + i + u * RL;
            if (!(i + u * RL < kCloudRows && r < rows)) continue;
            const float4 y = make_float4(q[u].x + ct.x, q[u].y + ct.y, q[u].z + ct.z, q[u].w + ct.w);
            if (nt) __builtin_nontemporal_store(cb_f4{y.x, y.y, y.z, y.w}, reinterpret_cast<cb_f4 *>(Y + r * C + cq));
            else *reinterpret_cast<float4 *>(Y + r * C + cq) = y;
            const float d[4] = {y.x - pv.x, y.y - pv.y, y.z - pv.z, y.w - pv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += d[e]; s2[e] = fmaf(d[e], d[e], s2[e]); }
        }
    }
    if (!stats) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cb_red[(rl * 2 + 0) * C + cq + e] = s1[e];
        cb_red[(rl * 2 + 1) * C + cq + e] = s2[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        float v = 0.f;
        for (int r = 0; r < RL; ++r) v += cb_red[r * 2 * C + e];
        stats[(long long)blockIdx.x * 2 * C + e] = v;
    }
}

__global__ __launch_bounds__(256) void cloud_bias_bwd_kernel(long long rows, int C, const float *__restrict__ G,
                                                             const float *__restrict__ Y, const float *__restrict__ p,
                                                             const float *__restrict__ q, const float *__restrict__ t,
                                                             float *__restrict__ dQ, float *__restrict__ part) {
    extern __shared__ float cb_red[];               // [RL][C]
    const int c4n = C / 4, RL = 256 / c4n;
    const int cq = (threadIdx.x % c4n) * 4, rl = threadIdx.x / c4n;
    const long long r0 = (long long)blockIdx.x * kCloudRows;
    const float4 cp = *reinterpret_cast<const float4 *>(p + cq), cqv = *reinterpret_cast<const float4 *>(q + cq),
                 ctv = *reinterpret_cast<const float4 *>(t + cq);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = rl; i < kCloudRows; i += 4 * RL) {
        float4 g[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = r0 + i + u * RL;
            const bool in = i + u * RL < kCloudRows && r < rows;
            g[u] = in ? *reinterpret_cast<const float4 *>(G + r * C + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
            y[u] = in ? *reinterpret_cast<const float4 *>(Y + r * C + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = r0 + i + u * RL;
            if (!(i + u * RL < kCloudRows && r < rows)) continue;
            const float4 d = make_float4(fmaf(cp.x, g[u].x, fmaf(cqv.x, y[u].x, ctv.x)), fmaf(cp.y, g[u].y, fmaf(cqv.y, y[u].y, ctv.y)),
                                         fmaf(cp.z, g[u].z, fmaf(cqv.z, y[u].z, ctv.z)), fmaf(cp.w, g[u].w, fmaf(cqv.w, y[u].w, ctv.w)));
            if (dQ) *reinterpret_cast<float4 *>(dQ + r * C + cq) = d;
            s[0] += d.x; s[1] += d.y; s[2] += d.z; s[3] += d.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) cb_red[rl * C + cq + e] = s[e];
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += 256) {
        float v = 0.f;
        for (int r = 0; r < RL; ++r) v += cb_red[r * C + e];
        part[(long long)blockIdx.x * C + e] = v;
    }
}

// dCtr[g][c] = sum of the group's partial rows (fixed order)
__global__ __launch_bounds__(256) void cloud_bias_reduce_kernel(int ppg, int C, const float *__restrict__ part, float *__restrict__ dCtr) {
    const int g = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = 0.f;
        for (int r = 0; r < ppg; ++r) v += part[((long long)g * ppg + r) * C + c];
        dCtr[(long long)g * C + c] = v;
    }
}

constexpr int kCsrGrid = 1024;      // persistent workgroups of the gather pass (= rows of its wpart)

// gradients of an ARITHMETIC first layer from a handful of sums (see pcops_mlp_gemm_dgrad_xyz in pcops.h):
//   dWxyz[i][c] = p[c] A[i][c] + q[c] B[i][c] + t[c] S[i],   B = M33 Wxyz + S^T b,   dbias[c] = p sumG + q sumY + t rows
// block = 32 channels x 32 row lanes; all sums in double, fixed order
__global__ __launch_bounds__(1024) void xyz_first_layer_grads_kernel(int P1, const float *__restrict__ xstats, int P2,
                                                                     const float *__restrict__ moments, int C,
                                                                     const float *__restrict__ Wxyz,
                                                                     const float *__restrict__ bias,
                                                                     const float *__restrict__ p,
                                                                     const float *__restrict__ q,
                                                                     const float *__restrict__ t,
                                                                     const float *__restrict__ sumG,
                                                                     const float *__restrict__ mean, double rows,
                                                                     float *__restrict__ dWxyz,
                                                                     float *__restrict__ dbias) {
    __shared__ double smA[3][32][32];
    __shared__ double smM[9][112];
    __shared__ double mom[9];
    const int tid = threadIdx.x, cl = tid & 31, g = tid >> 5;
    const int c = blockIdx.x * 32 + cl;
    // offset moments: 9 sums over P2 partial rows; thread = (row lane l < 112, moment k) reads element r*9 + k, so a
    // wave reads consecutive floats
    if (tid < 9 * 112) {
        const int l = tid / 9, k = tid % 9;
        // eight independent loads in flight per thread (a one-at-a-time loop over the 16 384 partial rows of the SSG
        // config made this tiny kernel take 130 us, all of it load latency); the summation order stays fixed
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int r = l;
        for (; r + 7 * 112 < P2; r += 8 * 112) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = moments[(long long)(r + u * 112) * 9 + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += (double)v[u];
        }
        for (; r < P2; r += 112) a[0] += (double)moments[(long long)r * 9 + k];
        smM[k][l] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (c < C)
        for (int r = g; r < P1; r += 32) {
            a0 += (double)xstats[((long long)r * 3 + 0) * C + c];
            a1 += (double)xstats[((long long)r * 3 + 1) * C + c];
            a2 += (double)xstats[((long long)r * 3 + 2) * C + c];
        }
    smA[0][g][cl] = a0; smA[1][g][cl] = a1; smA[2][g][cl] = a2;
    __syncthreads();
    if (tid < 9) {
        double a = 0.0;
        for (int l = 0; l < 112; ++l) a += smM[tid][l];
        mom[tid] = a;
    }
    __syncthreads();
    if (g != 0 || c >= C) return;
    double A[3] = {0.0, 0.0, 0.0};
    for (int l = 0; l < 32; ++l) { A[0] += smA[0][l][cl]; A[1] += smA[1][l][cl]; A[2] += smA[2][l][cl]; }
    const double M33[3][3] = {{mom[0], mom[1], mom[2]}, {mom[1], mom[3], mom[4]}, {mom[2], mom[4], mom[5]}};
    const double S[3] = {mom[6], mom[7], mom[8]};
    const double w[3] = {(double)Wxyz[0 * C + c], (double)Wxyz[1 * C + c], (double)Wxyz[2 * C + c]};
    const double b = bias ? (double)bias[c] : 0.0;
    const double pc = p[c], qc = q[c], tc = t[c];
    for (int i = 0; i < 3; ++i) {
        const double B = M33[i][0] * w[0] + M33[i][1] * w[1] + M33[i][2] * w[2] + S[i] * b;
        dWxyz[i * C + c] = (float)(pc * A[i] + qc * B + tc * S[i]);
    }
    if (dbias) dbias[c] = (float)(pc * (double)sumG[c] + qc * ((double)mean[c] * rows) + tc * rows);
}

// ---------------------------------------------------------------------------------------------------------
// EdgeConv as a single pooled layer, WITHOUT the (b, n, k, c) tensor in either direction.
//   y[g, s, :] = Q[idx[g, s], :] + Ctr[g, :]   ->  BN  ->  ReLU  ->  max over s                (dgcnn.py:39-48)
// Ctr is constant inside a group and BN + ReLU is monotone per channel (increasing for gamma >= 0), so
//   * the pooled value only needs  qsel = max_s (or min_s) Q[idx[g, s]]  and the first s attaining it,
//   * the batch statistics only need  SQ = sum_s Q[idx]  and  sum_s Q[idx]^2:
//       sum y = SQ + k Ctr,    sum y^2 = SQ2 + 2 Ctr SQ + k Ctr^2,
//   * the backward needs no y either:  dY = q y + t (+ p gpool at the arg row), so
//       dCtr[g] = q (SQ + k Ctr) + k t + a[g],   a = p gpool [relu(bn(ysel)) > 0]
//       dQ[i]   = cnt_i (q Q[i] + t) + q sum_{(g, s) -> i} Ctr[g] + sum_{g: arg row -> i} a[g]
//     where the middle sum walks the inverse index of idx (csr build above) over the L2-resident Ctr rows.
// thread = (group lane, column quad): one thread reduces the k neighbours of one (group, quad)
__global__ __launch_bounds__(256) void edge_pool_fwd_kernel(long long G, int n, int m, int S, int C,
                                                            const float *__restrict__ Q,
                                                            const float *__restrict__ Ctr,
                                                            const int *__restrict__ idx,
                                                            const float *__restrict__ gamma,
                                                            float *__restrict__ SQ, float *__restrict__ qsel,
                                                            unsigned char *__restrict__ arg,
                                                            float *__restrict__ stats,
                                                            const float *__restrict__ pivot, int groups_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [GL][2][C] statistics scratch
    const int c4n = C / 4;
    const int GL = 256 / c4n;                                    // groups in flight per block
    const int cq = (threadIdx.x % c4n) * 4, gl = threadIdx.x / c4n;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + cq);
    const bool up[4] = {!(ga.x < 0.f), !(ga.y < 0.f), !(ga.z < 0.f), !(ga.w < 0.f)};
    // shifted moments (pcops.h): the statistics are those of y - pivot, and the neighbour rows enter relative to ONE
    // sample row of Q (row 0 of cloud 0 -- any row is within a few standard deviations of the channel's mean), so the
    // three terms of  sum (q' + c')^2 = SQ2' + 2 c' SQ' + k c'^2  are all O(variance) and nothing cancels
    const float4 q0 = stats ? *reinterpret_cast<const float4 *>(Q + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pv4 = (stats && pivot) ? *reinterpret_cast<const float4 *>(pivot + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float qz[4] = {q0.x, q0.y, q0.z, q0.w};
    const float cz[4] = {q0.x - pv4.x, q0.y - pv4.y, q0.z - pv4.z, q0.w - pv4.w};     // c' = Ctr + cz
    if (gl < GL) {
        for (long long g = g0 + gl; g < g1; g += GL) {
            const long long b = g / m;
            const float4 ct = *reinterpret_cast<const float4 *>(Ctr + g * C + cq);
            float sq[4] = {0.f, 0.f, 0.f, 0.f}, sq2[4] = {0.f, 0.f, 0.f, 0.f};
            float ex[4];
            int ea[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) ex[e] = up[e] ? -INFINITY : INFINITY;
            // the k gathered rows are independent loads: four in flight per thread (one at a time left the kernel at
            // 0.15 of the HBM rate on L2-resident rows -- latency, not bandwidth)
            auto take = [&](const float4 &q, int s) {
                const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qd = qv[e] - qz[e];
                    sq[e] += qd;
                    sq2[e] = fmaf(qd, qd, sq2[e]);
                    const bool better = up[e] ? qv[e] > ex[e] : qv[e] < ex[e];    // strict: first extremum wins
                    ex[e] = better ? qv[e] : ex[e];
                    ea[e] = better ? s : ea[e];
                }
            };
            const int *ig = idx + g * S;
            const float *Qb = Q + b * n * (long long)C + cq;
            int s = 0;
            constexpr int UF = 10;                                // k = 20: two rounds
            for (; s + UF <= S; s += UF) {
                int iu[UF];
                float4 qu[UF];
#pragma unroll
                for (int u = 0; u < UF; ++u) iu[u] = ig[s + u];
#pragma unroll
                for (int u = 0; u < UF; ++u) qu[u] = *reinterpret_cast<const float4 *>(Qb + iu[u] * (long long)C);
#pragma unroll
                for (int u = 0; u < UF; ++u) take(qu[u], s + u);
            }
            for (; s + 4 <= S; s += 4) {
                const int i0 = ig[s], i1 = ig[s + 1], i2 = ig[s + 2], i3 = ig[s + 3];
                const float4 q0v = *reinterpret_cast<const float4 *>(Qb + i0 * (long long)C);
                const float4 q1v = *reinterpret_cast<const float4 *>(Qb + i1 * (long long)C);
                const float4 q2v = *reinterpret_cast<const float4 *>(Qb + i2 * (long long)C);
                const float4 q3v = *reinterpret_cast<const float4 *>(Qb + i3 * (long long)C);
                take(q0v, s); take(q1v, s + 1); take(q2v, s + 2); take(q3v, s + 3);
            }
            for (; s < S; ++s) take(*reinterpret_cast<const float4 *>(Qb + ig[s] * (long long)C), s);
            const float kf = (float)S;
            *reinterpret_cast<float4 *>(SQ + g * C + cq) =                       // the backward reads SQ = sum_s q
                make_float4(fmaf(kf, qz[0], sq[0]), fmaf(kf, qz[1], sq[1]), fmaf(kf, qz[2], sq[2]), fmaf(kf, qz[3], sq[3]));
            *reinterpret_cast<float4 *>(qsel + g * C + cq) = make_float4(ex[0], ex[1], ex[2], ex[3]);
            uchar4 a4;
            a4.x = (unsigned char)ea[0]; a4.y = (unsigned char)ea[1]; a4.z = (unsigned char)ea[2]; a4.w = (unsigned char)ea[3];
            *reinterpret_cast<uchar4 *>(arg + g * C + cq) = a4;
            const float cv[4] = {ct.x + cz[0], ct.y + cz[1], ct.z + cz[2], ct.w + cz[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += fmaf(kf, cv[e], sq[e]);
                s2[e] += fmaf(cv[e], fmaf(kf, cv[e], 2.f * sq[e]), sq2[e]);
            }
        }
    }
    if (stats == nullptr) return;
    if (gl < GL)
        for (int e = 0; e < 4; ++e) {
            sm[(gl * 2 + 0) * C + cq + e] = s1[e];
            sm[(gl * 2 + 1) * C + cq + e] = s2[e];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        float t = 0.f;
        for (int l = 0; l < GL; ++l) t += sm[(l * 2 + which) * C + c];
        stats[((long long)blockIdx.x * 2 + which) * C + c] = t;
    }
}

// out = relu(scale (qsel + Ctr) + shift), ysel = qsel + Ctr   (the same sum the forward of a stored y would hold)
__global__ __launch_bounds__(256) void edge_pool_out_kernel(long long total, int C, const float *__restrict__ qsel,
                                                            const float *__restrict__ Ctr,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            float *__restrict__ out, float *__restrict__ ysel) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float y = qsel[e] + Ctr[e];
        out[e] = fmaxf(fmaf(y, scale[c], shift[c]), 0.f);
        if (ysel) ysel[e] = y;
    }
}

// dCtr and the arg-row part of dQ (one atomic per (group, channel))
__global__ __launch_bounds__(256) void edge_pool_bwd_ctr_kernel(long long total, int n, int m, int S, int C,
                                                                const float *__restrict__ gpool,
                                                                const float *__restrict__ ysel,
                                                                const float *__restrict__ SQ,
                                                                const float *__restrict__ Ctr,
                                                                const unsigned char *__restrict__ arg,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift,
                                                                const float *__restrict__ p,
                                                                const float *__restrict__ q,
                                                                const float *__restrict__ t,
                                                                float *__restrict__ dCtr, float *__restrict__ dQ) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long g = e / C;
        const int c = (int)(e - g * C);
        const float a = fmaf(ysel[e], scale[c], shift[c]) > 0.f ? p[c] * gpool[e] : 0.f;
        const float kf = (float)S;
        dCtr[e] = fmaf(q[c], fmaf(kf, Ctr[e], SQ[e]), fmaf(kf, t[c], a));
        if (a != 0.f) {
            const long long b = g / m;
            atomicAdd(dQ + (b * n + idx[g * S + arg[e]]) * (long long)C + c, a);
        }
    }
}

// dense part of dQ over the inverse index: waves walk chunks of the sorted row list, sum the Ctr rows of the current
// point in registers and flush  q (cnt Q[i] + sum Ctr) + cnt t  with one atomic per channel when the point changes
template <int LPR>
__global__ __launch_bounds__(256) void edge_pool_bwd_q_kernel(int B, int n, int m, int S, int C,
                                                              const float *__restrict__ Q,
                                                              const float *__restrict__ Ctr,
                                                              const float *__restrict__ qv,
                                                              const float *__restrict__ tv,
                                                              const int2 *__restrict__ order,
                                                              float *__restrict__ dQ) {
    constexpr int RW = 64 / LPR, U = 8, CH = 64;       // U rows in flight per lane set (latency bound: L2 gathers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rsub = lane / LPR, quad = lane % LPR;
    const int mS = m * S;
    const int nch = (mS + CH - 1) / CH;
    const long long nchunks = (long long)B * nch;
    const long long wstride = (long long)gridDim.x * 4;
    for (int cb = 0; cb < C; cb += 4 * LPR) {
        const int c0 = cb + quad * 4;
        const float4 cq = *reinterpret_cast<const float4 *>(qv + c0);
        const float4 ct = *reinterpret_cast<const float4 *>(tv + c0);
        for (long long ch = (long long)blockIdx.x * 4 + wave; ch < nchunks; ch += wstride) {
            const int b = (int)(ch / nch);
            const int kb = (int)(ch - (long long)b * nch) * CH, ke = min(mS, kb + CH);
            const int2 *ob = order + (long long)b * mS;
            int cur = -1, cnt = 0;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            auto flush = [&]() {
                const float4 qi = *reinterpret_cast<const float4 *>(Q + ((long long)b * n + cur) * C + c0);
                const float kf = (float)cnt;
                float *dst = dQ + ((long long)b * n + cur) * C + c0;
                atomicAdd(dst + 0, fmaf(cq.x, fmaf(kf, qi.x, acc[0]), kf * ct.x));
                atomicAdd(dst + 1, fmaf(cq.y, fmaf(kf, qi.y, acc[1]), kf * ct.y));
                atomicAdd(dst + 2, fmaf(cq.z, fmaf(kf, qi.z, acc[2]), kf * ct.z));
                atomicAdd(dst + 3, fmaf(cq.w, fmaf(kf, qi.w, acc[3]), kf * ct.w));
            };
            const int ks = kb + rsub * (CH / RW), kse = min(ke, ks + CH / RW);
            for (int k0 = 0; k0 < CH / RW; k0 += U) {
                if (kb + k0 >= ke) break;
                int ii[U];
                float4 cc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = ks + k0 + u;
                    const int2 oe = ob[k < kse ? k : kb];
                    ii[u] = oe.y;
                    const int j = (int)((unsigned)oe.x / (unsigned)S);
                    cc[u] = *reinterpret_cast<const float4 *>(Ctr + ((long long)b * m + j) * C + c0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = ks + k0 + u;
                    if (k >= kse) continue;
                    if (ii[u] != cur) {
                        if (cur >= 0) flush();
                        cur = ii[u];
                        cnt = 0;
                        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
                    }
                    acc[0] += cc[u].x; acc[1] += cc[u].y; acc[2] += cc[u].z; acc[3] += cc[u].w;
                    ++cnt;
                }
            }
            if (cur >= 0) flush();
        }
    }
}

// ---- EdgeConv backward, second formulation (no global atomics) -----------------------------------------------------
// dQ[b,i,:] = sum over the rows (j,s) with idx[j,s] = i of dY[j,s,:],  dY = q (Q[i] + Ctr[j]) + t  everywhere, plus
// a[j,c] = p[c] relu'(.) gpool[j,c] at the ONE row s = arg[j,c] of every (group, channel).
//   edge_pool_bwd_sparse_kernel: the a-term.  One workgroup = (cloud, slice of 16 channels); the slice of dQ
//     (n x 16 floats, 128 KB at n = 2048) lives in LDS, the (group, channel) values are added there with LDS atomics
//     -- G x C of them, 20x fewer than the dense term, so their modest rate does not matter -- and the slice leaves
//     with plain coalesced stores: dQ needs no memset and no global atomic.  dCtr is written on the way.
//   edge_pool_bwd_dense_kernel: the dense term.  OWNER computes: a lane set per source point walks the point's list of
//     the inverse index (start[i] .. start[i+1]) and adds  q (cnt Q[i] + sum Ctr[j]) + cnt t  to dQ[b,i,:] with a
//     plain read-modify-write (it is the only writer of that row after the sparse kernel finished).
constexpr int kEdgeSlice = 16;

template <bool DET>     // DET: dCtr only -- the arg-row term is added by the owner walk (edge_pool_bwd_dense_kernel<., true>)
__global__ __launch_bounds__(1024) void edge_pool_bwd_sparse_kernel(int n, int m, int S, int C,
                                                                    const float *__restrict__ gpool,
                                                                    const float *__restrict__ ysel,
                                                                    const float *__restrict__ SQ,
                                                                    const float *__restrict__ Ctr,
                                                                    const unsigned char *__restrict__ arg,
                                                                    const int *__restrict__ idx,
                                                                    const float *__restrict__ scale,
                                                                    const float *__restrict__ shift,
                                                                    const float *__restrict__ p,
                                                                    const float *__restrict__ q,
                                                                    const float *__restrict__ t,
                                                                    float *__restrict__ dCtr, float *__restrict__ dQ) {
    extern __shared__ __attribute__((aligned(16))) float acc[];          // [n][kEdgeSlice]
    constexpr int NT = 1024, GL = NT / kEdgeSlice, U = 4;                // 64 groups per pass, 4 passes in flight
    const int nsl = C / kEdgeSlice;
    const int b = blockIdx.x / nsl, c0 = (blockIdx.x % nsl) * kEdgeSlice;
    const int tid = threadIdx.x, cl = tid % kEdgeSlice, gl = tid / kEdgeSlice;
    if (!DET) {
        for (int e = tid; e < n * kEdgeSlice; e += NT) acc[e] = 0.f;
        __syncthreads();
    }
    const int c = c0 + cl;
    const float sc = scale[c], sh = shift[c], pc = p[c], qc = q[c], tc = t[c];
    const float kf = (float)S;
    for (int j0 = gl; j0 < m; j0 += GL * U) {
        // all loads of U groups are requested before the first is used (this loop is nothing but load latency)
        float ys[U], gp[U], ce[U], sq[U];
        int ar[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * GL < m ? j0 + u * GL : j0;
            const long long e = ((long long)b * m + j) * C + c;
            ys[u] = ysel[e]; gp[u] = gpool[e]; ce[u] = Ctr[e]; sq[u] = SQ[e]; ar[u] = arg[e];
        }
        int di[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * GL < m ? j0 + u * GL : j0;
            di[u] = idx[((long long)b * m + j) * S + ar[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * GL;
            if (j >= m) continue;
            const long long e = ((long long)b * m + j) * C + c;
            const float a = fmaf(ys[u], sc, sh) > 0.f ? pc * gp[u] : 0.f;
            dCtr[e] = fmaf(qc, fmaf(kf, ce[u], sq[u]), fmaf(kf, tc, a));
            if (!DET && a != 0.f) atomicAdd(&acc[di[u] * kEdgeSlice + cl], a);
        }
    }
    if (DET) return;
    __syncthreads();
    float *dst = dQ + (long long)b * n * C + c0;
    for (int e = tid; e < n * (kEdgeSlice / 4); e += NT) {
        const int i = e / (kEdgeSlice / 4), quad = (e % (kEdgeSlice / 4)) * 4;
        *reinterpret_cast<float4 *>(dst + (long long)i * C + quad) = *reinterpret_cast<const float4 *>(&acc[i * kEdgeSlice + quad]);
    }
}

struct EdgeSparse {     // what the arg-row term needs (deterministic mode only)
    const float *gpool, *ysel, *scale, *shift, *p;
    const unsigned char *arg;
};

// DET: the lists are in ascending row order and the owner also adds the arg-row term p.g of every (group, channel)
// whose arg-max is this list entry -- more bytes per entry (gpool, ysel, arg next to Ctr), but every sum has one owner
// and a fixed order; dQ is written, not updated.
template <int LPR, bool DET>     // lanes per point: C = 4 LPR for LPR < 64; LPR == 64 walks C in blocks of 256
__global__ __launch_bounds__(256) void edge_pool_bwd_dense_kernel(int B, int n, int m, int S, int C,
                                                                  const float *__restrict__ Q,
                                                                  const float *__restrict__ Ctr,
                                                                  const float *__restrict__ qv,
                                                                  const float *__restrict__ tv,
                                                                  const int2 *__restrict__ order,
                                                                  const int *__restrict__ start,
                                                                  float *__restrict__ dQ, EdgeSparse sp) {
    constexpr int PW = 64 / LPR, U = 4;          // points per wave, list entries in flight per point
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int psub = lane / LPR, quad = lane % LPR;
    const long long npts = (long long)B * n;
    const long long pstride = (long long)gridDim.x * 4 * PW;
    const int mS = m * S;
    for (int cb = 0; cb < C; cb += 4 * LPR) {
        const int c0 = cb + quad * 4;
        const float4 cq = *reinterpret_cast<const float4 *>(qv + c0);
        const float4 ct = *reinterpret_cast<const float4 *>(tv + c0);
        float4 psc, psh, pp;
        if (DET) {
            psc = *reinterpret_cast<const float4 *>(sp.scale + c0);
            psh = *reinterpret_cast<const float4 *>(sp.shift + c0);
            pp = *reinterpret_cast<const float4 *>(sp.p + c0);
        }
        for (long long pt0 = ((long long)blockIdx.x * 4 + wave) * PW; pt0 < npts; pt0 += pstride) {
            const long long pt = pt0 + psub;
            const bool pin = pt < npts;
            const int b = (int)((pin ? pt : 0) / n), i = (int)((pin ? pt : 0) - (long long)b * n);
            const int *sb = start + (long long)b * (n + 1);
            const int k0 = pin ? sb[i] : 0, k1 = pin ? sb[i + 1] : 0;
            const int2 *ob = order + (long long)b * mS;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            float asp[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = k0; k < k1; k += U) {
                float4 cc[U], gg[DET ? U : 1], ys[DET ? U : 1];
                unsigned am[DET ? U : 1], sr[DET ? U : 1];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u < k1 ? k + u : k0;
                    const unsigned e = (unsigned)ob[kk].x;
                    const int j = (int)(e / (unsigned)S);
                    const long long ge = ((long long)b * m + j) * C + c0;
                    cc[u] = *reinterpret_cast<const float4 *>(Ctr + ge);
                    if (DET) {
                        gg[u] = *reinterpret_cast<const float4 *>(sp.gpool + ge);
                        ys[u] = *reinterpret_cast<const float4 *>(sp.ysel + ge);
                        am[u] = *reinterpret_cast<const unsigned *>(sp.arg + ge);
                        sr[u] = e - (unsigned)j * (unsigned)S;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (k + u < k1) {
                        acc[0] += cc[u].x; acc[1] += cc[u].y; acc[2] += cc[u].z; acc[3] += cc[u].w;
                        if (DET) {
                            const unsigned a4 = am[u], s = sr[u];
                            if ((a4 & 0xffu) == s && fmaf(ys[u].x, psc.x, psh.x) > 0.f) asp[0] += pp.x * gg[u].x;
                            if (((a4 >> 8) & 0xffu) == s && fmaf(ys[u].y, psc.y, psh.y) > 0.f) asp[1] += pp.y * gg[u].y;
                            if (((a4 >> 16) & 0xffu) == s && fmaf(ys[u].z, psc.z, psh.z) > 0.f) asp[2] += pp.z * gg[u].z;
                            if ((a4 >> 24) == s && fmaf(ys[u].w, psc.w, psh.w) > 0.f) asp[3] += pp.w * gg[u].w;
                        }
                    }
            }
            if (pin && (DET || k1 > k0)) {
                const float kf = (float)(k1 - k0);
                const float4 qi = *reinterpret_cast<const float4 *>(Q + pt * C + c0);
                float4 *dst = reinterpret_cast<float4 *>(dQ + pt * C + c0);
                float4 d = DET ? make_float4(asp[0], asp[1], asp[2], asp[3]) : *dst;
                d.x += fmaf(cq.x, fmaf(kf, qi.x, acc[0]), kf * ct.x);
                d.y += fmaf(cq.y, fmaf(kf, qi.y, acc[1]), kf * ct.y);
                d.z += fmaf(cq.z, fmaf(kf, qi.z, acc[2]), kf * ct.z);
                d.w += fmaf(cq.w, fmaf(kf, qi.w, acc[3]), kf * ct.w);
                *dst = d;
            }
        }
    }
}

// generic owner-walk scatter-add over sorted lists (deterministic mode of the unfused gradient ops):
//   out[b][d][ch] (+)= sum over the rows e of cloud b with idx[b][e] == d, ascending e, of  w[b][e] * src[b][e / div][ch]
// one thread per (destination point, channel); adjacent threads read adjacent channels of the same rows
__global__ __launch_bounds__(256) void scatter_rows_sorted_kernel(long long total, int rows, int ndst, int c, int div,
                                                                  int ld, const float *__restrict__ w,
                                                                  const float *__restrict__ src,
                                                                  const int2 *__restrict__ order,
                                                                  const int *__restrict__ start,
                                                                  float *__restrict__ out, int accumulate) {
    const int rows_src = rows / div;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pt = e / c;
        const int ch = (int)(e - pt * c);
        const int b = (int)(pt / ndst), d = (int)(pt - (long long)b * ndst);
        const int *sb = start + (long long)b * (ndst + 1);
        const int2 *ob = order + (long long)b * rows;
        const float *sp = src + (long long)b * rows_src * ld + ch;
        const float *wb = w ? w + (long long)b * rows : nullptr;
        float acc = 0.f;
        for (int k = sb[d]; k < sb[d + 1]; ++k) {
            const int r = ob[k].x;
            const float v = sp[(long long)(r / div) * ld];
            acc += wb ? wb[r] * v : v;
        }
        out[e] = accumulate ? out[e] + acc : acc;
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

static bool scatter_csr_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_SCATTER_CSR");
        return !(e && e[0] == '0');
    }();
    return on;
}

// owner walk outside deterministic mode: measured 2569 us against 592 us of the chunked form on the SA2 scatter of the
// headline step (a few low-index points own lists hundreds of rows long and serialise their waves) -- off by default
static bool scatter_owner_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_SCATTER_OWNER");
        return e && e[0] == '1';
    }();
    return on;
}

static bool scatter_lds_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_SCATTER_LDS");
        return !(e && e[0] == '0');
    }();
    return on;
}

extern "C" {

static int gather_groups_per_block(long long G) { return G >= 8192 ? 8 : 1; }   // few groups: one workgroup each
int pcops_sa_gather_stats_rows(long long G) {
    const int gpb = gather_groups_per_block(G);
    return (int)((G + gpb - 1) / gpb);
}
int pcops_sa_scatter_rows(int b, int m) {
    // rows of the caller's weight-gradient scratch: one per cloud (LDS kernel) or per 16 groups (fallback kernel)
    const long long fb = ((long long)b * m + 15) / 16;
    const long long r = fb > b ? fb : b;
    return (int)(r > kCsrGrid ? r : kCsrGrid);
}

unsigned long long pcops_sa_scatter_workspace_bytes(int b, int n, int m, int s) {
    // order | start | order rewritten in ascending row order (deterministic mode)
    return sizeof(int) * (4ull * b * m * s + (unsigned long long)b * (n + 1) + 2);
}

unsigned long long pcops_scatter_rows_workspace_bytes(int b, int rows, int ndst) {
    return sizeof(int) * (4ull * b * rows + (unsigned long long)b * (ndst + 1) + 2);
}

// the per-cloud counting sort keeps two ndst-sized tables (+ 1024 ints of scan scratch) in the 160 KB of LDS
int pcops_scatter_rows_sorted_max_ndst(void) { return (160 * 1024 / (int)sizeof(int) - 1024) / 2; }

int pcops_scatter_rows_sorted_supported(int rows, int ndst) {
    return (ndst >= 1 && ndst <= pcops_scatter_rows_sorted_max_ndst() && rows >= 0 && (long long)rows < (1ll << 30)) ? 1 : 0;
}

int pcops_scatter_rows_sorted(int b, int rows, int ndst, int c, int div, int ld_src, const int *idx, const float *w,
                              const float *src, float *out, int accumulate, void *workspace, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && rows >= 0 && ndst >= 1 && c >= 1 && div >= 1 && rows % div == 0 && ld_src >= c);
    if (b == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(out); PCOPS_REQUIRE_PTR(workspace);
    if (rows > 0) { PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(src); }
    if (reinterpret_cast<uintptr_t>(workspace) & 7) return PCOPS_ERR_UNSUPPORTED;
    const size_t blds = (2 * (size_t)ndst + 1024) * sizeof(int);
    if (!pcops_scatter_rows_sorted_supported(rows, ndst)) return PCOPS_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    int2 *order = static_cast<int2 *>(workspace);
    int *start = reinterpret_cast<int *>(order + (size_t)b * rows);
    int2 *sorted = reinterpret_cast<int2 *>(start + (((size_t)b * (ndst + 1) + 1) & ~(size_t)1));
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(sa_csr_build_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    hipLaunchKernelGGL(sa_csr_build_kernel, dim3(b), dim3(1024), blds, st, ndst, rows, idx, order, start, 1, rows > 0 ? rows : 1,
                       (const RowBlock *)nullptr, (const int *)nullptr, sorted);
    const long long total = (long long)b * ndst * c;
    const unsigned grid = cdiv(total, 256) < 32768u ? cdiv(total, 256) : 32768u;
    hipLaunchKernelGGL(scatter_rows_sorted_kernel, dim3(grid), dim3(256), 0, st, total, rows, ndst, c, div, ld_src, w, src,
                       sorted, start, out, accumulate);
    return pcops_launch_status();
}

unsigned long long pcops_rows_max_blocks(int b, int m, int s) {
    return (unsigned long long)b * m * (s / kBlk);
}

int pcops_rows_plan(int b, int m, int s, const int *pts_cnt, void *blocks, int *block_start, int *rows,
                    pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && m >= 1 && s >= kBlk && s % kBlk == 0 && s <= 256);
    PCOPS_REQUIRE_PTR(pts_cnt); PCOPS_REQUIRE_PTR(blocks); PCOPS_REQUIRE_PTR(block_start); PCOPS_REQUIRE_PTR(rows);
    if (reinterpret_cast<uintptr_t>(blocks) & 15) return PCOPS_ERR_UNSUPPORTED;
    const long long G = (long long)b * m;
    PCOPS_REQUIRE_SHAPE(G * (s / kBlk) < (1ll << 27));         // 16 x blocks rows fit an int
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(rows_plan_scan_kernel, dim3(1), dim3(1024), 0, st, (int)G, s, pts_cnt, block_start, rows);
    const long long total = G * (s / kBlk);
    const unsigned grid = cdiv(total, 256) < 4096u ? cdiv(total, 256) : 4096u;
    hipLaunchKernelGGL(rows_plan_fill_kernel, dim3(grid), dim3(256), 0, st, total, s, pts_cnt, block_start,
                       static_cast<RowBlock *>(blocks));
    return pcops_launch_status();
}

// shapes the gather (CSR) formulation of the feature gradient takes -- the only one that walks compacted rows
static bool scatter_csr_shape(int n, int m, int s, int c) {
    return (c == 32 || c == 64 || c == 128 || (c > 0 && c % 256 == 0)) && n <= 16384 && (long long)m * s < (1ll << 30) &&
           2 * (size_t)n * 4 + 4096 <= 160 * 1024;
}

int pcops_sa_scatter_rows_supported(int n, int m, int s, int c) {
    return (scatter_csr_enabled() && s % kBlk == 0 && scatter_csr_shape(n, m, s, c)) ? 1 : 0;
}

static int gather_rows_ok(const pcops_rows_t *rows, int s) {
    if (!rows) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(rows->blocks); PCOPS_REQUIRE_PTR(rows->block_start); PCOPS_REQUIRE_PTR(rows->rows);
    PCOPS_REQUIRE_SHAPE(s % kBlk == 0);
    return PCOPS_OK;
}

// the Q + Ctr form with a stored Y (the T-Net's first layer, dgcnn/models/transform_nets.py:18) runs on edgeconv.hip's
// forward kernel, which writes one row of partial statistics per 64 groups
static bool gather_fwd_is_ec(int b, int n, int m, int s, int c, bool has_q, bool has_ctr, bool other_terms, bool compacted) {
    return has_q && has_ctr && !other_terms && !compacted && ec_fwd_supported(b, n, m, s, c);
}
int pcops_sa_gather_fwd_stats_rows(int b, int n, int m, int s, int c, int has_q, int has_ctr, int other_terms, int compacted) {
    if (gather_fwd_is_ec(b, n, m, s, c, has_q != 0, has_ctr != 0, other_terms != 0, compacted != 0))
        return ec_stats_rows((long long)b * m);
    return pcops_sa_gather_stats_rows((long long)b * m);
}

int pcops_sa_gather_fwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const float *xyz,
                        const float *new_xyz, const float *Wxyz, const float *bias, const int *idx, float *Y,
                        float *off4, float *stats_partial, const float *stat_pivot, float *moments,
                        pcops_stream_t stream) {
    return pcops_sa_gather_fwd_rows(b, n, m, s, c, Q, Ctr, xyz, new_xyz, Wxyz, bias, idx, Y, off4, stats_partial,
                                    stat_pivot, moments, nullptr, stream);
}

int pcops_sa_gather_fwd_rows(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const float *xyz,
                             const float *new_xyz, const float *Wxyz, const float *bias, const int *idx, float *Y,
                             float *off4, float *stats_partial, const float *stat_pivot, float *moments,
                             const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && c >= 4 && c % 4 == 0);
    {
        const int rrc = gather_rows_ok(rows, s);
        if (rrc) return rrc;
    }
    PCOPS_REQUIRE_SHAPE(c <= 1024 && 256 % (c / 4) == 0);
    const long long G = (long long)b * m;
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(idx);
    PCOPS_REQUIRE_ARG(Y != nullptr || stats_partial != nullptr || off4 != nullptr);
    PCOPS_REQUIRE_ARG(Q != nullptr || Wxyz != nullptr);
    if (off4 || moments) PCOPS_REQUIRE_PTR(Wxyz);
    if (Wxyz) { PCOPS_REQUIRE_PTR(xyz); PCOPS_REQUIRE_PTR(new_xyz); }
    if (gather_fwd_is_ec(b, n, m, s, c, Q != nullptr, Ctr != nullptr, Wxyz != nullptr || bias != nullptr || off4 != nullptr ||
                         moments != nullptr, rows != nullptr) && Y != nullptr) {
        // (writes pcops_sa_gather_fwd_stats_rows(...) rows of partial statistics -- fewer than the shape-less upper bound
        // pcops_sa_gather_stats_rows(G): ABI version 4, pcops.h)
        return ec_gather_fwd(b, n, m, s, c, Q, c, Ctr, c, idx, Y, stats_partial, stat_pivot, as_stream(stream));
    }
    const int rl = 256 / (c / 4);
    static const bool nt_on = [] { const char *e = getenv("PCOPS_NT_STORE"); return !(e && e[0] == '0'); }();   // kernel A/B only
    const size_t staged = (size_t)(s >= 1024 ? s : 1024) * 4;      // floats: (dx, dy, dz, index) per staged row
    if (((size_t)rl * 2 * c + staged) * sizeof(float) > 64 * 1024) return PCOPS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sa_gather_fwd_kernel, dim3(pcops_sa_gather_stats_rows(G)), dim3(256),
                       ((size_t)rl * 2 * c + staged) * sizeof(float), as_stream(stream), G, n, m, s, c, Q, Ctr, xyz, new_xyz,
                       Wxyz, bias, idx, Y, off4, stats_partial, stat_pivot, moments, gather_groups_per_block(G),
                       rows ? static_cast<const RowBlock *>(rows->blocks) : nullptr, rows ? rows->block_start : nullptr,
                       (nt_on && Y && G * s * c * 4 >= (256ll << 20)) ? 1 : 0);
    return pcops_launch_status();
}

int pcops_sa_scatter_bwd(int b, int n, int m, int s, int c, const float *G, const float *Y, const float *p,
                         const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                         const float *pool_scale, const float *pool_shift, const int *idx, const float *xyz,
                         const float *new_xyz, float *dQ, float *dCtr, float *wpartial, float *dWxyz,
                         float *dbias, const float *fwd_Q, const float *fwd_Ctr, const float *fwd_Wxyz,
                         const float *fwd_bias, void *workspace, pcops_stream_t stream) {
    return pcops_sa_scatter_bwd_rows(b, n, m, s, c, G, Y, p, q, t, gpool, argmax, pool_scale, pool_shift, idx, xyz,
                                     new_xyz, dQ, dCtr, wpartial, dWxyz, dbias, fwd_Q, fwd_Ctr, fwd_Wxyz, fwd_bias,
                                     workspace, nullptr, stream);
}

int pcops_sa_scatter_bwd_rows(int b, int n, int m, int s, int c, const float *G, const float *Y, const float *p,
                              const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                              const float *pool_scale, const float *pool_shift, const int *idx, const float *xyz,
                              const float *new_xyz, float *dQ, float *dCtr, float *wpartial, float *dWxyz,
                              float *dbias, const float *fwd_Q, const float *fwd_Ctr, const float *fwd_Wxyz,
                              const float *fwd_bias, void *workspace, const pcops_rows_t *rows,
                              pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && c >= 4 && c % 4 == 0);
    {
        const int rrc = gather_rows_ok(rows, s);
        if (rrc) return rrc;
    }
    const RowBlock *rblocks = rows ? static_cast<const RowBlock *>(rows->blocks) : nullptr;
    const int *rbstart = rows ? rows->block_start : nullptr;
    // compacted rows: only the gather (CSR) formulation of the feature gradient walks them -- G materialised, no
    // per-group output; anything else is the caller's uncompacted path
    if (rows && (gpool || dCtr || !dQ || !workspace || !G)) return PCOPS_ERR_UNSUPPORTED;
    const bool rc_fwd = (fwd_Wxyz || fwd_bias) && !fwd_Q && !fwd_Ctr && !dQ;   // Y rebuilt, never read
    if (fwd_Wxyz) { PCOPS_REQUIRE_PTR(xyz); PCOPS_REQUIRE_PTR(new_xyz); }
    PCOPS_REQUIRE_SHAPE(c <= 1024 && (c >= 256 || 256 % c == 0));
    const long long Gn = (long long)b * m;
    hipStream_t st = as_stream(stream);
    if (Gn == 0) {
        if (dQ && hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess) return PCOPS_ERR_LAUNCH;
        return PCOPS_OK;
    }
    if (!rc_fwd) PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(idx);
    if (xyz) { PCOPS_REQUIRE_PTR(new_xyz); }
    if (dWxyz || dbias) PCOPS_REQUIRE_PTR(wpartial);
    float *wp = (dWxyz || dbias) ? wpartial : nullptr;
    if (gpool) {
        PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale); PCOPS_REQUIRE_PTR(pool_shift);
        PCOPS_REQUIRE_SHAPE(s <= 256);
    } else {
        PCOPS_REQUIRE_PTR(G);
    }
    if (!rows && !gpool && G && dQ && dCtr && fwd_Q && fwd_Ctr && !fwd_Wxyz && !fwd_bias && !wp && workspace &&
        !pcops_get_deterministic() && ec_bwd_supported(b, n, m, s, c)) {
        // round 5 (edgeconv.hip), the Q + Ctr form:  sum over the rows of a point of  q Y  is  q (cnt Q[i] + sum Ctr[g]),
        // so Y is not read at all and G is read twice (streamed per group for dCtr, gathered per point for dQ)
        int rc = ec_csr_build(b, n, m, s, idx, workspace, st);
        if (rc) return rc;
        rc = ec_tnet_ctr(b, n, m, s, c, fwd_Q, c, fwd_Ctr, c, G, idx, p, q, t, dCtr, c, st);
        if (rc) return rc;
        return ec_walk(b, n, m, s, c, fwd_Q, c, fwd_Ctr, c, G, p, q, t, workspace, dQ, c, st);
    }
    // gather formulation: feature gradient wanted, G materialised, no per-group output
    const int lpr = c <= 256 ? c / 4 : 64;
    const bool csr_shape = scatter_csr_shape(n, m, s, c);
    if (rows && !(scatter_csr_enabled() && csr_shape)) return PCOPS_ERR_UNSUPPORTED;
    const bool det = pcops_get_deterministic() != 0;
    // deterministic mode: only the owner walk below adds a feature gradient in a fixed order
    // (the pooled single-layer form adds its arg-row term with atomics: no ordered variant)
    if (det && dQ && !(scatter_csr_enabled() && workspace && csr_shape && !gpool)) return PCOPS_ERR_UNSUPPORTED;
    if (scatter_csr_enabled() && workspace && dQ && csr_shape) {
        const bool split = dCtr || gpool;        // per-group outputs / pooled form: streaming pass first
        const bool owner = !gpool && (det || scatter_owner_enabled());
        if (!owner && hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess) return PCOPS_ERR_LAUNCH;
        if (split) {
            size_t lb = 0;
            const int cs0 = scatter_lds_slice(n, c, false, xyz != nullptr, &lb);
            if (!cs0) return PCOPS_ERR_UNSUPPORTED;
            const int nsl = c / cs0;
            int gsplit = kCsrGrid / ((b + 7) / 8 * 8 * nsl);        // enough workgroups to fill the chip
            if (gsplit > (m + 15) / 16) gsplit = (m + 15) / 16;
            if (gsplit < 1) gsplit = 1;
            ScatterArgs sa = {b, n, m, s, c, nsl, gsplit, G, Y, p, q, t, gpool, argmax, pool_scale, pool_shift, idx,
                              xyz, new_xyz, nullptr, dCtr, wp, gpool ? dQ : nullptr, fwd_Q, fwd_Ctr, fwd_Wxyz,
                              fwd_bias};
            int rc0 = PCOPS_ERR_UNSUPPORTED;
            switch (cs0) {
                case 64: rc0 = gpool ? launch_scatter_lds<true, 64>(sa, lb, st) : launch_scatter_lds<false, 64>(sa, lb, st); break;
                case 32: rc0 = gpool ? launch_scatter_lds<true, 32>(sa, lb, st) : launch_scatter_lds<false, 32>(sa, lb, st); break;
                case 16: rc0 = gpool ? launch_scatter_lds<true, 16>(sa, lb, st) : launch_scatter_lds<false, 16>(sa, lb, st); break;
                case 8: rc0 = gpool ? launch_scatter_lds<true, 8>(sa, lb, st) : launch_scatter_lds<false, 8>(sa, lb, st); break;
            }
            if (rc0) return rc0;
            if (wp) {
                const int rows = b * gsplit;
                if (dWxyz) hipLaunchKernelGGL(sum_rows_kernel, dim3(3 * c), dim3(256), 0, st, rows, 4 * c, wp, dWxyz);
                if (dbias) hipLaunchKernelGGL(sum_rows_kernel, dim3(c), dim3(256), 0, st, rows, 4 * c, wp + 3 * c, dbias);
            }
        }
        int2 *order = static_cast<int2 *>(workspace);
        int *start = reinterpret_cast<int *>(order + (size_t)b * m * s);
        const size_t blds = (2 * (size_t)n + 1024) * sizeof(int);
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sa_csr_build_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        int2 *sorted = det ? reinterpret_cast<int2 *>(start + (((size_t)b * (n + 1) + 1) & ~(size_t)1)) : nullptr;
        hipLaunchKernelGGL(sa_csr_build_kernel, dim3(b), dim3(1024), blds, st, n, m * s, idx, order, start, m, s, rblocks,
                           rbstart, sorted);
        float *wp2 = split ? nullptr : wp;
        CsrArgs a = {b, n, m, s, c, G, Y, p, q, t, xyz, new_xyz, sorted ? sorted : order, dQ, wp2, rblocks, rbstart, start};
        if (owner) {
            switch (lpr) {
                case 8: hipLaunchKernelGGL(sa_scatter_owner_kernel<8>, dim3(kCsrGrid), dim3(256), 0, st, a); break;
                case 16: hipLaunchKernelGGL(sa_scatter_owner_kernel<16>, dim3(kCsrGrid), dim3(256), 0, st, a); break;
                case 32: hipLaunchKernelGGL(sa_scatter_owner_kernel<32>, dim3(kCsrGrid), dim3(256), 0, st, a); break;
                case 64: hipLaunchKernelGGL(sa_scatter_owner_kernel<64>, dim3(kCsrGrid), dim3(256), 0, st, a); break;
                default: return PCOPS_ERR_UNSUPPORTED;
            }
            int rc = pcops_launch_status();
            if (rc) return rc;
            if (wp2) {
                if (dWxyz) hipLaunchKernelGGL(sum_rows_kernel, dim3(3 * c), dim3(256), 0, st, kCsrGrid, 4 * c, wp2, dWxyz);
                if (dbias) hipLaunchKernelGGL(sum_rows_kernel, dim3(c), dim3(256), 0, st, kCsrGrid, 4 * c, wp2 + 3 * c, dbias);
                rc = pcops_launch_status();
            }
            return rc;
        }
        const bool small = (long long)b * ((m * s + 63) / 64) < 4 * kCsrGrid;   // fewer 64-row chunks than waves
#define PCOPS_CSR_LAUNCH(LPR_, Y_, CH_)                                                                            \
    hipLaunchKernelGGL((sa_scatter_csr_kernel<LPR_, Y_, CH_>), dim3(kCsrGrid), dim3(256), 0, st, a)
#define PCOPS_CSR_CASE(LPR_)                                             \
    case LPR_:                                                           \
        if (gpool) {                                                     \
            if (small) PCOPS_CSR_LAUNCH(LPR_, true, 16);                 \
            else PCOPS_CSR_LAUNCH(LPR_, true, 64);                       \
        } else {                                                         \
            if (small) PCOPS_CSR_LAUNCH(LPR_, false, 16);                \
            else PCOPS_CSR_LAUNCH(LPR_, false, 64);                      \
        }                                                                \
        break;
        switch (lpr) {
            PCOPS_CSR_CASE(8)
            PCOPS_CSR_CASE(16)
            PCOPS_CSR_CASE(32)
            PCOPS_CSR_CASE(64)
            default: return PCOPS_ERR_UNSUPPORTED;
        }
#undef PCOPS_CSR_LAUNCH
#undef PCOPS_CSR_CASE
        int rc = pcops_launch_status();
        if (rc) return rc;
        if (wp2) {
            if (dWxyz) hipLaunchKernelGGL(sum_rows_kernel, dim3(3 * c), dim3(256), 0, st, kCsrGrid, 4 * c, wp2, dWxyz);
            if (dbias) hipLaunchKernelGGL(sum_rows_kernel, dim3(c), dim3(256), 0, st, kCsrGrid, 4 * c, wp2 + 3 * c, dbias);
            rc = pcops_launch_status();
        }
        return rc;
    }
    size_t lds_bytes = 0;
    const int cs = scatter_lds_enabled() ? scatter_lds_slice(n, c, dQ != nullptr, xyz != nullptr, &lds_bytes) : 0;
    if (cs) {
        int gsplit = 1;
        if (!dQ) {                               // pure streaming: deal a cloud's groups to several workgroups
            gsplit = kCsrGrid / ((b + 7) / 8 * 8 * (c / cs));
            if (gsplit > (m + 15) / 16) gsplit = (m + 15) / 16;
            if (gsplit < 1) gsplit = 1;
        }
        ScatterArgs a = {b, n, m, s, c, c / cs, gsplit, G, Y, p, q, t, gpool, argmax, pool_scale, pool_shift, idx,
                         xyz, new_xyz, dQ, dCtr, wp, nullptr, fwd_Q, fwd_Ctr, fwd_Wxyz, fwd_bias};
        int rc;
#define PCOPS_SCATTER_CASE(CS_)                                                                   \
    case CS_:                                                                                     \
        rc = gpool ? launch_scatter_lds<true, CS_>(a, lds_bytes, st) : launch_scatter_lds<false, CS_>(a, lds_bytes, st); \
        break;
        switch (cs) {
            PCOPS_SCATTER_CASE(64)
            PCOPS_SCATTER_CASE(32)
            PCOPS_SCATTER_CASE(16)
            PCOPS_SCATTER_CASE(8)
            default: rc = PCOPS_ERR_UNSUPPORTED;
        }
#undef PCOPS_SCATTER_CASE
        if (rc) return rc;
        if (wp) {
            const int rows = b * gsplit;
            if (dWxyz) hipLaunchKernelGGL(sum_rows_kernel, dim3(3 * c), dim3(256), 0, st, rows, 4 * c, wp, dWxyz);
            if (dbias) hipLaunchKernelGGL(sum_rows_kernel, dim3(c), dim3(256), 0, st, rows, 4 * c, wp + 3 * c, dbias);
            rc = pcops_launch_status();
        }
        return rc;
    }
    // fallback: global atomics (clouds too large for an LDS-resident slice); reads the stored Y
    PCOPS_REQUIRE_PTR(Y);
    if (dQ && hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess) return PCOPS_ERR_LAUNCH;
    const int rl = c >= 256 ? 1 : 256 / c;
    const size_t lds = (size_t)rl * 6 * c * sizeof(float);
    const int gpb = 16;
    const unsigned grid = (unsigned)((Gn + 15) / 16);
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

int pcops_edge_pool_stats_rows(long long G) { return (int)((G + 63) / 64); }
int pcops_edge_pool_fwd_stats_rows(int b, int n, int m, int s, int c) {
    if (ec_fwd_supported(b, n, m, s, c) && s <= 256) return ec_edge_pool_stats_rows(b, n, m);
    return pcops_edge_pool_stats_rows((long long)b * m);
}

int pcops_edge_pool_fwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const int *idx,
                        const float *gamma, float *SQ, float *qsel, unsigned char *arg, float *stats_partial,
                        const float *stat_pivot, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && s <= 256 && c >= 4 && c % 4 == 0);
    PCOPS_REQUIRE_SHAPE(c <= 1024 && 256 % (c / 4) == 0);
    const long long G = (long long)b * m;
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(gamma);
    PCOPS_REQUIRE_PTR(SQ); PCOPS_REQUIRE_PTR(qsel); PCOPS_REQUIRE_PTR(arg);
    // round 5: 64-channel slices, offsets staged in LDS, XCD-contiguous clouds (edgeconv.hip); same outputs, and
    // b m / 64 = pcops_edge_pool_stats_rows(G) rows of partial statistics
    if (ec_fwd_supported(b, n, m, s, c) && s <= 256) {
        // (writes pcops_edge_pool_fwd_stats_rows(...) rows, not the shape-less upper bound: ABI version 4, pcops.h)
        return ec_edge_pool_fwd(b, n, m, s, c, Q, c, Ctr, c, idx, gamma, SQ, qsel, arg, stats_partial, stat_pivot, as_stream(stream));
    }
    const int gl = 256 / (c / 4);
    hipLaunchKernelGGL(edge_pool_fwd_kernel, dim3(pcops_edge_pool_stats_rows(G)), dim3(256),
                       (size_t)gl * 2 * c * sizeof(float), as_stream(stream), G, n, m, s, c, Q, Ctr, idx, gamma, SQ, qsel,
                       arg, stats_partial, stat_pivot, 64);
    return pcops_launch_status();
}

int pcops_edge_pool_out(long long G, int c, const float *qsel, const float *Ctr, const float *scale,
                        const float *shift, float *out, float *ysel, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 0 && c >= 1);
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(qsel); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift);
    PCOPS_REQUIRE_PTR(out);
    const long long total = G * c;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(edge_pool_out_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total, c, qsel, Ctr, scale,
                       shift, out, ysel);
    return pcops_launch_status();
}

int pcops_edge_pool_bwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const int *idx,
                        const float *gpool, const float *ysel, const float *SQ, const unsigned char *arg,
                        const float *scale, const float *shift, const float *p, const float *q, const float *t,
                        float *dQ, float *dCtr, void *workspace, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 0 && n >= 1 && m >= 0 && s >= 1 && s <= 256 && c >= 4);
    PCOPS_REQUIRE_SHAPE((c == 32 || c == 64 || c == 128 || c % 256 == 0) && n <= 16384);
    PCOPS_REQUIRE_PTR(dQ);
    hipStream_t st = as_stream(stream);
    const long long G = (long long)b * m;
    static const bool owner = [] { const char *e = getenv("PCOPS_EDGE_BWD_OWNER"); return !(e && e[0] == '0'); }();
    const bool det = pcops_get_deterministic() != 0;
    const size_t slice_lds = (size_t)n * kEdgeSlice * sizeof(float);
    const bool use_owner = G > 0 && c % kEdgeSlice == 0 && (det || (owner && slice_lds <= 160 * 1024));
    if (det && G > 0 && !use_owner) return PCOPS_ERR_UNSUPPORTED;
    if (!use_owner && hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)b * n * c, st) != hipSuccess) return PCOPS_ERR_LAUNCH;
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(gpool); PCOPS_REQUIRE_PTR(ysel);
    PCOPS_REQUIRE_PTR(SQ); PCOPS_REQUIRE_PTR(arg); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(p);
    PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(dCtr); PCOPS_REQUIRE_PTR(workspace);
    if (use_owner && !det && ec_bwd_supported(b, n, m, s, c) && ec_bwd_fused_ok(n, m, s, c)) {
        // round 5, second half: both terms and dCtr in ONE owner walk (edgeconv.hip ec_bwd_lds_kernel)
        int rc = ec_csr_build(b, n, m, s, idx, workspace, st);
        if (rc) return rc;
        return ec_bwd_fused(b, n, m, s, c, Q, c, Ctr, c, gpool, ysel, SQ, arg, scale, shift, p, q, t, workspace, dQ, c, dCtr, c, st);
    }
    if (use_owner && !det && ec_bwd_supported(b, n, m, s, c)) {
        // round 5 (edgeconv.hip): the arg-row term + dCtr as before (LDS slices, plain stores: initialises dQ), then the
        // dense term by a leaner owner walk over a packed inverse index, clouds XCD-contiguous
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(edge_pool_bwd_sparse_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        hipLaunchKernelGGL(edge_pool_bwd_sparse_kernel<false>, dim3(b * (c / kEdgeSlice)), dim3(1024), slice_lds, st, n, m, s, c,
                           gpool, ysel, SQ, Ctr, arg, idx, scale, shift, p, q, t, dCtr, dQ);
        int rc = ec_csr_build(b, n, m, s, idx, workspace, st);
        if (rc) return rc;
        return ec_walk(b, n, m, s, c, Q, c, Ctr, c, nullptr, p, q, t, workspace, dQ, c, st);
    }
    int2 *order = static_cast<int2 *>(workspace);
    int *start = reinterpret_cast<int *>(order + (size_t)b * m * s);
    const size_t blds = (2 * (size_t)n + 1024) * sizeof(int);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(sa_csr_build_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return PCOPS_ERR_LAUNCH;
    int2 *sorted = det ? reinterpret_cast<int2 *>(start + (((size_t)b * (n + 1) + 1) & ~(size_t)1)) : nullptr;
    hipLaunchKernelGGL(sa_csr_build_kernel, dim3(b), dim3(1024), blds, st, n, m * s, idx, order, start, m, s,
                       (const RowBlock *)nullptr, (const int *)nullptr, sorted);
    const int lpr = c <= 256 ? c / 4 : 64;
    if (use_owner) {
        // no global atomics: the sparse arg-row term through an LDS-resident slice (this also initialises dQ, no
        // memset), then the dense term by the owner of every source point.  Deterministic mode: the first kernel only
        // writes dCtr, the owner adds both terms in ascending row order.
        const EdgeSparse sp = {gpool, ysel, scale, shift, p, arg};
        const unsigned dgrid = 2048;
#define PCOPS_EDGE_DENSE(LPR_, DET_)                                                                                  \
    hipLaunchKernelGGL((edge_pool_bwd_dense_kernel<LPR_, DET_>), dim3(dgrid), dim3(256), 0, st, b, n, m, s, c, Q, Ctr, q, t, \
                       DET_ ? sorted : order, start, dQ, sp)
        if (det) {
            hipLaunchKernelGGL(edge_pool_bwd_sparse_kernel<true>, dim3(b * (c / kEdgeSlice)), dim3(1024), 0, st, n, m, s, c,
                               gpool, ysel, SQ, Ctr, arg, idx, scale, shift, p, q, t, dCtr, dQ);
            switch (lpr) {
                case 8: PCOPS_EDGE_DENSE(8, true); break;
                case 16: PCOPS_EDGE_DENSE(16, true); break;
                case 32: PCOPS_EDGE_DENSE(32, true); break;
                default: PCOPS_EDGE_DENSE(64, true); break;
            }
            return pcops_launch_status();
        }
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(edge_pool_bwd_sparse_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PCOPS_ERR_LAUNCH;
        hipLaunchKernelGGL(edge_pool_bwd_sparse_kernel<false>, dim3(b * (c / kEdgeSlice)), dim3(1024), slice_lds, st, n, m, s, c,
                           gpool, ysel, SQ, Ctr, arg, idx, scale, shift, p, q, t, dCtr, dQ);
        switch (lpr) {
            case 8: PCOPS_EDGE_DENSE(8, false); break;
            case 16: PCOPS_EDGE_DENSE(16, false); break;
            case 32: PCOPS_EDGE_DENSE(32, false); break;
            default: PCOPS_EDGE_DENSE(64, false); break;
        }
#undef PCOPS_EDGE_DENSE
        return pcops_launch_status();
    }
    const long long total = G * c;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(edge_pool_bwd_ctr_kernel, dim3(grid), dim3(256), 0, st, total, n, m, s, c, gpool, ysel, SQ, Ctr,
                       arg, idx, scale, shift, p, q, t, dCtr, dQ);
    switch (lpr) {
        case 8: hipLaunchKernelGGL(edge_pool_bwd_q_kernel<8>, dim3(2 * kCsrGrid), dim3(256), 0, st, b, n, m, s, c, Q, Ctr, q, t, order, dQ); break;
        case 16: hipLaunchKernelGGL(edge_pool_bwd_q_kernel<16>, dim3(2 * kCsrGrid), dim3(256), 0, st, b, n, m, s, c, Q, Ctr, q, t, order, dQ); break;
        case 32: hipLaunchKernelGGL(edge_pool_bwd_q_kernel<32>, dim3(2 * kCsrGrid), dim3(256), 0, st, b, n, m, s, c, Q, Ctr, q, t, order, dQ); break;
        default: hipLaunchKernelGGL(edge_pool_bwd_q_kernel<64>, dim3(2 * kCsrGrid), dim3(256), 0, st, b, n, m, s, c, Q, Ctr, q, t, order, dQ); break;
    }
    return pcops_launch_status();
}

// ---- the [Q | Ctr] forms (round 5): Q and Ctr are the column halves of ONE (b, n, 2 c) product of the layer's input
// with the concatenated weight [W_b | W_a - W_b] (dgcnn/tf_util.edge_conv_stack), so the two per-point GEMMs, their two
// weight gradients and the sum of their two data gradients become one of each.  Only the csrc/edgeconv.hip kernels take
// row strides: pcops_edge_ld_supported says whether a shape has them (the caller otherwise keeps two dense tensors).
// The QUERY also answers no in deterministic mode (the backward's arg-row sums are LDS float atomics: the caller then keeps the
// dense tensors and their ordered kernels); the ENTRY POINTS check the shape only, so that a step whose forward chose this family
// still has its backward when the switch is thrown in between (that one step is then correct but not bit-reproducible).
static bool edge_ld_shape_ok(int b, int n, int m, int s, int c) {
    return n == m && s <= 256 && ec_fwd_supported(b, n, m, s, c) && ec_bwd_supported(b, n, m, s, c) && ec_sparse_ok(n);
}
int pcops_edge_ld_supported(int b, int n, int m, int s, int c) {
    return (edge_ld_shape_ok(b, n, m, s, c) && !pcops_get_deterministic()) ? 1 : 0;
}

int pcops_edge_pool_fwd_ld(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc,
                           const int *idx, const float *gamma, float *SQ, float *qsel, unsigned char *arg,
                           float *stats_partial, const float *stat_pivot, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && n >= 1 && m >= 1 && s >= 1 && c >= 4 && ldq >= c && ldc >= c && ldq % 4 == 0 && ldc % 4 == 0);
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(gamma);
    PCOPS_REQUIRE_PTR(SQ); PCOPS_REQUIRE_PTR(qsel); PCOPS_REQUIRE_PTR(arg);
    if (!edge_ld_shape_ok(b, n, m, s, c) || (long long)b * n * ldq * 4 >= (1ll << 32)) return PCOPS_ERR_UNSUPPORTED;
    return ec_edge_pool_fwd(b, n, m, s, c, Q, ldq, Ctr, ldc, idx, gamma, SQ, qsel, arg, stats_partial, stat_pivot,
                            as_stream(stream));
}

namespace {
__global__ __launch_bounds__(256) void edge_pool_out_ld_kernel(long long total4, int C, int ldc, const float *__restrict__ qsel,
                                                               const float *__restrict__ Ctr, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, float *__restrict__ out,
                                                               float *__restrict__ ysel, float *__restrict__ out2, int ld2) {
    const int c4n = C / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const long long g = e / c4n;
        const int c = (int)(e - g * c4n) * 4;
        const float4 qs = *reinterpret_cast<const float4 *>(qsel + g * C + c);
        const float4 ct = *reinterpret_cast<const float4 *>(Ctr + g * ldc + c);
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
        const float4 y = make_float4(qs.x + ct.x, qs.y + ct.y, qs.z + ct.z, qs.w + ct.w);
        const float4 o = make_float4(fmaxf(fmaf(y.x, sc.x, sh.x), 0.f), fmaxf(fmaf(y.y, sc.y, sh.y), 0.f),
                                     fmaxf(fmaf(y.z, sc.z, sh.z), 0.f), fmaxf(fmaf(y.w, sc.w, sh.w), 0.f));
        *reinterpret_cast<float4 *>(out + g * C + c) = o;
        if (out2) *reinterpret_cast<float4 *>(out2 + g * ld2 + c) = o;       // the layer's column block of the concatenation
        if (ysel) *reinterpret_cast<float4 *>(ysel + g * C + c) = y;
    }
}
}  // namespace

int pcops_edge_pool_out_ld(long long G, int c, const float *qsel, const float *Ctr, int ldc, const float *scale,
                           const float *shift, float *out, float *ysel, pcops_stream_t stream) {
    return pcops_edge_pool_out_ld2(G, c, qsel, Ctr, ldc, scale, shift, out, ysel, nullptr, 0, stream);
}

int pcops_edge_pool_out_ld2(long long G, int c, const float *qsel, const float *Ctr, int ldc, const float *scale,
                            const float *shift, float *out, float *ysel, float *out2, int ld2, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 0 && c >= 4 && c % 4 == 0 && ldc >= c && ldc % 4 == 0);
    if (out2) PCOPS_REQUIRE_SHAPE(ld2 >= c && ld2 % 4 == 0 && (reinterpret_cast<uintptr_t>(out2) & 15) == 0);
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(qsel); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(out);
    const long long total4 = G * (c / 4);
    const unsigned grid = cdiv(total4, 256) < 16384u ? cdiv(total4, 256) : 16384u;
    hipLaunchKernelGGL(edge_pool_out_ld_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total4, c, ldc, qsel, Ctr, scale,
                       shift, out, ysel, out2, ld2);
    return pcops_launch_status();
}

int pcops_edge_pool_bwd_ld(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc,
                           const int *idx, const float *gpool, const float *ysel, const float *SQ, const unsigned char *arg,
                           const float *scale, const float *shift, const float *p, const float *q, const float *t,
                           float *dQ, int lddq, float *dCtr, int lddc, void *workspace, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && n >= 1 && m >= 1 && s >= 1 && c >= 4 && ldq >= c && ldc >= c && lddq >= c && lddc >= c);
    PCOPS_REQUIRE_SHAPE(ldq % 4 == 0 && ldc % 4 == 0 && lddq % 4 == 0 && lddc % 4 == 0);
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(gpool); PCOPS_REQUIRE_PTR(ysel);
    PCOPS_REQUIRE_PTR(SQ); PCOPS_REQUIRE_PTR(arg); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(p);
    PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(dQ); PCOPS_REQUIRE_PTR(dCtr); PCOPS_REQUIRE_PTR(workspace);
    if (!edge_ld_shape_ok(b, n, m, s, c) || (long long)b * n * ldq * 4 >= (1ll << 32)) return PCOPS_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (ec_bwd_fused_ok(n, m, s, c)) {
        int rc0 = ec_csr_build(b, n, m, s, idx, workspace, st);
        if (rc0) return rc0;
        return ec_bwd_fused(b, n, m, s, c, Q, ldq, Ctr, ldc, gpool, ysel, SQ, arg, scale, shift, p, q, t, workspace, dQ, lddq,
                            dCtr, lddc, st);
    }
    int rc = ec_sparse(b, n, m, s, c, gpool, ysel, SQ, Ctr, ldc, arg, idx, scale, shift, p, q, t, dCtr, lddc, dQ, lddq, st);
    if (rc) return rc;
    rc = ec_csr_build(b, n, m, s, idx, workspace, st);
    if (rc) return rc;
    return ec_walk(b, n, m, s, c, Q, ldq, Ctr, ldc, nullptr, p, q, t, workspace, dQ, lddq, st);
}

int pcops_sa_gather_fwd_ld(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc,
                           const int *idx, float *Y, float *stats_partial, const float *stat_pivot, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && n >= 1 && m >= 1 && s >= 1 && c >= 4 && ldq >= c && ldc >= c && ldq % 4 == 0 && ldc % 4 == 0);
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(Y);
    if (!edge_ld_shape_ok(b, n, m, s, c) || (long long)b * n * ldq * 4 >= (1ll << 32)) return PCOPS_ERR_UNSUPPORTED;
    return ec_gather_fwd(b, n, m, s, c, Q, ldq, Ctr, ldc, idx, Y, stats_partial, stat_pivot, as_stream(stream));
}

int pcops_sa_scatter_bwd_ld(int b, int n, int m, int s, int c, const float *G, const float *p, const float *q, const float *t,
                            const int *idx, const float *Q, int ldq, const float *Ctr, int ldc, float *dQ, int lddq,
                            float *dCtr, int lddc, void *workspace, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && n >= 1 && m >= 1 && s >= 1 && c >= 4 && ldq >= c && ldc >= c && lddq >= c && lddc >= c);
    PCOPS_REQUIRE_SHAPE(ldq % 4 == 0 && ldc % 4 == 0 && lddq % 4 == 0 && lddc % 4 == 0);
    PCOPS_REQUIRE_PTR(G); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(idx);
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(dQ); PCOPS_REQUIRE_PTR(dCtr); PCOPS_REQUIRE_PTR(workspace);
    if (!edge_ld_shape_ok(b, n, m, s, c) || (long long)b * n * ldq * 4 >= (1ll << 32)) return PCOPS_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    int rc = ec_csr_build(b, n, m, s, idx, workspace, st);
    if (rc) return rc;
    rc = ec_tnet_ctr(b, n, m, s, c, Q, ldq, Ctr, ldc, G, idx, p, q, t, dCtr, lddc, st);
    if (rc) return rc;
    return ec_walk(b, n, m, s, c, Q, ldq, Ctr, ldc, G, p, q, t, workspace, dQ, lddq, st);
}

// ---- first layer of a stack over whole clouds in their own order: Y = Q + Ctr[cloud]  (cloud_bias_*_kernel above)
int pcops_cloud_bias_supported(long long rows, int rows_per_group, int c) {
    return (rows >= 1 && rows_per_group >= kCloudRows && rows_per_group % kCloudRows == 0 && rows % rows_per_group == 0 && c >= 4 &&
            c % 4 == 0 && c <= 1024 && 256 % (c / 4) == 0 && rows / kCloudRows < (1ll << 31)) ? 1 : 0;
}
int pcops_cloud_bias_rows(long long rows) { return (int)((rows + kCloudRows - 1) / kCloudRows); }
int pcops_cloud_bias_fwd(long long rows, int rows_per_group, int c, const float *Q, const float *Ctr, float *Y,
                         float *stats_partial, const float *stat_pivot, pcops_stream_t stream) {
    PCOPS_REQUIRE_PTR(Q); PCOPS_REQUIRE_PTR(Ctr); PCOPS_REQUIRE_PTR(Y);
    if (!pcops_cloud_bias_supported(rows, rows_per_group, c)) return PCOPS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(Ctr) | reinterpret_cast<uintptr_t>(Y)) & 15) return PCOPS_ERR_UNSUPPORTED;
    static const bool nt_on = [] { const char *e = getenv("PCOPS_NT_STORE"); return !(e && e[0] == '0'); }();   // kernel A/B only
    const int rl = 256 / (c / 4);
    hipLaunchKernelGGL(cloud_bias_fwd_kernel, dim3(pcops_cloud_bias_rows(rows)), dim3(256), (size_t)rl * 2 * c * sizeof(float),
                       as_stream(stream), rows, rows_per_group, c, Q, Ctr, Y, stats_partial, stats_partial ? stat_pivot : nullptr,
                       (nt_on && rows * c * 4 >= (256ll << 20)) ? 1 : 0);
    return pcops_launch_status();
}
int pcops_cloud_bias_bwd(long long rows, int rows_per_group, int c, const float *G, const float *Y, const float *p, const float *q,
                         const float *t, float *dQ, float *dCtr, float *partial, pcops_stream_t stream) {
    PCOPS_REQUIRE_PTR(G); PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(dCtr); PCOPS_REQUIRE_PTR(partial);
    if (!pcops_cloud_bias_supported(rows, rows_per_group, c)) return PCOPS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dQ)) & 15) return PCOPS_ERR_UNSUPPORTED;
    const int rl = 256 / (c / 4);
    hipLaunchKernelGGL(cloud_bias_bwd_kernel, dim3(pcops_cloud_bias_rows(rows)), dim3(256), (size_t)rl * c * sizeof(float),
                       as_stream(stream), rows, c, G, Y, p, q, t, dQ, partial);
    hipLaunchKernelGGL(cloud_bias_reduce_kernel, dim3((unsigned)(rows / rows_per_group)), dim3(256), 0, as_stream(stream),
                       rows_per_group / kCloudRows, c, partial, dCtr);
    return pcops_launch_status();
}

// ---- first EdgeConv layer of a stack whose input needs no gradient (edgeconv.hip): weight gradient without a scatter
int pcops_edge_first_rows(void) { return ec_edge_first_rows(); }
int pcops_edge_first_supported(int b, int n, int m, int s, int c) { return ec_edge_first_supported(b, n, m, s, c) ? 1 : 0; }
int pcops_edge_first_moments(int b, int n, int m, int s, const float *xyz, const int *idx, float *moments_partial,
                             float *edge_rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && n >= 1 && m >= 1 && s >= 1);
    PCOPS_REQUIRE_PTR(xyz); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(moments_partial);
    if (m != n || (reinterpret_cast<uintptr_t>(edge_rows) & 15)) return PCOPS_ERR_UNSUPPORTED;   // EdgeConv graphs: group g is point g
    return ec_edge_first_moments(b, n, m, s, xyz, idx, moments_partial, edge_rows, as_stream(stream));
}
int pcops_edge_first_wgrad(int b, int n, int m, int s, int c, const float *G, const float *xyz, const int *idx,
                           float *wpartial, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(b >= 1 && n >= 1 && m >= 1 && s >= 1 && c >= 4);
    PCOPS_REQUIRE_PTR(G); PCOPS_REQUIRE_PTR(xyz); PCOPS_REQUIRE_PTR(idx); PCOPS_REQUIRE_PTR(wpartial);
    if (!ec_edge_first_supported(b, n, m, s, c) || (reinterpret_cast<uintptr_t>(G) & 15)) return PCOPS_ERR_UNSUPPORTED;
    return ec_edge_first_wgrad(b, n, m, s, c, G, xyz, idx, wpartial, as_stream(stream));
}
int pcops_edge_first_layer_grads(int P1, const float *wpartial, int P2, const float *moments_partial, int c, const float *W,
                                 const float *bias, const float *p, const float *q, const float *t, const float *sumG,
                                 const float *mean, long long rows, float *dW, float *dbias, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(P1 >= 1 && P2 >= 1 && c >= 1 && rows >= 1);
    PCOPS_REQUIRE_PTR(wpartial); PCOPS_REQUIRE_PTR(moments_partial); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(p);
    PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(sumG); PCOPS_REQUIRE_PTR(mean); PCOPS_REQUIRE_PTR(dW);
    return ec_edge_first_grads(P1, wpartial, P2, moments_partial, c, W, bias, p, q, t, sumG, mean, rows, dW, dbias,
                               as_stream(stream));
}

int pcops_xyz_first_layer_grads(int P1, const float *xyz_stats, int P2, const float *moments, int C,
                                const float *Wxyz, const float *bias, const float *p, const float *q, const float *t,
                                const float *sumG, const float *mean, long long rows, float *dWxyz, float *dbias,
                                pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(P1 >= 1 && P2 >= 1 && C >= 1 && rows >= 1);
    PCOPS_REQUIRE_PTR(xyz_stats); PCOPS_REQUIRE_PTR(moments); PCOPS_REQUIRE_PTR(Wxyz); PCOPS_REQUIRE_PTR(p);
    PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(sumG); PCOPS_REQUIRE_PTR(mean); PCOPS_REQUIRE_PTR(dWxyz);
    hipLaunchKernelGGL(xyz_first_layer_grads_kernel, dim3((C + 31) / 32), dim3(1024), 0, as_stream(stream), P1,
                       xyz_stats, P2, moments, C, Wxyz, bias, p, q, t, sumG, mean, (double)rows, dWxyz, dbias);
    return pcops_launch_status();
}

}  // extern "C"
