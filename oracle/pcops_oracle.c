/*
 * pcops_oracle.c -- CPU restatement of the reference's point-cloud ops.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (scanobjectnn_amd/) never links, imports or falls back to anything here.
 *
 * Each function restates the algorithm of one reference function and cites the
 * reference file:line it follows (paths relative to the reference checkout's
 * pointnet2/tf_ops/ unless stated).  Plain C, fp32, compiled with
 *     gcc -O2 -ffp-contract=off   (no -march: no FMA is ever emitted)
 * so the arithmetic is the same uncontracted IEEE fp32 the reference's CPU
 * twins execute under `g++ -O2`.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   pinned against the compiled reference (oracle/_ref, tests/golden npz files):
 *     query_ball_point idx, group_point, group_point_grad, selection_sort,
 *     three_nn, three_interpolate, three_interpolate_grad
 *   parity unpinned (no CPU code / not runnable in the reference; the
 *   restatement of the CUDA kernel or of the TF call sequence IS the pin):
 *     pts_cnt of query_ball_point, farthest_point_sample, gather_point(+grad), prob_sample,
 *     pairwise_distance / knn / get_edge_feature (TensorFlow 1.10, absent).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* grouping/tf_grouping_g.cu:3-36 (GPU) == grouping/test/query_ball_point.cpp:19-47
 * (CPU twin).  First `nsample` dataset indices, ascending, whose distance to the
 * query is < radius; the whole row is pre-filled with the first hit.  pts_cnt is
 * the `cnt` of tf_grouping_g.cu:34 (the CPU twin has no such output).
 * Rows with zero hits are left unwritten by the reference; this restatement (and
 * the product) define them as 0 and the caller passes zero-filled idx to _ref. */
void oracle_query_ball_point(int b, int n, int m, float radius, int nsample,
                             const float *xyz1, const float *xyz2, int *idx,
                             int *pts_cnt) {
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        int *row0 = idx + (size_t)i * m * nsample;
        for (int j = 0; j < m; ++j) {
            int *row = row0 + (size_t)j * nsample;
            for (int l = 0; l < nsample; ++l) row[l] = 0;
            const float qx = p2[j * 3 + 0], qy = p2[j * 3 + 1], qz = p2[j * 3 + 2];
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                const float dx = qx - p1[k * 3 + 0];
                const float dy = qy - p1[k * 3 + 1];
                const float dz = qz - p1[k * 3 + 2];
                float d = sqrtf(dx * dx + dy * dy + dz * dz);
                if (d < 1e-20f) d = 1e-20f;          /* max(.,1e-20f) */
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[cnt++] = k;
                }
            }
            if (pts_cnt) pts_cnt[(size_t)i * m + j] = cnt;
        }
    }
}

/* grouping/tf_grouping_g.cu:40-57 == test/query_ball_point.cpp:52-66 */
void oracle_group_point(int b, int n, int c, int m, int nsample,
                        const float *points, const int *idx, float *out) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                const int ii = idx[((size_t)i * m + j) * nsample + k];
                memcpy(out + (((size_t)i * m + j) * nsample + k) * c,
                       points + ((size_t)i * n + ii) * c, sizeof(float) * c);
            }
}

/* grouping/tf_grouping_g.cu:61-78 == test/query_ball_point.cpp:70-84.
 * Summation order here is (j, k) ascending like the CPU twin; the GPU reference
 * uses atomics (unordered) so comparisons against this are tolerance-based. */
void oracle_group_point_grad(int b, int n, int c, int m, int nsample,
                             const float *grad_out, const int *idx,
                             float *grad_points) {
    memset(grad_points, 0, sizeof(float) * (size_t)b * n * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                const int ii = idx[((size_t)i * m + j) * nsample + k];
                const float *g = grad_out + (((size_t)i * m + j) * nsample + k) * c;
                float *dst = grad_points + ((size_t)i * n + ii) * c;
                for (int l = 0; l < c; ++l) dst[l] += g[l];
            }
}

/* grouping/tf_grouping_g.cu:83-123 == test/selection_sort.cpp:20-63.
 * Literal, unstable selection sort of the first k positions of every row:
 * leftmost strict minimum of positions s..n-1 of the CURRENT row, swapped into
 * s.  Outputs are full (b,m,n) like the op (tf_grouping.cpp:124-126). */
void oracle_selection_sort(int b, int n, int m, int k, const float *dist,
                           int *outi, float *out) {
    for (size_t r = 0; r < (size_t)b * m; ++r) {
        const float *src = dist + r * n;
        float *v = out + r * n;
        int *ix = outi + r * n;
        for (int s = 0; s < n; ++s) { v[s] = src[s]; ix[s] = s; }
        for (int s = 0; s < k && s < n; ++s) {
            int mn = s;
            for (int t = s + 1; t < n; ++t)
                if (v[t] < v[mn]) mn = t;
            if (mn != s) {
                float tv = v[mn]; v[mn] = v[s]; v[s] = tv;
                int ti = ix[mn]; ix[mn] = ix[s]; ix[s] = ti;
            }
        }
    }
}

/* tf_grouping.py:49-74 knn_point: dist = sum_c (xyz1 - xyz2)^2 (c ascending,
 * the order TF leaves unspecified is fixed here), then selection sort, then the
 * first k columns. val (b,m,k), idx (b,m,k). */
void oracle_knn_point(int b, int n, int c, int m, int k, const float *xyz1,
                      const float *xyz2, float *val, int *idx) {
    float *dist = (float *)malloc(sizeof(float) * (size_t)m * n);
    float *sv = (float *)malloc(sizeof(float) * (size_t)m * n);
    int *si = (int *)malloc(sizeof(int) * (size_t)m * n);
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * c;
        const float *p2 = xyz2 + (size_t)i * m * c;
        for (int j = 0; j < m; ++j)
            for (int t = 0; t < n; ++t) {
                float acc = 0.f;
                for (int l = 0; l < c; ++l) {
                    const float d = p1[(size_t)t * c + l] - p2[(size_t)j * c + l];
                    acc = acc + d * d;
                }
                dist[(size_t)j * n + t] = acc;
            }
        oracle_selection_sort(1, n, m, k, dist, si, sv);
        for (int j = 0; j < m; ++j)
            for (int s = 0; s < k; ++s) {
                val[((size_t)i * m + j) * k + s] = sv[(size_t)j * n + s];
                idx[((size_t)i * m + j) * k + s] = si[(size_t)j * n + s];
            }
    }
    free(dist); free(sv); free(si);
}

/* ------------------------------------------------------------------------- */
/* sampling/tf_sampling_g.cu:105-170.  The CUDA kernel runs 512 threads per
 * cloud; thread t visits k = t, t+512, ... keeping the first strict maximum
 * (`d2 > best`, best=-1, besti=0), then a 9-level tree reduce in which the LEFT
 * slot wins ties (`if dists[i1] < dists[i2]` take right).  Net rule: argmax of the
 * running min-distance, ties -> smaller (k mod 512), then smaller k.  Threads
 * without a point contribute (-1, 0).  Restated literally (slots then tree). */
void oracle_farthest_point_sample(int b, int n, int m, const float *inp,
                                  int *out) {
    enum { BS = 512 };
    if (m <= 0) return;
    float *temp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float dists[BS];
    int dists_i[BS];
    for (int i = 0; i < b; ++i) {
        const float *p = inp + (size_t)i * n * 3;
        int old = 0;
        out[(size_t)i * m] = old;
        for (int k = 0; k < n; ++k) temp[k] = 1e38f;
        for (int j = 1; j < m; ++j) {
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int t = 0; t < BS; ++t) {
                float best = -1.f;
                int besti = 0;
                for (int k = t; k < n; k += BS) {
                    const float dx = p[k * 3 + 0] - x1;
                    const float dy = p[k * 3 + 1] - y1;
                    const float dz = p[k * 3 + 2] - z1;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const float td = temp[k];
                    const float d2 = d < td ? d : td;     /* min(d, td) */
                    if (d2 != td) temp[k] = d2;
                    if (d2 > best) { best = d2; besti = k; }
                }
                dists[t] = best;
                dists_i[t] = besti;
            }
            for (int u = 0; (1 << u) < BS; ++u)
                for (int t = 0; t < (BS >> (u + 1)); ++t) {
                    const int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
                    if (dists[i1] < dists[i2]) {
                        dists[i1] = dists[i2];
                        dists_i[i1] = dists_i[i2];
                    }
                }
            old = dists_i[0];
            out[(size_t)i * m + j] = old;
        }
    }
    free(temp);
}

/* sampling/tf_sampling_g.cu:7-81 (cumsumKernel).  An inclusive fp32 prefix sum of each row whose ASSOCIATION is part of
 * the contract, because prob_sample's integer output is read off these values: per chunk of 8192 elements, (1) groups of
 * four are summed serially -- v1, v1+v2, (v1+v2)+v3, (v3+v4)+(v1+v2) (:19-33); a ragged last group is the serial sum of
 * what exists, replicated (:34-44) -- (2) the group totals go through a work-efficient up-sweep / down-sweep (:46-67)
 * which leaves S[p] = T(aligned 2^u block ending at p) + S[p - 2^u], u = ctz(p + 1), T = balanced tree, (3) every element
 * of group g > 0 adds S[g-1] (:69-77), (4) the chunk is offset by a compensated running sum carried across chunks
 * (:79-85).  The thread/block indices of the CUDA kernel do not enter the result: every LDS slot has one writer per
 * level.  Restated as those four steps, sequentially. */
void oracle_cumsum(int b, int n, const float *inp, float *out) {
    enum { CHUNK = 8192 };
    float *b4 = (float *)malloc(sizeof(float) * CHUNK);
    float *tot = (float *)malloc(sizeof(float) * (CHUNK / 4));
    for (int i = 0; i < b; ++i) {
        const float *x = inp + (size_t)i * n;
        float *y = out + (size_t)i * n;
        float runningsum = 0.f, runningsum2 = 0.f;
        for (int j = 0; j < n; j += CHUNK) {
            const int len = n - j < CHUNK ? n - j : CHUNK;     /* n24_i */
            const int len4 = (len + 3) & ~3;                   /* n24   */
            const int n2 = len4 >> 2;
            for (int k = 0; k < len; k += 4) {
                if (k + 3 < len) {
                    float v1 = x[j + k], v2 = x[j + k + 1], v3 = x[j + k + 2], v4 = x[j + k + 3];
                    v2 += v1;
                    v4 += v3;
                    v3 += v2;
                    v4 += v2;
                    b4[k] = v1; b4[k + 1] = v2; b4[k + 2] = v3; b4[k + 3] = v4;
                    tot[k >> 2] = v4;
                } else {
                    float v = 0.f;
                    for (int k2 = k; k2 < len; ++k2) { v += x[j + k2]; b4[k2] = v; }
                    for (int k2 = len; k2 < len4; ++k2) b4[k2] = v;
                    tot[k >> 2] = v;
                }
            }
            int u = 0;
            for (; (2 << u) <= n2; ++u)
                for (int k = 0; k < (n2 >> (u + 1)); ++k)
                    tot[(((k << 1) + 2) << u) - 1] += tot[(((k << 1) + 1) << u) - 1];
            for (--u; u >= 0; --u)
                for (int k = 0; k < ((n2 - (1 << u)) >> (u + 1)); ++k)
                    tot[(((k << 1) + 3) << u) - 1] += tot[(((k << 1) + 2) << u) - 1];
            for (int k = 4; k < len4; k += 4) {
                const float p = tot[(k >> 2) - 1];
                b4[k] += p; b4[k + 1] += p; b4[k + 2] += p; b4[k + 3] += p;
            }
            for (int k = 0; k < len; ++k) y[j + k] = b4[k] + runningsum;
            const float t = tot[n2 - 1] + runningsum2;
            const float r2 = runningsum + t;
            runningsum2 = t - (r2 - runningsum);
            runningsum = r2;
        }
    }
    free(b4);
    free(tot);
}

/* sampling/tf_sampling_g.cu:83-103 (binarysearchKernel) behind probsampleLauncher (:197-200): q = r * cumsum[n-1] (one
 * fp32 product), then a descending power-of-two walk from n-1 to the SMALLEST index whose cumulative value is >= q.
 * inp_p (b,n) weights, inp_r (b,m) uniform numbers -> temp (b,n) = the cumsum above, out (b,m) int32. */
void oracle_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out) {
    oracle_cumsum(b, n, inp_p, temp);
    int base = 1;
    while (base < n) base <<= 1;
    for (int i = 0; i < b; ++i) {
        const float *d = temp + (size_t)i * n;
        for (int j = 0; j < m; ++j) {
            const float q = inp_r[(size_t)i * m + j] * d[n - 1];
            int r = n - 1;
            for (int k = base; k >= 1; k >>= 1)
                if (r >= k && d[r - k] >= q) r -= k;
            out[(size_t)i * m + j] = r;
        }
    }
}

/* sampling/tf_sampling_g.cu:172-181 */
void oracle_gather_point(int b, int n, int m, const float *inp, const int *idx,
                         float *out) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            for (int l = 0; l < 3; ++l)
                out[((size_t)i * m + j) * 3 + l] = inp[((size_t)i * n + a) * 3 + l];
        }
}

/* sampling/tf_sampling_g.cu:183-192 (atomics in the reference; j ascending here) */
void oracle_gather_point_grad(int b, int n, int m, const float *out_g,
                              const int *idx, float *inp_g) {
    memset(inp_g, 0, sizeof(float) * (size_t)b * n * 3);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            for (int l = 0; l < 3; ++l)
                inp_g[((size_t)i * n + a) * 3 + l] += out_g[((size_t)i * m + j) * 3 + l];
        }
}

/* ------------------------------------------------------------------------- */
/* 3d_interpolation/tf_interpolate.cpp:60-103.  xyz1 (b,n,3) unknown, xyz2
 * (b,m,3) known.  The reference widens a FLOAT expression to double and runs a
 * strict-< cascade from 1e40 sentinels; in pure fp32 with +inf sentinels this is
 * identical (1e40 -> +inf on the float store).  Squared distances. */
void oracle_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                     float *dist, int *idx) {
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        for (int j = 0; j < n; ++j) {
            const float x1 = p1[j * 3 + 0], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
            float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
            int i1 = 0, i2 = 0, i3 = 0;
            for (int k = 0; k < m; ++k) {
                const float dx = p2[k * 3 + 0] - x1;
                const float dy = p2[k * 3 + 1] - y1;
                const float dz = p2[k * 3 + 2] - z1;
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                else if (d < b3) { b3 = d; i3 = k; }
            }
            float *dd = dist + ((size_t)i * n + j) * 3;
            int *ii = idx + ((size_t)i * n + j) * 3;
            dd[0] = b1; dd[1] = b2; dd[2] = b3;
            ii[0] = i1; ii[1] = i2; ii[2] = i3;
        }
    }
}

/* tf_interpolate.cpp:107-127: out = p1*w1 + p2*w2 + p3*w3, left to right */
void oracle_three_interpolate(int b, int m, int c, int n, const float *points,
                              const int *idx, const float *weight, float *out) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *w = weight + ((size_t)i * n + j) * 3;
            const int *ii = idx + ((size_t)i * n + j) * 3;
            const float *q1 = points + ((size_t)i * m + ii[0]) * c;
            const float *q2 = points + ((size_t)i * m + ii[1]) * c;
            const float *q3 = points + ((size_t)i * m + ii[2]) * c;
            float *o = out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l)
                o[l] = q1[l] * w[0] + q2[l] * w[1] + q3[l] * w[2];
        }
}

/* tf_interpolate.cpp:131-153 (grad w.r.t. points only; zero-filled first like
 * the op's memset at :258) */
void oracle_three_interpolate_grad(int b, int n, int c, int m,
                                   const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points) {
    memset(grad_points, 0, sizeof(float) * (size_t)b * m * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *w = weight + ((size_t)i * n + j) * 3;
            const int *ii = idx + ((size_t)i * n + j) * 3;
            const float *g = grad_out + ((size_t)i * n + j) * c;
            for (int t = 0; t < 3; ++t) {
                float *dst = grad_points + ((size_t)i * m + ii[t]) * c;
                for (int l = 0; l < c; ++l) dst[l] += g[l] * w[t];
            }
        }
}

/* ------------------------------------------------------------------------- */
/* dgcnn/utils/tf_util.py:638-657 pairwise_distance.  TF/Eigen's accumulation
 * order is not recoverable (TensorFlow 1.10 is absent): this restatement FIXES
 * inner_ij = fmaf chain over c ascending from 0 (one rounding per product, the
 * exact arithmetic of gfx950's f32 MFMA / v_fmac chain), s_i likewise with
 * x_ic*x_ic, and the reference's association D = (s_i + (-2*inner)) + s_j.
 * "parity unpinned" w.r.t. TensorFlow. */
static float sq_norm(const float *x, int c) {
    float s = 0.f;
    for (int l = 0; l < c; ++l) s = fmaf(x[l], x[l], s);
    return s;
}
static float pair_dist(const float *xi, const float *xj, float si, float sj, int c) {
    float inner = 0.f;
    for (int l = 0; l < c; ++l) inner = fmaf(xi[l], xj[l], inner);
    return (si + (-2.f * inner)) + sj;
}
void oracle_pairwise_distance(int b, int n, int c, const float *x, float *adj) {
    float *s = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < b; ++i) {
        const float *p = x + (size_t)i * n * c;
        for (int j = 0; j < n; ++j) s[j] = sq_norm(p + (size_t)j * c, c);
        for (int j = 0; j < n; ++j)
            for (int t = 0; t < n; ++t)
                adj[((size_t)i * n + j) * n + t] =
                    pair_dist(p + (size_t)j * c, p + (size_t)t * c, s[j], s[t], c);
    }
    free(s);
}

/* dgcnn/utils/tf_util.py:660-671 knn = top_k(-adj, k): ascending distance,
 * ties -> lower index first (TF top_k contract).  rows x n -> rows x k. */
void oracle_knn_topk(int rows, int n, int k, const float *adj, int *nn_idx) {
    float *bv = (float *)malloc(sizeof(float) * (size_t)k);
    int *bi = (int *)malloc(sizeof(int) * (size_t)k);
    for (size_t r = 0; r < (size_t)rows; ++r) {
        const float *a = adj + r * n;
        int have = 0;
        for (int t = 0; t < n; ++t) {
            const float d = a[t];
            if (have == k && !(d < bv[k - 1])) continue;
            int pos = have < k ? have : k - 1;
            while (pos > 0 && d < bv[pos - 1]) {   /* strict: earlier index stays first */
                bv[pos] = bv[pos - 1]; bi[pos] = bi[pos - 1]; --pos;
            }
            bv[pos] = d; bi[pos] = t;
            if (have < k) ++have;
        }
        for (int s = 0; s < k; ++s) nn_idx[r * k + s] = s < have ? bi[s] : 0;
    }
    free(bv); free(bi);
}

/* fused pairwise_distance + knn, never materialising (n,n) per batch element */
void oracle_knn_graph(int b, int n, int c, int k, const float *x, int *nn_idx) {
    float *s = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float *row = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < b; ++i) {
        const float *p = x + (size_t)i * n * c;
        for (int j = 0; j < n; ++j) s[j] = sq_norm(p + (size_t)j * c, c);
        for (int j = 0; j < n; ++j) {
            for (int t = 0; t < n; ++t)
                row[t] = pair_dist(p + (size_t)j * c, p + (size_t)t * c, s[j], s[t], c);
            oracle_knn_topk(1, n, k, row, nn_idx + ((size_t)i * n + j) * k);
        }
    }
    free(s); free(row);
}

/* dgcnn/utils/tf_util.py:674-706 get_edge_feature: [x_i | x_j - x_i] */
void oracle_edge_feature(int b, int n, int c, int k, const float *x,
                         const int *nn_idx, float *out) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *xi = x + ((size_t)i * n + j) * c;
            for (int s = 0; s < k; ++s) {
                const int t = nn_idx[((size_t)i * n + j) * k + s];
                const float *xj = x + ((size_t)i * n + t) * c;
                float *o = out + (((size_t)i * n + j) * k + s) * 2 * c;
                for (int l = 0; l < c; ++l) { o[l] = xi[l]; o[c + l] = xj[l] - xi[l]; }
            }
        }
}

#ifdef __cplusplus
}
#endif
