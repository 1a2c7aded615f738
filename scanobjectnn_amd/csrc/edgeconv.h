// edgeconv.h -- internal interface of edgeconv.hip (round-5 gather / scatter kernels of the DGCNN path) to the C-ABI
// launchers in gather.hip.  Not part of include/pcops.h: the entry points keep their signatures, these are the kernels
// they dispatch to when the shape fits (64-channel slices, whole 64-group chunks; ec_*_supported).
#pragma once
#include <hip/hip_runtime.h>

bool ec_enabled();
bool ec_fwd_supported(int b, int n, int m, int s, int c);
bool ec_bwd_supported(int b, int n, int m, int s, int c);
int ec_stats_rows(long long G);      // partial-statistics rows the L2-gather forward kernels write: one per 64 groups
int ec_edge_pool_stats_rows(int b, int n, int m);     // ... and what pcops_edge_pool_fwd writes (one per cloud on LDS slices)
// ldq / ldc / lddq / lddc: row strides (floats) of Q / Ctr / dQ / dCtr: c for dense tensors, 2 c for the column halves
// of one (b, n, 2 c) product [Q | Ctr]
// pooled EdgeConv layer (pcops_edge_pool_fwd): SQ, qsel, arg, shifted moments
int ec_edge_pool_fwd(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const int *idx,
                     const float *gamma, float *SQ, float *qsel, unsigned char *arg, float *stats, const float *pivot,
                     hipStream_t st);
// stored first layer of a gather stack, Y = Q[idx] + Ctr (pcops_sa_gather_fwd_rows in its Q + Ctr form)
int ec_gather_fwd(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const int *idx,
                  float *Y, float *stats, const float *pivot, hipStream_t st);
// inverse index of idx into `workspace` (order | start | perm | codes), then the owner walk over it
int ec_csr_build(int b, int n, int m, int s, const int *idx, void *workspace, hipStream_t st);
int ec_walk(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const float *G,
            const float *p, const float *q, const float *t, const void *workspace, float *dQ, int lddq, hipStream_t st);
int ec_tnet_ctr(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const float *G,
                const int *idx, const float *p, const float *q, const float *t, float *dCtr, int lddc, hipStream_t st);
// arg-row term of the EdgeConv backward + dCtr (initialises dQ); LDS slice of n x 16 floats
bool ec_sparse_ok(int n);
int ec_sparse(int b, int n, int m, int s, int c, const float *gpool, const float *ysel, const float *SQ, const float *Ctr,
              int ldc, const unsigned char *arg, const int *idx, const float *scale, const float *shift, const float *p,
              const float *q, const float *t, float *dCtr, int lddc, float *dQ, int lddq, hipStream_t st);
// both terms of the EdgeConv backward + dCtr in one owner walk (after ec_csr_build): no LDS atomics, dQ written once
bool ec_bwd_fused_ok(int n, int m, int s, int c);
int ec_bwd_fused(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc, const float *gpool,
                 const float *ysel, const float *SQ, const unsigned char *arg, const float *scale, const float *shift,
                 const float *p, const float *q, const float *t, const void *workspace, float *dQ, int lddq, float *dCtr,
                 int lddc, hipStream_t st);
// first EdgeConv layer of a stack whose input needs no gradient: dW (6, c) / db from E^T Gm and the edge moments
int ec_edge_first_rows();
bool ec_edge_first_supported(int b, int n, int m, int s, int c);
int ec_edge_first_moments(int b, int n, int m, int s, const float *x, const int *idx, float *part, float *e8, hipStream_t st);
int ec_edge_first_wgrad(int b, int n, int m, int s, int c, const float *G, const float *x, const int *idx, float *part,
                        hipStream_t st);
int ec_edge_first_grads(int P1, const float *wpart, int P2, const float *mpart, int c, const float *W, const float *bias,
                        const float *p, const float *q, const float *t, const float *sumG, const float *mean, long long rows,
                        float *dW, float *dbias, hipStream_t st);
