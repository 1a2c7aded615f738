/*
 * pcops.h -- C ABI of libpcops.so: the MI355X (gfx950) point-cloud operator library.
 *
 * Drop-in boundary for the native layer of the reference's PointNet++ tf_ops and
 * the DGCNN kNN-graph path.  Every entry point keeps the scalar/pointer list and
 * argument meaning of the launcher it replaces (cited per function; paths are
 * relative to the reference's pointnet2/tf_ops/), adds an explicit stream, and
 * returns a status instead of void.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP), fp32 / int32, dense row-major, the
 *     (B,N,3) AoS layout of the reference at the boundary;
 *   - the caller owns every buffer; the library never allocates, frees or
 *     synchronises; work is enqueued on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream);
 *   - gradient launchers zero their output themselves (the reference's glue does a
 *     cudaMemset first: grouping/tf_grouping.cpp:204, sampling/tf_sampling.cpp:174,
 *     3d_interpolation/tf_interpolate.cpp:258);
 *   - return PCOPS_OK (0) or a negative pcops_status; argument checks mirror the
 *     OP_REQUIRES of the reference's OpKernels;
 *   - re-entrant.  The only process-wide mutable state is the pair of switches below (pcops_set_deterministic,
 *     pcops_set_option): both are atomics read at the START of every call, so a caller that wants a per-call choice
 *     sets the option, issues the call and may restore it; kernels already enqueued are not affected.
 */
#ifndef PCOPS_H
#define PCOPS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pcops_stream_t; /* hipStream_t */

/* A COMPACTED row set for the grouped shared MLP (see "compacted rows" below): device pointers only, the struct
 * itself lives on the host and is read at call time. */
typedef struct pcops_rows {
    const void *blocks;     /* device, 16 bytes per 16-row block: int group, int row_in_group of the block's first row,
                               float weight of that row (1 unless it is row 0 of its group), int 0; 16-byte aligned */
    const int *block_start; /* device [groups + 1]: first block of every group, last entry = number of blocks */
    const int *rows;        /* device [1]: number of rows = 16 * number of blocks */
} pcops_rows_t;

typedef enum pcops_status {
    PCOPS_OK = 0,
    PCOPS_ERR_NULL_POINTER = -1,
    PCOPS_ERR_BAD_SHAPE = -2,     /* negative/zero extents where the reference rejects them */
    PCOPS_ERR_BAD_ARGUMENT = -3,  /* radius <= 0, nsample <= 0, k <= 0, k > n ... */
    PCOPS_ERR_UNSUPPORTED = -4,   /* shape outside what the gfx950 kernels are built for */
    PCOPS_ERR_LAUNCH = -5         /* hipGetLastError() != hipSuccess after the launch */
} pcops_status;

const char *pcops_strerror(int status);
int pcops_abi_version(void);

/* Arithmetic options.  Each selects between two formulations of the SAME fp32-in / fp32-out product whose results are both
 * within the parity bars of tests/ (the split forms carry all 24 mantissa bits of each operand in three bf16 pieces and
 * drop products <= 2^-24 relative; the fp16 pre-filter only rejects candidates, survivors get exact distances).  Values:
 *   PCOPS_OPT_GEMM_SPLIT_BF16          0 fp32 MFMA, 1 (default) split operands on the bf16 matrix pipe, 2 split only where
 *                                      the weight pieces stay LDS-resident       (pcops_mlp_gemm_fwd* and the dX products)
 *   PCOPS_OPT_WGRAD_SPLIT_BF16         0 / 1 (default): pcops_mlp_wgrad* of layers wider than 64 on both sides
 *   PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16  0 fp32 MFMA, 1: the dX half of pcops_mlp_bwd_fused* on the bf16 pipe, 2 (default): the dW
 *                                      half too where that is faster (layers of 65..128 columns whose input rows are read)
 *   PCOPS_OPT_KNN_F16_PREFILTER        0 / 1 (default): pcops_knn_graph* at c == 64, k <= 20, n >= 256 (seeded or not)
 *   PCOPS_OPT_DGRAD_SPLIT_BF16         0 fp32 MFMA, 1 (default): pcops_mlp_gemm_dgrad* with 128..256 dY columns on the bf16 pipe in
 *                                      64-column passes (weight pieces LDS-resident), 2: 128-column passes where they fit.  The same option
 *                                      carries pcops_mlp_gemm_dgrad_top (Kp = 128 .. 1024, Kp % 32 == 0; weights resident or streamed)
 *   PCOPS_OPT_BWD_FUSED_GRAM_WGRAD     0 (default) / 1: pcops_mlp_bwd_fused_gw* take shapes (pcops_mlp_bwd_fused_gw_groups > 0);
 *                                      measured -4 % .. +1 % time against the direct form (its vector arg-row term costs what
 *                                      the halved matrix work saves); superseded by value 2 of the option above
 * pcops_set_option returns the PREVIOUS value (>= 0) or PCOPS_ERR_BAD_ARGUMENT.  The environment variables of rounds 3-4
 * (PCOPS_GEMM_BF3, PCOPS_WGRAD_BF3, PCOPS_BWD_FUSED_DX3, PCOPS_KNN_F16; round 6: PCOPS_DGRAD_BF3, PCOPS_BWD_FUSED_GW) only seed the initial values (test overrides). */
typedef enum pcops_option {
    PCOPS_OPT_GEMM_SPLIT_BF16 = 1,
    PCOPS_OPT_WGRAD_SPLIT_BF16 = 2,
    PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16 = 3,
    PCOPS_OPT_KNN_F16_PREFILTER = 4,
    PCOPS_OPT_DGRAD_SPLIT_BF16 = 5,
    PCOPS_OPT_BWD_FUSED_GRAM_WGRAD = 6,
    PCOPS_OPT_COUNT = 7
} pcops_option;
int pcops_set_option(int option, int value);
int pcops_get_option(int option);
/* diagnostics: the matrix pipe the calling thread's LAST matrix-product launch took (0 fp32 pipe or not a product, 1 bf16
 * pipe with split operands, 2 the one-pass backward's fp32 dW + split dX) -- what a roofline label should be priced on */
int pcops_last_launch_pipe(void);

/* ------------------------------------------------------------------ sampling */
/* farthestpointsamplingLauncher(b,n,m,inp,temp,out)   sampling/tf_sampling.cpp:94,
 * kernel sampling/tf_sampling_g.cu:105-170.  inp (b,n,3) -> out (b,m) int32.
 * `temp` is the reference's (32,n) scratch.  Up to n = 16384 this implementation keeps the
 * cloud AND the running min-distances in registers and ignores it (may be NULL);
 * larger clouds take the streamed form (the reference's path beyond its LDS-resident
 * points, tf_sampling_g.cu:133-141) and need pcops_farthest_point_sample_workspace_bytes(b, n)
 * = 4*b*n bytes there (0 for n <= 16384). */
int pcops_farthest_point_sample(int b, int n, int m, const float *inp, float *temp,
                                int *out, pcops_stream_t stream);
unsigned long long pcops_farthest_point_sample_workspace_bytes(int b, int n);

/* probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)   sampling/tf_sampling.cpp:65, kernels sampling/tf_sampling_g.cu:7-103,
 * launcher :197-200.  inp_p (b,n) non-negative weights, inp_r (b,m) numbers in [0,1] -> out (b,m) int32: the smallest
 * index whose cumulative weight is >= inp_r * total.  `temp` is the reference's (b,n) scratch (tf_sampling.cpp:86) and
 * receives the row cumsum; its fp32 association is the reference's (groups of four, up-/down-sweep over the group
 * totals, compensated carry across 8192-element chunks), so `out` is bit-identical, boundary cases included. */
int pcops_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp,
                      int *out, pcops_stream_t stream);

/* gatherpointLauncher(b,n,m,inp,idx,out)   sampling/tf_sampling.cpp:125, :172-181 */
int pcops_gather_point(int b, int n, int m, const float *inp, const int *idx,
                       float *out, pcops_stream_t stream);
/* scatteraddpointLauncher(b,n,m,out_g,idx,inp_g)   sampling/tf_sampling.cpp:150, :183-192 */
int pcops_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx,
                            float *inp_g, pcops_stream_t stream);

/* ------------------------------------------------------------------ grouping */
/* queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt)
 * grouping/tf_grouping.cpp:66, kernel grouping/tf_grouping_g.cu:3-36.
 * xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> idx (b,m,nsample), pts_cnt (b,m).
 * Rows with no hit are written as 0 (unspecified in the reference). */
int pcops_query_ball_point(int b, int n, int m, float radius, int nsample,
                           const float *xyz1, const float *xyz2, int *idx,
                           int *pts_cnt, pcops_stream_t stream);

/* MSG form: `nscale` radii over the same (xyz1, xyz2) pair in ONE pass over the
 * dataset (pointnet_sa_module_msg issues them back to back, utils/pointnet_util.py:
 * 175-179).  radius[s], nsample[s] are HOST arrays; idx[s] / pts_cnt[s] are host
 * arrays of device pointers.  nscale <= 4. */
int pcops_query_ball_point_multi(int b, int n, int m, int nscale, const float *radius,
                                 const int *nsample, const float *xyz1,
                                 const float *xyz2, int *const *idx,
                                 int *const *pts_cnt, pcops_stream_t stream);

/* groupPointLauncher(b,n,c,m,nsample,points,idx,out)   grouping/tf_grouping.cpp:142, :40-57 */
int pcops_group_point(int b, int n, int c, int m, int nsample, const float *points,
                      const int *idx, float *out, pcops_stream_t stream);
/* groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points)  tf_grouping.cpp:173, :61-78 */
int pcops_group_point_grad(int b, int n, int c, int m, int nsample,
                           const float *grad_out, const int *idx, float *grad_points,
                           pcops_stream_t stream);

/* selectionSortLauncher(b,n,m,k,dist,outi,out)   grouping/tf_grouping.cpp:108, :83-123.
 * Full (b,m,n) outputs like the op; first k columns sorted, literal unstable order. */
int pcops_selection_sort(int b, int n, int m, int k, const float *dist, int *outi,
                         float *out, pcops_stream_t stream);
/* knn_point(k, xyz1, xyz2) of the Python wrapper (grouping/tf_grouping.py:49-74: tile + subtract + square + reduce_sum,
 * SelectionSort, slice): xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries -> val (b,m,k) squared distances, idx (b,m,k).
 * dist[t] = sum_l (xyz1[t,l] - xyz2[j,l])^2 with l ascending and no contraction, then the literal selection sort above;
 * fused: a wave per query with its distance row in LDS, so neither the (b,m,n,c) difference tensor nor the (b,m,n)
 * matrix exists.  n <= 8192 (pcops_knn_point_supported); larger clouds: pcops_knn_point_dist + pcops_selection_sort. */
int pcops_knn_point_supported(int n);
int pcops_knn_point(int b, int n, int c, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                    pcops_stream_t stream);
int pcops_knn_point_dist(int b, int n, int c, int m, const float *xyz1, const float *xyz2, float *dist,
                         pcops_stream_t stream);

/* ------------------------------------------------------------ 3d_interpolation */
/* threenn_cpu(b,n,m,xyz1,xyz2,dist,idx)   3d_interpolation/tf_interpolate.cpp:60-103
 * (CPU-only in the reference).  xyz1 (b,n,3) unknown, xyz2 (b,m,3) known. */
int pcops_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                   float *dist, int *idx, pcops_stream_t stream);
/* threeinterpolate_cpu(b,m,c,n,points,idx,weight,out)   tf_interpolate.cpp:107-127 */
int pcops_three_interpolate(int b, int m, int c, int n, const float *points,
                            const int *idx, const float *weight, float *out,
                            pcops_stream_t stream);
/* threeinterpolate_grad_cpu(b,n,c,m,grad_out,idx,weight,grad_points)  tf_interpolate.cpp:131-153 */
int pcops_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out,
                                 const int *idx, const float *weight,
                                 float *grad_points, pcops_stream_t stream);

/* ------------------------------------------------------ deterministic backward passes
 * SURVEY section 5 / reference hazard: groupPointGrad, scatteraddpoint and threeinterpolate_grad add with float
 * atomics (tf_grouping_g.cu:61-78, tf_sampling_g.cu:183-192), so two runs of the reference differ in the last bits.
 * pcops_set_deterministic(1) (or PCOPS_DETERMINISTIC=1 in the environment) makes every backward pass of this library
 * bit-reproducible: scatter-adds are taken by ONE owner per destination point that walks the point's list of source
 * rows in ascending row order.  Entry points that only exist in an atomic form (the four *_grad launchers above and
 * below, the pooled / dCtr forms of pcops_sa_scatter_bwd) then return PCOPS_ERR_UNSUPPORTED instead of silently adding
 * in arrival order; pcops_scatter_rows_sorted is their ordered replacement. */
void pcops_set_deterministic(int on);
int pcops_get_deterministic(void);

/* out[b][d][0..c) (+)= sum over the rows e of cloud b with idx[b][e] == d, in ASCENDING e, of
 *                        w[b][e] * src[b][e / div][0..c)         (w == NULL: weight 1; accumulate != 0: "+=")
 * idx (b, rows) values in [0, ndst), src (b, rows / div, ld_src), out (b, ndst, c).  One call covers
 *   groupPointGrad      rows = m*nsample, ndst = n, div = 1          (tf_grouping.cpp:173)
 *   scatteraddpoint     rows = m, ndst = n, c = 3, div = 1           (tf_sampling.cpp:150)
 *   threeinterpolate_grad  rows = 3 n, ndst = m, div = 3, w = weight (tf_interpolate.cpp:131-153)
 *   the neighbour term of pcops_edge_feature_grad  rows = n*k, ndst = n, src = grad_out + c, ld_src = 2 c
 * workspace: pcops_scatter_rows_workspace_bytes(b, rows, ndst) bytes, 8-byte aligned.
 * Limits: ndst <= pcops_scatter_rows_sorted_max_ndst() (19 968: the per-cloud counting sort lives in LDS) and
 * rows < 2^30; pcops_scatter_rows_sorted_supported(rows, ndst) is the launcher's own predicate (PCOPS_ERR_UNSUPPORTED
 * otherwise) -- callers that may fall back to the atomic form ask it first. */
unsigned long long pcops_scatter_rows_workspace_bytes(int b, int rows, int ndst);
int pcops_scatter_rows_sorted_max_ndst(void);
int pcops_scatter_rows_sorted_supported(int rows, int ndst);
int pcops_scatter_rows_sorted(int b, int rows, int ndst, int c, int div, int ld_src, const int *idx, const float *w,
                              const float *src, float *out, int accumulate, void *workspace, pcops_stream_t stream);

/* ------------------------------------------------------------- DGCNN kNN graph */
/* The reference has no native code here (dgcnn/utils/tf_util.py:638-706 is
 * matmul + top_k + gather in TF-Python); these are the native units a TF-style
 * glue would bind for the same three functions.
 * pairwise_distance: x (b,n,c) -> adj (b,n,n), D_ij = (s_i + (-2 <x_i,x_j>)) + s_j. */
int pcops_pairwise_distance(int b, int n, int c, const float *x, float *adj,
                            pcops_stream_t stream);
/* knn = top_k(-adj,k): adj (rows,n) -> nn_idx (rows,k), ascending distance, ties ->
 * lower index. */
int pcops_knn_topk(int rows, int n, int k, const float *adj, int *nn_idx,
                   pcops_stream_t stream);
/* fused pairwise_distance + knn, never materialising (n,n): x (b,n,c) -> (b,n,k) */
int pcops_knn_graph(int b, int n, int c, int k, const float *x, int *nn_idx,
                    pcops_stream_t stream);
/* the same graph, with a hint: seed (b,n,k) names k DISTINCT points per query (DGCNN: the previous layer's neighbours,
 * dgcnn/models/dgcnn.py:31-71 rebuilds the graph on every layer's features).  The largest distance to them bounds the
 * k-th nearest distance from above, so the scan rejects everything beyond it with one compare and queues a fraction of
 * the candidates; the selection and its tie rule are untouched -- nn_idx is bit for bit pcops_knn_graph's.  The
 * precondition is CHECKED per query: a seed row with an entry outside [0, n) or with a repeated entry names fewer than k
 * distinct points and is ignored (that query is scanned without a bound) -- nn_idx is pcops_knn_graph's for ANY seed.
 * seed == NULL: pcops_knn_graph.
 * (c == 64, k <= 20, n >= 256, 16-byte aligned x, seeded or not: the pairs are first evaluated in fp16 on the 16-bit
 * matrix pipe and only those whose fp16 distance minus a rigorous error bound can still enter a list get the exact fp32
 * distance -- same indices, csrc/knn.hip knn_f16_kernel; PCOPS_OPT_KNN_F16_PREFILTER = 0 keeps the fp32-MFMA kernel.)
 * pcops_knn_graph_path says which kernel a call takes and whether that kernel USES a seed: bit 0-3 the kernel
 * (1 lane-per-query VALU, 2 fp32-MFMA distances, 3 fp16 pre-filter + exact survivors), bit 4 set when a seed is honoured
 * (the VALU kernel ignores it) -- callers decide from this whether passing the previous graph pays, instead of mirroring
 * the launcher's conditions. */
int pcops_knn_graph_seeded(int b, int n, int c, int k, const float *x, const int *seed, int *nn_idx,
                           pcops_stream_t stream);
int pcops_knn_graph_path(int b, int n, int c, int k, const float *x);
/* get_edge_feature: x (b,n,c), nn_idx (b,n,k) -> out (b,n,k,2c) = [x_i | x_j - x_i] */
int pcops_edge_feature(int b, int n, int c, int k, const float *x, const int *nn_idx,
                       float *out, pcops_stream_t stream);
/* gradient of the above w.r.t. x: grad_out (b,n,k,2c) -> grad_x (b,n,c) */
int pcops_edge_feature_grad(int b, int n, int c, int k, const float *grad_out,
                            const int *nn_idx, float *grad_x, pcops_stream_t stream);
/* the x_i half of the above only: grad_x[b,i,:] = sum_s (ga - gb)[b,i,s,:] with plain stores (deterministic mode: the
 * neighbour half is pcops_scatter_rows_sorted with accumulate = 1) */
int pcops_edge_feature_grad_central(int b, int n, int c, int k, const float *grad_out, float *grad_x,
                                    pcops_stream_t stream);

/* ------------------------------------------------- shared per-point MLP (1x1 conv + BN + ReLU [+ max-pool])
 * The reference has no native code for this stage: it is a chain of TensorFlow ops per layer
 * (pointnet2/utils/pointnet_util.py:117-127 conv2d stack + reduce_max, :223-227 FP stack;
 * pointnet2/utils/tf_util.py:120-185 conv2d = tf.nn.conv2d + bias_add + batch_norm + relu, :512-531;
 * dgcnn/models/dgcnn.py:39-48).  These are the native units a TF-style glue would bind instead: one fused
 * fp32-MFMA pass per layer and direction.  All matrices row-major; X/Y rows = (batch, point, sample)
 * flattened; per-channel vectors must be 16-byte aligned and padded to a multiple of 4 floats.
 *
 * forward:   Y[M,N] = f(X)[M,K] W[K,N] + bias,  f = identity (pro_scale == NULL) or relu(x*pro_scale[k] +
 *            pro_shift[k]) (= BN+ReLU of the previous layer applied on the fly).  stats_partial (may be NULL):
 *            float [pcops_mlp_stats_rows(M)][2][N] per-row-group column sums of (Y - pivot) and (Y - pivot)^2,
 *            pivot = stat_pivot [N] (NULL: 0).  SHIFTED MOMENTS: the variance of a batch-norm layer
 *            (pointnet2/utils/tf_util.py:512-531; TensorFlow computes it in two passes) is finalised from one-pass
 *            sums, and  E[y^2] - mean^2  on fp32 partial sums loses |mean|^2 / var digits.  Any per-channel estimate of
 *            the mean within a few standard deviations -- the layer's moving mean is the natural one -- removes that:
 *            the shift is algebraically neutral (pcops_mlp_bn_finalize adds it back to the mean), every producer of
 *            forward statistics (this family, pcops_sa_gather_fwd, pcops_edge_pool_fwd) takes the same argument.
 *            ARITHMETIC of the product (rows >= 8192, K a multiple of 8): fp32 in, fp32 out, evaluated on the bf16 matrix
 *            pipe with both operands split into three bf16 pieces (x = h + m + l to 2^-25 |x|; six exact partial
 *            products, the h.h products accumulated apart from the small ones) -- relative RMS error of an output ~5e-8
 *            ... 7e-8 for K = 64 ... 128, a third of a K-long fp32 fmaf chain's; it is NOT bit-identical to such a chain
 *            (environment PCOPS_GEMM_BF3=0 selects the fp32 matrix pipe, which is).  An infinite input gives NaN (inf - inf
 *            in the split) where the chain gives inf; NaN stays NaN. */
int pcops_mlp_stats_rows(int M);
unsigned long long pcops_mlp_reduce_workspace_bytes(int N);
int pcops_mlp_gemm_fwd(int M, int K, int N, const float *X, int ldx, const float *pro_scale,
                       const float *pro_shift, const float *W, const float *bias, float *Y,
                       float *stats_partial, const float *stat_pivot, pcops_stream_t stream);
/* forward of a max-pooled LAST layer with the neighbourhood reduction (the reference's reduce_max over nsample,
 * pointnet_util.py:127) fused into the GEMM epilogue.  BN+ReLU is monotone per channel -- increasing for
 * gamma >= 0, decreasing for gamma < 0 (scale = gamma * rstd) -- so per group of S consecutive rows
 *   ysel[g,c] = max_s y (gamma[c] >= 0) | min_s y (gamma[c] < 0),  argsel[g,c] = first row attaining it (8 bit)
 * and, once scale/shift are known, pcops_mlp_pool_select gives out = relu(scale*ysel + shift) without re-reading
 * Y.  S % 32 == 0, S <= 256, M % S == 0, ldx == K -- or (round 5) groups that are NOT whole 32-row tiles: S % 4 == 0,
 * 8 <= S < 256, lcm(S, 32) <= 256 and M a multiple of it (DGCNN's T-Net: S = 20, MSG: S = 16; a wave walks lcm(S, 32)
 * rows = whole groups, a lane's four consecutive rows always lie inside one group; split-operand kernels only);
 * PCOPS_ERR_UNSUPPORTED when the shape is outside what pcops_mlp_gemm_fwd_pool_supported(M,K,N,S) accepts.
 * pro_scale == pro_shift == NULL: X is the stack's raw input.  Y == NULL: the activation is not stored at all (it then
 * only exists as statistics and group extrema) -- enough for a forward without backward and for the algebraic backward
 * of a pooled top layer below. */
int pcops_mlp_gemm_fwd_pool_supported(int M, int K, int N, int S);
int pcops_mlp_gemm_fwd_pool(int M, int K, int N, int S, const float *X, int ldx, const float *pro_scale,
                            const float *pro_shift, const float *W, const float *bias, const float *gamma,
                            float *Y, float *stats_partial, const float *stat_pivot, float *ysel,
                            unsigned char *argsel, pcops_stream_t stream);
int pcops_mlp_pool_select(long long G, int C, const float *ysel, const float *scale, const float *shift,
                          float *out, pcops_stream_t stream);
/* batch statistics -> mean, rstd = 1/sqrt(var+eps) (biased var), scale = gamma*rstd, shift = beta - mean*scale;
 * moving_* (may be NULL) <- decay*moving + (1-decay)*batch (unbiased batch variance if unbiased_moving_var).
 * P = rows of stats_partial, R = rows the statistics run over, workspace >= pcops_mlp_reduce_workspace_bytes(N).
 * stat_pivot: the pivot the producer of stats_partial was given (NULL: 0):  mean = pivot + s1/R,  var = s2/R - (s1/R)^2,
 * reduced and finalised in fp64.  stat_pivot may alias moving_mean (each channel reads it before its update). */
int pcops_mlp_bn_finalize(int P, int N, long long R, const float *stats_partial, const float *stat_pivot,
                          void *workspace, const float *gamma, const float *beta, float eps, float decay,
                          int unbiased_moving_var, float *moving_mean, float *moving_var, float *mean,
                          float *rstd, float *scale, float *shift, pcops_stream_t stream);
/* eval mode: scale/shift from the moving statistics */
int pcops_mlp_bn_eval_coeffs(int N, const float *gamma, const float *beta, const float *moving_mean,
                             const float *moving_var, float eps, float *scale, float *shift,
                             pcops_stream_t stream);
/* out[g,c] = max_s relu(scale[c]*Y[g*S+s,c] + shift[c]), argmax[g,c] (may be NULL) = first s attaining it,
 * ysel[g,c] (may be NULL) = Y at that row.  S<=256 */
int pcops_mlp_bn_relu_maxpool(long long G, int S, int C, const float *Y, const float *scale,
                              const float *shift, float *out, unsigned char *argmax, float *ysel,
                              pcops_stream_t stream);
/* out = relu(scale*Y + shift) (stack output without pooling) */
int pcops_mlp_bn_relu_apply(long long R, int C, const float *Y, const float *scale, const float *shift,
                            float *out, pcops_stream_t stream);
/* backward.  BN backward is folded into dY = p.G + q.Y + t with G = upstream grad masked by the ReLU.
 * relu_mask_stats: Gm = Gout*[relu(bn(Y))>0] and partial (sum Gm, sum Gm*Y): [pcops_mlp_bwd_stats_rows(R)][2][C]
 * pool_bwd_stats : the same sums for a max-pooled output, from (gpool, ysel = y at the pooled row): [..pool_stats_rows(G)][2][C];
 *                  gmasked (may be NULL) [G][C] = gpool * [relu(scale*ysel + shift) > 0], the MASKED pooled gradient --
 *                  this is what the `gpool` argument of pcops_mlp_gemm_dgrad* / pcops_mlp_wgrad* has to be: those kernels
 *                  place it at the arg-max rows without looking at the ReLU again (the wave-stream data gradient adds the
 *                  one row per (group, channel) AFTER staging the dense part q.Y + t -- one multiply-add per element
 *                  instead of byte extract, two compares, select and two multiply-adds)
 * bn_bwd_coeffs  : sums -> dgamma, dbeta, p, q, t */
int pcops_mlp_bwd_stats_rows(long long R);
int pcops_mlp_bwd_pool_stats_rows(long long G);
int pcops_mlp_relu_mask_stats(long long R, int C, const float *Gout, const float *Y, const float *scale,
                              const float *shift, float *Gm, float *stats_partial, pcops_stream_t stream);
int pcops_mlp_pool_bwd_stats(long long G, int C, const float *gpool, const float *ysel, const float *scale,
                             const float *shift, float *stats_partial, float *gmasked, pcops_stream_t stream);
/* ... of a pooled gradient that arrives as two pieces and / or with a row stride (an EdgeConv output that feeds the next layer AND a
 * column block of the concatenation, dgcnn/models/dgcnn.py:39-81): the sums of (ga + gb) -- gb may be NULL -- with rows lda / ldb
 * floats apart (multiples of 4, 16-byte aligned pointers), and gsum [G][C] = ga + gb, UNMASKED and contiguous, for the data-gradient
 * kernel behind (pcops_edge_pool_bwd*); round 6: autograd's sum of the two was a launch over the tensor, a strided piece a copy. */
int pcops_mlp_pool_bwd_stats_sum(long long G, int C, const float *ga, long long lda, const float *gb, long long ldb,
                                 const float *ysel, const float *scale, const float *shift, float *stats_partial,
                                 float *gsum, pcops_stream_t stream);
int pcops_mlp_bn_bwd_coeffs(int P, int N, long long R, const float *stats_partial, void *workspace,
                            const float *gamma, const float *mean, const float *rstd, float *dgamma,
                            float *dbeta, float *p, float *q, float *t, pcops_stream_t stream);
/* dgrad: Gprev[M,Nout] = mask . (dY[M,K] Wt[K,Nout]); dY from (G,Y,p,q,t) or, when gpool != NULL, from the
 * pooled form (gpool = the MASKED pooled gradient of pcops_mlp_pool_bwd_stats, argmax, S, pool_scale, pool_shift).
 * CONTRACT since ABI version 3: gpool must ALREADY carry the ReLU mask of the pooled rows (pass the `gmasked` output of
 * pcops_mlp_pool_bwd_stats, never the raw upstream gradient) -- the dgrad / wgrad / scatter kernels no longer test
 * relu(pool_scale * y + pool_shift) themselves; pool_scale / pool_shift are still REQUIRED non-NULL with a pooled form
 * (argument validation, and room to move the test back) but are not read.  A caller written against version 2 that passes
 * the raw gradient gets wrong gradients for inactive pooled rows: compare pcops_abi_version() first (the Python binding
 * refuses a mismatch, _lib.load()).
 * Yprev != NULL: mask = [relu(bn_prev(Yprev)) > 0]
 * and stats_partial [pcops_mlp_stats_rows(M)][2][Nout] gets (sum Gprev, sum Gprev*Yprev); Yprev == NULL: plain. */
int pcops_mlp_gemm_dgrad(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                         const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                         int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                         const float *Yprev, const float *prev_scale, const float *prev_shift, float *Gprev,
                         float *stats_partial, pcops_stream_t stream);
/* wgrad: dW[K,N] = A^T dY, db[N] (may be NULL) = 1^T dY; A = X or relu(X*a_scale + a_shift); dY as above.
 * partial: caller scratch of pcops_mlp_wgrad_splits(M,K,N) * (K*N + N) floats.
 * ARITHMETIC (round 4): layers wider than 64 on both sides with >= 32 768 rows are reduced on the bf16 matrix pipe with
 * split operands (three bf16 pieces per fp32 value, six exact partial products, the large ones accumulated apart from the
 * small ones): fp32 in, fp32 out, 1.5e-7 ... 2.2e-7 relative RMS against float64 where the fp32-pipe kernel makes
 * 2.1e-7 ... 4.0e-7; not bit-identical to it (environment PCOPS_WGRAD_BF3=0 selects the fp32 pipe). */
int pcops_mlp_wgrad_splits(long long M, int K, int N);
int pcops_mlp_wgrad(long long M, int K, int N, const float *X, int ldx, const float *a_scale,
                    const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                    const float *t, const float *gpool, const unsigned char *argmax, int S,
                    const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                    pcops_stream_t stream);
int pcops_mlp_transpose(int K, int N, const float *W, float *Wt, pcops_stream_t stream);
/* Data AND weight gradient of a narrow layer in one pass (round 3): what pcops_mlp_gemm_dgrad (with Yprev) followed by
 * pcops_mlp_wgrad (A = relu(Yprev*a_scale + a_shift)) compute for the layer  Y = relu(bn(Yprev)) W + b,  W [K][N] in its
 * own layout (no transposed copy), with every tensor read ONCE.  The 64-wide layers of the reference's stacks are
 * bandwidth bound in both kernels and read the same bytes twice.
 *   pcops_mlp_bwd_fused_groups  0 when (M, K, N, S, pooled) is not taken (K <= 64, N <= 128, M >= 65536, both % 4 == 0);
 *                               otherwise the number of partial copies: partial = groups * (K N + N) floats,
 *                               stats_partial [groups][2][K] = (sum Gprev, sum Gprev*Yprev) -- the partial-row count to
 *                               hand to pcops_mlp_bn_bwd_coeffs
 *   G / Y / p / q / t / gpool / argmax / S   as pcops_mlp_gemm_dgrad
 * Same numbers as the two-kernel path up to rounding; sums in a fixed order (deterministic).  ARITHMETIC (round 4): the
 * data gradient dX = dY W^T is evaluated on the bf16 matrix pipe with split operands (three bf16 pieces per fp32 value,
 * six exact partial products, the large ones accumulated apart from the small ones): fp32 in, fp32 out, no less accurate
 * than the fp32 chain, not bit-identical to it (environment PCOPS_BWD_FUSED_DX3=0 selects the fp32 pipe); the weight
 * gradient half runs on the fp32 pipe -- round 6: on the bf16 pipe as well, from transposed pieces the staging waves split
 * (option value 2, the default), for 65..128 output columns over rows that are read (not the xyz form). */
int pcops_mlp_bwd_fused_groups(long long M, int K, int N, int S, int pooled);
int pcops_mlp_bwd_fused(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                        const float *G, const float *Y, const float *p, const float *q, const float *t,
                        const float *gpool, const unsigned char *argmax, int S, const float *W, float *partial,
                        float *dW, float *db, float *Gprev, float *stats_partial, pcops_stream_t stream);
/* ... over compacted rows (pcops_rows_t; M = the allocated row count, as for the other _rows entry points) */
int pcops_mlp_bwd_fused_rows(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                             const float *G, const float *Y, const float *p, const float *q, const float *t,
                             const float *gpool, const unsigned char *argmax, int S, const float *W, float *partial,
                             float *dW, float *db, float *Gprev, float *stats_partial, const pcops_rows_t *rows,
                             pcops_stream_t stream);
/* out [M][N] = (t + p.G) + q.Y row by row (N % 4 == 0, 16-byte aligned pointers; p / q / t [N]): the BatchNorm-backward
 * combination dY = p.G + q.Y + t as a tensor -- the data gradient of a first layer whose gather is the identity (the
 * whole-cloud group of sample_and_group_all, pointnet_util.py:59-84), where no scatter kernel is needed to form it. */
int pcops_mlp_dy_apply(long long M, int N, const float *G, const float *Y, const float *p, const float *q, const float *t,
                       float *out, pcops_stream_t stream);
/* C [M][N] = A [M][K] B [K][N], row-major with leading dimensions: the small weight x weight products and row vectors
 * around the big kernels (32 x 32 output tile per workgroup, fp32 MFMA, fixed summation order) */
int pcops_small_gemm(int M, int K, int N, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                     pcops_stream_t stream);
/* C = op(A) op(B) + bias: transA != 0 -> A is stored [K][M], transB != 0 -> B is stored [N][K]; bias [N] or NULL.
 * The three products of a fully connected layer on a few hundred rows (tf_util.py:187-213 fully_connected: Y = X W + b,
 * dX = dY W^T, dW = X^T dY) without a transposed copy of anything. */
int pcops_small_gemm_ex(int M, int K, int N, const float *A, int lda, int transA, const float *B, int ldb, int transB,
                        const float *bias, float *C, int ldc, pcops_stream_t stream);
/* ... and colsum [N] (may be NULL) = the column sums of op(B) over its K rows out of the same launch: the bias gradient
 * db = 1^T dY of fully_connected next to dW = X^T dY (tf_util.py:187-213; round 6 -- it was a separate reduction launch
 * of 12 us per layer).  Fixed summation order. */
int pcops_small_gemm_colsum(int M, int K, int N, const float *A, int lda, int transA, const float *B, int ldb, int transB,
                            const float *bias, float *C, int ldc, float *colsum, pcops_stream_t stream);
/* TWO independent products of that form in ONE launch -- p[0] and p[1]: the data and the weight gradient of a fully connected
 * layer (dX = dY W^T, dW = X^T dY with db as colsum), which share dY and each fill half the chip when launched alone; the
 * two K x K / K x N products of the algebraic top layer.  Same results as two pcops_small_gemm_colsum calls, bit for bit. */
typedef struct pcops_gemm_problem {
    int M, K, N;
    const float *A; int lda, transA;
    const float *B; int ldb, transB;
    const float *bias;              /* [N] or NULL */
    float *C; int ldc;
    float *colsum;                  /* [N] or NULL */
} pcops_gemm_problem_t;
int pcops_small_gemm_pair(const pcops_gemm_problem_t *p, pcops_stream_t stream);

/* ---- algebraic backward of a POOLED top layer (the last conv of a set-abstraction stack / of a stack pooled over a
 * whole cloud: pointnet_util.py:139-147, dgcnn.py "agg", transform_nets.py "tconv3").
 *   Y = X W + b,  X = relu(bn_prev(Yprev)) [M][Kp],  out[g] = max over the S rows of group g of relu(bn(Y))
 * dY = p.G + q.Y + t has ONE non-zero row of G per (group, channel).  Substituting Y = X W + b:
 *   dX = X (W diag(q) W^T) + 1 (W (q.b + t))^T + (p.G) W^T            Kp x Kp product instead of N x Kp
 *   dW = (X^T X) (W diag(q)) + (X^T 1)(q.b + t)^T + X^T (p.G)         Kp x Kp Gram matrix instead of Kp x N
 *   db = 1^T (p.G) + q.((X^T 1) W + M b) + M t
 * Neither needs Y; the (p.G) terms touch M/S * N rows.  N = 2 Kp .. 8 Kp in the reference's stacks, so the two large
 * products lose 1/2 .. 7/8 of their flops.  Same numbers as pcops_mlp_gemm_dgrad / pcops_mlp_wgrad up to fp32
 * rounding (different summation order).
 *   pcops_mlp_pool_top_supported  1 when the four entry points below take (M, Kp, N, S)
 *   pcops_mlp_pool_top_addend       addend [(M/S) min(S,N)][Kp] = the rows of (p.G) W^T that are not zero, compacted;
 *                                 rowmap [M] = slot of a row or -1.  gout / ysel / argmax [M/S][N] as in
 *                                 pcops_mlp_gemm_dgrad (gpool form), Wt [N][Kp] from pcops_mlp_transpose
 *   pcops_mlp_gemm_dgrad_top      Gprev [M][Kp] = mask_prev . (X Mq + addend[rowmap] + vconst), stats_partial as
 *                                 pcops_mlp_gemm_dgrad;  Mq [Kp][Kp] = W diag(q) W^T, vconst [Kp] = W (q.b + t)
 *   pcops_mlp_gram                gram [Kp][Kp] = X^T X, xsum [Kp] = X^T 1; partial: pcops_mlp_wgrad_splits(M,Kp,Kp)
 *                                 copies of (Kp Kp + Kp) floats
 *   pcops_mlp_pool_top_wsparse    Ssp [Kp][N] = X^T (p.G), cfsum [N] = 1^T (p.G)
 *   pcops_mlp_pool_top_prep       the small operands in one launch: Wt [N][Kp] = W^T, Wq [Kp][N] = W diag(q),
 *                                 u [N] = q.b + t, v [Kp] = W u (= vconst; v may be NULL)  (round 6: they were a transpose,
 *                                 two elementwise launches and a matrix-vector launch)
 *   pcops_mlp_pool_top_finish     closes the sums in place: dW [Kp][N] = (dW + Ssp) + xsum u^T with dW = gram Wq on
 *                                 entry, db [N] = (cfsum + q.(xsum^T W + M b)) + M t  (round 6: eleven launches)
 * prev_scale == prev_shift == NULL: X is the stack's raw input (a one-layer stack) -- no mask, no statistics.
 * All sums in a fixed order (deterministic). */
int pcops_mlp_pool_top_supported(int M, int Kp, int N, int S);
int pcops_mlp_pool_top_prep(int Kp, int N, const float *W, const float *b, const float *q, const float *t, float *Wt,
                            float *Wq, float *u, float *v, pcops_stream_t stream);
int pcops_mlp_pool_top_finish(int Kp, int N, long long M, float *dW, const float *Ssp, const float *xsum, const float *u,
                              const float *cfsum, const float *q, const float *W, const float *b, const float *t,
                              float *db, pcops_stream_t stream);
int pcops_mlp_pool_top_addend(int M, int Kp, int N, int S, const float *gout, const float *ysel,
                            const unsigned char *argmax, const float *pool_scale, const float *pool_shift,
                            const float *p, const float *Wt, float *addend, int *rowmap, pcops_stream_t stream);
int pcops_mlp_gemm_dgrad_top(int M, int Kp, const float *Yprev, const float *prev_scale, const float *prev_shift,
                             const float *Mq, const float *vconst, const float *addend, long long addend_rows,
                             const int *rowmap, float *Gprev, float *stats_partial, pcops_stream_t stream);
int pcops_mlp_gram(long long M, int Kp, const float *Yprev, int ldx, const float *a_scale, const float *a_shift,
                   float *partial, float *gram, float *xsum, pcops_stream_t stream);
int pcops_mlp_pool_top_wsparse(int M, int Kp, int N, int S, const float *gout, const float *ysel,
                               const unsigned char *argmax, const float *pool_scale, const float *pool_shift,
                               const float *p, const float *Yprev, const float *prev_scale, const float *prev_shift,
                               float *Ssp, float *cfsum, pcops_stream_t stream);
/* The layer that FOLLOWS an arithmetic first layer  y1[row][k] = fma(dz, w2[k], fma(dy, w1[k], fma(dx, w0[k], b[k])))
 * (grouped xyz offsets only: the first set-abstraction level, pointnet_util.py:44-54 with points == None).  y1 is
 * rebuilt from off4 [M][4] = (dx, dy, dz, 0) and xyzw [4][C1] (rows w0, w1, w2, b) wherever the plain entry points
 * would read it -- as the forward operand, as the weight gradient's A side, and in the data gradient's ReLU mask and
 * BN-backward statistics -- so the (rows, C1) first-layer tensor never exists.  Same arguments as the plain
 * functions otherwise.  Wave-stream shapes only: check pcops_mlp_xyz_supported(M, C1, N2) first
 * (PCOPS_ERR_UNSUPPORTED otherwise).
 * pcops_mlp_gemm_dgrad_xyz: xyz_stats (may be NULL) = float [pcops_mlp_stats_rows(M)][3][Nout] partial sums of
 * off[row][i] * Gprev[row][c].  The first layer's gradients are LINEAR in a handful of such sums,
 *   dWxyz[i][c] = p[c] A[i][c] + q[c] B[i][c] + t[c] S[i],   dbias[c] = p[c] sumG[c] + q[c] sumY[c] + t[c] rows
 * (A = these sums, sumG / sumGY = stats_partial, B = M33 Wxyz + S^T b and S from the offset moments the forward
 * gather emits), so with xyz_stats Gprev may be NULL: the (rows, C1) gradient is neither written nor scattered. */
int pcops_mlp_xyz_supported(int M, int C1, int N2);
/* pcops_mlp_bwd_fused over the xyz form (round 3): pcops_mlp_wgrad_xyz + pcops_mlp_gemm_dgrad_xyz (Gprev == NULL form) in
 * one pass -- dW, db of the layer, stats_partial [groups][2][K] and xyz_stats [groups][3][K] of the arithmetic layer
 * below; groups = pcops_mlp_bwd_fused_groups(M, K, N, S, pooled), 0 = not taken.  W [K][N] as the forward holds it. */
int pcops_mlp_bwd_fused_xyz_rows(long long M, int K, int N, const float *off4, const float *xyzw, const float *a_scale,
                                 const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                                 const float *t, const float *gpool, const unsigned char *argmax, int S, const float *W,
                                 float *partial, float *dW, float *db, float *stats_partial, float *xyz_stats,
                                 const pcops_rows_t *rows, pcops_stream_t stream);
int pcops_mlp_gemm_fwd_xyz(int M, int K, int N, const float *off4, const float *xyzw, const float *pro_scale,
                           const float *pro_shift, const float *W, const float *bias, float *Y,
                           float *stats_partial, const float *stat_pivot, pcops_stream_t stream);
int pcops_mlp_gemm_dgrad_xyz(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                             const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                             int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                             const float *off4, const float *xyzw, const float *prev_scale,
                             const float *prev_shift, float *Gprev, float *stats_partial, float *xyz_stats,
                             pcops_stream_t stream);
int pcops_mlp_wgrad_xyz(long long M, int K, int N, const float *off4, const float *xyzw, const float *a_scale,
                        const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                        const float *t, const float *gpool, const unsigned char *argmax, int S,
                        const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                        pcops_stream_t stream);

/* --------------------------------------------- first grouped layer in front of the grouping (gather.hip)
 * A 1x1 conv is linear: concat(xyz[idx] - new_xyz, points[idx]) W = (points W_f)[idx] + (xyz[idx] - new_xyz) W_xyz
 * (pointnet2/utils/pointnet_util.py:44-54,117-122; EdgeConv concat(x_i, x_j - x_i): dgcnn/utils/tf_util.py:699-705
 * + dgcnn.py:39-44), so the feature contraction runs once per SOURCE point and the (b,m,s,c) activation is
 *     Y[b,j,s,:] = Q[b, idx[b,j,s], :] + Ctr[b,j,:] + (xyz[b,idx,:] - new_xyz[b,j,:]) Wxyz + bias
 * Q (b,n,c), Ctr (b,m,c), xyz (b,n,3), new_xyz (b,m,3), Wxyz (3,c), bias (c), idx (b,m,s); Q, Ctr, the coordinate
 * term and bias are each optional (NULL); the coordinate term is evaluated on the centred offsets (no cancellation).
 * stats_partial (may be NULL): float [pcops_sa_gather_stats_rows(b*m)][2][c] partial sums of (Y - pivot) and
 * (Y - pivot)^2, pivot = stat_pivot [c] or 0 (shifted moments, see pcops_mlp_gemm_fwd).
 * Y may be NULL (statistics only) and off4 (may be NULL; needs the coordinate term) receives the centred offsets
 * (dx, dy, dz, 0) per grouped row, float [b*m*s][4]: when the layer has NO Q / Ctr term it is arithmetic in those
 * three numbers, and the pcops_mlp_*_xyz entry points rebuild it on the fly instead of reading a (b,m,s,c) tensor.
 * moments (may be NULL; needs the coordinate term): float [pcops_sa_gather_stats_rows(b*m)][9] partial sums of
 * (dx dx, dx dy, dx dz, dy dy, dy dz, dz dz, dx, dy, dz) -- see pcops_mlp_gemm_dgrad_xyz. */
int pcops_sa_gather_stats_rows(long long groups);
/* rows of stats_partial a pcops_sa_gather_fwd(_rows) call of this form writes: the Q + Ctr form with a stored Y runs on
 * the 64-channel-slice kernel of csrc/edgeconv.hip (one row per 64 groups) when the shape fits; everything else writes
 * pcops_sa_gather_stats_rows(b*m) rows.  has_q / has_ctr: the term is present; other_terms: any of Wxyz, bias, off4,
 * moments; compacted: a pcops_rows_t is passed. */
int pcops_sa_gather_fwd_stats_rows(int b, int n, int m, int s, int c, int has_q, int has_ctr, int other_terms,
                                   int compacted);
int pcops_sa_gather_fwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const float *xyz,
                        const float *new_xyz, const float *Wxyz, const float *bias, const int *idx, float *Y,
                        float *off4, float *stats_partial, const float *stat_pivot, float *moments,
                        pcops_stream_t stream);
/* backward through the BN+ReLU that follows: dY = p.G + q.Y + t (pooled form when gpool != NULL, as in
 * pcops_mlp_gemm_dgrad).  Outputs, each optional: dQ (b,n,c) = scatter-add of dY over idx (zeroed here),
 * dCtr (b,m,c) = sum over s, dWxyz (3,c) = sum (xyz[idx]-new_xyz)^T dY (needs xyz/new_xyz), dbias (c) = sum dY.
 * wpartial: caller scratch of pcops_sa_scatter_rows(b, m) * 4 * c floats (needed for dWxyz / dbias).
 * fwd_Q / fwd_Ctr / fwd_Wxyz / fwd_bias (all may be NULL): the arguments pcops_sa_gather_fwd was called with.  When
 * the forward had no Q / Ctr term (Y = (xyz[idx]-new_xyz) Wxyz + bias) the streaming kernel REBUILDS Y in the
 * forward's own operation order instead of reading it (bit-identical, one (b,m,s,c) tensor less to stream); Y may
 * be NULL only in that case with dQ == NULL.
 * workspace (may be NULL): pcops_sa_scatter_workspace_bytes(b,n,m,s) bytes, 16-byte aligned; with it the feature
 * gradient is computed as a GATHER over a per-cloud inverse index (counting sort of idx) -- one wave per source
 * point, no float atomics; without it (or when dCtr / the pooled form is requested) dQ is accumulated with atomics
 * in an LDS-resident slice per cloud. */
int pcops_sa_scatter_rows(int b, int m);
unsigned long long pcops_sa_scatter_workspace_bytes(int b, int n, int m, int s);
int pcops_sa_scatter_bwd(int b, int n, int m, int s, int c, const float *G, const float *Y, const float *p,
                         const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                         const float *pool_scale, const float *pool_shift, const int *idx, const float *xyz,
                         const float *new_xyz, float *dQ, float *dCtr, float *wpartial, float *dWxyz,
                         float *dbias, const float *fwd_Q, const float *fwd_Ctr, const float *fwd_Wxyz,
                         const float *fwd_bias, void *workspace, pcops_stream_t stream);

/* EdgeConv as ONE pooled layer without the (b,n,k,c) tensor in either direction (dgcnn/models/dgcnn.py:39-48 with the
 * first-layer restructuring above):  y[g,s,:] = Q[idx[g,s],:] + Ctr[g,:] -> BN -> ReLU -> max over s.  Ctr is constant
 * inside a group and BN+ReLU is monotone per channel, so
 *   forward : qsel[g,c] = max_s (gamma[c] >= 0) | min_s (gamma[c] < 0) of Q[idx[g,s],c], arg = first s attaining it,
 *             SQ[g,c] = sum_s Q[idx[g,s],c];  stats_partial [pcops_edge_pool_fwd_stats_rows(b,n,m,s,c)][2][c] = partial
 *             (sum y', sum y'^2), y' = y - stat_pivot (shifted moments, see pcops_mlp_gemm_fwd), evaluated as
 *             (SQ' + k c', SQ2' + 2 c' SQ' + k c'^2) with Q taken relative to its own first row, q' = q - Q[0,0,:], and
 *             c' = Ctr + Q[0,0,:] - pivot, so that neither a common offset of Q nor one of Ctr cancels in fp32
 *             -> pcops_mlp_bn_finalize with the same pivot
 *   out     : out = relu(scale (qsel + Ctr) + shift), ysel = qsel + Ctr
 *   backward: with (p, q, t) from pcops_mlp_pool_bwd_stats(gpool, ysel) + pcops_mlp_bn_bwd_coeffs,
 *             dCtr[g] = q (SQ + k Ctr) + k t + a[g],  a = p gpool [relu(bn(ysel)) > 0],
 *             dQ[i] = cnt_i (q Q[i] + t) + q sum_{(g,s)->i} Ctr[g] + sum_{g: arg row -> i} a[g]   (inverse index of idx)
 * workspace: pcops_sa_scatter_workspace_bytes(b,n,m,s) bytes.  s <= 256. */
int pcops_edge_pool_stats_rows(long long groups);     /* ABI >= 4: an UPPER BOUND (buffer size) only, see below */
/* ABI version 4 (round 6, was an unversioned change of round 5): pcops_edge_pool_fwd and pcops_sa_gather_fwd write the number
 * of rows THESE per-shape queries return -- rows beyond stay untouched -- so pcops_mlp_bn_finalize must be given that count,
 * not the shape-less one; a caller built against the round-4 header must be rebuilt (pcops_abi_version() tells).
 * rows of stats_partial THIS call shape writes (<= pcops_edge_pool_stats_rows(b*m)): the 64-channel-slice kernels of
 * csrc/edgeconv.hip write one row per 64 groups, or one per cloud when the cloud's slice is LDS-resident */
int pcops_edge_pool_fwd_stats_rows(int b, int n, int m, int s, int c);
int pcops_edge_pool_fwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const int *idx,
                        const float *gamma, float *SQ, float *qsel, unsigned char *arg, float *stats_partial,
                        const float *stat_pivot, pcops_stream_t stream);
int pcops_edge_pool_out(long long groups, int c, const float *qsel, const float *Ctr, const float *scale,
                        const float *shift, float *out, float *ysel, pcops_stream_t stream);
int pcops_edge_pool_bwd(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const int *idx,
                        const float *gpool, const float *ysel, const float *SQ, const unsigned char *arg,
                        const float *scale, const float *shift, const float *p, const float *q, const float *t,
                        float *dQ, float *dCtr, void *workspace, pcops_stream_t stream);
/* First layer of a stack whose groups are WHOLE CLOUDS in their own row order (round 5; reference: dgcnn_bga.py:118-128, the
 * segmentation head's concat of 1280 per-cloud channels with 320 per-point ones -- see DESIGN.md section 4.15):
 *   Y[r, :] = Q[r, :] + Ctr[r / rows_per_group, :]  with shifted-moment partials [pcops_cloud_bias_rows(rows)][2][c], and
 *   backward  dQ = p G + q Y + t  (may be NULL),  dCtr[g] = sum over the group's rows of it; partial: [..rows(rows)][c] scratch.
 * rows_per_group a multiple of 256, c % 4 == 0, c <= 1024, 256 % (c / 4) == 0.  The general form of the same layer is
 * pcops_sa_gather_fwd / pcops_sa_scatter_bwd with the identity index. */
int pcops_cloud_bias_supported(long long rows, int rows_per_group, int c);
int pcops_cloud_bias_rows(long long rows);
int pcops_cloud_bias_fwd(long long rows, int rows_per_group, int c, const float *Q, const float *Ctr, float *Y,
                         float *stats_partial, const float *stat_pivot, pcops_stream_t stream);
int pcops_cloud_bias_bwd(long long rows, int rows_per_group, int c, const float *G, const float *Y, const float *p, const float *q,
                         const float *t, float *dQ, float *dCtr, float *partial, pcops_stream_t stream);
/* First EdgeConv layer of a grouped stack whose INPUT needs no gradient (round 5; reference: the T-Net's tconv1 on the edge
 * features of the raw cloud, dgcnn/models/transform_nets.py:18-27 over tf_util.get_edge_feature, tf_util.py:660-706).  The
 * layer is linear in the six edge channels e = [x_g | x_j - x_g], j = idx[g, s]:  Y1 = e W + b, and with the BN backward
 * dY1 = p Gm + q Y1 + t of the layer
 *   dW (6, c) = p (E^T Gm) + q (E^T E W + E^T 1 b) + t E^T 1,     db = p sum Gm + q sum Y1 + t rows
 * -- one streaming pass over the masked gradient Gm (b m s, c) instead of the scatter to per-point gradients
 * (pcops_sa_scatter_bwd_ld: two passes over Gm, one a gather through an inverse index) and the GEMM backward behind it.
 *   pcops_edge_first_moments: moments_partial [pcops_edge_first_rows()][27] -- 21 second moments of e (upper triangle,
 *     row-major) then its 6 sums; forward time (xyz (b, n, 3), idx (b, m, s), m == n for an EdgeConv graph); edge_rows
 *     (may be NULL): the rows e themselves, (b m s, 8) with two pad floats, for pcops_mlp_bwd_fused_edge;
 *   pcops_edge_first_wgrad:   wpartial [pcops_edge_first_rows()][6][c] = partial sums of E^T Gm (c = 64 | 128);
 *   pcops_edge_first_layer_grads: dW (6, c), dbias (c, may be NULL) from both, p / q / t and sumG (= dbeta) from
 *     pcops_mlp_bn_bwd_coeffs, mean from pcops_mlp_bn_finalize, rows = b m s; sums in double, fixed order. */
int pcops_edge_first_rows(void);
int pcops_edge_first_supported(int b, int n, int m, int s, int c);
int pcops_edge_first_moments(int b, int n, int m, int s, const float *xyz, const int *idx, float *moments_partial,
                             float *edge_rows, pcops_stream_t stream);
int pcops_edge_first_wgrad(int b, int n, int m, int s, int c, const float *G, const float *xyz, const int *idx,
                           float *wpartial, pcops_stream_t stream);
/* the one-pass backward (pcops_mlp_bwd_fused_rows) of the POOLED layer above such a first layer: the masked gradient of
 * the first layer is not written -- E^T Gm leaves as edge_stats [pcops_mlp_bwd_fused_groups(M, K, N, S, 1)][6][K], in the
 * place of pcops_edge_first_wgrad's wpartial.  Groups of S rows that are not whole 32-row tiles (the k neighbours). */
int pcops_mlp_bwd_fused_edge(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                                  const float *Y, const float *p, const float *q, const float *t, const float *gpool,
                                  const unsigned char *argmax, int S, const float *W, float *partial, float *dW, float *db,
                                  float *stats_partial, const float *edge_rows, float *edge_stats, pcops_stream_t stream);

/* Round 6: pcops_mlp_bwd_fused / pcops_mlp_bwd_fused_edge of a POOLED layer with the weight gradient in its GRAM FORM.  With
 * dY = p.G + q.Y + t, one non-zero row of G per (group, channel), and Y = X W + bias:
 *     dW = X^T (p.G) + (X^T X) W diag(q) + (X^T 1)(q.bias + t)^T
 * -- a 64 x 64 Gram matrix on the matrix pipe instead of the K x N product (half the weight gradient's matrix time at
 * N = 128), the arg rows as vector work, the K x N product once on the summed partials.  Same reference op as
 * pcops_mlp_bwd_fused (tf.gradients of conv2d + batch_norm + reduce_max, pointnet2/utils/pointnet_util.py:117-127,
 * dgcnn/models/transform_nets.py:19-27); results equal to it to fp32 rounding.  Uncompacted rows, S % 32 == 0 or
 * 11 <= S <= 255, PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16 on; bias [N] may be NULL (a layer without bias).
 *   pcops_mlp_bwd_fused_gw_groups  workgroups = partial copies, 0 when the shape is not taken
 *   partial: groups (K N + N + K K + K) floats of scratch */
int pcops_mlp_bwd_fused_gw_groups(long long M, int K, int N, int S);
int pcops_mlp_bwd_fused_gw(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                           const float *Y, const float *p, const float *q, const float *t, const float *gpool,
                           const unsigned char *argmax, int S, const float *W, const float *bias, float *partial, float *dW,
                           float *db, float *Gprev, float *stats_partial, pcops_stream_t stream);
int pcops_mlp_bwd_fused_edge_gw(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                                const float *Y, const float *p, const float *q, const float *t, const float *gpool,
                                const unsigned char *argmax, int S, const float *W, const float *bias, float *partial,
                                float *dW, float *db, float *stats_partial, const float *edge_rows, float *edge_stats,
                                pcops_stream_t stream);
int pcops_edge_first_layer_grads(int P1, const float *wpartial, int P2, const float *moments_partial, int c, const float *W,
                                 const float *bias, const float *p, const float *q, const float *t, const float *sumG,
                                 const float *mean, long long rows, float *dW, float *dbias, pcops_stream_t stream);
/* dWxyz (3,c) and dbias (c, may be NULL) of an arithmetic first layer from the sums described at
 * pcops_mlp_gemm_dgrad_xyz: xyz_stats [P1][3][c], moments [P2][9] (pcops_sa_gather_fwd), p/q/t and sumG (= dbeta) from
 * pcops_mlp_bn_bwd_coeffs, mean from pcops_mlp_bn_finalize, rows = b*m*s. */
int pcops_xyz_first_layer_grads(int P1, const float *xyz_stats, int P2, const float *moments, int C,
                                const float *Wxyz, const float *bias, const float *p, const float *q, const float *t,
                                const float *sumG, const float *mean, long long rows, float *dWxyz, float *dbias,
                                pcops_stream_t stream);

/* ------------------------------------------------------------------ compacted rows
 * query_ball_point pads a neighbourhood that holds fewer than nsample points with copies of its first member
 * (grouping/tf_grouping_g.cu:26-32), and the reference pushes every copy through the whole shared MLP
 * (pointnet2/utils/pointnet_util.py:117-127): at the PB_T50_RS-shaped SSG config 39 % of the rows of the second
 * set-abstraction level are such copies.  A copy computes exactly what its original computes, so the grouped stack
 * runs on a COMPACTED row set instead -- per group its pts_cnt real members, rounded up to whole blocks of 16 rows
 * with copies of member 0 (16 * ceil(cnt / 16) <= nsample rows instead of nsample) -- and the first row of every
 * group carries the weight w = nsample - rows_of_the_group + 1 of the copies that were dropped:
 *   forward   BN batch statistics weigh that row with w (the max-pool ignores copies anyway);
 *   backward  the row stands for the SUM of the gradients of its w twins: dY = p.G + w (q.Y + t), everything
 *             downstream of dY (ReLU mask, dgrad, wgrad, the scatter-add) is linear in it.
 * Same mathematics as the uncompacted stack (only the summation order of the statistics changes).
 * The number of rows depends on the data: it lives on the device (pcops_rows_t.rows); the M / b*m*s arguments of the
 * *_rows entry points are the UNCOMPACTED row count, i.e. the upper bound the buffers and launches are sized with,
 * and batch-norm divisors (pcops_mlp_bn_finalize / _bn_bwd_coeffs R) stay the uncompacted count.
 * rows == NULL: exactly the entry point without the suffix.  Compacted rows need the wave-stream shapes
 * (>= 8192 rows, channel counts multiples of 8): PCOPS_ERR_UNSUPPORTED otherwise. */
/* TensorFlow-flavoured Adam on flat fp32 buffers of n floats (n % 4 == 0, 16-byte aligned), one launch:
 *   m <- beta1 m + (1 - beta1) g;  v <- beta2 v + (1 - beta2) g^2;  p <- p - lr_t m / (sqrt(v) + epsilon)
 * lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t) comes from the caller (tf.train.AdamOptimizer: epsilon outside the
 * bias-corrected root; reference trainers pointnet2/train.py:165-168). */
int pcops_adam_step(long long n, float *p, const float *g, float *m, float *v, float beta1, float beta2, float lr_t,
                    float epsilon, pcops_stream_t stream);
/* 1 when every launch of a grouped stack over compacted rows has a kernel for its shape (the *_rows entry points have
 * no tiled fallback): b clouds x m groups x s slots, n source points per cloud, has_q = the first layer has a feature
 * term (its gradient then walks the rows through the inverse index), widths [nlayers] = the layers' output widths
 * (HOST array).  pcops_sa_scatter_rows_supported: the first layer's part of that answer. */
int pcops_gather_stack_rows_supported(int b, int n, int m, int s, int has_q, int nlayers, const int *widths);
int pcops_sa_scatter_rows_supported(int n, int m, int s, int c);
unsigned long long pcops_rows_max_blocks(int b, int m, int s);
/* builds blocks / block_start / rows from pts_cnt (b,m) of pcops_query_ball_point; s % 16 == 0.
 * blocks: 16 * pcops_rows_max_blocks(b,m,s) bytes, block_start: b*m + 1 ints, rows: 1 int. */
int pcops_rows_plan(int b, int m, int s, const int *pts_cnt, void *blocks, int *block_start, int *rows,
                    pcops_stream_t stream);
int pcops_mlp_gemm_fwd_rows(int M, int K, int N, const float *X, int ldx, const float *pro_scale,
                            const float *pro_shift, const float *W, const float *bias, float *Y,
                            float *stats_partial, const float *stat_pivot, const pcops_rows_t *rows,
                            pcops_stream_t stream);
int pcops_mlp_gemm_fwd_xyz_rows(int M, int K, int N, const float *off4, const float *xyzw, const float *pro_scale,
                                const float *pro_shift, const float *W, const float *bias, float *Y,
                                float *stats_partial, const float *stat_pivot, const pcops_rows_t *rows,
                                pcops_stream_t stream);
int pcops_mlp_gemm_dgrad_rows(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                              const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                              int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                              const float *Yprev, const float *prev_scale, const float *prev_shift, float *Gprev,
                              float *stats_partial, const pcops_rows_t *rows, pcops_stream_t stream);
int pcops_mlp_gemm_dgrad_xyz_rows(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                                  const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                                  int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                                  const float *off4, const float *xyzw, const float *prev_scale,
                                  const float *prev_shift, float *Gprev, float *stats_partial, float *xyz_stats,
                                  const pcops_rows_t *rows, pcops_stream_t stream);
int pcops_mlp_wgrad_rows(long long M, int K, int N, const float *X, int ldx, const float *a_scale,
                         const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                         const float *t, const float *gpool, const unsigned char *argmax, int S,
                         const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                         const pcops_rows_t *rows, pcops_stream_t stream);
int pcops_mlp_wgrad_xyz_rows(long long M, int K, int N, const float *off4, const float *xyzw, const float *a_scale,
                             const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                             const float *t, const float *gpool, const unsigned char *argmax, int S,
                             const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                             const pcops_rows_t *rows, pcops_stream_t stream);
/* last layer of a max-pooled stack over compacted rows, pooling fused into the epilogue: ypart / ppart
 * [pcops_rows_max_blocks][N] receive, per 16-row block, the selected raw value (max for gamma >= 0, min otherwise)
 * and its row-in-group; pcops_mlp_pool_combine_rows picks per group -> out = relu(scale*ysel + shift), argmax, ysel */
int pcops_mlp_gemm_fwd_pool_rows_supported(int M, int K, int N);
int pcops_mlp_gemm_fwd_pool_rows(int M, int K, int N, const float *X, int ldx, const float *pro_scale,
                                 const float *pro_shift, const float *W, const float *bias, const float *gamma,
                                 float *Y, float *stats_partial, const float *stat_pivot, float *ypart,
                                 unsigned char *ppart, const pcops_rows_t *rows, pcops_stream_t stream);
int pcops_mlp_pool_combine_rows(long long G, int C, const float *ypart, const unsigned char *ppart,
                                const float *gamma, const float *scale, const float *shift,
                                const pcops_rows_t *rows, float *out, unsigned char *argmax, float *ysel,
                                pcops_stream_t stream);
/* out[g] = max over the rows of group g of relu(scale*Y + shift); argmax = row-in-group of the first maximiser */
int pcops_mlp_bn_relu_maxpool_rows(long long G, int C, const float *Y, const float *scale, const float *shift,
                                   const pcops_rows_t *rows, float *out, unsigned char *argmax, float *ysel,
                                   pcops_stream_t stream);
/* pcops_sa_gather_fwd / pcops_sa_scatter_bwd over compacted rows: Y, off4, G are (rows, c) / (rows, 4) in the
 * compacted order; idx stays the (b,m,s) tensor of the ball query */
int pcops_sa_gather_fwd_rows(int b, int n, int m, int s, int c, const float *Q, const float *Ctr, const float *xyz,
                             const float *new_xyz, const float *Wxyz, const float *bias, const int *idx, float *Y,
                             float *off4, float *stats_partial, const float *stat_pivot, float *moments,
                             const pcops_rows_t *rows, pcops_stream_t stream);
int pcops_sa_scatter_bwd_rows(int b, int n, int m, int s, int c, const float *G, const float *Y, const float *p,
                              const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                              const float *pool_scale, const float *pool_shift, const int *idx, const float *xyz,
                              const float *new_xyz, float *dQ, float *dCtr, float *wpartial, float *dWxyz,
                              float *dbias, const float *fwd_Q, const float *fwd_Ctr, const float *fwd_Wxyz,
                              const float *fwd_bias, void *workspace, const pcops_rows_t *rows,
                              pcops_stream_t stream);

/* ---- the [Q | Ctr] forms of the five entry points above (csrc/edgeconv.hip).  EdgeConv's first conv is linear in the
 * edge feature, concat(x_i, x_j - x_i) W = x_j W_b + x_i (W_a - W_b) (dgcnn/utils/tf_util.py:699-705 + dgcnn.py:39-44):
 * Q = X W_b and Ctr = X (W_a - W_b) + bias are the column halves of ONE product X [W_b | W_a - W_b] with row stride 2 c,
 * and dQ / dCtr the halves of one gradient, so the per-point GEMM, its weight gradient and its data gradient run once
 * instead of twice (and autograd has nothing to add up).  ldq / ldc / lddq / lddc: row strides in floats (multiples of 4,
 * >= c) of Q / Ctr / dQ / dCtr; everything else as in the dense entry points; n == m.  Only shapes
 * pcops_edge_ld_supported() accepts have kernels (64-channel slices, whole 64-group chunks, LDS-resident clouds);
 * PCOPS_ERR_UNSUPPORTED otherwise.  In deterministic mode the QUERY answers 0 (the backward's arg-row sums are LDS float
 * atomics; keep the dense entry points and their ordered kernels); the entry points themselves check the shape only, so a
 * backward whose forward chose this family still runs when the switch is thrown in between.  The sa_* pair is the Q + Ctr form of
 * pcops_sa_gather_fwd / pcops_sa_scatter_bwd (no coordinate term, G materialised, dQ and dCtr both produced). */
int pcops_edge_ld_supported(int b, int n, int m, int s, int c);
int pcops_edge_pool_fwd_ld(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc,
                           const int *idx, const float *gamma, float *SQ, float *qsel, unsigned char *arg,
                           float *stats_partial, const float *stat_pivot, pcops_stream_t stream);
int pcops_edge_pool_out_ld(long long groups, int c, const float *qsel, const float *Ctr, int ldc, const float *scale,
                           const float *shift, float *out, float *ysel, pcops_stream_t stream);
/* ... and a SECOND copy of `out` into a column block of a wider row-major tensor (out2 + g * ld2): DGCNN concatenates the
 * four EdgeConv outputs (dgcnn/models/dgcnn.py:83), so each layer stores its block of the (b, n, 320) tensor on the way
 * out and no concatenation pass (671 MB read + written) runs.  out2 may be NULL (= pcops_edge_pool_out_ld). */
int pcops_edge_pool_out_ld2(long long groups, int c, const float *qsel, const float *Ctr, int ldc, const float *scale,
                            const float *shift, float *out, float *ysel, float *out2, int ld2, pcops_stream_t stream);
int pcops_edge_pool_bwd_ld(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc,
                           const int *idx, const float *gpool, const float *ysel, const float *SQ,
                           const unsigned char *arg, const float *scale, const float *shift, const float *p,
                           const float *q, const float *t, float *dQ, int lddq, float *dCtr, int lddc, void *workspace,
                           pcops_stream_t stream);
int pcops_sa_gather_fwd_ld(int b, int n, int m, int s, int c, const float *Q, int ldq, const float *Ctr, int ldc,
                           const int *idx, float *Y, float *stats_partial, const float *stat_pivot,
                           pcops_stream_t stream);
int pcops_sa_scatter_bwd_ld(int b, int n, int m, int s, int c, const float *G, const float *p, const float *q,
                            const float *t, const int *idx, const float *Q, int ldq, const float *Ctr, int ldc,
                            float *dQ, int lddq, float *dCtr, int lddc, void *workspace, pcops_stream_t stream);

/* the concatenated weight of the [Q | Ctr] form: W1 (2 c, cp) = the reference's EdgeConv kernel [W_a ; W_b] (rows 0..c-1
 * multiply x_i, rows c..2c-1 multiply x_j - x_i), b1 (cp) or NULL -> Wcat (kp, 2 cp) = [W_b | W_a - W_b] (rows c..kp-1
 * zero: the input zero-padded to kp channels), bcat (2 cp) = [0 | b1]; and the backward map dWcat, dbcat -> dW1, db1. */
int pcops_edge_weights_fwd(int c, int cp, int kp, const float *W1, const float *b1, float *Wcat, float *bcat,
                           pcops_stream_t stream);
int pcops_edge_weights_bwd(int c, int cp, const float *dWcat, const float *dbcat, float *dW1, float *db1,
                           pcops_stream_t stream);

/* ------------------------------------------------------------- classifier / T-Net heads (csrc/head.hip)
 * BatchNorm (+ ReLU) of a fully connected layer's output over R rows (the batch) as one launch per direction:
 * fully_connected(..., bn=True) of pointnet2/utils/tf_util.py:327-363 (batch_norm_for_fc :534-546) and
 * dgcnn/utils/tf_util.py:317-354 (batch_norm_template :462-499).  x, y, dy, dx (R, C) dense.
 *   training != 0: batch mean and BIASED variance normalise; moving_mean / moving_var are updated in place,
 *                  m <- decay m + (1 - decay) batch, the variance fed being var R / (R - 1) when unbiased_moving_var
 *                  (the pointnet2 flavour) and var otherwise (the DGCNN flavour);
 *   training == 0: the moving statistics normalise and nothing is updated.
 * save_mean / save_rstd (C) receive the statistics used (the backward's inputs).  relu != 0: y = max(., 0) and the
 * backward masks dy with y > 0.  Backward: dgamma = sum g xhat, dbeta = sum g, dx = gamma rstd (g - dbeta / R - xhat
 * dgamma / R), or gamma rstd g through frozen statistics (training == 0). */
int pcops_fc_bn_fwd(int R, int C, const float *x, const float *gamma, const float *beta, float *moving_mean,
                    float *moving_var, int training, float decay, float eps, int unbiased_moving_var, int relu,
                    float *y, float *save_mean, float *save_rstd, pcops_stream_t stream);
int pcops_fc_bn_bwd(int R, int C, const float *dy, const float *x, const float *y, const float *gamma,
                    const float *save_mean, const float *save_rstd, int training, int relu, float *dx,
                    float *dgamma, float *dbeta, pcops_stream_t stream);

/* The classification loss of a batch and its gradient in one launch: mean softmax cross entropy against
 * q = s / C + (1 - s) onehot(labels) -- tf.losses.softmax_cross_entropy(..., label_smoothing = s) of dgcnn/models/dgcnn.py:99-105;
 * s = 0 is the sparse softmax cross entropy + reduce_mean of pointnet2/models/pointnet2_cls_ssg.py:47-53 and, over b n rows of
 * two classes, the per-point mask loss of pointnet2_cls_bga.py:94-98 (equal point counts: the mean of the clouds' means is the
 * mean of all rows).  dlogits [R][C] = (softmax - q) / R (the gradient for an upstream factor of 1; the caller scales).
 * loss [pcops_softmax_ce_blocks(R)]: one workgroup up to 4096 rows -- loss[0] is the loss --, several beyond, each leaving
 * its rows' share of the mean (the caller adds them up).  Fixed summation order. */
int pcops_softmax_ce_blocks(int R);
int pcops_softmax_ce(int R, int C, const float *logits, const int *labels, float label_smoothing, float *loss,
                     float *dlogits, pcops_stream_t stream);
/* The interpolation weights of pointnet_fp_module (pointnet_util.py:212-215) from three_nn's squared distances:
 * weight (b, n, 3) = inv / sum(inv), inv = 1 / max(dist, 1e-10); dist = +inf (fewer than three known points) -> 0. */
int pcops_three_nn_weights(int b, int n, const float *dist, float *weight, pcops_stream_t stream);

/* the learned 3 x 3 input transform applied to a cloud (tf.matmul(point_cloud, transform): dgcnn/models/dgcnn.py:37,
 * pointnet/models/pointnet_cls.py:27): out (b, n, 3) = x (b, n, 3) T (b, 3, 3); backward dT (b, 3, 3) = x^T grad_out per cloud in a
 * fixed order, dx = grad_out T^T (dx may be NULL: the input cloud needs no gradient). */
int pcops_transform3_fwd(int b, int n, const float *x, const float *T, float *out, pcops_stream_t stream);
int pcops_transform3_bwd(int b, int n, const float *x, const float *T, const float *grad_out, float *dT, float *dx,
                         pcops_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCOPS_H */
