// mlp.hip -- the shared per-point MLP (1x1 conv + bias + BatchNorm + ReLU [+ max-pool over the
// neighbourhood]) of PointNet++ set abstraction / feature propagation and DGCNN EdgeConv, forward
// and backward, as fused fp32-MFMA kernels for gfx950.
//
// Reference call sites (TensorFlow ops, no native code of its own): pointnet2/utils/pointnet_util.py:
// 117-127 (conv2d stack + reduce_max), :223-227 (FP stack); pointnet2/utils/tf_util.py:120-185
// (conv2d = tf.nn.conv2d + bias_add + batch_norm + relu), :512-531 (BN); dgcnn/models/dgcnn.py:39-48.
// Every activation tensor there makes a full HBM round trip per op; here one layer = ONE pass:
//
//   fwd   Y_l = relu(bn_{l-1}(Y_{l-1})) W_l + b_l        BN+ReLU of the PREVIOUS layer is applied while
//                                                        the A tile is staged (prologue); the epilogue
//                                                        emits per-channel sum / sum-of-squares partials
//   dgrad G_{l-1} = relu'_{l-1} . (dY_l W_l^T)           dY_l = p.G_l + q.Y_l + t (BN backward folded into
//                                                        three per-channel vectors) computed in the
//                                                        prologue; ReLU mask + the two BN-backward
//                                                        reductions of layer l-1 in the epilogue
//   wgrad dW_l = A_{l-1}^T dY_l, db_l = 1^T dY_l         both operands rebuilt in registers from the raw
//                                                        tensors, fragment-shaped straight from HBM
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, k ascending within a chunk) for the data and weight
// gradients and the small-row kernels; the FORWARD products of the wave-stream kernel run on the bf16 matrix pipe with
// every fp32 operand split into three bf16 pieces (six exact partial products, the large ones accumulated apart from the
// small ones: a third of the fp32 chain's rounding error; PCOPS_GEMM_BF3=0 takes the fp32 pipe) --
// no reduced-precision RESULT anywhere.  Tiles: 128 rows x (64|128) cols per workgroup of 4 waves, K in chunks
// of 32 staged through LDS with 16-byte reads; the A fragment uses the "label permutation" trick: a
// lane reads 4 consecutive k of its row with ONE ds_read_b128 and feeds them to 4 MFMAs whose k labels
// are matched on the B side, so no transpose is ever needed.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

// ---- build parts.  This file is compiled SEVEN times (Makefile: -DPCOPS_MLP_PART=0..6, in parallel): every part parses the
// whole file, but only emits its share of the kernel instantiations -- part 0 the C ABI and the small kernels, parts
// 1, 2, 3, 6 the wave-stream / tiled GEMM launchers by (operand, epilogue) mode, part 4 the weight-gradient kernels,
// part 5 the one-pass backward and the Gram kernel.  The launchers are the only symbols that cross parts (hidden
// visibility: none of them is part of the C ABI).  Without the macro (tools/ builds) everything is one translation unit.
#ifndef PCOPS_MLP_PART
#define PCOPS_MLP_PART (-1)
#endif
#define PCOPS_PART(p_) (PCOPS_MLP_PART < 0 || PCOPS_MLP_PART == (p_))
#define PCOPS_HIDDEN __attribute__((visibility("hidden")))

#ifndef PCOPS_BF_RM
#define PCOPS_BF_RM 1      // one-pass backward with the split-operand dW half, EdgeConv (SIDE) form: dY's pieces row-major as well (bwd_fused_kernel, RM)
#endif

namespace pcops_mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 -> three bf16 pieces, x = h + m + l up to 2^-25 |x| (each residual is exact in fp32; round-to-nearest pieces carry
// their own signs, so 3 x 8 significant bits cover the 24 of the operand; below 2^-100 the residuals turn denormal and the
// identity holds to an absolute 2^-133 instead).  Products of two pieces are exact in fp32.  (tests/test_split_operands_cpu.py)
// Written on PAIRS of values (round 6): one v_cvt_pk_bf16_f32 per pair and piece, the widening as shift / mask of the packed word,
// the residuals as v_pk_add_f32 -- 36 instructions for eight values.  As a loop over single values hipcc paired what it could and
// left 46 (17 conversions, ten of the sixteen subtractions unpacked); same arithmetic, bit for bit.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float a, float b, bf16x2 &h, bf16x2 &m, bf16x2 &l) {
    const f32x2 v = {a, b};
    h = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    l = __builtin_convertvector(r2, bf16x2);
}
__device__ __forceinline__ void split3(const float4 &x0, const float4 &x1, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        bf16x2 hh, mm, ll;
        split3_pair(x[2 * p], x[2 * p + 1], hh, mm, ll);
        h[2 * p] = hh.x; h[2 * p + 1] = hh.y;
        m[2 * p] = mm.x; m[2 * p + 1] = mm.y;
        l[2 * p] = ll.x; l[2 * p + 1] = ll.y;
    }
}

constexpr int kBM = 128;   // rows per tile (4 waves x 32)
constexpr int kBK = 32;    // K chunk
constexpr int kLdA = 36;   // A tile row stride in floats (16 B aligned, conflict-free b128 reads)

enum AMode { A_PLAIN = 0, A_BNRELU = 1, A_DY = 2, A_DYPOOL = 3, A_DYPOOLU = 4 };
// A_DYPOOLU: A_DYPOOL with S % 32 == 0 -- a 32-row tile / stripe lies inside ONE pooling group, so its gpool / arg-max
// quad is loaded once per tile and the row-in-group is (row0 % S) + r (wave-stream kernels only)
// A_DYPOOLB: the pooled form over COMPACTED rows (pcops_rows_t, see pcops.h): rows come in blocks of kBlk = 16 that lie
// inside one pooling group each, so a 32-row tile / stripe needs two (gpool, arg-max) quads, one per block, and the
// row-in-group is block.s0 + (r % 16); the block table replaces the division by S
constexpr int A_DYPOOLB = 6;
// A_DYW (weight-gradient kernel only): A_DY over compacted rows, i.e. with the block weights.  A separate mode so
// that the uncompacted instantiations carry none of it (the 64 x 64 tile lives on a 128-register budget)
constexpr int A_DYW = 7;
constexpr bool is_pool(int am) { return am == A_DYPOOL || am == A_DYPOOLU || am == A_DYPOOLB; }
constexpr int kBlk = 16;        // rows per block of a compacted row set
struct RowBlock { int g, s0; float w; int pad; };   // group, row-in-group of the block's first row, weight of that row
// A block record read at a WAVE-UNIFORM index, through the constant address space: one s_load_dwordx4 on the scalar unit.  As
// a plain global read hipcc issues a vector load (the table may alias the kernel's stores for all it knows), and the
// (pooled gradient, arg-max) row reads that depend on it then sit behind an s_waitcnt vmcnt: two exposed L2 round trips in
// every chunk's issue(), in front of the stripe prefetch.  The table is written by an earlier kernel (rows_plan_fill_kernel).
__device__ __forceinline__ RowBlock uniform_block(const RowBlock *table, long long i) {
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) i32x4_ *kptr;
    const i32x4_ v = *(kptr)(table + i);
    RowBlock r;
    r.g = v.x; r.s0 = v.y; r.w = __int_as_float(v.z); r.pad = v.w;
    return r;
}
// A_XYZ: the operand is the BN+ReLU of a first layer that is ARITHMETIC in three per-row offsets,
//   y[row][k] = fma(dz, w2[k], fma(dy, w1[k], fma(dx, w0[k], b[k])))      (csrc/gather.hip first_layer_quad order)
// rebuilt from off4[row] = (dx, dy, dz, 0) instead of being read: 16 bytes per row instead of 4 K (wave-stream only)
constexpr int A_XYZ = 5;
// weight-gradient kernels only: the "dY" side is the A operand itself, relu(bn(X)) -- the Gram matrix X^T X of a layer's
// input (pcops_mlp_gram), which is what the algebraic form of a pooled top layer's weight gradient is built from
constexpr int A_SELFD = 8;
constexpr bool is_dy(int am) { return am == A_DY || is_pool(am); }
__device__ __forceinline__ float xyz_y(float4 o, float w0, float w1, float w2, float b) {
    return fmaf(o.z, w2, fmaf(o.y, w1, fmaf(o.x, w0, b)));
}

// ---- buffer-resource addressing (gfx950): ONE 32-bit VGPR offset per lane + a scalar offset per access, and the
// hardware bounds check (offset >= num_records -> loads return 0, stores are dropped) replaces every row guard.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long long bytes) {
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    // (aux 2 = non-temporal on the streamed operand reads: SSG 23.05 vs 23.05 k, DGCNN 9.85 vs 9.85 k -- nothing; stores: see buf_store4)
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// nt: non-temporal (aux bit 1 on gfx94x / gfx950): the output is a stream far larger than the L2 and the memory-side cache
// that nothing reads back soon -- not allocating its lines measured 7 % on a pure 2.7 GB write stream (edgeconv.hip MODE 1)
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 x, bool nt = false) {
    u32x4 v;
    v.x = __float_as_uint(x.x); v.y = __float_as_uint(x.y); v.z = __float_as_uint(x.z); v.w = __float_as_uint(x.w);
    if (nt) __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    // gfx950 store-data hazard hipcc does not pad: a VALU write to the data registers of a 16-byte buffer store in the
    // very next issue slot still reaches the store (observed: the .w word of the last four lanes of every 16-lane group
    // took the NEXT row's value whenever a v_pk_add_f32 rewrote v[n+2:n+3] right behind the store).  LLVM's recogniser
    // only pads MUBUF stores WITHOUT an SGPR soffset; these use one.  The statement below reads the four registers, so
    // nothing can be allocated into them before it, and it carries the wait states itself.  (Round 2 met the same
    // corruption when it unrolled the MFMA loop and put it down to a code-generation accident.)
    asm volatile("s_nop 1" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_u32(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
constexpr unsigned kOOB = 0x7FFFFFF0u;   // a voffset no tensor reaches: forces the bounds check to fail
enum EMode { E_FWD = 0, E_MASK = 1, E_PLAIN = 2, E_MASKX = 3, E_MASKA = 4 };
// E_MASKX: E_MASK with the previous layer in xyz form
// E_MASKA: E_MASK of  acc + addend[rowmap[row]] + vconst  -- the data gradient of a POOLED top layer in its algebraic
//          form (pcops_mlp_gemm_dgrad_top): acc = relu(bn(Yprev)) . (W diag(q) W^T), the few arg-max rows come in
//          through the compact addend, the constant row W (q.b + t) through vconst
constexpr int E_PLAINA = 5;      // E_PLAIN of acc + addend[rowmap[row]] + vconst (the same gradient w.r.t. a stack's raw input)
constexpr bool is_mask(int em) { return em == E_MASK || em == E_MASKX || em == E_MASKA; }
constexpr bool has_add(int em) { return em == E_MASKA || em == E_PLAINA; }

// -DPCOPS_PHASE_PROF (tools/ only, never the shipped build): every wave of gemm_ws_kernel adds the shader cycles it spent
// staging / in the MFMA loop / in the epilogue to these counters; pcops_debug_phase_prof reads and clears them
#ifdef PCOPS_PHASE_PROF
__device__ unsigned long long g_phase_prof[8];
#define PROF_T() __builtin_amdgcn_s_memtime()
#endif

struct GemmArgs {
    int M, K, N;
    int tiles_per_block;
    // A operand
    const float *X;      // A_PLAIN/A_BNRELU: input rows;  A_DY: G (masked upstream grad);  A_DYPOOL: unused
    int ldx;
    const float *X2;     // A_DY/A_DYPOOL: raw Y of this layer (same shape/ld as X)
    const float *v0;     // A_BNRELU: scale[k];  A_DY*: p[k]
    const float *v1;     // A_BNRELU: shift[k];  A_DY*: q[k]
    const float *v2;     //                      A_DY*: t[k]
    const float *v3;     // A_DYPOOL: bn scale[k] of this layer (for the relu mask at the argmax)
    const float *v4;     // A_DYPOOL: bn shift[k]
    const float *gpool;  // A_DYPOOL: upstream grad of the pooled output [M/S][K]
    const unsigned char *argmax;  // A_DYPOOL: [M/S][K]
    int S;
    // B operand: row-major [K][N]
    const float *W;
    // epilogue
    const float *bias;   // E_FWD (may be null)
    float *Y;            // output [M][N]
    int ldy;
    const float *off4;   // A_XYZ / E_MASKX: per-row offsets [M][4]
    const float *xw;     // A_XYZ / E_MASKX: [4][xw_ld] rows w0, w1, w2, b of the arithmetic first layer
    int xw_ld;
    const float *Yprev;  // E_MASK: raw Y of the previous layer [M][N]
    const float *msc;    // E_MASK: bn scale / shift of the previous layer
    const float *msh;
    float *stats;        // [gridDim.y][2][N] partial column sums (null: none)
    const float *pivot;  // E_FWD: [N] per-channel pivot the forward statistics are taken relative to -- sums of
                         // (y - pivot) and (y - pivot)^2 (null: 0).  Any estimate of the channel mean within a few
                         // standard deviations (the BN layer's moving mean) removes the cancellation of the one-pass
                         // variance  E[y^2] - mean^2  in fp32; pcops_mlp_bn_finalize undoes the shift exactly
    int nrowgrp;         // wave-stream kernel: row groups that own tiles; workgroups beyond only zero their statistics row
    float *xstats;       // E_MASKX: [rows of stats][3][N] partial sums of off (x) Gprev -- with them the arithmetic first
                         // layer's weight gradient needs neither Gprev nor a pass over it (Y may then be NULL)
    // fused neighbourhood pooling of the RAW outputs (forward, wave-stream kernel only): BN+ReLU is monotone per
    // channel -- increasing for gamma >= 0, decreasing otherwise (scale = gamma * rstd) -- so per group of
    // 32*pool_sub rows the pooled activation is relu(scale * ysel + shift) with ysel the max (gamma >= 0) or min
    // (gamma < 0) of y over the group; psel gets the first row attaining it
    int pool_sub;
    // round 5: groups of pool_s4 rows with pool_s4 % 4 == 0 and NOT a multiple of 32 (DGCNN's T-Net: 20 neighbours): a wave
    // walks 32 pool_sub = lcm(pool_s4, 32) rows = 32 pool_sub / pool_s4 WHOLE groups; pool_inv = 65536 / pool_s4 + 1
    // (row / pool_s4 as a multiplication, exact for the rows of one walk: checked by the launcher).  0: one group per walk
    int pool_s4, pool_inv;
    int nt_out;                         // Y leaves with non-temporal stores (set by the launchers: outputs of 256 MB and more)
    const float *pgamma;                // [N]
    float *ysel;                        // [M / (32 pool_sub)][N]
    unsigned char *psel;                // [M / (32 pool_sub)][N]
    // compacted rows (wave-stream kernels only): the row count lives on the device (M above is the upper bound the
    // launch is sized with), blocks[i] describes rows 16 i .. 16 i + 15.  The first row of a group stands for w rows of
    // the uncompacted tensor: statistics weigh it with w, and its dY is p.G + w (q.Y + t)
    const RowBlock *blocks;
    const int *Mdev;
    // E_MASKA
    const float *addend;     // [slots][add_ld] compact rows added before the mask
    const int *rowmap;       // [M] slot of a row, or -1
    int add_ld;
    long long add_bytes;     // size of addend (< kOOB: a slot of -1 becomes an offset the bounds check rejects)
    const float *vconst;     // [N] added to every row
};

__device__ __forceinline__ float4 ld4(const float *p, bool vec, int k, int K) {
    // 4 consecutive floats starting at column k of a row (zero beyond K); vec: 16-byte aligned fast path
    if (vec) return (k < K) ? *reinterpret_cast<const float4 *>(p + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r;
    r.x = k + 0 < K ? p[k + 0] : 0.f;
    r.y = k + 1 < K ? p[k + 1] : 0.f;
    r.z = k + 2 < K ? p[k + 2] : 0.f;
    r.w = k + 3 < K ? p[k + 3] : 0.f;
    return r;
}

template <int NT, int AM, int EM>
__global__ __launch_bounds__(256) void gemm_rt_kernel(GemmArgs a) {
    constexpr int BN = NT * 32;
    __shared__ __attribute__((aligned(16))) float As[kBM * kLdA];
    __shared__ __attribute__((aligned(16))) float Bs[kBK * BN];
    __shared__ float red[4][2][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BN;
    const int M = a.M, K = a.K, N = a.N;
    const bool xvec = (a.ldx % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0) &&
                      (AM < A_DY || (reinterpret_cast<uintptr_t>(a.X2) & 15) == 0);
    const bool wvec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.W) & 15) == 0);

    // staging coordinates (fixed per thread)
    const int a_kq = (tid & 7) * 4;    // k offset inside the chunk
    const int a_r = tid >> 3;          // row 0..31 (+32 per pass)
    const int b_nq = (tid % (BN / 4)) * 4;
    const int b_k = tid / (BN / 4);    // 0..(1024/BN - 1), + (256*4/BN) per pass
    constexpr int B_PASS = (kBK * BN / 4) / 256;   // float4 per thread: 2 (BN=64) or 4 (BN=128)
    constexpr int B_KSTEP = 256 / (BN / 4);

    float s1[NT], s2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) s1[i] = s2[i] = 0.f;

    const int tile0 = blockIdx.y * a.tiles_per_block;
    for (int tt = 0; tt < a.tiles_per_block; ++tt) {
        const long long row0 = (long long)(tile0 + tt) * kBM;
        if (row0 >= M) break;

        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;

        float4 ra[4], rb[B_PASS];
        auto load_chunk = [&](int kc) {
            const int k = kc * kBK + a_kq;
            float4 c0, c1, c2, c3, c4;
            if (AM != A_PLAIN) {
                c0 = ld4(a.v0, true, k, (K + 3) & ~3);
                c1 = ld4(a.v1, true, k, (K + 3) & ~3);
                if (AM >= A_DY) c2 = ld4(a.v2, true, k, (K + 3) & ~3);
                if (AM == A_DYPOOL) {
                    c3 = ld4(a.v3, true, k, (K + 3) & ~3);
                    c4 = ld4(a.v4, true, k, (K + 3) & ~3);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const long long r = row0 + a_r + 32 * p;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < M) {
                    if (AM == A_PLAIN) {
                        x = ld4(a.X + r * a.ldx, xvec, k, K);
                    } else if (AM == A_BNRELU) {
                        x = ld4(a.X + r * a.ldx, xvec, k, K);
                        x.x = fmaxf(fmaf(x.x, c0.x, c1.x), 0.f);
                        x.y = fmaxf(fmaf(x.y, c0.y, c1.y), 0.f);
                        x.z = fmaxf(fmaf(x.z, c0.z, c1.z), 0.f);
                        x.w = fmaxf(fmaf(x.w, c0.w, c1.w), 0.f);
                    } else if (AM == A_DY) {
                        const float4 g = ld4(a.X + r * a.ldx, xvec, k, K);
                        const float4 y = ld4(a.X2 + r * a.ldx, xvec, k, K);
                        x.x = fmaf(c0.x, g.x, fmaf(c1.x, y.x, c2.x));
                        x.y = fmaf(c0.y, g.y, fmaf(c1.y, y.y, c2.y));
                        x.z = fmaf(c0.z, g.z, fmaf(c1.z, y.z, c2.z));
                        x.w = fmaf(c0.w, g.w, fmaf(c1.w, y.w, c2.w));
                    } else {  // A_DYPOOL: G is non-zero only at the arg-max row of each (group, channel)
                        const float4 y = ld4(a.X2 + r * a.ldx, xvec, k, K);
                        const long long g = (long long)((unsigned)r / (unsigned)a.S);   // rows < 2^31 (launcher)
                        const int s = (int)((unsigned)r - (unsigned)g * (unsigned)a.S);
                        const float yy[4] = {y.x, y.y, y.z, y.w};
                        const float cc0[4] = {c0.x, c0.y, c0.z, c0.w}, cc1[4] = {c1.x, c1.y, c1.z, c1.w};
                        const float cc2[4] = {c2.x, c2.y, c2.z, c2.w}, cc3[4] = {c3.x, c3.y, c3.z, c3.w};
                        const float cc4[4] = {c4.x, c4.y, c4.z, c4.w};
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float gm = 0.f;
                            if (k + e < K && a.argmax[g * K + k + e] == s && fmaf(yy[e], cc3[e], cc4[e]) > 0.f)
                                gm = a.gpool[g * K + k + e];
                            o[e] = (k + e < K) ? fmaf(cc0[e], gm, fmaf(cc1[e], yy[e], cc2[e])) : 0.f;
                        }
                        x = make_float4(o[0], o[1], o[2], o[3]);
                    }
                    if (AM >= A_DY) {  // rows are exact zeros beyond K (t[k] would leak otherwise)
                        if (k + 0 >= K) x.x = 0.f;
                        if (k + 1 >= K) x.y = 0.f;
                        if (k + 2 >= K) x.z = 0.f;
                        if (k + 3 >= K) x.w = 0.f;
                    }
                }
                ra[p] = x;
            }
#pragma unroll
            for (int p = 0; p < B_PASS; ++p) {
                const int kk = kc * kBK + b_k + B_KSTEP * p;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < K) w = ld4(a.W + (long long)kk * N, wvec, n0 + b_nq, N);
                rb[p] = w;
            }
        };

        const int nchunk = (K + kBK - 1) / kBK;
        load_chunk(0);
        for (int kc = 0; kc < nchunk; ++kc) {
            __syncthreads();  // previous chunk's MFMA reads are done
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<float4 *>(&As[(a_r + 32 * p) * kLdA + a_kq]) = ra[p];
#pragma unroll
            for (int p = 0; p < B_PASS; ++p)
                *reinterpret_cast<float4 *>(&Bs[(b_k + B_KSTEP * p) * BN + b_nq]) = rb[p];
            __syncthreads();
            if (kc + 1 < nchunk) load_chunk(kc + 1);  // global loads fly under the MFMAs below

            const float *arow = &As[(wave * 32 + (lane & 31)) * kLdA + 4 * (lane >> 5)];
            const float *bcol = &Bs[(4 * (lane >> 5)) * BN + (lane & 31)];
            // BLOCKED summation over K (round 4): every 32-wide K chunk is summed into its own accumulator (a 32-term
            // fmaf chain) and the chunk sums are added up -- the rounding error of a dot product then grows like
            // sqrt(32 + K / 32) instead of sqrt(K) of one K-long chain (K = 512: 3.3x smaller).  These are the small-row
            // layers (group_all stacks, FP stacks, heads of small batches) whose output feeds a batch norm over a few
            // hundred rows, where a library GEMM with its K split was measurably more accurate than the single chain
            // (tools/diag_stage_noise.py: +8 % RMS error from layer3 on); the extra 16 NT adds per chunk are free here.
            f32x16 part[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) part[i][v] = 0.f;
#pragma unroll
            for (int it = 0; it < kBK / 8; ++it) {
                const float4 av = *reinterpret_cast<const float4 *>(arow + 8 * it);
                const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float bv = bcol[(8 * it + t) * BN + 32 * nt];
                        part[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[t], bv, part[nt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][v] += part[i][v];
        }

        // ------------------------------------------------------------------ epilogue
        // acc[nt][v]: col = n0 + 32 nt + (lane & 31), row = row0 + 32 wave + (v&3) + 8 (v>>2) + 4 (lane>>5)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + 32 * nt + (lane & 31);
            const bool ncol = n < N;
            float bias = 0.f, msc = 0.f, msh = 0.f, pv = 0.f;
            if (EM == E_FWD && a.bias && ncol) bias = a.bias[n];
            if (EM == E_FWD && a.pivot && ncol) pv = a.pivot[n];
            if (EM == E_MASK && ncol) { msc = a.msc[n]; msh = a.msh[n]; }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = row0 + 32 * wave + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                if (r < M && ncol) {
                    float o = acc[nt][v];
                    if (EM == E_FWD) {
                        o += bias;
                        const float d = o - pv;              // statistics relative to the pivot (GemmArgs::pivot)
                        s1[nt] += d;
                        s2[nt] = fmaf(d, d, s2[nt]);
                    } else if (EM == E_MASK) {
                        const float yp = a.Yprev[r * a.ldy + n];
                        o = fmaf(yp, msc, msh) > 0.f ? o : 0.f;
                        s1[nt] += o;
                        s2[nt] = fmaf(o, yp, s2[nt]);
                    }
                    a.Y[r * a.ldy + n] = o;
                }
            }
        }
    }

    if (EM != E_PLAIN && a.stats) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s1[nt] += __shfl_xor(s1[nt], 32, 64);
            s2[nt] += __shfl_xor(s2[nt], 32, 64);
            if (lane < 32) {
                red[wave][0][32 * nt + lane] = s1[nt];
                red[wave][1][32 * nt + lane] = s2[nt];
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            if (n0 + c < N) {
                const float v = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
                a.stats[((long long)blockIdx.y * 2 + which) * N + n0 + c] = v;
            }
        }
    }
}

template <int AM, int EM>
PCOPS_HIDDEN int launch_gemm_rt(GemmArgs &a, hipStream_t st) {
    const int tiles = (a.M + kBM - 1) / kBM;
    // enough row-tile groups to fill the chip a few times over, few enough that the partial-statistics
    // buffer stays small
    int tpb = 1;
    while (tiles / tpb > 512) tpb *= 2;
    a.tiles_per_block = tpb;
    const int gy = (tiles + tpb - 1) / tpb;
    if (a.N <= 64) {
        hipLaunchKernelGGL((gemm_rt_kernel<2, AM, EM>), dim3((a.N + 63) / 64, gy), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((gemm_rt_kernel<4, AM, EM>), dim3((a.N + 127) / 128, gy), dim3(256), 0, st, a);
    }
    return pcops_launch_status();
}

// ---------------------------------------------------------------------------------------------
// gemm_ws: "wave-stream" variant of the same GEMM for the big-row / small-K layers (K <= 256 staged in
// stripes of <= 128, N tile 64|128): the weight tile and every per-channel coefficient vector stay RESIDENT
// in LDS for the life of a persistent workgroup, and every wave streams its own 32-row tiles independently:
//   * all global reads of a tile (operand stripes, the mask tensor of the epilogue, pooled-gradient side
//     inputs) are issued one stage AHEAD as coalesced 16-byte loads into registers, with clamped addresses
//     instead of branches; nothing is loaded right before it is needed, so the only vmcnt waits are for data
//     requested a whole tile earlier and the epilogue's stores never have to drain (gfx950 counts loads and
//     stores on one vmcnt);
//   * the operand transform is applied on the way into a wave-private LDS stripe; fragments come back with
//     ds_read_b128 (label-permutation trick, see header);
//   * the accumulator tile is transposed through the same stripe so that outputs leave as 16-byte stores of
//     whole 512-byte row segments, and the per-channel statistics accumulate in the lane that owns the column;
//   * the main loop contains NO workgroup barrier.
// (group, row-in-group) of row row0 + r for r < 64 without a per-element integer division: one division per
// tile for row0 (wave-uniform), then t = s0 + r < S + 64 splits exactly through a float reciprocal
// ((t + 0.5) / S is at least 1/(2S) >= 1/512 away from an integer; fp32 rounding of a value < 2^10 is far below)
struct PoolRows {
    long long g0;
    int s0, S;
    float invS;
    __device__ PoolRows(long long row0, int S_) : S(S_) {
        // row indices fit 32 bits (M is an int at the ABI): a 32-bit division is ~25 scalar instructions, the 64-bit one
        // this replaced ~100 -- twice per stripe and wave it was most of the 80 M SALU instructions of the SA1 wgrad
        const unsigned q = (unsigned)row0 / (unsigned)S_;
        g0 = (long long)q;
        s0 = (int)((unsigned)row0 - q * (unsigned)S_);
        invS = 1.0f / (float)S_;
    }
    // ... from values the caller keeps up to date itself (persistent loops: no division per stripe)
    __device__ PoolRows(long long g0_, int s0_, int S_) : g0(g0_), s0(s0_), S(S_), invS(1.0f / (float)S_) {}
    __device__ void split(int r, long long glast, long long &g, unsigned &s) const {
        const int t = s0 + r;
        const int dg = (int)(((float)t + 0.5f) * invS);
        s = (unsigned)(t - dg * S);
        g = g0 + dg;
        g = g < glast ? g : glast;            // rows beyond M (masked by the caller) stay inside gpool / argmax
    }
};

template <int NT, int AM, int EM, int KC, int WAVES, int EH, int VAR>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void gemm_ws_kernel(GemmArgs a) {
    wave_prio_stagger();
    // WAVES waves per workgroup (4: one per SIMD, 8: two per SIMD so that one wave's staging / epilogue hides under
    // its partner's MFMA phase); EH: the epilogue transposes the accumulator tile in EH column passes so that the
    // wave stripe only needs max(KC, BN/EH) columns.
    // VAR 0: weights resident in LDS for the life of the workgroup (K <= 256)
    //     1: + neighbourhood pooling fused into the forward epilogue
    //     2: weights STREAMED -- K chunks of KC rows double-buffered in LDS, refilled from L2 by the whole workgroup
    //        while the MFMAs of the current chunk run; all waves then walk the chunks in lockstep (one workgroup
    //        barrier per chunk), which is what K > 256 costs
    //  +4: the product on the 16-bit matrix pipe with SPLIT operands: each fp32 operand as three
    //      bf16 pieces, six of the nine partial products on v_mfma_f32_32x32x16_bf16 (6/16 of the fp32 pipe time), the
    //      h.h products in the tile's accumulators and the five small ones in a second set that is added once per tile --
    //      the large accumulator is rounded K/16 times instead of K/2, a third of the fp32 chain's error (measured,
    //      tools/ubench/gemm_bf16x3.hip).  The weight pieces sit in LDS in fragment order (6 bytes per element), the
    //      operand stripe stays fp32 and is split in registers on the way to the matrix pipe (every element is read by
    //      exactly one lane, so the split costs the same there as at staging time and needs no second stripe).
    constexpr bool BF3 = (VAR & 4) != 0;
    constexpr bool S4 = (VAR & 8) != 0;       // pooled forward over groups that are not whole tiles (GemmArgs::pool_s4)
    constexpr bool POOL = (VAR & 3) == 1 || (VAR & 3) == 3, WST = (VAR & 3) == 2 || (VAR & 3) == 3;   // 3: pooled AND streamed
    constexpr int BN = NT * 32;
    constexpr int NTH = NT / EH;                   // accumulator tiles per epilogue pass
    constexpr int BNH = BN / EH;                   // columns per epilogue pass
    constexpr int LDW = (KC > BNH ? KC : BNH) + 4; // stripe row stride (floats): 16 B aligned, conflict-free b128
    constexpr int C4 = KC / 4;                     // float4 per operand stripe row
    constexpr int NLD = KC / 8;                    // float4 per lane per 32 x KC operand stripe
    constexpr int O4 = BNH / 4;                    // float4 per output row per pass
    constexpr int NST = BNH / 8;                   // float4 per lane per pass
    static_assert(O4 % 4 == 0 && 64 % O4 == 0, "epilogue pass: whole rows per step, whole steps per 16-row block");
    constexpr int NCOEF = (AM == A_PLAIN) ? 0 : (AM == A_BNRELU ? 2 : (AM == A_DY ? 3 : (AM == A_XYZ ? 6 : 5)));
    constexpr int NTHR = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave id as a SCALAR: everything derived from it (tile index, buffer descriptors) is then provably wave-uniform
    // and hipcc does not wrap each buffer access in a waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool compact = a.blocks != nullptr;
    const int M = a.Mdev ? __builtin_amdgcn_readfirstlane(*a.Mdev) : a.M;
    const int K = a.K, N = a.N;
    const int nchunk = (K + KC - 1) / KC;
    const int Kp = nchunk * KC;
    constexpr int CROWS = WST ? (NCOEF > 0 ? NCOEF : 1) : 6;
    constexpr bool B_ = AM == A_DYPOOLB;             // two pooling blocks per tile
    static_assert(KC == 64 || KC == 32, "stripe rows of 16 or 8 float4");
    static_assert(!BF3 || KC % 16 == 0, "split operands: 16 k per matrix instruction");
    float *Ws = lds;                                        // [Kp][BN], or [2][KC][BN] when streamed (BF3: 6 bytes / element)
    float *coef = Ws + (size_t)(WST ? 2 * KC : Kp) * BN * (BF3 ? 3 : 2) / 2;    // [CROWS][Kp]
    bf16x8 *Wf = reinterpret_cast<bf16x8 *>(lds);           // BF3: [3 pieces][KSW k-steps][NT][64 lanes] fragments of 8 bf16
    const int KSW = (WST ? 2 * KC : Kp) / 16;
    float *ecoef = coef + CROWS * Kp;                       // [6][BN]: bias | (mask scale, mask shift) | xyz-form w0 w1 w2 b
    float *Aw = ecoef + 6 * BN + wave * 32 * LDW;           // [32][LDW] per wave
    float *red = ecoef + 6 * BN + WAVES * 32 * LDW;         // [WAVES][2][BN]
    // grid = (row groups, column blocks): the column blocks of one row group have linear ids that differ by a
    // multiple of 8, i.e. they run on the SAME XCD and the second reader of a stripe hits that XCD's L2
    const int n0 = blockIdx.y * BN;
    const int rowgrp = blockIdx.x, nrowgrp = a.nrowgrp;
    if (rowgrp >= nrowgrp) {
        // padding workgroup: the statistics buffer has a fixed number of partial rows (pcops_mlp_stats_rows), the rows
        // no tile owner writes are zeroed here instead of by a separate memset launch
        if (EM != E_PLAIN && EM != E_PLAINA && a.stats)
            for (int i = tid; i < 2 * BN; i += NTHR)
                if (n0 + i % BN < N) a.stats[((long long)rowgrp * 2 + i / BN) * N + n0 + i % BN] = 0.f;
        if (EM == E_MASKX && a.xstats)
            for (int i = tid; i < 3 * BN; i += NTHR)
                if (n0 + i % BN < N) a.xstats[((long long)rowgrp * 3 + i / BN) * N + n0 + i % BN] = 0.f;
        return;
    }

    // ---- weight tile in LDS, K-QUAD MAJOR:  Ws[k / 4][n][k % 4]  -- the four k a lane's MFMA steps consume for one
    // column sit in ONE 16-byte word, so a B fragment is one ds_read_b128 per 32 columns (it was four ds_read_b32), and
    // the 32 lanes of a half-wave read 512 contiguous bytes (conflict-free)
    // ---- streamed weights: KC x BN chunk in 4 x 4 blocks (4 k-rows x 4 columns), WPB blocks per thread: global (L2) ->
    // registers -> transposed in registers -> LDS buffer
    constexpr int WBLK = (KC / 4) * (BN / 4);
    constexpr int WPB = (WBLK + NTHR - 1) / NTHR;
    float4 wreg[(WST && !BF3) ? 4 * WPB : 1];
    // split operands: a chunk is (KC / 16) NT 64 fragments of 8 k x 1 column, FPB per thread: global (L2) -> registers ->
    // three bf16 pieces -> LDS buffer, already in the order the matrix instruction reads them
    constexpr int FR = (KC / 16) * NT * 64;
    constexpr int FPB = (FR + NTHR - 1) / NTHR;
    float wfr[(WST && BF3) ? 8 * FPB : 1];
    // POOLED forward: column n of the weight tile (and the accumulator start value) is multiplied by sign(gamma[n]), so
    // the tile leaves the matrix pipe as  s (y - pivot):  BN + ReLU is increasing in y for gamma >= 0 and decreasing
    // otherwise, i.e. the pooled row of a group is ALWAYS the arg-max of the accumulator values -- no per-element sign
    // multiply, and the max is taken on the accumulator registers themselves (below).  s = +-1: exact.
    auto colsign = [&](int n) { return (POOL && n < N && a.pgamma[n] < 0.f) ? -1.f : 1.f; };
    float4 wsg[(WST && POOL) ? WPB : 1];
#pragma unroll
    for (int j = 0; j < ((WST && POOL) ? WPB : 1); ++j) {
        const int n = n0 + ((tid + NTHR * j) % (BN / 4)) * 4;
        wsg[j] = make_float4(colsign(n), colsign(n + 1), colsign(n + 2), colsign(n + 3));
    }
    float wsgf[(WST && POOL && BF3) ? FPB : 1];
#pragma unroll
    for (int j = 0; j < ((WST && POOL && BF3) ? FPB : 1); ++j) {
        const int f = tid + NTHR * j;
        wsgf[j] = colsign(n0 + 32 * ((f >> 6) % NT) + (f & 31));
    }
    // (split operands: eight single words per fragment, rows k .. k + 7 of one column.  Through a descriptor over the K N words
    // of W -- a row beyond K answers 0 by the bounds check, a column beyond N or a fragment beyond the chunk gets an offset no
    // tensor reaches -- these are eight loads and nothing else; as predicated pointer loads hipcc made each its own exec-masked
    // branch with 64-bit address arithmetic and waited for every second one: four exposed L2 round trips per chunk, with all
    // eight waves of the workgroup in step)
    const __amdgpu_buffer_rsrc_t rw = make_rsrc_u32(a.W, (WST && BF3) ? (unsigned)K * (unsigned)N * 4u : 0u);
    unsigned wfoff[(WST && BF3) ? FPB : 1];
#pragma unroll
    for (int j = 0; j < ((WST && BF3) ? FPB : 1); ++j) {
        const int f = tid + NTHR * j;
        const int ln = f & 63, nt = (f >> 6) % NT, ksl = f / (64 * NT);
        const int n = n0 + 32 * nt + (ln & 31);
        wfoff[j] = (f < FR && n < N) ? (unsigned)((16 * ksl + 8 * (ln >> 5)) * N + n) * 4u : kOOB;
    }
    auto wload = [&](int kc) {
        if constexpr (BF3) {
            const unsigned cbase = (unsigned)(kc * KC) * (unsigned)N * 4u;      // wave-uniform: the chunk's first row
#pragma unroll
            for (int j = 0; j < FPB; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    wfr[8 * j + e] = __uint_as_float(
                        __builtin_amdgcn_raw_buffer_load_b32(rw, wfoff[j], cbase + (unsigned)e * (unsigned)N * 4u, 0));
            return;
        }
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
            const int e = tid + NTHR * j;
            const int kq = e / (BN / 4), n = n0 + (e % (BN / 4)) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kc * KC + kq * 4 + i;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < WBLK && k < K && n < N) w = *reinterpret_cast<const float4 *>(a.W + (long long)k * N + n);
                wreg[4 * j + i] = w;
            }
        }
    };
    auto wstore = [&](int buf) {
        if constexpr (BF3) {
#pragma unroll
            for (int j = 0; j < FPB; ++j) {
                const int f = tid + NTHR * j;
                const int ln = f & 63, nt = (f >> 6) % NT, ksl = f / (64 * NT);
                if (f < FR) {
                    const float sg = POOL ? wsgf[POOL ? j : 0] : 1.f;         // pooled forward: columns carry sign(gamma)
                    bf16x8 h, m, l;
                    split3(make_float4(wfr[8 * j] * sg, wfr[8 * j + 1] * sg, wfr[8 * j + 2] * sg, wfr[8 * j + 3] * sg),
                           make_float4(wfr[8 * j + 4] * sg, wfr[8 * j + 5] * sg, wfr[8 * j + 6] * sg, wfr[8 * j + 7] * sg),
                           h, m, l);
                    const int at = ((buf * (KC / 16) + ksl) * NT + nt) * 64 + ln;
                    Wf[at] = h;
                    Wf[KSW * NT * 64 + at] = m;
                    Wf[2 * KSW * NT * 64 + at] = l;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
            const int e = tid + NTHR * j;
            if (e < WBLK) {
                float4 *dst = reinterpret_cast<float4 *>(&Ws[((buf * (KC / 4) + e / (BN / 4)) * BN + (e % (BN / 4)) * 4) * 4]);
                const float4 w0 = wreg[4 * j], w1 = wreg[4 * j + 1], w2 = wreg[4 * j + 2], w3 = wreg[4 * j + 3];
                const float4 sg = wsg[POOL ? j : 0];               // pooled forward: columns carry sign(gamma), see below
                dst[0] = make_float4(w0.x * sg.x, w1.x * sg.x, w2.x * sg.x, w3.x * sg.x);
                dst[1] = make_float4(w0.y * sg.y, w1.y * sg.y, w2.y * sg.y, w3.y * sg.y);
                dst[2] = make_float4(w0.z * sg.z, w1.z * sg.z, w2.z * sg.z, w3.z * sg.z);
                dst[3] = make_float4(w0.w * sg.w, w1.w * sg.w, w2.w * sg.w, w3.w * sg.w);
            }
        }
    };
    // ---- resident data: weights + coefficient vectors, loaded once per workgroup
    if (WST) {
        wload(0);
        wstore(0);
    }
    for (int f = tid; f < ((BF3 && !WST) ? KSW * NT * 64 : 0); f += NTHR) {
        const int ln = f & 63, nt = (f >> 6) % NT, ks = f / (64 * NT);
        const int n = n0 + 32 * nt + (ln & 31);
        const float sg = colsign(n);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * (ln >> 5) + e;
            x[e] = (k < K && n < N) ? a.W[(long long)k * N + n] * sg : 0.f;
        }
        bf16x8 h, m, l;
        split3(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), h, m, l);
        Wf[f] = h;
        Wf[KSW * NT * 64 + f] = m;
        Wf[2 * KSW * NT * 64 + f] = l;
    }
    for (int e = tid; e < ((WST || BF3) ? 0 : Kp * (BN / 4)); e += NTHR) {
        const int k = e / (BN / 4), nq = (e % (BN / 4)) * 4;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            const float *src = a.W + (long long)k * N;
            const int n = n0 + nq;
            if (n + 3 < N && (N % 4 == 0)) {
                w = *reinterpret_cast<const float4 *>(src + n);
            } else {
                w.x = n + 0 < N ? src[n + 0] : 0.f;
                w.y = n + 1 < N ? src[n + 1] : 0.f;
                w.z = n + 2 < N ? src[n + 2] : 0.f;
                w.w = n + 3 < N ? src[n + 3] : 0.f;
            }
        }
        if (POOL) {
            const int n = n0 + nq;
            w.x *= colsign(n); w.y *= colsign(n + 1); w.z *= colsign(n + 2); w.w *= colsign(n + 3);
        }
        float *dst = &Ws[((k >> 2) * BN + nq) * 4 + (k & 3)];      // k-quad major (once per workgroup)
        dst[0] = w.x; dst[4] = w.y; dst[8] = w.z; dst[12] = w.w;
    }
    {
        const float *vs[6] = {a.v0, a.v1, a.v2, a.v3, a.v4, nullptr};
        if (AM == A_XYZ) {
            vs[2] = a.xw; vs[3] = a.xw + a.xw_ld; vs[4] = a.xw + 2 * a.xw_ld; vs[5] = a.xw + 3 * a.xw_ld;
        }
        for (int e = tid; e < NCOEF * Kp; e += NTHR) {
            const int which = e / Kp, k = e % Kp;
            coef[which * Kp + k] = k < K ? vs[which][k] : 0.f;
        }
        for (int e = tid; e < BN; e += NTHR) {
            const int n = n0 + e;
            float e0 = 0.f, e1 = 0.f;
            float pv = 0.f;
            if (n < N) {
                if (EM == E_FWD && a.bias) e0 = a.bias[n];
                if (EM == E_FWD && a.pivot) pv = a.pivot[n];
                if (is_mask(EM)) { e0 = a.msc[n]; e1 = a.msh[n]; }
                if (POOL) e1 = a.pgamma[n] < 0.f ? -1.f : 1.f;
            }
            // forward: the accumulators START at bias - pivot, so the tile comes out of the matrix pipe as y - pivot --
            // what the statistics sum -- and the pivot is added back on the way to the store (the add the bias used to
            // be): shifted moments at no extra instruction
            ecoef[e] = EM == E_FWD ? (POOL ? e1 * (e0 - pv) : e0 - pv) : e0;
            ecoef[BN + e] = e1;
            if (EM == E_FWD) ecoef[2 * BN + e] = pv;
            if (EM == E_MASKX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ecoef[(2 + i) * BN + e] = n < N ? a.xw[i * a.xw_ld + n] : 0.f;
            }
            if (has_add(EM)) ecoef[2 * BN + e] = n < N ? a.vconst[n] : 0.f;
        }
    }
    __syncthreads();

    float s1[EH][4], s2[EH][4];                      // statistics of this lane's 4 columns, per epilogue pass
    float sx[(EM == E_MASKX) ? EH : 1][3][4];        // E_MASKX: sums of offset (x) masked gradient
#pragma unroll
    for (int h = 0; h < EH; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[h][e] = s2[h][e] = 0.f;
#pragma unroll
    for (int h = 0; h < ((EM == E_MASKX) ? EH : 1); ++h)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sx[h][i][e] = 0.f;
    const int ocl = (lane % O4) * 4;                 // column quad inside a pass (fixed: 64 % O4 == 0)

    const long long ntiles = ((long long)M + 31) / 32;
    const long long tstride = (long long)nrowgrp * WAVES;
    float4 pa[NLD];                                  // A_PLAIN/A_BNRELU: X;  A_DY: G;  A_DYPOOL: gpool
    float4 pb[is_dy(AM) ? NLD : 1];                  // A_DY*: raw Y
    unsigned pm[(is_pool(AM)) ? NLD : 1];         // A_DYPOOL: 4 arg-max bytes
    float bw[2] = {1.f, 1.f};                        // compacted rows: weight of the first row of the tile's two blocks
    int bs0[2] = {0, 0};                             //                 row-in-group of the first row of each block

    // per-lane byte offsets inside a tile (the row part of element e = lane + 64 j is added as a SCALAR offset)
    const unsigned xvoff = (unsigned)((lane / C4) * a.ldx + (lane % C4) * 4) * 4u;
    const unsigned xrowstep = (unsigned)(64 / C4) * (unsigned)a.ldx * 4u;          // bytes per j
    const unsigned yrowstep = (unsigned)(64 / O4) * (unsigned)a.ldy * 4u;
    const long long glast = is_pool(AM) ? ((long long)M - 1) / a.S : 0;
    constexpr bool U_ = AM == A_DYPOOLU;             // one pooling group per tile
    // any other group size S >= 11: a 32-row tile meets at most FOUR groups (floor(31 / S) + 2); the 16-lane set
    // lane / C4 keeps the (gpool, arg-max) quad of group g0 + lane / C4 and adds its arg rows after the dense staging,
    // like U_ / B_ -- no per-row group arithmetic, one quad per lane instead of one per staged float4 (that form
    // needed 250+ registers and spilled: DGCNN's T-Net, S = 20)
    constexpr bool G_ = AM == A_DYPOOL;
    auto issue = [&](long long tile, int kc) {       // global -> registers, one stripe ahead, branch-free
        const long long row0 = tile * 32;
        if (compact && (is_dy(AM) || B_)) {
            const int nblk = (M + kBlk - 1) / kBlk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                long long bi = tile * 2 + h;
                bi = bi < nblk ? bi : nblk - 1;
                const RowBlock rb = uniform_block(a.blocks, bi);
                bw[h] = rb.w;
                bs0[h] = rb.s0;
                if (B_) {
                    int c = (lane % C4) * 4 + kc * KC;
                    c = c < K ? c : K - 4;
                    pa[h] = *reinterpret_cast<const float4 *>(a.gpool + (long long)rb.g * K + c);
                    pm[h] = *reinterpret_cast<const unsigned *>(a.argmax + (long long)rb.g * K + c);
                }
            }
        }
        if (is_pool(AM) && !B_) {
            const PoolRows pr(row0, a.S);
            {
                int c = (lane % C4) * 4 + kc * KC;
                c = c < K ? c : K - 4;
                long long gi = pr.g0 + (G_ ? lane / C4 : 0);
                gi = gi < glast ? gi : glast;
                pa[0] = *reinterpret_cast<const float4 *>(a.gpool + gi * K + c);
                pm[0] = *reinterpret_cast<const unsigned *>(a.argmax + gi * K + c);
            }
        }
        const long long left = ((long long)M - row0) * a.ldx * 4;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + row0 * a.ldx, left);
        const __amdgpu_buffer_rsrc_t rx2 = make_rsrc((is_dy(AM) ? a.X2 : a.X) + row0 * a.ldx, left);
        const unsigned kbytes = (unsigned)(kc * KC) * 4u;
        if (AM == A_XYZ) {       // 16 bytes per ROW (broadcast over the C4 lanes of a row)
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.off4 + row0 * 4, ((long long)M - row0) * 16);
#pragma unroll
            for (int j = 0; j < NLD; ++j) pa[j] = buf_load4(ro, (unsigned)(lane / C4) * 16u, (unsigned)j * (64 / C4) * 16u);
            return;
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            if (!is_pool(AM)) pa[j] = buf_load4(rx, xvoff, kbytes + (unsigned)j * xrowstep);
            if (is_dy(AM)) pb[j] = buf_load4(rx2, xvoff, kbytes + (unsigned)j * xrowstep);
        }
    };
    // FULL_: the tile has all 32 rows and K == Kp -- wave-uniform, true for every tile but the last; the variant without
    // it carries the per-element range checks (4 v_cndmask per float4 here, an exec-masked branch per row group in the
    // epilogue: per-row exec masking had turned the epilogue into ~170 basic blocks of 5-10 instructions, each with its
    // own s_and_saveexec / s_cbranch / s_waitcnt -- 2 000 of the 3 300 non-MFMA instructions of a 128 -> 256 tile)
    auto stage = [&](long long tile, int kc, auto full_) {       // registers -> transform -> wave stripe
        constexpr bool FULL = decltype(full_)::value;
        const long long row0 = tile * 32;
        const PoolRows prs(is_pool(AM) ? row0 : 0, is_pool(AM) ? a.S : 1);
        const int cl = (lane % C4) * 4;              // fixed per lane (64 % C4 == 0)
        const int c = cl + kc * KC;
        float4 c0, c1, c2, c3, c4, c5;
        if (NCOEF >= 6) c5 = *reinterpret_cast<const float4 *>(&coef[5 * Kp + c]);
        if (NCOEF >= 2) {
            c0 = *reinterpret_cast<const float4 *>(&coef[0 * Kp + c]);
            c1 = *reinterpret_cast<const float4 *>(&coef[1 * Kp + c]);
        }
        if (NCOEF >= 3) c2 = *reinterpret_cast<const float4 *>(&coef[2 * Kp + c]);
        if (NCOEF >= 5) {
            c3 = *reinterpret_cast<const float4 *>(&coef[3 * Kp + c]);
            c4 = *reinterpret_cast<const float4 *>(&coef[4 * Kp + c]);
        }
        const int rem = (int)((long long)M - row0 < 32 ? (long long)M - row0 : 32);   // rows of this tile (scalar, 32 bit)
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int r = (lane + 64 * j) / C4;
            const bool in = FULL || ((r < rem) && (c < K));
            float4 x = pa[(U_ || G_) ? 0 : (B_ ? j / (C4 / 4) : j)];     // (a 16-row block = C4 / 4 staging steps)
            if (AM == A_BNRELU) {
                x.x = fmaxf(fmaf(x.x, c0.x, c1.x), 0.f);
                x.y = fmaxf(fmaf(x.y, c0.y, c1.y), 0.f);
                x.z = fmaxf(fmaf(x.z, c0.z, c1.z), 0.f);
                x.w = fmaxf(fmaf(x.w, c0.w, c1.w), 0.f);
            } else if (AM == A_XYZ) {
                const float4 o = x;
                x.x = fmaxf(fmaf(xyz_y(o, c2.x, c3.x, c4.x, c5.x), c0.x, c1.x), 0.f);
                x.y = fmaxf(fmaf(xyz_y(o, c2.y, c3.y, c4.y, c5.y), c0.y, c1.y), 0.f);
                x.z = fmaxf(fmaf(xyz_y(o, c2.z, c3.z, c4.z, c5.z), c0.z, c1.z), 0.f);
                x.w = fmaxf(fmaf(xyz_y(o, c2.w, c3.w, c4.w, c5.w), c0.w, c1.w), 0.f);
            } else if (is_dy(AM)) {
                const float4 y = pb[j];
                float4 g = x;
                if (is_pool(AM)) {
                    // pooled form: the p.G term has ONE row per (group, channel); the rows get the dense part
                    // q.Y + t  here and the few arg rows are added afterwards (below)
                    g = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (compact && j % (C4 / 4) == 0) {
                    // rows 0 and 16 of the tile (the first C4 lanes of every C4 / 4-th step) open a block:
                    // dY = p.G + w (q.Y + t)
                    const float w = lane < C4 ? bw[j / (C4 / 4)] : 1.f;
                    x.x = fmaf(c0.x, g.x, w * fmaf(c1.x, y.x, c2.x));
                    x.y = fmaf(c0.y, g.y, w * fmaf(c1.y, y.y, c2.y));
                    x.z = fmaf(c0.z, g.z, w * fmaf(c1.z, y.z, c2.z));
                    x.w = fmaf(c0.w, g.w, w * fmaf(c1.w, y.w, c2.w));
                } else {
                    x.x = fmaf(c0.x, g.x, fmaf(c1.x, y.x, c2.x));
                    x.y = fmaf(c0.y, g.y, fmaf(c1.y, y.y, c2.y));
                    x.z = fmaf(c0.z, g.z, fmaf(c1.z, y.z, c2.z));
                    x.w = fmaf(c0.w, g.w, fmaf(c1.w, y.w, c2.w));
                }
            }
            if (!in) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&Aw[r * LDW + cl]) = x;
        }
        if (is_pool(AM)) {
            // the arg rows: lanes 0..C4-1 (one per column quad) take the tile's group / its first block, the next C4 lanes
            // the second block; each adds  p . gpool  to the stripe row the arg-max byte names, if that row is in this tile.
            // (Per element this replaces byte extract + two compares + and + select + multiply-add of round 2 -- 1 150
            // of the 1 540 vector instructions of a 256 -> 128 tile -- by ONE fused multiply-add; the sparse pass is ~40
            // instructions on half a wave.)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (any group size: the 16-lane set lane / 16 takes group g0 + lane / 16, whose row s sits at tile row
            // (lane / 16) S - s0 + s)
            const int grp = lane / C4;
            const bool gvalid = !G_ || prs.g0 + grp <= glast;
            if (lane < (G_ ? 4 * C4 : (B_ ? 2 * C4 : C4)) && c < K && gvalid) {
                const bool second = B_ && lane >= C4;
                const float4 gp = second ? pa[B_ ? 1 : 0] : pa[0];
                const unsigned am = second ? pm[B_ ? 1 : 0] : pm[0];
                // row-in-group of the tile's (block's) first row; for G_ the tile row of the group's row 0, negated
                const int base = B_ ? (second ? bs0[1] : bs0[0]) : (G_ ? prs.s0 - grp * a.S : prs.s0);
                const int span = B_ ? kBlk : 32, roff = second ? kBlk : 0;
                const float gv[4] = {gp.x, gp.y, gp.z, gp.w};
                const float pv[4] = {c0.x, c0.y, c0.z, c0.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rel = (int)((am >> (8 * e)) & 0xffu) - base;
                    if ((unsigned)rel < (unsigned)span && roff + rel < rem && gv[e] != 0.f) {
                        float *dst = &Aw[(roff + rel) * LDW + cl + e];
                        *dst = fmaf(pv[e], gv[e], *dst);
                    }
                }
            }
        }
    };
    // a wave walks whole pooling groups: SUB consecutive 32-row tiles (SUB = 1 without pooling)
    const int SUB = POOL ? a.pool_sub : 1;
    const long long nsuper = (ntiles + SUB - 1) / SUB;
    // ---- pooled forward: the neighbourhood max is taken on the ACCUMULATOR registers, before the tile is transposed.
    // acc[nt][v] of a lane is (row (v & 3) + 8 (v >> 2) + 4 (lane >> 5), column 32 nt + (lane & 31)): sixteen rows of ONE
    // column, ascending in v, so the running (max, first row) of a column is a chain of  v_cmp + 2 v_cndmask  on
    // registers -- four independent chains per lane (NT) -- and the two half-waves meet in ONE exchange per result.
    // (Round 2 took the max after the transposition, on float4 row segments: sign multiply, compare and two selects per
    // element inside the LDS-latency chain of the epilogue, plus a two-step cross-lane combine per 16-row block -- the
    // pooled epilogue then cost a wave 18 500 cycles per tile against 7 000 without pooling, tools/phase_prof.py.)
    float gmx[POOL ? NT : 1];                         // uncompacted groups: running max over the group's tiles, its row
    int grw[POOL ? NT : 1];
    float gmy[S4 ? NT : 1];                           // pool_s4: the running maxima of the group AFTER the oldest open one (gmx)
    unsigned prw = 0u, pry = 0u;                      // ... and the rows of both, one byte per column block
    int pgo = 0;                                      // pool_s4: the oldest open group of the walk
    static_assert(!S4 || NT <= 4, "pool_s4: four row bytes per register");
#pragma unroll
    for (int i = 0; i < (POOL ? NT : 1); ++i) { gmx[i] = -INFINITY; grw[i] = 0; }
#pragma unroll
    for (int i = 0; i < (S4 ? NT : 1); ++i) gmy[i] = -INFINITY;
    const int peer32 = (lane ^ 32) << 2;
    // value + row of the better of (this half-wave, the other): larger value, then lower row
    auto meet = [&](float &m, int &r) {
        const float om = __int_as_float(__builtin_amdgcn_ds_bpermute(peer32, __float_as_int(m)));
        const int orow = __builtin_amdgcn_ds_bpermute(peer32, r);
        const bool take = (om > m) | ((om == m) & (orow < r));
        m = take ? om : m;
        r = take ? orow : r;
    };
    // the pooled pair of group g, column block nt.  The column part of the address is rebuilt from an opaque copy of the lane
    // number at every store: hoisted, hipcc kept the two per-lane 64-bit column addresses alive across the kernel -- in scratch in the
    // S4 kernels, reloaded behind s_waitcnt vmcnt(0) in front of every store, which also waited for the next tile's stripe request.
    // (Descriptor stores were tried first: their eight scalar registers push the 128-column kernels into scratch elsewhere.)
    auto pool_store = [&](long long g, int nt, float yv, int r) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int n = n0 + 32 * nt + (ln & 31);
        if (ln < 32 && n < N) {
            a.ysel[g * N + n] = yv;
            a.psel[g * N + n] = (unsigned char)r;
        }
    };
    auto pool_acc = [&](const f32x16 (&acc)[NT], long long tile, int sub, long long st) {
        const int hrow = 4 * (lane >> 5);
        if (compact) {
            // 16-row blocks: block b = accumulator registers 8 b .. 8 b + 7; one partial (value, row-in-group) per block
            const int nblk = (M + kBlk - 1) / kBlk;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long long blk = tile * 2 + b;
                if (blk >= nblk) continue;                                   // wave-uniform
                const int s0 = uniform_block(a.blocks, blk).s0;              // row-in-group of the block's first row
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float m = acc[nt][8 * b];
                    int r = 0;
#pragma unroll
                    for (int i = 1; i < 8; ++i) {
                        const float x = acc[nt][8 * b + i];
                        const bool gt = x > m;                               // strict: the first maximum stays
                        m = gt ? x : m;
                        r = gt ? (i & 3) + 8 * (i >> 2) : r;
                    }
                    r += hrow;
                    meet(m, r);
                    {
                        const float sg = ecoef[BN + 32 * nt + (lane & 31)], pv = ecoef[2 * BN + 32 * nt + (lane & 31)];
                        pool_store(blk, nt, fmaf(m, sg, pv), s0 + r);        // the raw y at that row, as it is stored
                    }
                }
            }
        } else if constexpr (S4) {
            // Groups of S rows, S % 4 == 0: the four rows a lane holds in registers 4 seg .. 4 seg + 3 (rows 8 seg + hrow .. + 3 of
            // the tile) lie inside ONE group, so the lane takes their (max, first row) on registers and merges it into the
            // running pair of the OLDEST OPEN group (gmx / grw) or of the one after it (gmy / gry) -- the two half-waves may be
            // in neighbouring groups, never further apart (S >= 8).  Rows 8 seg .. 8 seg + 7 are done in both half-waves after
            // segment seg: a group that ends there is met across the half-waves and stored at once, and the next group's
            // pair moves up.
            const int S_ = a.pool_s4;
            const int gps = 32 * SUB / S_;                                   // groups per walk
            if (sub == 0) pgo = 0;
#pragma unroll
            for (int seg = 0; seg < 4; ++seg) {
                const int rs = 32 * sub + 8 * seg + hrow;                    // first row of the lane's segment, in the walk
                const int g = (rs * a.pool_inv) >> 16;                       // rs / S_
                const int rg0 = rs - g * S_;
                const bool nxt = g != pgo;                                   // the group after the oldest open one
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float m = acc[nt][4 * seg];
                    int r = 0;
#pragma unroll
                    for (int i = 1; i < 4; ++i) {
                        const float x = acc[nt][4 * seg + i];
                        const bool gt = x > m;                               // strict: the first maximum stays
                        m = gt ? x : m;
                        r = gt ? i : r;
                    }
                    r += rg0;
                    // (the rows of the NT column blocks share ONE register per pair, a byte each: the walk is at most 256 rows)
                    const bool t0 = !nxt && m > gmx[nt], t1 = nxt && m > gmy[nt];
                    gmx[nt] = t0 ? m : gmx[nt];
                    gmy[nt] = t1 ? m : gmy[nt];
                    const unsigned bm = 0xffu << (8 * nt), rb = (unsigned)r << (8 * nt);
                    prw = t0 ? ((prw & ~bm) | rb) : prw;
                    pry = t1 ? ((pry & ~bm) | rb) : pry;
                }
                const int pe = 32 * sub + 8 * seg + 8;                       // rows of the walk done so far (wave-uniform)
                if ((pgo + 1) * S_ <= pe) {                                  // the oldest open group is complete in both half-waves
                    const long long gg = st * gps + pgo;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float m = gmx[nt];
                        int r = (int)((prw >> (8 * nt)) & 0xffu);
                        meet(m, r);
                        {
                            const float sg = ecoef[BN + 32 * nt + (lane & 31)], pv = ecoef[2 * BN + 32 * nt + (lane & 31)];
                            pool_store(gg, nt, fmaf(m, sg, pv), r);
                        }
                        gmx[nt] = gmy[nt];
                        gmy[nt] = -INFINITY;
                    }
                    prw = pry;
                    pry = 0u;
                    ++pgo;
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float m = gmx[nt];
                int r = grw[nt];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float x = acc[nt][i];
                    const bool gt = x > m;
                    m = gt ? x : m;
                    r = gt ? sub * 32 + (i & 3) + 8 * (i >> 2) + hrow : r;
                }
                gmx[nt] = m;
                grw[nt] = r;
            }
            if (sub == SUB - 1) {                                            // group complete
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float m = gmx[nt];
                    int r = grw[nt];
                    meet(m, r);
                    {
                        const float sg = ecoef[BN + 32 * nt + (lane & 31)], pv = ecoef[2 * BN + 32 * nt + (lane & 31)];
                        pool_store(st, nt, fmaf(m, sg, pv), r);
                    }
                    gmx[nt] = -INFINITY;
                    grw[nt] = 0;
                }
            }
        }
    };
#ifdef PCOPS_PHASE_PROF
    unsigned long long pf_stage = 0, pf_mfma = 0, pf_epi = 0, pf_tiles = 0, pf_t;
    const unsigned long long pf_start = PROF_T();
#endif
    long long st = (long long)rowgrp * WAVES + wave;
    int sub = 0;
    if (st < nsuper) issue(st * SUB, 0);
    // streamed weights: every wave runs the workgroup's number of rounds (the chunk barriers are workgroup-wide);
    // a wave without a tile in the last round only helps refilling the weight buffers
    const long long st0 = (long long)rowgrp * WAVES;
    const long long nrounds = st0 < nsuper ? (nsuper - st0 + tstride - 1) / tstride * SUB : 0;   // SUB tiles per group
    long long round = 0;
    int wt = 0;                                      // streamed chunks consumed so far (buffer = wt & 1)
    while (WST ? round < nrounds : st < nsuper) {
        const bool active = !WST || st < nsuper;
        const long long tile = st * SUB + sub;
        long long nst = st;
        int nsub = sub + 1;
        if (nsub == SUB) { nsub = 0; nst = st + tstride; }
        const bool more = nst < nsuper;
        const long long next_tile = nst * SUB + nsub;
        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = EM == E_FWD ? ecoef[32 * i + (lane & 31)] : 0.f;   // bias - pivot
        f32x16 sm[BF3 ? NT : 1];                        // split operands: the sum of the small partial products
#pragma unroll
        for (int i = 0; i < (BF3 ? NT : 1); ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) sm[i][v] = 0.f;

        for (int kc = 0; kc < nchunk; ++kc) {
            // streamed weights: the next chunk's words are requested BEHIND this chunk's stage() and the next stripe's request, so
            // that the stage's wait for its stripe (in-order counter) does not also wait for them; wstore() at the bottom, a whole
            // matrix phase later, takes both
            // (split-operand kernels only: the fp32 ones, whose float4 weight loads the stage does not meet, measured 7 % slower so)
            const bool wl = WST && !(round + 1 == nrounds && kc + 1 == nchunk);
            if (wl && !(BF3 && active)) wload(kc + 1 < nchunk ? kc + 1 : 0);
            if (active) {
#ifdef PCOPS_PHASE_PROF
            pf_t = PROF_T();
#endif
            __builtin_amdgcn_wave_barrier();
#ifndef PCOPS_NO_STAGE_FULL
            if ((long long)M - tile * 32 >= 32 && Kp == K) stage(tile, kc, std::true_type{});
            else
#endif
                stage(tile, kc, std::false_type{});
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // next stripe: the next K chunk of this tile, or chunk 0 of this wave's next tile
            // (two call sites on purpose.  As ONE -- issue(same ? tile : next_tile, same ? kc + 1 : 0) -- the pooled data gradients
            // lose the register copies that meet the two inlined bodies behind their loads, but the 128-column forward kernel
            // gets its scratch reload back: dgrad 256 -> 128 946 -> 974 us, fwd 128 -> 256 673 -> 705 us, round 6)
            if (kc + 1 < nchunk) issue(tile, kc + 1);
            else if (more) issue(next_tile, 0);
            if (BF3 && wl) wload(kc + 1 < nchunk ? kc + 1 : 0);

#ifdef PCOPS_PHASE_PROF
            { const unsigned long long n_ = PROF_T(); pf_stage += n_ - pf_t; pf_t = n_; }
#endif
            if constexpr (BF3) {
                // split operands: a step is 16 k -- the lane's eight stripe values of its row (two ds_read_b128) split into
                // three bf16 pieces, the weight pieces as 16-byte fragments, six matrix instructions per column block.  The
                // fragments of one weight piece are requested while the products of the previous one issue.
                // (the fragment addresses are rebuilt from an opaque copy of the lane number per chunk: hoisted out of the tile loop
                // they were the registers that went to scratch in the 128-column pooled forward kernel -- and a scratch RELOAD is a
                // vector-memory load, so the first read of the reloaded address, right behind issue(), carried s_waitcnt vmcnt(0):
                // every chunk waited for the stripe it had just requested)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const float *arow = &Aw[(ln & 31) * LDW + 8 * (ln >> 5)];
                const bf16x8 *wf = Wf + ((WST ? (wt & 1) : kc) * (KC / 16)) * NT * 64 + ln;
                const int pst = KSW * NT * 64;                       // piece stride (fragments)
                const int kleft = K - kc * KC;
                const int nstep = kleft >= KC ? KC / 16 : (kleft + 15) / 16;
#pragma unroll
                for (int st_ = 0; st_ < KC / 16; ++st_) {
                    if (st_ < nstep) {
                        const float4 x0 = *reinterpret_cast<const float4 *>(arow + 16 * st_);
                        const float4 x1 = *reinterpret_cast<const float4 *>(arow + 16 * st_ + 4);
                        bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bh[nt] = wf[(st_ * NT + nt) * 64];
                        bf16x8 ah, am, al;
                        split3(x0, x1, ah, am, al);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bm[nt] = wf[pst + (st_ * NT + nt) * 64];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sm[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nt], sm[nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sm[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[nt], sm[nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nt], acc[nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bl[nt] = wf[2 * pst + (st_ * NT + nt) * 64];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sm[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[nt], sm[nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sm[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[nt], sm[nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sm[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nt], sm[nt], 0, 0, 0);
                    }
                }
                // (the steps of a chunk as ONE basic block -- no wave-uniform branch per step, so that a step's LDS reads may move
                // under the previous step's matrix instructions -- measured in round 6: the 128-column forward kernels then
                // spill (fwd_pool 64 -> 128: 683 -> 2 250 us), the 64-column data gradients do not move (1 127 -> 1 148 us))
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
            const float *arow = &Aw[(lane & 31) * LDW + 4 * (lane >> 5)];
            // B fragments: quad row (chunk base) + 2 it + (lane >> 5), column 32 nt + (lane & 31)
            const float4 *bq = reinterpret_cast<const float4 *>(Ws) +
                               ((WST ? (wt & 1) : kc) * (KC / 4) + (lane >> 5)) * BN + (lane & 31);
            // two register sets in ping-pong: while the 4 NT MFMAs of step it run, the fragments of step it + 1 are in
            // flight from LDS into the other set -- no copies between the sets, one loop body = two steps
            float4 a0 = *reinterpret_cast<const float4 *>(arow), a1;
            float4 b0[NT], b1[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = bq[32 * nt];
            auto mfma16 = [&](const float4 &av, const float4 (&bv)[NT]) {
                const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float be = t == 0 ? bv[nt].x : (t == 1 ? bv[nt].y : (t == 2 ? bv[nt].z : bv[nt].w));
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[t], be, acc[nt], 0, 0, 0);
                    }
            };
            static_assert((KC / 8) % 2 == 0, "two steps per loop body");
            // a chunk that reaches beyond K (K = 96: the second chunk holds 32 real columns and 32 of zeros) only runs the
            // steps that have something to multiply -- wave-uniform trip count, the skipped products are exact zeros
            const int kleft = K - kc * KC;
            const int nit = kleft >= KC ? KC / 8 : ((kleft + 15) / 16) * 2;
#pragma unroll 1
            for (int it = 0; it < nit; it += 2) {
                a1 = *reinterpret_cast<const float4 *>(arow + 8 * (it + 1));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b1[nt] = bq[(2 * (it + 1)) * BN + 32 * nt];
                mfma16(a0, b0);
                // (the last body re-reads its own step instead of branching: same addresses, nobody uses the result)
                const int nx = it + 2 < nit ? it + 2 : it + 1;
                a0 = *reinterpret_cast<const float4 *>(arow + 8 * nx);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b0[nt] = bq[(2 * nx) * BN + 32 * nt];
                mfma16(a1, b1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#ifdef PCOPS_PHASE_PROF
            {   // (the MFMAs are asynchronous: read one accumulator register so that the clock is taken after the last one)
                float sink_ = acc[NT - 1][15];
                asm volatile("" ::"v"(sink_));
                const unsigned long long n_ = PROF_T(); pf_mfma += n_ - pf_t; pf_t = n_;
            }
#endif
            }
            if (WST) {      // refill the other buffer (last read one chunk ago, before the previous barrier)
                wstore((wt + 1) & 1);
                __syncthreads();
                ++wt;
            }
        }
        if (active) {
        if constexpr (BF3) {
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] += sm[i];
        }
        if (POOL) pool_acc(acc, tile, sub, st);
        // ---- epilogue: accumulators -> stripe (transposed, EH column passes) -> 16-byte row-segment stores
        const long long row0 = tile * 32;
        const int erem = (int)((long long)M - row0 < 32 ? (long long)M - row0 : 32);   // rows of this tile (scalar, 32 bit)
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.Y + row0 * a.ldy, ((long long)M - row0) * a.ldy * 4);
        const bool ynt = a.nt_out != 0;
        const __amdgpu_buffer_rsrc_t rprev =
            make_rsrc(((EM == E_MASK || EM == E_MASKA) ? a.Yprev : a.Y) + row0 * a.ldy, ((long long)M - row0) * a.ldy * 4);
        auto epilogue = [&](auto full_) {
        constexpr bool FULL = decltype(full_)::value;        // all 32 rows and all BN columns exist: no range checks
#pragma unroll
        for (int h = 0; h < EH; ++h) {
            const int ocq = h * BNH + ocl;                   // this lane's column quad in the BN-wide tile
            const bool ocin = FULL || (n0 + ocq < N);        // N % 4 == 0 (launcher)
            const unsigned yvoff = ocin ? (unsigned)((lane / O4) * a.ldy + n0 + ocq) * 4u : kOOB;
            float4 py[is_mask(EM) ? NST : 1];
            float4 po[(EM == E_MASKX) ? NST : 1];
            if (EM == E_MASKX) {   // the previous layer's raw output rebuilt from the row offsets
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.off4 + row0 * 4, ((long long)M - row0) * 16);
                const float4 xw0 = *reinterpret_cast<const float4 *>(&ecoef[2 * BN + ocq]);
                const float4 xw1 = *reinterpret_cast<const float4 *>(&ecoef[3 * BN + ocq]);
                const float4 xw2 = *reinterpret_cast<const float4 *>(&ecoef[4 * BN + ocq]);
                const float4 xb = *reinterpret_cast<const float4 *>(&ecoef[5 * BN + ocq]);
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    const float4 o = buf_load4(ro, (unsigned)(lane / O4) * 16u, (unsigned)j * (64 / O4) * 16u);
                    po[j] = o;
                    py[j] = make_float4(xyz_y(o, xw0.x, xw1.x, xw2.x, xb.x), xyz_y(o, xw0.y, xw1.y, xw2.y, xb.y),
                                        xyz_y(o, xw0.z, xw1.z, xw2.z, xb.z), xyz_y(o, xw0.w, xw1.w, xw2.w, xb.w));
                }
            }
            if (EM == E_MASK || EM == E_MASKA) {
                // the mask tensor is requested HERE (not a tile ahead): it would cost NST more live float4 across the
                // whole MFMA phase, and with two waves per SIMD the partner wave covers this latency.  (Round 6 measured the
                // request a phase ahead on the split-operand data gradients, which have the registers: 1 127 -> 1 148 us and
                // 587 -> 579 us -- nothing.)
#pragma unroll
                for (int j = 0; j < NST; ++j) py[j] = buf_load4(rprev, yvoff, (unsigned)j * yrowstep);
            }
            float4 pad[has_add(EM) ? NST : 1];
            if (has_add(EM)) {
                // compact addend rows: most rows have none (slot -1 -> an offset the bounds check rejects -> zeros)
                const __amdgpu_buffer_rsrc_t radd = make_rsrc(a.addend, a.add_bytes);
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    const int rr = (lane + 64 * j) / O4;
                    const int slot = (FULL || rr < erem) ? a.rowmap[row0 + rr] : -1;
                    const unsigned off = (slot >= 0 && ocin) ? ((unsigned)slot * (unsigned)a.add_ld + (unsigned)(n0 + ocq)) * 4u : kOOB;
                    pad[j] = buf_load4(radd, off, 0u);
                }
            }
            const float4 eb = *reinterpret_cast<const float4 *>(&ecoef[ocq]);
            const float4 em = *reinterpret_cast<const float4 *>(&ecoef[BN + ocq]);
            const float4 epv = EM == E_FWD ? *reinterpret_cast<const float4 *>(&ecoef[2 * BN + ocq])
                                           : make_float4(0.f, 0.f, 0.f, 0.f);             // forward: the pivot
            float ew[2] = {1.f, 1.f};                        // compacted rows: statistics weight of rows 0 / 16
            if (EM == E_FWD && compact) {
                const int nblk = (M + kBlk - 1) / kBlk;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    long long bi = tile * 2 + hb;
                    bi = bi < nblk ? bi : nblk - 1;
                    const RowBlock rb = uniform_block(a.blocks, bi);
                    ew[hb] = rb.w;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    Aw[((v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)) * LDW + 32 * nt + (lane & 31)] = acc[h * NTH + nt][v];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                // branch-free the whole pass is ONE basic block: without a fence the scheduler hoists all NST stripe
                // reads (and everything that depends on them) to the top -- 32+ live registers more, which spilled
                if (FULL && j % 2 == 0) __builtin_amdgcn_sched_barrier(0);
                const int r = (lane + 64 * j) / O4;
                float4 o = *reinterpret_cast<const float4 *>(&Aw[r * LDW + ocl]);
                if (FULL || (r < erem && ocin)) {
                    if (EM == E_FWD) {
                        // o = y - pivot here (accumulator start value): the statistics take it as it is
                        // rows 0 / 16 of the tile stand for w rows.  A pass step covers 64 / O4 rows: a 16-row block opens
                        // every O4 / 4 steps, in the lanes of the step's first row
                        if (compact && j % (O4 / 4) == 0) {
                            const float w = lane < O4 ? ew[j / (O4 / 4)] : 1.f;
                            const float4 wo = make_float4(w * o.x, w * o.y, w * o.z, w * o.w);
                            s1[h][0] += wo.x; s1[h][1] += wo.y; s1[h][2] += wo.z; s1[h][3] += wo.w;
                            s2[h][0] = fmaf(wo.x, o.x, s2[h][0]); s2[h][1] = fmaf(wo.y, o.y, s2[h][1]);
                            s2[h][2] = fmaf(wo.z, o.z, s2[h][2]); s2[h][3] = fmaf(wo.w, o.w, s2[h][3]);
                        } else {
                        s1[h][0] += o.x; s1[h][1] += o.y; s1[h][2] += o.z; s1[h][3] += o.w;
                        s2[h][0] = fmaf(o.x, o.x, s2[h][0]); s2[h][1] = fmaf(o.y, o.y, s2[h][1]);
                        s2[h][2] = fmaf(o.z, o.z, s2[h][2]); s2[h][3] = fmaf(o.w, o.w, s2[h][3]);
                        }
                        if (POOL) {   // the tile is  s (y - pivot)  (column signs folded into the weights): back to y
                            o.x = fmaf(o.x, em.x, epv.x); o.y = fmaf(o.y, em.y, epv.y);
                            o.z = fmaf(o.z, em.z, epv.z); o.w = fmaf(o.w, em.w, epv.w);
                        } else {
                            o.x += epv.x; o.y += epv.y; o.z += epv.z; o.w += epv.w;      // back to y
                        }
                    } else if (EM == E_PLAINA) {
                        const float4 vc = *reinterpret_cast<const float4 *>(&ecoef[2 * BN + ocq]);
                        o.x += pad[j].x + vc.x; o.y += pad[j].y + vc.y; o.z += pad[j].z + vc.z; o.w += pad[j].w + vc.w;
                    } else if (is_mask(EM)) {
                        const float4 yp = py[j];
                        if (EM == E_MASKA) {
                            const float4 vc = *reinterpret_cast<const float4 *>(&ecoef[2 * BN + ocq]);
                            o.x += pad[j].x + vc.x; o.y += pad[j].y + vc.y; o.z += pad[j].z + vc.z; o.w += pad[j].w + vc.w;
                        }
                        o.x = fmaf(yp.x, eb.x, em.x) > 0.f ? o.x : 0.f;
                        o.y = fmaf(yp.y, eb.y, em.y) > 0.f ? o.y : 0.f;
                        o.z = fmaf(yp.z, eb.z, em.z) > 0.f ? o.z : 0.f;
                        o.w = fmaf(yp.w, eb.w, em.w) > 0.f ? o.w : 0.f;
                        s1[h][0] += o.x; s1[h][1] += o.y; s1[h][2] += o.z; s1[h][3] += o.w;
                        s2[h][0] = fmaf(o.x, yp.x, s2[h][0]); s2[h][1] = fmaf(o.y, yp.y, s2[h][1]);
                        s2[h][2] = fmaf(o.z, yp.z, s2[h][2]); s2[h][3] = fmaf(o.w, yp.w, s2[h][3]);
                        if (EM == E_MASKX) {
                            const float4 of = po[j];
                            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                sx[h][0][e] = fmaf(of.x, ov[e], sx[h][0][e]);
                                sx[h][1][e] = fmaf(of.y, ov[e], sx[h][1][e]);
                                sx[h][2][e] = fmaf(of.z, ov[e], sx[h][2][e]);
                            }
                        }
                    }
                }
                // (a pooled forward whose backward is algebraic, or that has no backward, passes Y == NULL: the
                // activation only exists as statistics and group extrema)
                if ((EM != E_MASKX && !(EM == E_FWD && POOL)) || a.Y)
                    buf_store4(rout, yvoff, (unsigned)j * yrowstep, o, ynt);  // rows >= M / columns >= N: dropped by the bounds check
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        };
        // (the xyz-mask epilogue keeps two more float4 arrays alive per pass: branch-free it needs ~40 registers more
        // than the 256 a wave has at two waves per SIMD and spills -- 748 us instead of 517 for SA1's layer -- so it
        // stays on the range-checked form)
        if (EM != E_MASKX && erem == 32 && n0 + BN <= N) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        }
#ifdef PCOPS_PHASE_PROF
        if (active) { const unsigned long long n_ = PROF_T(); pf_epi += n_ - pf_t; ++pf_tiles; }
#endif
        st = nst;
        sub = nsub;
        ++round;
    }
#ifdef PCOPS_PHASE_PROF
    if (lane == 0) {
        atomicAdd(&g_phase_prof[0], pf_stage);
        atomicAdd(&g_phase_prof[1], pf_mfma);
        atomicAdd(&g_phase_prof[2], pf_epi);
        atomicAdd(&g_phase_prof[3], PROF_T() - pf_start);
        atomicAdd(&g_phase_prof[4], pf_tiles);
        atomicAdd(&g_phase_prof[5], 1ull);
    }
#endif

    if (EM != E_PLAIN && EM != E_PLAINA && a.stats) {
        // lanes l, l + O4, l + 2 O4 ... own the same 4 columns
#pragma unroll
        for (int h = 0; h < EH; ++h) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int off = 32; off >= O4; off >>= 1) {
                    s1[h][e] += __shfl_xor(s1[h][e], off, 64);
                    s2[h][e] += __shfl_xor(s2[h][e], off, 64);
                }
            }
            if (lane < O4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (pooled forward: the sums ran over  s (y - pivot);  s^2 = 1, and the first moment gets its sign back)
                    red[(wave * 2 + 0) * BN + h * BNH + ocl + e] = POOL ? s1[h][e] * ecoef[BN + h * BNH + ocl + e] : s1[h][e];
                    red[(wave * 2 + 1) * BN + h * BNH + ocl + e] = s2[h][e];
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NTHR) {
            const int which = i / BN, c = i % BN;
            if (n0 + c < N) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) v += red[(w * 2 + which) * BN + c];
                a.stats[((long long)rowgrp * 2 + which) * N + n0 + c] = v;
            }
        }
    }
    if (EM == E_MASKX && a.xstats) {
        // the three offset sums go through the same scratch, one at a time
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < EH; ++h) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = sx[h][i][e];
#pragma unroll
                    for (int off = 32; off >= O4; off >>= 1) v += __shfl_xor(v, off, 64);
                    if (lane < O4) red[wave * BN + h * BNH + ocl + e] = v;
                }
            }
            __syncthreads();
            for (int c = tid; c < BN; c += NTHR) {
                if (n0 + c < N) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) v += red[w * BN + c];
                    a.xstats[((long long)rowgrp * 3 + i) * N + n0 + c] = v;
                }
            }
        }
    }
}

struct WsPlan {
    int kc, bn, waves, eh, ncb, gy;
    bool wst;        // weights streamed (K > 256)
    bool bf3;        // split operands on the 16-bit matrix pipe (kernel: VAR + 4)
    size_t lds;
};

// split-operand form: weights as three bf16 pieces (6 bytes per element), 32-wide operand stripes, 32-column epilogue passes
static size_t ws_lds_bytes_bf3(int Kp, int bn, int waves, bool wst, int ncoef) {
    const int ldw = 32 + 4;
    const int crows = wst ? (ncoef > 0 ? ncoef : 1) : 6;
    return (size_t)((wst ? 2 * 32 : Kp) * bn * 3 / 2 + crows * Kp + 6 * bn + waves * 32 * ldw + waves * 2 * bn) * sizeof(float);
}

// PCOPS_OPT_GEMM_SPLIT_BF16 = 0: off, 1 (default): on, 2: only where the weight pieces stay resident (pcops_set_option; the
// environment variable PCOPS_GEMM_BF3 only seeds the table)
static int ws_bf3_mode() { return pcops_get_option(PCOPS_OPT_GEMM_SPLIT_BF16); }

// PCOPS_OPT_DGRAD_SPLIT_BF16 = 0: the data gradients stay on the fp32 pipe, 1 (default): split operands in 64-column passes
// for K >= 128 dY columns, 2: 128-column passes where the weight pieces fit (kernel A/B)
static int dgrad_bf3_mode() { return pcops_get_option(PCOPS_OPT_DGRAD_SPLIT_BF16); }
static int dgrad_bf3_kmin() {
    static const int v = [] { const char *e = getenv("PCOPS_DGRAD_BF3_KMIN"); return e ? atoi(e) : 128; }();
    return v;
}

static int ws_ncoef(int am) { return am == A_PLAIN ? 0 : (am == A_BNRELU ? 2 : (am == A_DY ? 3 : (am == A_XYZ ? 6 : 5))); }

static size_t ws_lds_bytes_streamed(int Kp, int kc, int bn, int waves, int eh, int ncoef) {
    const int ldw = (kc > bn / eh ? kc : bn / eh) + 4;
    const int crows = ncoef > 0 ? ncoef : 1;
    return (size_t)(2 * kc * bn + crows * Kp + 6 * bn + waves * 32 * ldw + waves * 2 * bn) * sizeof(float);
}

static size_t ws_lds_bytes(int Kp, int kc, int bn, int waves, int eh) {
    // weights + 5 coefficient vectors + 2 epilogue vectors + wave stripes + statistics scratch
    const int ldw = (kc > bn / eh ? kc : bn / eh) + 4;
    return (size_t)(Kp * bn + 6 * Kp + 6 * bn + waves * 32 * ldw + waves * 2 * bn) * sizeof(float);
}

static bool ws_stream256_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_WS_STREAM256");
        return !(e && e[0] == '0');
    }();
    return on;
}

static bool ws_mask_bn64_enabled() {
    static const bool on = [] { const char *e = getenv("PCOPS_WS_MASK_BN64"); return !(e && e[0] == '0'); }();   // kernel A/B only
    return on;
}

static bool ws_n96_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_WS_N96");
        return !(e && e[0] == '0');
    }();
    return on;
}

// kind 1: the launch is a forward product (E_FWD epilogue); kind 2: a masked data gradient (E_MASK epilogue, dY operand).
// Round 4 built the split-operand form for the forward products only: the data gradients' 128-column variants need a
// second accumulator set beside two prefetched operands (they spilled) and streamed weights at K = 256.  Round 6: the data
// gradients of the layers that are matrix-pipe bound (K >= 128 dY columns) take the split form in 64-column passes --
// 32 + 32 accumulator registers, and the weight pieces of 64 columns stay RESIDENT up to K = 256 (96 KB): no streaming, no
// workgroup barrier; the two column blocks of a row group run on one XCD, so the second reader of a dY stripe hits its L2
static int dgrad_top_bf3_cols() {
    static const int v = [] {
        const char *e = getenv("PCOPS_DGRAD_TOP_BF3");
        return e ? atoi(e) : 128;
    }();
    return v;
}
// what ws_plan is asked for: 1 forward, 2 masked data gradient, 3 its xyz form, 4 / 5 the algebraic top-layer forms (masked / plain)
constexpr int ws_kind(int em) {
    return em == E_FWD ? 1 : (em == E_MASK ? 2 : (em == E_MASKA ? 4 : (em == E_PLAINA ? 5 : (is_mask(em) ? 3 : 0))));
}
static bool ws_plan(const GemmArgs &a, int am, WsPlan *pl, int kind = 0) {
    const bool fwd = kind == 1;
    if (a.M < 8 * 1024) return false;                        // small problems: the tiled kernel is fine
    if (a.K % 8 != 0 || a.K > 4096 || a.ldx % 4 != 0 || a.N % 4 != 0 || a.ldy % 4 != 0) return false;
    if (a.K > 256 && (am == A_XYZ || (reinterpret_cast<uintptr_t>(a.W) & 15))) return false;
    if ((reinterpret_cast<uintptr_t>(a.X) & 15) || (reinterpret_cast<uintptr_t>(a.X2) & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(a.Y) & 15) || (reinterpret_cast<uintptr_t>(a.Yprev) & 15)) return false;
    if (is_pool(am) && ((reinterpret_cast<uintptr_t>(a.gpool) & 15) || (reinterpret_cast<uintptr_t>(a.argmax) & 3)))
        return false;
    // pooled operand with arbitrary groups: a 32-row tile may meet at most four of them (kernel: G_)
    if (is_pool(am) && !a.blocks && a.S % 32 != 0 && a.S < 11) return false;
    // 8 waves (two per SIMD) with 64-wide stripes; 128 output columns when the resident weight tile fits
    pl->kc = 64;
    pl->waves = 8;
    const int Kp = (a.K + 63) / 64 * 64;
    pl->bn = 64;
    pl->eh = 1;
    pl->wst = a.K > 256;
    // K = 193..256 with more than 64 output columns: resident weights would only fit 64 columns at a time, i.e. the
    // operand would be streamed from HBM once per 64-column block; streaming the WEIGHTS (from L2) keeps 128 columns
    if (!pl->wst && a.N > 64 && am != A_XYZ && !(reinterpret_cast<uintptr_t>(a.W) & 15) &&
        ws_lds_bytes(Kp, 64, 128, 8, 2) > 160 * 1024 && ws_stream256_enabled())
        pl->wst = true;
    if (pl->wst) {
        const int nc = ws_ncoef(am);
        if (a.N > 64 && ws_lds_bytes_streamed(Kp, 64, 128, 8, 2, nc) <= 160 * 1024) { pl->bn = 128; pl->eh = 2; }
        if (pl->bn == 128 && a.N <= 96 && ws_n96_enabled()) { pl->bn = 96; pl->eh = 3; }
        pl->lds = ws_lds_bytes_streamed(Kp, pl->kc, pl->bn, pl->waves, pl->eh, nc);
    } else {
        if (a.N > 64 && ws_lds_bytes(Kp, 64, 128, 8, 2) <= 160 * 1024) { pl->bn = 128; pl->eh = 2; }
        // 65..96 output columns (MSG's 64 -> 96 -> 128 scale): THREE 32-column accumulator blocks instead of four -- a
        // quarter of the matrix-pipe work of the 128-wide tile was on zero columns; the epilogue then runs three
        // 32-column passes
        if (pl->bn == 128 && a.N <= 96 && ws_n96_enabled()) { pl->bn = 96; pl->eh = 3; }
        pl->lds = ws_lds_bytes(Kp, pl->kc, pl->bn, pl->waves, pl->eh);
    }
    pl->bf3 = false;
    static const int bf3_kmin = [] { const char *e = getenv("PCOPS_GEMM_BF3_KMIN"); return e ? atoi(e) : 0; }();
    static const int bf3_kmax = [] { const char *e = getenv("PCOPS_GEMM_BF3_KMAX"); return e ? atoi(e) : 1 << 30; }();
    if (fwd && ws_bf3_mode() && !(reinterpret_cast<uintptr_t>(a.W) & 3) && a.K >= bf3_kmin && a.K <= bf3_kmax) {
        const int Kp3 = (a.K + 31) / 32 * 32;
        const int nc = ws_ncoef(am);
        const int bn3 = a.N > 96 ? 128 : (a.N > 64 ? (ws_n96_enabled() ? 96 : 128) : 64);
        bool wst3 = ws_lds_bytes_bf3(Kp3, bn3, 8, false, nc) > 160 * 1024;
        if (wst3 && (am == A_XYZ || ws_bf3_mode() == 2)) wst3 = false, pl->bf3 = false;
        else pl->bf3 = true;
        if (pl->bf3 && ws_lds_bytes_bf3(Kp3, bn3, 8, wst3, nc) <= 160 * 1024) {
            pl->kc = 32; pl->bn = bn3; pl->eh = bn3 / 32; pl->wst = wst3;
            pl->lds = ws_lds_bytes_bf3(Kp3, bn3, 8, wst3, nc);
        } else {
            pl->bf3 = false;
        }
    }
    if (kind == 2 && is_dy(am) && dgrad_bf3_mode() && !(reinterpret_cast<uintptr_t>(a.W) & 3) && a.K >= dgrad_bf3_kmin() &&
        a.K <= 256 && a.K % 32 == 0) {
        const int nc = ws_ncoef(am);
        // 64-column passes (two accumulator sets of two blocks); 65..96 output columns as ONE pass of three blocks (MSG's
        // 128 -> 96: 241..256 registers, no spill) instead of a second pass over half-empty columns
        const int bn3 = (a.N > 64 && a.N <= 96 && ws_n96_enabled()) ? 96 : ((a.N > 64 && dgrad_bf3_mode() == 2) ? 128 : 64);
        if (ws_lds_bytes_bf3(a.K, bn3, 8, false, nc) <= 160 * 1024) {
            pl->bf3 = true; pl->kc = 32; pl->bn = bn3; pl->eh = bn3 / 32; pl->wst = false;
            pl->lds = ws_lds_bytes_bf3(a.K, bn3, 8, false, nc);
        }
    }
    // kinds 4, 5: the algebraic top-layer data gradients (E_MASKA / E_PLAINA: acc + addend[rowmap[row]] + vconst).  Their product
    // relu(bn(Yprev)) . (W diag(q) W^T) is K x K with K = 128 .. 1024: split operands, weights resident where 64 columns of all
    // K rows fit and streamed (from L2, one workgroup barrier per 32-row chunk) beyond that.  PCOPS_DGRAD_TOP_BF3 = 0 / 64 / 128
    // (columns per pass; default 128)
    if ((kind == 4 || kind == 5) && dgrad_bf3_mode() && dgrad_top_bf3_cols() && !(reinterpret_cast<uintptr_t>(a.W) & 15) && a.K >= 128 &&
        a.K % 32 == 0) {
        const int nc = ws_ncoef(am);
        // (the masked form's 128-column variant spills: 184 .. 228 bytes of scratch; the plain form's does not)
        const int bn3 = (a.N > 64 && dgrad_top_bf3_cols() == 128 && kind == 5) ? 128 : 64;
        const bool wst3 = ws_lds_bytes_bf3(a.K, bn3, 8, false, nc) > 160 * 1024;
        if (ws_lds_bytes_bf3(a.K, bn3, 8, wst3, nc) <= 160 * 1024) {
            pl->bf3 = true; pl->kc = 32; pl->bn = bn3; pl->eh = bn3 / 32; pl->wst = wst3;
            pl->lds = ws_lds_bytes_bf3(a.K, bn3, 8, wst3, nc);
        }
    }
    // masked data gradients that stay on the fp32 pipe (K > 256, the algebraic top-layer forms): their 128-column variants
    // SPILL -- hipcc reports 148 .. 388 bytes of scratch per lane (36 .. 96 registers) for gemm_ws_kernel<4, *, E_MASK*, 64, ..>,
    // which is what round 5's "1.46 x counted traffic" of SA2's 256 -> 128 data gradient was (scratch stores and reloads are
    // global memory: profiles/r06_pmc_dgrad_f32.txt) -- the 64-column variants (195 .. 240 registers) do not
    if (kind >= 2 && kind != 5 && !pl->bf3 && pl->bn == 128 && ws_mask_bn64_enabled()) {
        pl->bn = 64; pl->eh = 1;
        pl->lds = pl->wst ? ws_lds_bytes_streamed((a.K + 63) / 64 * 64, pl->kc, 64, pl->waves, 1, ws_ncoef(am))
                          : ws_lds_bytes((a.K + 63) / 64 * 64, pl->kc, 64, pl->waves, 1);
    }
    if (pl->lds > 160 * 1024) return false;
    pl->ncb = (a.N + pl->bn - 1) / pl->bn;
    const long long ntiles = (((long long)a.M + 31) / 32 + (a.pool_sub > 1 ? a.pool_sub - 1 : 0)) /
                             (a.pool_sub > 1 ? a.pool_sub : 1);   // units a wave walks: tiles, or pooling groups
    long long want = 256 / pl->ncb;                          // one persistent workgroup per CU over the whole grid
    if (want < 1) want = 1;
    const long long maxg = (ntiles + pl->waves - 1) / pl->waves;
    pl->gy = (int)(want < maxg ? want : maxg);
    if (pl->gy >= 8) pl->gy &= ~7;                             // multiple of 8: see the grid comment in the kernel
    return true;
}

static bool nt_stores_enabled() {
    static const bool on = [] { const char *e = getenv("PCOPS_NT_STORE"); return !(e && e[0] == '0'); }();   // kernel A/B only
    return on;
}
static int nt_for_bytes(long long bytes) { return (nt_stores_enabled() && bytes >= (256ll << 20)) ? 1 : 0; }

template <int AM, int EM>
PCOPS_HIDDEN int launch_gemm_ws(GemmArgs &a, const WsPlan &pl, hipStream_t st) {
    a.nt_out = a.Y ? nt_for_bytes((long long)a.M * a.ldy * 4) : 0;
#define PCOPS_WS_LAUNCH(NT_, EH_)                                                                     \
    do {                                                                                              \
        const bool pool_ = EM == E_FWD && a.pool_sub > 0;                                             \
        auto kern = (pl.wst && pool_) ? gemm_ws_kernel<NT_, AM, EM, 64, 8, EH_, ((AM == A_XYZ || EM != E_FWD) ? 0 : 3)> \
                    : pl.wst ? gemm_ws_kernel<NT_, AM, EM, 64, 8, EH_, (AM == A_XYZ ? 0 : 2)>            \
                    : pool_ ? gemm_ws_kernel<NT_, AM, EM, 64, 8, EH_, (EM == E_FWD ? 1 : 0)>              \
                            : gemm_ws_kernel<NT_, AM, EM, 64, 8, EH_, 0>;                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
            return PCOPS_ERR_LAUNCH;                                                                  \
        a.nrowgrp = pl.gy;                                                                            \
        const int P_ = (a.stats && EM != E_PLAIN && EM != E_PLAINA) ? pcops_mlp_stats_rows(a.M) : 0;  \
        hipLaunchKernelGGL(kern, dim3(pl.gy > P_ ? pl.gy : P_, pl.ncb), dim3(512), pl.lds, st, a);    \
    } while (0)
    pcops_note_pipe(pl.bf3 ? 1 : 0);
    if constexpr (EM == E_FWD || (EM == E_MASK && is_dy(AM)) || has_add(EM)) {
    if (pl.bf3) {
#define PCOPS_WS3_LAUNCH(NT_, EH_)                                                                    \
    do {                                                                                              \
        const bool pool_ = EM == E_FWD && a.pool_sub > 0;                                             \
        auto kern = (pl.wst && pool_) ? gemm_ws_kernel<NT_, AM, EM, 32, 8, EH_, ((AM == A_XYZ || EM != E_FWD) ? 4 : 7)> \
                    : pl.wst ? gemm_ws_kernel<NT_, AM, EM, 32, 8, EH_, (AM == A_XYZ ? 4 : 6)>            \
                    : pool_ ? gemm_ws_kernel<NT_, AM, EM, 32, 8, EH_, (EM == E_FWD ? 5 : 4)>              \
                            : gemm_ws_kernel<NT_, AM, EM, 32, 8, EH_, 4>;                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
            return PCOPS_ERR_LAUNCH;                                                                  \
        a.nrowgrp = pl.gy;                                                                            \
        const int P_ = (a.stats && EM != E_PLAIN && EM != E_PLAINA) ? pcops_mlp_stats_rows(a.M) : 0;  \
        hipLaunchKernelGGL(kern, dim3(pl.gy > P_ ? pl.gy : P_, pl.ncb), dim3(512), pl.lds, st, a);    \
    } while (0)
        if constexpr ((AM == A_BNRELU || AM == A_PLAIN) && EM == E_FWD) {
            if (a.pool_s4 > 0) {     // groups that are not whole tiles: their own instantiations (VAR | 8), split operands only
#define PCOPS_WS3S4_LAUNCH(NT_, EH_)                                                                  \
    do {                                                                                              \
        auto kern = pl.wst ? gemm_ws_kernel<NT_, AM, EM, 32, 8, EH_, 15> : gemm_ws_kernel<NT_, AM, EM, 32, 8, EH_, 13>; \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
            return PCOPS_ERR_LAUNCH;                                                                  \
        a.nrowgrp = pl.gy;                                                                            \
        const int P_ = a.stats ? pcops_mlp_stats_rows(a.M) : 0;                                       \
        hipLaunchKernelGGL(kern, dim3(pl.gy > P_ ? pl.gy : P_, pl.ncb), dim3(512), pl.lds, st, a);    \
    } while (0)
                if (pl.bn == 128) PCOPS_WS3S4_LAUNCH(4, 4);
                else if (pl.bn == 96) PCOPS_WS3S4_LAUNCH(3, 3);
                else PCOPS_WS3S4_LAUNCH(2, 2);
#undef PCOPS_WS3S4_LAUNCH
                return pcops_launch_status();
            }
        }
        if (pl.bn == 128) PCOPS_WS3_LAUNCH(4, 4);
        else if (pl.bn == 96) PCOPS_WS3_LAUNCH(3, 3);
        else PCOPS_WS3_LAUNCH(2, 2);
#undef PCOPS_WS3_LAUNCH
        return pcops_launch_status();
    }
    }
    if (a.pool_s4 > 0) return PCOPS_ERR_UNSUPPORTED;      // (only the split-operand kernels carry that epilogue)
    if (pl.bn == 128) PCOPS_WS_LAUNCH(4, 2);
    else if (pl.bn == 96) PCOPS_WS_LAUNCH(3, 3);
    else PCOPS_WS_LAUNCH(2, 1);
#undef PCOPS_WS_LAUNCH
    return pcops_launch_status();
}

// ---- which build part emits which launcher (see PCOPS_MLP_PART at the top): explicit instantiation in the owning part,
// `extern template` (no implicit instantiation, hence no kernels) in the others
#if PCOPS_MLP_PART >= 0
#if PCOPS_MLP_PART == 1
#define PCOPS_X_ template
#else
#define PCOPS_X_ extern template
#endif
PCOPS_X_ int launch_gemm_ws<A_BNRELU, E_FWD>(GemmArgs &, const WsPlan &, hipStream_t);
#undef PCOPS_X_
#if PCOPS_MLP_PART == 2
#define PCOPS_X_ template
#else
#define PCOPS_X_ extern template
#endif
PCOPS_X_ int launch_gemm_ws<A_DY, E_MASK>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DY, E_PLAIN>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOL, E_MASK>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOL, E_PLAIN>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOLB, E_MASK>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOLB, E_PLAIN>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOLU, E_MASK>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOLU, E_PLAIN>(GemmArgs &, const WsPlan &, hipStream_t);
#undef PCOPS_X_
#if PCOPS_MLP_PART == 3
#define PCOPS_X_ template
#else
#define PCOPS_X_ extern template
#endif
PCOPS_X_ int launch_gemm_ws<A_DY, E_MASKX>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOL, E_MASKX>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOLB, E_MASKX>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_DYPOOLU, E_MASKX>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_PLAIN, E_PLAINA>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_BNRELU, E_MASKA>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_rt<A_BNRELU, E_FWD>(GemmArgs &, hipStream_t);
PCOPS_X_ int launch_gemm_rt<A_PLAIN, E_FWD>(GemmArgs &, hipStream_t);
PCOPS_X_ int launch_gemm_rt<A_DYPOOL, E_MASK>(GemmArgs &, hipStream_t);
PCOPS_X_ int launch_gemm_rt<A_DYPOOL, E_PLAIN>(GemmArgs &, hipStream_t);
PCOPS_X_ int launch_gemm_rt<A_DY, E_MASK>(GemmArgs &, hipStream_t);
PCOPS_X_ int launch_gemm_rt<A_DY, E_PLAIN>(GemmArgs &, hipStream_t);
#undef PCOPS_X_
#if PCOPS_MLP_PART == 6
#define PCOPS_X_ template
#else
#define PCOPS_X_ extern template
#endif
PCOPS_X_ int launch_gemm_ws<A_PLAIN, E_FWD>(GemmArgs &, const WsPlan &, hipStream_t);
PCOPS_X_ int launch_gemm_ws<A_XYZ, E_FWD>(GemmArgs &, const WsPlan &, hipStream_t);
#undef PCOPS_X_
#endif

extern "C" int pcops_mlp_stats_rows(int M);

static bool ws_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_GEMM_WS");
        return !(e && e[0] == '0');
    }();
    return on;
}

// one entry for both variants: the wave-stream kernel when the shape suits it, the tiled kernel otherwise.
// The partial-statistics buffer always has pcops_mlp_stats_rows(M) rows; rows a kernel does not emit are zeroed.
template <int AM, int EM>
static int launch_gemm(GemmArgs &a, hipStream_t st) {
    WsPlan pl;
    if (ws_enabled() && ws_plan(a, AM, &pl, ws_kind(EM))) {
        int rc;
        if (AM == A_DYPOOL && a.blocks) rc = launch_gemm_ws<(AM == A_DYPOOL ? A_DYPOOLB : AM), EM>(a, pl, st);
        else if (AM == A_DYPOOL && a.S % 32 == 0) rc = launch_gemm_ws<(AM == A_DYPOOL ? A_DYPOOLU : AM), EM>(a, pl, st);
        else rc = launch_gemm_ws<AM, EM>(a, pl, st);
        return rc;
    }
    if (a.blocks) return PCOPS_ERR_UNSUPPORTED;      // compacted rows: wave-stream kernels only
    pcops_note_pipe(0);                              // the tiled kernel: fp32 MFMA
    return launch_gemm_rt<AM, EM>(a, st);
}

// ---------------------------------------------------------------------------------------------
// column reduction of partial statistics [P][2][N] -> double [2][N], two stages, deterministic
constexpr int kRedSlices = 32;
namespace {   // (non-template kernels: internal linkage, every build part carries its own copy)

__global__ __launch_bounds__(256) void colreduce_stage1(int P, int N, const float *__restrict__ part,
                                                        double *__restrict__ ws) {
    // grid (ceil(N/32), 2, kRedSlices); thread (c = tid%32, g = tid/32)
    __shared__ double sm[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    const int which = blockIdx.y, slice = blockIdx.z;
    const int per = (P + kRedSlices - 1) / kRedSlices;
    const int p0 = slice * per, p1 = min(P, p0 + per);
    double s = 0.0;
    if (c < N)
        for (int p = p0 + g; p < p1; p += 8) s += (double)part[((long long)p * 2 + which) * N + c];
    sm[g][threadIdx.x & 31] = s;
    __syncthreads();
    if (g == 0 && c < N) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += sm[i][threadIdx.x];
        ws[((long long)slice * 2 + which) * N + c] = t;
    }
}

// forward BN: statistics -> (mean, rstd, scale, shift) and the moving-average update
__global__ __launch_bounds__(256) void bn_finalize_kernel(int N, double R, const double *__restrict__ ws,
                                                          const float *__restrict__ pivot,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float decay, int unbiased,
                                                          float *__restrict__ moving_mean,
                                                          float *__restrict__ moving_var,
                                                          float *__restrict__ mean_o, float *__restrict__ rstd_o,
                                                          float *__restrict__ scale_o, float *__restrict__ shift_o) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < kRedSlices; ++s) {
        s1 += ws[((long long)s * 2 + 0) * N + c];
        s2 += ws[((long long)s * 2 + 1) * N + c];
    }
    // the sums are those of (y - pivot) and (y - pivot)^2: the shift leaves the variance untouched and is added back
    // to the mean; with the pivot near the mean the subtraction below no longer cancels leading digits
    const double dm = s1 / R;
    const double mean = dm + (pivot ? (double)pivot[c] : 0.0);
    double var = s2 / R - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    scale_o[c] = sc;
    shift_o[c] = beta[c] - (float)mean * sc;
    if (moving_mean) {
        const double uv = (unbiased && R > 1.0) ? var * (R / (R - 1.0)) : var;
        moving_mean[c] = decay * moving_mean[c] + (1.f - decay) * (float)mean;
        moving_var[c] = decay * moving_var[c] + (1.f - decay) * (float)uv;
    }
}

// eval-mode BN: moving statistics -> (scale, shift)
__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(int N, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta,
                                                             const float *__restrict__ mm,
                                                             const float *__restrict__ mv, float eps,
                                                             float *__restrict__ scale_o,
                                                             float *__restrict__ shift_o) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const float sc = gamma[c] * (1.0f / sqrtf(mv[c] + eps));
    scale_o[c] = sc;
    shift_o[c] = beta[c] - mm[c] * sc;
}

// backward BN: sums (sum G, sum G*Y) -> dgamma, dbeta and the dY = p*G + q*Y + t coefficient vectors
__global__ __launch_bounds__(256) void bn_bwd_coeffs_kernel(int N, double R, const double *__restrict__ ws,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ rstd,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                            float *__restrict__ p_o, float *__restrict__ q_o,
                                                            float *__restrict__ t_o) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    double sg = 0.0, sgy = 0.0;
    for (int s = 0; s < kRedSlices; ++s) {
        sg += ws[((long long)s * 2 + 0) * N + c];
        sgy += ws[((long long)s * 2 + 1) * N + c];
    }
    const double mu = mean[c], rs = rstd[c], g = gamma[c];
    const double dga = (sgy - mu * sg) * rs;
    const double p = g * rs;
    const double q = -p * rs * dga / R;
    const double t = -p * sg / R - q * mu;
    dgamma[c] = (float)dga;
    dbeta[c] = (float)sg;
    p_o[c] = (float)p;
    q_o[c] = (float)q;
    t_o[c] = (float)t;
}

// ---------------------------------------------------------------------------------------------
// out[g][c] = max_s relu(scale[c]*Y[g*S+s][c] + shift[c]); argmax = first s reaching it
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(long long G, int S, int C,
                                                              const float *__restrict__ Y,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              float *__restrict__ out,
                                                              unsigned char *__restrict__ argmax,
                                                              float *__restrict__ ysel) {
    const int c4n = C / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < G * c4n; e += (long long)gridDim.x * 256) {
        const long long g = e / c4n;
        const int c = (int)(e - g * c4n) * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        float m[4] = {-1.f, -1.f, -1.f, -1.f};
        float ys[4] = {0.f, 0.f, 0.f, 0.f};
        int am[4] = {0, 0, 0, 0};
        const float *base = Y + (g * S) * C + c;
        for (int s = 0; s < S; ++s) {
            const float4 y = *reinterpret_cast<const float4 *>(base + (long long)s * C);
            const float a0 = fmaxf(fmaf(y.x, sc.x, sh.x), 0.f), a1 = fmaxf(fmaf(y.y, sc.y, sh.y), 0.f);
            const float a2 = fmaxf(fmaf(y.z, sc.z, sh.z), 0.f), a3 = fmaxf(fmaf(y.w, sc.w, sh.w), 0.f);
            if (a0 > m[0]) { m[0] = a0; am[0] = s; ys[0] = y.x; }
            if (a1 > m[1]) { m[1] = a1; am[1] = s; ys[1] = y.y; }
            if (a2 > m[2]) { m[2] = a2; am[2] = s; ys[2] = y.z; }
            if (a3 > m[3]) { m[3] = a3; am[3] = s; ys[3] = y.w; }
        }
        *reinterpret_cast<float4 *>(out + g * C + c) = make_float4(m[0], m[1], m[2], m[3]);
        if (argmax) {
            uchar4 q;
            q.x = (unsigned char)am[0]; q.y = (unsigned char)am[1];
            q.z = (unsigned char)am[2]; q.w = (unsigned char)am[3];
            *reinterpret_cast<uchar4 *>(argmax + g * C + c) = q;
        }
        if (ysel) *reinterpret_cast<float4 *>(ysel + g * C + c) = make_float4(ys[0], ys[1], ys[2], ys[3]);
    }
}

// the same over COMPACTED rows: group g owns rows 16 bstart[g] .. 16 bstart[g+1] - 1 (its real members followed by
// copies of member 0 up to the block boundary; the first maximiser wins, so a copy is never the arg-max)
__global__ __launch_bounds__(256) void bn_relu_maxpool_rows_kernel(long long G, int C, const float *__restrict__ Y,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift,
                                                                   const int *__restrict__ bstart,
                                                                   float *__restrict__ out,
                                                                   unsigned char *__restrict__ argmax,
                                                                   float *__restrict__ ysel) {
    const int c4n = C / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < G * c4n; e += (long long)gridDim.x * 256) {
        const long long g = e / c4n;
        const int c = (int)(e - g * c4n) * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        float m[4] = {-1.f, -1.f, -1.f, -1.f};
        float ys[4] = {0.f, 0.f, 0.f, 0.f};
        int am[4] = {0, 0, 0, 0};
        const long long r0 = (long long)bstart[g] * kBlk;
        const int S = (bstart[g + 1] - bstart[g]) * kBlk;
        const float *base = Y + r0 * C + c;
        // S is a multiple of 16: eight rows are requested before the first is looked at (a row-at-a-time loop has ONE
        // 16-byte load in flight per lane and spends 93 % of its wave cycles in s_waitcnt)
        for (int s0 = 0; s0 < S; s0 += 8) {
            float4 yv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) yv[u] = *reinterpret_cast<const float4 *>(base + (long long)(s0 + u) * C);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 y = yv[u];
                const int s = s0 + u;
                const float a0 = fmaxf(fmaf(y.x, sc.x, sh.x), 0.f), a1 = fmaxf(fmaf(y.y, sc.y, sh.y), 0.f);
                const float a2 = fmaxf(fmaf(y.z, sc.z, sh.z), 0.f), a3 = fmaxf(fmaf(y.w, sc.w, sh.w), 0.f);
                if (a0 > m[0]) { m[0] = a0; am[0] = s; ys[0] = y.x; }
                if (a1 > m[1]) { m[1] = a1; am[1] = s; ys[1] = y.y; }
                if (a2 > m[2]) { m[2] = a2; am[2] = s; ys[2] = y.z; }
                if (a3 > m[3]) { m[3] = a3; am[3] = s; ys[3] = y.w; }
            }
        }
        *reinterpret_cast<float4 *>(out + g * C + c) = make_float4(m[0], m[1], m[2], m[3]);
        if (argmax) {
            uchar4 q;
            q.x = (unsigned char)am[0]; q.y = (unsigned char)am[1];
            q.z = (unsigned char)am[2]; q.w = (unsigned char)am[3];
            *reinterpret_cast<uchar4 *>(argmax + g * C + c) = q;
        }
        if (ysel) *reinterpret_cast<float4 *>(ysel + g * C + c) = make_float4(ys[0], ys[1], ys[2], ys[3]);
    }
}

// compacted rows: per group the best of its blocks' partial extrema (written by the fused epilogue per 16-row block:
// the selected raw value and its row-in-group); sign(gamma) decides between max and min, the first block wins ties
// (blocks are in row order) -> ysel, arg-max, out = relu(scale * ysel + shift)
__global__ __launch_bounds__(256) void pool_combine_rows_kernel(long long total, int C, const float *__restrict__ ypart,
                                                                const unsigned char *__restrict__ ppart,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift,
                                                                const int *__restrict__ bstart, float *__restrict__ out,
                                                                unsigned char *__restrict__ argmax,
                                                                float *__restrict__ ysel) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long g = e / C;
        const int c = (int)(e - g * C);
        const float sg = gamma[c] < 0.f ? -1.f : 1.f;
        const int b0 = bstart[g], b1 = bstart[g + 1];
        float best = ypart[(long long)b0 * C + c];
        int arg = ppart[(long long)b0 * C + c];
        for (int bk = b0 + 1; bk < b1; ++bk) {
            const float y = ypart[(long long)bk * C + c];
            if (y * sg > best * sg) { best = y; arg = ppart[(long long)bk * C + c]; }
        }
        out[e] = fmaxf(fmaf(best, scale[c], shift[c]), 0.f);
        if (argmax) argmax[e] = (unsigned char)arg;
        if (ysel) ysel[e] = best;
    }
}

// pooled output from the fused epilogue's selected raw values: out = relu(scale * ysel + shift)
__global__ __launch_bounds__(256) void pool_select_kernel(long long total, int C, const float *__restrict__ ysel,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        out[e] = fmaxf(fmaf(ysel[e], scale[c], shift[c]), 0.f);
    }
}

// the two BN-backward sums of a max-pooled layer from (gpool, y at the selected row): stats [P][2][C];
// one workgroup per `groups_per_block` groups, threads = (row lane, column quad), 16-byte loads
__global__ __launch_bounds__(256) void pool_bwd_stats_sel_kernel(long long G, int C, const float *__restrict__ gpool,
                                                                 const float *__restrict__ ysel,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 float *__restrict__ stats, int groups_per_block,
                                                                 float *__restrict__ gmasked) {
    extern __shared__ float sm[];                     // [RL][2][C]
    const int c4n = C / 4;
    const int RL = c4n >= 256 ? 1 : 256 / c4n;
    const int rl = c4n >= 256 ? 0 : threadIdx.x / c4n;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    for (int cq = threadIdx.x % (c4n >= 256 ? 256 : c4n); cq < c4n; cq += 256) {
        const int c = cq * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < RL)
            for (long long g = g0 + rl; g < g1; g += RL) {
                const float4 y = *reinterpret_cast<const float4 *>(ysel + g * C + c);
                const float4 gp = *reinterpret_cast<const float4 *>(gpool + g * C + c);
                const float gm0 = fmaf(y.x, sc.x, sh.x) > 0.f ? gp.x : 0.f;
                const float gm1 = fmaf(y.y, sc.y, sh.y) > 0.f ? gp.y : 0.f;
                const float gm2 = fmaf(y.z, sc.z, sh.z) > 0.f ? gp.z : 0.f;
                const float gm3 = fmaf(y.w, sc.w, sh.w) > 0.f ? gp.w : 0.f;
                if (gmasked) *reinterpret_cast<float4 *>(gmasked + g * C + c) = make_float4(gm0, gm1, gm2, gm3);
                a1[0] += gm0; a1[1] += gm1; a1[2] += gm2; a1[3] += gm3;
                a2[0] = fmaf(gm0, y.x, a2[0]); a2[1] = fmaf(gm1, y.y, a2[1]);
                a2[2] = fmaf(gm2, y.z, a2[2]); a2[3] = fmaf(gm3, y.w, a2[3]);
            }
        if (rl < RL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sm[(rl * 2 + 0) * C + c + e] = a1[e];
                sm[(rl * 2 + 1) * C + c + e] = a2[e];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        float t = 0.f;
        for (int l = 0; l < RL; ++l) t += sm[(l * 2 + which) * C + c];
        stats[((long long)blockIdx.x * 2 + which) * C + c] = t;
    }
}

// ... of a pooled gradient that arrives in TWO pieces and / or strided (an EdgeConv output that feeds the next layer and a column
// block of the concatenation, dgcnn.py:39-81: autograd's sum was a 134 MB launch of its own, a strided piece a copy): the pieces
// are added while the sums are taken and leave as ONE contiguous tensor gsum [G][C] for the data-gradient kernel behind
__global__ __launch_bounds__(256) void pool_bwd_stats_sum_kernel(long long G, int C, const float *__restrict__ ga, long long lda,
                                                                 const float *__restrict__ gb, long long ldb,
                                                                 const float *__restrict__ ysel,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 float *__restrict__ stats, int groups_per_block,
                                                                 float *__restrict__ gsum) {
    extern __shared__ float sm[];                     // [RL][2][C]
    const int c4n = C / 4;
    const int RL = c4n >= 256 ? 1 : 256 / c4n;
    const int rl = c4n >= 256 ? 0 : threadIdx.x / c4n;
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    for (int cq = threadIdx.x % (c4n >= 256 ? 256 : c4n); cq < c4n; cq += 256) {
        const int c = cq * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < RL)
            for (long long g = g0 + rl; g < g1; g += RL) {
                const float4 y = *reinterpret_cast<const float4 *>(ysel + g * C + c);
                float4 gp = *reinterpret_cast<const float4 *>(ga + g * lda + c);
                if (gb) {
                    const float4 g2 = *reinterpret_cast<const float4 *>(gb + g * ldb + c);
                    gp.x += g2.x; gp.y += g2.y; gp.z += g2.z; gp.w += g2.w;
                }
                *reinterpret_cast<float4 *>(gsum + g * C + c) = gp;
                const float gm0 = fmaf(y.x, sc.x, sh.x) > 0.f ? gp.x : 0.f;
                const float gm1 = fmaf(y.y, sc.y, sh.y) > 0.f ? gp.y : 0.f;
                const float gm2 = fmaf(y.z, sc.z, sh.z) > 0.f ? gp.z : 0.f;
                const float gm3 = fmaf(y.w, sc.w, sh.w) > 0.f ? gp.w : 0.f;
                a1[0] += gm0; a1[1] += gm1; a1[2] += gm2; a1[3] += gm3;
                a2[0] = fmaf(gm0, y.x, a2[0]); a2[1] = fmaf(gm1, y.y, a2[1]);
                a2[2] = fmaf(gm2, y.z, a2[2]); a2[3] = fmaf(gm3, y.w, a2[3]);
            }
        if (rl < RL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sm[(rl * 2 + 0) * C + c + e] = a1[e];
                sm[(rl * 2 + 1) * C + c + e] = a2[e];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        float t = 0.f;
        for (int l = 0; l < RL; ++l) t += sm[(l * 2 + which) * C + c];
        stats[((long long)blockIdx.x * 2 + which) * C + c] = t;
    }
}

// A[r][c] = relu(scale[c]*Y[r][c] + shift[c])   (stack output without pooling)
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(long long total4, int C,
                                                            const float *__restrict__ Y,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int c = (int)((e * 4) % C);
        const float4 y = reinterpret_cast<const float4 *>(Y)[e];
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        float4 o;
        o.x = fmaxf(fmaf(y.x, sc.x, sh.x), 0.f);
        o.y = fmaxf(fmaf(y.y, sc.y, sh.y), 0.f);
        o.z = fmaxf(fmaf(y.z, sc.z, sh.z), 0.f);
        o.w = fmaxf(fmaf(y.w, sc.w, sh.w), 0.f);
        reinterpret_cast<float4 *>(out)[e] = o;
    }
}

// backward of relu(bn(Y)) for a materialised upstream grad Gout: Gm = Gout*[a>0], partial sums
// (sum Gm, sum Gm*Y) per 128-row tile: stats [P][2][C]
__global__ __launch_bounds__(256) void relu_mask_stats_kernel(long long R, int C, const float *__restrict__ Gout,
                                                              const float *__restrict__ Y,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              float *__restrict__ Gm, float *__restrict__ stats,
                                                              int rows_per_block) {
    // block: 256 threads = (C/4 column-quads) x (256/(C/4)) row lanes ... generic: thread owns column quad
    // cq = tid % c4n and strides rows by 256 / c4n; requires c4n <= 256 and 256 % c4n == 0 or loops columns.
    extern __shared__ float sm[];  // [rl][2][C]
    const int c4n = C / 4;
    const int rl = max(1, 256 / c4n);  // row lanes
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    for (int cq = threadIdx.x % min(c4n, 256); cq < c4n; cq += 256) {
        const int c = cq * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        const int lane_r = (c4n >= 256) ? 0 : threadIdx.x / c4n;
        if (lane_r < rl) {
            for (long long r = r0 + lane_r; r < r1; r += rl) {
                const float4 g = *reinterpret_cast<const float4 *>(Gout + r * C + c);
                const float4 y = *reinterpret_cast<const float4 *>(Y + r * C + c);
                float4 o;
                o.x = fmaf(y.x, sc.x, sh.x) > 0.f ? g.x : 0.f;
                o.y = fmaf(y.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
                o.z = fmaf(y.z, sc.z, sh.z) > 0.f ? g.z : 0.f;
                o.w = fmaf(y.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
                *reinterpret_cast<float4 *>(Gm + r * C + c) = o;
                a1[0] += o.x; a1[1] += o.y; a1[2] += o.z; a1[3] += o.w;
                a2[0] = fmaf(o.x, y.x, a2[0]); a2[1] = fmaf(o.y, y.y, a2[1]);
                a2[2] = fmaf(o.z, y.z, a2[2]); a2[3] = fmaf(o.w, y.w, a2[3]);
            }
            for (int e = 0; e < 4; ++e) {
                sm[(lane_r * 2 + 0) * C + c + e] = a1[e];
                sm[(lane_r * 2 + 1) * C + c + e] = a2[e];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        float t = 0.f;
        for (int l = 0; l < rl; ++l) t += sm[(l * 2 + which) * C + c];
        stats[((long long)blockIdx.x * 2 + which) * C + c] = t;
    }
}

}  // namespace
// ---------------------------------------------------------------------------------------------
// wgrad: dW[K][N] (+)= A^T dY over a slice of rows; both operands rebuilt from raw tensors and loaded
// fragment-shaped straight from HBM (lane = (row parity, channel group); VK/VN consecutive channels per
// lane feed VK*VN MFMAs whose row/column labels are the permuted channels 32-strided... see header).
struct WgradArgs {
    long long M;
    int K, N;
    int rows_per_block;
    int amode;               // A_PLAIN | A_BNRELU | A_XYZ for the A^T side
    const float *X; int ldx; const float *asc; const float *ash;
    const float *off4; const float *xw; int xw_ld;   // A_XYZ: see the GEMM's A_XYZ
    int dmode;               // A_DY | A_DYPOOL for the dY side
    const float *G; const float *Y; int ldy;
    const float *p; const float *q; const float *t;
    const float *dsc; const float *dsh; const float *gpool; const unsigned char *argmax; int S;
    float *part;             // [gridDim.z][K][N] partial dW
    float *dbpart;           // [gridDim.z][N] partial db (written by blockIdx.x == 0)
    const RowBlock *blocks;  // compacted rows (producer/consumer kernel only), see GemmArgs
    const int *Mdev;
    // bwd_fused_kernel only: the layer's weights, the masked data gradient it also writes, its column statistics
    const float *W; float *Gprev; float *gstats; float *xstats;
    int nt_out;              // Gprev leaves with non-temporal stores
    const float *side;       // SIDE: [M][8] per-row inputs (six used) of the reduced first layer below
    // GW (bwd_fused_kernel, pooled layers): the weight gradient in its Gram form -- part receives X^T (p.G) (the arg rows
    // only), gram_part [groups][K][K] the workgroup's X^T X, xsum_part [groups][K] its X^T 1
    float *gram_part; float *xsum_part;
};

template <int VK, int VN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    // block tile: channels k0..k0+32*VK of A, n0..n0+32*VN of dY; 4 waves take interleaved row pairs
    __shared__ float red[32 * VK * 32 * VN];
    __shared__ float redb[32 * VN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 32 * VK, n0 = blockIdx.y * 32 * VN;
    const int K = a.K, N = a.N;
    const long long r0 = (long long)blockIdx.z * a.rows_per_block;
    const long long r1 = min(a.M, r0 + a.rows_per_block);
    const int half = lane >> 5, li = lane & 31;

    // per-lane channel constants
    float asc[VK], ash[VK], cp[VN], cq[VN], ct[VN], dsc[VN], dsh[VN];
    bool kin[VK], nin[VN];
#pragma unroll
    for (int e = 0; e < VK; ++e) {
        const int k = k0 + li * VK + e;
        kin[e] = k < K;
        asc[e] = (a.amode == A_BNRELU && kin[e]) ? a.asc[k] : 1.f;
        ash[e] = (a.amode == A_BNRELU && kin[e]) ? a.ash[k] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        const int n = n0 + li * VN + e;
        nin[e] = n < N;
        cp[e] = nin[e] ? a.p[n] : 0.f;
        cq[e] = nin[e] ? a.q[n] : 0.f;
        ct[e] = nin[e] ? a.t[n] : 0.f;
        dsc[e] = (a.dmode == A_DYPOOL && nin[e]) ? a.dsc[n] : 0.f;
        dsh[e] = (a.dmode == A_DYPOOL && nin[e]) ? a.dsh[n] : 0.f;
    }

    f32x16 acc[VK][VN];
#pragma unroll
    for (int i = 0; i < VK; ++i)
#pragma unroll
        for (int j = 0; j < VN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    float dbs[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) dbs[e] = 0.f;

    for (long long rb = r0 + 2 * wave; rb < r1; rb += 8) {
        // wave-uniform trip count: lanes 0-31 take row rb, lanes 32-63 row rb+1 of the pair
        const long long r = rb + half;
        const bool rin = r < r1;
        float av[VK], dv[VN];
#pragma unroll
        for (int e = 0; e < VK; ++e) {
            float x = 0.f;
            if (rin && kin[e]) {
                x = a.X[r * a.ldx + k0 + li * VK + e];
                if (a.amode == A_BNRELU) x = fmaxf(fmaf(x, asc[e], ash[e]), 0.f);
            }
            av[e] = x;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float d = 0.f;
            if (rin && nin[e]) {
                const int n = n0 + li * VN + e;
                const float y = a.Y[r * a.ldy + n];
                float gm;
                if (a.dmode == A_DY) {
                    gm = a.G[r * a.ldy + n];
                } else {
                    const long long g = (long long)((unsigned)r / (unsigned)a.S);   // rows < 2^31 (launcher)
                    const int s = (int)((unsigned)r - (unsigned)g * (unsigned)a.S);
                    gm = (a.argmax[g * N + n] == s && fmaf(y, dsc[e], dsh[e]) > 0.f) ? a.gpool[g * N + n] : 0.f;
                }
                d = fmaf(cp[e], gm, fmaf(cq[e], y, ct[e]));
            }
            dv[e] = d;
            dbs[e] += d;
        }
#pragma unroll
        for (int i = 0; i < VK; ++i)
#pragma unroll
            for (int j = 0; j < VN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], dv[j], acc[i][j], 0, 0, 0);
    }

    // acc[i][j][v]: A-channel k0 + VK*(row label) + i, row label = (v&3) + 8 (v>>2) + 4 half;
    //               dY-channel n0 + VN*(lane&31) + j
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < VK; ++i)
#pragma unroll
                for (int j = 0; j < VN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int kl = ((v & 3) + 8 * (v >> 2) + 4 * half) * VK + i;
                        const int nl = li * VN + j;
                        float *dst = &red[kl * 32 * VN + nl];
                        *dst = (w == 0 ? 0.f : *dst) + acc[i][j][v];
                    }
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float s = dbs[e] + __shfl_xor(dbs[e], 32, 64);
                if (half == 0) redb[li * VN + e] = (w == 0 ? 0.f : redb[li * VN + e]) + s;
            }
        }
    }
    __syncthreads();
    float *out = a.part + (long long)blockIdx.z * K * N;
    for (int i = tid; i < 32 * VK * 32 * VN; i += 256) {
        const int kl = i / (32 * VN), nl = i % (32 * VN);
        if (k0 + kl < K && n0 + nl < N) out[(long long)(k0 + kl) * N + n0 + nl] = red[i];
    }
    if (blockIdx.x == 0 && a.dbpart)
        for (int i = tid; i < 32 * VN; i += 256)
            if (n0 + i < N) a.dbpart[(long long)blockIdx.z * N + n0 + i] = redb[i];
}

// ---------------------------------------------------------------------------------------------
// wgrad_ws: wave-stream weight gradient.  Same operand contract as wgrad_kernel, different machine mapping:
// persistent workgroups, every wave owns its own row stripes and a PRIVATE (32 TK x 32 TN) accumulator tile;
// the stripes of all three source tensors are prefetched one stripe ahead with 16-byte loads, transformed
// (BN+ReLU on the A side, dY = p.G + q.Y + t on the other) on their way into wave-private LDS, and read back
// as fragments with conflict-free ds_read_b32 (lane = channel, half-wave = row parity).  No workgroup barrier
// in the main loop; coefficient vectors live in LDS; the four wave tiles are summed through LDS at the end
// and each workgroup writes ONE partial tile (summed deterministically by sum_partials_kernel).
template <int TK, int TN, int RS, int AMODE, int DMODE>
__global__ __launch_bounds__(256, 1) void wgrad_ws_kernel(WgradArgs a) {
    constexpr int KB = 32 * TK, NB = 32 * TN;
    constexpr int A4 = KB / 4, D4 = NB / 4;            // float4 per stripe row
    constexpr int NA = RS * A4 / 64, ND = RS * D4 / 64; // float4 per lane per stripe
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: keeps the buffer descriptors wave-uniform
    const int K = a.K, N = a.N;
    const long long M = a.M;
    // grid = (row groups, K blocks, N blocks): the (K,N) blocks of one row group sit on the same XCD (linear ids
    // differ by multiples of 8) so that re-reads of a stripe by a sibling block hit that XCD's L2
    const int k0 = blockIdx.y * KB, n0 = blockIdx.z * NB;
    const int grp = blockIdx.x, ngrp = gridDim.x;
    float *coefA = lds;                       // [2][KB]  scale, shift (A side)
    float *coefD = coefA + 2 * KB;            // [5][NB]  p, q, t, pool scale, pool shift
    float *As = coefD + 5 * NB + wave * RS * (KB + NB);   // [RS][KB]
    float *Ds = As + RS * KB;                 // [RS][NB]
    float *red = coefD + 5 * NB + 4 * RS * (KB + NB);     // [KB][NB] + [NB]

    for (int e = tid; e < KB; e += 256) {
        const int k = k0 + e;
        const bool in = k < K;
        coefA[e] = (AMODE == A_BNRELU && in) ? a.asc[k] : 0.f;
        coefA[KB + e] = (AMODE == A_BNRELU && in) ? a.ash[k] : 0.f;
    }
    for (int e = tid; e < NB; e += 256) {
        const int n = n0 + e;
        const bool in = n < N;
        coefD[e] = (in && a.p) ? a.p[n] : 0.f;
        coefD[NB + e] = (in && a.q) ? a.q[n] : 0.f;
        coefD[2 * NB + e] = (in && a.t) ? a.t[n] : 0.f;
        coefD[3 * NB + e] = (DMODE == A_DYPOOL && in) ? a.dsc[n] : 0.f;
        coefD[4 * NB + e] = (DMODE == A_DYPOOL && in) ? a.dsh[n] : 0.f;
    }
    __syncthreads();

    // fixed per-lane stripe coordinates: element e = lane + 64 j -> (row e / X4, column quad e % X4)
    const int acq = (lane % A4) * 4, dcq = (lane % D4) * 4;
    const bool ain = k0 + acq < K, din = n0 + dcq < N;        // K % 4 == 0 and N % 4 == 0 (launcher)
    const int acl = ain ? k0 + acq : 0, dcl = din ? n0 + dcq : 0;
    const float4 casc = *reinterpret_cast<const float4 *>(&coefA[acq]);
    const float4 cash = *reinterpret_cast<const float4 *>(&coefA[KB + acq]);
    const float4 cp = *reinterpret_cast<const float4 *>(&coefD[dcq]);
    const float4 cq = *reinterpret_cast<const float4 *>(&coefD[NB + dcq]);
    const float4 ct = *reinterpret_cast<const float4 *>(&coefD[2 * NB + dcq]);
    const float4 cds = *reinterpret_cast<const float4 *>(&coefD[3 * NB + dcq]);
    const float4 cdh = *reinterpret_cast<const float4 *>(&coefD[4 * NB + dcq]);

    f32x16 acc[TK][TN];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};

    float4 px[NA], pg[ND], py[ND];
    unsigned pm[(DMODE == A_DYPOOL) ? ND : 1];
    const long long nstripes = (M + RS - 1) / RS;
    const long long sstride = (long long)ngrp * 4;

    // one 32-bit lane offset per tensor + a scalar offset per access; rows beyond M fail the hardware bounds check
    const unsigned xvoff = ain ? (unsigned)((lane / A4) * a.ldx + acl) * 4u : kOOB;
    const unsigned dvoff = din ? (unsigned)((lane / D4) * a.ldy + dcl) * 4u : kOOB;
    const unsigned xstep = (unsigned)(64 / A4) * (unsigned)a.ldx * 4u, dstep = (unsigned)(64 / D4) * (unsigned)a.ldy * 4u;
    const long long glast = DMODE == A_DYPOOL ? (M - 1) / a.S : 0;
    auto issue = [&](long long stripe) {
        const long long row0 = stripe * RS;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + row0 * a.ldx, (M - row0) * a.ldx * 4);
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.Y + row0 * a.ldy, (M - row0) * a.ldy * 4);
        const __amdgpu_buffer_rsrc_t rg = make_rsrc((DMODE == A_DYPOOL ? a.Y : a.G) + row0 * a.ldy, (M - row0) * a.ldy * 4);
#pragma unroll
        for (int j = 0; j < NA; ++j) px[j] = buf_load4(rx, xvoff, (unsigned)j * xstep);
        const PoolRows pr(DMODE == A_DYPOOL ? row0 : 0, DMODE == A_DYPOOL ? a.S : 1);
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            py[j] = buf_load4(ry, dvoff, (unsigned)j * dstep);
            if (DMODE == A_DYPOOL) {
                long long gi;
                unsigned sdummy;
                pr.split((lane + 64 * j) / D4, glast, gi, sdummy);
                pg[j] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                pm[j] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
            } else {
                pg[j] = buf_load4(rg, dvoff, (unsigned)j * dstep);
            }
        }
    };
    auto stage = [&](long long stripe) {
        const long long row0 = stripe * RS;
        const PoolRows prs(DMODE == A_DYPOOL ? row0 : 0, DMODE == A_DYPOOL ? a.S : 1);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int r = (lane + 64 * j) / A4;
            float4 x = px[j];
            if (AMODE == A_BNRELU) {
                x.x = fmaxf(fmaf(x.x, casc.x, cash.x), 0.f);
                x.y = fmaxf(fmaf(x.y, casc.y, cash.y), 0.f);
                x.z = fmaxf(fmaf(x.z, casc.z, cash.z), 0.f);
                x.w = fmaxf(fmaf(x.w, casc.w, cash.w), 0.f);
            }
            if (!(ain && row0 + r < M)) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&As[r * KB + acq]) = x;
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int r = (lane + 64 * j) / D4;
            const float4 y = py[j];
            float4 g = pg[j];
            if (DMODE == A_DYPOOL) {
                long long gdummy;
                unsigned s;
                prs.split(r, glast, gdummy, s);
                const unsigned am = pm[j];
                g.x = ((am & 0xffu) == s && fmaf(y.x, cds.x, cdh.x) > 0.f) ? g.x : 0.f;
                g.y = (((am >> 8) & 0xffu) == s && fmaf(y.y, cds.y, cdh.y) > 0.f) ? g.y : 0.f;
                g.z = (((am >> 16) & 0xffu) == s && fmaf(y.z, cds.z, cdh.z) > 0.f) ? g.z : 0.f;
                g.w = ((am >> 24) == s && fmaf(y.w, cds.w, cdh.w) > 0.f) ? g.w : 0.f;
            }
            float4 d;
            d.x = fmaf(cp.x, g.x, fmaf(cq.x, y.x, ct.x));
            d.y = fmaf(cp.y, g.y, fmaf(cq.y, y.y, ct.y));
            d.z = fmaf(cp.z, g.z, fmaf(cq.z, y.z, ct.z));
            d.w = fmaf(cp.w, g.w, fmaf(cq.w, y.w, ct.w));
            if (!(din && row0 + r < M)) d = make_float4(0.f, 0.f, 0.f, 0.f);
            dbs[0] += d.x; dbs[1] += d.y; dbs[2] += d.z; dbs[3] += d.w;
            *reinterpret_cast<float4 *>(&Ds[r * NB + dcq]) = d;
        }
    };

    long long stripe = (long long)grp * 4 + wave;
    if (stripe < nstripes) issue(stripe);
    const int half = lane >> 5, li = lane & 31;
    for (; stripe < nstripes; stripe += sstride) {
        __builtin_amdgcn_wave_barrier();
        stage(stripe);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (stripe + sstride < nstripes) issue(stripe + sstride);
        float av_n[TK], dv_n[TN];
#pragma unroll
        for (int i = 0; i < TK; ++i) av_n[i] = As[half * KB + 32 * i + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) dv_n[j] = Ds[half * NB + 32 * j + li];
#pragma unroll 2
        for (int it = 0; it < RS / 2; ++it) {
            float av[TK], dv[TN];
#pragma unroll
            for (int i = 0; i < TK; ++i) av[i] = av_n[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) dv[j] = dv_n[j];
            if (it + 1 < RS / 2) {       // fragments of the next row pair are requested before this pair's MFMAs
#pragma unroll
                for (int i = 0; i < TK; ++i) av_n[i] = As[(2 * (it + 1) + half) * KB + 32 * i + li];
#pragma unroll
                for (int j = 0; j < TN; ++j) dv_n[j] = Ds[(2 * (it + 1) + half) * NB + 32 * j + li];
            }
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], dv[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- sum the four wave tiles (and db) through LDS, write this workgroup's partial
    // acc[i][j][v]: A channel 32 i + (v&3) + 8 (v>>2) + 4 half, dY channel 32 j + li
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float *dst = &red[(32 * i + (v & 3) + 8 * (v >> 2) + 4 * half) * NB + 32 * j + li];
                        *dst = (w == 0 ? 0.f : *dst) + acc[i][j][v];
                    }
            // lanes l, l + D4, ... own the same 4 dY columns
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = dbs[e];
#pragma unroll
                for (int off = 32; off >= D4; off >>= 1) s += __shfl_xor(s, off, 64);
                if (lane < D4) red[KB * NB + dcq + e] = (w == 0 ? 0.f : red[KB * NB + dcq + e]) + s;
            }
        }
    }
    __syncthreads();
    float *out = a.part + (long long)grp * K * N;
    for (int i = tid; i < KB * NB; i += 256) {
        const int kl = i / NB, nl = i % NB;
        if (k0 + kl < K && n0 + nl < N) out[(long long)(k0 + kl) * N + n0 + nl] = red[i];
    }
    if (blockIdx.y == 0 && a.dbpart)
        for (int i = tid; i < NB; i += 256)
            if (n0 + i < N) a.dbpart[(long long)grp * N + n0 + i] = red[KB * NB + i];
}

// ---------------------------------------------------------------------------------------------
// wgrad, producer / consumer variant (the default for large M).  The single-role kernel above alternates between
// rebuilding operands (VALU + LDS writes) and MFMAs inside one wave per SIMD, so the matrix pipe idles while
// operands are staged.  Here a workgroup is 8 waves = 2 per SIMD with fixed roles:
//   waves 4..7  PRODUCERS: HBM -> registers (one stripe ahead) -> BN/ReLU resp. dY rebuild -> LDS stripe [RS][KB+NB]
//   waves 0..3  CONSUMERS: LDS fragments -> MFMA, nothing else; consumer (ck, cn) owns the (TK x TN) x 32x32 sub-block
//               (ck, cn) of the workgroup's KB x NB output tile, so no cross-wave reduction is needed at the end
// Stripes are double buffered and handed over with ONE workgroup barrier per stripe.  Per 32-row stripe a consumer
// issues 16 TK TN MFMAs (64 cycles each) while the producers need a few hundred VALU cycles: the kernel is MFMA
// bound for K >= 128 and HBM bound for the 64-wide layers.
// K96 (65..96 input channels against 65..128 gradient columns -- MSG's 96 -> 128 layer): the staged tile keeps its 128 +
// 128 columns (the producers' lane mapping wants 256 % (KB / 4) == 0; channels >= K are never requested and arrive as
// zeros), but the four consumers split the OUTPUT as 96 x 32 each (three A blocks against one dY block) instead of
// 64 x 64: no wave multiplies the empty fourth block -- 3 MFMAs per row pair and wave instead of 4.
template <int TK, int TN, int AMODE, int DMODE, bool K96 = false>
__global__ __launch_bounds__(512, (TK * TN == 1) ? 2 : 1) void wgrad_pc_kernel(WgradArgs a) {
    static_assert(!K96 || (TK == 2 && TN == 2), "96-channel consumer layout: 128 + 128 staged columns");
    constexpr int KB = 64 * TK, NB = 64 * TN, LD = KB + NB;
    // rows per stripe: the small tiles get longer stripes (fewer barriers) unless the pooled form wants stripes that
    // stay inside one 32-row-aligned group
    constexpr int RS = (TK * TN <= 2 && !is_pool(DMODE)) ? 64 : 32;
    constexpr int A4 = KB / 4, D4 = NB / 4;                   // float4 per stripe row
    constexpr int NA = RS * A4 / 256, ND = RS * D4 / 256;     // float4 per producer lane per stripe
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wave_prio_consumer(wave < 4);
    const int K = a.K, N = a.N;
    constexpr bool compact = DMODE == A_DYW || DMODE == A_DYPOOLB;      // compacted rows (block table + device row count)
    const long long M = compact ? (long long)__builtin_amdgcn_readfirstlane(*a.Mdev) : a.M;
    const int k0 = blockIdx.y * KB, n0 = blockIdx.z * NB;
    const int grp = blockIdx.x, ngrp = gridDim.x;
    float *coefA = lds;                        // [6][KB]  scale, shift | xyz-form w0 w1 w2 b
    float *coefD = coefA + 6 * KB;             // [5][NB]
    float *buf = coefD + 5 * NB;               // [2][RS][LD]   | afterwards: db scratch [256][4]

    for (int e = tid; e < KB; e += 512) {
        const int k = k0 + e;
        const bool in = k < K;
        coefA[e] = (AMODE != A_PLAIN && in) ? a.asc[k] : 0.f;
        coefA[KB + e] = (AMODE != A_PLAIN && in) ? a.ash[k] : 0.f;
        if (AMODE == A_XYZ) {
#pragma unroll
            for (int i = 0; i < 4; ++i) coefA[(2 + i) * KB + e] = in ? a.xw[i * a.xw_ld + k] : 0.f;
        }
    }
    for (int e = tid; e < NB; e += 512) {
        const int n = n0 + e;
        const bool in = n < N;
        coefD[e] = (in && a.p) ? a.p[n] : 0.f;
        coefD[NB + e] = (in && a.q) ? a.q[n] : 0.f;
        coefD[2 * NB + e] = (in && a.t) ? a.t[n] : 0.f;
        coefD[3 * NB + e] = (is_pool(DMODE) && in) ? a.dsc[n] : 0.f;
        coefD[4 * NB + e] = (is_pool(DMODE) && in) ? a.dsh[n] : 0.f;
    }
    __syncthreads();

    const long long nstripes = (M + RS - 1) / RS;
    const long long cnt = grp < nstripes ? (nstripes - grp + ngrp - 1) / ngrp : 0;   // stripes of this workgroup

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const int acq = (pt % A4) * 4, dcq = (pt % D4) * 4;
        const bool ain = k0 + acq < K, din = n0 + dcq < N;    // K % 4 == 0 and N % 4 == 0 (launcher)
        const int acl = ain ? k0 + acq : 0, dcl = din ? n0 + dcq : 0;
        const float4 casc = *reinterpret_cast<const float4 *>(&coefA[acq]);
        const float4 cash = *reinterpret_cast<const float4 *>(&coefA[KB + acq]);
        float4 xw0 = make_float4(0.f, 0.f, 0.f, 0.f), xw1 = xw0, xw2 = xw0, xb = xw0;
        if (AMODE == A_XYZ) {
            xw0 = *reinterpret_cast<const float4 *>(&coefA[2 * KB + acq]);
            xw1 = *reinterpret_cast<const float4 *>(&coefA[3 * KB + acq]);
            xw2 = *reinterpret_cast<const float4 *>(&coefA[4 * KB + acq]);
            xb = *reinterpret_cast<const float4 *>(&coefA[5 * KB + acq]);
        }
        const float4 cp = *reinterpret_cast<const float4 *>(&coefD[dcq]);
        const float4 cq = *reinterpret_cast<const float4 *>(&coefD[NB + dcq]);
        const float4 ct = *reinterpret_cast<const float4 *>(&coefD[2 * NB + dcq]);
        float dbs[4] = {0.f, 0.f, 0.f, 0.f};
        float4 px[NA], pg[ND], py[ND];
        unsigned pm[(is_pool(DMODE)) ? ND : 1];
        const unsigned xvoff = ain ? (unsigned)((pt / A4) * a.ldx + acl) * 4u : kOOB;
        const unsigned dvoff = din ? (unsigned)((pt / D4) * a.ldy + dcl) * 4u : kOOB;
        const unsigned xstep = (unsigned)(256 / A4) * (unsigned)a.ldx * 4u;
        const unsigned dstep = (unsigned)(256 / D4) * (unsigned)a.ldy * 4u;
        const long long glast = is_pool(DMODE) ? (M - 1) / a.S : 0;
        constexpr bool U_ = DMODE == A_DYPOOLU;                // one pooling group per stripe
        constexpr bool B_ = DMODE == A_DYPOOLB;                // compacted rows: one pooling group per 16-row block
        constexpr int NBLK = RS / kBlk;                        // blocks per stripe
        constexpr int QD = 256 / D4;                           // rows between a lane's consecutive D rows (divides 16)
        static_assert(QD <= kBlk && kBlk % QD == 0, "block of row pt / D4 + j QD is (j QD) / 16");
        float bw[compact ? NBLK : 1];                          // weight of the first row of each block of the stripe
        int bs0[B_ ? NBLK : 1];
#pragma unroll
        for (int h = 0; h < (compact ? NBLK : 1); ++h) bw[h] = 1.f;
#pragma unroll
        for (int h = 0; h < (B_ ? NBLK : 1); ++h) bs0[h] = 0;
        int us0 = 0;                                           // U_: row-in-group of the stripe's first row (set by issue)
        auto issue = [&](long long stripe) {
            const long long row0 = stripe * RS;
            if (compact) {
                const long long nblk = (M + kBlk - 1) / kBlk;
#pragma unroll
                for (int h = 0; h < NBLK; ++h) {
                    long long bi = stripe * NBLK + h;
                    bi = bi < nblk ? bi : nblk - 1;
                    const RowBlock rb = uniform_block(a.blocks, bi);
                    bw[h] = rb.w;
                    if (B_) {
                        bs0[h] = rb.s0;
                        pg[h] = *reinterpret_cast<const float4 *>(a.gpool + (long long)rb.g * N + dcl);
                        pm[h] = *reinterpret_cast<const unsigned *>(a.argmax + (long long)rb.g * N + dcl);
                    }
                }
            }
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + row0 * a.ldx, (M - row0) * a.ldx * 4);
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.Y + row0 * a.ldy, (M - row0) * a.ldy * 4);
            const __amdgpu_buffer_rsrc_t rg =
                make_rsrc((is_pool(DMODE) ? a.Y : a.G) + row0 * a.ldy, (M - row0) * a.ldy * 4);
            if (AMODE == A_XYZ) {     // 16 bytes per ROW, broadcast over the A4 lanes of a row
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.off4 + row0 * 4, (M - row0) * 16);
#pragma unroll
                for (int j = 0; j < NA; ++j) px[j] = buf_load4(ro, (unsigned)(pt / A4) * 16u, (unsigned)j * (256 / A4) * 16u);
            } else {
#pragma unroll
                for (int j = 0; j < NA; ++j) px[j] = buf_load4(rx, xvoff, (unsigned)j * xstep);
            }
            const PoolRows pr(is_pool(DMODE) ? row0 : 0, is_pool(DMODE) ? a.S : 1);
            us0 = pr.s0;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                py[j] = buf_load4(ry, dvoff, (unsigned)j * dstep);
                if (B_) {
                    // loaded per block above
                } else if (is_pool(DMODE)) {
                    if (U_) {
                        if (j == 0) {
                            const long long gi = pr.g0 < glast ? pr.g0 : glast;
                            pg[0] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                            pm[0] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
                        }
                    } else {
                        long long gi;
                        unsigned sdummy;
                        pr.split(pt / D4 + j * (256 / D4), glast, gi, sdummy);
                        pg[j] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                        pm[j] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
                    }
                } else if (DMODE != A_SELFD) {
                    pg[j] = buf_load4(rg, dvoff, (unsigned)j * dstep);
                }
            }
        };
        // FULL_: every row of the stripe exists and the tile is inside the layer -- wave-uniform, true for all stripes but
        // the last of an aligned layer: no range selects (the producers' instructions compete with the consumer wave of the
        // same SIMD for issue slots; the same diet took the one-pass backward kernel from 1 418 to 1 316 us)
        auto stage_ = [&](long long stripe, float *dst, auto full_) {
            constexpr bool FULL = decltype(full_)::value;
            const long long row0 = stripe * RS;
            constexpr bool G_ = DMODE == A_DYPOOL;             // any group size: per-row group arithmetic
            const PoolRows prs(G_ ? row0 : 0, G_ ? a.S : 1);
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int r = pt / A4 + j * (256 / A4);
                float4 x = px[j];
                if (AMODE == A_XYZ) {
                    const float4 o = x;
                    x = make_float4(xyz_y(o, xw0.x, xw1.x, xw2.x, xb.x), xyz_y(o, xw0.y, xw1.y, xw2.y, xb.y),
                                    xyz_y(o, xw0.z, xw1.z, xw2.z, xb.z), xyz_y(o, xw0.w, xw1.w, xw2.w, xb.w));
                }
                if (AMODE != A_PLAIN) {
                    x.x = fmaxf(fmaf(x.x, casc.x, cash.x), 0.f);
                    x.y = fmaxf(fmaf(x.y, casc.y, cash.y), 0.f);
                    x.z = fmaxf(fmaf(x.z, casc.z, cash.z), 0.f);
                    x.w = fmaxf(fmaf(x.w, casc.w, cash.w), 0.f);
                }
                if (!FULL && !(ain && row0 + r < M)) x = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(&dst[r * LD + acq]) = x;
            }
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const int r = pt / D4 + j * (256 / D4);
                const float4 y = py[j];
                const int hb = (j * QD) / kBlk;                // block of this row inside the stripe (compile time)
                float4 g = DMODE == A_SELFD ? make_float4(0.f, 0.f, 0.f, 0.f) : pg[U_ ? 0 : (B_ ? hb : j)];
                if (is_pool(DMODE)) {
                    long long gdummy;
                    unsigned s;
                    if (U_) s = (unsigned)(us0 + r);
                    else if (B_) s = (unsigned)(bs0[B_ ? hb : 0] + (r & (kBlk - 1)));
                    else prs.split(r, glast, gdummy, s);
                    const unsigned am = pm[U_ ? 0 : (B_ ? hb : j)];
                    // gpool arrives MASKED (pcops.h: upstream gradient x [relu(bn(ysel)) > 0]): only the row test is left
                    g.x = ((am & 0xffu) == s) ? g.x : 0.f;
                    g.y = (((am >> 8) & 0xffu) == s) ? g.y : 0.f;
                    g.z = (((am >> 16) & 0xffu) == s) ? g.z : 0.f;
                    g.w = ((am >> 24) == s) ? g.w : 0.f;
                }
                float4 d;
                if (DMODE == A_SELFD && AMODE == A_PLAIN) {
                    d = y;
                } else if (DMODE == A_SELFD) {
                    d.x = fmaxf(fmaf(y.x, cq.x, ct.x), 0.f);
                    d.y = fmaxf(fmaf(y.y, cq.y, ct.y), 0.f);
                    d.z = fmaxf(fmaf(y.z, cq.z, ct.z), 0.f);
                    d.w = fmaxf(fmaf(y.w, cq.w, ct.w), 0.f);
                } else if (compact && (j * QD) % kBlk == 0) {
                    // a row that opens a block (r % 16 == 0): dY = p.G + w (q.Y + t)
                    const float w = (r & (kBlk - 1)) == 0 ? bw[compact ? hb : 0] : 1.f;
                    d.x = fmaf(cp.x, g.x, w * fmaf(cq.x, y.x, ct.x));
                    d.y = fmaf(cp.y, g.y, w * fmaf(cq.y, y.y, ct.y));
                    d.z = fmaf(cp.z, g.z, w * fmaf(cq.z, y.z, ct.z));
                    d.w = fmaf(cp.w, g.w, w * fmaf(cq.w, y.w, ct.w));
                } else {
                    d.x = fmaf(cp.x, g.x, fmaf(cq.x, y.x, ct.x));
                    d.y = fmaf(cp.y, g.y, fmaf(cq.y, y.y, ct.y));
                    d.z = fmaf(cp.z, g.z, fmaf(cq.z, y.z, ct.z));
                    d.w = fmaf(cp.w, g.w, fmaf(cq.w, y.w, ct.w));
                }
                if (!FULL && !(din && row0 + r < M)) d = make_float4(0.f, 0.f, 0.f, 0.f);
                dbs[0] += d.x; dbs[1] += d.y; dbs[2] += d.z; dbs[3] += d.w;
                *reinterpret_cast<float4 *>(&dst[r * LD + KB + dcq]) = d;
            }
        };
        auto stage = [&](long long stripe, float *dst) {
            if (stripe * RS + RS <= M && k0 + KB <= K && n0 + NB <= N) stage_(stripe, dst, std::true_type{});
            else stage_(stripe, dst, std::false_type{});
        };
        if (cnt > 0) {
            issue(grp);
            stage(grp, buf);
            if (cnt > 1) issue(grp + ngrp);
        }
        __syncthreads();                                       // stripe 0 is in buf[0]
        for (long long i = 0; i < cnt; ++i) {
            if (i + 1 < cnt) {
                stage(grp + (i + 1) * ngrp, buf + ((i + 1) & 1) * RS * LD);
                if (i + 2 < cnt) issue(grp + (i + 2) * ngrp);
            }
            __syncthreads();
        }
        // column sums of dY: lanes pt, pt + D4, ... own the same 4 columns
        float *sdb = buf;
#pragma unroll
        for (int e = 0; e < 4; ++e) sdb[pt * 4 + e] = dbs[e];
        __syncthreads();
    } else {
        // ------------------------------------------------------------------ consumers
        constexpr int CN = K96 ? 4 : 2;                        // consumers across the dY columns
        constexpr int KX = K96 ? 3 : TK, NY = K96 ? 1 : TN;    // 32 x 32 blocks of a consumer: KX of A against NY of dY
        const int ck = wave / CN, cn = wave % CN;
        const int half = lane >> 5, li = lane & 31;
        f32x16 acc[KX][NY];
#pragma unroll
        for (int i = 0; i < KX; ++i)
#pragma unroll
            for (int j = 0; j < NY; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
        const int aoff = half * LD + ck * KX * 32 + li;
        const int doff = half * LD + KB + cn * NY * 32 + li;
        __syncthreads();
        for (long long i = 0; i < cnt; ++i) {
            const float *sb = buf + (i & 1) * RS * LD;
            float av_n[KX], dv_n[NY];
#pragma unroll
            for (int x = 0; x < KX; ++x) av_n[x] = sb[aoff + 32 * x];
#pragma unroll
            for (int y = 0; y < NY; ++y) dv_n[y] = sb[doff + 32 * y];
#pragma unroll
            for (int it = 0; it < RS / 2; ++it) {
                float av[KX], dv[NY];
#pragma unroll
                for (int x = 0; x < KX; ++x) av[x] = av_n[x];
#pragma unroll
                for (int y = 0; y < NY; ++y) dv[y] = dv_n[y];
                if (it + 1 < RS / 2) {
#pragma unroll
                    for (int x = 0; x < KX; ++x) av_n[x] = sb[aoff + 2 * (it + 1) * LD + 32 * x];
#pragma unroll
                    for (int y = 0; y < NY; ++y) dv_n[y] = sb[doff + 2 * (it + 1) * LD + 32 * y];
                }
                // the next row pair's fragments are REQUESTED before this pair's MFMAs issue (the scheduler would
                // otherwise sink the reads to just in front of their first use and expose the LDS latency)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int x = 0; x < KX; ++x)
#pragma unroll
                    for (int y = 0; y < NY; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], dv[y], acc[x][y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        // acc[x][y][v]: A channel 32 x + (v&3) + 8 (v>>2) + 4 half, dY channel 32 y + li  (of this consumer's block)
        float *out = a.part + (long long)grp * K * N;
#pragma unroll
        for (int x = 0; x < KX; ++x)
#pragma unroll
            for (int y = 0; y < NY; ++y) {
                const int nn = n0 + (cn * NY + y) * 32 + li;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int kk = k0 + (ck * KX + x) * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
                    if (kk < K && nn < N) out[(long long)kk * N + nn] = acc[x][y][v];
                }
            }
        __syncthreads();                                       // matches the producers' db hand-over
        if (blockIdx.y == 0 && a.dbpart) {
            const float *sdb = buf;
            for (int c = tid; c < NB; c += 256) {
                const int quad = c >> 2, e = c & 3;
                float sum = 0.f;
                for (int r = quad; r < 256; r += D4) sum += sdb[r * 4 + e];
                if (n0 + c < N) a.dbpart[(long long)grp * N + n0 + c] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad with SPLIT OPERANDS on the bf16 matrix pipe (round 4): dW = A^T dY reduces over ROWS, so a
// matrix instruction wants, per lane, eight consecutive rows of ONE channel.  The producers therefore hand the stripe
// over TRANSPOSED and already split: T[piece][column slot][row] in bf16, three pieces per operand; a lane of a consumer
// reads its fragment (8 rows x 1 column) with one ds_read_b128 per piece.
//   waves 4..7  PRODUCERS: lane (cq = column quad, rg = group of FOUR CONSECUTIVE rows): 16-byte loads of rows 4 rg .. 4 rg + 3
//               (a row of 128 channels is read by 32 lanes: coalesced), transform as in wgrad_pc_kernel, split, and per
//               column and piece ONE 8-byte store of the four rows.  Column c of an operand lives in slot
//               32 (c % 4) + c / 4: the 32 lanes of a row then write 32 consecutive slots (80 bytes apart: two passes
//               for 64 lanes x 8 bytes, the minimum), and a consumer block of 32 slots holds the channels 4 m + e --
//               a permutation that only shows in the index arithmetic of the final store.
//   waves 0..3  CONSUMERS: 2 x 2 blocks of 32 x 32 each; per 16 rows and block pair six v_mfma_f32_32x32x16_bf16 -- the
//               h.h product into the block's accumulator, the five small ones into a second set that is added once at
//               the end (the large accumulator is then rounded once per 16 rows instead of once per 2: a long reduction
//               -- 8 192 rows per workgroup at SA2's size -- is where that matters most).
// Tile 128 x 128 (K or N beyond: more workgroups along y / z; the other operand is then read once per block).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3x4(float v0, float v1, float v2, float v3, bf16x4 &h, bf16x4 &m, bf16x4 &l) {
    const float x[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        bf16x2 hh, mm, ll;
        split3_pair(x[2 * p], x[2 * p + 1], hh, mm, ll);
        h[2 * p] = hh.x; h[2 * p + 1] = hh.y;
        m[2 * p] = mm.x; m[2 * p + 1] = mm.y;
        l[2 * p] = ll.x; l[2 * p + 1] = ll.y;
    }
}

// K96 (round 6; 65 .. 96 input channels, MSG's 96 -> 128): the X columns take slots 24 e + cq = 0 .. 95 (the eight lanes per row without
// columns put nothing), and the four consumer waves split the OUTPUT as 96 x 32 each -- three A blocks against ONE dY block, 18
// matrix instructions per 16-row step instead of 24 (a quarter of the 128-slot layout's products were against 32 empty channels,
// which its slot order 32 e + cq spread over all four blocks).
template <int AMODE, int DMODE, bool K96 = false>
__global__ __launch_bounds__(512, 1) void wgrad_bf3_kernel(WgradArgs a) {
    constexpr int KB = 128, NB = 128, RS = 32;
    constexpr int RSP = 40;                    // bf16 per slot: 32 rows + 8 (80 bytes: 16-byte aligned, 5 x 16 -> b128 reads of
                                               // 16 consecutive slots fall into 16 different 16-byte bank groups)
    constexpr int SLOTS = KB + NB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, N = a.N;
    constexpr bool compact = DMODE == A_DYW || DMODE == A_DYPOOLB;
    const long long M = compact ? (long long)__builtin_amdgcn_readfirstlane(*a.Mdev) : a.M;
    const int k0 = blockIdx.y * KB, n0 = blockIdx.z * NB;
    const int grp = blockIdx.x, ngrp = gridDim.x;
    float *coefA = lds;                        // [6][KB]
    float *coefD = coefA + 6 * KB;             // [5][NB]
    __bf16 *T = reinterpret_cast<__bf16 *>(coefD + 5 * NB);    // [2][3][SLOTS][RSP]   | afterwards: db scratch [256][4]

    for (int e = tid; e < KB; e += 512) {
        const int k = k0 + e;
        const bool in = k < K;
        coefA[e] = (AMODE != A_PLAIN && in) ? a.asc[k] : 0.f;
        coefA[KB + e] = (AMODE != A_PLAIN && in) ? a.ash[k] : 0.f;
        if (AMODE == A_XYZ) {
#pragma unroll
            for (int i = 0; i < 4; ++i) coefA[(2 + i) * KB + e] = in ? a.xw[i * a.xw_ld + k] : 0.f;
        }
    }
    for (int e = tid; e < NB; e += 512) {
        const int n = n0 + e;
        const bool in = n < N;
        coefD[e] = (in && a.p) ? a.p[n] : 0.f;
        coefD[NB + e] = (in && a.q) ? a.q[n] : 0.f;
        coefD[2 * NB + e] = (in && a.t) ? a.t[n] : 0.f;
        coefD[3 * NB + e] = (is_pool(DMODE) && in) ? a.dsc[n] : 0.f;
        coefD[4 * NB + e] = (is_pool(DMODE) && in) ? a.dsh[n] : 0.f;
    }
    __syncthreads();

    const long long nstripes = (M + RS - 1) / RS;
    const long long cnt = grp < nstripes ? (nstripes - grp + ngrp - 1) / ngrp : 0;   // stripes of this workgroup

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const int cq = pt & 31, rg = pt >> 5;                 // column quad; rows 4 rg .. 4 rg + 3 of the stripe
        const int acq = cq * 4, dcq = cq * 4;
        const bool ain = k0 + acq < K, din = n0 + dcq < N;    // K % 4 == 0 and N % 4 == 0 (launcher)
        const int acl = ain ? k0 + acq : 0, dcl = din ? n0 + dcq : 0;
        const float4 casc = *reinterpret_cast<const float4 *>(&coefA[acq]);
        const float4 cash = *reinterpret_cast<const float4 *>(&coefA[KB + acq]);
        float4 xw0 = make_float4(0.f, 0.f, 0.f, 0.f), xw1 = xw0, xw2 = xw0, xb = xw0;
        if (AMODE == A_XYZ) {
            xw0 = *reinterpret_cast<const float4 *>(&coefA[2 * KB + acq]);
            xw1 = *reinterpret_cast<const float4 *>(&coefA[3 * KB + acq]);
            xw2 = *reinterpret_cast<const float4 *>(&coefA[4 * KB + acq]);
            xb = *reinterpret_cast<const float4 *>(&coefA[5 * KB + acq]);
        }
        const float4 cp = *reinterpret_cast<const float4 *>(&coefD[dcq]);
        const float4 cqv = *reinterpret_cast<const float4 *>(&coefD[NB + dcq]);
        const float4 ct = *reinterpret_cast<const float4 *>(&coefD[2 * NB + dcq]);
        float dbs[4] = {0.f, 0.f, 0.f, 0.f};
        float4 px[4], pg[4], py[4];
        unsigned pm[(is_pool(DMODE)) ? 4 : 1];
        const unsigned xvoff = ain ? (unsigned)((4 * rg) * a.ldx + acl) * 4u : kOOB;
        const unsigned dvoff = din ? (unsigned)((4 * rg) * a.ldy + dcl) * 4u : kOOB;
        const unsigned xstep = (unsigned)a.ldx * 4u, dstep = (unsigned)a.ldy * 4u;
        const long long glast = is_pool(DMODE) ? (M - 1) / a.S : 0;
        constexpr bool U_ = DMODE == A_DYPOOLU;               // one pooling group per stripe
        constexpr bool B_ = DMODE == A_DYPOOLB;               // compacted rows: one pooling group per 16-row block
        constexpr bool G_ = DMODE == A_DYPOOL;                // any group size: per-row group arithmetic
        const int hb = rg >> 2;                               // the 16-row block this lane's four rows lie in
        float bw = 1.f;                                       // weight of that block's first row
        int bs0 = 0, us0 = 0;
        auto issue = [&](long long stripe) {
            const long long row0 = stripe * RS;
            if (compact) {
                const long long nblk = (M + kBlk - 1) / kBlk;
                // (the lane's block: both records of the stripe are scalar loads, the lane keeps its own)
                static_assert(RS / kBlk == 2, "two 16-row blocks per stripe");
                long long b0 = stripe * (RS / kBlk), b1 = b0 + 1;
                b0 = b0 < nblk ? b0 : nblk - 1;
                b1 = b1 < nblk ? b1 : nblk - 1;
                const RowBlock r0 = uniform_block(a.blocks, b0), r1 = uniform_block(a.blocks, b1);
                bw = hb ? r1.w : r0.w;
                if (B_) {
                    const long long gsel = hb ? r1.g : r0.g;
                    bs0 = hb ? r1.s0 : r0.s0;
                    pg[0] = *reinterpret_cast<const float4 *>(a.gpool + gsel * N + dcl);
                    pm[0] = *reinterpret_cast<const unsigned *>(a.argmax + gsel * N + dcl);
                }
            }
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + row0 * a.ldx, (M - row0) * a.ldx * 4);
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.Y + row0 * a.ldy, (M - row0) * a.ldy * 4);
            const __amdgpu_buffer_rsrc_t rgs =
                make_rsrc((is_pool(DMODE) ? a.Y : a.G) + row0 * a.ldy, (M - row0) * a.ldy * 4);
            if (AMODE == A_XYZ) {     // 16 bytes per ROW, broadcast over the 32 lanes of a row
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.off4 + row0 * 4, (M - row0) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) px[j] = buf_load4(ro, (unsigned)(4 * rg) * 16u, (unsigned)j * 16u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) px[j] = buf_load4(rx, xvoff, (unsigned)j * xstep);
            }
            const PoolRows pr(is_pool(DMODE) ? row0 : 0, is_pool(DMODE) ? a.S : 1);
            us0 = pr.s0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                py[j] = buf_load4(ry, dvoff, (unsigned)j * dstep);
                if (B_) {
                    // loaded per block above
                } else if (is_pool(DMODE)) {
                    if (U_) {
                        if (j == 0) {
                            const long long gi = pr.g0 < glast ? pr.g0 : glast;
                            pg[0] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                            pm[0] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
                        }
                    } else {
                        long long gi;
                        unsigned sdummy;
                        pr.split(4 * rg + j, glast, gi, sdummy);
                        pg[j] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                        pm[j] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
                    }
                } else if (DMODE != A_SELFD) {
                    pg[j] = buf_load4(rgs, dvoff, (unsigned)j * dstep);
                }
            }
        };
        // four rows x four columns of one operand -> three pieces, one 8-byte store per column and piece
        auto put = [&](const float4 (&v)[4], __bf16 *tbuf, int slot0, int q = 32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16x4 h, m, l;
                split3x4(e == 0 ? v[0].x : (e == 1 ? v[0].y : (e == 2 ? v[0].z : v[0].w)),
                         e == 0 ? v[1].x : (e == 1 ? v[1].y : (e == 2 ? v[1].z : v[1].w)),
                         e == 0 ? v[2].x : (e == 1 ? v[2].y : (e == 2 ? v[2].z : v[2].w)),
                         e == 0 ? v[3].x : (e == 1 ? v[3].y : (e == 2 ? v[3].z : v[3].w)), h, m, l);
                __bf16 *dst = tbuf + (slot0 + q * e + cq) * RSP + 4 * rg;
                *reinterpret_cast<bf16x4 *>(dst) = h;
                *reinterpret_cast<bf16x4 *>(dst + SLOTS * RSP) = m;
                *reinterpret_cast<bf16x4 *>(dst + 2 * SLOTS * RSP) = l;
            }
        };
        auto stage_ = [&](long long stripe, __bf16 *tbuf, auto full_) {
            constexpr bool FULL = decltype(full_)::value;
            const long long row0 = stripe * RS;
            const PoolRows prs(G_ ? row0 : 0, G_ ? a.S : 1);
            float4 ax[4], dd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * rg + j;
                float4 x = px[j];
                if (AMODE == A_XYZ) {
                    const float4 o = x;
                    x = make_float4(xyz_y(o, xw0.x, xw1.x, xw2.x, xb.x), xyz_y(o, xw0.y, xw1.y, xw2.y, xb.y),
                                    xyz_y(o, xw0.z, xw1.z, xw2.z, xb.z), xyz_y(o, xw0.w, xw1.w, xw2.w, xb.w));
                }
                if (AMODE != A_PLAIN) {
                    x.x = fmaxf(fmaf(x.x, casc.x, cash.x), 0.f);
                    x.y = fmaxf(fmaf(x.y, casc.y, cash.y), 0.f);
                    x.z = fmaxf(fmaf(x.z, casc.z, cash.z), 0.f);
                    x.w = fmaxf(fmaf(x.w, casc.w, cash.w), 0.f);
                }
                if (!FULL && !(ain && row0 + r < M)) x = make_float4(0.f, 0.f, 0.f, 0.f);
                ax[j] = x;
            }
            if (!K96) put(ax, tbuf, 0);
            else if (cq < 24) put(ax, tbuf, 0, 24);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * rg + j;
                const float4 y = py[j];
                float4 g = DMODE == A_SELFD ? make_float4(0.f, 0.f, 0.f, 0.f) : pg[(U_ || B_) ? 0 : j];
                if (is_pool(DMODE)) {
                    long long gdummy;
                    unsigned s;
                    if (U_) s = (unsigned)(us0 + r);
                    else if (B_) s = (unsigned)(bs0 + (r & (kBlk - 1)));
                    else prs.split(r, glast, gdummy, s);
                    const unsigned am = pm[(U_ || B_) ? 0 : j];
                    // gpool arrives MASKED (pcops.h: upstream gradient x [relu(bn(ysel)) > 0]): only the row test is left
                    g.x = ((am & 0xffu) == s) ? g.x : 0.f;
                    g.y = (((am >> 8) & 0xffu) == s) ? g.y : 0.f;
                    g.z = (((am >> 16) & 0xffu) == s) ? g.z : 0.f;
                    g.w = ((am >> 24) == s) ? g.w : 0.f;
                }
                float4 d;
                if (DMODE == A_SELFD && AMODE == A_PLAIN) {
                    d = y;
                } else if (DMODE == A_SELFD) {
                    d.x = fmaxf(fmaf(y.x, cqv.x, ct.x), 0.f);
                    d.y = fmaxf(fmaf(y.y, cqv.y, ct.y), 0.f);
                    d.z = fmaxf(fmaf(y.z, cqv.z, ct.z), 0.f);
                    d.w = fmaxf(fmaf(y.w, cqv.w, ct.w), 0.f);
                } else if (compact && j == 0) {
                    // a row that opens a block (r % 16 == 0): dY = p.G + w (q.Y + t)
                    const float w = (rg & 3) == 0 ? bw : 1.f;
                    d.x = fmaf(cp.x, g.x, w * fmaf(cqv.x, y.x, ct.x));
                    d.y = fmaf(cp.y, g.y, w * fmaf(cqv.y, y.y, ct.y));
                    d.z = fmaf(cp.z, g.z, w * fmaf(cqv.z, y.z, ct.z));
                    d.w = fmaf(cp.w, g.w, w * fmaf(cqv.w, y.w, ct.w));
                } else {
                    d.x = fmaf(cp.x, g.x, fmaf(cqv.x, y.x, ct.x));
                    d.y = fmaf(cp.y, g.y, fmaf(cqv.y, y.y, ct.y));
                    d.z = fmaf(cp.z, g.z, fmaf(cqv.z, y.z, ct.z));
                    d.w = fmaf(cp.w, g.w, fmaf(cqv.w, y.w, ct.w));
                }
                if (!FULL && !(din && row0 + r < M)) d = make_float4(0.f, 0.f, 0.f, 0.f);
                dbs[0] += d.x; dbs[1] += d.y; dbs[2] += d.z; dbs[3] += d.w;
                dd[j] = d;
            }
            put(dd, tbuf, KB);
        };
        auto stage = [&](long long stripe, __bf16 *tbuf) {
            if (stripe * RS + RS <= M && k0 + KB <= K && n0 + NB <= N) stage_(stripe, tbuf, std::true_type{});
            else stage_(stripe, tbuf, std::false_type{});
        };
        if (cnt > 0) {
            issue(grp);
            stage(grp, T);
            if (cnt > 1) issue(grp + ngrp);
        }
        __syncthreads();                                       // stripe 0 is in buffer 0
        for (long long i = 0; i < cnt; ++i) {
            if (i + 1 < cnt) {
                stage(grp + (i + 1) * ngrp, T + ((i + 1) & 1) * 3 * SLOTS * RSP);
                if (i + 2 < cnt) issue(grp + (i + 2) * ngrp);
            }
            __syncthreads();
        }
        // column sums of dY: lanes pt, pt + 32, ... own the same 4 columns
        float *sdb = reinterpret_cast<float *>(T);
#pragma unroll
        for (int e = 0; e < 4; ++e) sdb[pt * 4 + e] = dbs[e];
        __syncthreads();
    } else {
        // ------------------------------------------------------------------ consumers
        const int ck = wave >> 1, cn = wave & 1;
        const int half = lane >> 5, li = lane & 31;
        constexpr int NXB = K96 ? 3 : 2, NYB = K96 ? 1 : 2;    // A blocks x dY blocks of a consumer wave
        f32x16 acc[NXB][NYB], sm[NXB][NYB];
#pragma unroll
        for (int i = 0; i < NXB; ++i)
#pragma unroll
            for (int j = 0; j < NYB; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = sm[i][j][v] = 0.f;
        const int aslot = K96 ? li : (2 * ck) * 32 + li;                            // + 32 x
        const int dslot = K96 ? KB + wave * 32 + li : KB + (2 * cn) * 32 + li;      // + 32 y
        __syncthreads();
        for (long long i = 0; i < cnt; ++i) {
            const __bf16 *tb = T + (i & 1) * 3 * SLOTS * RSP;
#pragma unroll
            for (int s = 0; s < RS / 16; ++s) {
                const int ro = 16 * s + 8 * half;
                bf16x8 ah[NXB], am[NXB], al[NXB], dh[NYB], dm[NYB], dl[NYB];
#pragma unroll
                for (int x = 0; x < NXB; ++x) {
                    const __bf16 *p0 = tb + (aslot + 32 * x) * RSP + ro;
                    ah[x] = *reinterpret_cast<const bf16x8 *>(p0);
                    am[x] = *reinterpret_cast<const bf16x8 *>(p0 + SLOTS * RSP);
                    al[x] = *reinterpret_cast<const bf16x8 *>(p0 + 2 * SLOTS * RSP);
                }
#pragma unroll
                for (int y = 0; y < NYB; ++y) {
                    const __bf16 *p0 = tb + (dslot + 32 * y) * RSP + ro;
                    dh[y] = *reinterpret_cast<const bf16x8 *>(p0);
                    dm[y] = *reinterpret_cast<const bf16x8 *>(p0 + SLOTS * RSP);
                    dl[y] = *reinterpret_cast<const bf16x8 *>(p0 + 2 * SLOTS * RSP);
                }
                // one product at a time over the four blocks: consecutive matrix instructions never share an accumulator
#define PCOPS_MMX(A_, D_, C_)                                                                              \
    _Pragma("unroll") for (int x = 0; x < NXB; ++x) _Pragma("unroll") for (int y = 0; y < NYB; ++y)        \
        C_[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x], D_[y], C_[x][y], 0, 0, 0)
                PCOPS_MMX(al, dh, sm);
                PCOPS_MMX(ah, dl, sm);
                PCOPS_MMX(am, dm, sm);
                PCOPS_MMX(am, dh, sm);
                PCOPS_MMX(ah, dm, sm);
                PCOPS_MMX(ah, dh, acc);
#undef PCOPS_MMX
            }
            __syncthreads();
        }
        // acc[x][y][v]: A slot 32 (2 ck + x) + m with m = (v&3) + 8 (v>>2) + 4 half  ->  channel 4 m + (2 ck + x);
        //               dY slot 32 (2 cn + y) + li                                     ->  column  4 li + (2 cn + y)
        float *out = a.part + (long long)grp * K * N;
        if constexpr (K96) {
            // A slot 32 x + m = 24 e + cq  ->  channel 4 cq + e;  dY slot 32 wave + li  ->  column 4 li + wave
            const int nn = n0 + 4 * li + wave;
#pragma unroll
            for (int x = 0; x < NXB; ++x)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int sl = 32 * x + (v & 3) + 8 * (v >> 2) + 4 * half;
                    const int kk = 4 * (sl % 24) + sl / 24;                   // (k0 = 0: one block of input channels)
                    if (kk < K && nn < N) out[(long long)kk * N + nn] = acc[x][0][v] + sm[x][0][v];
                }
        } else {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int nn = n0 + 4 * li + 2 * cn + y;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int kk = k0 + 4 * ((v & 3) + 8 * (v >> 2) + 4 * half) + 2 * ck + x;
                    if (kk < K && nn < N) out[(long long)kk * N + nn] = acc[x][y][v] + sm[x][y][v];
                }
            }
        }
        __syncthreads();                                       // matches the producers' db hand-over
        if (blockIdx.y == 0 && a.dbpart) {
            const float *sdb = reinterpret_cast<const float *>(T);
            for (int c = tid; c < NB; c += 256) {
                const int quad = c >> 2, e = c & 3;
                float sum = 0.f;
                for (int r = quad; r < 256; r += 32) sum += sdb[r * 4 + e];
                if (n0 + c < N) a.dbpart[(long long)grp * N + n0 + c] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Data AND weight gradient of a 64-wide layer in ONE pass over its tensors (round 3).  The two kernels of such a layer
// are bandwidth bound and read the same bytes: SA1's 64 -> 128 layer moves 4.3 GB for the data gradient (Y, the mask
// tensor, Gprev) and 3.2 GB for the weight gradient (Y again, the mask tensor again as the A operand).  Here the
// producer / consumer weight-gradient kernel keeps what it has already staged -- relu(bn(Yprev)) and
// dY = p.G + q.Y + t of a 32-row stripe -- and its consumer waves ALSO multiply the dY stripe with the resident weights:
//   waves 4..7  PRODUCERS: HBM -> registers a stripe ahead -> LDS stripe  [32][ X (64) | raw Yprev (64) | dY (NB) ]
//   waves 0..3  CONSUMERS: dW += X^T dY  (v_mfma_f32_32x32x2_f32, 32 x (NB/2) block per wave, as in wgrad_pc_kernel)
//                          dX  = dY W^T  (v_mfma_f32_16x16x4_f32: two 16 x 16 blocks per wave and stripe; W sits in LDS
//                                         n-quad major, which is how its rows lie in memory -- no transposition), then
//                          Gprev = dX . [bn(Yprev) > 0]  straight from the accumulators, with the column sums
//                          (sum Gprev, sum Gprev Yprev) the BN backward of the layer below needs; the mask and the
//                          statistics read the RAW Yprev the producers left beside X -- no second trip to memory.
// Per stripe a consumer issues 16 NB/64 MFMAs of 64 cycles and 4 NB/16 of 32: equal matrix-pipe time for the two products
// (4 096 cycles at NB = 128); the pass is pipe bound at about 0.7 x the time of the two kernels it replaces.
// NSK (N <= NB - 32, MSG's 64 -> 96 layer on the 128-column tile): the consumers skip what only multiplies the zero
// columns -- the dW block of dY columns >= N and the data-gradient steps over them (wave-uniform tests on N).
// DX3: the data-gradient half (dX = dY W^T, a reduction over the dY columns) on the bf16 matrix pipe with split operands
// (DESIGN.md section 4.10): its A operand is the row-major dY the stripe already holds -- a lane reads eight consecutive
// columns of its row and splits them in registers -- and the wave's W slice is split once into register-resident pieces;
// per 32 dY columns six v_mfma_f32_16x16x32_bf16 per 16 x 16 block instead of eight v_mfma_f32_16x16x4_f32 of twice the
// length (768 instead of 2 048 matrix cycles per stripe and wave, ~700 cycles of split arithmetic in exchange), the h.h
// products in the block's accumulator, the small ones in a second one.  The dW half stays on the fp32 pipe (its operands
// would have to be staged transposed).
// SIDE (round 5): the layer below is the first EdgeConv layer of a stack whose input needs no gradient (pcops.h
// pcops_edge_first_*): its masked gradient is not written either -- its weight gradient is linear in E^T Gprev, E the six edge
// channels of a row, which ride along as 32 bytes per row (a.side) and are reduced exactly like the xyz form's offsets.
// GW (round 6): the weight gradient of a POOLED layer in its Gram form.  dY = p.G + q.Y + t has ONE non-zero row of G per
// (group, channel) and Y = X W + b, so  dW = X^T (p.G) + (X^T X) W diag(q) + (X^T 1)(q.b + t)^T:  the consumers multiply a
// 64 x 64 Gram matrix (16 matrix instructions per stripe and wave instead of 16 NB / 32 -- half the dW matrix time at
// NB = 128) and add the arg rows as vector work: lane (channel c, k slice) reads its group's (gpool, arg row) pair and adds
// p g X[arg row][slice] -- NB / 4 fused multiply-adds per group and lane.  The K x N product with W diag(q) happens once,
// on the summed partials (bwd_fused_gw_finish_kernel).  Kernel time is issue time on this chip (matrix and vector
// instructions of the two waves of a SIMD add up), so half the dW matrix cycles is ~16 % of the pass.
// DW3 (round 6): the dW half on the bf16 matrix pipe with split operands as well.  dW = X^T dY reduces over ROWS, so the producers
// hand X and dY over a second time, TRANSPOSED and pre-split (T[piece][column slot][row], wgrad_bf3_kernel's layout: a lane owns
// two / four CONSECUTIVE rows of a column quad and stores each column's rows as one 4- / 8-byte word per piece); a consumer
// fragment is one ds_read_b128 per piece, 24 v_mfma_f32_32x32x16_bf16 per stripe and wave (768 cycles) where 32
// v_mfma_f32_32x32x2_f32 (2 048) ran, h.h products in the block's accumulator, the five small ones in a second set.  X leaves the
// fp32 stripe (only dW read it): [raw | dY | pad] + the pieces = 144 KB; the W staging area lies under the second piece buffer.
template <int TN, int DMODE, bool XYZ, bool NSK = false, bool DX3 = false, bool SIDE = false, bool GW = false, bool DW3 = false>
__global__ __launch_bounds__(512, 1) void bwd_fused_kernel(WgradArgs a) {
    static_assert(!GW || (is_pool(DMODE) && DMODE != A_DYPOOLB && !XYZ), "Gram form: pooled, uncompacted rows");
    static_assert(!DW3 || (DX3 && !GW), "split-operand dW: with the split-operand dX half, not with the Gram form");
    // XYZ: the layer below is the arithmetic first layer (A_XYZ above): its raw rows are rebuilt from 16 bytes of offsets,
    // its masked gradient is never written -- only the sums its own gradients are linear in leave (gstats, xstats)
    constexpr int KB = 64, NB = 64 * TN, RS = 32;
    static_assert(!(XYZ && SIDE), "one reduced first layer below");
    constexpr int NX = XYZ ? 3 : (SIDE ? 6 : 0);              // per-row inputs the masked gradient is reduced against
    // fp32 stripe row:  X | raw | dY | pad  (row stride = 4 banks mod 32; SIDE: 12 mod 32);  DW3: raw | dY | pad
    // RM (DW3 of the EdgeConv form, -DPCOPS_BF_RM=0 for the layout before it): dY ALSO leaves the fp32 stripe -- the producers hand its three pieces over a
    // second time ROW-major, R[piece][row][column], and a consumer's dX operand is three 16-byte reads instead of two fp32 reads and a
    // split3 (36 vector instructions per step and wave that four waves repeated on values the producers had split already).  The LDS
    // for it comes out of the transposed pieces' padding: 32 rows per slot instead of 40, with the 16-byte unit of a slot XOR-ed with
    // (slot / 4) % 4 so that the 16 lanes of a read phase still meet 16 different bank groups.
    // Measured same-box (profiles/r06_bwd_fused_rm_ab.txt): the EdgeConv form 3 843 -> 3 790 us (DGCNN +0.3 .. 0.9 % over three
    // alternations); the plain forms 1 373 -> 1 411 us (SA1) and 3 750 -> 3 862 us (T-Net) -- their producers spill ten registers
    // with the second hand-over where the EdgeConv producers (which hold less per row) do not pay for it -- so they keep the fp32 dY.
    constexpr bool RM = DW3 && SIDE && (PCOPS_BF_RM != 0);
    constexpr int OXC = 0, ORC = DW3 ? 0 : KB, ODC = DW3 ? KB : 2 * KB, OPC = RM ? KB : ODC + NB;
    constexpr int LD = OPC + (SIDE ? 12 : 4);
    constexpr int RLD = NB + 8;                               // RM: bf16 per row of the row-major pieces (272 bytes: a row shifts one bank group)
    constexpr int RBUF = 3 * RS * RLD;                        // RM: bf16 per buffer
    constexpr int A4 = KB / 4, D4 = NB / 4;
    constexpr int NA = RS * A4 / 256, ND = RS * D4 / 256;
    constexpr int RSP = RM ? 32 : 40, SLOTS = KB + NB;        // DW3: bf16 per slot (32 rows + 8: 80-byte slots; RM: 32, swizzled), slots per piece
    constexpr int TBUF = 3 * SLOTS * RSP;                     // bf16 per piece buffer
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wave_prio_consumer(wave < 4);
    const int K = a.K, N = a.N;
    constexpr bool compact = DMODE == A_DYW || DMODE == A_DYPOOLB;      // compacted rows (block table + device row count)
    const long long M = compact ? (long long)__builtin_amdgcn_readfirstlane(*a.Mdev) : a.M;
    const int grp = blockIdx.x, ngrp = gridDim.x;
#ifdef PCOPS_BF_DEBUG
    const int dbg = a.rows_per_block;          // ablation experiments (tools/ only)
#else
    constexpr int dbg = 0;
#endif
    constexpr int CA = XYZ ? 6 : 2;
    float *coefA = lds;                        // [CA][KB]  scale, shift of the layer below | xyz form: w0 w1 w2 b
    float *coefD = coefA + CA * KB;            // [3][NB]  p, q, t
    // DW3: [2][RS][LD] stripes, then the piece buffers T[2][3][SLOTS][RSP]; the W staging area lies under T[1] (dead once the
    // consumers hold their slices in registers, and T[1] is first written after the barrier that follows)
    float *buf = DW3 ? coefD + 3 * NB : coefD + 3 * NB + NB * KB;   // [2][RS][LD]   | afterwards: db scratch [256][4], statistics [2][5][KB]
    __bf16 *Tp = reinterpret_cast<__bf16 *>(buf + 2 * RS * LD);
    __bf16 *Rp = Tp + 2 * TBUF;                               // RM: [2][3][RS][RLD]
    float *wq = DW3 ? reinterpret_cast<float *>(Tp + TBUF) : coefD + 3 * NB;     // [NB / 4][KB][4]   W[k][4 nq .. 4 nq + 3]

    for (int e = tid; e < KB; e += 512) {
        coefA[e] = e < K ? a.asc[e] : 0.f;
        coefA[KB + e] = e < K ? a.ash[e] : 0.f;
        if (XYZ) {
#pragma unroll
            for (int i = 0; i < 4; ++i) coefA[(2 + i) * KB + e] = e < K ? a.xw[i * a.xw_ld + e] : 0.f;
        }
    }
    for (int e = tid; e < NB; e += 512) {
        const bool in = e < N;
        coefD[e] = (in && a.p) ? a.p[e] : 0.f;
        coefD[NB + e] = (in && a.q) ? a.q[e] : 0.f;
        coefD[2 * NB + e] = (in && a.t) ? a.t[e] : 0.f;
    }
    for (int e = tid; e < D4 * KB; e += 512) {
        const int nq = e / KB, k = e % KB;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && 4 * nq < N) w = *reinterpret_cast<const float4 *>(a.W + (long long)k * N + 4 * nq);   // N % 4 == 0
        *reinterpret_cast<float4 *>(&wq[(nq * KB + k) * 4]) = w;
    }
    __syncthreads();

    const long long nstripes = (M + RS - 1) / RS;
    const long long cnt = grp < nstripes ? (nstripes - grp + ngrp - 1) / ngrp : 0;   // stripes of this workgroup

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const int acq = (pt % A4) * 4, dcq = (pt % D4) * 4;
        const bool ain = acq < K, din = dcq < N;
        const float4 casc = *reinterpret_cast<const float4 *>(&coefA[acq]);
        const float4 cash = *reinterpret_cast<const float4 *>(&coefA[KB + acq]);
        float4 xw0 = make_float4(0.f, 0.f, 0.f, 0.f), xw1 = xw0, xw2 = xw0, xb = xw0;
        if (XYZ) {
            xw0 = *reinterpret_cast<const float4 *>(&coefA[2 * KB + acq]);
            xw1 = *reinterpret_cast<const float4 *>(&coefA[3 * KB + acq]);
            xw2 = *reinterpret_cast<const float4 *>(&coefA[4 * KB + acq]);
            xb = *reinterpret_cast<const float4 *>(&coefA[5 * KB + acq]);
        }
        const float4 cp = *reinterpret_cast<const float4 *>(&coefD[dcq]);
        const float4 cq = *reinterpret_cast<const float4 *>(&coefD[NB + dcq]);
        const float4 ct = *reinterpret_cast<const float4 *>(&coefD[2 * NB + dcq]);
        float dbs[4] = {0.f, 0.f, 0.f, 0.f};
        float xs[4] = {0.f, 0.f, 0.f, 0.f};                    // GW: column sums of X (this lane's quad)
        // NSET register sets: the loads of stripes i + 2 .. i + NSET are in flight while stripe i + 1 is staged (29 KB per CU
        // and set).  Round 3 measured two sets within noise of one; round 6 measured three and four (-DPCOPS_BF_NSET=3 / 4,
        // no spills at three): SA1's layer 1 430 -> 1 515 -> 1 605 us -- more requests in flight make the pass SLOWER, so it
        // is not bound by latency x bytes in flight either (profiles/r06_bwd_fused_gw.txt).  Again with the dW half on split operands
        // (lighter consumers): three sets 1 320 -> 1 408-1 426 us, SSG 24.35 -> 23.89 k clouds/s over three alternations.
        constexpr bool B_ = DMODE == A_DYPOOLB;                // compacted rows: one pooling group per 16-row block
        constexpr int NBLK = RS / kBlk;                        // blocks per stripe
        constexpr int QD = 256 / D4;                           // rows between a lane's consecutive D rows (divides 16)
        static_assert(QD <= kBlk && kBlk % QD == 0, "block of row pt / D4 + j QD is (j QD) / 16");
        static_assert(!DW3 || (kBlk % ND == 0), "DW3: a lane's consecutive rows lie inside one 16-row block");
        const int dhb = DW3 ? (ND * (pt / D4)) / kBlk : 0;     // DW3: the block of this lane's rows (0 / 1)
        struct Regs {
            float4 px[NA], pg[ND], py[ND];
            float4 pe;                                         // SIDE: one float4 of the stripe's 32 x 8 side rows (lanes 0..63)
            unsigned pm[(is_pool(DMODE)) ? ND : 1];
            float bw[compact ? NBLK : 1];                      // weight of the first row of each block of the stripe
            int bs0[B_ ? NBLK : 1];
            int s0;                                            // U_: row-in-group of the stripe's first row
        };
#ifndef PCOPS_BF_NSET
#define PCOPS_BF_NSET 2
#endif
        constexpr int NSET = PCOPS_BF_NSET;                   // stripes in flight per producer wave (register sets)
        Regs rs0, rs1, rs2, rs3;                               // (named objects: an array of sets went to scratch)
        // rows of a lane: pt / A4 + j (256 / A4) (strided) -- DW3: NA (ND) CONSECUTIVE rows NA (pt / A4) + j, so that a column's
        // rows leave as one word per piece
        constexpr int XR0 = DW3 ? NA : 1, XRS = DW3 ? 1 : 256 / A4;       // row of (lane, j) = XR0 (pt / A4) + XRS j
        constexpr int DR0 = DW3 ? ND : 1, DRS = DW3 ? 1 : 256 / D4;
        const unsigned xvoff = ain ? (unsigned)(XR0 * (pt / A4) * a.ldx + acq) * 4u : kOOB;
        const unsigned dvoff = din ? (unsigned)(DR0 * (pt / D4) * a.ldy + dcq) * 4u : kOOB;
        const unsigned xstep = (unsigned)XRS * (unsigned)a.ldx * 4u;
        const unsigned dstep = (unsigned)DRS * (unsigned)a.ldy * 4u;
        const long long glast = is_pool(DMODE) ? (M - 1) / a.S : 0;
        const int dcl = din ? dcq : 0;
        constexpr bool U_ = DMODE == A_DYPOOLU;                // one pooling group per stripe
        // Stripes are issued (and staged) in ascending order, so everything a stripe's addresses depend on is kept as a
        // RUNNING wave-uniform value -- base pointers, rows left, (group, row-in-group) of its first row -- and advanced by
        // constants: computed from the stripe number each time (64-bit products, a division, range clamps) it was ~130
        // scalar instructions per stripe, more than the vector work, and the producers' instructions compete with the
        // consumer wave of their SIMD for issue slots.  A descriptor only covers the rows of ITS stripe (32-bit size).
        const int Mi = (int)M, rstep = ngrp * RS;
        int irow = grp * RS;                                   // first row of the next stripe to issue
        const float *ixp = XYZ ? a.off4 + (long long)irow * 4 : a.X + (long long)irow * a.ldx;
        const float *iyp = a.Y + (long long)irow * a.ldy;
        const float *igp = (is_pool(DMODE) ? a.Y : a.G) + (long long)irow * a.ldy;
        const float *iep = SIDE ? a.side + (long long)irow * 8 : nullptr;
        const long long xadv = (long long)rstep * (XYZ ? 4 : a.ldx), yadv = (long long)rstep * a.ldy;
        const int Sg = is_pool(DMODE) ? a.S : 1;
        const int dq = rstep / Sg, dr = rstep % Sg;            // (once per kernel)
        int ig0 = irow / Sg, is0 = irow % Sg;                  // group / row-in-group of irow
        const int nblk = compact ? (Mi + kBlk - 1) / kBlk : 0;
        auto issue = [&](Regs &rg_) {
            if (dbg & 16) return;
            const int left = Mi - irow;
            const unsigned rows_here = (unsigned)(left < RS ? left : RS);
            if (compact && DW3) {
                // a lane's rows lie in ONE block of the stripe: both block records are wave-uniform (scalar) loads, the
                // lane keeps its own block's
                static_assert(NBLK == 2, "two 16-row blocks per stripe");
                int b0 = irow / kBlk, b1 = b0 + 1;
                b0 = b0 < nblk ? b0 : nblk - 1;
                b1 = b1 < nblk ? b1 : nblk - 1;
                const RowBlock r0 = uniform_block(a.blocks, b0), r1 = uniform_block(a.blocks, b1);
                rg_.bw[0] = dhb ? r1.w : r0.w;
                if (B_) {
                    const long long gsel = dhb ? r1.g : r0.g;
                    rg_.bs0[0] = dhb ? r1.s0 : r0.s0;
                    rg_.pg[0] = *reinterpret_cast<const float4 *>(a.gpool + gsel * N + dcl);
                    rg_.pm[0] = *reinterpret_cast<const unsigned *>(a.argmax + gsel * N + dcl);
                }
            } else if (compact) {
#pragma unroll
                for (int h = 0; h < NBLK; ++h) {
                    int bi = irow / kBlk + h;
                    bi = bi < nblk ? bi : nblk - 1;
                    const RowBlock rb = uniform_block(a.blocks, bi);
                    rg_.bw[h] = rb.w;
                    if (B_) {
                        rg_.bs0[h] = rb.s0;
                        rg_.pg[h] = *reinterpret_cast<const float4 *>(a.gpool + (long long)rb.g * N + dcl);
                        rg_.pm[h] = *reinterpret_cast<const unsigned *>(a.argmax + (long long)rb.g * N + dcl);
                    }
                }
            }
            const __amdgpu_buffer_rsrc_t rx = make_rsrc_u32(ixp, rows_here * (XYZ ? 16u : (unsigned)a.ldx * 4u));
            const __amdgpu_buffer_rsrc_t ry = make_rsrc_u32(iyp, rows_here * (unsigned)a.ldy * 4u);
            const __amdgpu_buffer_rsrc_t rg = make_rsrc_u32(igp, rows_here * (unsigned)a.ldy * 4u);
            if (XYZ) {     // 16 bytes per ROW, broadcast over the A4 lanes of a row
#pragma unroll
                for (int j = 0; j < NA; ++j)
                    rg_.px[j] = buf_load4(rx, (unsigned)(XR0 * (pt / A4)) * 16u, (unsigned)j * XRS * 16u);
            } else {
#pragma unroll
                for (int j = 0; j < NA; ++j) rg_.px[j] = buf_load4(rx, xvoff, (unsigned)j * xstep);
            }
            if (SIDE) {
                const __amdgpu_buffer_rsrc_t re = make_rsrc_u32(iep, rows_here * 32u);   // rows beyond M read as zeros
                rg_.pe = buf_load4(re, pt < 64 ? (unsigned)pt * 16u : kOOB, 0u);
                iep += (long long)rstep * 8;
            }
            const PoolRows pr((long long)ig0, is0, Sg);
            rg_.s0 = is0;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                rg_.py[j] = buf_load4(ry, dvoff, (unsigned)j * dstep);
                if (B_) {
                    // loaded per block above
                } else if (is_pool(DMODE)) {
                    if (U_) {
                        if (j == 0) {
                            const long long gi = pr.g0 < glast ? pr.g0 : glast;
                            rg_.pg[0] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                            rg_.pm[0] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
                        }
                    } else {
                        long long gi;
                        unsigned sdummy;
                        pr.split(DR0 * (pt / D4) + j * DRS, glast, gi, sdummy);
                        rg_.pg[j] = *reinterpret_cast<const float4 *>(a.gpool + gi * N + dcl);
                        rg_.pm[j] = *reinterpret_cast<const unsigned *>(a.argmax + gi * N + dcl);
                    }
                } else {
                    rg_.pg[j] = buf_load4(rg, dvoff, (unsigned)j * dstep);
                }
            }
            irow += rstep; ixp += xadv; iyp += yadv; igp += yadv;
            if (is_pool(DMODE)) {
                ig0 += dq; is0 += dr;
                if (is0 >= Sg) { is0 -= Sg; ++ig0; }
            }
        };
        // FULL_: all 32 rows exist and the tile is as wide as the layer -- wave-uniform, true for every stripe but the
        // last; the other variant carries the range selects (a sixth of the producers' instructions, which compete with
        // the consumer wave of the same SIMD for issue slots)
        int srow = grp * RS, sg0 = srow / Sg, ss0 = srow % Sg;  // the stripe stage() is at (same running form)
        // DW3: RW consecutive rows x four columns of one operand -> three pieces, one (2 RW)-byte store per column and piece;
        // column c of an operand of width 4 Q lives in slot  Q (c % 4) + c / 4  (a wave's stores then fall on distinct banks)
        auto put_rows = [&](const auto &v, __bf16 *tb, int slot0, int Q, int cq, int r0, __bf16 *rb = nullptr) {
            constexpr int RW = (int)(sizeof(v) / sizeof(float4));
            __bf16 rh[RM ? RW : 1][4], rm[RM ? RW : 1][4], rl[RM ? RW : 1][4];      // RM: the pieces again, row by row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int slot = slot0 + Q * e + cq;
                // RM: unit (8 rows = 16 bytes) u of slot s sits at unit u ^ ((s / 4) % 4)
                __bf16 *dstp = RM ? tb + slot * RSP + ((((r0 >> 3) ^ (slot >> 2)) & 3) << 3) + (r0 & 7) : tb + slot * RSP + r0;
                __bf16 h[RW], m[RW], l[RW];
#pragma unroll
                for (int i = 0; i < RW; i += 2) {
                    const float xa = e == 0 ? v[i].x : (e == 1 ? v[i].y : (e == 2 ? v[i].z : v[i].w));
                    const float xb = e == 0 ? v[i + 1].x : (e == 1 ? v[i + 1].y : (e == 2 ? v[i + 1].z : v[i + 1].w));
                    bf16x2 hh, mm, ll;
                    split3_pair(xa, xb, hh, mm, ll);
                    h[i] = hh.x; h[i + 1] = hh.y; m[i] = mm.x; m[i + 1] = mm.y; l[i] = ll.x; l[i + 1] = ll.y;
                }
                if constexpr (RM) {
#pragma unroll
                    for (int i = 0; i < RW; ++i) { rh[i][e] = h[i]; rm[i][e] = m[i]; rl[i][e] = l[i]; }
                }
                if constexpr (RW == 4) {
                    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
                    bf16x4_ hv = {h[0], h[1], h[2], h[3]}, mv = {m[0], m[1], m[2], m[3]}, lv = {l[0], l[1], l[2], l[3]};
                    *reinterpret_cast<bf16x4_ *>(dstp) = hv;
                    *reinterpret_cast<bf16x4_ *>(dstp + SLOTS * RSP) = mv;
                    *reinterpret_cast<bf16x4_ *>(dstp + 2 * SLOTS * RSP) = lv;
                } else {
                    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                    static_assert(RW == 2, "two or four consecutive rows per lane");
                    bf16x2_ hv = {h[0], h[1]}, mv = {m[0], m[1]}, lv = {l[0], l[1]};
                    *reinterpret_cast<bf16x2_ *>(dstp) = hv;
                    *reinterpret_cast<bf16x2_ *>(dstp + SLOTS * RSP) = mv;
                    *reinterpret_cast<bf16x2_ *>(dstp + 2 * SLOTS * RSP) = lv;
                }
            }
            if constexpr (RM) {
                if (rb) {                                      // (the dY operand only)
                    typedef __bf16 bf16x4r __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int i = 0; i < RW; ++i) {
                        __bf16 *dr = rb + (r0 + i) * RLD + 4 * cq;
                        const bf16x4r hv = {rh[i][0], rh[i][1], rh[i][2], rh[i][3]}, mv = {rm[i][0], rm[i][1], rm[i][2], rm[i][3]},
                                      lv = {rl[i][0], rl[i][1], rl[i][2], rl[i][3]};
                        *reinterpret_cast<bf16x4r *>(dr) = hv;
                        *reinterpret_cast<bf16x4r *>(dr + RS * RLD) = mv;
                        *reinterpret_cast<bf16x4r *>(dr + 2 * RS * RLD) = lv;
                    }
                }
            }
        };
        auto stage_ = [&](float *dst, __bf16 *tdst, __bf16 *rdst, const Regs &rg_, auto full_) {
            constexpr bool FULL = decltype(full_)::value;
            if (dbg & 8) return;
            const int row0 = srow;
            const PoolRows prs((long long)sg0, ss0, Sg);
            if (SIDE && pt < 64) *reinterpret_cast<float4 *>(&dst[(pt >> 1) * LD + OPC + 4 * (pt & 1)]) = rg_.pe;
            float4 xt[DW3 ? NA : 1];                           // DW3: the lane's X rows, for the transposed pieces
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int r = XR0 * (pt / A4) + j * XRS;
                float4 y = rg_.px[j], x;
                if (XYZ) {
                    float4 o = y;
                    if (!FULL && !(row0 + r < M)) o = make_float4(0.f, 0.f, 0.f, 0.f);
                    y = make_float4(xyz_y(o, xw0.x, xw1.x, xw2.x, xb.x), xyz_y(o, xw0.y, xw1.y, xw2.y, xb.y),
                                    xyz_y(o, xw0.z, xw1.z, xw2.z, xb.z), xyz_y(o, xw0.w, xw1.w, xw2.w, xb.w));
                    if (acq == 0) *reinterpret_cast<float4 *>(&dst[r * LD + OPC]) = o;   // the row's offsets, beside dY
                }
                if (!FULL && !(ain && row0 + r < M)) y = make_float4(0.f, 0.f, 0.f, 0.f);
                x.x = fmaxf(fmaf(y.x, casc.x, cash.x), 0.f);
                x.y = fmaxf(fmaf(y.y, casc.y, cash.y), 0.f);
                x.z = fmaxf(fmaf(y.z, casc.z, cash.z), 0.f);
                x.w = fmaxf(fmaf(y.w, casc.w, cash.w), 0.f);
                if (!FULL && !(ain && row0 + r < M)) x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (GW) { xs[0] += x.x; xs[1] += x.y; xs[2] += x.z; xs[3] += x.w; }
                if (DW3) xt[DW3 ? j : 0] = x;
                else *reinterpret_cast<float4 *>(&dst[r * LD + OXC + acq]) = x;
                *reinterpret_cast<float4 *>(&dst[r * LD + ORC + acq]) = y;
            }
            if constexpr (DW3) put_rows(xt, tdst, 0, KB / 4, pt % A4, NA * (pt / A4));
            float4 dt[DW3 ? ND : 1];
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const int r = DR0 * (pt / D4) + j * DRS;
                const float4 y = rg_.py[j];
                const int hb = DW3 ? 0 : (j * QD) / kBlk;      // block of this row inside the stripe (compile time; DW3: the lane's
                                                               // own block was selected at issue time and sits in entry 0)
                float4 g = rg_.pg[U_ ? 0 : (B_ ? hb : j)];
                if (is_pool(DMODE)) {
                    long long gdummy;
                    unsigned s;
                    if (U_) s = (unsigned)(rg_.s0 + r);
                    else if (B_) s = (unsigned)(rg_.bs0[B_ ? hb : 0] + (r & (kBlk - 1)));
                    else prs.split(r, glast, gdummy, s);
                    const unsigned am = rg_.pm[U_ ? 0 : (B_ ? hb : j)];
                    // gpool arrives MASKED (pcops.h, pcops_mlp_pool_bwd_stats): only the row test is left
                    g.x = ((am & 0xffu) == s) ? g.x : 0.f;
                    g.y = (((am >> 8) & 0xffu) == s) ? g.y : 0.f;
                    g.z = (((am >> 16) & 0xffu) == s) ? g.z : 0.f;
                    g.w = ((am >> 24) == s) ? g.w : 0.f;
                }
                float4 d;
                if (compact && (DW3 ? j == 0 : (j * QD) % kBlk == 0)) {
                    // a row that opens a block (r % 16 == 0) stands for w rows: dY = p.G + w (q.Y + t)
                    const float w = (r & (kBlk - 1)) == 0 ? rg_.bw[compact ? hb : 0] : 1.f;
                    d.x = fmaf(cp.x, g.x, w * fmaf(cq.x, y.x, ct.x));
                    d.y = fmaf(cp.y, g.y, w * fmaf(cq.y, y.y, ct.y));
                    d.z = fmaf(cp.z, g.z, w * fmaf(cq.z, y.z, ct.z));
                    d.w = fmaf(cp.w, g.w, w * fmaf(cq.w, y.w, ct.w));
                } else {
                d.x = fmaf(cp.x, g.x, fmaf(cq.x, y.x, ct.x));
                d.y = fmaf(cp.y, g.y, fmaf(cq.y, y.y, ct.y));
                d.z = fmaf(cp.z, g.z, fmaf(cq.z, y.z, ct.z));
                d.w = fmaf(cp.w, g.w, fmaf(cq.w, y.w, ct.w));
                }
                if (!FULL && !(din && row0 + r < M)) d = make_float4(0.f, 0.f, 0.f, 0.f);
                dbs[0] += d.x; dbs[1] += d.y; dbs[2] += d.z; dbs[3] += d.w;
                if (!RM) *reinterpret_cast<float4 *>(&dst[r * LD + ODC + dcq]) = d;
                if (DW3) dt[DW3 ? j : 0] = d;
            }
            if constexpr (DW3) put_rows(dt, tdst, KB, NB / 4, pt % D4, ND * (pt / D4), rdst);
        };
        auto stage = [&](float *dst, __bf16 *tdst, __bf16 *rdst, const Regs &rg_) {
            if (srow + RS <= Mi && K == KB && N == NB) stage_(dst, tdst, rdst, rg_, std::true_type{});
            else stage_(dst, tdst, rdst, rg_, std::false_type{});
            srow += rstep;
            if (DMODE == A_DYPOOL) {
                sg0 += dq; ss0 += dr;
                if (ss0 >= Sg) { ss0 -= Sg; ++sg0; }
            }
        };
        static_assert(NSET >= 2 && NSET <= 4, "register sets of the producers");
        if (cnt > 0) issue(rs0);
        if (cnt > 1) issue(rs1);
        if (NSET > 2 && cnt > 2) issue(rs2);
        if (NSET > 3 && cnt > 3) issue(rs3);
        if (cnt > 0) {
            stage(buf, Tp, Rp, rs0);
            if (cnt > NSET) issue(rs0);
        }
        __syncthreads();                                       // stripe 0 is in buf[0]
        // stripe j lives in register set j % NSET and goes to stripe buffer j & 1; it is staged while the consumers work on
        // stripe j - 1, and its set is handed to stripe j + NSET at once
        int si = 1;
        for (long long j = 1; j <= cnt; ++j) {
            if (j < cnt) {
                // (a set is named by a wave-uniform index: one copy of the code per set, so that the sets stay in registers)
                auto body = [&](Regs &rg_) {
                    stage(buf + (j & 1) * RS * LD, Tp + (j & 1) * TBUF, Rp + (j & 1) * RBUF, rg_);
                    if (j + NSET < cnt) issue(rg_);
                };
                if (si == 0) body(rs0);
                else if (si == 1) body(rs1);
                else if (NSET > 2 && si == 2) body(rs2);
                else if (NSET > 3) body(rs3);
            }
            __syncthreads();
            si = si + 1 == NSET ? 0 : si + 1;
        }
        float *sdb = buf;
#pragma unroll
        for (int e = 0; e < 4; ++e) sdb[pt * 4 + e] = dbs[e];
        if (GW) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sdb[2048 + pt * 4 + e] = xs[e];      // behind the db scratch and the statistics
        }
        __syncthreads();
        __syncthreads();                                       // (the consumers' statistics hand-over)
    } else {
        // ------------------------------------------------------------------ consumers
        const int ck = wave >> 1, cn = wave & 1;               // dW block: k rows 32 ck .., n columns (NB / 2) cn ..
        const int half = lane >> 5, li = lane & 31;
        const int rh = wave >> 1, cbp = wave & 1;              // dX blocks: rows 16 rh .., columns 32 cbp + {0, 16} ..
        const int c16 = lane & 15, g4 = lane >> 4;
        f32x16 accw[GW ? 1 : TN];                              // GW: the wave's 32 x 32 block (ck, cn) of X^T X
        f32x16 smw[DW3 ? TN : 1];                              // DW3: the five small partial products of each block
        f32x4 accd[2];
#pragma unroll
        for (int j = 0; j < (GW ? 1 : TN); ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) accw[j][v] = 0.f;
#pragma unroll
        for (int j = 0; j < (DW3 ? TN : 1); ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) smw[j][v] = 0.f;
        // GW: the arg-row term X^T (p.G): lane (channel sc, k slice skq) of the 256 consumer lanes holds KPL sums
        constexpr int KPL = NB / 4;
        const int sc = tid & (NB - 1), skq = tid / NB;
        float ssp[GW ? KPL : 1];
#pragma unroll
        for (int j = 0; j < (GW ? KPL : 1); ++j) ssp[j] = 0.f;
        const float spc = (GW && sc < N) ? coefD[sc] : 0.f;    // p of the lane's channel
        constexpr int NGS = DMODE == A_DYPOOLU ? 1 : 4;        // groups a 32-row stripe can meet (S >= 11: launcher)
        const int Sgc = is_pool(DMODE) ? a.S : 1;
        const long long glastc = is_pool(DMODE) ? (M - 1) / Sgc : 0;
        const int crstep = ngrp * RS, cdq = crstep / Sgc, cdr = crstep % Sgc;
        int cg0 = (grp * RS) / Sgc, cs0 = (grp * RS) % Sgc;    // group / row-in-group of the stripe's first row (running)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) accd[b][v] = 0.f;
        const int aoff = half * LD + OXC + ck * 32 + li;
        const int doff = GW ? half * LD + cn * 32 + li : half * LD + ODC + cn * TN * 32 + li;   // GW: the X column block cn
        const int daoff = (16 * rh + c16) * LD + ODC + (DX3 ? 8 : 4) * g4;   // + 16 J (split operands: + 32 J, and + 4)
        // DW3: fragments of the transposed pieces: A = slots 32 ck + li, B = slots KB + 32 (cn TN + y) + li; rows 16 s + 8 half ..
        const int taoff = (32 * ck + li) * RSP + (RM ? 0 : 8 * half), tdoff = (KB + 32 * cn * TN + li) * RSP + (RM ? 0 : 8 * half);
        // RM: rows 16 s + 8 half .. = unit 2 s + half of a slot, stored at unit ^ ((slot / 4) % 4) = unit ^ ((li / 4) % 4) (the blocks
        // start at multiples of 32 slots)
        const int tsw0 = (((0 + half) ^ (li >> 2)) & 3) << 3, tsw1 = (((2 + half) ^ (li >> 2)) & 3) << 3;
        const int reoff = (16 * rh + c16) * RLD + 8 * g4;      // RM: the lane's dX operand fragment in a row-major piece: + 32 J
        const int nyw = NSK ? (N - cn * TN * 32 + 31) / 32 : TN;            // dW blocks of this wave with real columns
        const int jreal = NSK ? (DX3 ? (N + 31) / 32 : (N + 15) / 16) : (DX3 ? NB / 32 : NB / 16);   // steps with real columns
        const int wboff = (g4 * KB + 32 * cbp + c16) * 4;                   // + 16 b * 4, + J * 4 KB * 4
        // this wave's slice of W (NB x 32 columns) stays in REGISTERS for the life of the workgroup: re-read from LDS per
        // stripe it was 64 of the 154 KB of LDS reads a stripe cost -- at 128 B/clk the LDS pipe, not the matrix pipe,
        // set the pace (measured: the kernel without any MFMA or global access still took a quarter of its time)
        constexpr int JN = DX3 ? NB / 32 : NB / 16;            // column groups of dY (16, split operands: 32): one data-gradient step each
        float4 wreg[DX3 ? 1 : JN][2];
        bf16x8 wph[DX3 ? JN : 1][2], wpm[DX3 ? JN : 1][2], wpl[DX3 ? JN : 1][2];      // DX3: the slice as three bf16 pieces
        f32x4 smd[2];                                          // DX3: the small partial products of the two blocks
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) smd[b][v] = 0.f;
        if constexpr (DX3) {
            // B operand of v_mfma_f32_16x16x32_bf16: lane (c16, g4) = output column 32 cbp + 16 b + c16 (row kk of W), eight
            // consecutive dY columns 32 J + 8 g4 .. + 7 = two n-quads of the staging layout wq[n / 4][k][n % 4]
#pragma unroll
            for (int J = 0; J < JN; ++J)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int kk = 32 * cbp + 16 * b + c16, nq = 8 * J + 2 * g4;
                    const float4 f0 = *reinterpret_cast<const float4 *>(&wq[(nq * KB + kk) * 4]);
                    const float4 f1 = *reinterpret_cast<const float4 *>(&wq[((nq + 1) * KB + kk) * 4]);
                    split3(f0, f1, wph[J][b], wpm[J][b], wpl[J][b]);
                }
        } else {
#pragma unroll
        for (int J = 0; J < JN; ++J)
#pragma unroll
            for (int b = 0; b < 2; ++b) wreg[J][b] = *reinterpret_cast<const float4 *>(&wq[wboff + 64 * b + J * 4 * KB * 4]);
        }
        unsigned goff[2][4];                                   // byte offsets of this lane's Gprev elements inside a stripe
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int col = 32 * cbp + 16 * b + c16;
                goff[b][v] = col < K ? (unsigned)((16 * rh + 4 * g4 + v) * K + col) * 4u : kOOB;
            }
        float msc[2], msh[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            msc[b] = coefA[32 * cbp + 16 * b + c16];
            msh[b] = coefA[KB + 32 * cbp + 16 * b + c16];
        }
        float sx[NX ? 2 : 1][NX ? NX : 1];                     // xyz / side form: sums of row input (x) masked gradient
#pragma unroll
        for (int b = 0; b < (NX ? 2 : 1); ++b)
#pragma unroll
            for (int i = 0; i < (NX ? NX : 1); ++i) sx[b][i] = 0.f;
        __syncthreads();
        for (long long i = 0; i < cnt; ++i) {
            const float *sb = buf + (i & 1) * RS * LD;
            const long long row0 = (grp + i * ngrp) * RS;
            constexpr int TW = GW ? 1 : TN;                    // B blocks of the weight-gradient product per wave
            float av_n = DW3 ? 0.f : sb[aoff], dv_n[TW];
#pragma unroll
            for (int y = 0; y < TW; ++y) dv_n[y] = DW3 ? 0.f : sb[doff + 32 * y];
            // GW: this stripe's groups' (masked pooled gradient, arg row) of the lane's channel, requested here and used
            // behind the matrix loop
            float sgv[GW ? NGS : 1];
            unsigned sam[GW ? NGS : 1];
            if constexpr (GW) {
#pragma unroll
                for (int gi = 0; gi < NGS; ++gi) {
                    long long g = (long long)cg0 + gi;
                    g = g < glastc ? g : glastc;
                    const int cc = sc < N ? sc : 0;
                    sgv[gi] = a.gpool[g * N + cc];
                    sam[gi] = a.argmax[g * N + cc];
                }
            }
            constexpr int JP = (RS / 2) / JN;                  // data-gradient steps spread over the RS / 2 row pairs
            // every LDS fragment is requested one use AHEAD (the scheduler would otherwise sink the reads to just in
            // front of their first use and expose the LDS latency)
            float4 da_n = RM ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(&sb[daoff]);
            float4 da2_n = RM ? da_n : (DX3 ? *reinterpret_cast<const float4 *>(&sb[daoff + 4]) : da_n);     // split operands: columns + 4 .. + 7
            float yr[2][4];                                    // the raw Yprev under this wave's Gprev elements
            if constexpr (DW3) {
                // JN data-gradient steps (12 x 16 cycles each), the two 16-row steps of the weight gradient (6 TN x 32 cycles)
                // behind steps 0 and JN / 2; a step's fragments are requested in front of the data-gradient MFMAs before it
                const __bf16 *tb = Tp + (i & 1) * TBUF;
                const __bf16 *rbp = Rp + (i & 1) * RBUF;
                bf16x8 eh_n, em_n, el_n;
                if constexpr (RM) {
                    eh_n = *reinterpret_cast<const bf16x8 *>(rbp + reoff);
                    em_n = *reinterpret_cast<const bf16x8 *>(rbp + RS * RLD + reoff);
                    el_n = *reinterpret_cast<const bf16x8 *>(rbp + 2 * RS * RLD + reoff);
                }
                constexpr int SJ = JN / 2;
#pragma unroll
                for (int J = 0; J < JN; ++J) {
                    const bool wstep = J % SJ == 0;
                    const int s_ = J / SJ;
                    bf16x8 fa[3], fb[TN][3];
                    if (wstep) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            const int so = RM ? (s_ ? tsw1 : tsw0) : 16 * s_;
                            fa[pc] = *reinterpret_cast<const bf16x8 *>(tb + pc * SLOTS * RSP + taoff + so);
#pragma unroll
                            for (int y = 0; y < TN; ++y)
                                fb[y][pc] = *reinterpret_cast<const bf16x8 *>(tb + pc * SLOTS * RSP + tdoff + 32 * y * RSP + so);
                        }
                    }
                    const float4 da = da_n, da2 = da2_n;
                    bf16x8 eh, em, el;
                    if constexpr (RM) { eh = eh_n; em = em_n; el = el_n; }
                    if (J + 1 < JN) {
                        if constexpr (RM) {
                            eh_n = *reinterpret_cast<const bf16x8 *>(rbp + reoff + 32 * (J + 1));
                            em_n = *reinterpret_cast<const bf16x8 *>(rbp + RS * RLD + reoff + 32 * (J + 1));
                            el_n = *reinterpret_cast<const bf16x8 *>(rbp + 2 * RS * RLD + reoff + 32 * (J + 1));
                        } else {
                        da_n = *reinterpret_cast<const float4 *>(&sb[daoff + 32 * (J + 1)]);
                        da2_n = *reinterpret_cast<const float4 *>(&sb[daoff + 32 * (J + 1) + 4]);
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                yr[b][v] = sb[(16 * rh + 4 * g4 + v) * LD + ORC + 32 * cbp + 16 * b + c16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if ((!NSK || J < jreal) && !(dbg & 1)) {
                        if constexpr (!RM) split3(da, da2, eh, em, el);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(el, wph[J][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, wpl[J][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(em, wpm[J][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(em, wph[J][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, wpm[J][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) accd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, wph[J][b], accd[b], 0, 0, 0);
                    }
                    if (wstep && !(dbg & 4)) {
                        // one product at a time over the blocks: consecutive matrix instructions do not share an accumulator
#define PCOPS_MMW(A_, B_, C_)                                                                              \
    _Pragma("unroll") for (int y = 0; y < TN; ++y)                                                         \
        C_[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[A_], fb[y][B_], C_[y], 0, 0, 0)
                        PCOPS_MMW(2, 0, smw);
                        PCOPS_MMW(0, 2, smw);
                        PCOPS_MMW(1, 1, smw);
                        PCOPS_MMW(1, 0, smw);
                        PCOPS_MMW(0, 1, smw);
                        PCOPS_MMW(0, 0, accw);
#undef PCOPS_MMW
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int it = 0; it < RS / 2; ++it) {
                const float av = av_n;
                float dv[TW];
#pragma unroll
                for (int y = 0; y < TW; ++y) dv[y] = dv_n[y];
                if (it + 1 < RS / 2) {
                    av_n = sb[aoff + 2 * (it + 1) * LD];
#pragma unroll
                    for (int y = 0; y < TW; ++y) dv_n[y] = sb[doff + 2 * (it + 1) * LD + 32 * y];
                }
                const bool dostep = it % JP == 0;
                const int J = it / JP;
                const float4 da = da_n, da2 = da2_n;
                if (dostep && J + 1 < JN) {
                    da_n = *reinterpret_cast<const float4 *>(&sb[daoff + (DX3 ? 32 : 16) * (J + 1)]);
                    if (DX3) da2_n = *reinterpret_cast<const float4 *>(&sb[daoff + 32 * (J + 1) + 4]);
                }
                if (it == RS / 2 - 1) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            yr[b][v] = sb[(16 * rh + 4 * g4 + v) * LD + ORC + 32 * cbp + 16 * b + c16];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(dbg & 4)) {
#pragma unroll
                for (int y = 0; y < TW; ++y)
                    if (GW || !NSK || y < nyw) accw[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dv[y], accw[y], 0, 0, 0);
                }
                if (dostep && (!NSK || J < jreal) && !(dbg & 1)) {
                    if constexpr (DX3) {
                        bf16x8 eh, em, el;
                        split3(da, da2, eh, em, el);
                        const int Jc = J < JN ? J : 0;          // (J is a compile-time constant of the unrolled loop)
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(el, wph[Jc][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, wpl[Jc][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(em, wpm[Jc][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(em, wph[Jc][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) smd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, wpm[Jc][b], smd[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 2; ++b) accd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, wph[Jc][b], accd[b], 0, 0, 0);
                    } else {
                    const float de[4] = {da.x, da.y, da.z, da.w};
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float4 w4 = wreg[DX3 ? 0 : J][b];
                            const float we = s_ == 0 ? w4.x : (s_ == 1 ? w4.y : (s_ == 2 ? w4.z : w4.w));
                            accd[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(de[s_], we, accd[b], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if constexpr (GW) {
                // ---- the arg rows of this stripe's groups: row (gi S - cs0 + arg) of the stripe, if it lies inside
                const int rows_here = (int)(M - row0 < RS ? M - row0 : RS);
#pragma unroll
                for (int gi = 0; gi < NGS; ++gi) {
                    const int r = gi * Sgc - cs0 + (int)sam[gi];
                    const bool ok = sc < N && (long long)cg0 + gi <= glastc && (unsigned)r < (unsigned)rows_here && sgv[gi] != 0.f;
                    if (ok) {
                        const float cf = spc * sgv[gi];
                        const float *xr = &sb[r * LD + KPL * skq];
#pragma unroll
                        for (int j = 0; j < KPL; j += 4) {
                            const float4 x4 = *reinterpret_cast<const float4 *>(xr + j);
                            ssp[j] = fmaf(cf, x4.x, ssp[j]); ssp[j + 1] = fmaf(cf, x4.y, ssp[j + 1]);
                            ssp[j + 2] = fmaf(cf, x4.z, ssp[j + 2]); ssp[j + 3] = fmaf(cf, x4.w, ssp[j + 3]);
                        }
                    }
                }
                cg0 += cdq; cs0 += cdr;
                if (cs0 >= Sgc) { cs0 -= Sgc; ++cg0; }
            }
            // ---- Gprev rows of this stripe: mask, column sums, store (before the stripe buffer is handed back).  (Handing
            // them to the next stripe's loop so that they issue under its first MFMAs was measured: no difference.)
            if (!(dbg & 2)) {
                const __amdgpu_buffer_rsrc_t rgp = make_rsrc(NX ? nullptr : a.Gprev + row0 * K, NX ? 0 : (M - row0) * K * 4);
                const bool gnt = a.nt_out != 0;
                float4 ofs[NX ? 4 : 1], of2[SIDE ? 4 : 1];
                if (NX) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        ofs[v] = *reinterpret_cast<const float4 *>(&sb[(16 * rh + 4 * g4 + v) * LD + OPC]);
                        if (SIDE) of2[v] = *reinterpret_cast<const float4 *>(&sb[(16 * rh + 4 * g4 + v) * LD + OPC + 4]);
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float gv = fmaf(yr[b][v], msc[b], msh[b]) > 0.f ? (DX3 ? accd[b][v] + smd[b][v] : accd[b][v]) : 0.f;
                        if (DX3) smd[b][v] = 0.f;
                        s1[b] += gv;
                        s2[b] = fmaf(gv, yr[b][v], s2[b]);
                        if (NX) {
                            sx[b][0] = fmaf(ofs[v].x, gv, sx[b][0]);
                            sx[b][1] = fmaf(ofs[v].y, gv, sx[b][1]);
                            sx[b][2] = fmaf(ofs[v].z, gv, sx[b][2]);
                            if (SIDE) {
                                sx[b][3] = fmaf(ofs[v].w, gv, sx[b][3]);
                                sx[b][4] = fmaf(of2[v].x, gv, sx[b][4]);
                                sx[b][5] = fmaf(of2[v].y, gv, sx[b][5]);
                            }
                        } else {
                            if (gnt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gv), rgp, goff[b][v], 0, 2);
                            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gv), rgp, goff[b][v], 0, 0);
                        }
                        accd[b][v] = 0.f;
                    }
            }
            __syncthreads();
        }
        // dW partial of this workgroup: accw[y][v] = (k = 32 ck + (v&3) + 8 (v>>2) + 4 half, n = (NB/2) cn + 32 y + li)
        float *out = a.part + (long long)grp * K * N;
        if constexpr (GW) {
            // the arg-row term [K][N] from the lanes' sums, the Gram block (ck, cn) [K][K] from the accumulators
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int kk = KPL * skq + j;
                if (kk < K && sc < N) out[(long long)kk * N + sc] = ssp[j];
            }
            float *gout = a.gram_part + (long long)grp * K * K;
            const int jj = cn * 32 + li;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int kk = ck * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
                if (kk < K && jj < K) gout[(long long)kk * K + jj] = accw[0][v];
            }
        } else if constexpr (DW3) {
            // slots back to channels: X slot 16 e + m -> channel 4 m + e; dY slot (NB / 4) e + m -> column 4 m + e
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                const int sn = (cn * TN + y) * 32 + li, nn = 4 * (sn % D4) + sn / D4;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int sk = ck * 32 + (v & 3) + 8 * (v >> 2) + 4 * half, kk = 4 * (sk % A4) + sk / A4;
                    if (kk < K && nn < N) out[(long long)kk * N + nn] = accw[y][v] + smw[y][v];
                }
            }
        } else {
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int nn = (cn * TN + y) * 32 + li;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int kk = ck * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
                if (kk < K && nn < N) out[(long long)kk * N + nn] = accw[y][v];
            }
        }
        }
        // column statistics of Gprev: the four 16-lane sets of a wave own the same columns, the two row halves (waves
        // w, w ^ 2) meet in LDS
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            s1[b] += __shfl_xor(s1[b], 16, 64); s1[b] += __shfl_xor(s1[b], 32, 64);
            s2[b] += __shfl_xor(s2[b], 16, 64); s2[b] += __shfl_xor(s2[b], 32, 64);
        }
#pragma unroll
        for (int b = 0; b < (NX ? 2 : 1); ++b)
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                sx[b][i] += __shfl_xor(sx[b][i], 16, 64);
                sx[b][i] += __shfl_xor(sx[b][i], 32, 64);
            }
        __syncthreads();                                       // matches the producers' db hand-over
        constexpr int NSR = 2 + (NX > 3 ? NX : 3);             // statistics rows per row half: s1, s2, the NX input sums
        float *sst = buf + 256 * 4;                            // [2 row halves][NSR][KB], behind the db scratch
        if (lane < 16) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                sst[(rh * NSR + 0) * KB + 32 * cbp + 16 * b + c16] = s1[b];
                sst[(rh * NSR + 1) * KB + 32 * cbp + 16 * b + c16] = s2[b];
                if (NX) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) sst[(rh * NSR + 2 + i) * KB + 32 * cbp + 16 * b + c16] = sx[b][i];
                }
            }
        }
        if (a.dbpart) {
            const float *sdb = buf;
            for (int c = tid; c < NB; c += 256) {
                const int quad = c >> 2, e = c & 3;
                float sum = 0.f;
                for (int r = quad; r < 256; r += D4) sum += sdb[r * 4 + e];
                if (c < N) a.dbpart[(long long)grp * N + c] = sum;
            }
        }
        if constexpr (GW) {
            const float *sxs = buf + 2048;
            for (int c = tid; c < KB; c += 256) {
                const int quad = c >> 2, e = c & 3;
                float sum = 0.f;
                for (int r = quad; r < 256; r += A4) sum += sxs[r * 4 + e];
                if (c < K) a.xsum_part[(long long)grp * K + c] = sum;
            }
        }
        __syncthreads();
        for (int i = tid; i < (2 + NX) * KB; i += 256) {
            const int which = i / KB, c = i % KB;
            const float v = sst[which * KB + c] + sst[(NSR + which) * KB + c];
            if (c < K) {
                if (which < 2) a.gstats[((long long)grp * 2 + which) * K + c] = v;
                else a.xstats[((long long)grp * NX + which - 2) * K + c] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Gram matrix X^T X of a layer's input in ONE pass over X (round 3), for widths up to 320.  The tiled producer / consumer
// kernel reads a 128-column slice of X per tile side: nine 128 x 128 tiles at K = 320 stream the 671 MB of DGCNN's
// aggregation input more than three times over (the pass sat at ~2-3 TB/s of strided slices, 1.58 ms, and computing only
// the upper tiles changed nothing).  Here a workgroup stages WHOLE rows -- a stripe of 32 x K -- and its four consumer
// waves hold all 32 x 32 blocks of the UPPER triangle in accumulators (K = 320: 55 blocks, 14 per wave, 224 registers):
// X is read once, the lower triangle is mirrored by the launcher, the MFMA work is 55 / 100 of the full square.
struct GramArgs {
    long long M;
    int K, ldx;
    const float *X, *asc, *ash;      // X = raw input; asc == NULL: plain, else relu(X asc + ash)
    float *part;                     // [groups][K][K] partial Gram (upper 32 x 32 blocks written)
    float *xpart;                    // [groups][K] partial column sums
};

constexpr int gram_blocks(int nbk) { return nbk * (nbk + 1) / 2; }
// u-th block of the upper triangle in row-major order -> its block row
constexpr int gram_row_of(int u, int nbk) {
    int i = 0, left = u;
    while (left >= nbk - i) { left -= nbk - i; ++i; }
    return i;
}
constexpr int gram_col_of(int u, int nbk) {
    int i = 0, left = u;
    while (left >= nbk - i) { left -= nbk - i; ++i; }
    return i + left;
}

// (block row, block column) of the U-th upper block as CONSTANTS -- the fragment registers must be named statically
template <int U, int NBK>
struct GramBlk {
    static constexpr int i = gram_row_of(U, NBK), j = gram_col_of(U, NBK);
};
template <int NBK, int U0, typename AccT, int... B>
__device__ __forceinline__ void gram_mfma_step(AccT &acc, const float (&f)[NBK], std::integer_sequence<int, B...>) {
    ((acc[B] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[GramBlk<U0 + B, NBK>::i], f[GramBlk<U0 + B, NBK>::j], acc[B], 0, 0, 0)), ...);
}
template <int NBK, int U0, typename AccT, int... B>
__device__ __forceinline__ void gram_store(const AccT &acc, float *out, int K, int half, int li,
                                           std::integer_sequence<int, B...>) {
    auto one = [&](auto b_) {
        constexpr int b = decltype(b_)::value;
        constexpr int bi = GramBlk<U0 + b, NBK>::i, bj = GramBlk<U0 + b, NBK>::j;
        const int nn = 32 * bj + li;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int kk = 32 * bi + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (kk < K && nn < K) out[(long long)kk * K + nn] = acc[b][v];
        }
    };
    (one(std::integral_constant<int, B>{}), ...);
}

template <int NBK, bool BNRELU>
__global__ __launch_bounds__(512, 1) void gram_full_kernel(GramArgs a) {
    constexpr int KP = 32 * NBK, RS = 32, LD = KP + 4;
    constexpr int NU = gram_blocks(NBK), PER = (NU + 3) / 4;       // upper blocks, blocks per consumer wave
    constexpr int K4 = KP / 4, NV = RS * K4 / 256;                   // float4 per stripe row / per producer thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K;
    const long long M = a.M;
    const int grp = blockIdx.x, ngrp = gridDim.x;
    float *coef = lds;                 // [2][KP]
    float *buf = coef + 2 * KP;        // [2][RS][LD]  | afterwards: column-sum scratch [256][4]
    for (int e = tid; e < KP; e += 512) {
        coef[e] = (BNRELU && e < K) ? a.asc[e] : 0.f;
        coef[KP + e] = (BNRELU && e < K) ? a.ash[e] : 0.f;
    }
    __syncthreads();
    const long long nstripes = (M + RS - 1) / RS;
    const long long cnt = grp < nstripes ? (nstripes - grp + ngrp - 1) / ngrp : 0;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers: whole rows of X
        const int pt = tid - 256;
        float4 px[NV];
        float cs[NV][4];
#pragma unroll
        for (int j = 0; j < NV; ++j) cs[j][0] = cs[j][1] = cs[j][2] = cs[j][3] = 0.f;
        // element e = pt + 256 j of the stripe's [RS][K4] float4 grid: row e / K4, column quad e % K4
        auto issue = [&](long long stripe) {
            const long long row0 = stripe * RS;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + row0 * a.ldx, (M - row0) * a.ldx * 4);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int e = pt + 256 * j, r = e / K4, c = (e % K4) * 4;
                px[j] = buf_load4(rx, c < K ? (unsigned)(r * a.ldx + c) * 4u : kOOB, 0u);
            }
        };
        auto stage = [&](long long stripe, float *dst) {
            const long long row0 = stripe * RS;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int e = pt + 256 * j, r = e / K4, c = (e % K4) * 4;
                float4 x = px[j];
                if (BNRELU) {
                    const float4 sc = *reinterpret_cast<const float4 *>(&coef[c]);
                    const float4 sh = *reinterpret_cast<const float4 *>(&coef[KP + c]);
                    x.x = fmaxf(fmaf(x.x, sc.x, sh.x), 0.f); x.y = fmaxf(fmaf(x.y, sc.y, sh.y), 0.f);
                    x.z = fmaxf(fmaf(x.z, sc.z, sh.z), 0.f); x.w = fmaxf(fmaf(x.w, sc.w, sh.w), 0.f);
                    if (!(c < K && row0 + r < M)) x = make_float4(0.f, 0.f, 0.f, 0.f);   // (relu(shift) of a padded element)
                }
                cs[j][0] += x.x; cs[j][1] += x.y; cs[j][2] += x.z; cs[j][3] += x.w;
                *reinterpret_cast<float4 *>(&dst[r * LD + c]) = x;
            }
        };
        if (cnt > 0) {
            issue(grp);
            stage(grp, buf);
            if (cnt > 1) issue(grp + ngrp);
        }
        __syncthreads();
        for (long long i = 0; i < cnt; ++i) {
            if (i + 1 < cnt) {
                stage(grp + (i + 1) * ngrp, buf + ((i + 1) & 1) * RS * LD);
                if (i + 2 < cnt) issue(grp + (i + 2) * ngrp);
            }
            __syncthreads();
        }
        // column sums: element idx = pt + 256 j of the [RS][K4] grid always sits in column quad idx % K4 -- every (thread, j)
        // partial goes to LDS and each column is summed by ONE thread over its RS partials in a fixed order (deterministic)
        float *scr = buf;                                      // [256 NV][4]
#pragma unroll
        for (int j = 0; j < NV; ++j)
            *reinterpret_cast<float4 *>(&scr[(pt + 256 * j) * 4]) = make_float4(cs[j][0], cs[j][1], cs[j][2], cs[j][3]);
        __syncthreads();
        for (int e = pt; e < K; e += 256) {
            float sum = 0.f;
            for (int idx = e >> 2; idx < 256 * NV; idx += K4) sum += scr[idx * 4 + (e & 3)];
            a.xpart[(long long)grp * K + e] = sum;
        }
        __syncthreads();
    } else {
        // ------------------------------------------------------------------ consumers: the upper blocks, PER per wave
        const int half = lane >> 5, li = lane & 31;
        auto run = [&](auto wv_) {
            constexpr int WV = decltype(wv_)::value;
            constexpr int U0 = WV * PER, U1 = (U0 + PER < NU) ? U0 + PER : NU;
            constexpr int NB_ = U1 > U0 ? U1 - U0 : 0;
            f32x16 acc[NB_ > 0 ? NB_ : 1];
#pragma unroll
            for (int b = 0; b < NB_; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
            __syncthreads();
            for (long long i = 0; i < cnt; ++i) {
                const float *sb = buf + (i & 1) * RS * LD + half * LD + li;
                float fn[NBK];
#pragma unroll
                for (int c = 0; c < NBK; ++c) fn[c] = sb[32 * c];
#pragma unroll
                for (int it = 0; it < RS / 2; ++it) {
                    float f[NBK];
#pragma unroll
                    for (int c = 0; c < NBK; ++c) f[c] = fn[c];
                    if (it + 1 < RS / 2) {
#pragma unroll
                        for (int c = 0; c < NBK; ++c) fn[c] = sb[2 * (it + 1) * LD + 32 * c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    gram_mfma_step<NBK, U0>(acc, f, std::make_integer_sequence<int, NB_>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
            // acc[b][v]: k = 32 i + (v&3) + 8 (v>>2) + 4 half, n = 32 j + li
            gram_store<NBK, U0>(acc, a.part + (long long)grp * K * K, K, half, li, std::make_integer_sequence<int, NB_>{});
            __syncthreads();               // the producers' two column-sum barriers
            __syncthreads();
        };
        if (wave == 0) run(std::integral_constant<int, 0>{});
        else if (wave == 1) run(std::integral_constant<int, 1>{});
        else if (wave == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 3>{});
    }
}

// lower triangle of a symmetric K x K result from its upper one
static __global__ __launch_bounds__(256) void mirror_lower_kernel(int K, float *__restrict__ g) {
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < i; j += 256) g[(long long)i * K + j] = g[(long long)j * K + i];
}

static bool gram_full_on() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_GRAM_FULL");
        return !(e && e[0] == '0');
    }();
    return on;
}

struct PcWgradPlan {
    int tk, tn, kblocks, nblocks, groups;
    bool k96;        // 65..96 input channels: the 96 x 32 consumer layout (wgrad_pc_kernel)
    size_t lds;
};

static bool wgrad_k96_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_WGRAD_K96");
        return !(e && e[0] == '0');
    }();
    return on;
}

static bool wgrad_pc_plan(long long M, int K, int N, int ldx, const void *X, const void *G, const void *Y,
                          const void *gpool, const void *argmax, PcWgradPlan *pl, bool narrow = false) {
    if (M < 8 * 1024) return false;
    if (K % 4 != 0 || N % 4 != 0 || ldx % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(G) & 15) ||
        (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(gpool) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3))
        return false;
    pl->tk = K <= 64 ? 1 : 2;
    pl->tn = N <= 64 ? 1 : (N <= 128 ? 2 : 4);
    if (narrow && pl->tn == 4 && (N + 127) / 128 * 128 < (N + 255) / 256 * 256) pl->tn = 2;   // less padding (N = 320)
    pl->k96 = K > 64 && K <= 96 && pl->tn == 2 && wgrad_k96_enabled();
    const int KB = 64 * pl->tk, NB = 64 * pl->tn;
    pl->kblocks = (K + KB - 1) / KB;
    pl->nblocks = (N + NB - 1) / NB;
    // one persistent workgroup per CU over the whole grid; two for the 64x64 tile (its LDS and register footprints
    // allow it, and one workgroup's stripe barrier then hides under the other's MFMAs; measured slower for 64x128)
    int groups = (pl->tk * pl->tn == 1 ? 512 : 256) / (pl->kblocks * pl->nblocks);
    if (groups < 1) groups = 1;
    const long long maxg = (M + 31) / 32;
    if (groups > maxg) groups = (int)maxg;
    if (groups >= 8) groups &= ~7;
    pl->groups = groups;
    const int rs = pl->tk * pl->tn <= 2 ? 64 : 32;
    pl->lds = (size_t)(6 * KB + 5 * NB + 2 * rs * (KB + NB)) * sizeof(float);
    return pl->lds <= 160 * 1024;
}

struct Bf3WgradPlan {
    int kblocks, nblocks, groups;
    size_t lds;
};

// split-operand weight gradient (wgrad_bf3_kernel): the large layers wider than 64 on both sides
static bool wgrad_bf3_plan(long long M, int K, int N, int ldx, const void *X, const void *G, const void *Y,
                           const void *gpool, const void *argmax, Bf3WgradPlan *pl) {
    const bool on = pcops_get_option(PCOPS_OPT_WGRAD_SPLIT_BF16) != 0;
    if (!on || M < 32768 || K <= 64 || N <= 64) return false;
    if (K % 4 != 0 || N % 4 != 0 || ldx % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(G) & 15) ||
        (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(gpool) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3))
        return false;
    pl->kblocks = (K + 127) / 128;
    pl->nblocks = (N + 127) / 128;
    int groups = 256 / (pl->kblocks * pl->nblocks);
    if (groups < 1) groups = 1;
    const long long maxg = (M + 31) / 32;
    if (groups > maxg) groups = (int)maxg;
    if (groups >= 8) groups &= ~7;
    pl->groups = groups;
    pl->lds = (size_t)(6 * 128 + 5 * 128) * sizeof(float) + (size_t)2 * 3 * 256 * 40 * 2;
    return true;
}

static bool wgrad_pc_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_WGRAD_PC");
        return !(e && e[0] == '0');
    }();
    return on;
}

struct WsWgradPlan {
    int tk, tn, rs, kblocks, nblocks, groups;
    size_t lds;
};

static bool wgrad_ws_plan(long long M, int K, int N, int ldx, const void *X, const void *G, const void *Y,
                          const void *gpool, const void *argmax, WsWgradPlan *pl) {
    if (M < 8 * 1024) return false;
    if (K % 4 != 0 || N % 4 != 0 || ldx % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(G) & 15) ||
        (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(gpool) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3))
        return false;
    if (K <= 64 && N <= 64) { pl->tk = 2; pl->tn = 2; pl->rs = 32; }
    else if (K <= 64) { pl->tk = 2; pl->tn = 4; pl->rs = 32; }
    else { pl->tk = 4; pl->tn = 4; pl->rs = 16; }
    const int KB = 32 * pl->tk, NB = 32 * pl->tn;
    pl->kblocks = (K + KB - 1) / KB;
    pl->nblocks = (N + NB - 1) / NB;
    int groups = 256 / (pl->kblocks * pl->nblocks);    // one persistent workgroup per CU over the whole grid
    if (groups < 1) groups = 1;
    const long long maxg = ((M + pl->rs - 1) / pl->rs + 3) / 4;
    if (groups > maxg) groups = (int)maxg;
    if (groups >= 8) groups &= ~7;
    pl->groups = groups;
    pl->lds = (size_t)(2 * KB + 5 * NB + 4 * pl->rs * (KB + NB) + KB * NB + NB) * sizeof(float);
    return pl->lds <= 160 * 1024;
}

namespace {
// sum partials [P][L] -> out[L] (deterministic): 64 consecutive elements x 4 partial lanes per workgroup
__global__ __launch_bounds__(256) void sum_partials_kernel(int P, long long L, const float *__restrict__ part,
                                                           float *__restrict__ out) {
    __shared__ double sm[4][64];
    const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + c;
    double s = 0.0;
    if (i < L)
        for (int p = pl; p < P; p += 4) s += (double)part[(long long)p * L + i];
    sm[pl][c] = s;
    __syncthreads();
    if (pl == 0 && i < L) out[i] = (float)((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]));
}

// two adjacent reductions in one launch: blocks [0, ceil(L/64)) sum part -> out, the rest sum part2 -> out2
__global__ __launch_bounds__(1024) void sum_partials2_kernel(int P, long long L, const float *__restrict__ part,
                                                             float *__restrict__ out, long long L2,
                                                             const float *__restrict__ part2,
                                                             float *__restrict__ out2) {
    __shared__ double sm[16][64];
    const long long nb1 = (L + 63) / 64;
    const bool second = (long long)blockIdx.x >= nb1;
    const long long len = second ? L2 : L;
    const float *src = second ? part2 : part;
    float *dst = second ? out2 : out;
    const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x - (second ? nb1 : 0)) * 64 + c;
    double s = 0.0;
    if (i < len)
        for (int p = pl; p < P; p += 16) s += (double)src[(long long)p * len + i];
    sm[pl][c] = s;
    __syncthreads();
    if (pl == 0 && i < len) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][c];
        dst[i] = (float)t;
    }
}

// Wt[n][k] = W[k][n]
__global__ __launch_bounds__(256) void transpose_kernel(int K, int N, const float *__restrict__ W,
                                                        float *__restrict__ Wt) {
    __shared__ float t[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (k0 + i < K && n0 + tx < N) t[i][tx] = W[(long long)(k0 + i) * N + n0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (n0 + i < N && k0 + tx < K) Wt[(long long)(n0 + i) * K + k0 + tx] = t[tx][i];
}

// dY = (t + p.G) + q.Y over whole rows (the data gradient of a first layer whose "gather" is the identity: the whole-cloud
// group of sample_and_group_all, pointnet_util.py:59-84) -- it was an addcmul and an in-place addcmul over the tensor
__global__ __launch_bounds__(256) void dy_apply_kernel(long long total4, int N4, const float4 *__restrict__ G,
                                                       const float4 *__restrict__ Y, const float4 *__restrict__ p,
                                                       const float4 *__restrict__ q, const float4 *__restrict__ t,
                                                       float4 *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % N4);
        const float4 g = G[e], y = Y[e], pp = p[c], qq = q[c], tt = t[c];
        out[e] = make_float4((tt.x + g.x * pp.x) + y.x * qq.x, (tt.y + g.y * pp.y) + y.y * qq.y,
                             (tt.z + g.z * pp.z) + y.z * qq.z, (tt.w + g.w * pp.w) + y.w * qq.w);
    }
}

// The algebraic top layer's small operands in ONE launch (they were a transpose, an elementwise product, an addcmul and a
// matrix-vector launch of the small-GEMM kernel):
//   Wt[n][k] = W[k][n],  Wq[k][n] = W[k][n] q[n],  u[n] = q[n] b[n] + t[n]  (tile row 0 writes u),
//   v[k] = sum_n W[k][n] u[n]  (the workgroups behind the tiles: eight rows each, a row per half-wave pair, lanes over n)
__global__ __launch_bounds__(256) void pool_top_prep_kernel(int K, int N, const float *__restrict__ W,
                                                            const float *__restrict__ b, const float *__restrict__ q,
                                                            const float *__restrict__ tt, float *__restrict__ Wt,
                                                            float *__restrict__ Wq, float *__restrict__ u,
                                                            float *__restrict__ v) {
    __shared__ float t[32][33];
    const int tiles_y = (K + 31) / 32;
    if ((int)blockIdx.y >= tiles_y) {                    // v: eight rows per workgroup, numbered along x then y
        if (v == nullptr) return;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int vb = ((int)blockIdx.y - tiles_y) * (int)gridDim.x + (int)blockIdx.x;
        for (int rr = 0; rr < 2; ++rr) {
            const int k = vb * 8 + wave * 2 + rr;
            if (k >= K) continue;                        // (wave-uniform)
            float acc = 0.f;
            for (int n = lane; n < N; n += 64) acc = fmaf(W[(long long)k * N + n], q[n] * b[n] + tt[n], acc);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) v[k] = acc;
        }
        return;
    }
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float qn = n0 + tx < N ? q[n0 + tx] : 0.f;
    for (int i = ty; i < 32; i += 8)
        if (k0 + i < K && n0 + tx < N) {
            const float w = W[(long long)(k0 + i) * N + n0 + tx];
            t[i][tx] = w;
            Wq[(long long)(k0 + i) * N + n0 + tx] = w * qn;
        }
    if (blockIdx.y == 0 && ty == 0 && n0 + tx < N) u[n0 + tx] = qn * b[n0 + tx] + tt[n0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (n0 + i < N && k0 + tx < K) Wt[(long long)(n0 + i) * K + k0 + tx] = t[tx][i];
}

// ... and the sums that close its weight and bias gradient (an in-place add, an outer product, six vector launches and the
// matrix-vector launch for xw = xsum^T W, which the first row of workgroups now adds up per column, k ascending):
//   dW[k][n] = (dW[k][n] + Ssp[k][n]) + xsum[k] u[n],   db[n] = (cfsum[n] + q[n] (xw[n] + R b[n])) + R t[n]
__global__ __launch_bounds__(256) void pool_top_finish_kernel(int K, int N, float R, float *__restrict__ dW,
                                                              const float *__restrict__ Ssp, const float *__restrict__ xsum,
                                                              const float *__restrict__ u, const float *__restrict__ cfsum,
                                                              const float *__restrict__ q, const float *__restrict__ W,
                                                              const float *__restrict__ b, const float *__restrict__ tt,
                                                              float *__restrict__ db) {
    if (blockIdx.y < 8) {
        // db: 32 columns per workgroup (block 8 x + y of the 256-column strip x), eight lanes of k per column, four chains each
        // (a column's K loads would otherwise wait on one another: 61 us for K = 512 with one thread per column)
        __shared__ float red[8][32];
        const int c = threadIdx.x & 31, kl = threadIdx.x >> 5;
        const int n = (blockIdx.x * 8 + blockIdx.y) * 32 + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (n < N) {
            int k = kl;
            for (; k + 24 < K; k += 32) {
                a0 = fmaf(xsum[k], W[(long long)k * N + n], a0);
                a1 = fmaf(xsum[k + 8], W[(long long)(k + 8) * N + n], a1);
                a2 = fmaf(xsum[k + 16], W[(long long)(k + 16) * N + n], a2);
                a3 = fmaf(xsum[k + 24], W[(long long)(k + 24) * N + n], a3);
            }
            for (; k < K; k += 8) a0 = fmaf(xsum[k], W[(long long)k * N + n], a0);
        }
        red[kl][c] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (kl == 0 && n < N) {
            const float xw = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
            db[n] = (cfsum[n] + q[n] * (xw + R * b[n])) + R * tt[n];
        }
        return;
    }
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float un = u[n];
    const int kb = ((int)blockIdx.y - 8) * 8;
    for (int k = kb; k < min(K, kb + 8); ++k) {
        const long long e = (long long)k * N + n;
        dW[e] = (dW[e] + Ssp[e]) + xsum[k] * un;
    }
}

// few partial rows (P <= kFusedRows): column reduction and the per-channel finalisation in ONE launch.
// block = 32 columns x 32 row lanes; the row-lane totals meet in LDS in a fixed order (deterministic)
[[maybe_unused]] constexpr int kFusedRows = 1024;

__device__ __forceinline__ bool fused_col_sums(int P, int N, const float *__restrict__ part, double &s1, double &s2,
                                               int &c) {
    __shared__ double sm[2][32][32];
    c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int g = threadIdx.x >> 5;
    double a1 = 0.0, a2 = 0.0;
    if (c < N)
        for (int p = g; p < P; p += 32) {
            a1 += (double)part[((long long)p * 2 + 0) * N + c];
            a2 += (double)part[((long long)p * 2 + 1) * N + c];
        }
    sm[0][g][threadIdx.x & 31] = a1;
    sm[1][g][threadIdx.x & 31] = a2;
    __syncthreads();
    if (g != 0 || c >= N) return false;
    s1 = s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) { s1 += sm[0][i][threadIdx.x]; s2 += sm[1][i][threadIdx.x]; }
    return true;
}

__global__ __launch_bounds__(1024) void bn_finalize_fused_kernel(int P, int N, double R, const float *__restrict__ part,
                                                                const float *__restrict__ pivot,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float eps,
                                                                float decay, int unbiased,
                                                                float *__restrict__ moving_mean,
                                                                float *__restrict__ moving_var,
                                                                float *__restrict__ mean_o,
                                                                float *__restrict__ rstd_o,
                                                                float *__restrict__ scale_o,
                                                                float *__restrict__ shift_o) {
    double s1, s2;
    int c;
    if (!fused_col_sums(P, N, part, s1, s2, c)) return;
    const double dm = s1 / R;                                    // sums of (y - pivot), (y - pivot)^2: see bn_finalize_kernel
    const double mean = dm + (pivot ? (double)pivot[c] : 0.0);
    double var = s2 / R - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    scale_o[c] = sc;
    shift_o[c] = beta[c] - (float)mean * sc;
    if (moving_mean) {
        const double uv = (unbiased && R > 1.0) ? var * (R / (R - 1.0)) : var;
        moving_mean[c] = decay * moving_mean[c] + (1.f - decay) * (float)mean;
        moving_var[c] = decay * moving_var[c] + (1.f - decay) * (float)uv;
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_coeffs_fused_kernel(int P, int N, double R,
                                                                  const float *__restrict__ part,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd,
                                                                  float *__restrict__ dgamma,
                                                                  float *__restrict__ dbeta, float *__restrict__ p_o,
                                                                  float *__restrict__ q_o, float *__restrict__ t_o) {
    double sg, sgy;
    int c;
    if (!fused_col_sums(P, N, part, sg, sgy, c)) return;
    const double mu = mean[c], rs = rstd[c], g = gamma[c];
    const double dga = (sgy - mu * sg) * rs;
    const double p = g * rs;
    const double q = -p * rs * dga / R;
    const double t = -p * sg / R - q * mu;
    dgamma[c] = (float)dga;
    dbeta[c] = (float)sg;
    p_o[c] = (float)p;
    q_o[c] = (float)q;
    t_o[c] = (float)t;
}

// Gram form of the one-pass backward's weight gradient (bwd_fused_kernel<..., GW>): the P workgroups' partials of
// S = X^T (p.G) [K][N], Gr = X^T X [K][K], xs = X^T 1 [K] and db [N] are summed in a fixed order (in double) and combined,
//   dW[k][n] = S[k][n] + q[n] sum_j Gr[k][j] W[j][n] + xs[k] (q[n] b[n] + t[n]),
// one workgroup per (row k, 128 columns): K <= 64.
__global__ __launch_bounds__(128) void bwd_fused_gw_finish_kernel(int P, int K, int N, const float *__restrict__ spart,
                                                                  const float *__restrict__ dbpart,
                                                                  const float *__restrict__ gpart,
                                                                  const float *__restrict__ xpart, const float *__restrict__ W,
                                                                  const float *__restrict__ bias, const float *__restrict__ q,
                                                                  const float *__restrict__ t, float *__restrict__ dW,
                                                                  float *__restrict__ db) {
    __shared__ double red[2][64];
    __shared__ double gr[64];
    __shared__ double xsr[128];
    const int tid = threadIdx.x, k = blockIdx.x, n = blockIdx.y * 128 + tid;
    {
        const int j = tid & 63, h = tid >> 6;
        double s = 0.0;
        if (j < K)
            for (int p = h; p < P; p += 2) s += (double)gpart[((long long)p * K + k) * K + j];
        red[h][j] = s;
        double x = 0.0;
        for (int p = tid; p < P; p += 128) x += (double)xpart[(long long)p * K + k];
        xsr[tid] = x;
    }
    __syncthreads();
    if (tid < 64) gr[tid] = red[0][tid] + red[1][tid];
    for (int off = 64; off >= 1; off >>= 1) {
        if (tid < off) xsr[tid] += xsr[tid + off];
        __syncthreads();
    }
    const double xs = xsr[0];
    if (n < N) {
        double sS = 0.0;
        for (int p = 0; p < P; ++p) sS += (double)spart[((long long)p * K + k) * N + n];
        double dot = 0.0;
        for (int j = 0; j < K; ++j) dot += gr[j] * (double)W[(long long)j * N + n];
        const double qn = (double)q[n], bn = bias ? (double)bias[n] : 0.0;
        dW[(long long)k * N + n] = (float)(sS + qn * dot + xs * (qn * bn + (double)t[n]));
        if (k == 0 && db) {
            double d = 0.0;
            for (int p = 0; p < P; ++p) d += (double)dbpart[(long long)p * N + n];
            db[n] = (float)d;
        }
    }
}

int reduce_stats(int P, int N, const float *part, double *ws, hipStream_t st) {
    hipLaunchKernelGGL(colreduce_stage1, dim3((N + 31) / 32, 2, kRedSlices), dim3(256), 0, st, P, N, part, ws);
    return pcops_launch_status();
}

}  // namespace
}  // namespace pcops_mlp
using namespace pcops_mlp;

#if PCOPS_PART(0)

// ---------------------------------------------------------------------------------------------
// Algebraic backward of a POOLED top layer  Y = X W + b,  X = relu(bn(Yprev)),  out = max_group relu(bn(Y)).
// Its dY = p.G + q.Y + t has a DENSE part that is affine in X and a SPARSE part (one row per group and channel):
//   dX = (p.G) W^T + X (W diag(q) W^T) + 1 (W (q.b + t))^T
//   dW = X^T (p.G) + (X^T X) (W diag(q)) + (X^T 1) (q.b + t)^T
// so neither gradient needs Y, and the two big products shrink from K x N to K x K (N = 2K in every SA module, N = 3.2K
// .. 8K for the layers pooled over whole clouds).  The sparse parts are G x N row operations:
//   pool_top_addend_kernel   per group: arg-max rows -> compact addend rows  sum_c cf[c] Wt[c][:]  + the row -> slot map
//   pool_top_wsparse_kernel per channel: sum over the groups of cf X[arg row][:]
// cf[g][c] = p[c] gout[g][c] [relu(bn(ysel[g][c])) > 0].  Sums run in ascending channel / group order (deterministic).
template <int NT>      // threads: 1024 for few large groups (a wave set per segment needs the waves), 256 otherwise
__global__ __launch_bounds__(NT) void pool_top_addend_kernel(int S, int C, int Kp, int slots_per_group,
                                                              const float *__restrict__ gout,
                                                              const float *__restrict__ ysel,
                                                              const unsigned char *__restrict__ arg,
                                                              const float *__restrict__ sc, const float *__restrict__ sh,
                                                              const float *__restrict__ p, const float *__restrict__ Wt,
                                                              float *__restrict__ addend, int *__restrict__ rowmap) {
    // hit-driven: the (row, channel) pairs with a non-zero coefficient are compacted, sorted by (row, channel) with a
    // bitonic network in LDS (<= 1024 keys), cut into one segment per row, and a 16-lane set per segment adds the
    // segment's Wt rows in ascending channel order.  Work is O(hits log hits), not O(S C).
    __shared__ float cf[1024];
    __shared__ unsigned key[1024];                 // row << 10 | channel, 0xFFFFFFFF padding
    __shared__ int segstart[1025];                 // first sorted position of segment i
    __shared__ int wcount[16];
    constexpr int CPT = 1024 / NT, NW = NT / 64;     // channels per thread, waves
    const long long g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int s = tid; s < S; s += NT) rowmap[g * S + s] = -1;
    // ---- coefficients and compaction of the hits (C <= 1024: four channels per thread)
    float v[CPT];
    unsigned row[CPT];
    unsigned long long hm[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = i * NT + tid;
        v[i] = 0.f;
        row[i] = 0;
        if (c < C) {
            const long long e = g * C + c;
            v[i] = fmaf(ysel[e], sc[c], sh[c]) > 0.f ? p[c] * gout[e] : 0.f;
            row[i] = arg[e];
            cf[c] = v[i];
        }
        hm[i] = __ballot(v[i] != 0.f);
        if (lane == 0) wcount[i * NW + wave] = __popcll(hm[i]);
        key[c] = 0xFFFFFFFFu;
    }
    __syncthreads();
    int nh = 0, base[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        for (int w = 0; w < NW; ++w) {
            if (w == wave) base[i] = nh;
            nh += wcount[i * NW + w];
        }
    if (nh == 0) return;                           // a chunk that won no channel (layers pooled over whole clouds)
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (v[i] != 0.f)
            key[base[i] + __popcll(hm[i] & ((1ull << lane) - 1ull))] = (row[i] << 10) | (unsigned)(i * NT + tid);
    int n2 = 64;
    while (n2 < nh) n2 <<= 1;
    // ---- bitonic sort of key[0 .. n2): n2 / 2 compare-exchanges per stage
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < n2 / 2; t += NT) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned a = key[lo], b = key[hi];
                if ((a > b) == ((lo & k) == 0)) { key[lo] = b; key[hi] = a; }
            }
        }
    __syncthreads();
    // ---- segments: a new one wherever the row changes (positions i * 256 + tid, i.e. ascending over (i, wave, lane))
    bool first[CPT];
    unsigned long long fm[CPT];
    unsigned mykey[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int pos = i * NT + tid;
        mykey[i] = pos < nh ? key[pos] : 0u;
        first[i] = pos < nh && (pos == 0 || (key[pos - 1] >> 10) != (mykey[i] >> 10));
        fm[i] = __ballot(first[i]);
    }
    __syncthreads();                               // wcount is reused
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < CPT; ++i) wcount[i * NW + wave] = __popcll(fm[i]);
    __syncthreads();
    int nseg = 0, sbase[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        for (int w = 0; w < NW; ++w) {
            if (w == wave) sbase[i] = nseg;
            nseg += wcount[i * NW + w];
        }
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (first[i]) {
            const int si = sbase[i] + __popcll(fm[i] & ((1ull << lane) - 1ull));
            segstart[si] = i * NT + tid;
            rowmap[g * S + (mykey[i] >> 10)] = (int)(g * slots_per_group) + si;
        }
    if (tid == 0) segstart[nseg] = nh;
    __syncthreads();
    // ---- a 16-lane set per segment; lane q of the set owns the float4 columns q, q + 16, ... of the Kp-wide row
    constexpr int KQ = 8;                          // Kp <= 512
    const int sub = lane >> 4, q = lane & 15;
    const int kq = (Kp + 63) / 64;
    for (int si = wave * 4 + sub; si < nseg; si += NW * 4) {
        const int e0 = segstart[si], e1 = segstart[si + 1];
        float4 acc[KQ];
#pragma unroll
        for (int i = 0; i < KQ; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = e0; e < e1; ++e) {
            const int c = key[e] & 1023u;
            const float w = cf[c];
            const float *src = Wt + (long long)c * Kp;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const int k = (i * 16 + q) * 4;
                if (i < kq && k < Kp) {
                    const float4 wt = *reinterpret_cast<const float4 *>(src + k);
                    acc[i].x = fmaf(w, wt.x, acc[i].x); acc[i].y = fmaf(w, wt.y, acc[i].y);
                    acc[i].z = fmaf(w, wt.z, acc[i].z); acc[i].w = fmaf(w, wt.w, acc[i].w);
                }
            }
        }
        float *dst = addend + (g * slots_per_group + si) * (long long)Kp;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const int k = (i * 16 + q) * 4;
            if (i < kq && k < Kp) *reinterpret_cast<float4 *>(dst + k) = acc[i];
        }
    }
}

// one workgroup per channel c, its 4 waves take a quarter of the groups each:
//   Ssp[k][c] = sum_g cf[g][c] X[g S + arg[g][c]][k],  cfsum[c] = sum_g cf[g][c]      (wave partials added in order)
__global__ __launch_bounds__(256) void pool_top_wsparse_kernel(long long G, int S, int C, int Kp,
                                                               const float *__restrict__ gout,
                                                               const float *__restrict__ ysel,
                                                               const unsigned char *__restrict__ arg,
                                                               const float *__restrict__ sc, const float *__restrict__ sh,
                                                               const float *__restrict__ p,
                                                               const float *__restrict__ Yprev,
                                                               const float *__restrict__ psc, const float *__restrict__ psh,
                                                               float *__restrict__ Ssp, float *__restrict__ cfsum) {
    constexpr int KR = 8, U = 8;                                    // Kp <= 64 KR; hit rows in flight
    __shared__ float red[4][64 * KR + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const float scc = sc[c], shc = sh[c], pc = p[c];
    const int kr = (Kp + 63) / 64;                                  // registers in use (wave-uniform)
    float asc[KR], ash[KR], acc[KR];
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        const int k = lane + 64 * i;
        asc[i] = (k < Kp && psc) ? psc[k] : 1.f;
        ash[i] = (k < Kp && psc) ? psh[k] : 0.f;
        acc[i] = 0.f;
    }
    float csum = 0.f;
    const long long gq = (G + 3) / 4;
    const long long gend = (wave + 1) * gq < G ? (wave + 1) * gq : G;
    for (long long g0 = wave * gq; g0 < gend; g0 += 64) {
        const long long g = g0 + lane;
        float cf = 0.f;
        int row = 0;
        if (g < gend) {
            const long long e = g * C + c;
            cf = fmaf(ysel[e], scc, shc) > 0.f ? pc * gout[e] : 0.f;
            row = arg[e];
        }
        unsigned long long hits = __ballot(cf != 0.f);
        while (hits) {                                              // wave-uniform: U hit groups at a time
            float w[U];
            long long r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                w[u] = 0.f;
                r[u] = 0;
                if (hits) {
                    const int l = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    w[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cf), l));
                    r[u] = (g0 + l) * S + __builtin_amdgcn_readlane(row, l);
                }
            }
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                if (i >= kr) break;
                const int k = lane + 64 * i;
                float x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) x[u] = (k < Kp) ? Yprev[r[u] * Kp + k] : 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float xv = psc ? fmaxf(fmaf(x[u], asc[i], ash[i]), 0.f) : x[u];
                    acc[i] = fmaf(w[u], xv, acc[i]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) csum += w[u];
        }
    }
#pragma unroll
    for (int i = 0; i < KR; ++i) red[wave][lane + 64 * i] = acc[i];
    if (lane == 0) red[wave][64 * KR] = csum;
    __syncthreads();
    for (int k = threadIdx.x; k < Kp; k += 256)
        Ssp[(long long)k * C + c] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    if (threadIdx.x == 0)
        cfsum[c] = (red[0][64 * KR] + red[1][64 * KR]) + (red[2][64 * KR] + red[3][64 * KR]);
}

// C [M][N] = A [M][K] B [K][N] for the SMALL products around the big kernels (weights x weights: K x K Gram algebra,
// matrix-vector rows).  One 32 x 32 tile per workgroup so that even a 512 x 512 result fills the chip; the 4 waves split
// K and add their accumulators through LDS in a fixed order.  (Eight waves over K for the long reductions -- the partial tiles
// meeting in the staging area -- were measured in round 6: the SSG and DGCNN steps within noise, 24.55 / 24.46 / 24.63 against
// 24.47 / 24.44 / 25.03 k clouds/s; not kept.  Three chunks per wave in flight instead of one, 48 more registers: 19.4 against
// 15.5 us per launch in the SSG step, profiles/r06_tail_fold_ab.txt; not kept either.)
__device__ __forceinline__ void small_gemm_tile(int bx, int by, int M, int K, int N, const float *__restrict__ A, int lda,
                                                         const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                         int ldc, int transA, int transB, const float *__restrict__ bias,
                                                         float *__restrict__ colsum) {
    // colsum (optional): column sums of B over its K rows next to the product -- the bias gradient of a fully connected layer
    // out of its weight-gradient launch (dW = X^T dY, db = 1^T dY); the tile row 0 adds up what it stages anyway
    __shared__ float As[4][32][17], Bs[4][16][33], red[4][32][33], cs[4][2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = by * 32, n0 = bx * 32;
    const int kchunks = (K + 15) / 16;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    float ra[8], rb[8];
    auto fetch = [&](int ch) {
        const int k0 = ch * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = lane + 64 * i;
            const int r = e >> 4, kk = e & 15;
            // transA: A is stored [K][M] (the product is A^T ...), transB: B is stored [N][K]
            const long long ia = transA ? (long long)(k0 + kk) * lda + m0 + r : (long long)(m0 + r) * lda + k0 + kk;
            ra[i] = (ch < kchunks && m0 + r < M && k0 + kk < K) ? A[ia] : 0.f;
            const int kb = e >> 5, c = e & 31;
            const long long ib = transB ? (long long)(n0 + c) * ldb + k0 + kb : (long long)(k0 + kb) * ldb + n0 + c;
            rb[i] = (ch < kchunks && k0 + kb < K && n0 + c < N) ? B[ib] : 0.f;
        }
    };
    fetch(wave);
    const bool sums = colsum != nullptr && by == 0;
    float bsum = 0.f;                                // column lane & 31 over the rows (lane >> 5) + 2 i of this wave's chunks
    for (int ch = wave; ch < kchunks; ch += 4) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = lane + 64 * i;
            As[wave][e >> 4][e & 15] = ra[i];
            Bs[wave][e >> 5][e & 31] = rb[i];
        }
        if (sums) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bsum += rb[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fetch(ch + 4);                              // the next chunk's loads fly under this chunk's MFMAs
#pragma unroll
        for (int st = 0; st < 8; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wave][lane & 31][2 * st + (lane >> 5)],
                                                       Bs[wave][2 * st + (lane >> 5)][lane & 31], acc, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) red[wave][(v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)][lane & 31] = acc[v];
    cs[wave][lane >> 5][lane & 31] = bsum;
    __syncthreads();
    if (sums && tid < 32 && n0 + tid < N)            // fixed order: the run-to-run bits do not depend on the schedule
        colsum[n0 + tid] = ((cs[0][0][tid] + cs[0][1][tid]) + (cs[1][0][tid] + cs[1][1][tid])) +
                           ((cs[2][0][tid] + cs[2][1][tid]) + (cs[3][0][tid] + cs[3][1][tid]));
    for (int e = tid; e < 32 * 32; e += 256) {
        const int r = e >> 5, c = e & 31;
        if (m0 + r < M && n0 + c < N)
            C[(long long)(m0 + r) * ldc + n0 + c] =
                ((red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c])) + (bias ? bias[n0 + c] : 0.f);
    }
}

__global__ __launch_bounds__(256) void small_gemm_kernel(int M, int K, int N, const float *__restrict__ A, int lda,
                                                         const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                         int ldc, int transA, int transB, const float *__restrict__ bias,
                                                         float *__restrict__ colsum) {
    small_gemm_tile(blockIdx.x, blockIdx.y, M, K, N, A, lda, B, ldb, C, ldc, transA, transB, bias, colsum);
}

// TWO independent products in one launch (the data and the weight gradient of a fully connected layer, dX = dY W^T and
// dW = X^T dY: each alone fills half the chip with 16 us of dependent chunks; the pooled top layer's W diag(q) W^T and gram W diag(q)):
// workgroups [0, tiles0) take the first problem's tiles, the rest the second's -- one call site, the operands chosen by a uniform test
struct SgProblem {
    int M, K, N, lda, ldb, ldc, transA, transB;
    const float *A, *B, *bias;
    float *C, *colsum;
};
__global__ __launch_bounds__(256) void small_gemm_pair_kernel(SgProblem p0, SgProblem p1, int tiles0, int gx0, int gx1) {
    const bool second = (int)blockIdx.x >= tiles0;
    const int t = second ? (int)blockIdx.x - tiles0 : (int)blockIdx.x, gx = second ? gx1 : gx0;
    small_gemm_tile(t % gx, t / gx, second ? p1.M : p0.M, second ? p1.K : p0.K, second ? p1.N : p0.N, second ? p1.A : p0.A,
                    second ? p1.lda : p0.lda, second ? p1.B : p0.B, second ? p1.ldb : p0.ldb, second ? p1.C : p0.C,
                    second ? p1.ldc : p0.ldc, second ? p1.transA : p0.transA, second ? p1.transB : p0.transB,
                    second ? p1.bias : p0.bias, second ? p1.colsum : p0.colsum);
}

#endif  // PCOPS_PART(0)

// ============================================================================ C ABI
// wave-stream kernel or nothing (the xyz-form modes have no tiled fallback)
template <int AM, int EM>
static int launch_gemm_ws_only(GemmArgs &a, hipStream_t st) {
    WsPlan pl;
    if (!(ws_enabled() && ws_plan(a, AM, &pl, ws_kind(EM)))) return PCOPS_ERR_UNSUPPORTED;
    int rc;
    if (AM == A_DYPOOL && a.blocks) rc = launch_gemm_ws<(AM == A_DYPOOL ? A_DYPOOLB : AM), EM>(a, pl, st);
    else if (AM == A_DYPOOL && a.S % 32 == 0) rc = launch_gemm_ws<(AM == A_DYPOOL ? A_DYPOOLU : AM), EM>(a, pl, st);
    else rc = launch_gemm_ws<AM, EM>(a, pl, st);
    return rc;
}

// ---- launchers of the weight-gradient, one-pass-backward and Gram kernels: build parts 4 and 5 (PCOPS_MLP_PART)
PCOPS_HIDDEN int wgrad_impl(WgradArgs &a, float *partial, float *dW, float *db, hipStream_t st);
PCOPS_HIDDEN int bwd_fused_launch(WgradArgs &a, bool xyz, int groups, float *partial, float *dW, float *db, hipStream_t st,
                                  bool side = false, const float *gw_bias = nullptr);
PCOPS_HIDDEN int gram_full_launch(GramArgs &g, int nbk, bool bnrelu, int gg, size_t lds, hipStream_t st);

static int wgrad_legacy_splits(long long M, int K, int N) {
    const int kb = (K + 63) / 64, nb = (N + 127) / 128;
    long long want = (1024 + (long long)kb * nb - 1) / ((long long)kb * nb);
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    long long rows = (M + want - 1) / want;
    rows = ((rows + 7) / 8) * 8;
    if (rows < 8) rows = 8;
    return (int)((M + rows - 1) / rows);
}

#if PCOPS_PART(4)
/* dW[K][N] = A^T dY, db[N] = 1^T dY;  A = X (a_scale==NULL) or relu(X*a_scale + a_shift);
 * dY as in pcops_mlp_gemm_dgrad.  partial: float [splits][K][N] + [splits][N] scratch (caller). */
int wgrad_impl(WgradArgs &a, float *partial, float *dW, float *db, hipStream_t st) {
    const long long M = a.M;
    const int K = a.K, N = a.N, ldx = a.ldx;
    const float *X = a.X, *G = a.G, *Y = a.Y, *gpool = a.gpool;
    const unsigned char *argmax = a.argmax;
    int splits;
    WsWgradPlan pl;
    PcWgradPlan pc;
    // producer/consumer kernel for the pooled forms and the widest tile; the single-role kernel (256 accumulator
    // registers per wave) is ahead on the narrow materialised-G shapes
    if (a.blocks && !(ws_enabled() && wgrad_pc_enabled() && wgrad_pc_plan(M, K, N, ldx, X, G, Y, gpool, argmax, &pc)))
        return PCOPS_ERR_UNSUPPORTED;            // compacted rows: producer/consumer kernel only
    const bool self = a.dmode == A_SELFD;
    if (self && !(ws_enabled() && wgrad_pc_enabled() && wgrad_pc_plan(M, K, N, ldx, X, G, Y, gpool, argmax, &pc, true)))
        return PCOPS_ERR_UNSUPPORTED;            // Gram matrix: producer/consumer kernel only
    Bf3WgradPlan b3;
    pcops_note_pipe(0);
    if (ws_enabled() && wgrad_pc_enabled() && !self && a.amode != A_XYZ &&
        wgrad_bf3_plan(M, K, N, ldx, X, G, Y, gpool, argmax, &b3)) {
        pcops_note_pipe(1);
        splits = b3.groups;
        a.part = partial; a.dbpart = partial + (long long)splits * K * N;
        const dim3 grid(b3.groups, b3.kblocks, b3.nblocks);
        // 65 .. 96 input channels behind a BN + ReLU (MSG's 96 -> 128): the 96 x 32 consumer layout (PCOPS_WGRAD_BF3_K96=0: off)
        static const bool k96_on = [] { const char *e = getenv("PCOPS_WGRAD_BF3_K96"); return !(e && e[0] == '0'); }();
        const bool k96 = k96_on && K <= 96 && K % 4 == 0;
#define PCOPS_B3_LAUNCH(AM_, DM_)                                                                          \
    do {                                                                                                   \
        auto kern = (k96 && AM_ == A_BNRELU) ? wgrad_bf3_kernel<AM_, DM_, AM_ == A_BNRELU> : wgrad_bf3_kernel<AM_, DM_>;   \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
            return PCOPS_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(512), b3.lds, st, a);                                          \
    } while (0)
        if (a.amode == A_BNRELU && a.dmode == A_DY && a.blocks) PCOPS_B3_LAUNCH(A_BNRELU, A_DYW);
        else if (a.amode == A_PLAIN && a.dmode == A_DY && a.blocks) PCOPS_B3_LAUNCH(A_PLAIN, A_DYW);
        else if (a.amode == A_BNRELU && a.dmode != A_DY && a.blocks) PCOPS_B3_LAUNCH(A_BNRELU, A_DYPOOLB);
        else if (a.amode == A_PLAIN && a.dmode != A_DY && a.blocks) PCOPS_B3_LAUNCH(A_PLAIN, A_DYPOOLB);
        else if (a.amode == A_BNRELU && a.dmode == A_DY) PCOPS_B3_LAUNCH(A_BNRELU, A_DY);
        else if (a.amode == A_BNRELU && a.S % 32 == 0) PCOPS_B3_LAUNCH(A_BNRELU, A_DYPOOLU);
        else if (a.amode == A_BNRELU) PCOPS_B3_LAUNCH(A_BNRELU, A_DYPOOL);
        else if (a.dmode == A_DY) PCOPS_B3_LAUNCH(A_PLAIN, A_DY);
        else if (a.S % 32 == 0) PCOPS_B3_LAUNCH(A_PLAIN, A_DYPOOLU);
        else PCOPS_B3_LAUNCH(A_PLAIN, A_DYPOOL);
#undef PCOPS_B3_LAUNCH
    } else if (ws_enabled() && wgrad_pc_enabled() && wgrad_pc_plan(M, K, N, ldx, X, G, Y, gpool, argmax, &pc, self) &&
        (gpool || pc.tn == 4 || a.amode == A_XYZ || a.blocks || self ||
         !wgrad_ws_plan(M, K, N, ldx, X, G, Y, gpool, argmax, &pl))) {
        splits = pc.groups;
        a.part = partial; a.dbpart = partial + (long long)splits * K * N;
        const dim3 grid(pc.groups, pc.kblocks, pc.nblocks);
#define PCOPS_PC_LAUNCH(TK_, TN_, AM_, DM_)                                                                \
    do {                                                                                                   \
        auto kern = wgrad_pc_kernel<TK_, TN_, AM_, DM_, K96_>;                                                \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
            return PCOPS_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(512), pc.lds, st, a);                                          \
    } while (0)
#define PCOPS_PC_MODES(TK_, TN_, K96V_)                                                                    \
    do {                                                                                                   \
        constexpr bool K96_ = K96V_;                                                                       \
        if (a.dmode == A_SELFD && a.amode == A_PLAIN) PCOPS_PC_LAUNCH(TK_, TN_, A_PLAIN, A_SELFD);         \
        else if (a.dmode == A_SELFD) PCOPS_PC_LAUNCH(TK_, TN_, A_BNRELU, A_SELFD);                         \
        else if (a.amode == A_XYZ && a.dmode == A_DY && a.blocks) PCOPS_PC_LAUNCH(TK_, TN_, A_XYZ, A_DYW); \
        else if (a.amode == A_BNRELU && a.dmode == A_DY && a.blocks) PCOPS_PC_LAUNCH(TK_, TN_, A_BNRELU, A_DYW); \
        else if (a.amode == A_PLAIN && a.dmode == A_DY && a.blocks) PCOPS_PC_LAUNCH(TK_, TN_, A_PLAIN, A_DYW);   \
        else if (a.amode == A_XYZ && a.dmode == A_DY) PCOPS_PC_LAUNCH(TK_, TN_, A_XYZ, A_DY);              \
        else if (a.amode == A_XYZ && a.blocks) PCOPS_PC_LAUNCH(TK_, TN_, A_XYZ, A_DYPOOLB);                \
        else if (a.amode == A_BNRELU && a.dmode != A_DY && a.blocks) PCOPS_PC_LAUNCH(TK_, TN_, A_BNRELU, A_DYPOOLB); \
        else if (a.amode == A_PLAIN && a.dmode != A_DY && a.blocks) PCOPS_PC_LAUNCH(TK_, TN_, A_PLAIN, A_DYPOOLB);   \
        else if (a.amode == A_XYZ && a.S % 32 == 0) PCOPS_PC_LAUNCH(TK_, TN_, A_XYZ, A_DYPOOLU);           \
        else if (a.amode == A_XYZ) PCOPS_PC_LAUNCH(TK_, TN_, A_XYZ, A_DYPOOL);                             \
        else if (a.amode == A_BNRELU && a.dmode == A_DY) PCOPS_PC_LAUNCH(TK_, TN_, A_BNRELU, A_DY);        \
        else if (a.amode == A_BNRELU && a.S % 32 == 0) PCOPS_PC_LAUNCH(TK_, TN_, A_BNRELU, A_DYPOOLU);     \
        else if (a.amode == A_BNRELU) PCOPS_PC_LAUNCH(TK_, TN_, A_BNRELU, A_DYPOOL);                       \
        else if (a.dmode == A_DY) PCOPS_PC_LAUNCH(TK_, TN_, A_PLAIN, A_DY);                                \
        else if (a.S % 32 == 0) PCOPS_PC_LAUNCH(TK_, TN_, A_PLAIN, A_DYPOOLU);                             \
        else PCOPS_PC_LAUNCH(TK_, TN_, A_PLAIN, A_DYPOOL);                                                 \
    } while (0)
        if (pc.tk == 1 && pc.tn == 1) PCOPS_PC_MODES(1, 1, false);
        else if (pc.tk == 1 && pc.tn == 2) PCOPS_PC_MODES(1, 2, false);
        else if (pc.tk == 1) PCOPS_PC_MODES(1, 4, false);
        else if (pc.tn == 1) PCOPS_PC_MODES(2, 1, false);
        else if (pc.tn == 2 && pc.k96) PCOPS_PC_MODES(2, 2, true);
        else if (pc.tn == 2) PCOPS_PC_MODES(2, 2, false);
        else PCOPS_PC_MODES(2, 4, false);
#undef PCOPS_PC_MODES
#undef PCOPS_PC_LAUNCH
    } else if (a.amode == A_XYZ) {
        return PCOPS_ERR_UNSUPPORTED;
    } else if (ws_enabled() && wgrad_ws_plan(M, K, N, ldx, X, G, Y, gpool, argmax, &pl)) {
        splits = pl.groups;
        a.part = partial; a.dbpart = partial + (long long)splits * K * N;
        const dim3 grid(pl.groups, pl.kblocks, pl.nblocks);
#define PCOPS_WG_LAUNCH(TK_, TN_, RS_, AM_, DM_)                                                           \
    do {                                                                                                   \
        auto kern = wgrad_ws_kernel<TK_, TN_, RS_, AM_, DM_>;                                              \
        static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                 \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)once;                                                                                        \
        hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, st, a);                                          \
    } while (0)
#define PCOPS_WG_MODES(TK_, TN_, RS_)                                                                      \
    do {                                                                                                   \
        if (a.amode == A_BNRELU && a.dmode == A_DY) PCOPS_WG_LAUNCH(TK_, TN_, RS_, A_BNRELU, A_DY);        \
        else if (a.amode == A_BNRELU) PCOPS_WG_LAUNCH(TK_, TN_, RS_, A_BNRELU, A_DYPOOL);                  \
        else if (a.dmode == A_DY) PCOPS_WG_LAUNCH(TK_, TN_, RS_, A_PLAIN, A_DY);                           \
        else PCOPS_WG_LAUNCH(TK_, TN_, RS_, A_PLAIN, A_DYPOOL);                                            \
    } while (0)
        if (pl.tk == 2 && pl.tn == 2) PCOPS_WG_MODES(2, 2, 32);
        else if (pl.tk == 2) PCOPS_WG_MODES(2, 4, 32);
        else PCOPS_WG_MODES(4, 4, 16);
#undef PCOPS_WG_MODES
#undef PCOPS_WG_LAUNCH
    } else {
        splits = wgrad_legacy_splits(M, K, N);
        a.rows_per_block = (int)((((M + splits - 1) / splits) + 7) / 8 * 8);
        a.part = partial; a.dbpart = partial + (long long)splits * K * N;
        hipLaunchKernelGGL((wgrad_kernel<2, 4>), dim3((K + 63) / 64, (N + 127) / 128, splits), dim3(256), 0, st, a);
    }
    int rc = pcops_launch_status();
    if (rc) return rc;
    // dW and db partials are adjacent ([splits][K*N] then [splits][N]): one launch sums both
    const long long L = (long long)K * N;
    hipLaunchKernelGGL(sum_partials2_kernel, dim3(cdiv(L, 64) + (db ? cdiv(N, 64) : 0)), dim3(1024), 0, st, splits, L,
                       partial, dW, (long long)N, a.dbpart, db);
    return pcops_launch_status();
}
#endif

#if PCOPS_PART(5)
static bool nsk_on() {
    static const bool on = [] {
        const char *e = getenv("PCOPS_BWD_FUSED_NSKIP");
        return !(e && e[0] == '0');
    }();
    return on;
}
// (the split-operand dW variant exists for the 128-column tile over read rows only -- see bwd_fused_launch)
template <int TN, int DM, bool X>
static auto bwd_fused_dw3_kernel() -> void (*)(WgradArgs) {
    if constexpr (TN == 2 && !X) return bwd_fused_kernel<TN, DM, X, false, true, false, false, true>;
    else return nullptr;
}
template <int TN>
static auto bwd_fused_dw3_side_kernel() -> void (*)(WgradArgs) {
    if constexpr (TN == 2) return bwd_fused_kernel<TN, A_DYPOOL, false, false, true, true, false, true>;
    else return nullptr;
}
int bwd_fused_launch(WgradArgs &a, bool xyz, int groups, float *partial, float *dW, float *db, hipStream_t st,
                     bool side, const float *gw_bias) {
    const int K = a.K, N = a.N;
#ifdef PCOPS_BF_DEBUG
    { const char *e = getenv("PCOPS_BF_DEBUG"); a.rows_per_block = e ? atoi(e) : 0; }      // tools/ablate_bwd_fused.py
#endif
    a.part = partial; a.dbpart = db ? partial + (long long)groups * K * N : nullptr;
    const bool gw = a.gram_part != nullptr;     // Gram form of the weight gradient (pcops_mlp_bwd_fused_gw*): partial also
                                                // holds [groups][K][K] + [groups][K] behind the dW / db partials
    const int tn = N <= 64 ? 1 : 2;
    const int NB = 64 * tn;
    const int optv = pcops_get_option(PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16);
    const bool dx3 = optv != 0;
    // split-operand dW half (bwd_fused_kernel<.., DW3>; option value 2, the default): on the 128-column tile of layers whose
    // input is READ.  Measured where it is not built (round 6, profiles/r06_bwd_fused_dw3.txt): the 64-column tile (+7 %: two
    // waves' worth of matrix work saved, the producers' piece splitting added) and the xyz forms (+9 %: their producers
    // already rebuild the first layer per row) are slower with it; nor with the Gram form or the column skip (N <= 96).
    const bool dw3 = optv >= 2 && tn == 2 && !xyz && !a.gram_part && !(nsk_on() && N <= 96);
    const size_t lds = (dw3 && side && PCOPS_BF_RM) ? (size_t)((xyz ? 6 : 2) * 64 + 3 * NB + 2 * 32 * (64 + (side ? 12 : 4))) * sizeof(float) +
                                 (size_t)2 * 3 * (64 + NB) * 32 * 2 + (size_t)2 * 3 * 32 * (NB + 8) * 2
                     : dw3 ? (size_t)((xyz ? 6 : 2) * 64 + 3 * NB + 2 * 32 * (64 + NB + (side ? 12 : 4))) * sizeof(float) +
                                 (size_t)2 * 3 * (64 + NB) * 40 * 2
                           : (size_t)((xyz ? 6 : 2) * 64 + 3 * NB + NB * 64 + 2 * 32 * (2 * 64 + NB + (side ? 12 : 4))) * sizeof(float);
    const bool pooled = a.gpool != nullptr;
    const bool nsk = nsk_on() && tn == 2 && N <= 96;
    pcops_note_pipe(dw3 ? 1 : (dx3 ? 2 : 0));
    if (gw) {
        if (!dx3 || xyz || a.blocks || !a.gpool) return PCOPS_ERR_UNSUPPORTED;
#define PCOPS_BFG_LAUNCH(TN_, DM_, NSK_, SIDE_)                                                            \
    do {                                                                                                   \
        auto kern = bwd_fused_kernel<TN_, DM_, false, NSK_, true, SIDE_, true>;                            \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
            return PCOPS_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL(kern, dim3(groups), dim3(512), lds, st, a);                                     \
    } while (0)
        const bool u = a.S % 32 == 0;
        if (side) {
            if (tn == 1) PCOPS_BFG_LAUNCH(1, A_DYPOOL, false, true);
            else PCOPS_BFG_LAUNCH(2, A_DYPOOL, false, true);
        } else if (tn == 1) {
            if (u) PCOPS_BFG_LAUNCH(1, A_DYPOOLU, false, false);
            else PCOPS_BFG_LAUNCH(1, A_DYPOOL, false, false);
        } else if (nsk) {
            if (u) PCOPS_BFG_LAUNCH(2, A_DYPOOLU, true, false);
            else PCOPS_BFG_LAUNCH(2, A_DYPOOL, true, false);
        } else {
            if (u) PCOPS_BFG_LAUNCH(2, A_DYPOOLU, false, false);
            else PCOPS_BFG_LAUNCH(2, A_DYPOOL, false, false);
        }
#undef PCOPS_BFG_LAUNCH
        int rcg = pcops_launch_status();
        if (rcg) return rcg;
        hipLaunchKernelGGL(bwd_fused_gw_finish_kernel, dim3(K, cdiv(N, 128)), dim3(128), 0, st, groups, K, N, a.part,
                           a.dbpart, a.gram_part, a.xsum_part, a.W, gw_bias, a.q, a.t, dW, db);
        return pcops_launch_status();
    }
#define PCOPS_BF_LAUNCH(TN_, DM_, X_)                                                                      \
    do {                                                                                                   \
        auto kern = dw3 ? bwd_fused_dw3_kernel<TN_, DM_, X_>()                                               \
                  : dx3 ? ((TN_ == 2 && nsk) ? bwd_fused_kernel<TN_, DM_, X_, TN_ == 2, true>                  \
                                              : bwd_fused_kernel<TN_, DM_, X_, false, true>)                    \
                        : ((TN_ == 2 && nsk) ? bwd_fused_kernel<TN_, DM_, X_, TN_ == 2> : bwd_fused_kernel<TN_, DM_, X_>);                                                      \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
            return PCOPS_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL(kern, dim3(groups), dim3(512), lds, st, a);                                     \
    } while (0)
#define PCOPS_BF_MODES(TN_, X_)                                                                            \
    do {                                                                                                   \
        if (!pooled && a.blocks) PCOPS_BF_LAUNCH(TN_, A_DYW, X_);                                          \
        else if (a.blocks) PCOPS_BF_LAUNCH(TN_, A_DYPOOLB, X_);                                            \
        else if (!pooled) PCOPS_BF_LAUNCH(TN_, A_DY, X_);                                                  \
        else if (a.S % 32 == 0) PCOPS_BF_LAUNCH(TN_, A_DYPOOLU, X_);                                       \
        else PCOPS_BF_LAUNCH(TN_, A_DYPOOL, X_);                                                           \
    } while (0)
    if (side) {
        // (the first EdgeConv layer below: pooled groups of k neighbours, plain rows -- pcops_mlp_bwd_fused_edge checks)
#define PCOPS_BF_SIDE(TN_)                                                                                 \
    do {                                                                                                   \
        auto kern = dw3 ? bwd_fused_dw3_side_kernel<TN_>()                                                 \
                  : dx3 ? bwd_fused_kernel<TN_, A_DYPOOL, false, false, true, true>                        \
                        : bwd_fused_kernel<TN_, A_DYPOOL, false, false, false, true>;                      \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
            return PCOPS_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL(kern, dim3(groups), dim3(512), lds, st, a);                                     \
    } while (0)
        if (tn == 1) PCOPS_BF_SIDE(1);
        else PCOPS_BF_SIDE(2);
#undef PCOPS_BF_SIDE
    } else
    if (tn == 1 && xyz) PCOPS_BF_MODES(1, true);
    else if (tn == 1) PCOPS_BF_MODES(1, false);
    else if (xyz) PCOPS_BF_MODES(2, true);
    else PCOPS_BF_MODES(2, false);
#undef PCOPS_BF_MODES
#undef PCOPS_BF_LAUNCH
    int rc = pcops_launch_status();
    if (rc) return rc;
    const long long L = (long long)K * N;
    hipLaunchKernelGGL(sum_partials2_kernel, dim3(cdiv(L, 64) + (db ? cdiv(N, 64) : 0)), dim3(1024), 0, st, groups, L,
                       partial, dW, (long long)N, a.dbpart, db);
    return pcops_launch_status();
}

int gram_full_launch(GramArgs &g, int nbk, bool bnrelu, int gg, size_t lds, hipStream_t st) {
#define PCOPS_GRAM_LAUNCH(NBK_)                                                                            \
    do {                                                                                                   \
        auto kern = bnrelu ? gram_full_kernel<NBK_, true> : gram_full_kernel<NBK_, false>;               \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
            return PCOPS_ERR_LAUNCH;                                                                       \
        hipLaunchKernelGGL(kern, dim3(gg), dim3(512), lds, st, g);                                         \
    } while (0)
        switch (nbk) {
            case 1: PCOPS_GRAM_LAUNCH(1); break;
            case 2: PCOPS_GRAM_LAUNCH(2); break;
            case 3: PCOPS_GRAM_LAUNCH(3); break;
            case 4: PCOPS_GRAM_LAUNCH(4); break;
            case 5: PCOPS_GRAM_LAUNCH(5); break;
            case 6: PCOPS_GRAM_LAUNCH(6); break;
            case 7: PCOPS_GRAM_LAUNCH(7); break;
            case 8: PCOPS_GRAM_LAUNCH(8); break;
            case 9: PCOPS_GRAM_LAUNCH(9); break;
            default: PCOPS_GRAM_LAUNCH(10); break;
        }
#undef PCOPS_GRAM_LAUNCH
    return PCOPS_OK;
}
#endif

#if PCOPS_PART(0)
extern "C" {

int pcops_mlp_stats_rows(int M) {
    // upper bound of the partial-statistics rows any gemm kernel emits for M rows (buffers are sized with
    // it; kernels that launch fewer row groups leave the tail ZERO -- see zero_stats_tail)
    const int tiles = (M + kBM - 1) / kBM;
    int tpb = 1;
    while (tiles / tpb > 512) tpb *= 2;
    return (tiles + tpb - 1) / tpb;
}

unsigned long long pcops_mlp_reduce_workspace_bytes(int N) {
    return (unsigned long long)kRedSlices * 2 * (unsigned long long)N * sizeof(double);
}

static int rows_ok(const pcops_rows_t *rows) {
    if (!rows) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(rows->blocks); PCOPS_REQUIRE_PTR(rows->block_start); PCOPS_REQUIRE_PTR(rows->rows);
    if (reinterpret_cast<uintptr_t>(rows->blocks) & 15) return PCOPS_ERR_UNSUPPORTED;
    return PCOPS_OK;
}
#define PCOPS_ROWS(args_, rows_)                                                      \
    do {                                                                              \
        const int rrc_ = rows_ok(rows_);                                              \
        if (rrc_) return rrc_;                                                        \
        if (rows_) {                                                                  \
            (args_).blocks = static_cast<const RowBlock *>((rows_)->blocks);          \
            (args_).Mdev = (rows_)->rows;                                             \
        }                                                                             \
    } while (0)

int pcops_mlp_gemm_fwd_rows(int M, int K, int N, const float *X, int ldx, const float *pro_scale,
                            const float *pro_shift, const float *W, const float *bias, float *Y,
                            float *stats_partial, const float *stat_pivot, const pcops_rows_t *rows,
                            pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 0 && K >= 1 && N >= 1 && ldx >= K);
    if (M == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(X); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_ARG((pro_scale == nullptr) == (pro_shift == nullptr));
    if (pro_scale) PCOPS_REQUIRE_SHAPE(K % 4 == 0);  // coefficient vectors are read 4 at a time
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = N; a.X = X; a.ldx = ldx; a.v0 = pro_scale; a.v1 = pro_shift;
    a.W = W; a.bias = bias; a.Y = Y; a.ldy = N; a.stats = stats_partial; a.pivot = stats_partial ? stat_pivot : nullptr;
    PCOPS_ROWS(a, rows);
    if (pro_scale) return launch_gemm<A_BNRELU, E_FWD>(a, as_stream(stream));
    return launch_gemm<A_PLAIN, E_FWD>(a, as_stream(stream));
}

int pcops_mlp_gemm_fwd(int M, int K, int N, const float *X, int ldx, const float *pro_scale,
                       const float *pro_shift, const float *W, const float *bias, float *Y,
                       float *stats_partial, const float *stat_pivot, pcops_stream_t stream) {
    return pcops_mlp_gemm_fwd_rows(M, K, N, X, ldx, pro_scale, pro_shift, W, bias, Y, stats_partial, stat_pivot, nullptr,
                                   stream);
}

// groups of S rows that are not whole 32-row tiles (round 5): S % 4 == 0, 8 <= S < 256, a walk of lcm(S, 32) <= 256 rows
static int pool_s4_sub(int S) {
    if (S % 32 == 0 || S % 4 != 0 || S < 8 || S > 255) return 0;
    int l = S;
    while (l % 32 != 0) l += S;
    if (l > 256) return 0;
    const int inv = 65536 / S + 1;
    for (int r = 0; r <= l; ++r)
        if (((r * inv) >> 16) != r / S) return 0;
    return l / 32;
}
static bool pool_s4_enabled() {
    static const bool on = [] { const char *e = getenv("PCOPS_POOL_S4"); return !(e && e[0] == '0'); }();   // kernel A/B only
    return on;
}
static void set_pool_group(GemmArgs &a, int S) {
    const int sub4 = pool_s4_sub(S);
    a.pool_sub = sub4 ? sub4 : S / 32;
    a.pool_s4 = sub4 ? S : 0;
    a.pool_inv = sub4 ? 65536 / S + 1 : 0;
}
static bool fwd_pool_shape_ok(int M, int K, int N, int S) {
    const int sub4 = pool_s4_enabled() ? pool_s4_sub(S) : 0;
    if (sub4) {
        if (M % (32 * sub4) != 0) return false;
    } else if (S < 32 || S > 256 || S % 32 != 0 || M % S != 0) {
        return false;
    }
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = N; a.ldx = K; a.ldy = N;
    set_pool_group(a, S);
    WsPlan pl;
    if (!(ws_enabled() && ws_plan(a, A_BNRELU, &pl, 1))) return false;
    return !sub4 || pl.bf3;                                 // groups that are not whole tiles: split-operand kernels only
}

int pcops_mlp_gemm_fwd_pool_supported(int M, int K, int N, int S) { return fwd_pool_shape_ok(M, K, N, S) ? 1 : 0; }

int pcops_mlp_gemm_fwd_pool(int M, int K, int N, int S, const float *X, int ldx, const float *pro_scale,
                            const float *pro_shift, const float *W, const float *bias, const float *gamma,
                            float *Y, float *stats_partial, const float *stat_pivot, float *ysel,
                            unsigned char *argsel, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && ldx >= K && S >= 1);
    PCOPS_REQUIRE_PTR(X); PCOPS_REQUIRE_PTR(W);
    PCOPS_REQUIRE_ARG((pro_scale == nullptr) == (pro_shift == nullptr));
    PCOPS_REQUIRE_PTR(gamma); PCOPS_REQUIRE_PTR(ysel); PCOPS_REQUIRE_PTR(argsel);
    if (!fwd_pool_shape_ok(M, K, N, S) || ldx != K) return PCOPS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(ysel) & 15) || (reinterpret_cast<uintptr_t>(argsel) & 3))
        return PCOPS_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = N; a.X = X; a.ldx = ldx; a.v0 = pro_scale; a.v1 = pro_shift;
    a.W = W; a.bias = bias; a.Y = Y; a.ldy = N; a.stats = stats_partial; a.pivot = stats_partial ? stat_pivot : nullptr;
    set_pool_group(a, S);
    a.pgamma = gamma; a.ysel = ysel; a.psel = argsel;
    WsPlan pl;
    if (!ws_plan(a, A_BNRELU, &pl)) return PCOPS_ERR_UNSUPPORTED;   // pointer alignment
    if (!pro_scale) return launch_gemm<A_PLAIN, E_FWD>(a, as_stream(stream));      // the stack's raw input
    return launch_gemm<A_BNRELU, E_FWD>(a, as_stream(stream));
}

int pcops_mlp_gemm_fwd_pool_rows(int M, int K, int N, const float *X, int ldx, const float *pro_scale,
                                 const float *pro_shift, const float *W, const float *bias, const float *gamma,
                                 float *Y, float *stats_partial, const float *stat_pivot, float *ypart,
                                 unsigned char *ppart, const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && ldx >= K);
    PCOPS_REQUIRE_PTR(X); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(pro_scale);
    PCOPS_REQUIRE_PTR(pro_shift); PCOPS_REQUIRE_PTR(gamma); PCOPS_REQUIRE_PTR(ypart); PCOPS_REQUIRE_PTR(ppart);
    PCOPS_REQUIRE_PTR(rows);
    if (ldx != K || (reinterpret_cast<uintptr_t>(ypart) & 15) || (reinterpret_cast<uintptr_t>(ppart) & 3))
        return PCOPS_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = N; a.X = X; a.ldx = ldx; a.v0 = pro_scale; a.v1 = pro_shift;
    a.W = W; a.bias = bias; a.Y = Y; a.ldy = N; a.stats = stats_partial; a.pivot = stats_partial ? stat_pivot : nullptr;
    a.pool_sub = 1; a.pgamma = gamma; a.ysel = ypart; a.psel = ppart;       // one partial per 16-row block
    PCOPS_ROWS(a, rows);
    WsPlan pl;
    if (!(ws_enabled() && ws_plan(a, A_BNRELU, &pl))) return PCOPS_ERR_UNSUPPORTED;
    return launch_gemm<A_BNRELU, E_FWD>(a, as_stream(stream));
}

int pcops_mlp_gemm_fwd_pool_rows_supported(int M, int K, int N) {
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = N; a.ldx = K; a.ldy = N; a.pool_sub = 1;
    WsPlan pl;
    return ws_enabled() && ws_plan(a, A_BNRELU, &pl) ? 1 : 0;
}

int pcops_mlp_pool_combine_rows(long long G, int C, const float *ypart, const unsigned char *ppart,
                                const float *gamma, const float *scale, const float *shift,
                                const pcops_rows_t *rows, float *out, unsigned char *argmax, float *ysel,
                                pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 0 && C >= 1);
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(ypart); PCOPS_REQUIRE_PTR(ppart); PCOPS_REQUIRE_PTR(gamma); PCOPS_REQUIRE_PTR(scale);
    PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(out); PCOPS_REQUIRE_PTR(rows);
    const int rrc = rows_ok(rows);
    if (rrc) return rrc;
    const long long total = G * C;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(pool_combine_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total, C, ypart, ppart,
                       gamma, scale, shift, rows->block_start, out, argmax, ysel);
    return pcops_launch_status();
}

int pcops_mlp_pool_select(long long G, int C, const float *ysel, const float *scale, const float *shift,
                          float *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 0 && C >= 1);
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(ysel); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(out);
    const long long total = G * C;
    const unsigned grid = cdiv(total, 256) < 16384u ? cdiv(total, 256) : 16384u;
    hipLaunchKernelGGL(pool_select_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total, C, ysel, scale,
                       shift, out);
    return pcops_launch_status();
}

int pcops_mlp_bn_finalize(int P, int N, long long R, const float *stats_partial, const float *stat_pivot,
                          void *workspace, const float *gamma, const float *beta, float eps, float decay,
                          int unbiased_moving_var, float *moving_mean, float *moving_var, float *mean,
                          float *rstd, float *scale, float *shift, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(P >= 1 && N >= 1 && R >= 1);
    PCOPS_REQUIRE_PTR(stats_partial); PCOPS_REQUIRE_PTR(workspace); PCOPS_REQUIRE_PTR(gamma);
    PCOPS_REQUIRE_PTR(beta); PCOPS_REQUIRE_PTR(mean); PCOPS_REQUIRE_PTR(rstd); PCOPS_REQUIRE_PTR(scale);
    PCOPS_REQUIRE_PTR(shift);
    PCOPS_REQUIRE_ARG((moving_mean == nullptr) == (moving_var == nullptr));
    hipStream_t st = as_stream(stream);
    if (P <= kFusedRows) {
        hipLaunchKernelGGL(bn_finalize_fused_kernel, dim3((N + 31) / 32), dim3(1024), 0, st, P, N, (double)R,
                           stats_partial, stat_pivot, gamma, beta, eps, decay, unbiased_moving_var, moving_mean, moving_var, mean,
                           rstd, scale, shift);
        return pcops_launch_status();
    }
    double *ws = static_cast<double *>(workspace);
    int rc = reduce_stats(P, N, stats_partial, ws, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, (double)R, ws, stat_pivot,
                       gamma, beta, eps, decay, unbiased_moving_var, moving_mean, moving_var, mean, rstd, scale, shift);
    return pcops_launch_status();
}

int pcops_mlp_bn_eval_coeffs(int N, const float *gamma, const float *beta, const float *moving_mean,
                             const float *moving_var, float eps, float *scale, float *shift,
                             pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(N >= 1);
    PCOPS_REQUIRE_PTR(gamma); PCOPS_REQUIRE_PTR(beta); PCOPS_REQUIRE_PTR(moving_mean);
    PCOPS_REQUIRE_PTR(moving_var); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift);
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), N, gamma,
                       beta, moving_mean, moving_var, eps, scale, shift);
    return pcops_launch_status();
}

int pcops_mlp_bn_relu_maxpool(long long G, int S, int C, const float *Y, const float *scale,
                              const float *shift, float *out, unsigned char *argmax, float *ysel,
                              pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 0 && S >= 1 && S <= 256 && C >= 4 && C % 4 == 0);
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(out);
    const long long total = G * (C / 4);
    const unsigned grid = cdiv(total, 256) < 32768u ? cdiv(total, 256) : 32768u;
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(grid), dim3(256), 0, as_stream(stream), G, S, C, Y, scale,
                       shift, out, argmax, ysel);
    return pcops_launch_status();
}

int pcops_mlp_bn_relu_maxpool_rows(long long G, int C, const float *Y, const float *scale, const float *shift,
                                   const pcops_rows_t *rows, float *out, unsigned char *argmax, float *ysel,
                                   pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 0 && C >= 4 && C % 4 == 0);
    if (G == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(out);
    PCOPS_REQUIRE_PTR(rows);
    const int rrc = rows_ok(rows);
    if (rrc) return rrc;
    const long long total = G * (C / 4);
    const unsigned grid = cdiv(total, 256) < 32768u ? cdiv(total, 256) : 32768u;
    hipLaunchKernelGGL(bn_relu_maxpool_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), G, C, Y, scale,
                       shift, rows->block_start, out, argmax, ysel);
    return pcops_launch_status();
}

int pcops_mlp_bn_relu_apply(long long R, int C, const float *Y, const float *scale, const float *shift,
                            float *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(R >= 0 && C >= 4 && C % 4 == 0);
    if (R == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(out);
    const long long total4 = R * C / 4;
    const unsigned grid = cdiv(total4, 256) < 32768u ? cdiv(total4, 256) : 32768u;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total4, C, Y, scale,
                       shift, out);
    return pcops_launch_status();
}

// ---- backward ---------------------------------------------------------------------------------
int pcops_mlp_bwd_stats_rows(long long R) { return (int)((R + 511) / 512); }
int pcops_mlp_bwd_pool_stats_rows(long long G) { return (int)((G + 15) / 16); }

/* Gm = Gout * [relu(bn(Y)) > 0]; partial sums (sum Gm, sum Gm*Y) -> stats [pcops_mlp_bwd_stats_rows(R)][2][C] */
int pcops_mlp_relu_mask_stats(long long R, int C, const float *Gout, const float *Y, const float *scale,
                              const float *shift, float *Gm, float *stats_partial, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(R >= 1 && C >= 4 && C % 4 == 0);
    PCOPS_REQUIRE_PTR(Gout); PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(scale); PCOPS_REQUIRE_PTR(shift);
    PCOPS_REQUIRE_PTR(Gm); PCOPS_REQUIRE_PTR(stats_partial);
    const int c4n = C / 4;
    PCOPS_REQUIRE_SHAPE(c4n >= 256 || 256 % c4n == 0);
    const int rl = c4n >= 256 ? 1 : 256 / c4n;
    const size_t lds = (size_t)rl * 2 * C * sizeof(float);
    if (lds > 64 * 1024) return PCOPS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(relu_mask_stats_kernel, dim3(pcops_mlp_bwd_stats_rows(R)), dim3(256), lds,
                       as_stream(stream), R, C, Gout, Y, scale, shift, Gm, stats_partial, 512);
    return pcops_launch_status();
}

int pcops_mlp_pool_bwd_stats(long long G, int C, const float *gpool, const float *ysel, const float *scale,
                             const float *shift, float *stats_partial, float *gmasked, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 1 && C >= 4 && C % 4 == 0);
    PCOPS_REQUIRE_PTR(gpool); PCOPS_REQUIRE_PTR(ysel); PCOPS_REQUIRE_PTR(scale);
    PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(stats_partial);
    const int c4n = C / 4;
    const int rl = c4n >= 256 ? 1 : 256 / c4n;        // threads beyond rl * c4n idle when c4n does not divide 256
    const size_t lds = (size_t)rl * 2 * C * sizeof(float);
    if (lds > 64 * 1024) return PCOPS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_bwd_stats_sel_kernel, dim3(pcops_mlp_bwd_pool_stats_rows(G)), dim3(256), lds,
                       as_stream(stream), G, C, gpool, ysel, scale, shift, stats_partial, 16, gmasked);
    return pcops_launch_status();
}

int pcops_mlp_pool_bwd_stats_sum(long long G, int C, const float *ga, long long lda, const float *gb, long long ldb,
                                 const float *ysel, const float *scale, const float *shift, float *stats_partial,
                                 float *gsum, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(G >= 1 && C >= 4 && C % 4 == 0 && lda >= C && lda % 4 == 0 && (!gb || (ldb >= C && ldb % 4 == 0)));
    PCOPS_REQUIRE_PTR(ga); PCOPS_REQUIRE_PTR(ysel); PCOPS_REQUIRE_PTR(scale);
    PCOPS_REQUIRE_PTR(shift); PCOPS_REQUIRE_PTR(stats_partial); PCOPS_REQUIRE_PTR(gsum);
    PCOPS_REQUIRE_ARG(((uintptr_t)ga | (uintptr_t)gb | (uintptr_t)gsum) % 16 == 0);
    const int c4n = C / 4;
    const int rl = c4n >= 256 ? 1 : 256 / c4n;
    const size_t lds = (size_t)rl * 2 * C * sizeof(float);
    if (lds > 64 * 1024) return PCOPS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_bwd_stats_sum_kernel, dim3(pcops_mlp_bwd_pool_stats_rows(G)), dim3(256), lds,
                       as_stream(stream), G, C, ga, lda, gb, ldb, ysel, scale, shift, stats_partial, 16, gsum);
    return pcops_launch_status();
}

int pcops_mlp_bn_bwd_coeffs(int P, int N, long long R, const float *stats_partial, void *workspace,
                            const float *gamma, const float *mean, const float *rstd, float *dgamma,
                            float *dbeta, float *p, float *q, float *t, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(P >= 1 && N >= 1 && R >= 1);
    PCOPS_REQUIRE_PTR(stats_partial); PCOPS_REQUIRE_PTR(workspace); PCOPS_REQUIRE_PTR(gamma);
    PCOPS_REQUIRE_PTR(mean); PCOPS_REQUIRE_PTR(rstd); PCOPS_REQUIRE_PTR(dgamma); PCOPS_REQUIRE_PTR(dbeta);
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    hipStream_t st = as_stream(stream);
    if (P <= kFusedRows) {
        hipLaunchKernelGGL(bn_bwd_coeffs_fused_kernel, dim3((N + 31) / 32), dim3(1024), 0, st, P, N, (double)R,
                           stats_partial, gamma, mean, rstd, dgamma, dbeta, p, q, t);
        return pcops_launch_status();
    }
    double *ws = static_cast<double *>(workspace);
    int rc = reduce_stats(P, N, stats_partial, ws, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, (double)R, ws, gamma,
                       mean, rstd, dgamma, dbeta, p, q, t);
    return pcops_launch_status();
}

/* Gprev[M][Nout] = mask . (dY[M][K] Wt[K][Nout]),  dY = p.G + q.Y + t  (or the pooled form when gpool!=NULL)
 * mask = [relu(bn_prev(Yprev)) > 0] when Yprev != NULL (then stats_partial gets (sum Gprev, sum Gprev*Yprev)),
 * no mask / no stats when Yprev == NULL (first layer: plain dX). */
int pcops_mlp_gemm_dgrad(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                         const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                         int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                         const float *Yprev, const float *prev_scale, const float *prev_shift, float *Gprev,
                         float *stats_partial, pcops_stream_t stream) {
    return pcops_mlp_gemm_dgrad_rows(M, K, Nout, G, Y, p, q, t, gpool, argmax, S, pool_scale, pool_shift, Wt, Yprev,
                                     prev_scale, prev_shift, Gprev, stats_partial, nullptr, stream);
}

int pcops_mlp_gemm_dgrad_rows(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                              const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                              int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                              const float *Yprev, const float *prev_scale, const float *prev_shift, float *Gprev,
                              float *stats_partial, const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 0 && K >= 1 && Nout >= 1);
    if (M == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(Wt); PCOPS_REQUIRE_PTR(Gprev);
    PCOPS_REQUIRE_SHAPE(K % 4 == 0);
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = Nout; a.X = G; a.X2 = Y; a.ldx = K; a.v0 = p; a.v1 = q; a.v2 = t;
    a.v3 = pool_scale; a.v4 = pool_shift; a.gpool = gpool; a.argmax = argmax; a.S = S > 0 ? S : 1;
    a.W = Wt; a.Y = Gprev; a.ldy = Nout; a.Yprev = Yprev; a.msc = prev_scale; a.msh = prev_shift;
    a.stats = stats_partial;
    PCOPS_ROWS(a, rows);
    hipStream_t st = as_stream(stream);
    if (gpool) {
        PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale); PCOPS_REQUIRE_PTR(pool_shift);
        a.X = Y;  // alignment probe only
        if (Yprev) return launch_gemm<A_DYPOOL, E_MASK>(a, st);
        return launch_gemm<A_DYPOOL, E_PLAIN>(a, st);
    }
    PCOPS_REQUIRE_PTR(G);
    if (Yprev) return launch_gemm<A_DY, E_MASK>(a, st);
    return launch_gemm<A_DY, E_PLAIN>(a, st);
}

int pcops_mlp_xyz_supported(int M, int C1, int N2) {
    // second layer (C1 -> N2) forward / weight gradient with an A_XYZ operand, and its data gradient (N2 -> C1) with
    // the E_MASKX epilogue, all on the wave-stream kernels
    GemmArgs f = {};
    f.M = M; f.K = C1; f.N = N2; f.ldx = C1; f.ldy = N2;
    GemmArgs d = {};
    d.M = M; d.K = N2; d.N = C1; d.ldx = N2; d.ldy = C1;
    WsPlan pl;
    PcWgradPlan pc;
    return ws_enabled() && wgrad_pc_enabled() && C1 % 4 == 0 && ws_plan(f, A_XYZ, &pl) && ws_plan(d, A_DY, &pl) &&
           wgrad_pc_plan(M, C1, N2, C1, nullptr, nullptr, nullptr, nullptr, nullptr, &pc) ? 1 : 0;
}

/* forward of the layer FOLLOWING an arithmetic first layer: X = relu(pro_scale * y + pro_shift), y rebuilt from
 * off4 [M][4] and xyzw [4][K]; otherwise pcops_mlp_gemm_fwd */
int pcops_mlp_gemm_fwd_xyz(int M, int K, int N, const float *off4, const float *xyzw, const float *pro_scale,
                           const float *pro_shift, const float *W, const float *bias, float *Y,
                           float *stats_partial, const float *stat_pivot, pcops_stream_t stream) {
    return pcops_mlp_gemm_fwd_xyz_rows(M, K, N, off4, xyzw, pro_scale, pro_shift, W, bias, Y, stats_partial, stat_pivot,
                                       nullptr, stream);
}

int pcops_mlp_gemm_fwd_xyz_rows(int M, int K, int N, const float *off4, const float *xyzw, const float *pro_scale,
                                const float *pro_shift, const float *W, const float *bias, float *Y,
                                float *stats_partial, const float *stat_pivot, const pcops_rows_t *rows,
                                pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 4 && K % 4 == 0 && N >= 1);
    PCOPS_REQUIRE_PTR(off4); PCOPS_REQUIRE_PTR(xyzw); PCOPS_REQUIRE_PTR(pro_scale); PCOPS_REQUIRE_PTR(pro_shift);
    PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(Y);
    if (reinterpret_cast<uintptr_t>(off4) & 15) return PCOPS_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = N; a.X = nullptr; a.ldx = K; a.v0 = pro_scale; a.v1 = pro_shift;
    a.off4 = off4; a.xw = xyzw; a.xw_ld = K;
    a.W = W; a.bias = bias; a.Y = Y; a.ldy = N; a.stats = stats_partial; a.pivot = stats_partial ? stat_pivot : nullptr;
    PCOPS_ROWS(a, rows);
    return launch_gemm_ws_only<A_XYZ, E_FWD>(a, as_stream(stream));
}

/* pcops_mlp_gemm_dgrad whose PREVIOUS layer is the arithmetic first layer: the ReLU mask and the (sum G, sum G*Y)
 * statistics use y rebuilt from off4 [M][4] and xyzw [4][Nout] instead of a stored Yprev */
int pcops_mlp_gemm_dgrad_xyz(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                             const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                             int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                             const float *off4, const float *xyzw, const float *prev_scale,
                             const float *prev_shift, float *Gprev, float *stats_partial, float *xyz_stats,
                             pcops_stream_t stream) {
    return pcops_mlp_gemm_dgrad_xyz_rows(M, K, Nout, G, Y, p, q, t, gpool, argmax, S, pool_scale, pool_shift, Wt, off4,
                                         xyzw, prev_scale, prev_shift, Gprev, stats_partial, xyz_stats, nullptr, stream);
}

int pcops_mlp_gemm_dgrad_xyz_rows(int M, int K, int Nout, const float *G, const float *Y, const float *p,
                                  const float *q, const float *t, const float *gpool, const unsigned char *argmax,
                                  int S, const float *pool_scale, const float *pool_shift, const float *Wt,
                                  const float *off4, const float *xyzw, const float *prev_scale,
                                  const float *prev_shift, float *Gprev, float *stats_partial, float *xyz_stats,
                                  const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 4 && K % 4 == 0 && Nout >= 1);
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(Wt); PCOPS_REQUIRE_PTR(off4); PCOPS_REQUIRE_PTR(xyzw);
    PCOPS_REQUIRE_ARG(Gprev != nullptr || xyz_stats != nullptr);
    if (xyz_stats) PCOPS_REQUIRE_PTR(stats_partial);
    PCOPS_REQUIRE_PTR(prev_scale); PCOPS_REQUIRE_PTR(prev_shift);
    if (reinterpret_cast<uintptr_t>(off4) & 15) return PCOPS_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.M = M; a.K = K; a.N = Nout; a.X = G; a.X2 = Y; a.ldx = K; a.v0 = p; a.v1 = q; a.v2 = t;
    a.v3 = pool_scale; a.v4 = pool_shift; a.gpool = gpool; a.argmax = argmax; a.S = S > 0 ? S : 1;
    a.W = Wt; a.Y = Gprev; a.ldy = Nout; a.msc = prev_scale; a.msh = prev_shift;
    a.off4 = off4; a.xw = xyzw; a.xw_ld = Nout;
    a.stats = stats_partial; a.xstats = xyz_stats;
    PCOPS_ROWS(a, rows);
    hipStream_t st = as_stream(stream);
    if (gpool) {
        PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale); PCOPS_REQUIRE_PTR(pool_shift);
        a.X = Y;  // alignment probe only
        return launch_gemm_ws_only<A_DYPOOL, E_MASKX>(a, st);
    }
    PCOPS_REQUIRE_PTR(G);
    return launch_gemm_ws_only<A_DY, E_MASKX>(a, st);
}


// workgroups (= partial copies) of the single-pass Gram kernel, 0 when it does not take the shape
static int gram_full_groups(long long M, int K, int ldx, const void *X) {
    if (!gram_full_on() || !ws_enabled() || M < 65536 || K > 320 || K < 32 || K % 4 != 0 || ldx % 4 != 0) return 0;
    if (reinterpret_cast<uintptr_t>(X) & 15) return 0;
    long long g = 256;
    const long long ns = (M + 31) / 32;
    if (g > ns) g = ns;
    if (g >= 8) g &= ~7ll;
    return (int)g;
}

int pcops_mlp_wgrad_splits(long long M, int K, int N) {
    // upper bound of the partial copies ANY of the three wgrad kernels writes for this shape (the scratch is sized
    // with it): the group counts of wgrad_pc_plan / wgrad_ws_plan before their M clamp, and the legacy split count.
    // (It used to be a flat 512, i.e. 1 GiB of scratch for the 512 -> 1024 layer whose kernels write 16 partials.)
    int best = wgrad_legacy_splits(M, K, N);
    {   // wgrad_pc_plan
        const int tk = K <= 64 ? 1 : 2, tn = N <= 64 ? 1 : (N <= 128 ? 2 : 4);
        const int kb = (K + 64 * tk - 1) / (64 * tk), nb = (N + 64 * tn - 1) / (64 * tn);
        int g = (tk * tn == 1 ? 512 : 256) / (kb * nb);
        if (g < 1) g = 1;
        if (g > best) best = g;
        if (K == N) {                 // pcops_mlp_gram of this width may take the single-pass kernel: a copy per workgroup
            const int gg = gram_full_groups(M, K, 4, nullptr);
            if (gg > best) best = gg;
        }
    }
    {   // wgrad_ws_plan
        int tk, tn;
        if (K <= 64 && N <= 64) { tk = 2; tn = 2; }
        else if (K <= 64) { tk = 2; tn = 4; }
        else { tk = 4; tn = 4; }
        const int kb = (K + 32 * tk - 1) / (32 * tk), nb = (N + 32 * tn - 1) / (32 * tn);
        int g = 256 / (kb * nb);
        if (g < 1) g = 1;
        if (g > best) best = g;
    }
    return best;
}



/* 1 when EVERY launch of a grouped stack on compacted rows has a kernel for its shape -- the *_rows entry points have no
 * tiled fallback, so a caller decides with this (and nothing else) whether to compact: group size a multiple of the
 * 16-row block and within the 8-bit arg index, the wave-stream forward / data-gradient and the producer / consumer
 * weight-gradient plans of every layer >= 1 (16-byte aligned operands assumed), the gather formulation of the first
 * layer's feature gradient when the stack has a feature term (c1 = widths[0], n source points per cloud). */
int pcops_gather_stack_rows_supported(int b, int n, int m, int s, int has_q, int nlayers, const int *widths) {
    if (b < 1 || m < 1 || s < kBlk || s % kBlk != 0 || s > 256 || nlayers < 2 || !widths) return 0;
    if (!ws_enabled() || !wgrad_pc_enabled()) return 0;
    const long long M = (long long)b * m * s;
    if (M > 0x7fffffffll) return 0;
    for (int l = 1; l < nlayers; ++l) {
        const int K = widths[l - 1], N = widths[l];
        GemmArgs f = {};                 // forward: (M, K) -> (M, N)
        f.M = (int)M; f.K = K; f.N = N; f.ldx = K; f.ldy = N;
        GemmArgs d = {};                 // data gradient: (M, N) -> (M, K)
        d.M = (int)M; d.K = N; d.N = K; d.ldx = N; d.ldy = K; d.S = s;
        WsPlan pl;
        PcWgradPlan pc;
        if (!ws_plan(f, A_BNRELU, &pl) || !ws_plan(d, A_DY, &pl) ||
            !wgrad_pc_plan(M, K, N, K, nullptr, nullptr, nullptr, nullptr, nullptr, &pc))
            return 0;
    }
    if (has_q && !pcops_sa_scatter_rows_supported(n, m, s, widths[0])) return 0;
    return 1;
}

static int bwd_fused_groups(long long M, int K, int N, int S, int pooled) {
    if (!ws_enabled() || !wgrad_pc_enabled()) return 0;
    static const bool on = [] {
        const char *e = getenv("PCOPS_BWD_FUSED");
        return !(e && e[0] == '0');
    }();
    if (!on) return 0;
    // the bandwidth-bound 64-wide layers only: wider ones are matrix-pipe bound in both kernels and gain nothing
    if (M < 65536 || M > 0x7fffff00ll || K > 64 || K % 4 != 0 || N > 128 || N % 4 != 0) return 0;
    if (pooled && (S < 1 || S > 255)) return 0;
    long long g = 256;
    const long long ns = (M + 31) / 32;
    if (g > ns) g = ns;
    if (g >= 8) g &= ~7ll;
    return (int)g;
}

int pcops_mlp_bwd_fused_groups(long long M, int K, int N, int S, int pooled) { return bwd_fused_groups(M, K, N, S, pooled); }

int pcops_mlp_bwd_fused(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                        const float *G, const float *Y, const float *p, const float *q, const float *t,
                        const float *gpool, const unsigned char *argmax, int S, const float *W, float *partial,
                        float *dW, float *db, float *Gprev, float *stats_partial, pcops_stream_t stream) {
    return pcops_mlp_bwd_fused_rows(M, K, N, Yprev, a_scale, a_shift, G, Y, p, q, t, gpool, argmax, S, W, partial, dW, db,
                                    Gprev, stats_partial, nullptr, stream);
}


int pcops_mlp_bwd_fused_rows(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                             const float *G, const float *Y, const float *p, const float *q, const float *t,
                             const float *gpool, const unsigned char *argmax, int S, const float *W, float *partial,
                             float *dW, float *db, float *Gprev, float *stats_partial, const pcops_rows_t *rows,
                             pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1);
    PCOPS_REQUIRE_PTR(Yprev); PCOPS_REQUIRE_PTR(a_scale); PCOPS_REQUIRE_PTR(a_shift); PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(W);
    PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW); PCOPS_REQUIRE_PTR(Gprev); PCOPS_REQUIRE_PTR(stats_partial);
    if (!gpool) PCOPS_REQUIRE_PTR(G);
    if (gpool) PCOPS_REQUIRE_PTR(argmax);
    const int groups = bwd_fused_groups(M, K, N, S, gpool != nullptr);
    if (groups == 0) return PCOPS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Yprev) & 15) || (reinterpret_cast<uintptr_t>(G) & 15) ||
        (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(gpool) & 15) ||
        (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(argmax) & 3))
        return PCOPS_ERR_UNSUPPORTED;
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = A_BNRELU; a.X = Yprev; a.ldx = K; a.asc = a_scale; a.ash = a_shift;
    a.dmode = gpool ? A_DYPOOL : A_DY; a.G = G; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.gpool = gpool; a.argmax = argmax; a.S = S > 0 ? S : 1;
    a.W = W; a.Gprev = Gprev; a.gstats = stats_partial;
    a.nt_out = nt_for_bytes((long long)M * K * 4);
    PCOPS_ROWS(a, rows);
    return bwd_fused_launch(a, false, groups, partial, dW, db, as_stream(stream));
}

int pcops_mlp_bwd_fused_edge(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                                  const float *Y, const float *p, const float *q, const float *t, const float *gpool,
                                  const unsigned char *argmax, int S, const float *W, float *partial, float *dW, float *db,
                                  float *stats_partial, const float *edge_rows, float *edge_stats, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && S >= 1);
    PCOPS_REQUIRE_PTR(Yprev); PCOPS_REQUIRE_PTR(a_scale); PCOPS_REQUIRE_PTR(a_shift); PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(gpool);
    PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW); PCOPS_REQUIRE_PTR(stats_partial);
    PCOPS_REQUIRE_PTR(edge_rows); PCOPS_REQUIRE_PTR(edge_stats);
    const int groups = bwd_fused_groups(M, K, N, S, true);
    if (groups == 0 || S % 32 == 0) return PCOPS_ERR_UNSUPPORTED;       // (whole-tile groups take another operand form)
    if ((reinterpret_cast<uintptr_t>(Yprev) & 15) || (reinterpret_cast<uintptr_t>(Y) & 15) ||
        (reinterpret_cast<uintptr_t>(gpool) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3) || (reinterpret_cast<uintptr_t>(edge_rows) & 15))
        return PCOPS_ERR_UNSUPPORTED;
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = A_BNRELU; a.X = Yprev; a.ldx = K; a.asc = a_scale; a.ash = a_shift;
    a.dmode = A_DYPOOL; a.G = nullptr; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.gpool = gpool; a.argmax = argmax; a.S = S;
    a.W = W; a.Gprev = nullptr; a.gstats = stats_partial; a.xstats = edge_stats; a.side = edge_rows;
    return bwd_fused_launch(a, false, groups, partial, dW, db, as_stream(stream), true);
}

/* ---- the one-pass backward of a POOLED layer with its weight gradient in the Gram form (round 6; bwd_fused_kernel<.., GW>):
 * uncompacted rows, groups of S % 32 == 0 or 11 <= S <= 255 rows, split-operand dX half.  bias [N] (may be NULL) is the
 * layer's own bias -- Y = X W + bias is what the form substitutes.  partial: groups (K N + N + K K + K) floats. */
static int bwd_fused_gw_groups(long long M, int K, int N, int S) {
    if (pcops_get_option(PCOPS_OPT_BWD_FUSED_GRAM_WGRAD) == 0 || S < 11 || S > 255) return 0;
    if (pcops_get_option(PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16) == 0) return 0;
    return bwd_fused_groups(M, K, N, S, 1);
}

int pcops_mlp_bwd_fused_gw_groups(long long M, int K, int N, int S) { return bwd_fused_gw_groups(M, K, N, S); }

static void gw_carve(WgradArgs &a, float *partial, int groups, int K, int N) {
    a.gram_part = partial + (long long)groups * ((long long)K * N + N);
    a.xsum_part = a.gram_part + (long long)groups * K * K;
}

int pcops_mlp_bwd_fused_gw(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                           const float *Y, const float *p, const float *q, const float *t, const float *gpool,
                           const unsigned char *argmax, int S, const float *W, const float *bias, float *partial, float *dW,
                           float *db, float *Gprev, float *stats_partial, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && S >= 1);
    PCOPS_REQUIRE_PTR(Yprev); PCOPS_REQUIRE_PTR(a_scale); PCOPS_REQUIRE_PTR(a_shift); PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(gpool);
    PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW); PCOPS_REQUIRE_PTR(db);
    PCOPS_REQUIRE_PTR(Gprev); PCOPS_REQUIRE_PTR(stats_partial);
    const int groups = bwd_fused_gw_groups(M, K, N, S);
    if (groups == 0) return PCOPS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Yprev) & 15) || (reinterpret_cast<uintptr_t>(Y) & 15) ||
        (reinterpret_cast<uintptr_t>(gpool) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3))
        return PCOPS_ERR_UNSUPPORTED;
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = A_BNRELU; a.X = Yprev; a.ldx = K; a.asc = a_scale; a.ash = a_shift;
    a.dmode = A_DYPOOL; a.G = nullptr; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.gpool = gpool; a.argmax = argmax; a.S = S;
    a.W = W; a.Gprev = Gprev; a.gstats = stats_partial;
    a.nt_out = nt_for_bytes((long long)M * K * 4);
    gw_carve(a, partial, groups, K, N);
    return bwd_fused_launch(a, false, groups, partial, dW, db, as_stream(stream), false, bias);
}

int pcops_mlp_bwd_fused_edge_gw(long long M, int K, int N, const float *Yprev, const float *a_scale, const float *a_shift,
                                const float *Y, const float *p, const float *q, const float *t, const float *gpool,
                                const unsigned char *argmax, int S, const float *W, const float *bias, float *partial,
                                float *dW, float *db, float *stats_partial, const float *edge_rows, float *edge_stats,
                                pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && S >= 1);
    PCOPS_REQUIRE_PTR(Yprev); PCOPS_REQUIRE_PTR(a_scale); PCOPS_REQUIRE_PTR(a_shift); PCOPS_REQUIRE_PTR(Y);
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(gpool);
    PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW); PCOPS_REQUIRE_PTR(db);
    PCOPS_REQUIRE_PTR(stats_partial); PCOPS_REQUIRE_PTR(edge_rows); PCOPS_REQUIRE_PTR(edge_stats);
    const int groups = bwd_fused_gw_groups(M, K, N, S);
    if (groups == 0 || S % 32 == 0) return PCOPS_ERR_UNSUPPORTED;       // (whole-tile groups take another operand form)
    if ((reinterpret_cast<uintptr_t>(Yprev) & 15) || (reinterpret_cast<uintptr_t>(Y) & 15) ||
        (reinterpret_cast<uintptr_t>(gpool) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3) || (reinterpret_cast<uintptr_t>(edge_rows) & 15))
        return PCOPS_ERR_UNSUPPORTED;
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = A_BNRELU; a.X = Yprev; a.ldx = K; a.asc = a_scale; a.ash = a_shift;
    a.dmode = A_DYPOOL; a.G = nullptr; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.gpool = gpool; a.argmax = argmax; a.S = S;
    a.W = W; a.Gprev = nullptr; a.gstats = stats_partial; a.xstats = edge_stats; a.side = edge_rows;
    gw_carve(a, partial, groups, K, N);
    return bwd_fused_launch(a, false, groups, partial, dW, db, as_stream(stream), true, bias);
}

int pcops_mlp_bwd_fused_xyz_rows(long long M, int K, int N, const float *off4, const float *xyzw, const float *a_scale,
                                 const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                                 const float *t, const float *gpool, const unsigned char *argmax, int S, const float *W,
                                 float *partial, float *dW, float *db, float *stats_partial, float *xyz_stats,
                                 const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 4 && K % 4 == 0 && N >= 1);
    PCOPS_REQUIRE_PTR(off4); PCOPS_REQUIRE_PTR(xyzw); PCOPS_REQUIRE_PTR(a_scale); PCOPS_REQUIRE_PTR(a_shift);
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(W);
    PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW); PCOPS_REQUIRE_PTR(stats_partial); PCOPS_REQUIRE_PTR(xyz_stats);
    if (!gpool) PCOPS_REQUIRE_PTR(G);
    if (gpool) PCOPS_REQUIRE_PTR(argmax);
    const int groups = bwd_fused_groups(M, K, N, S, gpool != nullptr);
    if (groups == 0) return PCOPS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(off4) & 15) || (reinterpret_cast<uintptr_t>(G) & 15) ||
        (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(gpool) & 15) ||
        (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(argmax) & 3))
        return PCOPS_ERR_UNSUPPORTED;
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = A_XYZ; a.X = off4; a.ldx = 4; a.off4 = off4; a.xw = xyzw; a.xw_ld = K; a.asc = a_scale; a.ash = a_shift;
    a.dmode = gpool ? A_DYPOOL : A_DY; a.G = G; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.gpool = gpool; a.argmax = argmax; a.S = S > 0 ? S : 1;
    a.W = W; a.Gprev = nullptr; a.gstats = stats_partial; a.xstats = xyz_stats;
    PCOPS_ROWS(a, rows);
    return bwd_fused_launch(a, true, groups, partial, dW, db, as_stream(stream));
}

int pcops_mlp_wgrad(long long M, int K, int N, const float *X, int ldx, const float *a_scale,
                    const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                    const float *t, const float *gpool, const unsigned char *argmax, int S,
                    const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                    pcops_stream_t stream) {
    return pcops_mlp_wgrad_rows(M, K, N, X, ldx, a_scale, a_shift, G, Y, p, q, t, gpool, argmax, S, pool_scale,
                                pool_shift, partial, dW, db, nullptr, stream);
}

int pcops_mlp_wgrad_rows(long long M, int K, int N, const float *X, int ldx, const float *a_scale,
                         const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                         const float *t, const float *gpool, const unsigned char *argmax, int S,
                         const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                         const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && ldx >= K);
    PCOPS_REQUIRE_PTR(X); PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW);
    if (!gpool) PCOPS_REQUIRE_PTR(G);
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = a_scale ? A_BNRELU : A_PLAIN; a.X = X; a.ldx = ldx; a.asc = a_scale; a.ash = a_shift;
    a.dmode = gpool ? A_DYPOOL : A_DY; a.G = G; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.dsc = pool_scale; a.dsh = pool_shift; a.gpool = gpool; a.argmax = argmax; a.S = S > 0 ? S : 1;
    PCOPS_ROWS(a, rows);
    return wgrad_impl(a, partial, dW, db, as_stream(stream));
}

/* A = relu(a_scale * y + a_shift) with y the ARITHMETIC first layer rebuilt from off4 [M][4] and xyzw [4][K]
 * (rows w0, w1, w2, b): the grouped first-layer tensor is never read.  Wave-stream shapes only. */
int pcops_mlp_wgrad_xyz(long long M, int K, int N, const float *off4, const float *xyzw, const float *a_scale,
                        const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                        const float *t, const float *gpool, const unsigned char *argmax, int S,
                        const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                        pcops_stream_t stream) {
    return pcops_mlp_wgrad_xyz_rows(M, K, N, off4, xyzw, a_scale, a_shift, G, Y, p, q, t, gpool, argmax, S, pool_scale,
                                    pool_shift, partial, dW, db, nullptr, stream);
}

int pcops_mlp_wgrad_xyz_rows(long long M, int K, int N, const float *off4, const float *xyzw, const float *a_scale,
                             const float *a_shift, const float *G, const float *Y, const float *p, const float *q,
                             const float *t, const float *gpool, const unsigned char *argmax, int S,
                             const float *pool_scale, const float *pool_shift, float *partial, float *dW, float *db,
                             const pcops_rows_t *rows, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1);
    PCOPS_REQUIRE_PTR(off4); PCOPS_REQUIRE_PTR(xyzw); PCOPS_REQUIRE_PTR(a_scale); PCOPS_REQUIRE_PTR(a_shift);
    PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(dW);
    if (!gpool) PCOPS_REQUIRE_PTR(G);
    if (reinterpret_cast<uintptr_t>(off4) & 15) return PCOPS_ERR_UNSUPPORTED;
    WgradArgs a = {};
    a.M = M; a.K = K; a.N = N;
    a.amode = A_XYZ; a.X = nullptr; a.ldx = K; a.asc = a_scale; a.ash = a_shift;
    a.off4 = off4; a.xw = xyzw; a.xw_ld = K;
    a.dmode = gpool ? A_DYPOOL : A_DY; a.G = G; a.Y = Y; a.ldy = N; a.p = p; a.q = q; a.t = t;
    a.dsc = pool_scale; a.dsh = pool_shift; a.gpool = gpool; a.argmax = argmax; a.S = S > 0 ? S : 1;
    PCOPS_ROWS(a, rows);
    return wgrad_impl(a, partial, dW, db, as_stream(stream));
}

/* ---- algebraic backward of a pooled top layer (see pool_top_addend_kernel) */
static bool dgrad_top_plan(int M, int Kp, WsPlan *pl) {
    GemmArgs a = {};
    a.M = M; a.K = Kp; a.N = Kp; a.ldx = Kp; a.ldy = Kp;
    return ws_enabled() && Kp % 64 == 0 && Kp <= 512 && ws_plan(a, A_BNRELU, pl);
}

int pcops_mlp_pool_top_supported(int M, int Kp, int N, int S) {
    WsPlan pl;
    PcWgradPlan pc;
    if (S < 1 || S > 256 || M % S != 0 || N % 4 != 0 || N > 1024) return 0;
    if (!dgrad_top_plan(M, Kp, &pl)) return 0;
    if (!(wgrad_pc_enabled() && wgrad_pc_plan(M, Kp, Kp, Kp, nullptr, nullptr, nullptr, nullptr, nullptr, &pc, true))) return 0;
    const long long slots = (long long)(M / S) * (S < N ? S : N);
    return slots * Kp * 4 < (long long)kOOB ? 1 : 0;
}

/* compact arg-max rows of the data gradient: addend [(M/S) min(S,N)][Kp], rowmap [M] (slot or -1) */
int pcops_mlp_pool_top_addend(int M, int Kp, int N, int S, const float *gout, const float *ysel,
                            const unsigned char *argmax, const float *pool_scale, const float *pool_shift,
                            const float *p, const float *Wt, float *addend, int *rowmap, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && S >= 1 && S <= 256 && M % S == 0 && Kp % 4 == 0 && N >= 1 && N <= 65535);
    PCOPS_REQUIRE_PTR(gout); PCOPS_REQUIRE_PTR(ysel); PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale);
    PCOPS_REQUIRE_PTR(pool_shift); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(Wt); PCOPS_REQUIRE_PTR(addend);
    PCOPS_REQUIRE_PTR(rowmap);
    if (reinterpret_cast<uintptr_t>(Wt) & 15 || reinterpret_cast<uintptr_t>(addend) & 15) return PCOPS_ERR_UNSUPPORTED;
    if (N > 1024 || Kp > 512) return PCOPS_ERR_UNSUPPORTED;
    if (M / S <= 512)
        hipLaunchKernelGGL(pool_top_addend_kernel<1024>, dim3(M / S), dim3(1024), 0, as_stream(stream), S, N, Kp,
                           S < N ? S : N, gout, ysel, argmax, pool_scale, pool_shift, p, Wt, addend, rowmap);
    else
        hipLaunchKernelGGL(pool_top_addend_kernel<256>, dim3(M / S), dim3(256), 0, as_stream(stream), S, N, Kp,
                           S < N ? S : N, gout, ysel, argmax, pool_scale, pool_shift, p, Wt, addend, rowmap);
    return pcops_launch_status();
}

/* Gprev = mask . (relu(bn(Yprev)) Mq + addend[rowmap] + vconst), statistics as pcops_mlp_gemm_dgrad */
int pcops_mlp_gemm_dgrad_top(int M, int Kp, const float *Yprev, const float *prev_scale, const float *prev_shift,
                             const float *Mq, const float *vconst, const float *addend, long long addend_rows,
                             const int *rowmap, float *Gprev, float *stats_partial, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && Kp >= 1);
    PCOPS_REQUIRE_PTR(Yprev); PCOPS_REQUIRE_PTR(Mq);
    PCOPS_REQUIRE_PTR(vconst); PCOPS_REQUIRE_PTR(addend); PCOPS_REQUIRE_PTR(rowmap); PCOPS_REQUIRE_PTR(Gprev);
    PCOPS_REQUIRE_ARG((prev_scale == nullptr) == (prev_shift == nullptr));
    WsPlan pl;
    if (!dgrad_top_plan(M, Kp, &pl)) return PCOPS_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.M = M; a.K = Kp; a.N = Kp; a.X = Yprev; a.ldx = Kp; a.v0 = prev_scale; a.v1 = prev_shift;
    a.W = Mq; a.Y = Gprev; a.ldy = Kp; a.Yprev = Yprev; a.msc = prev_scale; a.msh = prev_shift;
    a.stats = stats_partial; a.addend = addend; a.rowmap = rowmap; a.add_ld = Kp; a.vconst = vconst;
    a.add_bytes = addend_rows * Kp * 4;
    if (addend_rows < 1 || a.add_bytes >= (long long)kOOB) return PCOPS_ERR_UNSUPPORTED;
    if (!prev_scale) {      // X is the stack's raw input: plain dX, no mask, no statistics
        a.stats = nullptr;
        return launch_gemm_ws_only<A_PLAIN, E_PLAINA>(a, as_stream(stream));
    }
    return launch_gemm_ws_only<A_BNRELU, E_MASKA>(a, as_stream(stream));
}

/* gram [Kp][Kp] = X^T X, xsum [Kp] = X^T 1 with X = relu(Yprev * a_scale + a_shift);  partial as pcops_mlp_wgrad
 * (pcops_mlp_wgrad_splits(M, Kp, Kp) copies) */
int pcops_mlp_gram(long long M, int Kp, const float *Yprev, int ldx, const float *a_scale, const float *a_shift,
                   float *partial, float *gram, float *xsum, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && Kp >= 1 && ldx >= Kp);
    PCOPS_REQUIRE_PTR(Yprev); PCOPS_REQUIRE_PTR(partial); PCOPS_REQUIRE_PTR(gram);
    PCOPS_REQUIRE_ARG((a_scale == nullptr) == (a_shift == nullptr));
    const int gg = gram_full_groups(M, Kp, ldx, Yprev);
    pcops_note_pipe(0);
    if (gg > 0) {
        // one pass over X: whole rows staged, every upper 32 x 32 block in accumulators (gram_full_kernel)
        hipStream_t st = as_stream(stream);
        GramArgs g = {M, Kp, ldx, Yprev, a_scale, a_shift, partial, partial + (long long)gg * Kp * Kp};
        const int nbk = (Kp + 31) / 32;
        const size_t lds = (size_t)(2 * 32 * nbk + 2 * 32 * (32 * nbk + 4)) * sizeof(float);
        {
            const int rcl = gram_full_launch(g, nbk, a_scale != nullptr, gg, lds, st);
            if (rcl) return rcl;
        }
        int rc = pcops_launch_status();
        if (rc) return rc;
        const long long L = (long long)Kp * Kp;
        hipLaunchKernelGGL(sum_partials2_kernel, dim3(cdiv(L, 64) + (xsum ? cdiv(Kp, 64) : 0)), dim3(1024), 0, st, gg, L,
                           partial, gram, (long long)Kp, g.xpart, xsum);
        hipLaunchKernelGGL(mirror_lower_kernel, dim3(Kp), dim3(256), 0, st, Kp, gram);
        return pcops_launch_status();
    }
    WgradArgs a = {};
    a.M = M; a.K = Kp; a.N = Kp;
    a.amode = a_scale ? A_BNRELU : A_PLAIN; a.X = Yprev; a.ldx = ldx; a.asc = a_scale; a.ash = a_shift;
    a.dmode = A_SELFD; a.G = Yprev; a.Y = Yprev; a.ldy = ldx; a.p = a_scale; a.q = a_scale; a.t = a_shift; a.S = 1;
    return wgrad_impl(a, partial, gram, xsum, as_stream(stream));
}

/* Ssp [Kp][N] = X^T (p.G) (the arg-max rows only), cfsum [N] = column sums of p.G */
int pcops_mlp_pool_top_wsparse(int M, int Kp, int N, int S, const float *gout, const float *ysel,
                               const unsigned char *argmax, const float *pool_scale, const float *pool_shift,
                               const float *p, const float *Yprev, const float *prev_scale, const float *prev_shift,
                               float *Ssp, float *cfsum, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && S >= 1 && S <= 256 && M % S == 0 && Kp >= 1 && Kp <= 512 && N >= 1);
    PCOPS_REQUIRE_PTR(gout); PCOPS_REQUIRE_PTR(ysel); PCOPS_REQUIRE_PTR(argmax); PCOPS_REQUIRE_PTR(pool_scale);
    PCOPS_REQUIRE_PTR(pool_shift); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(Yprev);
    PCOPS_REQUIRE_ARG((prev_scale == nullptr) == (prev_shift == nullptr));
    PCOPS_REQUIRE_PTR(Ssp); PCOPS_REQUIRE_PTR(cfsum);
    hipLaunchKernelGGL(pool_top_wsparse_kernel, dim3(N), dim3(256), 0, as_stream(stream), (long long)(M / S), S, N,
                       Kp, gout, ysel, argmax, pool_scale, pool_shift, p, Yprev, prev_scale, prev_shift, Ssp, cfsum);
    return pcops_launch_status();
}

int pcops_mlp_dy_apply(long long M, int N, const float *G, const float *Y, const float *p, const float *q, const float *t,
                       float *out, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 0 && N >= 4 && N % 4 == 0);
    if (M == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(G); PCOPS_REQUIRE_PTR(Y); PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(out);
    PCOPS_REQUIRE_ARG(((uintptr_t)G | (uintptr_t)Y | (uintptr_t)p | (uintptr_t)q | (uintptr_t)t | (uintptr_t)out) % 16 == 0);
    const long long total4 = M * (N / 4);
    const unsigned grid = (unsigned)(cdiv(total4, 256) < 8192 ? cdiv(total4, 256) : 8192);
    hipLaunchKernelGGL(dy_apply_kernel, dim3(grid), dim3(256), 0, as_stream(stream), total4, N / 4,
                       reinterpret_cast<const float4 *>(G), reinterpret_cast<const float4 *>(Y),
                       reinterpret_cast<const float4 *>(p), reinterpret_cast<const float4 *>(q),
                       reinterpret_cast<const float4 *>(t), reinterpret_cast<float4 *>(out));
    return pcops_launch_status();
}

int pcops_mlp_pool_top_prep(int Kp, int N, const float *W, const float *b, const float *q, const float *t, float *Wt,
                            float *Wq, float *u, float *v, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(Kp >= 1 && N >= 1);
    PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(b); PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(t);
    PCOPS_REQUIRE_PTR(Wt); PCOPS_REQUIRE_PTR(Wq); PCOPS_REQUIRE_PTR(u);
    const int gx = (N + 31) / 32;
    hipLaunchKernelGGL(pool_top_prep_kernel, dim3(gx, (Kp + 31) / 32 + (v ? ((Kp + 7) / 8 + gx - 1) / gx : 0)), dim3(256), 0,
                       as_stream(stream), Kp, N, W, b, q, t, Wt, Wq, u, v);
    return pcops_launch_status();
}

int pcops_mlp_pool_top_finish(int Kp, int N, long long M, float *dW, const float *Ssp, const float *xsum, const float *u,
                              const float *cfsum, const float *q, const float *W, const float *b, const float *t,
                              float *db, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(Kp >= 1 && N >= 1 && M >= 1);
    PCOPS_REQUIRE_PTR(dW); PCOPS_REQUIRE_PTR(Ssp); PCOPS_REQUIRE_PTR(xsum); PCOPS_REQUIRE_PTR(u); PCOPS_REQUIRE_PTR(cfsum);
    PCOPS_REQUIRE_PTR(q); PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(b); PCOPS_REQUIRE_PTR(t); PCOPS_REQUIRE_PTR(db);
    hipLaunchKernelGGL(pool_top_finish_kernel, dim3((N + 255) / 256, 8 + (Kp + 7) / 8), dim3(256), 0, as_stream(stream), Kp, N,
                       (float)M, dW, Ssp, xsum, u, cfsum, q, W, b, t, db);
    return pcops_launch_status();
}

int pcops_small_gemm(int M, int K, int N, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                     pcops_stream_t stream) {
    return pcops_small_gemm_ex(M, K, N, A, lda, 0, B, ldb, 0, nullptr, C, ldc, stream);
}

int pcops_small_gemm_ex(int M, int K, int N, const float *A, int lda, int transA, const float *B, int ldb, int transB,
                        const float *bias, float *C, int ldc, pcops_stream_t stream) {
    return pcops_small_gemm_colsum(M, K, N, A, lda, transA, B, ldb, transB, bias, C, ldc, nullptr, stream);
}

int pcops_small_gemm_colsum(int M, int K, int N, const float *A, int lda, int transA, const float *B, int ldb, int transB,
                            const float *bias, float *C, int ldc, float *colsum, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(M >= 1 && K >= 1 && N >= 1 && lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    PCOPS_REQUIRE_PTR(A); PCOPS_REQUIRE_PTR(B); PCOPS_REQUIRE_PTR(C);
    pcops_note_pipe(0);                                     // fp32 MFMA (bench labels read pcops_last_launch_pipe per launch)
    hipLaunchKernelGGL(small_gemm_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, as_stream(stream), M, K, N, A,
                       lda, B, ldb, C, ldc, transA ? 1 : 0, transB ? 1 : 0, bias, colsum);
    return pcops_launch_status();
}

int pcops_small_gemm_pair(const pcops_gemm_problem_t *p, pcops_stream_t stream) {
    PCOPS_REQUIRE_PTR(p);
    SgProblem q[2];
    int tiles[2], gx[2];
    for (int i = 0; i < 2; ++i) {
        const pcops_gemm_problem_t &a = p[i];
        PCOPS_REQUIRE_SHAPE(a.M >= 1 && a.K >= 1 && a.N >= 1 && a.lda >= (a.transA ? a.M : a.K) && a.ldb >= (a.transB ? a.K : a.N) &&
                            a.ldc >= a.N);
        PCOPS_REQUIRE_PTR(a.A); PCOPS_REQUIRE_PTR(a.B); PCOPS_REQUIRE_PTR(a.C);
        q[i] = SgProblem{a.M, a.K, a.N, a.lda, a.ldb, a.ldc, a.transA ? 1 : 0, a.transB ? 1 : 0, a.A, a.B, a.bias, a.C, a.colsum};
        gx[i] = (a.N + 31) / 32;
        tiles[i] = gx[i] * ((a.M + 31) / 32);
    }
    pcops_note_pipe(0);
    hipLaunchKernelGGL(small_gemm_pair_kernel, dim3(tiles[0] + tiles[1]), dim3(256), 0, as_stream(stream), q[0], q[1], tiles[0],
                       gx[0], gx[1]);
    return pcops_launch_status();
}

#ifdef PCOPS_PHASE_PROF
/* tools only: out[0..5] = summed shader cycles (stage, MFMA loop, epilogue, whole kernel), tiles, waves; then cleared */
int pcops_debug_phase_prof(unsigned long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return PCOPS_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_prof), sizeof(unsigned long long) * 8) != hipSuccess) return PCOPS_ERR_LAUNCH;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_prof), z, sizeof(z)) != hipSuccess) return PCOPS_ERR_LAUNCH;
    return PCOPS_OK;
}
#endif

int pcops_mlp_transpose(int K, int N, const float *W, float *Wt, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(K >= 1 && N >= 1);
    PCOPS_REQUIRE_PTR(W); PCOPS_REQUIRE_PTR(Wt);
    hipLaunchKernelGGL(transpose_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, as_stream(stream), K, N,
                       W, Wt);
    return pcops_launch_status();
}

}  // extern "C"
#endif  // PCOPS_PART(0)
