// abi.hip -- status strings, ABI version and the process-wide switches of libpcops.
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "common.h"

extern "C" const char *pcops_strerror(int status) {
    switch (status) {
        case PCOPS_OK: return "ok";
        case PCOPS_ERR_NULL_POINTER: return "null pointer argument";
        case PCOPS_ERR_BAD_SHAPE: return "invalid tensor extents";
        case PCOPS_ERR_BAD_ARGUMENT: return "invalid attribute (radius/nsample/npoint/k)";
        case PCOPS_ERR_UNSUPPORTED: return "shape outside the range the gfx950 kernels are built for";
        case PCOPS_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown pcops status";
    }
}

extern "C" int pcops_abi_version(void) { return 4; }   // 2: stat_pivot (shifted BN moments) on the forward-statistics producers
                                                        // 3: one-pass backward (pcops_mlp_bwd_fused*), pcops_adam_step
                                                        // 4: pcops_edge_pool_fwd / pcops_sa_gather_fwd write pcops_*_fwd_stats_rows(shape)
                                                        //    rows of partial statistics; the shape-less queries are upper bounds only

// bit-reproducible backward passes (reference hazard tf_grouping_g.cu:61-78, tf_sampling_g.cu:183-192:
// float atomics).  Off: scatter-adds may use atomics / unordered lists.  On: every sum is taken by one owner in
// ascending row order; entry points that only have an atomic form return PCOPS_ERR_UNSUPPORTED.
static std::atomic<int> g_deterministic{[] {
    const char *e = getenv("PCOPS_DETERMINISTIC");
    return (e && e[0] == '1') ? 1 : 0;
}()};
extern "C" void pcops_set_deterministic(int on) { g_deterministic.store(on ? 1 : 0); }

// ---- arithmetic options (pcops.h pcops_set_option).  The ONLY process-wide mutable state besides the deterministic
// switch: which of two exact-to-fp32 formulations a product runs in.  Rounds 3-4 read these from the environment once per
// process inside the launchers (a TF-side caller could not choose per call); now the environment
// only provides the initial value (test override) and the launchers read the table at every call.
namespace {
int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
std::atomic<int> g_options[PCOPS_OPT_COUNT] = {};
std::atomic<int> g_options_init{0};
void options_init() {
    if (g_options_init.load(std::memory_order_acquire)) return;
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    while (busy.test_and_set(std::memory_order_acquire)) {}
    if (!g_options_init.load(std::memory_order_relaxed)) {
        g_options[PCOPS_OPT_GEMM_SPLIT_BF16].store(env_int("PCOPS_GEMM_BF3", 1));
        g_options[PCOPS_OPT_WGRAD_SPLIT_BF16].store(env_int("PCOPS_WGRAD_BF3", 1) != 0);
        g_options[PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16].store(env_int("PCOPS_BWD_FUSED_DX3", 2));
        g_options[PCOPS_OPT_KNN_F16_PREFILTER].store(env_int("PCOPS_KNN_F16", 1) != 0);
        g_options[PCOPS_OPT_DGRAD_SPLIT_BF16].store(env_int("PCOPS_DGRAD_BF3", 1));
        g_options[PCOPS_OPT_BWD_FUSED_GRAM_WGRAD].store(env_int("PCOPS_BWD_FUSED_GW", 0) != 0);
        g_options_init.store(1, std::memory_order_release);
    }
    busy.clear(std::memory_order_release);
}
}  // namespace

extern "C" int pcops_get_option(int option) {
    if (option <= 0 || option >= PCOPS_OPT_COUNT) return PCOPS_ERR_BAD_ARGUMENT;
    options_init();
    return g_options[option].load();
}

extern "C" int pcops_set_option(int option, int value) {
    if (option <= 0 || option >= PCOPS_OPT_COUNT || value < 0) return PCOPS_ERR_BAD_ARGUMENT;
    if ((option == PCOPS_OPT_GEMM_SPLIT_BF16 || option == PCOPS_OPT_DGRAD_SPLIT_BF16 ||
         option == PCOPS_OPT_BWD_FUSED_DX_SPLIT_BF16) ? value > 2 : value > 1)
        return PCOPS_ERR_BAD_ARGUMENT;
    options_init();
    return g_options[option].exchange(value);
}
extern "C" int pcops_get_deterministic(void) { return g_deterministic.load(); }

// diagnostics (bench.py): which matrix pipe the LAST matrix-product launch of the calling thread took -- 0 the fp32 pipe
// (or no product), 1 the bf16 pipe with split operands, 2 half and half (one-pass backward: dW fp32, dX split; 1 where dW is split too).  The
// launchers note it where they decide, so a roofline label is the library's own decision, not a mirror of its rules.
static thread_local int t_last_pipe = 0;
void pcops_note_pipe(int pipe) { t_last_pipe = pipe; }
extern "C" int pcops_last_launch_pipe(void) { return t_last_pipe; }

// ---------------------------------------------------------------------------------------------------------------
// The training step's parameter update as ONE launch over the flat buffers (host: train_util.TFAdam).  TensorFlow's Adam
// (`tf.train.AdamOptimizer`, reference trainers `pointnet2/train.py:165-168`): m <- b1 m + (1 - b1) g,
// v <- b2 v + (1 - b2) g^2, p <- p - lr_t m / (sqrt(v) + eps) with lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) from the host
// (epsilon OUTSIDE the bias-corrected root).  It replaced seven elementwise launches per step.
namespace {
__global__ __launch_bounds__(256) void adam_step_kernel(long long n4, float4 *__restrict__ p, const float4 *__restrict__ g,
                                                        float4 *__restrict__ m, float4 *__restrict__ v, float b1, float b2,
                                                        float lr_t, float eps) {
    const float c1 = 1.f - b1, c2 = 1.f - b2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 gi = g[i];
        float4 mi = m[i], vi = v[i], pi = p[i];
        mi.x = fmaf(c1, gi.x, b1 * mi.x); mi.y = fmaf(c1, gi.y, b1 * mi.y);
        mi.z = fmaf(c1, gi.z, b1 * mi.z); mi.w = fmaf(c1, gi.w, b1 * mi.w);
        vi.x = fmaf(c2 * gi.x, gi.x, b2 * vi.x); vi.y = fmaf(c2 * gi.y, gi.y, b2 * vi.y);
        vi.z = fmaf(c2 * gi.z, gi.z, b2 * vi.z); vi.w = fmaf(c2 * gi.w, gi.w, b2 * vi.w);
        pi.x -= lr_t * mi.x / (sqrtf(vi.x) + eps); pi.y -= lr_t * mi.y / (sqrtf(vi.y) + eps);
        pi.z -= lr_t * mi.z / (sqrtf(vi.z) + eps); pi.w -= lr_t * mi.w / (sqrtf(vi.w) + eps);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}
}  // namespace

extern "C" int pcops_adam_step(long long n, float *p, const float *g, float *m, float *v, float beta1, float beta2,
                               float lr_t, float epsilon, pcops_stream_t stream) {
    PCOPS_REQUIRE_SHAPE(n >= 0 && n % 4 == 0);
    if (n == 0) return PCOPS_OK;
    PCOPS_REQUIRE_PTR(p); PCOPS_REQUIRE_PTR(g); PCOPS_REQUIRE_PTR(m); PCOPS_REQUIRE_PTR(v);
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return PCOPS_ERR_UNSUPPORTED;
    const long long n4 = n / 4;
    unsigned grid = cdiv(n4, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, as_stream(stream), n4, reinterpret_cast<float4 *>(p),
                       reinterpret_cast<const float4 *>(g), reinterpret_cast<float4 *>(m), reinterpret_cast<float4 *>(v),
                       beta1, beta2, lr_t, epsilon);
    return pcops_launch_status();
}
