// api.hip -- the extern "C" entry points of libciao_hip.so (declared in include/ciao_hip.h).
//
// Each entry point validates its arguments on the host (shapes must match what the kernels and their grids assume
// BEFORE anything is launched), builds the kernel argument structs and enqueues the kernels on the ctx's stream.
// Nothing here computes on the CPU: if the HIP runtime or the device is missing the calls fail with CIAO_ERR_HIP.

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "ciao_ctx.h"
#include "launch.h"

namespace ciao {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int32_t hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return CIAO_ERR_HIP;
}

int32_t ensure(ciao_ctx *ctx, void **buf, size_t *have, size_t need)
{
    if (*have >= need) return CIAO_OK;
    // growing a workspace buffer: wait for work that may still read the old one, then reallocate
    CIAO_HIP(hipStreamSynchronize(ctx->stream));
    if (*buf) CIAO_HIP(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    size_t cap = need + need / 4 + 256;
    hipError_t e = hipMalloc(buf, cap);
    if (e != hipSuccess) {
        set_error("workspace allocation of %zu bytes failed: %s", cap, hipGetErrorString(e));
        return CIAO_ERR_ALLOC;
    }
    *have = cap;
    return CIAO_OK;
}

// ---- synthetic data -------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// standard normal keyed by (seed, row, col): Box-Muller on two 32-bit uniforms of one 64-bit hash
__device__ __forceinline__ float normal_at(uint64_t seed, uint64_t row, uint64_t col)
{
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ULL * (row * 0x100000001B3ULL + col + 1));
    h = mix64(h ^ (row << 1) ^ 0xD1B54A32D192ED03ULL);
    const float u1 = ((float)(uint32_t)(h >> 32) + 1.0f) * (1.0f / 4294967296.0f);   // (0,1]
    const float u2 = (float)(uint32_t)h * (1.0f / 4294967296.0f);                    // [0,1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

template <typename T>
__global__ void __launch_bounds__(256)
    synth_normal_kernel(T *out, int64_t nrows, int64_t d, int64_t ld, int64_t row0, uint64_t seed, T scale)
{
    const int64_t total = nrows * d;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t r = t / d, c = t - r * d;
        out[r * ld + c] = scale * (T)normal_at(seed, (uint64_t)(row0 + r), (uint64_t)c);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) synth_targets_kernel(const T *A, int64_t N, int64_t d, int64_t ld, const T *x_true,
                                                           T noise, int labels, int64_t row0, uint64_t seed, T *b)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += nw) {
        const T *ap = A + i * ld;
        T dot = T(0);
        for (int64_t e = lane; e < d; e += WAVE) dot += ap[e] * x_true[e];
        dot = wave_allsum(dot);
        T v = dot + noise * (T)normal_at(seed ^ 0xABCDEF0123456789ULL, (uint64_t)(row0 + i), 0xFFFFFFFFULL);
        if (labels) v = v >= T(0) ? T(1) : T(-1);
        if (lane == 0) b[i] = v;
    }
}

// ---- helpers ----------------------------------------------------------------------------------------------------------
static int32_t check_problem(ciao_ctx *ctx, const ciao_problem *p)
{
    CIAO_REQUIRE(ctx, "ctx is NULL");
    // Any entry point may overwrite the vector the cached a_i'z_full belong to; only ciao_svrg_iterate (which restores
    // the key after this check) and ciao_svrg_inner (which never writes z_full) keep the cache alive.
    ctx->rowdot_A = nullptr;
    CIAO_REQUIRE(p, "problem is NULL");
    CIAO_REQUIRE(p->dtype == CIAO_F32 || p->dtype == CIAO_F64, "problem.dtype must be CIAO_F32 or CIAO_F64 (got %d)", p->dtype);
    CIAO_REQUIRE(p->loss >= CIAO_LOSS_LS && p->loss <= CIAO_LOSS_LS_COMPLEX, "unknown loss kind %d", p->loss);
    CIAO_REQUIRE(p->loss != CIAO_LOSS_LS_COMPLEX || (p->d % 2 == 0 && p->ld % 2 == 0),
                 "complex problems store (re, im) pairs: d and ld count reals and must be even (got d=%lld ld=%lld)", (long long)p->d,
                 (long long)p->ld);
    CIAO_REQUIRE(p->N >= 0 && p->d >= 1, "need N >= 0 and d >= 1 (got N=%lld d=%lld)", (long long)p->N, (long long)p->d);
    CIAO_REQUIRE(p->ld >= p->d, "row stride ld=%lld < d=%lld", (long long)p->ld, (long long)p->d);
    CIAO_REQUIRE(p->N_total >= p->N && p->N_total >= 1, "N_total=%lld must be >= max(N,1)", (long long)p->N_total);
    if (p->N > 0 && p->loss != CIAO_LOSS_ZERO) {   // Zero() terms carry no data (SVRG.jl:58)
        CIAO_REQUIRE(p->A, "problem.A is NULL");
        CIAO_REQUIRE(p->b, "problem.b is NULL");
    }
    return CIAO_OK;
}

static int32_t check_prox(const ciao_prox_desc *g)
{
    if (!g) return CIAO_OK;
    CIAO_REQUIRE(g->kind >= CIAO_PROX_ZERO && g->kind <= CIAO_PROX_L1_COMPLEX, "unknown prox kind %d", g->kind);
    if (g->kind == CIAO_PROX_L1 || g->kind == CIAO_PROX_L1_COMPLEX) CIAO_REQUIRE(g->lam >= 0.0, "NormL1 lambda must be >= 0");
    return CIAO_OK;
}

// a complex problem takes g = Zero or the complex NormL1; a real problem never the complex NormL1
static int32_t check_pair(const ciao_problem *p, const ciao_prox_desc *g)
{
    const int kind = g ? g->kind : CIAO_PROX_ZERO;
    if (p->loss == CIAO_LOSS_LS_COMPLEX)
        CIAO_REQUIRE(kind == CIAO_PROX_ZERO || kind == CIAO_PROX_L1_COMPLEX,
                     "a complex problem takes g = Zero or CIAO_PROX_L1_COMPLEX (got prox kind %d)", kind);
    else
        CIAO_REQUIRE(kind != CIAO_PROX_L1_COMPLEX, "CIAO_PROX_L1_COMPLEX needs a complex problem (CIAO_LOSS_LS_COMPLEX)");
    return CIAO_OK;
}

// the sharing problem's iterates are real: the complex NormL1 has no meaning for it (and no ProShI kernel implements it)
static int32_t check_real_prox(const ciao_prox_desc *g, const char *who)
{
    if (g && g->kind == CIAO_PROX_L1_COMPLEX) {
        set_error("%s: CIAO_PROX_L1_COMPLEX is not supported for the sharing problem (real iterates only)", who);
        return CIAO_ERR_UNSUPPORTED;
    }
    return CIAO_OK;
}

template <typename T>
static RowsArgs<T> rows_args(const ciao_problem *p)
{
    RowsArgs<T> a{};
    a.A = (p->loss == CIAO_LOSS_ZERO) ? nullptr : (const T *)p->A;
    a.b = (p->loss == CIAO_LOSS_ZERO) ? nullptr : (const T *)p->b;
    a.ld = p->ld;
    a.d = p->d;
    a.loss = p->loss;
    a.lam = (T)p->lam;
    a.row0 = 0;
    a.nrows = p->N;
    a.idx = nullptr;
    a.invN = T(1) / (T)p->N_total;
    a.N = p->N;
    a.gam_uniform = T(1);
    a.hat_gamma = T(0);
    return a;
}

template <typename T>
static ChainArgs<T> chain_args(const ciao_problem *p, const ciao_prox_desc *g)
{
    ChainArgs<T> a{};
    a.A = (p->loss == CIAO_LOSS_ZERO) ? nullptr : (const T *)p->A;
    a.b = (p->loss == CIAO_LOSS_ZERO) ? nullptr : (const T *)p->b;
    a.ld = p->ld;
    a.d = p->d;
    a.loss = p->loss;
    a.lam = (T)p->lam;
    a.batch = 1;
    a.invN = T(1) / (T)p->N_total;
    a.gam_uniform = T(1);
    a.g = make_prox<T>(g);
    a.N = p->N;
    a.nshards = 0;
    return a;
}

// the chain over a shard table: indices are GLOBAL rows, every shard's base pointer is valid on this device
template <typename T>
static void chain_shards(const ciao_ctx *ctx, const ciao_problem *p, ChainArgs<T> &a, bool with_table)
{
    const ciao_shard_table &sh = ctx->shards;
    a.nshards = sh.nshards;
    a.N = p->N_total;
    for (int k = 0; k < CIAO_MAX_SHARDS; ++k) {
        const bool on = k < sh.nshards;
        a.shA[k] = on ? (const T *)sh.A[k] : nullptr;
        a.shb[k] = on ? (const T *)sh.b[k] : nullptr;
        a.shT[k] = (on && with_table) ? (T *)sh.table[k] : nullptr;
    }
    for (int k = 0; k <= CIAO_MAX_SHARDS; ++k) a.sh_row0[k] = k <= sh.nshards ? sh.row0[k] : sh.row0[sh.nshards];
}

// Row-sharded chains: after the owner's chain, vectors u and v (d each) are all-reduced with the non-owners contributing
// zeros, so every rank ends with the owner's values (the sum adds exact zeros: bitwise the owner's).
// `nsteps` = the length of the chain the owner runs first: the other ranks enqueue their half of the reduction at once and WAIT in
// it for the owner's whole chain kernel -- seconds at 10^7 steps.  RCCL and a host hook tolerate any skew; the peer mailboxes'
// wait is bounded in wall-clock time, so this one reduction is given the chain's worth of it (2 us per step is several times
// the slowest chain, remote rows included) on top of the ordinary limit.
template <typename T>
static int32_t broadcast_from_owner(ciao_ctx *ctx, int64_t d, void *u, void *v, int64_t nsteps)
{
    struct Once {
        ciao_ctx *c;
        ~Once() { c->peer_wait_s_once = 0; }
    } once{ctx};
    ctx->peer_wait_s_once = ctx->peer_timeout_s + 60 + nsteps / 500000;
    const size_t bytes = (size_t)d * sizeof(T);
    CIAO_TRY(ensure(ctx, &ctx->sumbuf, &ctx->sumbuf_bytes, 2 * bytes + sizeof(T)));
    char *buf = (char *)ctx->sumbuf;
    if (ctx->shards.owner) {
        CIAO_HIP(hipMemcpyAsync(buf, u, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        CIAO_HIP(hipMemcpyAsync(buf + bytes, v, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        CIAO_HIP(hipMemsetAsync(buf, 0, 2 * bytes, ctx->stream));
    }
    const int32_t hs = ctx->hook(ctx->hook_user, buf, 2 * d, sizeof(T) == 8 ? CIAO_F64 : CIAO_F32, (void *)ctx->stream);
    if (hs != 0) {
        set_error("all-reduce hook failed with status %d", hs);
        return CIAO_ERR_HOOK;
    }
    CIAO_HIP(hipMemcpyAsync(u, buf, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    CIAO_HIP(hipMemcpyAsync(v, buf + bytes, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CIAO_OK;
}

template <typename T>
static Epilogue<T> epi_zero()
{
    Epilogue<T> e{};
    e.p0 = T(1);
    return e;
}

template <typename T>
static int32_t prox_launch(ciao_ctx *ctx, int64_t d, const ciao_prox_desc *g, const T *x, T gamma, T scale, T *y)
{
    hipLaunchKernelGGL((prox_kernel<T>), dim3((unsigned)((d + 255) / 256)), dim3(256), 0, ctx->stream, d, make_prox<T>(g), x,
                       gamma, scale, y);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

// ---- objective monitor (ciao_ctx_set_monitor) ------------------------------------------------------
// While a monitor is set, every FULL pass over the rows (mode GRAD, no index list) also sums the values f_i(x) the
// reference's gradient! returns and discards: they ride on the same sweep as the extra reduced scalar (all-reduced with the
// d-vector on a row-sharded problem), the epilogue leaves (1/N) sum_i f_i(x) in obj[1], and one small launch adds g(x).
template <typename T>
static void monitor_begin(const ciao_ctx *ctx, const ciao_problem *p, RowsArgs<T> &a, Epilogue<T> &e)
{
    if (!ctx->monitor) return;
    a.want_fval = 1;
    e.obj_out = ctx->monitor;
    e.obj_scale = 1.0 / (double)p->N_total;
}

template <typename T>
static int32_t monitor_end(ciao_ctx *ctx, int64_t d, const void *x)
{
    if (!ctx->monitor) return CIAO_OK;
    hipLaunchKernelGGL((gvalue_kernel<T>), dim3(1), dim3(256), 0, ctx->stream, d, make_prox<T>(ctx->monitor_has_g ? &ctx->monitor_g : nullptr),
                       (const T *)x, ctx->monitor + 2, ctx->monitor);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

// ---- typed implementations -------------------------------------------------------------------------------------------
template <typename T>
static int32_t full_gradient_t(ciao_ctx *ctx, const ciao_problem *p, const void *x, void *av, bool keep_rowdots = false)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x;
    ctx->rowdot_A = nullptr;   // whatever was cached belongs to an older pass
    if (keep_rowdots && ctx->svrg_cache_rowdots && !ctx->hook && p->N > 0 && p->loss != CIAO_LOSS_ZERO) {
        CIAO_TRY(ensure(ctx, &ctx->rowdot, &ctx->rowdot_bytes, (size_t)p->N * sizeof(T)));
        a.rowdot_out = (T *)ctx->rowdot;
        ctx->rowdot_A = p->A;
        ctx->rowdot_x = x;
        ctx->rowdot_N = p->N;
    }
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = a.invN;   // av = sum / N
    e.av_out = (T *)av;
    monitor_begin<T>(ctx, p, a, e);
    CIAO_TRY(launch_rows<T>(ctx, RM_GRAD, a, e));
    return monitor_end<T>(ctx, p->d, x);
}

template <typename T>
static int32_t proxgrad_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, const void *x, void *av,
                          void *y)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x;
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = a.invN;
    e.av_out = (T *)av;
    e.z_out = (T *)y;   // y = prox_{gamma g}(x - gamma*av)
    e.tau = (T)gamma;
    e.p0 = -(T)gamma;
    e.p1 = T(1);
    e.pw = (const T *)x;
    e.g = make_prox<T>(g);
    if (y == x && ctx->monitor) {   // the monitor reads x after the sweep: keep a copy when the step overwrites it
        CIAO_TRY(ensure(ctx, &ctx->monx, &ctx->monx_bytes, (size_t)p->d * sizeof(T)));
        CIAO_HIP(hipMemcpyAsync(ctx->monx, x, (size_t)p->d * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    }
    monitor_begin<T>(ctx, p, a, e);
    CIAO_TRY(launch_rows<T>(ctx, RM_GRAD, a, e));
    return monitor_end<T>(ctx, p->d, (y == x && ctx->monitor) ? ctx->monx : x);
}

template <typename T>
static int32_t svrg_inner_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int64_t m,
                            const int64_t *idx, const void *av, void *z, const void *z_full, void *w, bool use_rowdots = false)
{
    ChainArgs<T> a = chain_args<T>(p, g);
    a.nsteps = m;
    a.idx = idx;
    a.gamma = (T)gamma;
    a.av = (T *)av;   // read-only for SVRG
    a.z = (T *)z;
    a.zf = (T *)z_full;
    a.w = (T *)w;
    if (ctx->shards.nshards > 0) {   // row-sharded problem: the owner runs the one chain over every shard's rows
        if (ctx->shards.owner) {
            chain_shards<T>(ctx, p, a, false);
            CIAO_TRY(launch_chain<T>(ctx, CA_SVRG, a));
        }
        return ctx->hook ? broadcast_from_owner<T>(ctx, p->d, z, w, m) : CIAO_OK;
    }
    // a_i'z_full for every row is already known if the last full pass on this ctx was the one at this z_full
    if (use_rowdots && ctx->rowdot_A == p->A && ctx->rowdot_x == z_full && ctx->rowdot_N == p->N && ctx->rowdot_A) {
        a.gam = (const T *)ctx->rowdot;
        return launch_chain<T>(ctx, CA_SVRGC, a);
    }
    return launch_chain<T>(ctx, CA_SVRG, a);
}

// SVRG_basic.jl:84-92: z_full = z / m; basic: w = z_full; z = 0; av = (1/N) sum_i grad f_i(z_full)
template <typename T>
static int32_t svrg_epoch_tail_t(ciao_ctx *ctx, const ciao_problem *p, int64_t m, int32_t plus, void *av, void *z, void *z_full, void *w)
{
    ctx->rowdot_A = nullptr;   // z_full is about to change
    hipLaunchKernelGGL((svrg_tail_kernel<T>), dim3((unsigned)((p->d + 255) / 256)), dim3(256), 0, ctx->stream, p->d, (T)m,
                       (int)plus, (T *)z, (T *)z_full, (T *)w);
    CIAO_HIP(hipGetLastError());
    return full_gradient_t<T>(ctx, p, z_full, av, true);
}

// K full passes at K iterates: ONE pass over A on the matrix cores where the shape allows (mrhs_kernels.h), K single sweeps otherwise
template <typename T>
static int32_t full_gradient_multi_t(ciao_ctx *ctx, const ciao_problem *p, int32_t K, const void *const *x, void *const *av)
{
    ctx->rowdot_A = nullptr;   // whatever was cached belongs to an older pass
    if (mrhs_supported<T>(ctx, p, K, x, av)) return launch_mrhs<T>(ctx, p, K, x, av);
    for (int k = 0; k < K; ++k) CIAO_TRY(full_gradient_t<T>(ctx, p, x[k], av[k]));
    return CIAO_OK;
}

// SVRG_basic.jl:84-92 for K solves over the same rows: every solve's tail (z_full = z / m; w = z_full; z = 0), then the K full
// passes at the K new z_full as one multi-right-hand-side pass
template <typename T>
static int32_t svrg_epoch_tail_multi_t(ciao_ctx *ctx, const ciao_problem *p, int32_t K, int64_t m, int32_t plus, void *const *av,
                                       void *const *z, void *const *z_full, void *const *w)
{
    ctx->rowdot_A = nullptr;
    for (int k = 0; k < K; ++k) {
        hipLaunchKernelGGL((svrg_tail_kernel<T>), dim3((unsigned)((p->d + 255) / 256)), dim3(256), 0, ctx->stream, p->d, (T)m, (int)plus,
                           (T *)z[k], (T *)z_full[k], (T *)w[k]);
    }
    CIAO_HIP(hipGetLastError());
    return full_gradient_multi_t<T>(ctx, p, K, const_cast<const void *const *>(z_full), av);
}

template <typename T>
static int32_t svrg_iterate_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int64_t m,
                              const int64_t *idx, int32_t plus, int32_t reuse_rowdots, void *av, void *z, void *z_full, void *w)
{
    CIAO_TRY(svrg_inner_t<T>(ctx, p, g, gamma, m, idx, av, z, z_full, w, reuse_rowdots != 0));
    return svrg_epoch_tail_t<T>(ctx, p, m, plus, av, z, z_full, w);
}

template <typename T>
static int32_t saga_init_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, const void *x0,
                           void *table, void *av, void *z)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x0;
    a.table = (T *)table;
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = a.invN;
    e.av_out = (T *)av;
    CIAO_TRY(launch_rows<T>(ctx, RM_SAGA_INIT, a, e));
    // z = prox_{gamma g}((1 - gamma) x0)      SAGA_basic.jl:48
    return prox_launch<T>(ctx, p->d, g, (const T *)x0, (T)gamma, T(1) - (T)gamma, (T *)z);
}

template <typename T>
static int32_t saga_steps_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int32_t sag,
                            int64_t nsteps, const int64_t *idx, void *table, void *av, void *z)
{
    ChainArgs<T> a = chain_args<T>(p, g);
    a.nsteps = nsteps;
    a.idx = idx;
    a.gamma = (T)gamma;
    a.sag = sag;
    a.table = (T *)table;
    a.av = (T *)av;
    a.z = (T *)z;
    if (ctx->shards.nshards > 0) {   // row-sharded problem: data rows AND table rows of the other shards are peer memory
        if (ctx->shards.owner) {
            chain_shards<T>(ctx, p, a, true);
            CIAO_TRY(launch_chain<T>(ctx, CA_SAGA, a));
        }
        return ctx->hook ? broadcast_from_owner<T>(ctx, p->d, z, av, nsteps) : CIAO_OK;
    }
    return launch_chain<T>(ctx, CA_SAGA, a);
}

template <typename T>
static int32_t finito_init_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                             const void *x0, void *table, void *av, void *z)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x0;
    a.table = (T *)table;
    a.gam = (const T *)gam;
    a.hat_gamma = (T)hat_gamma;
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = (T)hat_gamma;   // av = hat_gamma * sum_i s_i/gam_i
    e.av_out = (T *)av;
    e.z_out = (T *)z;         // z = prox_{hat_gamma g}(av)
    e.tau = (T)hat_gamma;
    e.g = make_prox<T>(g);
    return launch_rows<T>(ctx, RM_FINITO_INIT, a, e);
}

// Where the batches of a call come from: index lists (bidx[bptr[t] .. bptr[t+1]), device int64) or contiguous row blocks
// (first[t] .. first[t]+len[t]: the static batches of Finito_basic.jl:52-58, no index array at all).
struct BatchSrc {
    const int64_t *bptr = nullptr, *bidx = nullptr;    // index lists
    const int64_t *first = nullptr, *len = nullptr;    // row blocks (host arrays)
    bool blocks() const { return first != nullptr; }
    int64_t size(int64_t t) const { return blocks() ? len[t] : bptr[t + 1] - bptr[t]; }
    const int64_t *idx(int64_t t) const { return blocks() ? nullptr : bidx + bptr[t]; }
    int64_t row0(int64_t t) const { return blocks() ? first[t] : 0; }
};

// Row blocks that run as a sequential chain need their indices spelled out on the device: first[t] .. first[t]+r-1 for
// t in [t0, t1), r each, into the ctx's index workspace (host-built, one upload per run; pageable memory, so the copy is
// complete when hipMemcpyAsync returns and the host buffer may go).
static int32_t block_indices(ciao_ctx *ctx, const BatchSrc &src, int64_t t0, int64_t t1, int64_t r, const int64_t **out)
{
    const size_t n = (size_t)((t1 - t0) * r);
    CIAO_TRY(ensure(ctx, &ctx->idxbuf, &ctx->idxbuf_bytes, n * sizeof(int64_t)));
    std::vector<int64_t> h;
    try {
        h.resize(n);
    } catch (...) {
        set_error("out of host memory (%zu block indices)", n);
        return CIAO_ERR_ALLOC;
    }
    size_t k = 0;
    for (int64_t t = t0; t < t1; ++t)
        for (int64_t q = 0; q < r; ++q) h[k++] = src.first[t] + q;
    CIAO_HIP(hipMemcpyAsync(ctx->idxbuf, h.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    CIAO_HIP(hipStreamSynchronize(ctx->stream));   // h goes out of scope: be certain the staging copy has consumed it
    *out = (const int64_t *)ctx->idxbuf;
    return CIAO_OK;
}

// Batches of r samples: r dependent chain steps in one persistent workgroup, or one batch-parallel rows launch?
// "chain_max_batch" >= 0 fixes the crossover; -1 (default) derives it from measurements on MI355X: a batch-parallel step
// costs ~10 us of launches + latency whatever r (tools/finito_batch_time.py), a chain step 0.45-1.5 us growing with the row.
template <typename T>
static bool batch_as_chain(const ciao_ctx *ctx, const ciao_problem *p, int64_t r)
{
    if (ctx->hook) return false;                       // sharded batches need the all-reduce between kernels
    // beyond 8192 elements: tiny batches only -- the several-workgroup chain (chain_wide_kernel, up to 131 072 elements) takes about 3 us
    // per sample, the one-workgroup any-d chain 12 us and more, a batch-parallel step of such rows 10 us and more
    if (p->d > 32 * CHAIN_NT) return r <= ((p->d <= 131072 && !ctx->chain_no_wide) ? 3 : 2);
    if (sizeof(T) == 8 && p->d > 16 * CHAIN_NT && !ctx->chain_no_wide && ctx->chain_max_batch < 0) return r <= 3;   // (fp64 rows of 4097 .. 8192 elements: the same chain)
    int64_t lim = ctx->chain_max_batch;
    if (lim < 0) {
        // measured after the chain work of round 2 (tools/gpu_s34.sh, profiles/r02_batch_chain_crossover.txt; Finito, fp32): a
        // batch-parallel step costs 7.5-9.5 us up to r = 64 whatever r; a chain step 0.39 / 0.50 / 0.82 / 1.6 us per sample at
        // 4 / 8 / 16 / 32 KiB rows, 0.29-0.34 us on the single-wave shapes (rows of up to 2 KiB)
        const int64_t rowb = p->d * (int64_t)sizeof(T);
        if (rowb % 4096 == 0 && rowb <= 32768 && rowb != 12288 && rowb != 20480 && rowb != 24576 && rowb != 28672)
            lim = rowb == 4096 ? 22 : (rowb == 8192 ? 15 : (rowb == 16384 ? 10 : 5));
        else if (rowb < 1024)
            lim = 28;   // rows under 1 KiB: chain steps 0.2-0.5 us against ~14 us per batch on the scalar generic kernel
        else if (rowb <= 2048 && rowb % 16 == 0)
            lim = 22;
        else
            lim = 14;
    }
    return r <= lim;
}

template <typename T>
static int32_t finito_steps_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                              int64_t nit, const BatchSrc &src, void *table, void *av, void *z)
{
    if (ctx->batch_open && nit > 0) {
        // recorded into a chain batch: the whole call must be ONE chain launch -- every iteration's batch of the same small size
        // (a second launch of the same solve would have to wait for the first: not something a batch can express)
        const int64_t r0 = src.size(0);
        bool uniform = !src.blocks() && r0 >= 1;
        for (int64_t tt = 1; uniform && tt < nit; ++tt) uniform = src.size(tt) == r0;
        if (!uniform || !batch_as_chain<T>(ctx, p, r0)) {
            set_error("a chain batch takes Finito steps whose batches all have the same size, small enough to run as a sequential "
                      "chain (index-list form; option chain_max_batch), got %lld iterations starting with a batch of %lld",
                      (long long)nit, (long long)r0);
            return CIAO_ERR_UNSUPPORTED;
        }
    }
    int64_t t = 0;
    while (t < nit) {
        const int64_t r = src.size(t);
        // on a row-sharded problem a batch may have no member on this rank: it still joins the all-reduce with a zero sum
        CIAO_REQUIRE(r >= 1 || (ctx->hook && r == 0), "Finito batch %lld is empty", (long long)t);
        // a run of consecutive batches of the same size
        int64_t t1 = t + 1;
        while (t1 < nit && src.size(t1) == r) ++t1;
        if (batch_as_chain<T>(ctx, p, r)) {
            ChainArgs<T> a = chain_args<T>(p, g);
            a.nsteps = (t1 - t) * r;
            a.idx = src.idx(t);
            if (src.blocks()) CIAO_TRY(block_indices(ctx, src, t, t1, r, &a.idx));
            a.batch = r;
            a.gam = (const T *)gam;
            a.hat_gamma = (T)hat_gamma;
            a.table = (T *)table;
            a.av = (T *)av;
            a.z = (T *)z;
            CIAO_TRY(launch_chain<T>(ctx, CA_FINITO, a));
        } else {
            auto one = [&](int64_t tt) -> int32_t {
                RowsArgs<T> a = rows_args<T>(p);
                a.nrows = r;
                a.idx = src.idx(tt);
                a.row0 = src.row0(tt);
                a.x1 = (const T *)z;
                a.table = (T *)table;
                a.gam = (const T *)gam;
                a.hat_gamma = (T)hat_gamma;
                Epilogue<T> e = epi_zero<T>();
                e.c_acc = T(1);   // av += sum_i (t_i - s_i) hat_gamma/gam_i
                e.acc_in = (const T *)av;
                e.c_sum = T(1);
                e.av_out = (T *)av;
                e.z_out = (T *)z;   // z = prox_{hat_gamma g}(av)
                e.tau = (T)hat_gamma;
                e.g = make_prox<T>(g);
                return launch_rows<T>(ctx, RM_FINITO_BATCH, a, e);
            };
            {
                for (int64_t tt = t; tt < t1; ++tt) CIAO_TRY(one(tt));
            }
        }
        t = t1;
    }
    return CIAO_OK;
}

template <typename T>
static int32_t lfinito_init_t(ciao_ctx *ctx, const ciao_problem *p, double hat_gamma, const void *x0, void *av, void *z,
                              void *z_full)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x0;
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = -((T)hat_gamma * a.invN);   // av = x0 - (hat_gamma/N) sum grad f_i(x0)
    e.c_u = T(1);
    e.u = (const T *)x0;
    e.av_out = (T *)av;
    monitor_begin<T>(ctx, p, a, e);
    CIAO_TRY(launch_rows<T>(ctx, RM_GRAD, a, e));
    CIAO_TRY(monitor_end<T>(ctx, p->d, x0));
    const size_t bytes = (size_t)p->d * sizeof(T);
    CIAO_HIP(hipMemcpyAsync(z, av, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    CIAO_HIP(hipMemcpyAsync(z_full, av, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CIAO_OK;
}

template <typename T>
static int32_t lfinito_iterate_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam,
                                 double hat_gamma, int64_t nb, const BatchSrc &src, void *av, void *z, void *z_full)
{
    const T hg = (T)hat_gamma;
    // z_full = prox_{hg g}(av)                                   Finito_LFinito.jl:83
    CIAO_TRY(prox_launch<T>(ctx, p->d, g, (const T *)av, hg, T(1), (T *)z_full));
    {   // av = z_full - (hg/N) sum_i grad f_i(z_full)            :84-88
        RowsArgs<T> a = rows_args<T>(p);
        a.x1 = (const T *)z_full;
        Epilogue<T> e = epi_zero<T>();
        e.c_sum = -(hg * a.invN);
        e.c_u = T(1);
        e.u = (const T *)z_full;
        e.av_out = (T *)av;
        monitor_begin<T>(ctx, p, a, e);
        CIAO_TRY(launch_rows<T>(ctx, RM_GRAD, a, e));
        CIAO_TRY(monitor_end<T>(ctx, p->d, z_full));
    }
    int64_t t = 0;
    bool z_ready = false;
    while (t < nb) {
        const int64_t r = src.size(t);
        CIAO_REQUIRE(r >= 1 || (ctx->hook && r == 0), "LFinito batch %lld is empty", (long long)t);
        int64_t t1 = t + 1;
        while (t1 < nb && src.size(t1) == r) ++t1;
        if (batch_as_chain<T>(ctx, p, r)) {
            ChainArgs<T> a = chain_args<T>(p, g);
            a.nsteps = (t1 - t) * r;
            a.idx = src.idx(t);
            if (src.blocks()) CIAO_TRY(block_indices(ctx, src, t, t1, r, &a.idx));
            a.batch = r;
            a.gam = (const T *)gam;
            a.hat_gamma = hg;
            a.av = (T *)av;
            a.z = (T *)z;
            a.zf = (T *)z_full;
            CIAO_TRY(launch_chain<T>(ctx, CA_LFINITO, a));
            z_ready = false;
        } else {
            for (int64_t tt = t; tt < t1; ++tt) {
                // z = prox_{hg g}(av), :92 -- already written by the previous batch's epilogue when there was one
                if (!z_ready) CIAO_TRY(prox_launch<T>(ctx, p->d, g, (const T *)av, hg, T(1), (T *)z));
                RowsArgs<T> a = rows_args<T>(p);
                a.nrows = r;
                a.idx = src.idx(tt);
                a.row0 = src.row0(tt);
                a.x1 = (const T *)z_full;   // acc += (coef(z_full) - coef(z)) a_i
                a.x2 = (const T *)z;
                a.gam = (const T *)gam;
                a.hat_gamma = hg;
                Epilogue<T> e = epi_zero<T>();
                e.c_acc = T(1);
                e.acc_in = (const T *)av;
                e.c_sum = hg * a.invN;
                e.uv_extra = 1;   // + (sum_i hg/gam_i) * (z - z_full)
                e.c_u = T(1);
                e.u = (const T *)z;
                e.c_v = T(-1);
                e.v = (const T *)z_full;
                e.av_out = (T *)av;
                // every batch but the last also leaves the NEXT batch's z = prox_{hg g}(av) (one launch less per batch;
                // each thread reads its own z[k] above before it overwrites it).  After the last batch z stays the z that
                // batch worked with: that is the reference's `solution` (Finito_LFinito.jl:105).
                z_ready = (tt + 1 < nb);
                if (z_ready) {
                    e.z_out = (T *)z;
                    e.tau = hg;
                    e.g = make_prox<T>(g);
                }
                CIAO_TRY(launch_rows<T>(ctx, RM_GRAD2, a, e));
            }
        }
        t = t1;
    }
    return CIAO_OK;
}

template <typename T>
static int32_t afinito_init_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double alpha, const void *x0,
                              void *table, void *meta, void *av, void *z, void *hat_gamma_dev, const void *gam_override)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x0;
    a.table = (T *)table;
    a.meta = (T *)meta;
    a.gam = (const T *)gam_override;   // stepsizes the host resolved by re-probing (entries > 0 are used as they are)
    a.alpha = (T)alpha;
    a.Nd = (double)p->N_total;
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = T(1);
    e.inv_extra = 1;              // hat_gamma = 1 / sum_i 1/gamma_i ; av = hat_gamma * sum ; z = prox_{hat_gamma g}(av)
    e.hg_out = (T *)hat_gamma_dev;
    e.av_out = (T *)av;
    e.z_out = (T *)z;
    e.g = make_prox<T>(g);
    return launch_rows<T>(ctx, RM_AFINITO_INIT, a, e);
}

template <typename T>
static int32_t afinito_steps_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double alpha, double tol_b,
                               int64_t nsteps, const int64_t *idx, void *table, void *meta, void *av, void *z,
                               void *hat_gamma_dev, int64_t *done_host, int64_t *trials_host)
{
    AFinitoArgs<T> a{};
    a.A = (const T *)p->A;
    a.b = (const T *)p->b;
    a.ld = p->ld;
    a.d = p->d;
    a.N = p->N;
    a.lam = (T)p->lam;
    a.nsteps = nsteps;
    a.idx = idx;
    a.alpha = (T)alpha;
    a.tol_b = (T)tol_b;
    a.invN = T(1) / (T)p->N_total;
    a.Nf = (T)p->N_total;
    a.Nd = (double)p->N_total;
    a.g = make_prox<T>(g);
    a.table = (T *)table;
    a.meta = (T *)meta;
    a.av = (T *)av;
    a.z = (T *)z;
    a.hg = (T *)hat_gamma_dev;
    a.counters = reinterpret_cast<long long *>(ctx->scal);
    const ciao_shard_table &sh = ctx->shards;
    const bool sharded = sh.nshards > 0;
    if (sharded) {   // row-sharded problem: data rows, table rows and scalars of the other shards are peer memory (as the SAGA chain)
        a.nshards = sh.nshards;
        a.N = p->N_total;
        for (int k = 0; k < CIAO_MAX_SHARDS; ++k) {
            const bool on = k < sh.nshards;
            a.shA[k] = on ? (const T *)sh.A[k] : nullptr;
            a.shb[k] = on ? (const T *)sh.b[k] : nullptr;
            a.shT[k] = on ? (T *)sh.table[k] : nullptr;
            a.shM[k] = on ? (T *)sh.meta[k] : nullptr;
        }
        for (int k = 0; k <= CIAO_MAX_SHARDS; ++k) a.sh_row0[k] = k <= sh.nshards ? sh.row0[k] : sh.row0[sh.nshards];
    }
    long long c[2] = {0, 0};
    if (!sharded || sh.owner) {
        CIAO_TRY(launch_afinito<T>(ctx, p->loss, a));
        CIAO_HIP(hipMemcpyAsync(c, ctx->scal, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
        CIAO_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (sharded && ctx->hook) {
        // av and z as the chains' hand-over; hat_gamma and the two counters as three Float64 (a Float32 is exact in one, a count
        // of steps below 2^53 too) behind them, the non-owners contributing zeros
        CIAO_TRY(broadcast_from_owner<T>(ctx, p->d, av, z, nsteps));
        double t[3] = {0.0, 0.0, 0.0};
        if (sh.owner) {
            T hgv;
            CIAO_HIP(hipMemcpyAsync(&hgv, hat_gamma_dev, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            CIAO_HIP(hipStreamSynchronize(ctx->stream));
            t[0] = (double)hgv, t[1] = (double)c[0], t[2] = (double)c[1];
        }
        CIAO_HIP(hipMemcpyAsync(ctx->scal, t, sizeof t, hipMemcpyHostToDevice, ctx->stream));
        const int32_t hs = ctx->hook(ctx->hook_user, ctx->scal, 3, CIAO_F64, (void *)ctx->stream);
        if (hs != 0) {
            set_error("all-reduce hook failed with status %d", hs);
            return CIAO_ERR_HOOK;
        }
        CIAO_HIP(hipMemcpyAsync(t, ctx->scal, sizeof t, hipMemcpyDeviceToHost, ctx->stream));
        CIAO_HIP(hipStreamSynchronize(ctx->stream));
        const T hgv = (T)t[0];
        CIAO_HIP(hipMemcpyAsync(hat_gamma_dev, &hgv, sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        CIAO_HIP(hipStreamSynchronize(ctx->stream));
        c[0] = (long long)t[1], c[1] = (long long)t[2];
    }
    if (done_host) *done_host = c[0];
    if (trials_host) *trials_host = c[1];
    return CIAO_OK;
}

template <typename T>
static ProshiArgs<T> proshi_args(const ciao_sepquad *f, const void *gam, void *table)
{
    ProshiArgs<T> a{};
    a.Q = (const T *)f->Q;
    a.q = (const T *)f->q;
    a.ld = f->ld;
    a.d = f->d;
    a.N = f->N;
    a.eta = (T)f->eta;
    a.lo = (T)f->lo;
    a.hi = (T)f->hi;
    a.gam = (const T *)gam;
    a.invN = T(1) / (T)f->N_total;
    a.table = (T *)table;
    a.dense = f->dense;
    return a;
}

template <typename T>
static int32_t proshi_init_t(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam, const void *x0,
                             void *table, void *av, void *z, void *hat_gamma_dev)
{
    ProshiArgs<T> a = proshi_args<T>(f, gam, table);
    a.x = (const T *)x0;
    a.nrows = f->N;
    a.idx = nullptr;
    a.row0 = 0;
    Epilogue<T> e = epi_zero<T>();
    e.c_sum = T(1);            // av = sum_i s_i
    e.inv_extra = 2;           // hat_gamma = sum_i gam_i
    e.hg_out = (T *)hat_gamma_dev;
    e.av_out = (T *)av;
    e.z_out = (T *)z;          // z = (prox_{hat_gamma g}(av) - av) / hat_gamma
    e.zmode = 1;
    e.g = make_prox<T>(g);
    return launch_proshi<T>(ctx, true, a, e);
}

template <typename T>
static int32_t proshi_steps_t(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                              int64_t nit, const BatchSrc &src, void *table, void *av, void *z)
{
    for (int64_t t = 0; t < nit; ++t) {
        const int64_t r = src.size(t);
        CIAO_REQUIRE(r >= 1 || (ctx->hook && r == 0), "ProShI batch %lld is empty", (long long)t);
        // Small batches of separable agents: a whole run of equal-sized batches is ONE coordinate-parallel launch
        // (proshi_chain_kernel) instead of two launches per iteration.  Crossover (option proshi_chain_max_batch, -1 = automatic),
        // measured at d = 1024 (tools/proshi_chain_time.py): a visit costs the chain 0.31-0.35 us in fp64, 0.24-0.26 in fp32; a
        // batch-parallel iteration 6.1-6.7 us whatever r.
        const int64_t lim = ctx->proshi_chain_max_batch >= 0 ? ctx->proshi_chain_max_batch : 18;
        if (!ctx->hook && !f->dense && r >= 1 && r <= lim) {
            int64_t t1 = t + 1;
            while (t1 < nit && src.size(t1) == r) ++t1;
            ProshiChainArgs<T> c{};
            c.Q = (const T *)f->Q;
            c.q = (const T *)f->q;
            c.ld = f->ld;
            c.d = f->d;
            c.N = f->N;
            c.eta = (T)f->eta;
            c.lo = (T)f->lo;
            c.hi = (T)f->hi;
            c.gam = (const T *)gam;
            c.invN = T(1) / (T)f->N_total;
            c.hat_gamma = (T)hat_gamma;
            c.idx = src.idx(t);
            if (src.blocks()) CIAO_TRY(block_indices(ctx, src, t, t1, r, &c.idx));
            c.nvisits = (t1 - t) * r;
            c.batch = r;
            c.g = make_prox<T>(g);
            c.table = (T *)table;
            c.av = (T *)av;
            c.z = (T *)z;
            c.errflag = ctx->errflag;
            const int64_t grid = (f->d + 255) / 256;
            constexpr size_t lds = proshi_chain_lds_bytes<T>();
            auto kern = &proshi_chain_kernel<T>;
            if (lds > 60 * 1024)
                CIAO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, c);
            CIAO_HIP(hipGetLastError());
            char nm[128];
            snprintf(nm, sizeof nm, "proshi_chain_kernel<%s> grid=%lld block=256 visits=%lld batch=%lld", sizeof(T) == 8 ? "f64" : "f32",
                     (long long)grid, (long long)c.nvisits, (long long)r);
            ctx->last_kernel = nm;
            t = t1 - 1;
            continue;
        }
        ProshiArgs<T> a = proshi_args<T>(f, gam, table);
        a.x = (const T *)z;
        a.nrows = r;
        a.idx = src.idx(t);
        a.row0 = src.row0(t);
        Epilogue<T> e = epi_zero<T>();
        e.c_acc = T(1);        // av += sum_{i in batch} (s_i_new - s_i_old)
        e.acc_in = (const T *)av;
        e.c_sum = T(1);
        e.av_out = (T *)av;
        e.z_out = (T *)z;
        e.zmode = 1;
        e.tau = (T)hat_gamma;
        e.g = make_prox<T>(g);
        CIAO_TRY(launch_proshi<T>(ctx, false, a, e));
    }
    return CIAO_OK;
}

static int32_t check_sepquad(ciao_ctx *ctx, const ciao_sepquad *f)
{
    CIAO_REQUIRE(ctx && f, "ctx or problem is NULL");
    ctx->rowdot_A = nullptr;
    CIAO_REQUIRE(f->dtype == CIAO_F32 || f->dtype == CIAO_F64, "bad dtype %d", f->dtype);
    CIAO_REQUIRE(f->N >= 0 && f->d >= 1 && f->ld >= f->d && f->N_total >= f->N && f->N_total >= 1, "bad shape");
    CIAO_REQUIRE(f->dense == 0 || f->dense == 1, "ciao_sepquad.dense must be 0 or 1 (got %d)", f->dense);
    CIAO_REQUIRE(f->N == 0 || (f->Q && f->q), "Q or q is NULL");
    CIAO_REQUIRE(f->eta >= 0 && f->lo <= f->hi, "need eta >= 0 and lo <= hi");
    return CIAO_OK;
}

template <typename T>
static int32_t objective_t(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *x, double *obj)
{
    RowsArgs<T> a = rows_args<T>(p);
    a.x1 = (const T *)x;
    a.want_fval = 1;
    CIAO_TRY(launch_rows_raw<T>(ctx, RM_GRAD, a));
    hipLaunchKernelGGL((gvalue_kernel<T>), dim3(1), dim3(256), 0, ctx->stream, p->d, make_prox<T>(g), (const T *)x, ctx->scal,
                       (double *)nullptr);
    CIAO_HIP(hipGetLastError());
    T fsum;
    double gval;
    CIAO_HIP(hipMemcpyAsync(&fsum, (const T *)ctx->sumbuf + p->d, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    CIAO_HIP(hipMemcpyAsync(&gval, ctx->scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CIAO_HIP(hipStreamSynchronize(ctx->stream));
    *obj = (double)fsum / (double)p->N_total + gval;
    return CIAO_OK;
}

}  // namespace ciao

using namespace ciao;

#define DISPATCH(dtype, fn, ...) ((dtype) == CIAO_F64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

extern "C" {

int32_t ciao_abi_version(void) { return CIAO_ABI_VERSION; }

const char *ciao_last_error(void) { return g_err; }

#ifndef CIAO_BUILD_FLAGS
#define CIAO_BUILD_FLAGS ""
#endif
const char *ciao_build_flags(void) { return CIAO_BUILD_FLAGS; }

int32_t ciao_ctx_create(int32_t device, void *stream, ciao_ctx **out)
{
    CIAO_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    CIAO_HIP(hipGetDeviceCount(&ndev));
    CIAO_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    DeviceGuard dg(device);   // allocate on `device`, leave the caller's current device as it was
    hipDeviceProp_t prop;
    CIAO_HIP(hipGetDeviceProperties(&prop, device));
    ciao_ctx *ctx = new (std::nothrow) ciao_ctx();
    if (!ctx) {
        set_error("out of host memory");
        return CIAO_ERR_ALLOC;
    }
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipMalloc((void **)&ctx->scal, 4096 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->errflag, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->errflag, 0, sizeof(int), ctx->stream);
    if (e != hipSuccess) {
        delete ctx;
        return hip_fail(e, "ctx scratch allocation");
    }
    *out = ctx;
    return CIAO_OK;
}

int32_t ciao_ctx_destroy(ciao_ctx *ctx)
{
    if (!ctx) return CIAO_OK;
    DeviceGuard dg(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->partial) (void)hipFree(ctx->partial);
    if (ctx->mrhs_ptrs) (void)hipFree(ctx->mrhs_ptrs);
    if (ctx->wide_box) (void)hipFree(ctx->wide_box);
    if (ctx->pextra) (void)hipFree(ctx->pextra);
    if (ctx->sumbuf) (void)hipFree(ctx->sumbuf);
    if (ctx->rowdot) (void)hipFree(ctx->rowdot);
    if (ctx->monx) (void)hipFree(ctx->monx);
    if (ctx->idxbuf) (void)hipFree(ctx->idxbuf);
    if (ctx->scal) (void)hipFree(ctx->scal);
    if (ctx->errflag) (void)hipFree(ctx->errflag);
    if (ctx->peer_counter) (void)hipFree(ctx->peer_counter);
    if (ctx->batch_dev) (void)hipFree(ctx->batch_dev);
    if (ctx->batch_host) (void)hipHostFree(ctx->batch_host);
    if (ctx->batch_ev) (void)hipEventDestroy(ctx->batch_ev);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    delete ctx;
    return CIAO_OK;
}

int32_t ciao_ctx_set_stream(ciao_ctx *ctx, void *stream)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return CIAO_OK;
}

int32_t ciao_ctx_synchronize(ciao_ctx *ctx)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    int flag = 0;
    CIAO_HIP(hipMemcpyAsync(&flag, ctx->errflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CIAO_HIP(hipStreamSynchronize(ctx->stream));
    if (flag) {
        CIAO_HIP(hipMemsetAsync(ctx->errflag, 0, sizeof(int), ctx->stream));
        if (flag == 2) {
            set_error("adaptive Finito init: grad f_i(x0 .+ 1) == grad f_i(x0) for the samples with gamma_i = -1 in meta; the reference "
                      "now probes random points (Finito_adaptive.jl:78-85): resolve them with ciao_afinito_probe and the host's "
                      "draws, then repeat ciao_afinito_init with gam_override");
            return CIAO_ERR_UNSUPPORTED;
        }
        if (flag == 3) {
            set_error("internal: a wave of the wave-specialised chain kernel gave up waiting for another (chain_ws_kernel spin limit); "
                      "results since the last synchronize are invalid -- option chain_no_ws=1 selects the previous chain kernel");
            return CIAO_ERR_HIP;
        }
        if (flag == 4) {
            set_error("peer all-reduce: a rank's flag never arrived (ciao_ctx_set_peers: every rank must make the same reductions in "
                      "the same order); results since the last synchronize are invalid");
            return CIAO_ERR_HOOK;
        }
        if (flag == 5) {
            set_error("internal: a workgroup of the several-workgroup chain (chain_wide_kernel, rows beyond 8192 elements) never saw "
                      "another's partial dot product within 4 s -- the chain's workgroups were not all resident (a GPU shared with other "
                      "work?); results since the last synchronize are invalid -- option chain_no_wide=1 selects the one-workgroup kernel");
            return CIAO_ERR_HIP;
        }
        if (flag == 6) {
            set_error("internal: a workgroup of the cluster sweep (rows_long_kernel, rows beyond 64 KiB) never saw another's partial dot "
                      "product within 4 s -- the grid was not all resident (a GPU shared with other work?); results since the last "
                      "synchronize are invalid -- option long_rows=0 selects the generic kernel");
            return CIAO_ERR_HIP;
        }
        set_error("a sample index was outside [0, N): results since the last synchronize are invalid");
        return CIAO_ERR_ARG;
    }
    return CIAO_OK;
}

int32_t ciao_ctx_set_allreduce(ciao_ctx *ctx, ciao_allreduce_fn fn, void *user)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    ctx->hook = fn;
    ctx->hook_user = user;
    return CIAO_OK;
}

// the library's own all-reduce hook: RCCL called directly (no host-language callback per reduction)
static int32_t rccl_hook(void *user, void *buf, int64_t count, int32_t dtype, void *stream)
{
    ciao_ctx *ctx = static_cast<ciao_ctx *>(user);
    const int nccl_type = (dtype == CIAO_F64) ? 8 : 7;   // ncclFloat64 / ncclFloat32 (rccl.h)
    const int r = ctx->rccl_allreduce(buf, buf, (size_t)count, nccl_type, /*ncclSum*/ 0, ctx->rccl_comm, (hipStream_t)stream);
    if (r != 0) {
        set_error("ncclAllReduce failed with %d (%s)", r, ctx->rccl_errstr ? ctx->rccl_errstr(r) : "?");
        return r;
    }
    return 0;
}

int32_t ciao_ctx_set_rccl(ciao_ctx *ctx, void *comm, const char *librccl_path)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    if (!comm) {
        if (ctx->hook == rccl_hook) {
            ctx->hook = nullptr;
            ctx->hook_user = nullptr;
        }
        ctx->rccl_comm = nullptr;
        return CIAO_OK;
    }
    const char *path = librccl_path ? librccl_path : "librccl.so";
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);   // the same handle again if the host has loaded it already
    if (!h) {
        set_error("cannot load RCCL from '%s': %s", path, dlerror());
        return CIAO_ERR_ARG;
    }
    void *f = dlsym(h, "ncclAllReduce");
    if (!f) {
        set_error("'%s' has no ncclAllReduce", path);
        dlclose(h);
        return CIAO_ERR_ARG;
    }
    if (ctx->rccl_lib && ctx->rccl_lib != h) dlclose(ctx->rccl_lib);
    ctx->rccl_lib = h;
    ctx->rccl_allreduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(f);
    ctx->rccl_errstr = reinterpret_cast<const char *(*)(int)>(dlsym(h, "ncclGetErrorString"));
    ctx->rccl_comm = comm;
    ctx->hook = rccl_hook;
    ctx->hook_user = ctx;
    return CIAO_OK;
}

// ---- one-shot peer all-reduce (peer_kernels.h) ---------------------------------------------------------------------------
static int64_t peer_slot_bytes_for(int64_t max_elems) { return ((max_elems * 8 + 255) / 256) * 256; }

int32_t ciao_peer_mailbox_create(ciao_ctx *ctx, int64_t max_elems, void **mailbox_out, int64_t *bytes_out)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(mailbox_out && max_elems >= 1, "NULL argument or max_elems < 1");
    const size_t bytes = (size_t)(PEER_HDR + 2 * PEER_MAX * peer_slot_bytes_for(max_elems));
    void *m = nullptr;
    // fine-grained device memory: written by the other GPUs over xGMI while this GPU's kernels read it
    CIAO_HIP(hipExtMallocWithFlags(&m, bytes, hipDeviceMallocFinegrained));
    CIAO_HIP(hipMemset(m, 0, bytes));
    CIAO_HIP(hipDeviceSynchronize());
    *mailbox_out = m;
    if (bytes_out) *bytes_out = (int64_t)bytes;
    return CIAO_OK;
}

int32_t ciao_peer_mailbox_destroy(ciao_ctx *ctx, void *mailbox)
{
    CIAO_ENTER(ctx);
    if (mailbox) CIAO_HIP(hipFree(mailbox));
    return CIAO_OK;
}

// the library's peer hook on an arbitrary buffer (chain-owner broadcasts, bench): the two halves as kernels of their own
static int32_t peer_hook(void *user, void *buf, int64_t count, int32_t dtype, void *stream)
{
    ciao_ctx *ctx = static_cast<ciao_ctx *>(user);
    if (count > ctx->peer_max_elems) {
        set_error("the peer mailboxes hold %lld elements, this reduction has %lld", (long long)ctx->peer_max_elems, (long long)count);
        return CIAO_ERR_ARG;
    }
    const PeerDev pd = peer_dev(ctx);
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (dtype == CIAO_F64) {
        hipLaunchKernelGGL((peer_send_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double *)buf, count, pd);
        hipLaunchKernelGGL((peer_recv_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)buf, count, pd);
    } else {
        hipLaunchKernelGGL((peer_send_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)buf, count, pd);
        hipLaunchKernelGGL((peer_recv_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)buf, count, pd);
    }
    return hipGetLastError() == hipSuccess ? 0 : CIAO_ERR_HIP;
}

int32_t ciao_ctx_set_peers(ciao_ctx *ctx, int32_t rank, int32_t world, void *const *mailboxes, int64_t max_elems)
{
    CIAO_ENTER(ctx);
    if (!mailboxes || world <= 0) {   // off: what the peers displaced (a user hook, the RCCL communicator) is back in place
        if (ctx->hook == peer_hook) {
            ctx->hook = ctx->peer_saved ? ctx->peer_saved_hook : nullptr;
            ctx->hook_user = ctx->peer_saved ? ctx->peer_saved_user : nullptr;
        }
        ctx->peer_saved = false;
        ctx->peer_world = 0;
        return CIAO_OK;
    }
    CIAO_REQUIRE(world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world, "world must be in 1..%d and rank in [0, world) (got %d, %d)",
                 PEER_MAX, world, rank);
    CIAO_REQUIRE(max_elems >= 1, "max_elems < 1");
    for (int r = 0; r < world; ++r) CIAO_REQUIRE(mailboxes[r], "mailbox %d is NULL", r);
    if (!ctx->peer_counter) {
        CIAO_HIP(hipMalloc((void **)&ctx->peer_counter, sizeof(unsigned int)));
        CIAO_HIP(hipMemset(ctx->peer_counter, 0, sizeof(unsigned int)));
    }
    for (int r = 0; r < PEER_MAX; ++r) ctx->peer_mail[r] = r < world ? static_cast<unsigned char *>(mailboxes[r]) : nullptr;
    ctx->peer_world = world;
    ctx->peer_rank = rank;
    // The sequence number lives WITH the mailboxes, so that a group that is set again (off and on; another ctx) continues instead of
    // restarting at 1 and meeting flags of an earlier life that happen to equal the new numbers.  It is read from THIS
    // RANK'S OWN flag in its own mailbox -- the two words (one per parity) that only this rank's kernels ever write: every rank made
    // the same reductions, so every rank reads the same number, and no peer can change what is read.  (Until round 5 the maximum over
    // ALL ranks' flags was taken: a rank that finished set_peers first and started reduction last+1 wrote its flag into a slower
    // rank's mailbox before that rank read it, the slow rank started at last+1 and the two stayed one number -- one parity -- apart
    // until both timed out: ADVICE r4.)  Fresh mailboxes read 0.
    {
        CIAO_HIP(hipStreamSynchronize(ctx->stream));
        unsigned int flags[2 * PEER_MAX * 16];
        static_assert(sizeof flags == PEER_HDR, "the header is two parities of eight 64-byte flag lines");
        CIAO_HIP(hipMemcpy(flags, mailboxes[rank], sizeof flags, hipMemcpyDeviceToHost));
        unsigned int last = 0;
        for (int par = 0; par < 2; ++par) last = flags[(par * PEER_MAX + rank) * 16] > last ? flags[(par * PEER_MAX + rank) * 16] : last;
        ctx->peer_seq = last;
    }
    ctx->peer_max_elems = max_elems;
    ctx->peer_slot_bytes = peer_slot_bytes_for(max_elems);
    if (ctx->hook != peer_hook) {   // remember what is displaced (set twice in a row: the first one's memory stands)
        ctx->peer_saved_hook = ctx->hook;
        ctx->peer_saved_user = ctx->hook_user;
        ctx->peer_saved = true;
    }
    ctx->hook = peer_hook;
    ctx->hook_user = ctx;
    return CIAO_OK;
}

int32_t ciao_peer_allreduce(ciao_ctx *ctx, int32_t dtype, int64_t count, void *buf)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx->peer_world > 0, "no peers are set (ciao_ctx_set_peers)");
    CIAO_REQUIRE(buf && count >= 1 && (dtype == CIAO_F64 || dtype == CIAO_F32), "NULL buffer, count < 1 or unknown dtype");
    const int32_t st = peer_hook(ctx, buf, count, dtype, (void *)ctx->stream);
    return st == 0 ? CIAO_OK : (st < 0 ? st : CIAO_ERR_HOOK);
}

// ---- a batch of independent chains in one launch per kernel ---------------------------------------------------------------
int32_t ciao_ctx_chain_batch_begin(ciao_ctx *ctx)
{
    CIAO_ENTER(ctx);   // (refuses a second begin)
    ctx->batch.clear();
    ctx->batch_open = 1;
    return CIAO_OK;
}

int32_t ciao_ctx_chain_batch_end(ciao_ctx *ctx, int32_t launch)
{
    CIAO_ENTER_BATCHABLE(ctx);
    CIAO_REQUIRE(ctx->batch_open, "no chain batch is open on this ctx");
    ctx->batch_open = 0;
    std::vector<ciao_chain_rec> recs;
    recs.swap(ctx->batch);
    if (!launch || recs.empty()) return CIAO_OK;
    const size_t K = recs.size();
    // No chain may write what another one touches: sort the state ranges by their start and compare each with the ones that begin
    // inside it (read-only ranges -- a shared z_full, say -- may overlap)
    {
        struct Rg { const unsigned char *lo, *hi; bool wr; size_t chain; };
        std::vector<Rg> rg;
        rg.reserve(K * 5);
        for (size_t c = 0; c < K; ++c)
            for (int k = 0; k < 5; ++k)
                if (recs[c].hi[k] > recs[c].lo[k]) rg.push_back({recs[c].lo[k], recs[c].hi[k], recs[c].wr[k], c});
        std::sort(rg.begin(), rg.end(), [](const Rg &x, const Rg &y) { return x.lo < y.lo; });
        for (size_t i = 0; i < rg.size(); ++i)
            for (size_t j = i + 1; j < rg.size() && rg[j].lo < rg[i].hi; ++j)
                if (rg[i].chain != rg[j].chain && (rg[i].wr || rg[j].wr)) {
                    set_error("chains %zu and %zu of the batch overlap in a state vector / table that one of them writes: the chains of a "
                              "batch run concurrently and must own their av / z / w (SVRG: w, z; SAGA: z, av, table)", rg[i].chain, rg[j].chain);
                    return CIAO_ERR_ARG;
                }
    }
    // group by kernel (first appearance first); the argument blocks of a group are contiguous in the staging buffer
    std::vector<std::vector<size_t>> groups;
    for (size_t c = 0; c < K; ++c) {
        size_t gi = 0;
        for (; gi < groups.size(); ++gi) {
            const ciao_chain_rec &f = recs[groups[gi][0]];
            if (f.kern == recs[c].kern && f.block == recs[c].block && f.lds == recs[c].lds && f.args.size() == recs[c].args.size()) break;
        }
        if (gi == groups.size()) groups.emplace_back();
        groups[gi].push_back(c);
    }
    size_t total = 0;
    for (const auto &g : groups) total += (g.size() * recs[g[0]].args.size() + 255) & ~(size_t)255;
    CIAO_TRY(ensure(ctx, &ctx->batch_dev, &ctx->batch_dev_bytes, total));
    if (!ctx->batch_ev) CIAO_HIP(hipEventCreateWithFlags(&ctx->batch_ev, hipEventDisableTiming));
    CIAO_HIP(hipEventSynchronize(ctx->batch_ev));   // the previous batch's staging copy has been read (no-op for a fresh event)
    if (ctx->batch_host_bytes < total) {
        if (ctx->batch_host) (void)hipHostFree(ctx->batch_host);
        ctx->batch_host = nullptr;
        ctx->batch_host_bytes = 0;
        CIAO_HIP(hipHostMalloc(&ctx->batch_host, total, hipHostMallocDefault));
        ctx->batch_host_bytes = total;
    }
    std::vector<size_t> off(groups.size());
    size_t at = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        off[gi] = at;
        const size_t ab = recs[groups[gi][0]].args.size();
        for (size_t k = 0; k < groups[gi].size(); ++k)
            memcpy(static_cast<unsigned char *>(ctx->batch_host) + at + k * ab, recs[groups[gi][k]].args.data(), ab);
        at += (groups[gi].size() * ab + 255) & ~(size_t)255;
    }
    CIAO_HIP(hipMemcpyAsync(ctx->batch_dev, ctx->batch_host, total, hipMemcpyHostToDevice, ctx->stream));
    CIAO_HIP(hipEventRecord(ctx->batch_ev, ctx->stream));
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        ciao_chain_rec &f = recs[groups[gi][0]];
        // the by-value argument: the first chain's block with `multi` (its first field, whatever T) -> this group's blocks
        const void *multi = static_cast<unsigned char *>(ctx->batch_dev) + off[gi];
        memcpy(f.args.data(), &multi, sizeof multi);
        void *kargs[1] = {f.args.data()};
        CIAO_HIP(hipLaunchKernel(f.kern, dim3((unsigned)groups[gi].size()), dim3(f.block), kargs, f.lds, ctx->stream));
    }
    char nm[256];
    snprintf(nm, sizeof nm, "chain batch: %zu chains in %zu launch(es); first: %s grid=%zu", K, groups.size(),
             recs[groups[0][0]].name.c_str(), groups[0].size());
    ctx->last_kernel = nm;
    return CIAO_OK;
}

int32_t ciao_ctx_set_monitor(ciao_ctx *ctx, const ciao_prox_desc *g, double *obj_dev)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_prox(g));
    ctx->monitor = obj_dev;
    ctx->monitor_has_g = (g != nullptr);
    if (g) ctx->monitor_g = *g;
    return CIAO_OK;
}

int32_t ciao_ctx_set_option(ciao_ctx *ctx, const char *key, int64_t value)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx && key, "ctx or key is NULL");
    if (!strcmp(key, "sweep_blocks_per_cu")) {
        CIAO_REQUIRE(value >= 0 && value <= 16, "sweep_blocks_per_cu must be in 0..16 (0 = automatic)");
        ctx->sweep_blocks_per_cu = value;
    } else if (!strcmp(key, "sweep_multi")) {
        ctx->sweep_multi = value != 0;
    } else if (!strcmp(key, "split_max_rows")) {
        CIAO_REQUIRE(value >= -1, "split_max_rows must be >= -1");
        ctx->split_max_rows = value;
    } else if (!strcmp(key, "small_i")) {
        CIAO_REQUIRE(value == 0 || value == 8 || value == 16, "small_i must be 0, 8 or 16");
        ctx->small_i = value;
    } else if (!strcmp(key, "split_all")) {
        ctx->split_all = value != 0;
    } else if (!strcmp(key, "split_blocks_per_cu")) {
        CIAO_REQUIRE(value >= 0 && value <= 16, "split_blocks_per_cu must be in 0..16");
        ctx->split_blocks_per_cu = value;
    } else if (!strcmp(key, "long_rows")) {
        ctx->long_rows = value != 0;
    } else if (!strcmp(key, "long_j")) {
        CIAO_REQUIRE(value == 0 || value == 4 || value == 8, "long_j must be 0, 4 or 8");
        ctx->long_j = value;
    } else if (!strcmp(key, "sweep_grid")) {
        CIAO_REQUIRE(value >= 0 && value <= 65535, "sweep_grid must be in 0..65535");
        ctx->sweep_grid = value;
    } else if (!strcmp(key, "sweep_prefetch")) {
        ctx->sweep_prefetch = value < 0 ? -1 : (value != 0);
    } else if (!strcmp(key, "chain_max_batch")) {
        CIAO_REQUIRE(value >= -1, "chain_max_batch must be >= -1 (-1 = automatic)");
        ctx->chain_max_batch = value;
    } else if (!strcmp(key, "svrg_cache_rowdots")) {
        ctx->svrg_cache_rowdots = value != 0;
        ctx->rowdot_A = nullptr;
    } else if (!strcmp(key, "proshi_chain_max_batch")) {
        CIAO_REQUIRE(value >= -1, "proshi_chain_max_batch must be >= -1 (-1 = automatic)");
        ctx->proshi_chain_max_batch = value;
    } else if (!strcmp(key, "chain_four_waves")) {
        ctx->chain_four_waves = value;
    } else if (!strcmp(key, "chain_big")) {
        ctx->chain_big = value != 0;
    } else if (!strcmp(key, "chain_no_dma")) {
        ctx->chain_no_dma = value != 0;
    } else if (!strcmp(key, "chain_no_wide")) {
        ctx->chain_no_wide = value != 0;
    } else if (!strcmp(key, "chain_no_ws")) {
        ctx->chain_no_ws = value != 0;
    } else if (!strcmp(key, "small_mfma")) {
        ctx->small_mfma = value;
    } else if (!strcmp(key, "wrow_rows_per_wave")) {
        ctx->wrow_rows_per_wave = value;
    } else if (!strcmp(key, "small_wrow")) {
        ctx->small_wrow = value;
    } else if (!strcmp(key, "small_mfma_table")) {
        ctx->small_mfma_table = value;
    } else if (!strcmp(key, "small_nb")) {
        CIAO_REQUIRE(value == 0 || (value >= 2 && value <= 4), "small_nb must be 0 or 2..4");
        ctx->small_nb = value;
    } else if (!strcmp(key, "multi_rhs_off")) {
        ctx->mrhs_off = value != 0;
    } else if (!strcmp(key, "peer_timeout_s")) {
        CIAO_REQUIRE(value >= 1 && value <= 86400, "peer_timeout_s must be in 1..86400");
        ctx->peer_timeout_s = value;
    } else if (!strcmp(key, "chain_ws_issuers")) {
        CIAO_REQUIRE(value >= 0 && value <= 2, "chain_ws_issuers must be 0 (automatic), 1 or 2");
        ctx->chain_ws_issuers = value;
    } else if (!strcmp(key, "force_generic")) {
        ctx->force_generic = value != 0;
    } else {
        set_error("unknown option '%s'", key);
        return CIAO_ERR_ARG;
    }
    return CIAO_OK;
}

int32_t ciao_ctx_timing_enable(ciao_ctx *ctx, int32_t enable)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    ctx->timing = enable != 0;
    return CIAO_OK;
}

int32_t ciao_ctx_timing_read(ciao_ctx *ctx, double *total_ms_host, int64_t *launches_host)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx && total_ms_host && launches_host, "NULL argument");
    CIAO_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        CIAO_HIP(hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        tot += ms;
    }
    *total_ms_host = tot;
    *launches_host = (int64_t)(ctx->ev_used / 2);
    ctx->ev_used = 0;
    return CIAO_OK;
}

const char *ciao_ctx_last_kernel(ciao_ctx *ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

// a shard table must describe exactly the problem the call is about
static int32_t check_shards(const ciao_ctx *ctx, const ciao_problem *p, bool need_table)
{
    const ciao_shard_table &sh = ctx->shards;
    if (sh.nshards == 0) return CIAO_OK;
    CIAO_REQUIRE(sh.row0[sh.nshards] == p->N_total, "the shard table covers %lld rows but the problem has N_total = %lld",
                 (long long)sh.row0[sh.nshards], (long long)p->N_total);
    if (sh.owner && p->loss != CIAO_LOSS_ZERO)
        for (int k = 0; k < sh.nshards; ++k) {
            if (sh.row0[k + 1] == sh.row0[k]) continue;
            CIAO_REQUIRE(sh.A[k] && sh.b[k], "shard %d has rows but no data pointers on the chain owner", k);
            CIAO_REQUIRE(!need_table || sh.table[k], "shard %d has no table pointer on the chain owner (SAGA)", k);
        }
    return CIAO_OK;
}

int32_t ciao_ctx_set_shards(ciao_ctx *ctx, const ciao_shard_table *shards)
{
    CIAO_ENTER(ctx);
    ctx->rowdot_A = nullptr;
    if (!shards) {
        ctx->shards = ciao_shard_table{};
        return CIAO_OK;
    }
    CIAO_REQUIRE(shards->nshards >= 1 && shards->nshards <= CIAO_MAX_SHARDS, "nshards must be in 1..%d (got %d)", CIAO_MAX_SHARDS,
                 shards->nshards);
    CIAO_REQUIRE(shards->row0[0] == 0, "row0[0] must be 0");
    for (int k = 0; k < shards->nshards; ++k)
        CIAO_REQUIRE(shards->row0[k + 1] >= shards->row0[k], "row0 must be non-decreasing (shard %d)", k);
    ctx->shards = *shards;
    return CIAO_OK;
}

int32_t ciao_ipc_export(const void *dev_ptr, void *handle_out, int64_t *offset_out)
{
    CIAO_REQUIRE(dev_ptr && handle_out && offset_out, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI documents 64-byte handles");
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    CIAO_HIP(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev_ptr));
    hipIpcMemHandle_t h;
    CIAO_HIP(hipIpcGetMemHandle(&h, (void *)base));
    memcpy(handle_out, &h, sizeof h);
    *offset_out = (int64_t)((const char *)dev_ptr - (const char *)base);
    return CIAO_OK;
}

int32_t ciao_ipc_open(const void *handle, int64_t offset, void **dev_ptr_out)
{
    CIAO_REQUIRE(handle && dev_ptr_out && offset >= 0, "NULL argument or negative offset");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    void *base = nullptr;
    CIAO_HIP(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
    *dev_ptr_out = (char *)base + offset;
    return CIAO_OK;
}

int32_t ciao_ipc_close(void *dev_ptr, int64_t offset)
{
    CIAO_REQUIRE(dev_ptr && offset >= 0, "NULL argument or negative offset");
    CIAO_HIP(hipIpcCloseMemHandle((char *)dev_ptr - offset));
    return CIAO_OK;
}

int32_t ciao_gradient(ciao_ctx *ctx, const ciao_problem *p, int64_t i, const void *x, void *y, void *fval)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(x && y, "x or y is NULL");
    CIAO_REQUIRE(i >= 0 && i < p->N, "sample index %lld outside [0, %lld)", (long long)i, (long long)p->N);
    if (p->dtype == CIAO_F64)
        hipLaunchKernelGGL((gradient_kernel<double>), dim3(1), dim3(WAVE), 0, ctx->stream, p->loss == CIAO_LOSS_ZERO ? nullptr : (const double *)p->A,
                           p->loss == CIAO_LOSS_ZERO ? nullptr : (const double *)p->b, p->ld, p->d, p->loss, (double)p->lam, i,
                           (const double *)x, (double *)y, (double *)fval);
    else
        hipLaunchKernelGGL((gradient_kernel<float>), dim3(1), dim3(WAVE), 0, ctx->stream, p->loss == CIAO_LOSS_ZERO ? nullptr : (const float *)p->A,
                           p->loss == CIAO_LOSS_ZERO ? nullptr : (const float *)p->b, p->ld, p->d, p->loss, (float)p->lam, i,
                           (const float *)x, (float *)y, (float *)fval);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

int32_t ciao_prox(ciao_ctx *ctx, int32_t dtype, int64_t d, const ciao_prox_desc *g, const void *x, double gamma, void *y)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    ctx->rowdot_A = nullptr;
    CIAO_REQUIRE(dtype == CIAO_F32 || dtype == CIAO_F64, "bad dtype %d", dtype);
    CIAO_REQUIRE(d >= 0 && (d == 0 || (x && y)), "bad d or NULL vector");
    CIAO_TRY(check_prox(g));
    CIAO_REQUIRE(!(g && g->kind == CIAO_PROX_L1_COMPLEX) || d % 2 == 0, "CIAO_PROX_L1_COMPLEX works on (re, im) pairs: d must be even");
    if (d == 0) return CIAO_OK;
    if (dtype == CIAO_F64) return prox_launch<double>(ctx, d, g, (const double *)x, gamma, 1.0, (double *)y);
    return prox_launch<float>(ctx, d, g, (const float *)x, (float)gamma, 1.0f, (float *)y);
}

int32_t ciao_full_gradient(ciao_ctx *ctx, const ciao_problem *p, const void *x, void *av)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(x && av, "x or av is NULL");
    return DISPATCH(p->dtype, full_gradient_t, ctx, p, x, av);
}

int32_t ciao_full_gradient_multi(ciao_ctx *ctx, const ciao_problem *p, int32_t K, const void *const *x, void *const *av)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(K >= 1 && K <= 4096 && x && av, "K must be in 1..4096 and the pointer tables non-NULL");
    for (int k = 0; k < K; ++k) CIAO_REQUIRE(x[k] && av[k], "x[%d] or av[%d] is NULL", k, k);
    CIAO_REQUIRE(p->loss != CIAO_LOSS_LS_COMPLEX, "complex problems: one ciao_full_gradient per iterate");
    return DISPATCH(p->dtype, full_gradient_multi_t, ctx, p, K, x, av);
}

int32_t ciao_proxgrad_step(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, const void *x,
                           void *av, void *y)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(x && av && y, "x, av or y is NULL");
    CIAO_REQUIRE(gamma > 0, "gamma must be > 0");
    return DISPATCH(p->dtype, proxgrad_t, ctx, p, g, gamma, x, av, y);
}

int32_t ciao_objective(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *x, double *obj_host)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(x && obj_host, "x or obj_host is NULL");
    return DISPATCH(p->dtype, objective_t, ctx, p, g, x, obj_host);
}

int32_t ciao_svrg_init(ciao_ctx *ctx, const ciao_problem *p, const void *x0, void *av, void *z, void *z_full, void *w)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(x0 && av && z && z_full && w, "NULL state vector");
    const size_t bytes = (size_t)p->d * (p->dtype == CIAO_F64 ? 8 : 4);
    CIAO_TRY(p->dtype == CIAO_F64 ? full_gradient_t<double>(ctx, p, x0, av, true) : full_gradient_t<float>(ctx, p, x0, av, true));
    if (ctx->rowdot_A) ctx->rowdot_x = z_full;   // z_full becomes a copy of x0: the cached a_i'x0 are a_i'z_full
    if (z_full != x0) CIAO_HIP(hipMemcpyAsync(z_full, x0, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (w != x0) CIAO_HIP(hipMemcpyAsync(w, x0, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    CIAO_HIP(hipMemsetAsync(z, 0, bytes, ctx->stream));
    return CIAO_OK;
}

int32_t ciao_svrg_inner(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int64_t m,
                        const int64_t *idx, const void *av, void *z, const void *z_full, void *w)
{
    CIAO_ENTER_BATCHABLE(ctx);
    CIAO_REQUIRE(!ctx->batch_open || (!ctx->hook && ctx->shards.nshards == 0), "a chain batch cannot run on a row-sharded problem");
    const void *keep = ctx->rowdot_A;
    CIAO_TRY(check_problem(ctx, p));
    ctx->rowdot_A = keep;   // z_full is read-only here: a following ciao_svrg_iterate may still reuse the row dots
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(m >= 0 && (m == 0 || idx), "m < 0 or idx is NULL");
    CIAO_REQUIRE(m == 0 || p->N > 0 || ctx->shards.nshards > 0, "cannot sample from an empty problem");
    CIAO_REQUIRE(av && z && z_full && w, "NULL state vector");
    CIAO_REQUIRE(gamma > 0, "gamma must be > 0");
    CIAO_REQUIRE(!ctx->hook || ctx->shards.nshards > 0,
                 "the SVRG inner cycle is a sequential chain: on a row-sharded problem it needs a shard table (ciao_ctx_set_shards)");
    CIAO_TRY(check_shards(ctx, p, false));
    return DISPATCH(p->dtype, svrg_inner_t, ctx, p, g, gamma, m, idx, av, z, z_full, w);
}

int32_t ciao_svrg_iterate(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int64_t m,
                          const int64_t *idx, int32_t plus, int32_t reuse_rowdots, void *av, void *z, void *z_full, void *w)
{
    CIAO_ENTER(ctx);
    const void *keep = ctx->rowdot_A;
    CIAO_TRY(check_problem(ctx, p));
    // the a_i'z_full of the last full pass are used only when the CALLER vouches that A and z_full still hold what that
    // pass read (reuse_rowdots) AND the previous call on this ctx was svrg_init / svrg_iterate / svrg_inner on these buffers
    ctx->rowdot_A = reuse_rowdots ? keep : nullptr;
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(m >= 1 && idx, "m < 1 or idx is NULL");
    CIAO_REQUIRE(p->N > 0 || ctx->shards.nshards > 0, "cannot sample from an empty problem");
    CIAO_REQUIRE(av && z && z_full && w, "NULL state vector");
    CIAO_REQUIRE(gamma > 0, "gamma must be > 0");
    CIAO_REQUIRE(!ctx->hook || ctx->shards.nshards > 0,
                 "the SVRG inner cycle is a sequential chain: on a row-sharded problem it needs a shard table (ciao_ctx_set_shards)");
    CIAO_TRY(check_shards(ctx, p, false));
    return DISPATCH(p->dtype, svrg_iterate_t, ctx, p, g, gamma, m, idx, plus, reuse_rowdots, av, z, z_full, w);
}

int32_t ciao_svrg_epoch_tail(ciao_ctx *ctx, const ciao_problem *p, int64_t m, int32_t plus, void *av, void *z, void *z_full, void *w)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(m >= 1, "m < 1");
    CIAO_REQUIRE(av && z && z_full && w, "NULL state vector");
    return DISPATCH(p->dtype, svrg_epoch_tail_t, ctx, p, m, plus, av, z, z_full, w);
}

int32_t ciao_svrg_epoch_tail_multi(ciao_ctx *ctx, const ciao_problem *p, int32_t K, int64_t m, int32_t plus, void *const *av,
                                   void *const *z, void *const *z_full, void *const *w)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(m >= 1, "m < 1");
    CIAO_REQUIRE(K >= 1 && K <= 4096 && av && z && z_full && w, "K must be in 1..4096 and the pointer tables non-NULL");
    for (int k = 0; k < K; ++k) CIAO_REQUIRE(av[k] && z[k] && z_full[k] && w[k], "NULL state vector of solve %d", k);
    CIAO_REQUIRE(p->loss != CIAO_LOSS_LS_COMPLEX, "complex problems: one ciao_svrg_epoch_tail per solve");
    return DISPATCH(p->dtype, svrg_epoch_tail_multi_t, ctx, p, K, m, plus, av, z, z_full, w);
}

int32_t ciao_saga_init(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, const void *x0,
                       void *table, void *av, void *z)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(x0 && av && z && (table || p->N == 0), "NULL state vector / table");
    CIAO_REQUIRE(gamma > 0, "gamma must be > 0");
    return DISPATCH(p->dtype, saga_init_t, ctx, p, g, gamma, x0, table, av, z);
}

int32_t ciao_saga_steps(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int32_t sag,
                        int64_t nsteps, const int64_t *idx, void *table, void *av, void *z)
{
    CIAO_ENTER_BATCHABLE(ctx);
    CIAO_REQUIRE(!ctx->batch_open || (!ctx->hook && ctx->shards.nshards == 0), "a chain batch cannot run on a row-sharded problem");
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(nsteps >= 0 && (nsteps == 0 || idx), "nsteps < 0 or idx is NULL");
    CIAO_REQUIRE(nsteps == 0 || p->N > 0 || ctx->shards.nshards > 0, "cannot sample from an empty problem");
    CIAO_REQUIRE((table || ctx->shards.nshards > 0) && av && z, "NULL state vector / table");
    CIAO_REQUIRE(gamma > 0, "gamma must be > 0");
    CIAO_REQUIRE(!ctx->hook || ctx->shards.nshards > 0,
                 "SAGA steps are a sequential chain: on a row-sharded problem they need a shard table (ciao_ctx_set_shards)");
    CIAO_TRY(check_shards(ctx, p, true));
    return DISPATCH(p->dtype, saga_steps_t, ctx, p, g, gamma, sag, nsteps, idx, table, av, z);
}

int32_t ciao_hat_gamma(ciao_ctx *ctx, int32_t dtype, int64_t N, const void *gam, double *hat_gamma_host)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx && hat_gamma_host, "ctx or output is NULL");
    CIAO_REQUIRE(dtype == CIAO_F32 || dtype == CIAO_F64, "bad dtype %d", dtype);
    CIAO_REQUIRE(N >= 0 && (N == 0 || gam), "N < 0 or gam is NULL");
    const int nb = 256;
    if (dtype == CIAO_F64)
        hipLaunchKernelGGL((invsum_kernel<double>), dim3(nb), dim3(256), 0, ctx->stream, N, (const double *)gam, ctx->scal);
    else
        hipLaunchKernelGGL((invsum_kernel<float>), dim3(nb), dim3(256), 0, ctx->stream, N, (const float *)gam, ctx->scal);
    CIAO_HIP(hipGetLastError());
    double part_buf[256];
    double *const part_data = part_buf;
    CIAO_HIP(hipMemcpyAsync(part_data, ctx->scal, nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CIAO_HIP(hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part_data[i];
    if (ctx->hook) {
        CIAO_HIP(hipMemcpyAsync(ctx->scal, &s, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        const int32_t hs = ctx->hook(ctx->hook_user, ctx->scal, 1, CIAO_F64, (void *)ctx->stream);
        if (hs != 0) {
            set_error("all-reduce hook failed with status %d", hs);
            return CIAO_ERR_HOOK;
        }
        CIAO_HIP(hipMemcpyAsync(&s, ctx->scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CIAO_HIP(hipStreamSynchronize(ctx->stream));
    }
    *hat_gamma_host = 1.0 / s;
    return CIAO_OK;
}

int32_t ciao_finito_init(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                         const void *x0, void *table, void *av, void *z)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(x0 && av && z && ((table && gam) || p->N == 0), "NULL state vector / table / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    return DISPATCH(p->dtype, finito_init_t, ctx, p, g, gam, hat_gamma, x0, table, av, z);
}

// validation of a call's row blocks: inside [0, N), non-negative lengths (empty only on a row-sharded problem)
static int32_t check_blocks(const ciao_ctx *ctx, int64_t N, int64_t n, const int64_t *first_host, const int64_t *len_host, const char *what)
{
    CIAO_REQUIRE(n >= 0 && (n == 0 || (first_host && len_host)), "%s: n < 0 or NULL block arrays", what);
    for (int64_t t = 0; t < n; ++t) {
        CIAO_REQUIRE(len_host[t] >= 1 || (ctx->hook && len_host[t] == 0), "%s: batch %lld is empty", what, (long long)t);
        CIAO_REQUIRE(first_host[t] >= 0 && first_host[t] + len_host[t] <= N, "%s: batch %lld = rows [%lld, %lld) leaves [0, %lld)", what,
                     (long long)t, (long long)first_host[t], (long long)(first_host[t] + len_host[t]), (long long)N);
    }
    return CIAO_OK;
}

int32_t ciao_finito_steps(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                          int64_t nit, const int64_t *bptr_host, const int64_t *bidx, void *table, void *av, void *z)
{
    CIAO_ENTER_BATCHABLE(ctx);
    CIAO_REQUIRE(!ctx->batch_open || (!ctx->hook && ctx->shards.nshards == 0), "a chain batch cannot run on a row-sharded problem");
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(nit >= 0 && (nit == 0 || (bptr_host && (bidx || bptr_host[nit] == bptr_host[0]))), "nit < 0 or NULL batch arrays");
    CIAO_REQUIRE(nit == 0 || p->N > 0, "cannot sample from an empty problem");
    CIAO_REQUIRE(table && av && z && gam, "NULL state vector / table / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    BatchSrc src;
    src.bptr = bptr_host;
    src.bidx = bidx;
    return DISPATCH(p->dtype, finito_steps_t, ctx, p, g, gam, hat_gamma, nit, src, table, av, z);
}

int32_t ciao_finito_steps_blocks(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                                 int64_t nit, const int64_t *first_host, const int64_t *len_host, void *table, void *av, void *z)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_TRY(check_blocks(ctx, p->N, nit, first_host, len_host, "ciao_finito_steps_blocks"));
    CIAO_REQUIRE(nit == 0 || p->N > 0 || ctx->hook, "cannot take batches from an empty problem");
    CIAO_REQUIRE(table && av && z && gam, "NULL state vector / table / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    BatchSrc src;
    src.first = first_host;
    src.len = len_host;
    return DISPATCH(p->dtype, finito_steps_t, ctx, p, g, gam, hat_gamma, nit, src, table, av, z);
}

int32_t ciao_lfinito_init(ciao_ctx *ctx, const ciao_problem *p, double hat_gamma, const void *x0, void *av, void *z,
                          void *z_full)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(x0 && av && z && z_full, "NULL state vector");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    return DISPATCH(p->dtype, lfinito_init_t, ctx, p, hat_gamma, x0, av, z, z_full);
}

int32_t ciao_lfinito_iterate(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                             int64_t nb, const int64_t *bptr_host, const int64_t *bidx, void *av, void *z, void *z_full)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(nb >= 0 && (nb == 0 || (bptr_host && (bidx || bptr_host[nb] == bptr_host[0]))), "nb < 0 or NULL batch arrays");
    CIAO_REQUIRE(av && z && z_full && gam, "NULL state vector / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    BatchSrc src;
    src.bptr = bptr_host;
    src.bidx = bidx;
    return DISPATCH(p->dtype, lfinito_iterate_t, ctx, p, g, gam, hat_gamma, nb, src, av, z, z_full);
}

int32_t ciao_lfinito_iterate_blocks(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                                    int64_t nb, const int64_t *first_host, const int64_t *len_host, void *av, void *z, void *z_full)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_TRY(check_blocks(ctx, p->N, nb, first_host, len_host, "ciao_lfinito_iterate_blocks"));
    CIAO_REQUIRE(av && z && z_full && gam, "NULL state vector / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    BatchSrc src;
    src.first = first_host;
    src.len = len_host;
    return DISPATCH(p->dtype, lfinito_iterate_t, ctx, p, g, gam, hat_gamma, nb, src, av, z, z_full);
}

int32_t ciao_afinito_probe(ciao_ctx *ctx, const ciao_problem *p, int64_t i, const void *x0, const void *signs, double t,
                           double *nmg_host)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_REQUIRE(x0 && signs && nmg_host && t > 0, "NULL argument or t <= 0");
    CIAO_REQUIRE(p->loss != CIAO_LOSS_ZERO && i >= 0 && i < p->N, "sample index %lld outside [0, %lld) or no data terms", (long long)i,
                 (long long)p->N);
    if (p->dtype == CIAO_F64)
        hipLaunchKernelGGL((afinito_probe_kernel<double>), dim3(1), dim3(WAVE), 0, ctx->stream, (const double *)p->A, (const double *)p->b,
                           p->ld, p->d, p->loss, (double)p->lam, i, (const double *)x0, (const double *)signs, t, ctx->scal);
    else
        hipLaunchKernelGGL((afinito_probe_kernel<float>), dim3(1), dim3(WAVE), 0, ctx->stream, (const float *)p->A, (const float *)p->b,
                           p->ld, p->d, p->loss, (float)p->lam, i, (const float *)x0, (const float *)signs, (float)t, ctx->scal);
    CIAO_HIP(hipGetLastError());
    CIAO_HIP(hipMemcpyAsync(nmg_host, ctx->scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CIAO_HIP(hipStreamSynchronize(ctx->stream));
    return CIAO_OK;
}

int32_t ciao_afinito_init(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double alpha, const void *x0,
                          void *table, void *meta, void *av, void *z, void *hat_gamma_dev, const void *gam_override)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(x0 && av && z && hat_gamma_dev && ((table && meta) || p->N == 0), "NULL state vector / table / meta");
    CIAO_REQUIRE(alpha > 0 && alpha < 1, "alpha must be in (0, 1)");
    CIAO_REQUIRE(p->loss != CIAO_LOSS_ZERO, "adaptive Finito needs data terms (the Lipschitz probe of Zero() is degenerate)");
    CIAO_REQUIRE(p->N >= 1 || ctx->shards.nshards > 0, "adaptive Finito needs at least one term");
    CIAO_REQUIRE(!ctx->hook || ctx->shards.nshards > 0,
                 "adaptive Finito is a sequential chain: on a row-sharded problem it needs a shard table (ciao_ctx_set_shards) with "
                 "the table and meta shards");
    CIAO_TRY(check_shards(ctx, p, false));
    return DISPATCH(p->dtype, afinito_init_t, ctx, p, g, alpha, x0, table, meta, av, z, hat_gamma_dev, gam_override);
}

int32_t ciao_afinito_steps(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double alpha, double tol_b,
                           int64_t nsteps, const int64_t *idx, void *table, void *meta, void *av, void *z, void *hat_gamma_dev,
                           int64_t *done_host, int64_t *trials_host)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_problem(ctx, p));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_pair(p, g));
    CIAO_REQUIRE(nsteps >= 0 && (nsteps == 0 || idx), "nsteps < 0 or idx is NULL");
    CIAO_REQUIRE(((table && meta) || ctx->shards.nshards > 0) && av && z && hat_gamma_dev, "NULL state vector / table / meta");
    CIAO_REQUIRE(alpha > 0 && alpha < 1 && tol_b > 0, "need 0 < alpha < 1 and tol_b > 0");
    CIAO_REQUIRE(p->loss != CIAO_LOSS_ZERO && (p->N >= 1 || ctx->shards.nshards > 0), "adaptive Finito needs data terms");
    CIAO_REQUIRE(!ctx->hook || ctx->shards.nshards > 0,
                 "adaptive Finito is a sequential chain: on a row-sharded problem it needs a shard table (ciao_ctx_set_shards) with "
                 "the table and meta shards");
    CIAO_TRY(check_shards(ctx, p, true));
    if (ctx->shards.nshards > 0 && ctx->shards.owner)
        for (int k = 0; k < ctx->shards.nshards; ++k)
            CIAO_REQUIRE(ctx->shards.row0[k + 1] == ctx->shards.row0[k] || ctx->shards.meta[k],
                         "shard %d has no meta pointer on the chain owner (adaptive Finito)", k);
    if (nsteps == 0) {
        if (done_host) *done_host = 0;
        if (trials_host) *trials_host = 0;
        return CIAO_OK;
    }
    return DISPATCH(p->dtype, afinito_steps_t, ctx, p, g, alpha, tol_b, nsteps, idx, table, meta, av, z, hat_gamma_dev, done_host,
                    trials_host);
}

int32_t ciao_proshi_init(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam, const void *x0,
                         void *table, void *av, void *z, void *hat_gamma_dev)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_sepquad(ctx, f));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_real_prox(g, "ciao_proshi_init"));
    CIAO_REQUIRE(x0 && av && z && hat_gamma_dev && ((table && gam) || f->N == 0), "NULL state vector / table / gam");
    return DISPATCH(f->dtype, proshi_init_t, ctx, f, g, gam, x0, table, av, z, hat_gamma_dev);
}

int32_t ciao_proshi_steps(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                          int64_t nit, const int64_t *bptr_host, const int64_t *bidx, void *table, void *av, void *z)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_sepquad(ctx, f));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_real_prox(g, "ciao_proshi_steps"));
    CIAO_REQUIRE(nit >= 0 && (nit == 0 || (bptr_host && (bidx || bptr_host[nit] == bptr_host[0]))), "nit < 0 or NULL batch arrays");
    CIAO_REQUIRE(table && gam && av && z, "NULL state vector / table / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    BatchSrc src;
    src.bptr = bptr_host;
    src.bidx = bidx;
    return DISPATCH(f->dtype, proshi_steps_t, ctx, f, g, gam, hat_gamma, nit, src, table, av, z);
}

int32_t ciao_proshi_steps_blocks(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam, double hat_gamma,
                                 int64_t nit, const int64_t *first_host, const int64_t *len_host, void *table, void *av, void *z)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_sepquad(ctx, f));
    CIAO_TRY(check_prox(g));
    CIAO_TRY(check_real_prox(g, "ciao_proshi_steps_blocks"));
    CIAO_TRY(check_blocks(ctx, f->N, nit, first_host, len_host, "ciao_proshi_steps_blocks"));
    CIAO_REQUIRE(table && gam && av && z, "NULL state vector / table / gam");
    CIAO_REQUIRE(hat_gamma > 0, "hat_gamma must be > 0");
    BatchSrc src;
    src.first = first_host;
    src.len = len_host;
    return DISPATCH(f->dtype, proshi_steps_t, ctx, f, g, gam, hat_gamma, nit, src, table, av, z);
}

int32_t ciao_proshi_solution(ciao_ctx *ctx, const ciao_sepquad *f, const void *gam, const void *z, void *table)
{
    CIAO_ENTER(ctx);
    CIAO_TRY(check_sepquad(ctx, f));
    CIAO_REQUIRE((table && gam && z) || f->N == 0, "NULL argument");
    if (f->N == 0) return CIAO_OK;
    int64_t grid = (f->N * f->d + 255) / 256;
    if (grid > (int64_t)ctx->num_cu * 16) grid = (int64_t)ctx->num_cu * 16;
    if (f->dtype == CIAO_F64)
        hipLaunchKernelGGL((proshi_solution_kernel<double>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, f->N, f->d,
                           (const double *)gam, (const double *)z, (double *)table);
    else
        hipLaunchKernelGGL((proshi_solution_kernel<float>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, f->N, f->d,
                           (const float *)gam, (const float *)z, (float *)table);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

int32_t ciao_synth_normal(ciao_ctx *ctx, int32_t dtype, void *out, int64_t nrows, int64_t d, int64_t ld, int64_t row0,
                          uint64_t seed, double scale)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx && (out || nrows == 0), "ctx or out is NULL");
    CIAO_REQUIRE(dtype == CIAO_F32 || dtype == CIAO_F64, "bad dtype %d", dtype);
    CIAO_REQUIRE(nrows >= 0 && d >= 1 && ld >= d, "bad shape");
    if (nrows == 0) return CIAO_OK;
    const int64_t total = nrows * d;
    int64_t grid = (total + 255) / 256;
    if (grid > (int64_t)ctx->num_cu * 32) grid = (int64_t)ctx->num_cu * 32;
    if (dtype == CIAO_F64)
        hipLaunchKernelGGL((synth_normal_kernel<double>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (double *)out, nrows, d,
                           ld, row0, seed, scale);
    else
        hipLaunchKernelGGL((synth_normal_kernel<float>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (float *)out, nrows, d,
                           ld, row0, seed, (float)scale);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

int32_t ciao_synth_targets(ciao_ctx *ctx, const ciao_problem *p, const void *x_true, double noise, int32_t labels,
                           int64_t row0, uint64_t seed, void *b_out)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx && p && x_true && (b_out || p->N == 0), "NULL argument");
    CIAO_REQUIRE(p->dtype == CIAO_F32 || p->dtype == CIAO_F64, "bad dtype");
    CIAO_REQUIRE(p->N >= 0 && p->d >= 1 && p->ld >= p->d && (p->A || p->N == 0), "bad problem shape");
    if (p->N == 0) return CIAO_OK;
    int64_t grid = (p->N + 3) / 4;
    if (grid > (int64_t)ctx->num_cu * 8) grid = (int64_t)ctx->num_cu * 8;
    if (p->dtype == CIAO_F64)
        hipLaunchKernelGGL((synth_targets_kernel<double>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const double *)p->A,
                           p->N, p->d, p->ld, (const double *)x_true, noise, (int)labels, row0, seed, (double *)b_out);
    else
        hipLaunchKernelGGL((synth_targets_kernel<float>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const float *)p->A,
                           p->N, p->d, p->ld, (const float *)x_true, (float)noise, (int)labels, row0, seed, (float *)b_out);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

// ---- host helper: n x sample(1:N, r, replace=false) from the splitmix64 index stream (sampling.py is its reference) ----
namespace {
inline uint64_t splitmix_at(uint64_t seed, uint64_t k)   // output number k (0-based) of the stream: counter-based
{
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

namespace {
__global__ void __launch_bounds__(256) sample_uniform_kernel(uint64_t seed, uint64_t pos, uint64_t N, int64_t m, int64_t *out)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < m; k += (int64_t)gridDim.x * 256) {
        uint64_t z = seed + (pos + (uint64_t)k + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[k] = (int64_t)(((z >> 32) * N) >> 32);
    }
}
}  // namespace

int32_t ciao_sample_uniform(ciao_ctx *ctx, uint64_t seed, uint64_t pos, int64_t N, int64_t m, int64_t *out_dev)
{
    CIAO_ENTER(ctx);
    CIAO_REQUIRE(ctx, "ctx is NULL");
    CIAO_REQUIRE(N > 0 && N < (1ll << 32) && m >= 0 && (out_dev || m == 0), "need 0 < N < 2^32, m >= 0 and an output array");
    if (m == 0) return CIAO_OK;
    const int64_t grid = std::min<int64_t>((m + 255) / 256, (int64_t)ctx->num_cu * 8);
    hipLaunchKernelGGL(sample_uniform_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, seed, pos, (uint64_t)N, m, out_dev);
    CIAO_HIP(hipGetLastError());
    return CIAO_OK;
}

int32_t ciao_sample_batches(uint64_t seed, uint64_t pos, int64_t N, int64_t r, int64_t n, int64_t *out_host,
                            uint64_t *pos_out_host)
{
    CIAO_REQUIRE(N > 0 && N < (1ll << 32) && r >= 1 && 2 * r <= N && n >= 0, "need 0 < N < 2^32, r >= 1, 2r <= N, n >= 0");
    CIAO_REQUIRE((out_host || n == 0) && pos_out_host, "NULL output");
    // open-addressing set of the indices held by the current batch: one 8-byte word per slot = (batch tag << 32) | index, so
    // a probe touches one cache line and nothing is cleared between batches (a stale tag reads as an empty slot)
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * r)) cap <<= 1;
    std::vector<uint64_t> slots;
    try {
        slots.assign(cap, 0);
    } catch (...) {   // nothing may unwind through the C ABI
        set_error("out of host memory (%llu hash slots)", (unsigned long long)cap);
        return CIAO_ERR_ALLOC;
    }
    uint32_t tag = 0;
    for (int64_t t = 0; t < n; ++t) {
        if (++tag == 0) {   // 2^32 batches later the tags would repeat: start over with clean slots
            std::fill(slots.begin(), slots.end(), 0);
            tag = 1;
        }
        int64_t *out = out_host + t * r;
        int64_t have = 0;
        while (have < r) {
            const int64_t need = r - have;   // one round: as many candidates as are still missing (>= 1)
            for (int64_t c = 0; c < need; ++c) {
                const uint64_t u = splitmix_at(seed, pos++) >> 32;
                const uint64_t v = (u * (uint64_t)N) >> 32;   // uniform in 0..N-1, the rule of IndexStream.rand_indices
                const uint64_t word = ((uint64_t)tag << 32) | v;
                uint64_t h = ((v * 0x9E3779B97F4A7C15ull) >> 24) & (cap - 1);
                bool seen = false;
                while ((slots[h] >> 32) == tag) {
                    if (slots[h] == word) {
                        seen = true;
                        break;
                    }
                    h = (h + 1) & (cap - 1);
                }
                if (!seen) {
                    slots[h] = word;
                    out[have++] = (int64_t)v;
                }
            }
        }
    }
    *pos_out_host = pos;
    return CIAO_OK;
}

}  // extern "C"
