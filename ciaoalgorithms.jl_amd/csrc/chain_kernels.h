// chain_kernels.h -- the strictly sequential inner loops (SURVEY.md section 8a rows S3, G3, and F3/F4 with small
// batches) as ONE persistent 256-thread workgroup.
//
// Why one workgroup: every step reads the iterate the previous step wrote (SVRG_basic.jl:75,80; SAGA_basic.jl:56,64),
// so the chain is latency-bound, not bandwidth-bound.  A cross-CU hand-off costs microseconds on this chip (per-XCD
// L2s are not coherent; MI355X_MICROARCH.md "handoff" rows), a workgroup barrier costs tens of cycles, so the whole
// d-vector state lives in the registers of one workgroup: thread t owns elements t, t+256, ...  Per step:
//   row a_i (prefetched DEPTH steps ahead into registers -- all indices are known up front, SVRG_basic.jl:73),
//   per-thread partial dot -> DPP wave sum -> 4 partials through LDS (ONE barrier per step, double-buffered slots),
//   scalar link function, element-wise update + prox in the reference's own operation order.
// SAGA/Finito table rows are prefetched the same way; a row that an intervening step rewrites is detected by
// comparing indices at prefetch time and re-read at use time (same thread wrote it: program order).
#pragma once

#include "ciao_common.h"

namespace ciao {

enum ChainAlg { CA_SVRG = 0, CA_SAGA = 1, CA_FINITO = 2, CA_LFINITO = 3 };

template <typename T>
struct ChainArgs {
    const T *A;
    const T *b;
    int64_t ld, d;
    int loss;
    T lam;
    int64_t nsteps;        // number of samples in the flattened sequence
    const int64_t *idx;    // their rows
    int64_t batch;         // FINITO / LFINITO: prox every `batch` samples
    T gamma;               // SVRG / SAGA stepsize
    int sag;
    T invN;                // 1 / N_total
    const T *gam;          // FINITO / LFINITO per-sample stepsizes (nullptr -> gam_uniform)
    T gam_uniform, hat_gamma;
    T *table;
    ProxD<T> g;
    T *av, *z, *zf, *w;
    int64_t N;             // local rows (index validation)
    int *errflag;          // device word set to 1 on an out-of-range index
};

constexpr int CHAIN_NT = 256;
constexpr int CHAIN_NW = CHAIN_NT / WAVE;

template <int E>
struct ChainDepth {
    static constexpr int value = E <= 4 ? 8 : (E <= 8 ? 4 : 2);
};

template <typename T, int E, int ALG>
__global__ void __launch_bounds__(CHAIN_NT) chain_kernel(ChainArgs<T> a)
{
    constexpr int DEPTH = ChainDepth<E>::value;
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO);

    __shared__ T red[2][CHAIN_NW][2];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;

    bool valid[E];
    int64_t eidx[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        eidx[j] = tid + (int64_t)j * CHAIN_NT;
        valid[j] = eidx[j] < d;
    }

    // iterate state in registers:  p = the point the "moving" gradient is taken at (w for SVRG, z otherwise)
    T av[E], p[E], zf[E], zs[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        av[j] = valid[j] ? a.av[eidx[j]] : T(0);
        if (ALG == CA_SVRG) {
            p[j] = valid[j] ? a.w[eidx[j]] : T(0);
            zs[j] = valid[j] ? a.z[eidx[j]] : T(0);
        } else {
            p[j] = valid[j] ? a.z[eidx[j]] : T(0);
            zs[j] = T(0);
        }
        zf[j] = (TWO && valid[j]) ? a.zf[eidx[j]] : T(0);
    }

    // prefetch rings (statically indexed through full unrolling)
    T ar[DEPTH][E], sr[DEPTH][E];
    int64_t ring_row[DEPTH];
    T ring_b[DEPTH], ring_g[DEPTH];
    bool ring_stale[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
        ring_row[u] = -1;
        ring_b[u] = T(0);
        ring_g[u] = T(1);
        ring_stale[u] = false;
    }

    auto issue = [&](int u, int64_t step) {
        if (step < a.nsteps) {
            int64_t r = a.idx[step];
            if ((uint64_t)r >= (uint64_t)a.N) {   // memory-safe: flag it, use row 0 (results are void once flagged)
                if (tid == 0) *a.errflag = 1;
                r = 0;
            }
            bool stale = false;
            if (HAS_TABLE) {
#pragma unroll
                for (int u2 = 0; u2 < DEPTH; ++u2) stale |= (ring_row[u2] == r);
            }
            ring_row[u] = r;
            ring_b[u] = a.b ? a.b[r] : T(0);
            if (PER_SAMPLE_GAM) ring_g[u] = a.gam ? a.gam[r] : a.gam_uniform;
            ring_stale[u] = stale;
            const T *ap = a.A + r * a.ld;
#pragma unroll
            for (int j = 0; j < E; ++j) ar[u][j] = (valid[j] && a.A) ? ap[eidx[j]] : T(0);
            if (HAS_TABLE && !stale) {
                const T *sp = a.table + r * d;
#pragma unroll
                for (int j = 0; j < E; ++j) sr[u][j] = valid[j] ? sp[eidx[j]] : T(0);
            }
        }
    };

#pragma unroll
    for (int u = 0; u < DEPTH; ++u) issue(u, u);

    int par = 0;
    int64_t inb = 0;   // position of the current sample inside its batch (FINITO / LFINITO)
    for (int64_t base = 0; base < a.nsteps; base += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int64_t step = base + u;
            if (step >= a.nsteps) break;
            const int64_t row = ring_row[u];
            const T bi = ring_b[u];

            if (ALG == CA_LFINITO && inb == 0) {   // Finito_LFinito.jl:92  z = prox(av)
#pragma unroll
                for (int j = 0; j < E; ++j) p[j] = valid[j] ? prox_elem(a.g, av[j], a.hat_gamma, eidx[j]) : T(0);
            }
            if (HAS_TABLE && ring_stale[u]) {   // the row was rewritten after its prefetch slot was claimed
                const T *sp = a.table + row * d;
#pragma unroll
                for (int j = 0; j < E; ++j) sr[u][j] = valid[j] ? sp[eidx[j]] : T(0);
            }

            // block-wide dot products: a_i'p and (TWO) a_i'z_full
            T d1 = T(0), d2 = T(0);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                d1 += ar[u][j] * p[j];
                if (TWO) d2 += ar[u][j] * zf[j];
            }
            d1 = wave_allsum(d1);
            if (TWO) d2 = wave_allsum(d2);
            if (lane == 0) {
                red[par][wib][0] = d1;
                if (TWO) red[par][wib][1] = d2;
            }
            __syncthreads();
            d1 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
            if (TWO) d2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
            par ^= 1;

            const GradCoef<T> gp = grad_coef(a.loss, d1, bi, a.lam);
            if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                const GradCoef<T> gz = grad_coef(a.loss, d2, bi, a.lam);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    T t = gz.elem(ar[u][j]) - gp.elem(ar[u][j]);
                    t -= av[j];
                    t *= a.gamma;
                    t += p[j];
                    p[j] = valid[j] ? prox_elem(a.g, t, a.gamma, eidx[j]) : T(0);
                    zs[j] += p[j];
                }
            } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                T *sp = a.table + row * d;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const T gn = gp.elem(ar[u][j]);
                    const T del = (gn - sr[u][j]) * a.invN;
                    T wv;
                    if (a.sag) {
                        av[j] += del;
                        wv = p[j] - a.gamma * av[j];
                    } else {
                        wv = p[j] - a.gamma * (gn - sr[u][j] + av[j]);
                        av[j] += del;
                    }
                    p[j] = valid[j] ? prox_elem(a.g, wv, a.gamma, eidx[j]) : T(0);
                    if (valid[j]) sp[eidx[j]] = gn;
                }
            } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                const T gi = ring_g[u];
                const T cg = gi * a.invN;
                const T rr = a.hat_gamma / gi;
                T *sp = a.table + row * d;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const T t = p[j] - cg * gp.elem(ar[u][j]);
                    av[j] += (t - sr[u][j]) * rr;
                    if (valid[j]) sp[eidx[j]] = t;
                }
                if (inb + 1 == a.batch || step + 1 == a.nsteps) {
#pragma unroll
                    for (int j = 0; j < E; ++j) p[j] = valid[j] ? prox_elem(a.g, av[j], a.hat_gamma, eidx[j]) : T(0);
                }
            } else {                                                         // Finito_LFinito.jl:93-98
                const GradCoef<T> gzf = grad_coef(a.loss, d2, bi, a.lam);
                const T gi = ring_g[u];
                const T c = a.hat_gamma * a.invN;
                const T rr = a.hat_gamma / gi;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    av[j] += c * gzf.elem(ar[u][j]);
                    av[j] -= c * gp.elem(ar[u][j]);
                    av[j] += rr * (p[j] - zf[j]);
                }
            }

            if (++inb == a.batch) inb = 0;
            issue(u, step + DEPTH);   // refill this slot (after this step's table store: program order)
        }
    }

#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j]) continue;
        if (ALG == CA_SVRG) {
            a.w[eidx[j]] = p[j];
            a.z[eidx[j]] = zs[j];
        } else {
            a.z[eidx[j]] = p[j];
            a.av[eidx[j]] = av[j];
        }
    }
}

// single-sample gradient!(y, f_i, x) -- the L1 plugin call itself (one wave).
template <typename T>
__global__ void __launch_bounds__(WAVE)
    gradient_kernel(const T *A, const T *b, int64_t ld, int64_t d, int loss, T lam, int64_t i, const T *x, T *y, T *fval)
{
    const int lane = threadIdx.x;
    const T *ap = A ? A + i * ld : nullptr;
    T dot = T(0);
    for (int64_t e = lane; e < d; e += WAVE) dot += (ap ? ap[e] : T(0)) * x[e];
    dot = wave_allsum(dot);
    const T bi = b ? b[i] : T(0);
    const GradCoef<T> g = grad_coef(loss, dot, bi, lam);
    for (int64_t e = lane; e < d; e += WAVE) y[e] = g.elem(ap ? ap[e] : T(0));
    if (fval && lane == 0) *fval = loss_value(loss, dot, bi, lam);
}

// elementwise prox!(y, g, x, gamma)  and the two small vector helpers the epoch tails need
template <typename T>
__global__ void __launch_bounds__(256) prox_kernel(int64_t d, ProxD<T> g, const T *x, T gamma, T scale, T *y)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < d) y[k] = prox_elem(g, scale * x[k], gamma, k);
}

// SVRG epoch tail (SVRG_basic.jl:84-86): z_full = z/m ; basic: w = z_full ; z = 0
template <typename T>
__global__ void __launch_bounds__(256) svrg_tail_kernel(int64_t d, T m, int plus, T *z, T *z_full, T *w)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < d) {
        const T zf = z[k] / m;
        z_full[k] = zf;
        if (!plus) w[k] = zf;
        z[k] = T(0);
    }
}

// g(x) = lam*||x||_1 partial sums are tiny: one block
template <typename T>
__global__ void __launch_bounds__(256) gvalue_kernel(int64_t d, ProxD<T> g, const T *x, double *out)
{
    __shared__ double s[256];
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < d; k += 256) acc += (double)prox_value_elem(g, x[k]);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s[0];
}

// sum_i 1/gam_i  (two-pass deterministic): per-block partials, summed by finalize on the host side of the call
template <typename T>
__global__ void __launch_bounds__(256) invsum_kernel(int64_t n, const T *gam, double *partial)
{
    __shared__ double s[256];
    double acc = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        acc += 1.0 / (double)gam[k];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

}  // namespace ciao
