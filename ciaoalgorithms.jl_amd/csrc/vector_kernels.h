// vector_kernels.h -- the small one-launch kernels: gradient!(y, f_i, x) on one wave, the Lipschitz probe, prox, the SVRG epoch tail,
// g's value, the sum of 1/gamma_i.  Split out of chain_kernels.h in round 5.
#pragma once

#include "chain_common.h"

namespace ciao {

// single-sample gradient!(y, f_i, x) -- the L1 plugin call itself (one wave).
template <typename T>
__global__ void __launch_bounds__(WAVE)
    gradient_kernel(const T *A, const T *b, int64_t ld, int64_t d, int loss, T lam, int64_t i, const T *x, T *y, T *fval)
{
    const int lane = threadIdx.x;
    const T *ap = A ? A + i * ld : nullptr;
    if (loss == CIAO_LOSS_LS_COMPLEX) {   // (re, im) pairs: res = a.x - b, y_k = (conj(a_k) res) lam, f = lam/2 |res|^2
        T sr = T(0), si = T(0);
        for (int64_t e = lane; e < d / 2; e += WAVE) {
            sr += ap[2 * e] * x[2 * e] - ap[2 * e + 1] * x[2 * e + 1];
            si += ap[2 * e] * x[2 * e + 1] + ap[2 * e + 1] * x[2 * e];
        }
        sr = wave_allsum(sr) - b[2 * i];
        si = wave_allsum(si) - b[2 * i + 1];
        for (int64_t e = lane; e < d / 2; e += WAVE) cgrad_elem(ap[2 * e], ap[2 * e + 1], sr, si, lam, y[2 * e], y[2 * e + 1]);
        if (fval && lane == 0) *fval = (lam / T(2)) * (sr * sr + si * si);
        return;
    }
    T dot = T(0);
    for (int64_t e = lane; e < d; e += WAVE) dot += (ap ? ap[e] : T(0)) * x[e];
    dot = wave_allsum(dot);
    const T bi = b ? b[i] : T(0);
    const GradCoef<T> g = grad_coef(loss, dot, bi, lam);
    for (int64_t e = lane; e < d; e += WAVE) y[e] = g.elem(ap ? ap[e] : T(0));
    if (fval && lane == 0) *fval = loss_value(loss, dot, bi, lam);
}

// elementwise prox!(y, g, x, gamma)  and the two small vector helpers the epoch tails need
// One retry of adaptive Finito's Lipschitz probe for sample i (Finito_adaptive.jl:80-82): the probe point is x0 + t*signs
// (signs = the host's +-1 draws), both gradients are multiples of a_i, so
//   nmg = || grad f_i(x0 + t signs) - grad f_i(x0) || = | c(a_i'x0 + t a_i'signs) - c(a_i'x0) | * ||a_i||       (in R; one wave)
template <typename T>
__global__ void __launch_bounds__(WAVE)
    afinito_probe_kernel(const T *A, const T *b, int64_t ld, int64_t d, int loss, T lam, int64_t i, const T *x0, const T *signs, T t, double *out)
{
    const int lane = threadIdx.x;
    const T *ap = A + i * ld;
    if (loss == CIAO_LOSS_LS_COMPLEX) {   // complex T: `signs` has d/2 REAL entries, added to the real parts (rand(t*[-1,1], size(x0)))
        T sr = T(0), si = T(0), m2 = T(0);
        for (int64_t e = lane; e < d / 2; e += WAVE) {
            const T ar = ap[2 * e], ai = ap[2 * e + 1];
            sr += ar * signs[e];
            si += ai * signs[e];
            m2 += ar * ar + ai * ai;
        }
        sr = wave_allsum(sr);
        si = wave_allsum(si);
        m2 = wave_allsum(m2);
        if (lane == 0) *out = (double)(fhypot(lam * (t * sr), lam * (t * si)) * fsqrt(m2));   // |c1 - c0| ||a_i||, c1 - c0 = lam t a.signs
        return;
    }
    T d0 = T(0), ds = T(0), n2 = T(0);
    for (int64_t k = lane; k < d; k += WAVE) {
        const T ak = ap[k];
        d0 += ak * x0[k];
        ds += ak * signs[k];
        n2 += ak * ak;
    }
    d0 = wave_allsum(d0);
    ds = wave_allsum(ds);
    n2 = wave_allsum(n2);
    const T bi = b[i];
    const T c0 = grad_coef(loss, d0, bi, lam).coef();
    const T c1 = grad_coef(loss, d0 + t * ds, bi, lam).coef();
    if (lane == 0) *out = (double)(fabs2(c1 - c0) * fsqrt(n2));
}

template <typename T>
__global__ void __launch_bounds__(256) prox_kernel(int64_t d, ProxD<T> g, const T *x, T gamma, T scale, T *y)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g.kind == CIAO_PROX_L1_COMPLEX) {   // (re, im) pairs: thread k takes coordinates 2k and 2k+1 (d is even)
        if (2 * k + 1 < d) prox_cpair(gamma * g.lam, scale * x[2 * k], scale * x[2 * k + 1], y[2 * k], y[2 * k + 1]);
        return;
    }
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
__global__ void __launch_bounds__(256) gvalue_kernel(int64_t d, ProxD<T> g, const T *x, double *out, double *obj)
{
    __shared__ double s[256];
    double acc = 0.0;
    if (g.kind == CIAO_PROX_L1_COMPLEX) {   // lam * sum of complex moduli
        for (int64_t k = threadIdx.x; 2 * k + 1 < d; k += 256) acc += (double)(g.lam * fhypot(x[2 * k], x[2 * k + 1]));
    } else {
        for (int64_t k = threadIdx.x; k < d; k += 256) acc += (double)prox_value_elem(g, x[k]);
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *out = s[0];
        if (obj) obj[0] = obj[1] + s[0];   // monitor: F = (1/N) sum f_i (left in obj[1] by the sweep's epilogue) + g
    }
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
