// chain_reg_kernels.h -- the chains whose rows travel through registers: chain_kernel (the register-ring fallback for rows that are not
// whole aligned 16-byte chunks), chain_big_kernel (rows of any length, state in the caller's vectors), chain_cplx_kernel / chain_cplx_reg_kernel
// (complex T).  Split out of chain_kernels.h in round 5; see chain_common.h for the design.
#pragma once

#include "chain_common.h"

namespace ciao {

// NT = 256 (four waves, thread t owns elements t + 256 j) or 64: rows of up to 512 elements on ONE wave -- the reduced dot
// product reaches every lane through an SGPR and the LDS exchange + barrier of every step disappears (as in chain_dma_kernel).
template <typename T, int E, int ALG, int LOSS, bool FULL, int NT = CHAIN_NT>
__global__ void __launch_bounds__(NT) chain_kernel(ChainArgs<T> a)
{
    constexpr int NW = NT / WAVE;
    static_assert(NW == 1 || NW == CHAIN_NW, "one wave or four");
    constexpr int DEPTH = ChainDepth<E>::value;
    constexpr int CH = CHAIN_CHUNK;
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO);
    static_assert(CH % DEPTH == 0, "ring slots must line up with chunk starts");

    __shared__ T red[2][NW][2];
    // per-chunk staging of everything that is gathered by sample index: rows (with DEPTH entries of history in front
    // and DEPTH entries of look-ahead behind), b_i, gamma_i and the table-row hazard flags
    __shared__ int64_t s_row[CH + 2 * DEPTH];
    __shared__ T s_b[CH];
    __shared__ T s_g[PER_SAMPLE_GAM ? CH : 1];
    __shared__ int s_stale[HAS_TABLE ? CH : 1];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;

    bool valid[E];
    int64_t eidx[E], ecl[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        eidx[j] = tid + (int64_t)j * NT;
        valid[j] = FULL || eidx[j] < d;
        ecl[j] = valid[j] ? eidx[j] : d - 1;   // clamped: loads stay unconditional and in bounds
    }

    // iterate state in registers:  p = the point the "moving" gradient is taken at (w for SVRG, z otherwise)
    T av[E], p[E], zf[E], zs[E], plo[E], phi[E];
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        av[j] = valid[j] ? a.av[ecl[j]] : T(0);
        if (ALG == CA_SVRG) {
            p[j] = valid[j] ? a.w[ecl[j]] : T(0);
            zs[j] = valid[j] ? a.z[ecl[j]] : T(0);
        } else {
            p[j] = valid[j] ? a.z[ecl[j]] : T(0);
            zs[j] = T(0);
        }
        zf[j] = (TWO && valid[j]) ? a.zf[ecl[j]] : T(0);
        plo[j] = -INFINITY;
        phi[j] = INFINITY;
        if (a.g.kind == CIAO_PROX_BOX) {
            plo[j] = a.g.lo_vec ? a.g.lo_vec[ecl[j]] : a.g.lo;
            phi[j] = a.g.hi_vec ? a.g.hi_vec[ecl[j]] : a.g.hi;
        }
    }

    // register prefetch rings (statically indexed through full unrolling)
    T ar[DEPTH][E], sr[DEPTH][E];

    // all loads of the ring refill are unconditional and straight-line, so that the compiler can retire them with
    // counted s_waitcnt vmcnt(N) instead of draining the queue every step
    auto refill = [&](int u, int64_t r) {
        const T *ap = a.A + r * a.ld;   // never null here: Zero() terms alias a finite d-vector with ld = 0 (see launch)
#pragma unroll
        for (int j = 0; j < E; ++j) ar[u][j] = ap[ecl[j]];
        if (HAS_TABLE) {
            const T *sp = a.table + r * d;
#pragma unroll
            for (int j = 0; j < E; ++j) sr[u][j] = sp[ecl[j]];
        }
    };

    int par = 0;
    int64_t inb = 0;   // position of the current sample inside its batch (FINITO / LFINITO)
    for (int64_t base = 0; base < a.nsteps; base += CH) {
        const int nch = (int)((a.nsteps - base) < CH ? (a.nsteps - base) : CH);

        // ---- stage this chunk's gathers in LDS --------------------------------------------------------------------------
        __syncthreads();   // the previous chunk is fully consumed
        int64_t hist = -1;
        if (tid < DEPTH && base > 0) hist = s_row[CH + tid];   // last DEPTH rows of the previous (full) chunk
        __syncthreads();
        if (tid < DEPTH) s_row[tid] = hist;
        for (int e = tid; e < nch + DEPTH; e += NT) {
            int64_t st = base + e;
            if (st > a.nsteps - 1) st = a.nsteps - 1;   // look-ahead past the end repeats the last row (harmless loads)
            int64_t r = a.idx[st];
            if ((uint64_t)r >= (uint64_t)a.N) {   // memory-safe: flag it, use row 0 (results are void once flagged)
                *a.errflag = 1;
                r = 0;
            }
            s_row[DEPTH + e] = r;
            if (e < nch) {
                s_b[e] = a.b ? a.b[r] : T(0);
                if (PER_SAMPLE_GAM) s_g[e] = a.gam ? a.gam[r] : a.gam_uniform;
            }
        }
        __syncthreads();
        if (HAS_TABLE) {
            for (int e = tid; e < nch; e += NT) {
                const int64_t r = s_row[DEPTH + e];
                bool st = false;
#pragma unroll
                for (int k = 1; k <= DEPTH; ++k) st |= (s_row[DEPTH + e - k] == r);
                s_stale[e] = st ? 1 : 0;
            }
            __syncthreads();
        }
        if (base == 0) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) refill(u, uniform64(s_row[DEPTH + u]));
        }

        // ---- the dependent chain ----------------------------------------------------------------------------------------
        for (int s0 = 0; s0 < nch; s0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int s = s0 + u;
                if (s >= nch) break;
                const int64_t row = uniform64(s_row[DEPTH + s]);
                const int64_t row_n = uniform64(s_row[DEPTH + s + DEPTH]);
                const T bi = s_b[s];

                if (ALG == CA_LFINITO && inb == 0) {   // Finito_LFinito.jl:92  z = prox(av)
#pragma unroll
                    for (int j = 0; j < E; ++j) p[j] = valid[j] ? prox_bf(av[j], a.hat_gamma * plam, plo[j], phi[j]) : T(0);
                }
                if (HAS_TABLE && __builtin_amdgcn_readfirstlane(s_stale[s])) {
                    // an intervening step rewrote this table row after it was prefetched: re-read it (the same thread
                    // wrote these very elements, so program order makes the new values visible)
                    const T *sp = a.table + row * d;
#pragma unroll
                    for (int j = 0; j < E; ++j) sr[u][j] = sp[ecl[j]];
                }

                // block-wide dot products: a_i'p and (TWO) a_i'z_full
                T d1 = T(0), d2 = T(0);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const T aj = (FULL || valid[j]) ? ar[u][j] : T(0);
                    d1 += aj * p[j];
                    if (TWO) d2 += aj * zf[j];
                }
                d1 = wave_sum_lane63(d1);
                if (TWO) d2 = wave_sum_lane63(d2);
                if constexpr (NW == 1) {
                    d1 = readlane(d1, WAVE - 1);
                    if (TWO) d2 = readlane(d2, WAVE - 1);
                } else {
                    if (lane == WAVE - 1) {   // the lane that holds the wave's sum
                        red[par][wib][0] = d1;
                        if (TWO) red[par][wib][1] = d2;
                    }
                    __syncthreads();
                    d1 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
                    if (TWO) d2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
                    par ^= 1;
                }

                const GradCoef<T> gp = grad_coef_t<T, LOSS>(d1, bi, a.lam);
                if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                    const GradCoef<T> gz = grad_coef_t<T, LOSS>(d2, bi, a.lam);
                    const T gl = a.gamma * plam;
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        T t = gz.elem(ar[u][j]) - gp.elem(ar[u][j]);
                        t -= av[j];
                        t *= a.gamma;
                        t += p[j];
                        p[j] = valid[j] ? prox_bf(t, gl, plo[j], phi[j]) : T(0);
                        zs[j] += p[j];
                    }
                } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                    T *sp = a.table + row * d;
                    const T gl = a.gamma * plam;
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
                        p[j] = valid[j] ? prox_bf(wv, gl, plo[j], phi[j]) : T(0);
                        if (FULL || valid[j]) sp[eidx[j]] = gn;
                        if (!FULL && !valid[j]) av[j] = T(0);
                    }
                } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                    const T gi = s_g[s];
                    const T cg = gi * a.invN;
                    const T rr = a.hat_gamma / gi;
                    T *sp = a.table + row * d;
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        const T t = p[j] - cg * gp.elem(ar[u][j]);
                        av[j] += (t - sr[u][j]) * rr;
                        if (FULL || valid[j]) sp[eidx[j]] = t;
                        if (!FULL && !valid[j]) av[j] = T(0);
                    }
                    if (inb + 1 == a.batch || (base + s + 1) == a.nsteps) {
                        const T gl = a.hat_gamma * plam;
#pragma unroll
                        for (int j = 0; j < E; ++j) p[j] = valid[j] ? prox_bf(av[j], gl, plo[j], phi[j]) : T(0);
                    }
                } else {                                                         // Finito_LFinito.jl:93-98
                    const GradCoef<T> gzf = grad_coef_t<T, LOSS>(d2, bi, a.lam);
                    const T gi = s_g[s];
                    const T c = a.hat_gamma * a.invN;
                    const T rr = a.hat_gamma / gi;
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        av[j] += c * gzf.elem(ar[u][j]);
                        av[j] -= c * gp.elem(ar[u][j]);
                        av[j] += rr * (p[j] - zf[j]);
                        if (!FULL && !valid[j]) av[j] = T(0);
                    }
                }

                if (++inb == a.batch) inb = 0;
                refill(u, row_n);   // after this step's table store (program order); look-ahead entry always exists
            }
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

// ------------------------------------------------------------------------------------------------------------------
// Chains on rows of ANY length (d beyond 8192, where the per-thread register state of the kernels above no longer fits):
// one 1024-thread workgroup, the iterate state stays in the caller's d-vectors (L2-resident: a few hundred KiB), every step
// is two passes over the row -- dot product(s), then the element-wise update -- with one block-wide reduction in between.
// The arithmetic is the reference's own operation order (as chain_kernel).  Bandwidth of one CU bounds it: a step moves
// about 8 d-vectors through one L1 (measured: d = 16384 fp64, 128 KiB rows: a few microseconds per step) -- the point of
// this kernel is that the sequential solvers exist for every d, not speed.
// ------------------------------------------------------------------------------------------------------------------
constexpr int CHAIN_BIG_NT = 1024;

template <typename T, int ALG, int LOSS>
__global__ void __launch_bounds__(CHAIN_BIG_NT) chain_big_kernel(ChainArgs<T> a)
{
    constexpr int NW = CHAIN_BIG_NT / WAVE;
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    __shared__ T red[2][NW][2];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;
    T *p = (ALG == CA_SVRG) ? a.w : a.z;      // the point the moving gradient is taken at
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
    auto box = [&](int64_t k, T &lo, T &hi) {
        lo = -INFINITY;
        hi = INFINITY;
        if (a.g.kind == CIAO_PROX_BOX) {
            lo = a.g.lo_vec ? a.g.lo_vec[k] : a.g.lo;
            hi = a.g.hi_vec ? a.g.hi_vec[k] : a.g.hi;
        }
    };
    int par = 0;
    int64_t inb = 0;
    for (int64_t s = 0; s < a.nsteps; ++s) {
        int64_t row = a.idx[s];
        if ((uint64_t)row >= (uint64_t)a.N) {   // memory-safe: flag it, use row 0 (results are void once flagged)
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        const T *ap = a.A + row * a.ld;          // Zero() terms alias a finite d-vector with ld = 0 and lam = 0 (see launch)
        const T bi = a.b ? a.b[row] : T(0);
        T *sp = HAS_TABLE ? a.table + row * d : nullptr;
        if (ALG == CA_LFINITO && inb == 0) {     // Finito_LFinito.jl:92  z = prox(av)
            for (int64_t k = tid; k < d; k += CHAIN_BIG_NT) {
                T lo, hi;
                box(k, lo, hi);
                p[k] = prox_bf(a.av[k], a.hat_gamma * plam, lo, hi);
            }
            __syncthreads();
        }
        T d1 = T(0), d2 = T(0);
        for (int64_t k = tid; k < d; k += CHAIN_BIG_NT) {
            const T ak = ap[k];
            d1 += ak * p[k];
            if (TWO) d2 += ak * a.zf[k];
        }
        d1 = wave_sum_lane63(d1);
        if (TWO) d2 = wave_sum_lane63(d2);
        if (lane == WAVE - 1) {   // the lane that holds the wave's sum
            red[par][wib][0] = d1;
            if (TWO) red[par][wib][1] = d2;
        }
        __syncthreads();
        d1 = T(0);
        d2 = T(0);
#pragma unroll
        for (int w = 0; w < NW; w += 4) {        // fixed association order: groups of four
            d1 += (red[par][w][0] + red[par][w + 1][0]) + (red[par][w + 2][0] + red[par][w + 3][0]);
            if (TWO) d2 += (red[par][w][1] + red[par][w + 1][1]) + (red[par][w + 2][1] + red[par][w + 3][1]);
        }
        par ^= 1;
        const GradCoef<T> gp = grad_coef_t<T, LOSS>(d1, bi, a.lam);
        const GradCoef<T> gz = grad_coef_t<T, LOSS>(d2, bi, a.lam);
        const T gi = (ALG == CA_FINITO || ALG == CA_LFINITO) ? (a.gam ? a.gam[row] : a.gam_uniform) : T(1);
        const bool last_of_batch = (inb + 1 == a.batch) || (s + 1 == a.nsteps);
        for (int64_t k = tid; k < d; k += CHAIN_BIG_NT) {
            const T ak = ap[k];
            T lo, hi;
            box(k, lo, hi);
            if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                T t = gz.elem(ak) - gp.elem(ak);
                t -= a.av[k];
                t *= a.gamma;
                t += p[k];
                const T wn = prox_bf(t, a.gamma * plam, lo, hi);
                p[k] = wn;
                a.z[k] += wn;
            } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                const T gn = gp.elem(ak);
                const T sk = sp[k];
                const T del = (gn - sk) * a.invN;
                T avk = a.av[k], wv;
                if (a.sag) {
                    avk += del;
                    wv = p[k] - a.gamma * avk;
                } else {
                    wv = p[k] - a.gamma * (gn - sk + avk);
                    avk += del;
                }
                a.av[k] = avk;
                p[k] = prox_bf(wv, a.gamma * plam, lo, hi);
                sp[k] = gn;
            } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                const T t = p[k] - (gi * a.invN) * gp.elem(ak);
                const T avk = a.av[k] + (t - sp[k]) * (a.hat_gamma / gi);
                a.av[k] = avk;
                sp[k] = t;
                if (last_of_batch) p[k] = prox_bf(avk, a.hat_gamma * plam, lo, hi);
            } else {                                                         // Finito_LFinito.jl:93-98
                const T c = a.hat_gamma * a.invN;
                T avk = a.av[k];
                avk += c * gz.elem(ak);
                avk -= c * gp.elem(ak);
                avk += (a.hat_gamma / gi) * (p[k] - a.zf[k]);
                a.av[k] = avk;
            }
        }
        if (++inb == a.batch) inb = 0;
        __syncthreads();   // the next step's dot products read what this step wrote (same workgroup: one CU, one L1)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The chains for complex T (CIAO_LOSS_LS_COMPLEX; vectors are (re, im) pairs): chain_big_kernel's structure -- one
// 1024-thread workgroup, state in the caller's vectors, two passes over the row per step -- with the complex residual
// res = a_i . p - b_i, grad = (conj(a_k) res) lam, and the prox of g = Zero or complex NormL1 pair by pair.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int ALG>
__global__ void __launch_bounds__(CHAIN_BIG_NT) chain_cplx_kernel(ChainArgs<T> a)
{
    constexpr int NW = CHAIN_BIG_NT / WAVE;
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    __shared__ T red[2][NW][4];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d, dc = a.d / 2;
    T *p = (ALG == CA_SVRG) ? a.w : a.z;
    const bool l1 = (a.g.kind == CIAO_PROX_L1_COMPLEX);
    auto proxc = [&](T tau, T vr, T vi, T &yr, T &yi) {
        if (l1) {
            prox_cpair_chain(tau * a.g.lam, vr, vi, yr, yi);
        } else {
            yr = vr;
            yi = vi;
        }
    };
    int par = 0;
    int64_t inb = 0;
    for (int64_t s = 0; s < a.nsteps; ++s) {
        int64_t row = a.idx[s];
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        const T *ap = a.A + row * a.ld;
        const T br = a.b[2 * row], bi = a.b[2 * row + 1];
        T *sp = HAS_TABLE ? a.table + row * d : nullptr;
        if (ALG == CA_LFINITO && inb == 0) {     // Finito_LFinito.jl:92  z = prox(av)
            for (int64_t e = tid; e < dc; e += CHAIN_BIG_NT) proxc(a.hat_gamma, a.av[2 * e], a.av[2 * e + 1], p[2 * e], p[2 * e + 1]);
            __syncthreads();
        }
        T s1r = T(0), s1i = T(0), s2r = T(0), s2i = T(0);
        for (int64_t e = tid; e < dc; e += CHAIN_BIG_NT) {
            const T ar = ap[2 * e], ai = ap[2 * e + 1];
            const T xr = p[2 * e], xi = p[2 * e + 1];
            s1r += ar * xr - ai * xi;
            s1i += ar * xi + ai * xr;
            if (TWO) {
                const T yr = a.zf[2 * e], yi = a.zf[2 * e + 1];
                s2r += ar * yr - ai * yi;
                s2i += ar * yi + ai * yr;
            }
        }
        s1r = wave_sum_lane63(s1r);
        s1i = wave_sum_lane63(s1i);
        if (TWO) {
            s2r = wave_sum_lane63(s2r);
            s2i = wave_sum_lane63(s2i);
        }
        if (lane == WAVE - 1) {   // the lane that holds the wave's sum
            red[par][wib][0] = s1r;
            red[par][wib][1] = s1i;
            red[par][wib][2] = s2r;
            red[par][wib][3] = s2i;
        }
        __syncthreads();
        T t4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int w = 0; w < NW; w += 4)
                t4[c] += (red[par][w][c] + red[par][w + 1][c]) + (red[par][w + 2][c] + red[par][w + 3][c]);
        par ^= 1;
        const T rpr = t4[0] - br, rpi = t4[1] - bi;      // residual at p
        const T rzr = t4[2] - br, rzi = t4[3] - bi;      // residual at z_full (TWO)
        const T gi = (ALG == CA_FINITO || ALG == CA_LFINITO) ? (a.gam ? a.gam[row] : a.gam_uniform) : T(1);
        const bool last_of_batch = (inb + 1 == a.batch) || (s + 1 == a.nsteps);
        for (int64_t e = tid; e < dc; e += CHAIN_BIG_NT) {
            const int64_t k = 2 * e;
            const T ar = ap[k], ai = ap[k + 1];
            T gpr, gpi, gzr, gzi;
            cgrad_elem(ar, ai, rpr, rpi, a.lam, gpr, gpi);
            cgrad_elem(ar, ai, rzr, rzi, a.lam, gzr, gzi);
            if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                T tr = gzr - gpr, ti = gzi - gpi;
                tr -= a.av[k];
                ti -= a.av[k + 1];
                tr *= a.gamma;
                ti *= a.gamma;
                tr += p[k];
                ti += p[k + 1];
                T wr, wi;
                proxc(a.gamma, tr, ti, wr, wi);
                p[k] = wr;
                p[k + 1] = wi;
                a.z[k] += wr;
                a.z[k + 1] += wi;
            } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                const T sr = sp[k], si = sp[k + 1];
                const T delr = (gpr - sr) * a.invN, deli = (gpi - si) * a.invN;
                T avr = a.av[k], avi = a.av[k + 1], wr, wi;
                if (a.sag) {
                    avr += delr;
                    avi += deli;
                    wr = p[k] - a.gamma * avr;
                    wi = p[k + 1] - a.gamma * avi;
                } else {
                    wr = p[k] - a.gamma * (gpr - sr + avr);
                    wi = p[k + 1] - a.gamma * (gpi - si + avi);
                    avr += delr;
                    avi += deli;
                }
                a.av[k] = avr;
                a.av[k + 1] = avi;
                proxc(a.gamma, wr, wi, p[k], p[k + 1]);
                sp[k] = gpr;
                sp[k + 1] = gpi;
            } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                const T tr = p[k] - (gi * a.invN) * gpr, ti = p[k + 1] - (gi * a.invN) * gpi;
                const T avr = a.av[k] + (tr - sp[k]) * (a.hat_gamma / gi);
                const T avi = a.av[k + 1] + (ti - sp[k + 1]) * (a.hat_gamma / gi);
                a.av[k] = avr;
                a.av[k + 1] = avi;
                sp[k] = tr;
                sp[k + 1] = ti;
                if (last_of_batch) proxc(a.hat_gamma, avr, avi, p[k], p[k + 1]);
            } else {                                                         // Finito_LFinito.jl:93-98
                const T c = a.hat_gamma * a.invN;
                T avr = a.av[k], avi = a.av[k + 1];
                avr += c * gzr;
                avi += c * gzi;
                avr -= c * gpr;
                avi -= c * gpi;
                avr += (a.hat_gamma / gi) * (p[k] - a.zf[k]);
                avi += (a.hat_gamma / gi) * (p[k + 1] - a.zf[k + 1]);
                a.av[k] = avr;
                a.av[k + 1] = avi;
            }
        }
        if (++inb == a.batch) inb = 0;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Complex chains, register-resident: up to 2048 complex entries per row (EP pairs per thread, 256 threads).  The iterate
// state (p, av, z_full / the SVRG accumulator) lives in registers for the whole launch, thread t owning the pairs
// t + 256 j; the next step's row (and table row) is requested one step ahead and is in flight while this step computes; one
// raw barrier per step for the 4-wave exchange of the complex dot product(s).  Formulas as in chain_cplx_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int ALG, int EP>
__global__ void __launch_bounds__(CHAIN_NT) chain_cplx_reg_kernel(ChainArgs<T> a)
{
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    __shared__ T red[2][CHAIN_NW][4];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d, dc = a.d / 2;
    T *pmem = (ALG == CA_SVRG) ? a.w : a.z;
    const bool l1 = (a.g.kind == CIAO_PROX_L1_COMPLEX);
    auto proxc = [&](T tau, T vr, T vi, T &yr, T &yi) {
        if (l1) {
            prox_cpair_chain(tau * a.g.lam, vr, vi, yr, yi);
        } else {
            yr = vr;
            yi = vi;
        }
    };
    bool ok[EP];
    int64_t ke[EP];                                   // offset of the pair's real part; dead pairs point at pair 0 and are masked
    T pr[EP], pi[EP], avr[EP], avi[EP], qr[EP], qi[EP];   // q: z_full (SVRG, LFinito) ; the SVRG accumulator z rides in zr/zi
    T zr[EP], zi[EP];
#pragma unroll
    for (int j = 0; j < EP; ++j) {
        const int64_t e = tid + (int64_t)j * CHAIN_NT;
        ok[j] = e < dc;
        ke[j] = ok[j] ? 2 * e : 0;
        pr[j] = ok[j] ? pmem[ke[j]] : T(0);
        pi[j] = ok[j] ? pmem[ke[j] + 1] : T(0);
        avr[j] = ok[j] ? a.av[ke[j]] : T(0);
        avi[j] = ok[j] ? a.av[ke[j] + 1] : T(0);
        qr[j] = (TWO && ok[j]) ? a.zf[ke[j]] : T(0);
        qi[j] = (TWO && ok[j]) ? a.zf[ke[j] + 1] : T(0);
        zr[j] = (ALG == CA_SVRG && ok[j]) ? a.z[ke[j]] : T(0);
        zi[j] = (ALG == CA_SVRG && ok[j]) ? a.z[ke[j] + 1] : T(0);
    }
    auto row_of = [&](int64_t s) -> int64_t {
        int64_t r = a.idx[s];
        if ((uint64_t)r >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            r = 0;
        }
        return r;
    };
    auto load = [&](int64_t r, T(&xr)[EP], T(&xi)[EP], T(&tr)[EP], T(&ti)[EP], T &br, T &bi, T &gi) {
        const T *ap = a.A + r * a.ld;
        const T *sp = HAS_TABLE ? a.table + r * d : nullptr;
#pragma unroll
        for (int j = 0; j < EP; ++j) {
            xr[j] = ap[ke[j]];
            xi[j] = ap[ke[j] + 1];
            if (HAS_TABLE) {
                tr[j] = sp[ke[j]];
                ti[j] = sp[ke[j] + 1];
            }
        }
        br = a.b[2 * r];
        bi = a.b[2 * r + 1];
        gi = (ALG == CA_FINITO || ALG == CA_LFINITO) ? (a.gam ? a.gam[r] : a.gam_uniform) : T(1);
    };
    T ar[EP], ai[EP], sr[EP], si[EP], br = T(0), bi = T(0), gi = T(1);
    T arn[EP], ain[EP], srn[EP], sin_[EP], brn = T(0), bin = T(0), gin = T(1);
    int64_t row = 0, rown = 0;
    if (a.nsteps > 0) {
        row = row_of(0);
        load(row, ar, ai, sr, si, br, bi, gi);
    }
    int par = 0;
    int64_t inb = 0;
    for (int64_t s = 0; s < a.nsteps; ++s) {
        const bool more = s + 1 < a.nsteps;
        bool same = false;
        if (more) {
            rown = row_of(s + 1);
            same = (rown == row);
            if (!same) load(rown, arn, ain, srn, sin_, brn, bin, gin);   // in flight while this step computes
        }
        if (ALG == CA_LFINITO && inb == 0) {     // Finito_LFinito.jl:92  z = prox(av)
#pragma unroll
            for (int j = 0; j < EP; ++j) proxc(a.hat_gamma, avr[j], avi[j], pr[j], pi[j]);
        }
        T s1r = T(0), s1i = T(0), s2r = T(0), s2i = T(0);
#pragma unroll
        for (int j = 0; j < EP; ++j) {
            const T xr = ok[j] ? ar[j] : T(0), xi = ok[j] ? ai[j] : T(0);
            s1r += xr * pr[j] - xi * pi[j];
            s1i += xr * pi[j] + xi * pr[j];
            if (TWO) {
                s2r += xr * qr[j] - xi * qi[j];
                s2i += xr * qi[j] + xi * qr[j];
            }
        }
        s1r = wave_sum_lane63(s1r);
        s1i = wave_sum_lane63(s1i);
        if (TWO) {
            s2r = wave_sum_lane63(s2r);
            s2i = wave_sum_lane63(s2i);
        }
        if (lane == WAVE - 1) {   // the lane that holds the wave's sum
            red[par][wib][0] = s1r;
            red[par][wib][1] = s1i;
            if (TWO) {
                red[par][wib][2] = s2r;
                red[par][wib][3] = s2i;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // raw barrier: the next row's loads stay in flight across it
        const T t0 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
        const T t1 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
        T t2 = T(0), t3 = T(0);
        if (TWO) {
            t2 = (red[par][0][2] + red[par][1][2]) + (red[par][2][2] + red[par][3][2]);
            t3 = (red[par][0][3] + red[par][1][3]) + (red[par][2][3] + red[par][3][3]);
        }
        par ^= 1;
        const T rpr = t0 - br, rpi = t1 - bi;      // residual at p
        const T rzr = t2 - br, rzi = t3 - bi;      // residual at z_full (TWO)
        const bool last_of_batch = (inb + 1 == a.batch) || (s + 1 == a.nsteps);
        T *sp = HAS_TABLE ? a.table + row * d : nullptr;
#pragma unroll
        for (int j = 0; j < EP; ++j) {
            if (!ok[j]) continue;
            T gpr, gpi, gzr, gzi;
            cgrad_elem(ar[j], ai[j], rpr, rpi, a.lam, gpr, gpi);
            cgrad_elem(ar[j], ai[j], rzr, rzi, a.lam, gzr, gzi);
            if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                T tr = gzr - gpr, ti = gzi - gpi;
                tr -= avr[j];
                ti -= avi[j];
                tr *= a.gamma;
                ti *= a.gamma;
                tr += pr[j];
                ti += pi[j];
                proxc(a.gamma, tr, ti, pr[j], pi[j]);
                zr[j] += pr[j];
                zi[j] += pi[j];
            } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                const T delr = (gpr - sr[j]) * a.invN, deli = (gpi - si[j]) * a.invN;
                T wr, wi;
                if (a.sag) {
                    avr[j] += delr;
                    avi[j] += deli;
                    wr = pr[j] - a.gamma * avr[j];
                    wi = pi[j] - a.gamma * avi[j];
                } else {
                    wr = pr[j] - a.gamma * (gpr - sr[j] + avr[j]);
                    wi = pi[j] - a.gamma * (gpi - si[j] + avi[j]);
                    avr[j] += delr;
                    avi[j] += deli;
                }
                proxc(a.gamma, wr, wi, pr[j], pi[j]);
                sr[j] = gpr;                                                 // the row's new table entry (kept for `same`)
                si[j] = gpi;
                sp[ke[j]] = gpr;
                sp[ke[j] + 1] = gpi;
            } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                const T tr = pr[j] - (gi * a.invN) * gpr, ti = pi[j] - (gi * a.invN) * gpi;
                avr[j] += (tr - sr[j]) * (a.hat_gamma / gi);
                avi[j] += (ti - si[j]) * (a.hat_gamma / gi);
                sr[j] = tr;
                si[j] = ti;
                sp[ke[j]] = tr;
                sp[ke[j] + 1] = ti;
                if (last_of_batch) proxc(a.hat_gamma, avr[j], avi[j], pr[j], pi[j]);
            } else {                                                         // Finito_LFinito.jl:93-98
                const T c = a.hat_gamma * a.invN;
                avr[j] += c * gzr;
                avi[j] += c * gzi;
                avr[j] -= c * gpr;
                avi[j] -= c * gpi;
                avr[j] += (a.hat_gamma / gi) * (pr[j] - qr[j]);
                avi[j] += (a.hat_gamma / gi) * (pi[j] - qi[j]);
            }
        }
        if (++inb == a.batch) inb = 0;
        if (more && !same) {
#pragma unroll
            for (int j = 0; j < EP; ++j) {
                ar[j] = arn[j];
                ai[j] = ain[j];
                if (HAS_TABLE) {
                    sr[j] = srn[j];
                    si[j] = sin_[j];
                }
            }
            br = brn;
            bi = bin;
            gi = gin;
            row = rown;
        }
    }
#pragma unroll
    for (int j = 0; j < EP; ++j) {
        if (!ok[j]) continue;
        pmem[ke[j]] = pr[j];
        pmem[ke[j] + 1] = pi[j];
        a.av[ke[j]] = avr[j];
        a.av[ke[j] + 1] = avi[j];
        if (ALG == CA_SVRG) {
            a.z[ke[j]] = zr[j];
            a.z[ke[j] + 1] = zi[j];
        }
    }
}

}  // namespace ciao
