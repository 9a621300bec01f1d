// afinito_kernels.h -- adaptive Finito (Finito_adaptive.jl:59-152): afinito_chain_kernel (compiler-scheduled fallback), afinito_big_kernel
// (any length, complex), afinito_dma_kernel (LDS-DMA).  Split out of chain_kernels.h in round 5.
#pragma once

#include "chain_common.h"
#include "chain_reg_kernels.h"
#include "chain_dma_kernels.h"

namespace ciao {

// ------------------------------------------------------------------------------------------------------------------
// Adaptive Finito steps (Finito_adaptive.jl:118-150): one sample per iteration with a
// data-dependent backtracking loop on that sample's stepsize.  Same one-workgroup, state-in-registers structure as the
// chains above; the per-sample scalars live in `meta` ({c_i with grad f_i = c_i a_i, f_i(x_i), gamma_i, a_i'x_i}, kept in
// FOUR identical copies per sample, N x 4 x 4: wave w of the workgroup writes and reads only copy w, so every read of a
// scalar follows its last write in the SAME wave's program order and needs neither a barrier nor a drained memory queue),
// so the reference's N x d gradient table collapses to N scalars for these row-structured f_i.  The next sample's
// row, table row and scalars are loaded one step ahead (re-read when it is the sample being updated).  Every trial of
// the backtracking needs a_i'z and ||z - x_i||^2: one 2-value exchange per trial.  All branches are workgroup-uniform
// because every thread derives them from the same bitwise-identical reduced scalars.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
struct AFinitoArgs {
    const T *A;
    const T *b;
    int64_t ld, d, N;
    T lam;
    int64_t nsteps;
    const int64_t *idx;
    T alpha, tol_b, invN, Nf;
    double Nd;             // N_total as the reference uses it in `0.5 * iter.N * iter.α / γ` (Float64 whatever R, :128)
    ProxD<T> g;
    T *table, *meta, *av, *z;
    T *hg;                // device scalar: hat_gamma (in/out)
    long long *counters;  // [0] steps completed, [1] backtracking trials (out)
    int *errflag;
    // Row-sharded problem (ciao_ctx_set_shards, as ChainArgs): shard k = global rows [sh_row0[k], sh_row0[k+1]) with its data rows,
    // its rows of the s-table and its per-sample scalars in allocations of their own (possibly another GPU's); idx holds GLOBAL rows.
    int nshards;
    const T *shA[CIAO_MAX_SHARDS];
    const T *shb[CIAO_MAX_SHARDS];
    T *shT[CIAO_MAX_SHARDS];
    T *shM[CIAO_MAX_SHARDS];
    int64_t sh_row0[CIAO_MAX_SHARDS + 1];
};

// The shard table of an adaptive Finito chain (41 qwords: shA | shb | shT | shM | sh_row0), to LDS and searched there exactly as the
// chains' (shard_table_to_lds / shard_resolve above, and why).
constexpr int AF_SHARD_QW = 5 * CIAO_MAX_SHARDS + 1;
template <typename T>
struct AFShardRow {
    const T *arow;
    const T *bp;
    T *trow;
    T *mrow;   // the sample's 4 x 4 scalars
};
template <typename T>
__device__ __forceinline__ void af_shard_table_to_lds(int64_t *s_sh, int tid)
{
    static_assert(offsetof(AFinitoArgs<T>, shb) == offsetof(AFinitoArgs<T>, shA) + 8 * CIAO_MAX_SHARDS &&
                  offsetof(AFinitoArgs<T>, shT) == offsetof(AFinitoArgs<T>, shA) + 16 * CIAO_MAX_SHARDS &&
                  offsetof(AFinitoArgs<T>, shM) == offsetof(AFinitoArgs<T>, shA) + 24 * CIAO_MAX_SHARDS &&
                  offsetof(AFinitoArgs<T>, sh_row0) == offsetof(AFinitoArgs<T>, shA) + 32 * CIAO_MAX_SHARDS, "the table is 41 contiguous qwords");
    const unsigned char __attribute__((address_space(4))) *ka =
        (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    if (tid < AF_SHARD_QW) s_sh[tid] = reinterpret_cast<const int64_t __attribute__((address_space(4))) *>(ka + offsetof(AFinitoArgs<T>, shA))[tid];
}
template <typename T>
__device__ __forceinline__ AFShardRow<T> af_shard_resolve(const int64_t *s_sh, int nshards, int64_t r, int64_t ld, int64_t d)
{
    const int64_t *row0 = s_sh + 4 * CIAO_MAX_SHARDS;
    int k = 0;
#pragma unroll
    for (int j = 1; j < CIAO_MAX_SHARDS; ++j) k += (j < nshards && r >= row0[j]) ? 1 : 0;
    const int64_t local = r - row0[k];
    auto glob = [](int64_t q) { return (T *)(__attribute__((address_space(1))) T *)(uintptr_t)q; };
    AFShardRow<T> o;
    o.arow = glob(s_sh[k]) + local * ld;
    o.bp = s_sh[CIAO_MAX_SHARDS + k] ? glob(s_sh[CIAO_MAX_SHARDS + k]) + local : nullptr;
    o.trow = glob(s_sh[2 * CIAO_MAX_SHARDS + k]) + local * d;
    o.mrow = glob(s_sh[3 * CIAO_MAX_SHARDS + k]) + local * (CHAIN_NW * 4);
    return o;
}

template <typename T, int E, int LOSS>
__global__ void __launch_bounds__(CHAIN_NT) afinito_chain_kernel(AFinitoArgs<T> a)
{
    __shared__ T red[2][CHAIN_NW][2];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;

    bool valid[E];
    int64_t ecl[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = tid + (int64_t)j * CHAIN_NT;
        valid[j] = e < d;
        ecl[j] = valid[j] ? e : d - 1;
    }
    T av[E], z[E], plo[E], phi[E];
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        av[j] = valid[j] ? a.av[ecl[j]] : T(0);
        z[j] = valid[j] ? a.z[ecl[j]] : T(0);
        plo[j] = -INFINITY;
        phi[j] = INFINITY;
        if (a.g.kind == CIAO_PROX_BOX) {
            plo[j] = a.g.lo_vec ? a.g.lo_vec[ecl[j]] : a.g.lo;
            phi[j] = a.g.hi_vec ? a.g.hi_vec[ecl[j]] : a.g.hi;
        }
    }
    T hg = *a.hg;
    int par = 0;

    auto row_of = [&](int64_t s) -> int64_t {
        int64_t r = a.idx[s];
        if ((uint64_t)r >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            r = 0;
        }
        return r;
    };
    auto load = [&](int64_t r, T(&ar)[E], T(&sr)[E], T(&m)[4], T &bi) {
        const T *ap = a.A + r * a.ld;
        const T *sp = a.table + r * d;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            ar[j] = ap[ecl[j]];
            sr[j] = sp[ecl[j]];
        }
        // this wave's own copy of the per-sample scalars (written by this wave's lane 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = a.meta[(r * CHAIN_NW + wib) * 4 + q];
        bi = a.b ? a.b[r] : T(0);
    };

    T ar[E], sr[E], m[4], bi = T(0);
    T arn[E], srn[E], mn[4], bin = T(0);
    int64_t row = 0, rown = 0;
    // the sample updated by the step that has just finished, and the scalars it stored: a prefetch issued right after
    // that store (no barrier in between) must not read them back from memory -- other waves may run ahead of thread 0
    int64_t row_prev = -1;
    T m_prev[4] = {T(0), T(0), T(0), T(0)};
    if (a.nsteps > 0) {
        row = row_of(0);
        load(row, ar, sr, m, bi);
    }
    int64_t done = 0, trials = 0;
    for (int64_t s = 0; s < a.nsteps; ++s) {
        const bool more = s + 1 < a.nsteps;
        bool same = false;
        if (more) {
            rown = row_of(s + 1);
            same = (rown == row);
            if (!same) {
                load(rown, arn, srn, mn, bin);   // in flight while this step computes
                if (rown == row_prev) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) mn[q] = m_prev[q];
                }
            }
        }
        const T c_old = m[0], fi_x = m[1], as_i = m[3];
        T gi = m[2];
        T res[E];
#pragma unroll
        for (int j = 0; j < E; ++j) res[j] = valid[j] ? z[j] - sr[j] : T(0);
        T dz = T(0), fi_z = T(0);
        bool stop = false;
        while (true) {
            if (gi < a.tol_b * a.invN) {          // Finito_adaptive.jl:121-124: the stepsize collapsed
                stop = true;
                break;
            }
            ++trials;
            T p1 = T(0), p2 = T(0);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                p1 = fmad(valid[j] ? ar[j] : T(0), z[j], p1);
                p2 = fmad(res[j], res[j], p2);
            }
            p1 = wave_sum_lane63(p1);
            p2 = wave_sum_lane63(p2);
            if (lane == WAVE - 1) {   // the lane that holds the wave's sum
                red[par][wib][0] = p1;
                red[par][wib][1] = p2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // raw barrier: the next sample's loads stay in flight across it
            dz = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
            const T n2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
            par ^= 1;
            fi_z = loss_value(LOSS, dz, bi, a.lam);                                     // :125
            // Julia's promotions, which matter for R = Float32: `0.5 * iter.N * iter.α / γ` is Float64 (the literal 0.5), so the
            // model value and the comparison are Float64; `γ *= 0.8` multiplies in Float64 and rounds back to R.
            const double fi_model = (double)(fi_x + c_old * (dz - as_i)) + (0.5 * a.Nd * (double)a.alpha / (double)gi) * (double)n2;   // :126-129
            const T tol = T(10) * Eps<T>::value * (T(1) + fabs2(fi_z));                 // :130
            if ((double)fi_z <= fi_model + (double)tol) break;                          // :131
            const T gb = gi;                                                            // :133
            gi = (T)((double)gi * 0.8);                                                 // :134
            const T hg_old = hg;
            hg = T(1) / (T(1) / hg_old + T(1) / gi - T(1) / gb);                        // :139
            const T gl = hg * plam;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                T t = av[j] / hg_old;                                                   // :136
                t += sr[j] / gi;                                                        // :137
                t -= sr[j] / gb;                                                        // :138
                t *= hg;                                                                // :140
                av[j] = valid[j] ? t : T(0);
                z[j] = valid[j] ? prox_bf(av[j], gl, plo[j], phi[j]) : T(0);            // :141
                res[j] = valid[j] ? z[j] - sr[j] : T(0);                                // :142
            }
        }
        if (stop) break;
        // the main step, :145-150
        const GradCoef<T> gn = grad_coef_t<T, LOSS>(dz, bi, a.lam);
        const T c_new = gn.coef();
        const T r1 = hg / gi;
        const T r2 = (hg * a.invN) * (c_old - c_new);   // + (hg/N) grad_old - (hg/N) grad_new, both multiples of a_i
        const T gl = hg * plam;
        T *sp = a.table + row * d;
        T znew[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            znew[j] = z[j];
            if (valid[j]) sp[ecl[j]] = z[j];                                             // :146  s_i = z
            T t = fmad(r1, res[j], av[j]);                                               // :145
            t = fmad(r2, ar[j], t);                                                      // :147, :149
            av[j] = valid[j] ? t : T(0);
            z[j] = valid[j] ? prox_bf(av[j], gl, plo[j], phi[j]) : T(0);                // :150
        }
        if (lane == 0) {
            T *mp = a.meta + (row * CHAIN_NW + wib) * 4;
            mp[0] = c_new;
            mp[1] = fi_z;                                                                // :148 fi_x[i] = f_i(z)
            mp[2] = gi;
            mp[3] = dz;
        }
        ++done;
        row_prev = row;
        m_prev[0] = c_new;
        m_prev[1] = fi_z;
        m_prev[2] = gi;
        m_prev[3] = dz;
        if (more) {
            if (same) {
                // the next step works on the sample just updated: its row stays, its table row is the z stored above and
                // its scalars are the ones just computed (no memory round trip, and no cross-thread visibility question)
#pragma unroll
                for (int j = 0; j < E; ++j) sr[j] = znew[j];
                m[0] = c_new;
                m[1] = fi_z;
                m[2] = gi;
                m[3] = dz;
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    ar[j] = arn[j];
                    sr[j] = srn[j];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = mn[q];
                bi = bin;
                row = rown;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j]) continue;
        a.av[ecl[j]] = av[j];
        a.z[ecl[j]] = z[j];
    }
    if (tid == 0) {
        *a.hg = hg;
        a.counters[0] = done;
        a.counters[1] = trials;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Adaptive Finito on rows of ANY length, real or complex (CPLX: (re, im) pairs, CIAO_LOSS_LS_COMPLEX with g = Zero or the
// complex NormL1).  Same step as afinito_chain_kernel, same structure as chain_big_kernel: one 1024-thread workgroup, the
// state (av, z) and the table row stay in the caller's vectors (L2-resident), thread t owns coordinates t, t+1024, ... in
// every loop, so the only cross-thread traffic is the reduction of each trial (a.z, ||z - s_i||^2) and the per-sample
// scalars, which thread 0 stores and everybody reads after the barrier that opens the next step.
// Complex scalars: c = lam res and a.x_i are complex, the model's linear term is Re(conj(c) (a.z - a.x_i)); the meta slots
// are laid out as rows_cplx_kernel<AFINITO_INIT> leaves them (copies 0/2 real parts, 1/3 imaginary parts).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, bool CPLX>
__global__ void __launch_bounds__(CHAIN_BIG_NT) afinito_big_kernel(AFinitoArgs<T> a, int loss)
{
    constexpr int NW = CHAIN_BIG_NT / WAVE;
    __shared__ T red[2][NW][4];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;
    const int64_t units = CPLX ? d / 2 : d;        // coordinates a thread steps through: complex entries or reals
    const bool l1c = (a.g.kind == CIAO_PROX_L1_COMPLEX);
    // z = prox_{tau g}(av) for the thread's unit e
    auto prox_unit = [&](int64_t e, T tau) {
        if (CPLX) {
            const T vr = a.av[2 * e], vi = a.av[2 * e + 1];
            if (l1c) {
                prox_cpair(tau * a.g.lam, vr, vi, a.z[2 * e], a.z[2 * e + 1]);
            } else {
                a.z[2 * e] = vr;
                a.z[2 * e + 1] = vi;
            }
        } else {
            a.z[e] = prox_elem(a.g, a.av[e], tau, e);
        }
    };
    T hg = *a.hg;
    int par = 0;
    long long done = 0, trials = 0;
    for (int64_t s = 0; s < a.nsteps; ++s) {
        int64_t row = a.idx[s];
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        const T *ap = a.A + row * a.ld;
        T *sp = a.table + row * d;
        const T br = CPLX ? a.b[2 * row] : (a.b ? a.b[row] : T(0));
        const T bi = CPLX ? a.b[2 * row + 1] : T(0);
        __syncthreads();                                  // the scalars the previous step stored are visible
        const T *mp = a.meta + row * 16;
        const T c_or = mp[0], fi_x = mp[1], as_r = mp[3];
        const T c_oi = CPLX ? mp[4] : T(0), as_i = CPLX ? mp[7] : T(0);
        T gi = mp[2];
        T dzr = T(0), dzi = T(0), fi_z = T(0);
        bool stop = false;
        while (true) {
            if (gi < a.tol_b * a.invN) {                  // Finito_adaptive.jl:121-124: the stepsize collapsed
                stop = true;
                break;
            }
            ++trials;
            T p1r = T(0), p1i = T(0), p2 = T(0);
            for (int64_t e = tid; e < units; e += CHAIN_BIG_NT) {
                if (CPLX) {
                    const T ar = ap[2 * e], ai = ap[2 * e + 1];
                    const T zr = a.z[2 * e], zi = a.z[2 * e + 1];
                    p1r += ar * zr - ai * zi;
                    p1i += ar * zi + ai * zr;
                    const T rr = zr - sp[2 * e], ri = zi - sp[2 * e + 1];
                    p2 += rr * rr + ri * ri;
                } else {
                    const T zv = a.z[e];
                    p1r += ap[e] * zv;
                    const T rv = zv - sp[e];
                    p2 += rv * rv;
                }
            }
            p1r = wave_sum_lane63(p1r);
            if (CPLX) p1i = wave_sum_lane63(p1i);
            p2 = wave_sum_lane63(p2);
            if (lane == WAVE - 1) {   // the lane that holds the wave's sum
                red[par][wib][0] = p1r;
                red[par][wib][1] = p1i;
                red[par][wib][2] = p2;
            }
            __syncthreads();
            T t3[3] = {T(0), T(0), T(0)};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int w = 0; w < NW; w += 4)
                    t3[c] += (red[par][w][c] + red[par][w + 1][c]) + (red[par][w + 2][c] + red[par][w + 3][c]);
            par ^= 1;
            dzr = t3[0];
            dzi = t3[1];
            const T n2 = t3[2];
            T lin;
            if (CPLX) {
                const T rr = dzr - br, ri = dzi - bi;
                fi_z = (a.lam / T(2)) * (rr * rr + ri * ri);                             // :125
                lin = c_or * (dzr - as_r) + c_oi * (dzi - as_i);                         // real_dot(grad f_i(x_i), z - x_i)
            } else {
                fi_z = loss_value(loss, dzr, br, a.lam);
                lin = c_or * (dzr - as_r);
            }
            const double fi_model = (double)(fi_x + lin) + (0.5 * a.Nd * (double)a.alpha / (double)gi) * (double)n2;   // :126-129 (Float64)
            const T tol = T(10) * Eps<T>::value * (T(1) + fabs2(fi_z));                  // :130
            if ((double)fi_z <= fi_model + (double)tol) break;                           // :131
            const T gb = gi;                                                             // :133
            gi = (T)((double)gi * 0.8);                                                  // :134
            const T hg_old = hg;
            hg = T(1) / (T(1) / hg_old + T(1) / gi - T(1) / gb);                         // :139
            for (int64_t e = tid; e < units; e += CHAIN_BIG_NT) {
#pragma unroll
                for (int c = 0; c < (CPLX ? 2 : 1); ++c) {
                    const int64_t k = CPLX ? 2 * e + c : e;
                    T t = a.av[k] / hg_old;                                              // :136
                    t += sp[k] / gi;                                                     // :137
                    t -= sp[k] / gb;                                                     // :138
                    t *= hg;                                                             // :140
                    a.av[k] = t;
                }
                prox_unit(e, hg);                                                        // :141
            }
        }
        if (stop) break;
        // the main step, :145-150
        T c_nr, c_ni = T(0);
        if (CPLX) {
            c_nr = a.lam * (dzr - br);
            c_ni = a.lam * (dzi - bi);
        } else {
            c_nr = grad_coef(loss, dzr, br, a.lam).coef();
        }
        const T r1 = hg / gi;
        const T cc = hg * a.invN;
        const T dcr = c_or - c_nr, dci = c_oi - c_ni;       // + (hg/N) grad_old - (hg/N) grad_new, both multiples of conj(a_i)
        for (int64_t e = tid; e < units; e += CHAIN_BIG_NT) {
            if (CPLX) {
                const T ar = ap[2 * e], ai = ap[2 * e + 1];
                const T zr = a.z[2 * e], zi = a.z[2 * e + 1];
                T tr = a.av[2 * e] + r1 * (zr - sp[2 * e]);                              // :145
                T ti = a.av[2 * e + 1] + r1 * (zi - sp[2 * e + 1]);
                tr += cc * (ar * dcr + ai * dci);                                        // :147, :149   conj(a) (c_old - c_new)
                ti += cc * (ar * dci - ai * dcr);
                sp[2 * e] = zr;                                                          // :146  s_i = z
                sp[2 * e + 1] = zi;
                a.av[2 * e] = tr;
                a.av[2 * e + 1] = ti;
            } else {
                const T zv = a.z[e];
                T t = a.av[e] + r1 * (zv - sp[e]);
                t += (cc * dcr) * ap[e];
                sp[e] = zv;
                a.av[e] = t;
            }
            prox_unit(e, hg);                                                            // :150
        }
        if (tid < 4) {
            T *mw = a.meta + (row * 4 + tid) * 4;
            const bool im = CPLX && (tid & 1);
            mw[0] = im ? c_ni : c_nr;
            mw[1] = fi_z;                                                                // :148 fi_x[i] = f_i(z)
            mw[2] = gi;
            mw[3] = im ? dzi : dzr;
        }
        ++done;
    }
    if (tid == 0) {
        *a.hg = hg;
        a.counters[0] = done;
        a.counters[1] = trials;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Adaptive Finito, fast path: the same step as afinito_chain_kernel with the inputs of step s (row a_i, table row s_i,
// this wave's copy of the sample's scalars) brought in by LDS-DMA DEPTH steps ahead and retired with hand-counted waits,
// exactly as chain_dma_kernel does (the compiler-scheduled version above drains the whole memory queue twice per step:
// once behind the index load, once behind the prefetch it has just issued).  Ops per step and thread that are certain to
// be issued, in program order: J stores of the table row, then 2J + 1 LDS-DMA loads; the scalar stores of lane 0 are
// left out of the count, which only makes the waits stricter.  A sample that recurs within the look-ahead window has
// its table row / scalars re-read from memory at use (the same thread / the same wave wrote them: program order).
// Needs d*sizeof(T) == J*4096 and 16-byte aligned rows, table, scalars and vectors; otherwise the kernel above runs.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst)
{
    // m0 declared clobbered, not saved and restored (see glds16)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}

__device__ __forceinline__ void glds4_at(const void *gsrc, uint32_t lds_base, int off)   // (scalar base) + (immediate): glds16_at
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_add_i32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_base), "i"(off) : "memory", "m0", "scc");
#pragma clang diagnostic pop
}

constexpr int AF_CHUNK = 512;

template <typename T, int J, int NT = CHAIN_NT, bool SHARDED = false>
constexpr size_t afinito_dma_lds_bytes()
{
    constexpr int NW = NT / WAVE;
    constexpr int DEPTH = DmaDepth<(J * NT + 255) / 256, true>::value;
    return (size_t)2 * DEPTH * J * NT * 16 + (size_t)DEPTH * NW * 256 + (AF_CHUNK + 2 * DEPTH) * sizeof(int64_t) +
           AF_CHUNK * sizeof(T) + AF_CHUNK * sizeof(int) + 16 + 2 * NW * 2 * sizeof(T) +
           (SHARDED ? (2 * (AF_CHUNK + 2 * DEPTH) + AF_SHARD_QW) * sizeof(int64_t) : 0);
}

// NT = 256, or 64: rows of up to 2 KiB on ONE wave (J = 1 / 2), where the exchange of every trial disappears (as in
// chain_dma_kernel).  The per-sample scalars keep their N x 4 x 4 layout: the single wave reads copy 0 and writes all four.
// SHARDED: the rows live in several allocations (AFinitoArgs::sh*): where a step's data row, table row and scalars are is
// resolved when its index is staged, 512 steps at a time (the table row's ADDRESS is then what the steps and the hazard flags know
// the sample by, the other two addresses ride beside it in LDS); the step itself is the same instruction for instruction.
template <typename T, int J, int LOSS, bool MASKED, int NT = CHAIN_NT, bool SHARDED = false>
__global__ void __launch_bounds__(NT) afinito_dma_kernel(AFinitoArgs<T> a_by_value)
{
    // the arguments through the kernel-argument segment, field by field where they are used (chain_dma_kernel, and why)
    (void)a_by_value;
    typedef const __attribute__((address_space(4))) AFinitoArgs<T> KernArgs;
    KernArgs &a = *(KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NW = NT / WAVE;
    static_assert(NW == 1 || NW == CHAIN_NW, "one wave or four");
    static_assert(!SHARDED || NW == CHAIN_NW, "the sharded chain runs on four waves");
    using V = typename VecOfC<T>::type;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int DEPTH = DmaDepth<(J * NT + 255) / 256, true>::value;
    constexpr int CH = AF_CHUNK;
    constexpr int OPS_PER_STEP = (MASKED ? 2 * J : 3 * J) + 1;   // MASKED: predicated table stores are not counted (chain_dma_kernel)
    // (eight waves -- 32 KiB rows, 256 registers per wave -- spill 50-190 registers with the two register sets and are still the
    // fastest of what was measured: fp64 d = 4096 0.570 us per SVRG update against 0.590 without PIPE (no spill) and 0.755 on four
    // waves with twice the chunks per thread, profiles/r04_chain_32k_ab.txt)
    constexpr bool PIPE = DEPTH >= 4;
    constexpr int WAIT_N = (PIPE ? DEPTH - 2 : DEPTH - 1) * OPS_PER_STEP;
    constexpr int ROW_BYTES = J * NT * 16;
    constexpr int MDW = 4 * sizeof(T) / 4;   // dwords in one copy of a sample's scalars
    static_assert(CH % DEPTH == 0 && DEPTH % 2 == 0, "ring slots must line up with chunk starts; ping-pong needs even DEPTH");
    static_assert(WAIT_N <= 63, "vmcnt is a 6-bit counter");

    // ringA[DEPTH][ROW_BYTES] | ringT[DEPTH][ROW_BYTES] | ringM[DEPTH][NW][256 B] | s_row | s_b | s_stale | red
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *ringA = dsm;
    unsigned char *ringT = ringA + DEPTH * ROW_BYTES;
    unsigned char *ringM = ringT + DEPTH * ROW_BYTES;
    unsigned char *cur = ringM + DEPTH * NW * 256;
    int64_t *s_row = reinterpret_cast<int64_t *>(cur);
    cur += (CH + 2 * DEPTH) * sizeof(int64_t);
    T *s_b = reinterpret_cast<T *>(cur);
    cur += CH * sizeof(T);
    int *s_stale = reinterpret_cast<int *>(cur);
    cur += CH * sizeof(int);
    cur += (16 - (reinterpret_cast<uintptr_t>(cur) & 15)) & 15;
    T(*red)[NW][2] = reinterpret_cast<T(*)[NW][2]>(cur);
    cur += 2 * NW * 2 * sizeof(T);
    cur += (8 - (reinterpret_cast<uintptr_t>(cur) & 7)) & 7;
    int64_t *s_pa = reinterpret_cast<int64_t *>(cur);                    // SHARDED: the steps' data-row addresses ...
    int64_t *s_pm = s_pa + (SHARDED ? CH + 2 * DEPTH : 0);              // ... the addresses of their scalars ...
    int64_t *s_sh = s_pm + (SHARDED ? CH + 2 * DEPTH : 0);              // ... and the shard table

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;
    const uint32_t ringA_off = (uint32_t)(uintptr_t)ringA;
    const uint32_t ringT_off = (uint32_t)(uintptr_t)ringT;
    const uint32_t ringM_off = (uint32_t)(uintptr_t)ringM;
    // this wave's pieces of the ring slots: ONE scalar register per ring (glds16_at)
    const uint32_t ringA_w = sgpr_pin(ringA_off + (uint32_t)wib * 1024u);
    const uint32_t ringT_w = sgpr_pin(ringT_off + (uint32_t)wib * 1024u);
    const uint32_t ringM_w = sgpr_pin(ringM_off + (uint32_t)wib * 256u);
    // what the step loop reads of the argument block (sgpr_pin); everything else is read where it is used
    const int64_t nsteps = sgpr_pin(a.nsteps);
    const T lam = (LOSS == CIAO_LOSS_LOGISTIC) ? T(0) : sgpr_pin(a.lam);
    const T invN = sgpr_pin(a.invN);
    const T tol_stop = sgpr_pin_computed(a.tol_b * a.invN);                       // Finito_adaptive.jl:121
    const double half_N_alpha = sgpr_pin_computed(0.5 * a.Nd * (double)a.alpha);   // :128 (left to right: (0.5 N) alpha, then / gamma_i)
    const T *const Abase = SHARDED ? nullptr : sgpr_pin_global(a.A);
    const int64_t ld = SHARDED ? 0 : sgpr_pin(a.ld);
    T *const tbase = SHARDED ? nullptr : sgpr_pin_global(a.table);
    T *const mbase = SHARDED ? nullptr : sgpr_pin_global(a.meta);
    if constexpr (SHARDED) af_shard_table_to_lds<T>(s_sh, tid);   // (the staging's first __syncthreads orders it before its readers)

    // chunk ownership and dead chunks exactly as in chain_dma_kernel
    const int64_t nchunks = d / VEC;
    bool ok[J];
    int64_t cl[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = tid + (int64_t)j * NT;
        ok[j] = !MASKED || c < nchunks;
        cl[j] = ok[j] ? c : 0;
    }
    V av[J], p[J], plo[J], phi[J];
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
    const bool hasbox = (a.g.kind == CIAO_PROX_BOX);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = cl[j];
        av[j] = ok[j] ? reinterpret_cast<const V *>(a.av)[c] : V(T(0));
        p[j] = ok[j] ? reinterpret_cast<const V *>(a.z)[c] : V(T(0));
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            plo[j][v] = -INFINITY;
            phi[j][v] = INFINITY;
            if (hasbox && ok[j]) {
                plo[j][v] = a.g.lo_vec ? a.g.lo_vec[c * VEC + v] : a.g.lo;
                phi[j][v] = a.g.hi_vec ? a.g.hi_vec[c * VEC + v] : a.g.hi;
            }
        }
    }
    T hg = *a.hg;

    auto prox_all = [&](T gl) {   // p = prox_{hg g}(av): one workgroup-uniform branch instead of a clamp per coordinate
        if (hasbox) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v) p[j][v] = prox_bf(av[j][v], gl, plo[j][v], phi[j][v]);
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v) p[j][v] = prox_l1(av[j][v], gl);
        }
    };

    // where a sample's table row and scalars are: unsharded from its row number, SHARDED the staged addresses themselves
    auto table_row = [&](int64_t row) { return SHARDED ? (T *)(__attribute__((address_space(1))) T *)(uintptr_t)row : tbase + row * d; };
    auto meta_row = [&](int64_t row, int64_t pm) {
        return SHARDED ? (T *)(__attribute__((address_space(1))) T *)(uintptr_t)pm : mbase + row * (CHAIN_NW * 4);
    };
    // const_u: the slot number is a compile-time constant where the call is inlined (the unrolled steps): LDS destinations as the
    // wave's base + an immediate; the one-off first filling of the ring runs as a loop over the slots (chain_dma_kernel)
    auto refill = [&](auto const_u, int u, int64_t r, int64_t pa, int64_t pm) {
        constexpr bool CU = decltype(const_u)::value;
        const unsigned char *ap = SHARDED ? (const unsigned char *)(__attribute__((address_space(1))) const unsigned char *)(uintptr_t)pa
                                          : reinterpret_cast<const unsigned char *>(Abase + r * ld);
        const unsigned char *sp = reinterpret_cast<const unsigned char *>(table_row(r));
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int off = (u * J + j) * NW * 1024;
            if constexpr (CU) glds16_at(ap + cl[j] * 16, ringA_w, off);
            else glds16(ap + cl[j] * 16, ringA_w + (uint32_t)off);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int off = (u * J + j) * NW * 1024;
            if constexpr (CU) glds16_at(sp + cl[j] * 16, ringT_w, off);
            else glds16(sp + cl[j] * 16, ringT_w + (uint32_t)off);
        }
        // this wave's copy of the scalars: lanes l and l + MDW fetch the same dword, only the first MDW LDS dwords are read back
        const unsigned char *mp = reinterpret_cast<const unsigned char *>(meta_row(r, pm) + wib * 4);   // the layout's four copies, one per wave
        if constexpr (CU) glds4_at(mp + (lane & (MDW - 1)) * 4, ringM_w, u * NW * 256);
        else glds4(mp + (lane & (MDW - 1)) * 4, ringM_w + (uint32_t)(u * NW * 256));
    };

    struct StepIn {
        V ar[J], sr[J];
        T m[4];
        int64_t row, row_n;
        int64_t pm, pa_n, pm_n;   // SHARDED only
        T bi;
        int stale;
    };
    StepIn in[2];
    auto fetch = [&](StepIn &x, int u, int s) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            x.ar[j] = *reinterpret_cast<const V *>(ringA + (((u * J + j) * NW + wib) * 64 + lane) * 16);
            x.sr[j] = *reinterpret_cast<const V *>(ringT + (((u * J + j) * NW + wib) * 64 + lane) * 16);
            if (MASKED && !ok[j]) x.ar[j] = x.sr[j] = V(T(0));
        }
        const T *mp = reinterpret_cast<const T *>(ringM + (u * NW + wib) * 256);
#pragma unroll
        for (int q = 0; q < 4; ++q) x.m[q] = mp[q];
        x.row = s_row[DEPTH + s];
        x.row_n = s_row[DEPTH + s + DEPTH];
        if constexpr (SHARDED) {
            x.pm = s_pm[DEPTH + s];
            x.pa_n = s_pa[DEPTH + s + DEPTH];
            x.pm_n = s_pm[DEPTH + s + DEPTH];
        } else {
            x.pm = x.pa_n = x.pm_n = 0;
        }
        x.bi = s_b[s];
        x.stale = s_stale[s];
    };

    int par = 0;
    long long done = 0, trials = 0;
    bool stop = false;
    for (int64_t base = 0; base < nsteps && !stop; base += CH) {
        const int nch = (int)((nsteps - base) < CH ? (nsteps - base) : CH);

        __syncthreads();
        int64_t hist = -1;
        if (tid < DEPTH && base > 0) hist = s_row[CH + tid];
        __syncthreads();
        if (tid < DEPTH) s_row[tid] = hist;
        for (int e = tid; e < nch + DEPTH; e += NT) {
            int64_t st = base + e;
            if (st > nsteps - 1) st = nsteps - 1;
            int64_t r = a.idx[st];
            if ((uint64_t)r >= (uint64_t)a.N) {
                *a.errflag = 1;
                r = 0;
            }
            if constexpr (SHARDED) {   // global row -> its shard's memory (which may be another GPU's)
                const AFShardRow<T> sr = af_shard_resolve<T>(s_sh, a.nshards, r, a.ld, d);
                s_row[DEPTH + e] = (int64_t)(uintptr_t)sr.trow;
                s_pa[DEPTH + e] = (int64_t)(uintptr_t)sr.arow;
                s_pm[DEPTH + e] = (int64_t)(uintptr_t)sr.mrow;
                if (e < nch) s_b[e] = sr.bp ? *sr.bp : T(0);
            } else {
                s_row[DEPTH + e] = r;
                if (e < nch) s_b[e] = a.b ? a.b[r] : T(0);
            }
        }
        __syncthreads();
        for (int e = tid; e < nch; e += NT) {
            const int64_t r = s_row[DEPTH + e];
            bool st = false;
#pragma unroll
            for (int k = 1; k <= DEPTH; ++k) st |= (s_row[DEPTH + e - k] == r);
            s_stale[e] = st ? 1 : 0;
        }
        __syncthreads();
        if (base == 0) {
#pragma unroll 1
            for (int u = 0; u < DEPTH; ++u)   // once per launch: a loop
                refill(std::false_type{}, u, uniform64(s_row[DEPTH + u]), SHARDED ? uniform64(s_pa[DEPTH + u]) : 0,
                       SHARDED ? uniform64(s_pm[DEPTH + u]) : 0);
        }
        wait_vmcnt<0>();
        drain_vmcnt_visible();
        if (PIPE) fetch(in[0], 0, 0);

        for (int s0 = 0; s0 < nch && !stop; s0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int s = s0 + u;
                if (s >= nch || stop) break;
                StepIn &x = in[PIPE ? (u & 1) : 0];
                auto mask_dead = [&]() {
                    if constexpr (MASKED) {
#pragma unroll
                        for (int j = 0; j < J; ++j)
                            if (!ok[j]) x.ar[j] = x.sr[j] = V(T(0));
                    }
                };
                if (PIPE) {
                    mask_dead();   // read one step ago
                    if (s + 1 < nch) {
                        wait_vmcnt<WAIT_N>();
                        fetch(in[(u + 1) & 1], (u + 1) % DEPTH, s + 1);
                    }
                } else {
                    wait_vmcnt<WAIT_N>();
                    fetch(x, u, s);
                    mask_dead();
                }
                const int64_t row = uniform64(x.row);
                const int64_t row_n = uniform64(x.row_n);
                const int64_t pm = SHARDED ? uniform64(x.pm) : 0;
                const int64_t pa_n = SHARDED ? uniform64(x.pa_n) : 0;
                const int64_t pm_n = SHARDED ? uniform64(x.pm_n) : 0;
                const T bi = x.bi;
                if (__builtin_amdgcn_readfirstlane(x.stale)) {
                    const V *sp = reinterpret_cast<const V *>(table_row(row));
#pragma unroll
                    for (int j = 0; j < J; ++j) x.sr[j] = ok[j] ? sp[cl[j]] : V(T(0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) x.m[q] = meta_row(row, pm)[wib * 4 + q];
                    drain_vmcnt_visible();
                }
                const T c_old = x.m[0], fi_x = x.m[1], as_i = x.m[3];
                T gi = x.m[2];
                V res[J];
#pragma unroll
                for (int j = 0; j < J; ++j) res[j] = p[j] - x.sr[j];
                T dz = T(0), fi_z = T(0), r1_acc = T(0);
                while (true) {
                    if (gi < tol_stop) {          // Finito_adaptive.jl:121-124: the stepsize collapsed
                        stop = true;
                        break;
                    }
                    ++trials;
                    T p1 = T(0), p2 = T(0);
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            p1 = fmad(x.ar[j][v], p[j][v], p1);
                            p2 = fmad(res[j][v], res[j][v], p2);
                        }
                    p1 = wave_sum_lane63(p1);
                    p2 = wave_sum_lane63(p2);
                    if constexpr (NW > 1) {
                        if (lane == WAVE - 1) {   // the lane that holds the wave's sum
                            red[par][wib][0] = p1;
                            red[par][wib][1] = p2;
                        }
                    }
                    // the two divisions of the step depend only on gamma_i and hat_gamma: issued here, they run in the shadow of
                    // the exchange instead of behind it
                    const double qc = half_N_alpha / (double)gi;                // :128 (Float64 in the reference whatever R)
                    const T r1 = hg / gi;                                                       // :145
                    T n2;
                    if constexpr (NW == 1) {   // one wave: the sums reach every lane through SGPRs, no LDS exchange
                        dz = readlane(p1, WAVE - 1);
                        n2 = readlane(p2, WAVE - 1);
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();   // raw barrier: must not drain the DMA queue
                        dz = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
                        n2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
                        par ^= 1;
                    }
                    fi_z = loss_value(LOSS, dz, bi, lam);                                     // :125
                    const double fi_model = (double)(fi_x + c_old * (dz - as_i)) + qc * (double)n2;   // :126-129
                    const T tol = T(10) * Eps<T>::value * (T(1) + fabs2(fi_z));                 // :130
                    if ((double)fi_z <= fi_model + (double)tol) {                               // :131
                        r1_acc = r1;
                        break;
                    }
                    const T gb = gi;                                                            // :133
                    gi = (T)((double)gi * 0.8);                                                 // :134 (Float64 product, rounded to R)
                    const T hg_old = hg;
                    hg = T(1) / (T(1) / hg_old + T(1) / gi - T(1) / gb);                        // :139
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            T t = av[j][v] / hg_old;                                            // :136
                            t += x.sr[j][v] / gi;                                               // :137
                            t -= x.sr[j][v] / gb;                                               // :138
                            t *= hg;                                                            // :140
                            av[j][v] = t;
                        }
                    prox_all(hg * plam);                                                        // :141
#pragma unroll
                    for (int j = 0; j < J; ++j) res[j] = p[j] - x.sr[j];                        // :142
                }
                if (stop) break;
                // the main step, :145-150
                const GradCoef<T> gn = grad_coef_t<T, LOSS>(dz, bi, lam);
                const T c_new = gn.coef();
                const T r1 = r1_acc;
                const T r2 = (hg * invN) * (c_old - c_new);   // + (hg/N) grad_old - (hg/N) grad_new, both multiples of a_i
                V *sp = reinterpret_cast<V *>(table_row(row));
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (ok[j]) sp[cl[j]] = p[j];                                                // :146  s_i = z
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const T t = fmad(r1, res[j][v], av[j][v]);                              // :145
                        av[j][v] = fmad(r2, x.ar[j][v], t);                                     // :147, :149
                    }
                }
                prox_all(hg * plam);                                                            // :150
                if (NW == 1 ? lane < CHAIN_NW : lane == 0) {   // one wave keeps all four copies of the layout identical
                    T *mp = meta_row(row, pm) + (NW == 1 ? lane : wib) * 4;
                    mp[0] = c_new;
                    mp[1] = fi_z;                                                               // :148 fi_x[i] = f_i(z)
                    mp[2] = gi;
                    mp[3] = dz;
                }
                ++done;
                refill(std::true_type{}, u, row_n, pa_n, pm_n);   // after this step's stores (program order); the look-ahead entry always exists
            }
        }
    }
    wait_vmcnt<0>();   // nothing may still be writing LDS when the workgroup retires

#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (!ok[j]) continue;
        reinterpret_cast<V *>(a.av)[cl[j]] = av[j];
        reinterpret_cast<V *>(a.z)[cl[j]] = p[j];
    }
    if (tid == 0) {
        *a.hg = hg;
        a.counters[0] = done;
        a.counters[1] = trials;
    }
}

}  // namespace ciao
