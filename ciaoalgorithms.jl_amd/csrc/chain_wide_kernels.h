// chain_wide_kernels.h -- the sequential chains (SVRG, SAGA / SAG, small-batch Finito and LFinito) on rows LONGER than one workgroup's
// registers hold (more than 8192 elements): several workgroups share ONE chain.
//
// Why.  chain_big_kernel (chain_reg_kernels.h) keeps the whole state in the caller's vectors and streams row and state through one CU:
// 12-40 us per update at d = 9000 ... 32 768, a 10-30x cliff behind the register-resident chains (0.25-1.9 us).  A chain step is a
// dot product over the row and an element-wise update: both split by COLUMNS.  Workgroup g of G owns columns [g S, (g + 1) S) of
// every vector -- its slice of the state lives in registers for the whole launch, its slices of the next two rows are in flight while
// this one is used -- and the only thing the workgroups exchange per step is their partial dot products (SVRG: two, a'w and a'z_full;
// SAGA: one), through a mailbox in global memory.
//
// The exchange.  No grid-wide barrier exists inside a kernel; the G workgroups (at most 64 of 2048 columns each: always co-resident on
// an idle chip)
// synchronise through the data itself.  Every 64-bit mailbox word carries 32 bits of payload and the 32-bit step number, written
// and read with relaxed device-scope atomics (a plain store / load that bypasses the non-coherent caches: correct wherever the
// workgroups were placed; system scope times the same): a reader spins until the word it reads carries the step it is in -- no fence, no flag to order against
// the payload, nothing but the words themselves (a fence at system scope would write the L2 back, SAGA's table stores included).
// An fp64 dot is two such words, an fp32 dot one.  Two parities: a workgroup can be at most one step ahead of the slowest (it needs
// that one's word of the step before to get there).  Every workgroup adds the G partials in workgroup order: the same sum everywhere,
// bitwise reproducible.  Spins are bounded in wall-clock time (s_memrealtime); a timeout sets the error word and ends the chain in
// every workgroup (the others run into the same bound).
//
// Results: the chain's arithmetic is chain_big_kernel's, element for element; only the dot products are added in another order
// (slices, then workgroups).
#pragma once

#include "chain_kernels.h"

namespace ciao {

constexpr int WIDE_NT = 256;
constexpr int WIDE_GMAX = 64;
constexpr unsigned long long WIDE_TIMEOUT_TICKS = 400000000ull;   // 4 s of the 100 MHz constant clock

// mailbox: [2 parities][WIDE_GMAX workgroups][4 words]  (SVRG fp64: a'w hi/lo, a'z_full hi/lo)
struct WideArgs {
    unsigned long long *box;
    int G;
    int64_t slice;   // columns per workgroup (a multiple of WIDE_NT)
};

__device__ __forceinline__ void wide_put(unsigned long long *w, unsigned int payload, unsigned int seq)
{
    __hip_atomic_store(w, ((unsigned long long)payload << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// spins until the word carries `seq`; false on timeout
__device__ __forceinline__ bool wide_get(const unsigned long long *w, unsigned int seq, unsigned int &payload)
{
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned int)v == seq) {
            payload = (unsigned int)(v >> 32);
            return true;
        }
        if ((spins & 1023u) == 1023u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0)
                t0 = now;
            else if (now - t0 > WIDE_TIMEOUT_TICKS)
                return false;
        }
    }
}
// N consecutive words of one slot at once: all the loads of a poll are in flight together (one after the other they cost a memory
// round trip each -- an fp64 pair of sums is four words), the poll is repeated until every word carries `seq`
template <int N>
__device__ __forceinline__ bool wide_get_run(const unsigned long long *w, unsigned int seq, unsigned int (&payload)[N])
{
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned long long v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && ((unsigned int)v[i] == seq);
        if (all) {
#pragma unroll
            for (int i = 0; i < N; ++i) payload[i] = (unsigned int)(v[i] >> 32);
            return true;
        }
        if ((spins & 1023u) == 1023u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0)
                t0 = now;
            else if (now - t0 > WIDE_TIMEOUT_TICKS)
                return false;
        }
    }
}
// ... and a slot's first N words for every lane with M more behind them for the lanes that `want_more` -- in the SAME poll (two loops, one per
// kind of lane, would run one after the other in a wave: two round trips)
template <int N, int M>
__device__ __forceinline__ bool wide_get_run2(const unsigned long long *w, unsigned int seq, bool want_more, unsigned int (&payload)[N],
                                              unsigned int (&more)[M])
{
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned long long v[N], u[M];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (want_more) {
#pragma unroll
            for (int i = 0; i < M; ++i) u[i] = __hip_atomic_load(w + N + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) u[i] = seq;
        }
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && ((unsigned int)v[i] == seq);
#pragma unroll
        for (int i = 0; i < M; ++i) all = all && ((unsigned int)u[i] == seq);
        if (all) {
#pragma unroll
            for (int i = 0; i < N; ++i) payload[i] = (unsigned int)(v[i] >> 32);
#pragma unroll
            for (int i = 0; i < M; ++i) more[i] = (unsigned int)(u[i] >> 32);
            return true;
        }
        if ((spins & 1023u) == 1023u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0)
                t0 = now;
            else if (now - t0 > WIDE_TIMEOUT_TICKS)
                return false;
        }
    }
}
template <typename T>
struct WideWord;
template <>
struct WideWord<float> {
    static constexpr int N = 1;
    static __device__ __forceinline__ void put(unsigned long long *w, float v, unsigned int seq) { wide_put(w, __float_as_uint(v), seq); }
    static __device__ __forceinline__ bool get(const unsigned long long *w, unsigned int seq, float &v)
    {
        unsigned int p;
        if (!wide_get(w, seq, p)) return false;
        v = __uint_as_float(p);
        return true;
    }
    static __device__ __forceinline__ float decode(const unsigned int *p) { return __uint_as_float(p[0]); }
};
template <>
struct WideWord<double> {
    static constexpr int N = 2;
    static __device__ __forceinline__ void put(unsigned long long *w, double v, unsigned int seq)
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        wide_put(w, (unsigned int)(b >> 32), seq);
        wide_put(w + 1, (unsigned int)b, seq);
    }
    static __device__ __forceinline__ bool get(const unsigned long long *w, unsigned int seq, double &v)
    {
        unsigned int hi, lo;
        if (!wide_get(w, seq, hi) || !wide_get(w + 1, seq, lo)) return false;
        v = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        return true;
    }
    static __device__ __forceinline__ double decode(const unsigned int *p)
    {
        return __longlong_as_double((long long)(((unsigned long long)p[0] << 32) | p[1]));
    }
};

// E elements per thread: the workgroup's slice is WIDE_NT E columns
template <typename T, int E, int ALG, int LOSS>
__global__ void __launch_bounds__(WIDE_NT) chain_wide_kernel(ChainArgs<T> a_by_value, WideArgs wa)
{
    (void)a_by_value;
    CIAO_KERNARG0(ChainArgs<T>, a);
    // what the step loop reads of the argument block: loaded once and pinned in scalar registers (sgpr_pin); the rest where it is used
    struct {
        int64_t nsteps, ld, batch, N;
        const int64_t *idx;
        const T *A, *b, *gam;
        T *table;
        T gamma, invN, hat_gamma, lam;
        int sag;
    } h;
    h.nsteps = sgpr_pin(a.nsteps);
    h.ld = sgpr_pin(a.ld);
    h.batch = sgpr_pin(a.batch);
    h.N = sgpr_pin(a.N);
    h.idx = sgpr_pin_global(a.idx);
    h.A = sgpr_pin_global(a.A);
    h.b = sgpr_pin_global(a.b);
    h.gam = sgpr_pin_global(a.gam);
    h.table = sgpr_pin_global(a.table);
    h.gamma = sgpr_pin(a.gamma);
    h.invN = sgpr_pin(a.invN);
    h.hat_gamma = sgpr_pin(a.hat_gamma);
    h.lam = sgpr_pin(a.lam);
    h.sag = sgpr_pin(a.sag);
    static_assert(ALG == CA_SVRG || ALG == CA_SAGA || ALG == CA_FINITO || ALG == CA_LFINITO, "the four chains");
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool HAS_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO);
    constexpr int NW = WIDE_NT / WAVE;
    using W = WideWord<T>;
    __shared__ T red[2][NW][2];
    __shared__ T gath[WIDE_GMAX][2];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, G = wa.G;
    // (A fifth wave that does nothing but publish and poll -- so that the polls do not queue behind the rows in flight of a wave
    // that also holds state: a wave's memory operations complete in order -- was built and was SLOWER, 3.4 against 2.9 us at d = 9000
    // fp64: the step is the sum of its serial pieces -- dots, two barriers, the mailbox round trip of ~1.2 us, the update.)
    constexpr bool poller = false;
    const int64_t d = a.d;
    const int64_t base = (int64_t)g * wa.slice;
    if (tid == 0) s_fail = 0;

    bool valid[E];
    int64_t col[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        col[e] = base + (tid & (WIDE_NT - 1)) + (int64_t)e * WIDE_NT;
        valid[e] = !poller && col[e] < d && col[e] < base + wa.slice;
        if (!valid[e]) col[e] = d - 1;   // (a finite, in-bounds element; never written)
    }
    T *pp = (ALG == CA_SVRG) ? a.w : a.z;      // the point the moving gradient is taken at
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
    const T gl = (HAS_GAM ? h.hat_gamma : h.gamma) * plam;   // the prox's threshold: gamma lambda (SVRG, SAGA) / hat_gamma lambda (Finito, LFinito)
    const bool boxed = (a.g.kind == CIAO_PROX_BOX);   // (its bounds are read per step where they are vectors: 4 E registers otherwise)
    T p[E], av[E], zf[E], zacc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        p[e] = valid[e] ? pp[col[e]] : T(0);
        av[e] = valid[e] ? a.av[col[e]] : T(0);
        zf[e] = (TWO && valid[e]) ? a.zf[col[e]] : T(0);
        zacc[e] = (ALG == CA_SVRG && valid[e]) ? a.z[col[e]] : T(0);
    }
    auto prox_at = [&](T v, int e) {
        if (!boxed) return prox_l1(v, gl);
        const T lo = a.g.lo_vec ? a.g.lo_vec[col[e]] : a.g.lo;
        const T hi = a.g.hi_vec ? a.g.hi_vec[col[e]] : a.g.hi;
        return prox_bf(v, gl, lo, hi);
    };
    auto row_of = [&](int64_t s) {
        int64_t row = h.idx[s];
        if ((uint64_t)row >= (uint64_t)h.N) {   // memory-safe: flag it, use row 0 (results are void once flagged)
            if (tid == 0 && g == 0) *a.errflag = 1;
            row = 0;
        }
        return row;
    };
    auto load_row = [&](T(&o)[E], int64_t row) {
        const T *ap = h.A + row * h.ld;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = __builtin_nontemporal_load(ap + col[e]);
    };
    auto load_tab = [&](T(&o)[E], int64_t row) {
        const T *sp = h.table + row * d;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = sp[col[e]];
    };

    // The rows (and, SAGA, the table rows) of the next TWO steps are in flight while this one is worked on: a row is one dependent
    // HBM access away, and a step that has to see its successor's row land cannot be shorter than that latency (1.9 of the 3.1 us
    // of the first version at d = 9000 fp64).  Three register sets in rotation -- the loop is unrolled by three so that none is ever
    // copied (a copy waits for the load it copies).
    T B0[E], B1[E], B2[E], S0[E], S1[E], S2[E];
    int64_t q0 = 0, q1 = 0, q2 = 0, inext = 0;
    T c0 = T(0), c1 = T(0), c2 = T(0);
    T h0 = a.gam_uniform, h1 = a.gam_uniform, h2 = a.gam_uniform;   // gamma_i of the three steps in flight (Finito, LFinito)
    int64_t inb = 0;                                                // position in the current batch (Finito, LFinito)
    if (!poller) {
        q0 = row_of(0);
        q1 = h.nsteps > 1 ? row_of(1) : q0;
        q2 = q0;
        inext = h.nsteps > 2 ? row_of(2) : q0;            // the row of the step after next
        c0 = h.b ? h.b[q0] : T(0);
        c1 = h.b ? h.b[q1] : T(0);
        if (HAS_GAM && h.gam) {
            h0 = h.gam[q0];
            h1 = h.gam[q1];
        }
        load_row(B0, q0);
        if (HAS_TABLE) load_tab(S0, q0);
        if (h.nsteps > 1) {
            load_row(B1, q1);
            if (HAS_TABLE) load_tab(S1, q1);
        }
    }
    int par = 0;
    __syncthreads();
    // one step: works on (cur, scur, r0, b0); (n1, sn1, r1, b1) is the next step's, requested a step ago; (n2, sn2, r2, b2) is
    // requested now for the step after next.  false: a workgroup of the chain never arrived
    auto step = [&](int64_t s, T(&cur)[E], T(&n1)[E], T(&n2)[E], T(&scur)[E], T(&sn1)[E], T(&sn2)[E], int64_t &r0, int64_t &r1, int64_t &r2,
                    T &b0, T &b1, T &b2, T &g0, T &g1, T &g2) -> bool {
        (void)n1, (void)b1, (void)g1;
        const unsigned int seq = (unsigned int)(s + 1);
        const bool more1 = s + 1 < h.nsteps, more2 = s + 2 < h.nsteps;
        T d1 = T(0), d2 = T(0);
        const int64_t row = r0;
        const T bi = b0;
        if (!poller) {
            if (more2) {
                // (the index was read a step ago, BEFORE that step's row loads: loads return in order, so waiting for it here does not
                // wait for them; the next index and b_i are requested before this step's row loads for the same reason)
                r2 = inext;
                if (s + 3 < h.nsteps) inext = row_of(s + 3);
                b2 = h.b ? h.b[r2] : T(0);
                if (HAS_GAM && h.gam) g2 = h.gam[r2];
                asm volatile("" ::: "memory");
                load_row(n2, r2);
                if (HAS_TABLE) load_tab(sn2, r2);
            }
            if (ALG == CA_LFINITO && inb == 0) {     // Finito_LFinito.jl:92  z = prox(av)
#pragma unroll
                for (int e = 0; e < E; ++e) p[e] = prox_at(av[e], e);
            }
            // ---- this workgroup's share of the dot product(s)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T ak = valid[e] ? cur[e] : T(0);
                d1 += ak * p[e];
                if (TWO) d2 += ak * zf[e];
            }
            d1 = wave_sum_lane63(d1);
            if (TWO) d2 = wave_sum_lane63(d2);
            if (lane == WAVE - 1) {
                red[par][wib][0] = d1;
                if (TWO) red[par][wib][1] = d2;
            }
        }
        __syncthreads();
        // ---- published by one thread, gathered by one thread per workgroup of the chain (G <= 64: the first wave)
        unsigned long long *mine = wa.box + ((size_t)(s & 1) * WIDE_GMAX + g) * 4;
        if (tid == 0) {
            const T v1 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
            W::put(mine, v1, seq);
            if (TWO) {
                const T v2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
                W::put(mine + W::N, v2, seq);
            }
        }
        if (tid < G) {
            const unsigned long long *theirs = wa.box + ((size_t)(s & 1) * WIDE_GMAX + lane) * 4;
            unsigned int pw[(TWO ? 2 : 1) * W::N];
            if (!wide_get_run(theirs, seq, pw)) {
                s_fail = 1;
#pragma unroll
                for (int i = 0; i < (TWO ? 2 : 1) * W::N; ++i) pw[i] = 0;
            }
            gath[lane][0] = W::decode(pw);
            gath[lane][1] = TWO ? W::decode(pw + (TWO ? W::N : 0)) : T(0);
        }
        __syncthreads();
        if (s_fail) {   // a workgroup of the chain never arrived: give up everywhere (the others run into the same bound)
            if (tid == 0) *a.errflag = 5;
            return false;
        }
        par ^= 1;
        if (poller) return true;
        // the G partials added by every wave in the same fixed tree (lane q holds workgroup q's; a serial loop over G LDS reads was
        // 0.7 us of the step at G = 16)
        d1 = lane < G ? gath[lane][0] : T(0);
        d1 = readlane(wave_sum_lane63(d1), WAVE - 1);
        if (TWO) {
            d2 = lane < G ? gath[lane][1] : T(0);
            d2 = readlane(wave_sum_lane63(d2), WAVE - 1);
        } else {
            d2 = T(0);
        }
        // ---- the element-wise update of this workgroup's columns (chain_big_kernel's arithmetic)
        const GradCoef<T> gp = grad_coef_t<T, LOSS>(d1, bi, h.lam);
        const GradCoef<T> gz = grad_coef_t<T, LOSS>(d2, bi, h.lam);
        T *sp = HAS_TABLE ? h.table + row * d : nullptr;
        const T gi = g0;
        const bool last_of_batch = (inb + 1 == h.batch) || !more1;
        // SAGA: the table rows of the next two steps were requested BEFORE this step's store (the one for step s + 1 a step ago, the
        // one for step s + 2 at the top of this step): where they are this very sample's, they are stale -- the row is what this step writes
        const bool fix1 = HAS_TABLE && more1 && r1 == row, fix2 = HAS_TABLE && more2 && r2 == row;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const T ak = cur[e];
            if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                T t = gz.elem(ak) - gp.elem(ak);
                t -= av[e];
                t *= h.gamma;
                t += p[e];
                const T wn = prox_at(t, e);
                p[e] = wn;
                zacc[e] += wn;
            } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                const T gn = gp.elem(ak);
                const T sk = scur[e];
                const T del = (gn - sk) * h.invN;
                T wv;
                if (h.sag) {
                    av[e] += del;
                    wv = p[e] - h.gamma * av[e];
                } else {
                    wv = p[e] - h.gamma * (gn - sk + av[e]);
                    av[e] += del;
                }
                p[e] = prox_at(wv, e);
                if (valid[e]) sp[col[e]] = gn;
                if (fix1) sn1[e] = gn;
                if (fix2) sn2[e] = gn;
            } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                const T t = p[e] - (gi * h.invN) * gp.elem(ak);
                av[e] += (t - scur[e]) * (h.hat_gamma / gi);
                if (valid[e]) sp[col[e]] = t;
                if (fix1) sn1[e] = t;
                if (fix2) sn2[e] = t;
                if (last_of_batch) p[e] = prox_at(av[e], e);
            } else {                                                         // Finito_LFinito.jl:93-98
                const T c = h.hat_gamma * h.invN;
                T avk = av[e];
                avk += c * gz.elem(ak);
                avk -= c * gp.elem(ak);
                avk += (h.hat_gamma / gi) * (p[e] - zf[e]);
                av[e] = avk;
            }
        }
        if (++inb == h.batch) inb = 0;
        return true;
    };
    for (int64_t s = 0; s < h.nsteps; s += 3) {
        if (!step(s, B0, B1, B2, S0, S1, S2, q0, q1, q2, c0, c1, c2, h0, h1, h2)) break;
        if (s + 1 >= h.nsteps) break;
        if (!step(s + 1, B1, B2, B0, S1, S2, S0, q1, q2, q0, c1, c2, c0, h1, h2, h0)) break;
        if (s + 2 >= h.nsteps) break;
        if (!step(s + 2, B2, B0, B1, S2, S0, S1, q2, q0, q1, c2, c0, c1, h2, h0, h1)) break;
    }
    // ---- the slice of the state back to the caller's vectors
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (!valid[e]) continue;
        pp[col[e]] = p[e];
        if (ALG == CA_SVRG)
            a.z[col[e]] = zacc[e];
        else
            a.av[col[e]] = av[e];   // (SAGA, Finito, LFinito: the aggregate moves with the chain)
    }
}

}  // namespace ciao

namespace ciao {

// ------------------------------------------------------------------------------------------------------------------------------------
// Adaptive Finito (Finito_adaptive.jl:118-150) on rows beyond one workgroup's registers: the same column split and the same mailbox.
// One exchange per backtracking TRIAL -- both sums of the trial (a_i'z and |z - s_i|^2) travel in one mailbox slot -- and every
// workgroup takes the same decisions from the same totals (added in the same order everywhere), so the number of exchanges is the
// same in all of them; the exchange counter, not the step, numbers the mailbox words.  The sample's four scalars (c_i, f_i(x_i),
// gamma_i, a_i's_i) have ONE reader and writer, workgroup 0 (its own stores and loads: program order), which hands them to the others
// with the first exchange of every step in its slot's upper words.  hat_gamma lives in every workgroup and moves identically.
// The arithmetic is afinito_dma_kernel's, element for element; the dot products are added slices first, then workgroups.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int AFW_WORDS = 16;   // mailbox words per workgroup and parity: two sums + workgroup 0's four scalars (fp64: two words each)

template <typename T, int E, int LOSS>
__global__ void __launch_bounds__(WIDE_NT) afinito_wide_kernel(AFinitoArgs<T> a_by_value, WideArgs wa)
{
    (void)a_by_value;
    CIAO_KERNARG0(AFinitoArgs<T>, a);
    // Read where they are used, not pinned: this kernel's scalar registers are taken by its own uniform state (three sets of
    // per-sample scalars, rows, b_i in flight; eight validity masks), and a step is 4 us of mailbox round trips -- a scalar load
    // from the argument block is nothing beside them, a pinned field is a register for the whole launch (pinned: 27-34 spilled
    // scalar registers, read in place: 0-10).  The two products the backtracking loop needs are formed once.
    const T tol_stop = sgpr_pin_computed(a.tol_b * a.invN);                       // Finito_adaptive.jl:121
    const double half_N_alpha = sgpr_pin_computed(0.5 * a.Nd * (double)a.alpha);   // :128, left to right
    constexpr int NW = WIDE_NT / WAVE;
    using W = WideWord<T>;
    constexpr int WN = W::N;
    __shared__ T red[2][NW][2];
    __shared__ T gath[WIDE_GMAX][2];
    __shared__ T s_meta[4];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, G = wa.G;
    const int64_t d = a.d;
    const int64_t base = (int64_t)g * wa.slice;
    if (tid == 0) s_fail = 0;

    bool valid[E];
    int64_t col[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        col[e] = base + tid + (int64_t)e * WIDE_NT;
        valid[e] = col[e] < d && col[e] < base + wa.slice;
        if (!valid[e]) col[e] = d - 1;
    }
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
    const bool boxed = (a.g.kind == CIAO_PROX_BOX);
    T p[E], av[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        p[e] = valid[e] ? a.z[col[e]] : T(0);
        av[e] = valid[e] ? a.av[col[e]] : T(0);
    }
    T hg = *a.hg;
    auto prox_at = [&](T v, T gl, int e) {
        if (!boxed) return prox_l1(v, gl);
        const T lo = a.g.lo_vec ? a.g.lo_vec[col[e]] : a.g.lo;
        const T hi = a.g.hi_vec ? a.g.hi_vec[col[e]] : a.g.hi;
        return prox_bf(v, gl, lo, hi);
    };
    auto row_of = [&](int64_t s) {
        int64_t row = a.idx[s];
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0 && g == 0) *a.errflag = 1;
            row = 0;
        }
        return row;
    };
    auto load_row = [&](T(&o)[E], int64_t row) {
        const T *ap = a.A + row * a.ld;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = __builtin_nontemporal_load(ap + col[e]);
    };
    auto load_tab = [&](T(&o)[E], int64_t row) {
        const T *sp = a.table + row * d;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = sp[col[e]];
    };
    struct Meta {
        T c, f, gam, as;
    };
    auto load_meta = [&](Meta &m, int64_t row) {   // workgroup 0 only: every wave its own copy of the layout's four (written by its lane 0)
        const T *mp = a.meta + (row * CHAIN_NW + wib) * 4;
        m.c = mp[0], m.f = mp[1], m.gam = mp[2], m.as = mp[3];
    };

    // rows, table rows and (workgroup 0) scalars of the next two steps in flight, three register sets in rotation (chain_wide_kernel)
    T B0[E], B1[E], B2[E], S0[E], S1[E], S2[E];
    Meta M0{}, M1{}, M2{};
    int64_t q0 = row_of(0), q1 = a.nsteps > 1 ? row_of(1) : q0, q2 = q0, inext = a.nsteps > 2 ? row_of(2) : q0;
    T c0 = a.b ? a.b[q0] : T(0), c1 = a.b ? a.b[q1] : T(0), c2 = T(0);
    if (g == 0) {
        load_meta(M0, q0);
        if (a.nsteps > 1) load_meta(M1, q1);
    }
    load_row(B0, q0);
    load_tab(S0, q0);
    if (a.nsteps > 1) {
        load_row(B1, q1);
        load_tab(S1, q1);
    }
    unsigned int xc = 0;   // exchanges made so far: numbers the mailbox words, picks the parity
    long long done = 0, trials = 0;
    bool stop = false, fail = false;
    __syncthreads();

    // one exchange: this workgroup's (v1, v2) out, everybody's in, added in workgroup order; with_meta: workgroup 0's scalars ride along
    auto exchange = [&](T &v1, T &v2, bool with_meta, Meta &m) -> bool {
        const unsigned int seq = xc + 1;
        unsigned long long *mine = wa.box + ((size_t)(xc & 1) * WIDE_GMAX + g) * AFW_WORDS;
        if (lane == WAVE - 1) {
            red[xc & 1][wib][0] = v1;
            red[xc & 1][wib][1] = v2;
        }
        __syncthreads();
        if (tid == 0) {
            const int par = xc & 1;
            W::put(mine, (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]), seq);
            W::put(mine + WN, (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]), seq);
            if (with_meta && g == 0) {
                W::put(mine + 2 * WN, m.c, seq);
                W::put(mine + 3 * WN, m.f, seq);
                W::put(mine + 4 * WN, m.gam, seq);
                W::put(mine + 5 * WN, m.as, seq);
            }
        }
        if (tid < G) {
            const unsigned long long *theirs = wa.box + ((size_t)(xc & 1) * WIDE_GMAX + lane) * AFW_WORDS;
            // every lane its workgroup's two sums; lane 0 (workgroup 0's slot) also the four scalars behind them, in the same poll
            const bool want_meta = with_meta && lane == 0 && g != 0;
            unsigned int pw[2 * WN], pm[4 * WN];
            const bool ok = wide_get_run2(theirs, seq, want_meta, pw, pm);
            if (!ok) {
#pragma unroll
                for (int i = 0; i < 2 * WN; ++i) pw[i] = 0;
#pragma unroll
                for (int i = 0; i < 4 * WN; ++i) pm[i] = 0;
                s_fail = 1;
            }
            gath[lane][0] = W::decode(pw);
            gath[lane][1] = W::decode(pw + WN);
            if (want_meta) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s_meta[q] = W::decode(pm + q * WN);
            }
        }
        __syncthreads();
        ++xc;
        if (s_fail) {
            if (tid == 0) *a.errflag = 5;
            return false;
        }
        T t = lane < G ? gath[lane][0] : T(0);
        v1 = readlane(wave_sum_lane63(t), WAVE - 1);
        t = lane < G ? gath[lane][1] : T(0);
        v2 = readlane(wave_sum_lane63(t), WAVE - 1);
        if (with_meta && g != 0) m.c = s_meta[0], m.f = s_meta[1], m.gam = s_meta[2], m.as = s_meta[3];
        return true;
    };

    auto step = [&](int64_t s, T(&cur)[E], T(&n2)[E], T(&scur)[E], T(&sn1)[E], T(&sn2)[E], int64_t &w0, int64_t &w1, int64_t &w2, T &b0, T &b2,
                    Meta &m0, Meta &m1, Meta &m2) {
        const bool more1 = s + 1 < a.nsteps, more2 = s + 2 < a.nsteps;
        const int64_t row = w0;
        const T bi = b0;
        if (more2) {
            w2 = inext;
            if (s + 3 < a.nsteps) inext = row_of(s + 3);
            b2 = a.b ? a.b[w2] : T(0);
            if (g == 0) load_meta(m2, w2);
            asm volatile("" ::: "memory");
            load_row(n2, w2);
            load_tab(sn2, w2);
        }
        T res[E];
#pragma unroll
        for (int e = 0; e < E; ++e) res[e] = valid[e] ? p[e] - scur[e] : T(0);   // (columns beyond the slice count for nothing in |z - s_i|^2)
        Meta m = m0;
        T gi = T(0), dz = T(0), fi_z = T(0), r1_acc = T(0), c_old = T(0), fi_x = T(0), as_i = T(0);
        bool first = true;
        while (true) {
            if (!first && gi < tol_stop) {   // Finito_adaptive.jl:121-124 (checked before the first trial below, once gamma_i is known)
                stop = true;
                break;
            }
            T p1 = T(0), p2 = T(0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T ak = valid[e] ? cur[e] : T(0);
                p1 = fmad(ak, p[e], p1);
                p2 = fmad(res[e], res[e], p2);
            }
            p1 = wave_sum_lane63(p1);
            p2 = wave_sum_lane63(p2);
            if (!exchange(p1, p2, first, m)) {
                fail = true;
                break;
            }
            if (first) {   // the sample's scalars are here now (workgroup 0: its own; the others: from its slot)
                c_old = m.c, fi_x = m.f, gi = m.gam, as_i = m.as;
                first = false;
                if (gi < tol_stop) {
                    stop = true;
                    break;
                }
            }
            ++trials;
            dz = p1;
            const T n2v = p2;
            const double qc = half_N_alpha / (double)gi;                        // :128
            const T r1 = hg / gi;                                                               // :145
            fi_z = loss_value(LOSS, dz, bi, a.lam);                                             // :125
            const double fi_model = (double)(fi_x + c_old * (dz - as_i)) + qc * (double)n2v;    // :126-129
            const T tol = T(10) * Eps<T>::value * (T(1) + fabs2(fi_z));                         // :130
            if ((double)fi_z <= fi_model + (double)tol) {                                       // :131
                r1_acc = r1;
                break;
            }
            const T gb = gi;                                                                    // :133
            gi = (T)((double)gi * 0.8);                                                         // :134
            const T hg_old = hg;
            hg = T(1) / (T(1) / hg_old + T(1) / gi - T(1) / gb);                                // :139
#pragma unroll
            for (int e = 0; e < E; ++e) {
                T t = av[e] / hg_old;                                                           // :136
                t += scur[e] / gi;                                                              // :137
                t -= scur[e] / gb;                                                              // :138
                t *= hg;                                                                        // :140
                av[e] = t;
                p[e] = prox_at(t, hg * plam, e);                                                // :141
                res[e] = valid[e] ? p[e] - scur[e] : T(0);                                      // :142
            }
        }
        if (stop || fail) return;
        // the main step, :145-150
        const GradCoef<T> gn = grad_coef_t<T, LOSS>(dz, bi, a.lam);
        const T c_new = gn.coef();
        const T r1 = r1_acc;
        const T r2 = (hg * a.invN) * (c_old - c_new);
        T *sp = a.table + row * d;
        // the table rows (and, workgroup 0, the scalars) of the next two steps were requested before this step's stores: where they are
        // this very sample's, what this step writes replaces them
        const bool fix1 = more1 && w1 == row, fix2 = more2 && w2 == row;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (valid[e]) sp[col[e]] = p[e];                                                    // :146  s_i = z
            if (fix1) sn1[e] = p[e];
            if (fix2) sn2[e] = p[e];
            const T t = fmad(r1, res[e], av[e]);                                                // :145
            av[e] = fmad(r2, valid[e] ? cur[e] : T(0), t);                                      // :147, :149
        }
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = prox_at(av[e], hg * plam, e);                        // :150
        if (g == 0) {
            const Meta mn{c_new, fi_z, gi, dz};                                                 // :148 fi_x[i] = f_i(z)
            if (fix1) m1 = mn;
            if (fix2) m2 = mn;
            if (lane == 0) {   // every wave its own copy (it is the one that reads it back: program order)
                T *mp = a.meta + (row * CHAIN_NW + wib) * 4;
                mp[0] = c_new, mp[1] = fi_z, mp[2] = gi, mp[3] = dz;
            }
        }
        ++done;
    };
    for (int64_t s = 0; s < a.nsteps && !stop && !fail; s += 3) {
        step(s, B0, B2, S0, S1, S2, q0, q1, q2, c0, c2, M0, M1, M2);
        if (s + 1 >= a.nsteps || stop || fail) break;
        step(s + 1, B1, B0, S1, S2, S0, q1, q2, q0, c1, c0, M1, M2, M0);
        if (s + 2 >= a.nsteps || stop || fail) break;
        step(s + 2, B2, B1, S2, S0, S1, q2, q0, q1, c2, c1, M2, M0, M1);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (!valid[e]) continue;
        a.z[col[e]] = p[e];
        a.av[col[e]] = av[e];
    }
    if (g == 0 && tid == 0) {
        *a.hg = hg;
        a.counters[0] = done;
        a.counters[1] = trials;
    }
}

}  // namespace ciao
