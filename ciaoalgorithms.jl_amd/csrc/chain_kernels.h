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

#include <cstddef>
#include <type_traits>

#include "ciao_common.h"

namespace ciao {

enum ChainAlg {
    CA_SVRG = 0,
    CA_SAGA = 1,
    CA_FINITO = 2,
    CA_LFINITO = 3,
    CA_SVRGC = 4   // SVRG with a_i'z_full taken from the full pass that produced av (passed through `gam`): one dot per step
};

template <typename T>
struct ChainArgs {
    // A batch of independent chains in ONE launch (ciao_ctx_chain_batch_begin / _end): workgroup k runs multi[k] (device memory) and
    // everything else of the by-value argument is ignored.  nullptr: the one chain described by the fields below.  First field,
    // so that the host can patch it into a recorded argument block whatever T is.
    const ChainArgs<T> *multi;
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
    int64_t N;             // rows the indices may address (index validation): local rows, or N_total with a shard table
    int *errflag;          // device word set to 1 on an out-of-range index
    // Row-sharded problem (ciao_ctx_set_shards; SURVEY.md 8e "one chain on one GPU pulling remote rows over xGMI"): the rows
    // live in nshards allocations, shard k = global rows [sh_row0[k], sh_row0[k+1]); the pointers may be peer-mapped memory of
    // other GPUs.  idx then holds GLOBAL rows.  nshards = 0: A / b / table above are the whole problem.
    int nshards;
    const T *shA[CIAO_MAX_SHARDS];
    const T *shb[CIAO_MAX_SHARDS];
    T *shT[CIAO_MAX_SHARDS];
    int64_t sh_row0[CIAO_MAX_SHARDS + 1];
};

// The argument block a chain kernel reads its fields from, in the constant address space (scalar loads, where a field is used):
// the kernel-argument segment itself, or -- a batch of chains (ChainArgs::multi) -- workgroup k's own block in device memory, which
// the host wrote before the launch and nothing writes during it.
template <typename T>
using ChainArgsK = const __attribute__((address_space(4))) ChainArgs<T>;
template <typename T>
__device__ __forceinline__ ChainArgsK<T> *chain_args_block()
{
    ChainArgsK<T> *k = (ChainArgsK<T> *)__builtin_amdgcn_kernarg_segment_ptr();
    const ChainArgs<T> *m = k->multi;
    if (m) k = (ChainArgsK<T> *)(uintptr_t)(m + blockIdx.x);
    return k;
}

// A batch of chains (ChainArgs::multi): workgroup k takes its own argument block.  Word by word through v_readfirstlane, so that
// every field is in scalar registers exactly as a kernel argument would be (the inline asm of the chain kernels names SGPRs).
template <typename T>
__device__ __forceinline__ void chain_args_fetch(ChainArgs<T> &a)
{
    static_assert(sizeof(ChainArgs<T>) % 4 == 0, "whole dwords");
    if (!a.multi) return;
    const unsigned int *src = reinterpret_cast<const unsigned int *>(a.multi + blockIdx.x);
    unsigned int w[sizeof(ChainArgs<T>) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(ChainArgs<T>) / 4; ++i) w[i] = (unsigned int)__builtin_amdgcn_readfirstlane((int)src[i]);
    __builtin_memcpy(&a, w, sizeof a);
    // pointers read from memory are generic to the compiler (flat loads / stores, which count on BOTH memory counters and break
    // the hand-counted waits): say that they are global, as it knows of a kernel argument's
    auto glob = [](auto *&p) {
        using P = std::remove_reference_t<decltype(*p)>;
        p = (P *)(__attribute__((address_space(1))) P *)(uintptr_t)p;
    };
    glob(a.A), glob(a.b), glob(a.idx), glob(a.gam), glob(a.table), glob(a.g.lo_vec), glob(a.g.hi_vec);
    glob(a.av), glob(a.z), glob(a.zf), glob(a.w), glob(a.errflag);
}

// Where global row r of a row-sharded problem lives: its data row, its b entry (or nullptr), its table row.
// The shard table (33 qwords: shA[8] | shb[8] | shT[8] | sh_row0[9]) is copied to LDS once per kernel and searched THERE, with the
// row's own (per-lane) shard number as an index.  Two other ways were measured and dropped: indexing the kernel argument's arrays
// with the shard number makes the compiler copy the whole argument block to scratch memory and read it from there (round 3: 520
// bytes of scratch, 320 instructions per SAGA step against 154); a chain of selects over compile-time indexes keeps all 66 scalar
// registers of the table live through the whole kernel, and everything else spills to VGPR lanes (80-200 spilled SGPRs; the
// sharded SAGA step 0.44 us against 0.38 unsharded, and the wave-specialised kernel 0.39 against 0.36 with the search in its stager).
constexpr int SHARD_QW = 4 * CIAO_MAX_SHARDS + 1;
template <typename T>
struct ShardRow {
    const T *arow;
    const T *bp;
    T *trow;
};
// Lane j < 33 copies qword j of the table straight from the kernel-argument segment (a vector load from constant memory: no
// scalar registers at all; the chain kernels take their ChainArgs as the one kernel argument, at offset 0, and a chain over a
// shard table is never part of a batch, whose arguments would live elsewhere).
template <typename T>
__device__ __forceinline__ void shard_table_to_lds(int64_t *s_sh, int tid)
{
    static_assert(offsetof(ChainArgs<T>, shb) == offsetof(ChainArgs<T>, shA) + 8 * CIAO_MAX_SHARDS &&
                  offsetof(ChainArgs<T>, shT) == offsetof(ChainArgs<T>, shA) + 16 * CIAO_MAX_SHARDS &&
                  offsetof(ChainArgs<T>, sh_row0) == offsetof(ChainArgs<T>, shA) + 24 * CIAO_MAX_SHARDS, "the table is 33 contiguous qwords");
    const unsigned char __attribute__((address_space(4))) *ka =
        (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    if (tid < SHARD_QW) s_sh[tid] = reinterpret_cast<const int64_t __attribute__((address_space(4))) *>(ka + offsetof(ChainArgs<T>, shA))[tid];
}
template <typename T>
__device__ __forceinline__ ShardRow<T> shard_resolve(const int64_t *s_sh, int nshards, int64_t r, int64_t ld, int64_t d)
{
    const int64_t *row0 = s_sh + 3 * CIAO_MAX_SHARDS;
    int k = 0;   // sh_row0 ascends: the number of shards that start at or before r, minus one
#pragma unroll
    for (int j = 1; j < CIAO_MAX_SHARDS; ++j) k += (j < nshards && r >= row0[j]) ? 1 : 0;
    const int64_t local = r - row0[k];
    const T *A = reinterpret_cast<const T *>((uintptr_t)s_sh[k]);
    const T *b = reinterpret_cast<const T *>((uintptr_t)s_sh[CIAO_MAX_SHARDS + k]);
    T *tb = reinterpret_cast<T *>((uintptr_t)s_sh[2 * CIAO_MAX_SHARDS + k]);
    // pointers read from LDS are generic to the compiler; these are global memory (local or peer-mapped)
    auto glob = [](auto *p) {
        using P = std::remove_pointer_t<decltype(p)>;
        return (P *)(__attribute__((address_space(1))) P *)(uintptr_t)p;
    };
    ShardRow<T> o;
    o.arow = glob(A) + local * ld;
    o.bp = b ? glob(b) + local : nullptr;
    o.trow = tb ? glob(tb) + local * d : nullptr;
    return o;
}

template <typename T>
struct VecOfC;
template <>
struct VecOfC<float> {
    typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct VecOfC<double> {
    typedef double type __attribute__((ext_vector_type(2)));
};


constexpr int CHAIN_NT = 256;
constexpr int CHAIN_NW = CHAIN_NT / WAVE;
constexpr int CHAIN_CHUNK = 1024;   // steps whose indices / b_i / gamma_i (/ row addresses) are staged in LDS at a time

template <int E>
struct ChainDepth {
    static constexpr int value = E <= 4 ? 8 : (E <= 8 ? 4 : 2);
};

// Branch-free prox for one coordinate: soft threshold (gl = tau*lambda, 0 unless NormL1) then clamp (lo/hi = -/+inf
// unless IndBox).  One straight-line form for Zero / NormL1 / IndBox keeps the dependent chain free of branches.
// clamp(v, -t, t) on the chains: fmin/fmax make hipcc canonicalise their operands first (v_max_f64 x, x: three extra
// instructions per step); the two machine instructions themselves, with the negation as a source modifier, do not.
__device__ __forceinline__ float clamp_chain(float v, float t) { return clamp_sym(v, t); }
__device__ __forceinline__ double clamp_chain(double v, double t)
{
    double m, r;
    asm("v_max_f64 %0, %1, -%2" : "=v"(m) : "v"(v), "v"(t));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(m), "v"(t));
    return r;
}
template <typename T>
__device__ __forceinline__ T prox_bf(T v, T gl, T lo, T hi)
{
    // soft threshold as v - clamp(v, -gl, gl): the same value as the reference's three-way form for every finite v
    // (v > gl: v - gl; v < -gl: v + gl; else v - v = 0) in three instructions instead of compares + 64-bit selects
    const T s = v - clamp_chain(v, gl);
    return fmin2(fmax2(s, lo), hi);
}
// the same without the box (g = Zero or NormL1: lo/hi are -/+inf and the clamp would be the identity)
template <typename T>
__device__ __forceinline__ T prox_l1(T v, T gl)
{
    return v - clamp_chain(v, gl);
}

// LOSS is a template parameter here (CIAO_LOSS_LS also serves Zero(): lam = 0 and no data), FULL = every thread's E
// elements are inside the vector (d == E*256): no per-element masks anywhere.
template <typename T, int LOSS>
__device__ __forceinline__ GradCoef<T> grad_coef_t(T dot, T bi, T lam)
{
    GradCoef<T> g;
    if (LOSS == CIAO_LOSS_LOGISTIC) {
        g.s1 = -bi / (T(1) + fexp(bi * dot));
        g.s2 = T(1);
    } else {
        g.s1 = dot - bi;
        g.s2 = lam;
    }
    return g;
}

// wave-uniform 64-bit value -> SGPR pair
__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

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

// ------------------------------------------------------------------------------------------------------------------
// Fast chain: LDS-DMA row ring.
//
// The register-ring kernel above leaves the waits to hipcc, which drains the whole vector-memory queue once per ring
// revolution (its s_waitcnt bookkeeping is conservative across the loop back-edge).  Here the prefetched rows never
// touch a register on their way in: every thread issues `global_load_lds_dwordx4` (16 B per lane, LDS destination =
// wave base + lane*16) DEPTH steps ahead, and reads back ONLY the 16-byte chunks its own lanes loaded -- so the only
// ordering needed is the issuing wave's own counted `s_waitcnt vmcnt(N)`, placed by hand (the compiler does not see
// inline-asm memory operations, cdna_hip_programming.md section 5.7).  Counting, per step and per thread:
//   J LDS-DMA loads of a_i, and for SAGA/Finito J LDS-DMA loads of the table row + J 16-byte table stores,
// all unconditional and in program order; ops younger than the slot being consumed = (DEPTH-1) * that.  Anything the
// compiler adds (the rare hazard re-read, chunk staging) only makes the hardware counter drain further: safe.
//
// Ownership: thread t owns the 16-byte chunks t + 256*j, j < J, of every d-vector (d*sizeof(T) == J*256*16).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    // m0 (the LDS destination base of the DMA) is declared clobbered instead of saved and restored around every load: hipcc
    // never holds a value in m0 across statements (it sets it next to the few instructions that read it), and two scalar
    // moves per load are on the chain's issue path
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}

// The same with the row's (wave-uniform) base address in an SGPR pair and the thread's 32-bit byte offset in a VGPR: the
// 64-bit address addition per load disappears from the vector pipeline.
//
// THE SCALAR BASE IS COPIED BY A SCALAR INSTRUCTION INSIDE THE ASM, and the memory instruction reads the copy.  gfx9 rule (CDNA3/4
// ISA, manually inserted wait states): "VALU writes SGPR -> VMEM reads that SGPR: 5 wait states".  hipcc inserts the s_nops for the
// memory instructions it emits itself; the operands of an inline asm are opaque to its hazard recognizer.  A base that reaches the asm
// from a v_readfirstlane_b32 (uniform64 of a pointer read from LDS) or -- the case that faulted in round 4 -- from the v_readlane_b32
// that RESTORES a spilled scalar register, which hipcc puts directly in front of the use, is read STALE by the memory instruction:
// on MI355X 76-98 % of the loads of tools/micro/sgpr_hazard_lab.hip go through the old content of the register pair with 0-3 wait
// states in between, none with 4 or more, none with a scalar instruction in between (profiles/r05_sgpr_hazard_lab.txt).  A scalar
// instruction reading a VALU-written SGPR is interlocked by the hardware, and a VMEM instruction reading a SALU-written SGPR has no
// hazard: the copy makes the asm correct wherever the compiler puts the definition of its operand.  It takes the place of the s_nop
// that the m0 write needs before the LDS-DMA anyway: no instruction more.  tools/sgpr_vmem_hazard.py checks the built library.
__device__ __forceinline__ void glds16s(const void *sbase, uint32_t voff, uint32_t lds_dst)
{
    uint64_t base_copy;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %3\n\ts_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0"
                 : "=&s"(base_copy) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}

// The LDS destination as (a wave's base in ONE scalar register) + (a byte offset that is a compile-time constant once the ring's
// loops are unrolled): written as base + offset in C++, hipcc hoists every sum out of the step loop into a scalar register of its
// own -- 2 * DEPTH * J of them (32-64 for a table chain), the largest single consumer of the chain kernels' scalar registers and
// why they spilled.  The addition is one scalar instruction either way (s_add_i32 for s_mov_b32).
__device__ __forceinline__ void glds16_at(const void *gsrc, uint32_t lds_base, int off)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_add_i32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base), "i"(off) : "memory", "m0", "scc");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void glds16s_at(const void *sbase, uint32_t voff, uint32_t lds_base, int off)
{
    uint64_t base_copy;   // (glds16s: the memory instruction reads a scalar COPY of the base)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_add_i32 m0, %3, %4\n\ts_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0"
                 : "=&s"(base_copy) : "v"(voff), "s"(sbase), "s"(lds_base), "i"(off) : "memory", "m0", "scc");
#pragma clang diagnostic pop
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// vmcnt(0) that hipcc's own wait bookkeeping also sees (simm16: vmcnt[3:0]=0, expcnt[6:4]=7, lgkmcnt[11:8]=15,
// vmcnt[5:4] in bits 15:14 = 0).  Used where compiler-tracked loads must be retired BEFORE the hand-counted loop, so
// that hipcc does not re-insert a draining wait for them inside it.
__device__ __forceinline__ void drain_vmcnt_visible()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}

// The reads of the cross-wave exchange in two halves (issue, wait), so that work can be placed between them: N 16-byte reads of
// the partials from LDS byte address `addr`, then an lgkmcnt(0) that also (re)defines the registers -- no use of them can move
// above the wait.
template <typename V>
__device__ __forceinline__ void xchg_issue(uint32_t addr, V (&rv)[1])
{
    asm volatile("ds_read_b128 %0, %1" : "=&v"(rv[0]) : "v"(addr) : "memory");
}
template <typename V>
__device__ __forceinline__ void xchg_issue(uint32_t addr, V (&rv)[2])
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(rv[0]), "=&v"(rv[1]) : "v"(addr) : "memory");
}
template <typename V>
__device__ __forceinline__ void xchg_issue(uint32_t addr, V (&rv)[4])
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                 : "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3]) : "v"(addr) : "memory");
}
template <typename V>
__device__ __forceinline__ void xchg_issue(uint32_t addr, V (&rv)[8])
{
    xchg_issue(addr, reinterpret_cast<V (&)[4]>(rv[0]));
    xchg_issue(addr + 64, reinterpret_cast<V (&)[4]>(rv[4]));
}
// fp64, one value per wave in 16-byte slots {value, unused}: the four values by two ds_read2_b64 (8-byte units 0,2 and 4,6)
template <typename V>
__device__ __forceinline__ void xchg_issue_single64(uint32_t addr, V (&rv)[2])
{
    asm volatile("ds_read2_b64 %0, %2 offset1:2\n\tds_read2_b64 %1, %2 offset0:4 offset1:6" : "=&v"(rv[0]), "=&v"(rv[1]) : "v"(addr) : "memory");
}
template <typename V, int N>
__device__ __forceinline__ void xchg_wait(V (&rv)[N])
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(rv[i]));
}

template <int J, bool TABLE, bool SHARDED = false>   // J = row bytes / 4096
struct DmaDepth {   // ring slots: enough lead to cover an HBM miss at 0.4-0.9 us per step, within 128 KiB of LDS for the rings:
                    // with a table ring beside the row ring 64 KiB each, without one the row ring takes it all.
                    // Over a shard table most rows are another GPU's: a load over xGMI is a few us away, eight steps of 0.25 us are
                    // not -- where LDS allows (rows up to 8 KiB; with a table ring up to 4 KiB) the ring is sixteen deep.  (Not yet
                    // run across xGMI: the depth is by reasoning, the arithmetic does not depend on it.)
    static constexpr int value = (SHARDED && (TABLE ? J <= 1 : J <= 2)) ? 16 : (TABLE ? (J <= 2 ? 8 : (J <= 4 ? 4 : 2)) : (J <= 4 ? 8 : 4));
};

// Chains with a table: are BOTH addresses of a step's sample (data row, table row) resolved while staging and kept in LDS (the step then
// multiplies nothing: two 64-bit multiplies, eighteen scalar instructions, leave every step), or only the row index?  Always over a
// shard table (the step must not search it); on one allocation wherever the second address array (8 KiB) still fits the 160 KiB of
// LDS beside the rings -- everything but the 16 KiB-row Finito chains.  Round 5: the sharded SAGA chain, the same instructions but for
// this, ran 5-8 % FASTER than the unsharded one (fp64 d = 1024 0.479 against 0.507 us, fp32 d = 2048 0.443 against 0.483).
template <typename T, int J, int ALG, int NT, bool SHARDED>
constexpr bool chain_dma_stage_ptr()
{
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr int NW = NT / WAVE;
    constexpr int DEPTH = DmaDepth<J * NT / 256, HAS_TABLE, SHARDED>::value;
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO || ALG == CA_SVRGC);
    constexpr size_t with_ptr = (size_t)DEPTH * J * NT * 16 * 2 + 2 * (CHAIN_CHUNK + 2 * DEPTH) * sizeof(int64_t) +
                                CHAIN_CHUNK * sizeof(T) * (PER_SAMPLE_GAM ? 2 : 1) + CHAIN_CHUNK * sizeof(int) + 16 + 2 * NW * 2 * sizeof(T) +
                                (SHARDED ? SHARD_QW * sizeof(int64_t) : 0);
    return HAS_TABLE && (SHARDED || with_ptr <= 160 * 1024);
}

// NT threads (256 or 512): 32 KiB rows are shared by eight waves instead of four (a step costs ~0.38 us + ~0.06 us per
// 16-byte chunk a thread owns, but eight waves also pay more for the exchange: chain_launch.inc has the measurements).
// SHARDED: the rows live in several allocations (ChainArgs::sh*, ciao_ctx_set_shards): each step's row ADDRESS is resolved
// while staging and kept in LDS, table rows are addressed through the shard table.  A separate instantiation, so that the
// single-allocation chain keeps its instruction count (an always-present shard search cost it 0.07 us per SAGA step).
template <typename T, int J, int ALG, int LOSS, bool MASKED, int NT, bool SHARDED = false>
__global__ void __launch_bounds__(NT) chain_dma_kernel(ChainArgs<T> a_in)
{
    // The arguments are read THROUGH THE KERNEL-ARGUMENT SEGMENT (or, in a batch of chains, through this workgroup's own block of
    // ChainArgs::multi), field by field where they are used: hipcc loads every field of a by-value argument into scalar registers in
    // the entry block, where the fields only the staging or the final stores need stay live through the step loop and push 10-80
    // of them out to VGPR lanes (chain_ws_kernel: the same cure).  Both blocks are constant for the kernel's lifetime.
    (void)a_in;
    const ChainArgsK<T> &a = *chain_args_block<T>();
    constexpr int NW = NT / WAVE;
    static_assert(NW == 1 || NW == 4 || NW == 8, "one, four or eight waves");
    using V = typename VecOfC<T>::type;
    constexpr int VEC = 16 / sizeof(T);
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr int DEPTH = DmaDepth<J * NT / 256, HAS_TABLE, SHARDED>::value;   // by row bytes (J*NT*16), whatever the thread count
    constexpr int CH = CHAIN_CHUNK;
    constexpr bool SVRG_ANY = (ALG == CA_SVRG || ALG == CA_SVRGC);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO || ALG == CA_SVRGC);   // s_g staging
    // MASKED (rows shorter than J*4096 bytes): the table stores of chunk groups beyond the row are predicated off and may
    // not issue at all, so only the (always issued, address-clamped) LDS-DMA loads are counted -- stricter waits, still safe
    // (one wave issues four waves' worth of operations per step: counting its stores as well would pass the 6-bit counter)
    constexpr int OPS_PER_STEP = HAS_TABLE ? ((MASKED || NW == 1) ? 2 * J : 3 * J) : J;
    // PIPE: the LDS reads of step s+1 (its ring slot and its staged scalars) are issued at the top of step s and land
    // while step s reduces its dot product, so only one LDS round trip (the 4-partial exchange) stays on the
    // dependent path.  It costs one step of DMA lead, hence only with DEPTH >= 4.
    // (eight waves -- 32 KiB rows, 256 registers per wave -- spill 50-190 registers with the two register sets and are still the
    // fastest of what was measured: fp64 d = 4096 0.570 us per SVRG update against 0.590 without PIPE (no spill) and 0.755 on four
    // waves with twice the chunks per thread, profiles/r04_chain_32k_ab.txt)
    constexpr bool PIPE = DEPTH >= 4;
    constexpr int WAIT_N = (PIPE ? DEPTH - 2 : DEPTH - 1) * OPS_PER_STEP;
    // four waves: the ring's refill is issued in the shadow of the exchange (between the partial reads' issue and their wait) --
    // unless it is eight DMA instructions (table + row of 16 KiB): those outlast the shadow and are better left at the end of the
    // step (Finito r = 1 at d = 4096 fp32: 0.87 us there, 1.04 in the shadow)
    constexpr bool SHADOW_REFILL = (NW == 4) && (!HAS_TABLE || J <= 2);
    constexpr int ROW_BYTES = J * NT * 16;
    static_assert(!SHARDED || ALG == CA_SVRG || ALG == CA_SAGA, "only the SVRG and SAGA chains run over a shard table");
    // Chains without a table (SVRG, LFinito) need a step's row only as an ADDRESS: the staged entry is the row's address
    // itself (resolved while staging, with full parallelism), which takes the 64-bit multiply -- nine scalar instructions -- out
    // of every step.  Chains with a table need the sample's identity as well (table row, hazard flags): STAGE_PTR (over a shard
    // table always; on one allocation wherever LDS has room, chain_dma_stage_ptr) stages BOTH addresses -- s_row holds the TABLE
    // row's address (which identifies the sample as well as its index does: the hazard flags compare it) and s_ptr the data row's, so
    // that a step neither multiplies nor searches the shard table; the 16 KiB-row Finito chains keep the index and compute both
    // addresses in the step.
    constexpr bool PTR_IN_ROW = !HAS_TABLE;
    constexpr bool STAGE_PTR = chain_dma_stage_ptr<T, J, ALG, NT, SHARDED>();
    static_assert(CH % DEPTH == 0 && DEPTH % 2 == 0, "ring slots must line up with chunk starts; ping-pong needs even DEPTH");

    // one dynamic LDS block, carved by hand (16-byte aligned pieces):
    //   ringA[DEPTH][ROW_BYTES] | ringT[DEPTH][ROW_BYTES] (table algs) | s_row | s_ptr | s_b | s_g | s_stale | red
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *ringA = dsm;
    unsigned char *ringT = ringA + DEPTH * ROW_BYTES;
    unsigned char *cur = ringT + (HAS_TABLE ? DEPTH * ROW_BYTES : 0);
    int64_t *s_row = reinterpret_cast<int64_t *>(cur);
    cur += (CH + 2 * DEPTH) * sizeof(int64_t);
    // the data row's ADDRESS per step, resolved while staging (one multiply, or the shard search on a row-sharded problem):
    // the step itself only reads it back, so a remote (peer-mapped) row costs the step nothing extra to address
    const unsigned char **s_ptr = reinterpret_cast<const unsigned char **>(cur);
    cur += (STAGE_PTR ? CH + 2 * DEPTH : 0) * sizeof(int64_t);
    T *s_b = reinterpret_cast<T *>(cur);
    cur += CH * sizeof(T);
    T *s_g = reinterpret_cast<T *>(cur);
    cur += (PER_SAMPLE_GAM ? CH : 0) * sizeof(T);
    int *s_stale = reinterpret_cast<int *>(cur);
    cur += (HAS_TABLE ? CH : 0) * sizeof(int);
    cur += (16 - (reinterpret_cast<uintptr_t>(cur) & 15)) & 15;
    T(*red)[NW][2] = reinterpret_cast<T(*)[NW][2]>(cur);
    cur += 2 * NW * 2 * sizeof(T);
    int64_t *s_sh = reinterpret_cast<int64_t *>(cur);   // SHARDED: the shard table (shard_resolve)

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;
    if constexpr (SHARDED) shard_table_to_lds<T>(s_sh, tid);   // (the staging's first __syncthreads orders it before its readers)
    const uint32_t ringA_off = (uint32_t)(uintptr_t)ringA;   // LDS byte offsets (low 32 bits of the flat address)
    const uint32_t ringT_off = (uint32_t)(uintptr_t)ringT;
    // this wave's 1 KiB pieces of the ring slots start here: ONE scalar register per ring (glds16_at)
    const uint32_t ringA_w = sgpr_pin(ringA_off + (uint32_t)wib * 1024u);
    const uint32_t ringT_w = sgpr_pin(ringT_off + (uint32_t)wib * 1024u);
    // what the step loop reads of the argument block (sgpr_pin); everything else is read where it is used
    const int64_t nsteps = sgpr_pin(a.nsteps);
    const T gamma = (SVRG_ANY || ALG == CA_SAGA) ? sgpr_pin(a.gamma) : T(0);
    const T lam = (LOSS == CIAO_LOSS_LOGISTIC) ? T(0) : sgpr_pin(a.lam);
    const T invN = (ALG == CA_SVRG || ALG == CA_SVRGC) ? T(0) : sgpr_pin(a.invN);
    const T hat_gamma = (ALG == CA_FINITO || ALG == CA_LFINITO) ? sgpr_pin(a.hat_gamma) : T(0);
    const int64_t batch = (ALG == CA_FINITO || ALG == CA_LFINITO) ? sgpr_pin(a.batch) : 0;
    const bool sag = (ALG == CA_SAGA) && sgpr_pin(a.sag) != 0;
    // rows and table rows by index (chains with a table on ONE allocation): base pointers and strides
    const T *const Abase = (!PTR_IN_ROW && !STAGE_PTR) ? sgpr_pin_global(a.A) : nullptr;
    const int64_t ld = (!PTR_IN_ROW && !STAGE_PTR) ? sgpr_pin(a.ld) : 0;
    T *const tbase = (HAS_TABLE && !STAGE_PTR) ? sgpr_pin_global(a.table) : nullptr;
    const int64_t dtab = (HAS_TABLE && !STAGE_PTR) ? sgpr_pin(a.d) : 0;

    // chunk ownership: thread t owns 16-byte chunks t + 256*j; with MASKED those at or beyond the row's end are dead (their
    // state stays zero, their loads are redirected to chunk 0 and discarded, their stores are predicated off)
    const int64_t nchunks = d / VEC;
    T box_lo = a.g.lo, box_hi = a.g.hi;   // as VALUES (a select between "&a.g.lo" and the bound vector would keep `a` in memory)
    T gam_u = a.gam_uniform;              // ... likewise (fp64: 16 bytes of scratch and a flat load per staged step otherwise)
    asm volatile("" : "+v"(box_lo), "+v"(box_hi), "+v"(gam_u));
    bool ok[J];
    int64_t cl[J];   // chunk to address: own chunk, or 0 when dead
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = tid + (int64_t)j * NT;
        ok[j] = !MASKED || c < nchunks;
        cl[j] = ok[j] ? c : 0;
    }
    // iterate state, in 16-byte chunks
    V av[J], p[J], zf[J], zs[J], plo[J], phi[J];
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
    const bool hasbox = (a.g.kind == CIAO_PROX_BOX);   // wave-uniform: one branch per step selects the clamp-free prox
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = cl[j];
        av[j] = reinterpret_cast<const V *>(a.av)[c];
        if (SVRG_ANY) {
            p[j] = reinterpret_cast<const V *>(a.w)[c];
            zs[j] = reinterpret_cast<const V *>(a.z)[c];
        } else {
            p[j] = reinterpret_cast<const V *>(a.z)[c];
            zs[j] = V(T(0));
        }
        zf[j] = TWO ? reinterpret_cast<const V *>(a.zf)[c] : V(T(0));
        if (!ok[j]) av[j] = p[j] = zs[j] = zf[j] = V(T(0));
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            plo[j][v] = -INFINITY;
            phi[j][v] = INFINITY;
            if (a.g.kind == CIAO_PROX_BOX && ok[j]) {   // dead chunks keep -inf/+inf: their zeros stay zeros
                plo[j][v] = a.g.lo_vec ? a.g.lo_vec[c * VEC + v] : box_lo;
                phi[j][v] = a.g.hi_vec ? a.g.hi_vec[c * VEC + v] : box_hi;
            }
        }
    }

    // SVRG: av is constant over the inner cycle, so gamma*av is hoisted out of the chain
    V gav[J];
#pragma unroll
    for (int j = 0; j < J; ++j) gav[j] = gamma * av[j];

    // issue the DMA of row r (at address ap; STAGE_PTR: r IS its table row's address) into ring slot u: J (+J) wave-instructions of 1 KiB each
    // const_u: the slot number is a compile-time constant where the call is inlined (the unrolled step groups): the LDS destination
    // is then the wave's base + an immediate (glds16_at); the one-off first filling of the ring runs as a loop over the slots
    auto refill = [&](auto const_u, int u, int64_t r, const unsigned char *ap) {
        constexpr bool CU = decltype(const_u)::value;
        // table-free chains: base in SGPRs + 32-bit lane offset (-3 % per SVRG step); with a table ring beside it the plain
        // 64-bit VGPR addresses schedule better (measured: SAGA 0.416 us against 0.422 / 0.430 with the scalar base)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int off = (u * J + j) * NW * 1024;
            if constexpr (HAS_TABLE) {
                if constexpr (CU) glds16_at(ap + cl[j] * 16, ringA_w, off);
                else glds16(ap + cl[j] * 16, ringA_w + (uint32_t)off);
            } else {
                if constexpr (CU) glds16s_at(ap, (uint32_t)cl[j] * 16u, ringA_w, off);
                else glds16s(ap, (uint32_t)cl[j] * 16u, ringA_w + (uint32_t)off);
            }
        }
        if (HAS_TABLE) {
            const unsigned char *sp = STAGE_PTR ? reinterpret_cast<const unsigned char *>((uintptr_t)r)
                                                : reinterpret_cast<const unsigned char *>(tbase + r * dtab);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int off = (u * J + j) * NW * 1024;
                if constexpr (CU) glds16_at(sp + cl[j] * 16, ringT_w, off);
                else glds16(sp + cl[j] * 16, ringT_w + (uint32_t)off);
            }
        }
    };

    // the table row of the sample a step knows as `row`: its index, or (STAGE_PTR) the row's address itself (global memory, this
    // GPU's or a peer's: said so, or the stores through it are FLAT ones)
    auto trow_of = [&](int64_t row) -> T * {
        if constexpr (STAGE_PTR) return (T *)(__attribute__((address_space(1))) T *)(uintptr_t)row;
        else return tbase + row * dtab;
    };

    // everything step s needs from LDS: its ring slot and its staged scalars (two register sets, ping-pong by step parity)
    struct StepIn {
        V ar[J], sr[J];
        int64_t row, row_n;
        const unsigned char *ptr_n;
        T bi, gi;
        int stale;
    };
    StepIn in[2];
    auto fetch = [&](StepIn &x, int u, int s) {   // plain LDS reads; the caller has retired slot u's DMA
#pragma unroll
        for (int j = 0; j < J; ++j) {
            x.ar[j] = *reinterpret_cast<const V *>(ringA + (((u * J + j) * NW + wib) * 64 + lane) * 16);
            if (HAS_TABLE) x.sr[j] = *reinterpret_cast<const V *>(ringT + (((u * J + j) * NW + wib) * 64 + lane) * 16);
            // (MASKED: the dead chunks are zeroed by mask_dead() when the step that USES them begins -- zeroing them here would
            // make the wave wait for these reads right after issuing them, a whole LDS latency at the top of every step)
        }
        x.row = s_row[DEPTH + s];
        x.row_n = s_row[DEPTH + s + DEPTH];
        x.ptr_n = STAGE_PTR ? s_ptr[DEPTH + s + DEPTH] : nullptr;
        x.bi = s_b[s];
        x.gi = PER_SAMPLE_GAM ? s_g[s] : T(1);
        x.stale = HAS_TABLE ? s_stale[s] : 0;
    };

    auto mask_dead = [&](StepIn &x) {   // rows shorter than the threads' reach: what the ring holds for the dead chunks is discarded
        if constexpr (MASKED) {
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (!ok[j]) {
                    x.ar[j] = V(T(0));
                    if (HAS_TABLE) x.sr[j] = V(T(0));
                }
        }
    };

    int par = 0;
    int64_t inb = 0;
    for (int64_t base = 0; base < nsteps; base += CH) {
        const int nch = (int)((nsteps - base) < CH ? (nsteps - base) : CH);

        // ---- stage this chunk's gathers in LDS (ordinary loads: the compiler drains the queue here, once per chunk) ----
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
            const T *arow, *bp;
            int64_t ident = r;   // what the steps and the hazard flags know the sample by
            if (SHARDED) {   // global row -> its shard's memory (which may be another GPU's)
                const ShardRow<T> sr = shard_resolve<T>(s_sh, a.nshards, r, a.ld, a.d);
                arow = sr.arow;
                bp = sr.bp;
                if (STAGE_PTR) ident = (int64_t)(uintptr_t)sr.trow;
            } else {
                arow = a.A + r * a.ld;
                bp = a.b ? a.b + r : nullptr;
                if (STAGE_PTR) ident = (int64_t)(uintptr_t)(a.table + r * a.d);
            }
            s_row[DEPTH + e] = PTR_IN_ROW ? (int64_t)(uintptr_t)arow : ident;
            if (STAGE_PTR) s_ptr[DEPTH + e] = reinterpret_cast<const unsigned char *>(arow);
            if (e < nch) {
                s_b[e] = bp ? *bp : T(0);
                if (PER_SAMPLE_GAM) {
                    const T gv = a.gam ? a.gam[r] : gam_u;
                    // SVRG with cached row dots: what the step needs of a_i'z_full is the link-function coefficient at it,
                    // which does not depend on the chain -- evaluated HERE, 256 steps at a time, instead of once per step on
                    // the chain's only wave per SIMD (for the logistic loss that is an exp and a division per step)
                    s_g[e] = (ALG == CA_SVRGC) ? grad_coef_t<T, LOSS>(gv, bp ? *bp : T(0), lam).coef() : gv;
                }
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
#pragma unroll 1
            for (int u = 0; u < DEPTH; ++u) {   // once per launch: a loop (unrolled, its DEPTH sets of LDS addresses cost scalar registers)
                const int64_t r0 = uniform64(s_row[DEPTH + u]);
                refill(std::false_type{}, u, r0,
                       PTR_IN_ROW ? reinterpret_cast<const unsigned char *>((uintptr_t)r0)
                       : STAGE_PTR ? reinterpret_cast<const unsigned char *>(uniform64((int64_t)(uintptr_t)s_ptr[DEPTH + u]))
                                   : reinterpret_cast<const unsigned char *>(Abase + r0 * ld));
            }
        }
        wait_vmcnt<0>();          // ring fully landed: the counted waits below assume the steady-state op sequence
        drain_vmcnt_visible();    // ... and hipcc knows that the state / staging loads are retired too
        if (PIPE) fetch(in[0], 0, 0);

        // ---- the dependent chain ----------------------------------------------------------------------------------------
        // DEPTH steps (one ring revolution), in four versions selected ONCE per group instead of once per step: with / without
        // the IndBox clamp (HB), and with / without the end-of-chunk checks (CHK: a group whose every step exists and has a
        // successor in this chunk needs none -- all but the last group of a chunk).  The per-step tests and branches were
        // a sixth of the step's instructions.
        auto group = [&](auto hb_tag, auto chk_tag, auto sag_tag, const int s0) {
            constexpr bool HB = decltype(hb_tag)::value;
            constexpr bool CHK = decltype(chk_tag)::value;
            constexpr bool SAG = decltype(sag_tag)::value;   // SAGA chains only: SAG steps with the new average (a select per element otherwise)
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int s = s0 + u;
                if (CHK && s >= nch) return;
                StepIn &x = in[PIPE ? (u & 1) : 0];
                if (PIPE) {
                    mask_dead(x);                // read one step ago: long here
                    if (!CHK || s + 1 < nch) {   // next step's inputs: retire its DMA (one step less lead), read, do not wait
                        wait_vmcnt<WAIT_N>();
                        fetch(in[(u + 1) & 1], (u + 1) % DEPTH, s + 1);
                    }
                } else {
                    wait_vmcnt<WAIT_N>();
                    fetch(x, u, s);
                    mask_dead(x);
                }
                const int64_t row = uniform64(x.row);
                const int64_t row_n = uniform64(x.row_n);
                const unsigned char *ptr_n = PTR_IN_ROW ? reinterpret_cast<const unsigned char *>((uintptr_t)row_n)
                                             : STAGE_PTR ? reinterpret_cast<const unsigned char *>(uniform64((int64_t)(uintptr_t)x.ptr_n))
                                                         : reinterpret_cast<const unsigned char *>(Abase + row_n * ld);
                const T bi = x.bi;

                if (ALG == CA_LFINITO && inb == 0) {   // Finito_LFinito.jl:92  z = prox(av)
                    const T gl = hat_gamma * plam;
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) p[j][v] = hasbox ? prox_bf(av[j][v], gl, plo[j][v], phi[j][v]) : prox_l1(av[j][v], gl);
                }
                if (HAS_TABLE && __builtin_amdgcn_readfirstlane(x.stale)) {
                    // an intervening step rewrote this table row after its DMA was issued: re-read it from memory (this
                    // very thread stored these bytes, so program order makes them visible)
                    const V *sp = reinterpret_cast<const V *>(trow_of(row));
#pragma unroll
                    for (int j = 0; j < J; ++j) x.sr[j] = ok[j] ? sp[cl[j]] : V(T(0));
                    drain_vmcnt_visible();   // retire it HERE, or hipcc puts a draining vmcnt(0) on the common path
                }

                // Finito / LFinito: the per-sample stepsize's two scalars -- hat_gamma / gamma_i (a division: ten instructions) and
                // gamma_i / N -- need nothing of this step: written HERE, in front of the wave sum, they fill the wait states of its
                // DPP stages and the exchange's first shadow instead of standing behind the exchange
                T pre_rr = T(0), pre_gn = T(0);
                if (ALG == CA_FINITO || ALG == CA_LFINITO) {
                    pre_rr = hat_gamma / x.gi;
                    pre_gn = x.gi * invN;
                }
                T d1 = T(0), d2 = T(0);
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        d1 = fmad(x.ar[j][v], p[j][v], d1);
                        if (TWO) d2 = fmad(x.ar[j][v], zf[j][v], d2);
                    }
                V q1[J], q2[J];
                if constexpr (NW == 1) {
                    // ONE wave owns the whole row: the reduced dot is broadcast from lane 63 through an SGPR, and the LDS
                    // exchange (write, lgkmcnt(0), barrier, read: the largest piece of a four-wave step) does not exist.
                    // The sums go stage by stage -- with two dot products two independent dependency chains, each filling the
                    // other's latencies (one after the other, what hipcc makes of two calls, they cost twice six dependent
                    // stages) -- and between the stages, instead of wait states, what the update needs that does not depend on
                    // the dots: SVRG's q2 = w - gamma*av and q1 = gamma*a_i, element by element.  The empty asm statements
                    // keep that order; the additions are wave_sum_lane63's, in its order: bitwise the same sums.
                    int nq = 0;   // elements of (q2, q1) placed so far (compile-time after unrolling)
                    constexpr int NQ = SVRG_ANY ? 2 * J * VEC : 0, PER = (NQ + 5) / 6;
                    auto fill = [&](int n) {
                        for (int e = 0; e < n && nq < NQ; ++e, ++nq) {
                            const int k = nq >> 1, j = k / VEC, v = k % VEC;
                            if (nq & 1) {
                                q1[j][v] = gamma * x.ar[j][v];
                                asm volatile("" : "+v"(q1[j][v]));
                            } else {
                                q2[j][v] = p[j][v] - gav[j][v];
                                asm volatile("" : "+v"(q2[j][v]));
                            }
                        }
                    };
                    auto stage = [&](auto f) {
                        d1 = f(d1);
                        asm volatile("" : "+v"(d1));
                        if (TWO) {
                            d2 = f(d2);
                            asm volatile("" : "+v"(d2));
                        }
                        fill(PER);
                    };
                    stage([](T v) { return v + dpp_mov<0xB1>(v); });
                    stage([](T v) { return v + dpp_mov<0x4E>(v); });
                    stage([](T v) { return v + dpp_mov<0x141>(v); });
                    stage([](T v) { return v + dpp_mov<0x140>(v); });
                    stage([](T v) { return v + dpp_rows<0x142, 0xA>(v); });
                    stage([](T v) { return v + dpp_rows<0x143, 0xC>(v); });
                    fill(NQ);
                    d1 = readlane(d1, WAVE - 1);
                    if (TWO) d2 = readlane(d2, WAVE - 1);
                } else {
                d1 = wave_sum_lane63(d1);   // bitwise the same total, in lane 63 only: no v_readlane / scalar round trip
                if (TWO) d2 = wave_sum_lane63(d2);
                if (lane == WAVE - 1) {
                    red[par][wib][0] = d1;
                    if (TWO) red[par][wib][1] = d2;
                }
                }
                // work that does not need the dot product goes between the LDS write and the barrier, where it overlaps the
                // other waves' arrival:  temp = gamma*(a*dc - av) + w  =  (gamma*a)*dc + (w - gamma*av)
                if (SVRG_ANY && NW != 1) {
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        q1[j] = gamma * x.ar[j];
                        q2[j] = p[j] - gav[j];
                        // four waves: computed HERE, before the barrier (the empty asm is volatile and stays in front of the
                        // volatile wait below; hipcc otherwise sinks half of these eight instructions behind the exchange,
                        // onto the dependent path)
                        if constexpr (NW == 4) asm volatile("" : "+v"(q1[j]), "+v"(q2[j]));
                    }
                }
                if (ALG == CA_FINITO || ALG == CA_LFINITO) {
                    if constexpr (NW != 1) asm volatile("" : "+v"(pre_rr), "+v"(pre_gn));   // (computed above, complete by here)
                }
                if constexpr (NW == 4) {
                    // The exchange with its reads in two halves, and in between -- while the partials travel from LDS, about
                    // ninety cycles in which this wave has nothing else to do -- everything of the step that does not need
                    // the dot product: the DMA of the row DEPTH steps ahead (its slot's row is in registers since the last
                    // step) and SVRG's `z += w` (SVRG_basic.jl:81) for the iterate of the PREVIOUS step.  tools/micro/xchg_lab.hip:
                    // two dozen independent instructions cost 140 cycles after the exchange, 46 in its shadows.
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // raw barrier: must not drain the DMA queue
                    const uint32_t raddr = (uint32_t)(uintptr_t)&red[par][0][0];
                    constexpr bool SINGLE64 = (sizeof(T) == 8 && !TWO);
                    V rv[SINGLE64 ? 2 : (int)(NW * 2 * sizeof(T) / 16)];
                    if constexpr (SINGLE64) xchg_issue_single64(raddr, rv); else xchg_issue(raddr, rv);
                    if (SHADOW_REFILL) refill(std::true_type{}, u, row_n, ptr_n);
                    if (SVRG_ANY) {
                        if (u > 0 || s0 > 0 || base > 0) {   // compile-time true except in the first step of a ring revolution
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                zs[j] += p[j];
                                asm volatile("" : "+v"(zs[j]));
                            }
                        }
                    }
                    xchg_wait(rv);
                    // element k of the parity's slots [wave][2]: SINGLE64 holds {w0, w1}, {w2, w3}; otherwise the slots as they lie
                    auto val = [&](int w, int c) -> T {
                        if constexpr (SINGLE64) return rv[w / 2][w % 2];
                        const int k = w * 2 + c;
                        return rv[k / VEC][k % VEC];
                    };
                    {
                        T lo = val(0, 0) + val(1, 0), hi = val(2, 0) + val(3, 0);
                        // fp64: pin the two pair sums right behind the LDS read (two-dot SVRG step 0.351 -> 0.327 us; fp32 is better
                        // left to the compiler, 0.262 vs 0.268 with the pin)
                        if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(lo), "+v"(hi));
                        d1 = lo + hi;
                    }
                    if (TWO) d2 = (val(0, 1) + val(1, 1)) + (val(2, 1) + val(3, 1));
                } else if constexpr (NW > 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // raw barrier: must not drain the DMA queue
                {
                    T lo = red[par][0][0] + red[par][1][0], hi = red[par][2][0] + red[par][3][0];
                    if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(lo), "+v"(hi));
                    d1 = lo + hi;
                }
                if (TWO) d2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
                if constexpr (NW == 8) {   // fixed association order: two groups of four
                    d1 += (red[par][4][0] + red[par][5][0]) + (red[par][6][0] + red[par][7][0]);
                    if (TWO) d2 += (red[par][4][1] + red[par][5][1]) + (red[par][6][1] + red[par][7][1]);
                }
                }
                par ^= 1;

                // everything after the exchange, instantiated twice: with the IndBox clamp and without it (g = Zero / NormL1),
                // selected by ONE workgroup-uniform branch per step instead of a select per coordinate
                {
                    const GradCoef<T> gp = grad_coef_t<T, LOSS>(d1, bi, lam);
                    if (SVRG_ANY) {                                                  // SVRG_basic.jl:74-81
                        // a_i'z_full: recomputed (CA_SVRG) or the value the last full pass stored for this row (CA_SVRGC)
                        // the coefficient at a_i'z_full: staged ready-made (CA_SVRGC), or from this step's second dot product
                        const T cz = (ALG == CA_SVRGC) ? x.gi : grad_coef_t<T, LOSS>(d2, bi, lam).coef();
                        const T gl = gamma * plam;
                        const T dc = cz - gp.coef();
    #pragma unroll
                        for (int j = 0; j < J; ++j)
    #pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                const T t = fmad(q1[j][v], dc, q2[j][v]);
                                p[j][v] = HB ? prox_bf(t, gl, plo[j][v], phi[j][v]) : prox_l1(t, gl);
                                if (NW != 4) zs[j][v] += p[j][v];   // four waves: in the next step's exchange shadow
                            }
                    } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                        V *sp = reinterpret_cast<V *>(trow_of(row));
                        const T gl = gamma * plam;
                        const T cp = gp.coef();
                        const T ngam = -gamma;
    #pragma unroll
                        for (int j = 0; j < J; ++j) {
                            V gnv;
    #pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                const T gn = x.ar[j][v] * cp;
                                const T del = gn - x.sr[j][v];
                                // SAGA steps with (g_new - s_i + av_old), SAG with av_new (SAGA_basic.jl:58-62)
                                const T avn = fmad(del, invN, av[j][v]);
                                const T wv = fmad(ngam, SAG ? avn : del + av[j][v], p[j][v]);
                                av[j][v] = avn;
                                p[j][v] = HB ? prox_bf(wv, gl, plo[j][v], phi[j][v]) : prox_l1(wv, gl);
                                gnv[v] = gn;
                            }
                            if (ok[j]) sp[cl[j]] = gnv;
                        }
                    } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                        const T ncc = -pre_gn * gp.coef();   // t = z - (gamma_i/N) * c * a
                        const T rr = pre_rr;                  // hat_gamma / gamma_i
                        V *sp = reinterpret_cast<V *>(trow_of(row));
    #pragma unroll
                        for (int j = 0; j < J; ++j) {
                            V tv;
    #pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                tv[v] = fmad(ncc, x.ar[j][v], p[j][v]);
                                av[j][v] = fmad(tv[v] - x.sr[j][v], rr, av[j][v]);
                            }
                            if (ok[j]) sp[cl[j]] = tv;
                        }
                        if (inb + 1 == batch || (base + s + 1) == nsteps) {
                            const T gl = hat_gamma * plam;
    #pragma unroll
                            for (int j = 0; j < J; ++j)
    #pragma unroll
                                for (int v = 0; v < VEC; ++v) p[j][v] = HB ? prox_bf(av[j][v], gl, plo[j][v], phi[j][v]) : prox_l1(av[j][v], gl);
                        }
                    } else {                                                         // Finito_LFinito.jl:93-98
                        const GradCoef<T> gzf = grad_coef_t<T, LOSS>(d2, bi, lam);
                        const T dc = (hat_gamma * invN) * (gzf.coef() - gp.coef());
                        const T rr = pre_rr;                  // hat_gamma / gamma_i
    #pragma unroll
                        for (int j = 0; j < J; ++j)
    #pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                av[j][v] = fmad(x.ar[j][v], dc, av[j][v]);
                                av[j][v] = fmad(rr, p[j][v] - zf[j][v], av[j][v]);
                            }
                    }
                }

                if (++inb == batch) inb = 0;
                // one or eight waves: the refill at the end of the step (four waves: in the exchange's shadow, above -- a table row
                // it fetches that this step is about to rewrite is flagged stale either way: the flag compares DEPTH steps back)
                if (!SHADOW_REFILL) refill(std::true_type{}, u, row_n, ptr_n);
            }
        };
        // the run-time flags become compile-time tags of the group (SAG only exists for the SAGA chain)
        auto pick_sag = [&](auto hb_tag, auto chk_tag, const int s0) {
            if constexpr (ALG == CA_SAGA) {
                if (sag)
                    group(hb_tag, chk_tag, std::true_type{}, s0);
                else
                    group(hb_tag, chk_tag, std::false_type{}, s0);
            } else {
                group(hb_tag, chk_tag, std::false_type{}, s0);
            }
        };
        for (int s0 = 0; s0 < nch; s0 += DEPTH) {
            if (s0 + DEPTH < nch) {
                if (hasbox)
                    pick_sag(std::true_type{}, std::false_type{}, s0);
                else
                    pick_sag(std::false_type{}, std::false_type{}, s0);
            } else {
                if (hasbox)
                    pick_sag(std::true_type{}, std::true_type{}, s0);
                else
                    pick_sag(std::false_type{}, std::true_type{}, s0);
            }
        }
    }
    wait_vmcnt<0>();   // nothing may still be writing LDS when the workgroup retires
    if (SVRG_ANY && NW == 4 && nsteps > 0) {   // the last step's `z += w`
#pragma unroll
        for (int j = 0; j < J; ++j) zs[j] += p[j];
    }

#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (!ok[j]) continue;
        const int64_t c = cl[j];
        if (SVRG_ANY) {
            reinterpret_cast<V *>(a.w)[c] = p[j];
            reinterpret_cast<V *>(a.z)[c] = zs[j];
        } else {
            reinterpret_cast<V *>(a.z)[c] = p[j];
            reinterpret_cast<V *>(a.av)[c] = av[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Complex chains on the LDS-DMA ring.  chain_cplx_reg_kernel requests the next row ONE step ahead, so one
// HBM miss per step is exposed (1.6 us per SVRG update at 512 complex fp64 entries against 0.27 us for the real chain).  Here
// the rows (and SAGA / Finito table rows) travel exactly as in chain_dma_kernel -- LDS-DMA DEPTH steps ahead, hand-counted
// vmcnt waits, indices / b_i / gamma_i / hazard flags staged 1024 steps at a time -- and only the arithmetic is complex: thread t
// owns the 16-byte chunks t + 256 j of every (re, im)-interleaved vector (one complex entry per chunk in fp64, two in fp32), the
// complex dot product(s) are two (four) real wave sums and one exchange of 2 (4) values per wave, formulas and operation
// order those of chain_cplx_reg_kernel (bitwise the same results: tests).  Rows of whole 16-byte chunks up to 16 KiB.
// ------------------------------------------------------------------------------------------------------------------
constexpr int CDMA_CHUNK = 512;

template <typename T, int J, int ALG, bool MASKED>
__global__ void __launch_bounds__(CHAIN_NT) chain_cdma_kernel(ChainArgs<T> a_by_value)
{
    // the arguments through the kernel-argument segment, field by field where they are used (chain_dma_kernel, and why)
    (void)a_by_value;
    ChainArgsK<T> &a = *(ChainArgsK<T> *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NW = CHAIN_NW, NT = CHAIN_NT;
    using V = typename VecOfC<T>::type;
    constexpr int VEC = 16 / sizeof(T), PC = VEC / 2;          // reals / complex entries per chunk
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool TWO = (ALG == CA_SVRG || ALG == CA_LFINITO);
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO);
    constexpr int DEPTH = DmaDepth<J, HAS_TABLE>::value;
    constexpr int CH = CDMA_CHUNK;   // (half the real chains' chunk: b_i is a pair here, and two 64 KiB rings leave 32 KiB for the staging)
    constexpr int OPS_PER_STEP = HAS_TABLE ? (MASKED ? 2 * J : 3 * J) : J;
    // PIPE: the LDS reads of step s+1's ring slot are issued at the top of step s (one step less DMA lead), so that they have
    // landed when step s+1 begins instead of being waited for right after their issue (chain_dma_kernel does the same)
    // (eight waves -- 32 KiB rows, 256 registers per wave -- spill 50-190 registers with the two register sets and are still the
    // fastest of what was measured: fp64 d = 4096 0.570 us per SVRG update against 0.590 without PIPE (no spill) and 0.755 on four
    // waves with twice the chunks per thread, profiles/r04_chain_32k_ab.txt)
    constexpr bool PIPE = DEPTH >= 4;
    constexpr int WAIT_N = (PIPE ? DEPTH - 2 : DEPTH - 1) * OPS_PER_STEP;
    constexpr int ROW_BYTES = J * NT * 16;
    static_assert(CH % DEPTH == 0 && WAIT_N <= 63, "ring slots line up with chunk starts; vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *ringA = dsm;
    unsigned char *ringT = ringA + DEPTH * ROW_BYTES;
    unsigned char *cur = ringT + (HAS_TABLE ? DEPTH * ROW_BYTES : 0);
    int64_t *s_row = reinterpret_cast<int64_t *>(cur);
    cur += (CH + 2 * DEPTH) * sizeof(int64_t);
    T *s_b = reinterpret_cast<T *>(cur);           // (re, im) of b_i per step
    cur += 2 * CH * sizeof(T);
    T *s_g = reinterpret_cast<T *>(cur);
    cur += (PER_SAMPLE_GAM ? CH : 0) * sizeof(T);
    int *s_stale = reinterpret_cast<int *>(cur);
    cur += (HAS_TABLE ? CH : 0) * sizeof(int);
    cur += (16 - (reinterpret_cast<uintptr_t>(cur) & 15)) & 15;
    T(*red)[NW][4] = reinterpret_cast<T(*)[NW][4]>(cur);

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d = a.d;
    const uint32_t ringA_off = (uint32_t)(uintptr_t)ringA, ringT_off = (uint32_t)(uintptr_t)ringT;
    const uint32_t ringA_w = sgpr_pin(ringA_off + (uint32_t)wib * 1024u), ringT_w = sgpr_pin(ringT_off + (uint32_t)wib * 1024u);
    // what the step loop reads of the argument block (sgpr_pin); everything else is read where it is used
    const int64_t nsteps = sgpr_pin(a.nsteps);
    const T gamma = (ALG == CA_SVRG || ALG == CA_SAGA) ? sgpr_pin(a.gamma) : T(0);
    const T lam = sgpr_pin(a.lam);
    const T invN = (ALG == CA_SVRG) ? T(0) : sgpr_pin(a.invN);
    const T hat_gamma = PER_SAMPLE_GAM ? sgpr_pin(a.hat_gamma) : T(0);
    const int64_t batch = PER_SAMPLE_GAM ? sgpr_pin(a.batch) : 0;
    const bool sag = (ALG == CA_SAGA) && sgpr_pin(a.sag) != 0;
    const T glam = sgpr_pin(a.g.lam);
    const T *const Abase = sgpr_pin_global(a.A);
    const int64_t ld = sgpr_pin(a.ld);
    T *const tbase = HAS_TABLE ? sgpr_pin_global(a.table) : nullptr;
    const int64_t nchunks = d / VEC;
    bool ok[J];
    int64_t cl[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = tid + (int64_t)j * NT;
        ok[j] = !MASKED || c < nchunks;
        cl[j] = ok[j] ? c : 0;
    }
    T *pmem = (ALG == CA_SVRG) ? a.w : a.z;
    const bool l1 = (a.g.kind == CIAO_PROX_L1_COMPLEX);
    auto proxc = [&](T tau, T vr, T vi, T &yr, T &yi) {
        if (l1) {
            prox_cpair_chain(tau * glam, vr, vi, yr, yi);
        } else {
            yr = vr;
            yi = vi;
        }
    };
    V av[J], p[J], q[J], zs[J];          // q: z_full (SVRG, LFinito); zs: the SVRG accumulator z
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = cl[j];
        av[j] = reinterpret_cast<const V *>(a.av)[c];
        p[j] = reinterpret_cast<const V *>(pmem)[c];
        q[j] = TWO ? reinterpret_cast<const V *>(a.zf)[c] : V(T(0));
        zs[j] = (ALG == CA_SVRG) ? reinterpret_cast<const V *>(a.z)[c] : V(T(0));
        if (!ok[j]) av[j] = p[j] = q[j] = zs[j] = V(T(0));
    }
    // const_u: the slot number is a compile-time constant where the call is inlined (chain_dma_kernel's refill, and why)
    auto refill = [&](auto const_u, int u, int64_t r) {
        constexpr bool CU = decltype(const_u)::value;
        const unsigned char *ap = reinterpret_cast<const unsigned char *>(Abase + r * ld);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int off = (u * J + j) * NW * 1024;
            if constexpr (CU) glds16_at(ap + cl[j] * 16, ringA_w, off);
            else glds16(ap + cl[j] * 16, ringA_w + (uint32_t)off);
        }
        if (HAS_TABLE) {
            const unsigned char *sp = reinterpret_cast<const unsigned char *>(tbase + r * d);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int off = (u * J + j) * NW * 1024;
                if constexpr (CU) glds16_at(sp + cl[j] * 16, ringT_w, off);
                else glds16(sp + cl[j] * 16, ringT_w + (uint32_t)off);
            }
        }
    };
    int par = 0;
    int64_t inb = 0;
    for (int64_t base = 0; base < nsteps; base += CH) {
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
            s_row[DEPTH + e] = r;
            if (e < nch) {
                s_b[2 * e] = a.b[2 * r];
                s_b[2 * e + 1] = a.b[2 * r + 1];
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
#pragma unroll 1
            for (int u = 0; u < DEPTH; ++u) refill(std::false_type{}, u, uniform64(s_row[DEPTH + u]));   // once per launch: a loop
        }
        wait_vmcnt<0>();
        drain_vmcnt_visible();
        struct SlotIn {
            V ar[J], sr[J];
        };
        SlotIn in[2];
        auto fetch = [&](SlotIn &x, int u) {   // plain LDS reads of ring slot u; the caller has retired its DMA
#pragma unroll
            for (int j = 0; j < J; ++j) {
                x.ar[j] = *reinterpret_cast<const V *>(ringA + (((u * J + j) * NW + wib) * 64 + lane) * 16);
                if (HAS_TABLE) x.sr[j] = *reinterpret_cast<const V *>(ringT + (((u * J + j) * NW + wib) * 64 + lane) * 16);
            }
        };
        if (PIPE) fetch(in[0], 0);
        for (int s0 = 0; s0 < nch; s0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int s = s0 + u;
                if (s >= nch) break;
                SlotIn &x = in[PIPE ? (u & 1) : 0];   // (DEPTH is even: the buffers alternate across revolutions too)
                if (PIPE) {
                    if (s + 1 < nch) {
                        wait_vmcnt<WAIT_N>();                           // slot u+1's DMA (issued DEPTH-1 steps ago) has landed
                        fetch(in[(u + 1) & 1], (u + 1) % DEPTH);
                    }
                } else {
                    wait_vmcnt<WAIT_N>();                               // slot u's DMA (issued DEPTH steps ago) has landed
                    fetch(x, u);
                }
                V(&ar)[J] = x.ar;
                V(&sr)[J] = x.sr;
                if (MASKED) {   // dead chunks: what the ring holds for them is discarded
#pragma unroll
                    for (int j = 0; j < J; ++j)
                        if (!ok[j]) {
                            ar[j] = V(T(0));
                            if (HAS_TABLE) sr[j] = V(T(0));
                        }
                }
                const int64_t row = uniform64(s_row[DEPTH + s]);
                const int64_t row_n = uniform64(s_row[DEPTH + s + DEPTH]);
                const T br = s_b[2 * s], bi = s_b[2 * s + 1];
                const T gi = PER_SAMPLE_GAM ? s_g[s] : T(1);
                if (HAS_TABLE && __builtin_amdgcn_readfirstlane(s_stale[s])) {
                    const V *sp = reinterpret_cast<const V *>(tbase + row * d);
#pragma unroll
                    for (int j = 0; j < J; ++j) sr[j] = ok[j] ? sp[cl[j]] : V(T(0));
                    drain_vmcnt_visible();
                }
                if (ALG == CA_LFINITO && inb == 0) {                    // Finito_LFinito.jl:92  z = prox(av)
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int c = 0; c < PC; ++c) {
                            T yr, yi;
                            proxc(hat_gamma, av[j][2 * c], av[j][2 * c + 1], yr, yi);
                            p[j][2 * c] = yr;
                            p[j][2 * c + 1] = yi;
                        }
                }
                T s1r = T(0), s1i = T(0), s2r = T(0), s2i = T(0);
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int c = 0; c < PC; ++c) {
                        const T xr = ar[j][2 * c], xi = ar[j][2 * c + 1];
                        s1r += xr * p[j][2 * c] - xi * p[j][2 * c + 1];
                        s1i += xr * p[j][2 * c + 1] + xi * p[j][2 * c];
                        if (TWO) {
                            s2r += xr * q[j][2 * c] - xi * q[j][2 * c + 1];
                            s2i += xr * q[j][2 * c + 1] + xi * q[j][2 * c];
                        }
                    }
                s1r = wave_sum_lane63(s1r);
                s1i = wave_sum_lane63(s1i);
                if (TWO) {
                    s2r = wave_sum_lane63(s2r);
                    s2i = wave_sum_lane63(s2i);
                }
                if (lane == WAVE - 1) {
                    red[par][wib][0] = s1r;
                    red[par][wib][1] = s1i;
                    if (TWO) {
                        red[par][wib][2] = s2r;
                        red[par][wib][3] = s2i;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                           // raw barrier: must not drain the DMA queue
                const T t0 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
                const T t1 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
                T t2 = T(0), t3 = T(0);
                if (TWO) {
                    t2 = (red[par][0][2] + red[par][1][2]) + (red[par][2][2] + red[par][3][2]);
                    t3 = (red[par][0][3] + red[par][1][3]) + (red[par][2][3] + red[par][3][3]);
                }
                par ^= 1;
                const T rpr = t0 - br, rpi = t1 - bi;                   // residual at p
                const T rzr = t2 - br, rzi = t3 - bi;                   // residual at z_full (TWO)
                const bool last_of_batch = (inb + 1 == batch) || (base + s + 1 == nsteps);
                V *sp = HAS_TABLE ? reinterpret_cast<V *>(tbase + row * d) : nullptr;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    V tv = V(T(0));
#pragma unroll
                    for (int c = 0; c < PC; ++c) {
                        const T xr = ar[j][2 * c], xi = ar[j][2 * c + 1];
                        T pr = p[j][2 * c], pi = p[j][2 * c + 1];           // (vector elements cannot be bound by reference:
                        T avr = av[j][2 * c], avi = av[j][2 * c + 1];       //  scalar copies, written back at the end of the entry)
                        T gpr, gpi, gzr, gzi;
                        cgrad_elem(xr, xi, rpr, rpi, lam, gpr, gpi);
                        cgrad_elem(xr, xi, rzr, rzi, lam, gzr, gzi);
                        if (ALG == CA_SVRG) {                                            // SVRG_basic.jl:74-81
                            T tr = gzr - gpr, ti = gzi - gpi;
                            tr -= avr;
                            ti -= avi;
                            tr *= gamma;
                            ti *= gamma;
                            tr += pr;
                            ti += pi;
                            proxc(gamma, tr, ti, pr, pi);
                            zs[j][2 * c] += pr;
                            zs[j][2 * c + 1] += pi;
                        } else if (ALG == CA_SAGA) {                                     // SAGA_basic.jl:56-65
                            const T s_r = sr[j][2 * c], s_i = sr[j][2 * c + 1];
                            const T delr = (gpr - s_r) * invN, deli = (gpi - s_i) * invN;
                            T wr, wi;
                            if (sag) {
                                avr += delr;
                                avi += deli;
                                wr = pr - gamma * avr;
                                wi = pi - gamma * avi;
                            } else {
                                wr = pr - gamma * (gpr - s_r + avr);
                                wi = pi - gamma * (gpi - s_i + avi);
                                avr += delr;
                                avi += deli;
                            }
                            proxc(gamma, wr, wi, pr, pi);
                            tv[2 * c] = gpr;
                            tv[2 * c + 1] = gpi;
                        } else if (ALG == CA_FINITO) {                                   // Finito_basic.jl:110-118
                            const T s_r = sr[j][2 * c], s_i = sr[j][2 * c + 1];
                            const T tr = pr - (gi * invN) * gpr, ti = pi - (gi * invN) * gpi;
                            avr += (tr - s_r) * (hat_gamma / gi);
                            avi += (ti - s_i) * (hat_gamma / gi);
                            tv[2 * c] = tr;
                            tv[2 * c + 1] = ti;
                            if (last_of_batch) proxc(hat_gamma, avr, avi, pr, pi);
                        } else {                                                         // Finito_LFinito.jl:93-98
                            const T cc = hat_gamma * invN;
                            avr += cc * gzr;
                            avi += cc * gzi;
                            avr -= cc * gpr;
                            avi -= cc * gpi;
                            avr += (hat_gamma / gi) * (pr - q[j][2 * c]);
                            avi += (hat_gamma / gi) * (pi - q[j][2 * c + 1]);
                        }
                        p[j][2 * c] = pr;
                        p[j][2 * c + 1] = pi;
                        av[j][2 * c] = avr;
                        av[j][2 * c + 1] = avi;
                    }
                    if (HAS_TABLE && ok[j]) sp[cl[j]] = tv;
                    if (MASKED && !ok[j]) av[j] = p[j] = zs[j] = V(T(0));   // (dead chunks: keep the state exactly zero)
                }
                if (++inb == batch) inb = 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this lane's LDS reads of slot u are done before the DMA overwrites it
                refill(std::true_type{}, u, row_n);
            }
        }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (!ok[j]) continue;
        const int64_t c = cl[j];
        reinterpret_cast<V *>(pmem)[c] = p[j];
        reinterpret_cast<V *>(a.av)[c] = av[j];
        if (ALG == CA_SVRG) reinterpret_cast<V *>(a.z)[c] = zs[j];
    }
}

template <typename T, int J, int ALG>
constexpr size_t chain_cdma_lds_bytes()
{
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO);
    constexpr int DEPTH = DmaDepth<J, HAS_TABLE>::value;
    return (size_t)DEPTH * J * CHAIN_NT * 16 * (HAS_TABLE ? 2 : 1) + (CDMA_CHUNK + 2 * DEPTH) * sizeof(int64_t) + 2 * CDMA_CHUNK * sizeof(T) +
           (PER_SAMPLE_GAM ? CDMA_CHUNK * sizeof(T) : 0) + (HAS_TABLE ? CDMA_CHUNK * sizeof(int) : 0) + 16 + 2 * CHAIN_NW * 4 * sizeof(T);
}

template <typename T, int J, int ALG, int NT, bool SHARDED = false>
constexpr size_t chain_dma_lds_bytes()
{
    constexpr int NW = NT / WAVE;
    constexpr bool HAS_TABLE = (ALG == CA_SAGA || ALG == CA_FINITO);
    constexpr int DEPTH = DmaDepth<J * NT / 256, HAS_TABLE, SHARDED>::value;
    constexpr bool PER_SAMPLE_GAM = (ALG == CA_FINITO || ALG == CA_LFINITO || ALG == CA_SVRGC);
    constexpr bool STAGE_PTR = chain_dma_stage_ptr<T, J, ALG, NT, SHARDED>();
    return (size_t)DEPTH * J * NT * 16 * (HAS_TABLE ? 2 : 1) + (STAGE_PTR ? 2 : 1) * (CHAIN_CHUNK + 2 * DEPTH) * sizeof(int64_t) +
           CHAIN_CHUNK * sizeof(T) * (PER_SAMPLE_GAM ? 2 : 1) + (HAS_TABLE ? CHAIN_CHUNK * sizeof(int) : 0) + 16 +
           2 * NW * 2 * sizeof(T) + (SHARDED ? SHARD_QW * sizeof(int64_t) : 0);
}

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
