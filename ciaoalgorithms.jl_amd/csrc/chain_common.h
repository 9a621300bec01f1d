// chain_common.h -- the strictly sequential inner loops (SURVEY.md section 8a rows S3, G3, and F3/F4 with small
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
//
// This file: what every chain kernel shares -- the algorithms' numbers, the argument block, the argument-block readers, the small
// scalar helpers.  The kernels: chain_reg_kernels.h (register ring, any-length, complex), chain_dma_kernels.h (LDS-DMA ring, real and
// complex), afinito_kernels.h (adaptive Finito), vector_kernels.h (the one-launch element-wise kernels); chain_kernels.h includes all.
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

}  // namespace ciao
