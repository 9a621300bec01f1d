// proshi_kernels.h -- ProShI (src/algorithms/ProShI/ProShI_basic.jl) for the operator
// family of the reference's own test:  f_i = Sum(Quadratic(diagm(Q_i), q_i), SqrDistL2(IndBox(lo, hi), eta)).
// Every agent's update is element-wise in its own row (no dot product), agents of a batch are independent given z, and
// the only coupling is av = sum_i s_i: one wave per agent row, the same per-wave LDS accumulators, per-block partials and
// fixed-order finalize as the rows kernels.  HBM-bound: 3 rows read (Q_i, q_i, s_i) + 1 written per agent = 4*d*s bytes.
#pragma once

#include <type_traits>

#include "chain_kernels.h"
#include "ciao_common.h"

namespace ciao {

template <typename T>
struct ProshiArgs {
    const T *Q, *q;        // N x ld each
    int64_t ld, d, N;
    T eta, lo, hi;
    const T *gam;          // per-agent stepsizes
    T invN;
    const T *x;            // INIT: x0 ; STEP: z
    T *table;              // N x d
    int64_t nrows;
    const int64_t *idx;    // STEP: batch members, or nullptr = the contiguous rows row0 .. row0+nrows; INIT: nullptr (all rows)
    int64_t row0;
    T *partial;
    int64_t pstride;
    T *pextra;
    int *errflag;
    int dense;             // Q holds N dense d x d blocks (row k of agent i at Q + (i*d + k)*ld) instead of N diagonals
};

// grad f_i(x)_k = Q_k x_k + q_k + eta (x_k - clamp(x_k, lo, hi))
template <typename T>
__device__ __forceinline__ T sq_grad(T Qk, T qk, T eta, T lo, T hi, T x)
{
    const T pr = x < lo ? lo : (x > hi ? hi : x);
    return (Qk * x + qk) + eta * (x - pr);
}

template <typename T, bool INIT, int PROSHI_NW>
__global__ void __launch_bounds__(PROSHI_NW *WAVE) proshi_rows_kernel(ProshiArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw);          // x0 or z
    T *accs = xs + a.d;                               // [PROSHI_NW][d]
    __shared__ T red_extra[PROSHI_NW];
    const int64_t d = a.d;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * PROSHI_NW;
    T *acc = accs + (int64_t)wib * d;
    for (int64_t e = threadIdx.x; e < d; e += PROSHI_NW * WAVE) xs[e] = a.x[e];
    for (int64_t e = threadIdx.x; e < (int64_t)PROSHI_NW * d; e += PROSHI_NW * WAVE) accs[e] = T(0);
    __syncthreads();

    T extra = T(0);
    for (int64_t u = (int64_t)blockIdx.x * PROSHI_NW + wib; u < a.nrows; u += nwaves) {
        int64_t row = a.idx ? a.idx[u] : a.row0 + u;
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (lane == 0) *a.errflag = 1;
            row = 0;
        }
        const T *Qp = a.Q + row * a.ld, *qp = a.q + row * a.ld;
        T *sp = a.table + row * d;
        const T gi = a.gam[row];
        const T c = gi * a.invN;
        for (int64_t e = lane; e < d; e += WAVE) {
            if (INIT) {                                                     // ProShI_basic.jl:77-79
                const T x0 = xs[e];
                const T t = x0 - c * sq_grad(Qp[e], qp[e], a.eta, a.lo, a.hi, x0);
                sp[e] = t;
                acc[e] += t;
            } else {                                                        // :111-117
                const T s = sp[e];
                const T s2 = s + gi * xs[e];
                const T t = s2 - c * sq_grad(Qp[e], qp[e], a.eta, a.lo, a.hi, s2);
                sp[e] = t;
                acc[e] += t - s;
            }
        }
        if (INIT) extra += gi;                                              // :82  hat_γ = sum(γ)
    }
    if (lane == 0) red_extra[wib] = extra;
    __syncthreads();
    T *pout = a.partial + (int64_t)blockIdx.x * a.pstride;
    for (int64_t e = threadIdx.x; e < d; e += PROSHI_NW * WAVE) {
        T s = accs[e];
        for (int w = 1; w < PROSHI_NW; ++w) s += accs[(int64_t)w * d + e];
        pout[e] = s;
    }
    if (threadIdx.x == 0) {
        T ex = T(0);
        for (int w = 0; w < PROSHI_NW; ++w) ex += red_extra[w];
        a.pextra[blockIdx.x] = ex;
    }
}

// Vectorised variant: one workgroup per agent row, thread t owns the 16-byte chunks t + 256*j (j < J) of every d-vector,
// accumulators in registers, non-temporal 16-byte loads / stores (every byte is touched once per visit), no LDS and no
// barrier at all -- the update is element-wise.  Needs 16-byte aligned rows and d*sizeof(T) <= J*4096, J in {1,2,4,8};
// the tail of the last chunk group is masked.
template <typename T>
struct PVec;
template <>
struct PVec<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
};
template <>
struct PVec<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int N = 2;
};

template <typename T, bool INIT, int J>
__global__ void __launch_bounds__(256) proshi_vec_kernel(ProshiArgs<T> a)
{
    using V = typename PVec<T>::type;
    constexpr int VEC = PVec<T>::N;
    const int tid = threadIdx.x;
    const int64_t nchunks = a.d / VEC;
    bool ok[J];
    V xs[J], acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        ok[j] = tid + j * 256 < nchunks;
        xs[j] = ok[j] ? reinterpret_cast<const V *>(a.x)[tid + j * 256] : V(T(0));
        acc[j] = V(T(0));
    }
    T extra = T(0);
    // A workgroup's agents one after the other, the NEXT agent's three rows requested before this one's are used (two register sets,
    // the loop unrolled by two), and the index after that read one agent further ahead still (a uniform address: a scalar load, which
    // the wait for a row's vector loads does not wait for).  Without the look-ahead a workgroup has 24 KiB in flight and a batch of
    // 16 agents per workgroup is 16 dependent round trips (r = 4096, d = 1024 fp64: 38 us against 24 us of traffic).
    struct Agent {
        V Qv[J], qv[J], sv[J];
        V *sp;
        T gi;
    };
    auto resolve = [&](int64_t u) -> int64_t {
        int64_t row = a.idx ? a.idx[u] : a.row0 + u;
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        return row;
    };
    auto request = [&](Agent &g, int64_t row) {
        const V *Qp = reinterpret_cast<const V *>(a.Q + row * a.ld);
        const V *qp = reinterpret_cast<const V *>(a.q + row * a.ld);
        g.sp = reinterpret_cast<V *>(a.table + row * a.d);
        g.gi = a.gam[row];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (!ok[j]) continue;
            g.Qv[j] = __builtin_nontemporal_load(&Qp[tid + j * 256]);
            g.qv[j] = __builtin_nontemporal_load(&qp[tid + j * 256]);
            if (!INIT) g.sv[j] = __builtin_nontemporal_load(&g.sp[tid + j * 256]);
        }
    };
    auto update = [&](Agent &g) {
        const T gi = g.gi;
        const T c = gi * a.invN;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (!ok[j]) continue;
            V tv;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (INIT) {                                                 // ProShI_basic.jl:77-79
                    const T x0 = xs[j][v];
                    tv[v] = x0 - c * sq_grad(g.Qv[j][v], g.qv[j][v], a.eta, a.lo, a.hi, x0);
                    acc[j][v] += tv[v];
                } else {                                                    // :111-117
                    const T s = g.sv[j][v];
                    const T s2 = s + gi * xs[j][v];
                    tv[v] = s2 - c * sq_grad(g.Qv[j][v], g.qv[j][v], a.eta, a.lo, a.hi, s2);
                    acc[j][v] += tv[v] - s;
                }
            }
            __builtin_nontemporal_store(tv, &g.sp[tid + j * 256]);
        }
        if (INIT) extra += gi;                                              // :82  hat_γ = sum(γ)
    };
    const int64_t G = gridDim.x;
    if constexpr (J <= 4) {
        Agent g0, g1;
        int64_t u = blockIdx.x;
        int64_t row_n = u + G < a.nrows ? resolve(u + G) : 0;
        if (u < a.nrows) request(g0, resolve(u));
        while (u < a.nrows) {
            int64_t row_nn = u + 2 * G < a.nrows ? resolve(u + 2 * G) : 0;
            if (u + G < a.nrows) request(g1, row_n);
            update(g0);
            u += G;
            if (u >= a.nrows) break;
            row_n = u + 2 * G < a.nrows ? resolve(u + 2 * G) : 0;
            if (u + G < a.nrows) request(g0, row_nn);
            update(g1);
            u += G;
        }
    } else {   // 16-32 KiB rows: one register set (two would not fit)
        Agent g0;
        for (int64_t u = blockIdx.x; u < a.nrows; u += G) {
            request(g0, resolve(u));
            update(g0);
        }
    }
    V *pout = reinterpret_cast<V *>(a.partial + (int64_t)blockIdx.x * a.pstride);
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) pout[tid + j * 256] = acc[j];
    if (tid == 0) a.pextra[blockIdx.x] = extra;
}

// Dense Quadratic(Q_i, q_i): grad f_i(x) = Q_i x + q_i + eta (x - clamp(x)) needs one d x d matrix-vector product per
// visited agent, and the d*d*s bytes of Q_i are the traffic (the three d-vectors are noise next to them).  One workgroup per
// agent: the point the gradient is taken at (x0, or s_i + gam_i z) is staged in LDS, every wave streams whole rows of Q_i
// (consecutive rows of one agent are consecutive in memory, so the workgroup reads one contiguous d*ld block) and reduces
// each dot product across its lanes in a fixed order; lane 0 finishes the element.  Row k is always handled by wave k mod 4,
// so the workgroup's accumulator row in the workspace is updated without atomics.  Any d with d*sizeof(T) <= 64 KiB.
template <typename T, bool INIT>
__global__ void __launch_bounds__(256) proshi_dense_kernel(ProshiArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sv = reinterpret_cast<T *>(smem_raw);
    constexpr int NW = 4;
    const int64_t d = a.d;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    T *pout = a.partial + (int64_t)blockIdx.x * a.pstride;
    for (int64_t e = threadIdx.x; e < d; e += 256) pout[e] = T(0);
    T extra = T(0);
    for (int64_t u = blockIdx.x; u < a.nrows; u += gridDim.x) {
        int64_t row = a.idx ? a.idx[u] : a.row0 + u;
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (threadIdx.x == 0) *a.errflag = 1;
            row = 0;
        }
        const T gi = a.gam[row];
        const T c = gi * a.invN;
        T *sp = a.table + row * d;
        const T *qp = a.q + row * a.ld;
        const T *Qi = a.Q + row * d * a.ld;
        __syncthreads();                                                    // the previous agent's dot products are done with sv
        for (int64_t e = threadIdx.x; e < d; e += 256) sv[e] = INIT ? a.x[e] : sp[e] + gi * a.x[e];   // :77 / :112
        __syncthreads();
        for (int64_t k = wib; k < d; k += NW) {
            const T *qr = Qi + k * a.ld;
            T dot0 = T(0), dot1 = T(0), dot2 = T(0), dot3 = T(0);
            int64_t e = lane;
            for (; e + 3 * WAVE < d; e += 4 * WAVE) {
                const T q0 = qr[e], q1 = qr[e + WAVE], q2 = qr[e + 2 * WAVE], q3 = qr[e + 3 * WAVE];
                dot0 += q0 * sv[e];
                dot1 += q1 * sv[e + WAVE];
                dot2 += q2 * sv[e + 2 * WAVE];
                dot3 += q3 * sv[e + 3 * WAVE];
            }
            for (; e < d; e += WAVE) dot0 += qr[e] * sv[e];
            const T dot = wave_allsum((dot0 + dot1) + (dot2 + dot3));
            if (lane == 0) {
                const T x = sv[k];
                const T pr = x < a.lo ? a.lo : (x > a.hi ? a.hi : x);
                const T g = (dot + qp[k]) + a.eta * (x - pr);              // Quadratic: Q x + q ; SqrDistL2: eta (x - proj)
                const T t = x - c * g;                                      // :79 / :114-115
                if (INIT) {
                    pout[k] += t;
                } else {
                    pout[k] += t - sp[k];                                   // :111, :116
                }
                sp[k] = t;                                                  // :117
            }
        }
        if (INIT) extra += gi;                                              // :82  hat_γ = sum(γ)
    }
    if (threadIdx.x == 0) a.pextra[blockIdx.x] = extra;
}

// ------------------------------------------------------------------------------------------------------------------
// Small batches (the reference's default is ONE agent per iteration, ProShI.jl:27): with the separable operator family every
// step of ProShI_basic.jl:109-121 is element-wise -- agent update, av update AND the z update touch coordinate k only --
// so nothing couples the coordinates and a whole run of iterations is one launch in which thread k carries (av_k, z_k) in
// registers through every visited agent, in the reference's own operation order (:111-117 per agent, :119-121 per batch):
// no reduction, no barrier in the loop, no second kernel.  Against ~8 us per iteration for the launch pair of the
// batch-parallel path.  The visited agents' (Q, q, s, gamma) are prefetched PDEPTH visits ahead through register rings
// (indices and hazard flags staged in LDS per chunk of visits, exactly as the chain kernels do); an agent revisited inside
// the look-ahead window has its table entry re-read at use (this thread stored it: program order).
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
struct ProshiChainArgs {
    const T *Q, *q;
    int64_t ld, d, N;
    T eta, lo, hi;
    const T *gam;
    T invN, hat_gamma;
    const int64_t *idx;        // visited agents, batch after batch
    int64_t nvisits, batch;    // every batch has `batch` members
    ProxD<T> g;
    T *table, *av, *z;
    int *errflag;
};

constexpr int PROSHI_CH = 1008;   // visits staged at a time (a multiple of both look-ahead depths)
// look-ahead in visits: as deep as the 6-bit vmcnt allows -- (PD - 1) * (3 DW loads + 1 store) <= 63
template <typename T>
struct ProshiDepth {
    static constexpr int value = sizeof(T) == 8 ? 9 : 16;
};

template <typename T>
constexpr size_t proshi_chain_lds_bytes()
{
    constexpr int PD = ProshiDepth<T>::value;
    return (size_t)PD * 3 * (sizeof(T) / 4) * 4 * 256 + (PROSHI_CH + 2 * PD) * sizeof(int64_t) + PROSHI_CH * sizeof(int) +
           PROSHI_CH * sizeof(T) + (size_t)PD * 256 * sizeof(T) + 16;
}

// The prefetched (Q, q, s) of a visit travel by LDS-DMA (global_load_lds_dword: one dword per lane, two per fp64 value) into a
// ring of PD visits and are retired with hand-counted waits, as in chain_dma_kernel: with register rings hipcc drains the whole
// load queue at the loop's back-edge once per ring revolution (measured at d = 1024: 0.44-0.49 us per visit; here 0.31-0.35 fp64, 0.24-0.26 fp32).  gamma_i is staged
// in LDS with the indices.  No compiler-visible load is left inside the loop.
template <typename T>
__global__ void __launch_bounds__(256) proshi_chain_kernel(ProshiChainArgs<T> a_by_value)
{
    (void)a_by_value;
    CIAO_KERNARG0(ProshiChainArgs<T>, ka);
    // what a visit reads, loaded once and held in scalar registers (read in place hipcc re-loads fields inside the visit, a scalar-cache
    // round trip on the dependent path: 0.338 -> 0.403 us per visit at d = 1024 fp64, profiles/r05_kernarg_ab.txt); staging and
    // prologue fields are read where they are used
    const struct {
        const T *Q, *q;
        T *table;
        int64_t ld, d, batch, nvisits;
        T lo, hi, eta, invN, hat_gamma;
    } a = {sgpr_pin_global(ka.Q), sgpr_pin_global(ka.q), sgpr_pin_global(ka.table), sgpr_pin(ka.ld), sgpr_pin(ka.d), sgpr_pin(ka.batch),
           sgpr_pin(ka.nvisits), sgpr_pin(ka.lo), sgpr_pin(ka.hi), sgpr_pin(ka.eta), sgpr_pin(ka.invN), sgpr_pin(ka.hat_gamma)};
    constexpr int CH = PROSHI_CH, PD = ProshiDepth<T>::value, NW = 4;
    constexpr int DW = sizeof(T) / 4;                 // dwords per value
    constexpr int OPS = 3 * DW + 1;                   // per visit and wave: 3 DW LDS-DMA loads + the table store (every wave that visits has a live lane)
    constexpr int WAIT_N = (PD - 1) * OPS;
    static_assert(CH % PD == 0 && WAIT_N <= 63, "ring slots line up with chunk starts; vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    uint32_t *ring = reinterpret_cast<uint32_t *>(psm);                     // [PD][3][DW][NW][64]
    unsigned char *cur = psm + (size_t)PD * 3 * DW * NW * 256;
    int64_t *s_row = reinterpret_cast<int64_t *>(cur);
    cur += (CH + 2 * PD) * sizeof(int64_t);
    T *s_gam = reinterpret_cast<T *>(cur);
    cur += CH * sizeof(T);
    int *s_back = reinterpret_cast<int *>(cur);   // 0, or how many visits ago (1..PD) this agent was last updated: its prefetched entry is stale
    cur += CH * sizeof(int);
    cur += (16 - (reinterpret_cast<uintptr_t>(cur) & 15)) & 15;
    // hist[u][tid] = the table entry this thread's visit of slot u wrote: an agent revisited inside the window takes its value from
    // here.  In LDS, not in registers: indexed by (u - back) the compiler turns a register ring into scratch memory, and the
    // flat load + vmcnt(0) behind it drained the DMA queue on every visit (0.4-0.6 us per visit)
    T *hist = reinterpret_cast<T *>(cur);
    const uint32_t ring_off = (uint32_t)(uintptr_t)ring;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t k0 = (int64_t)blockIdx.x * 256 + tid;
    const bool live = k0 < a.d;
    // Dead threads of the last block (k0 >= d).  In a wave that also has live lanes they shadow coordinate d-1 -- loads only, their
    // stores are predicated off, so the wave issues the same instructions as any other and the counted waits stay exact.  A wave
    // with no live lane at all takes no part in the visits (it has nothing to compute, and a whole wave shadowing d-1 would race
    // with the wave that owns it: the waves of this kernel are not synchronised inside a chunk); it only helps with the staging.
    const int64_t k = live ? k0 : a.d - 1;
    const bool wave_dead = ((int64_t)blockIdx.x * 256 + (int64_t)wib * WAVE) >= a.d;   // wave-uniform
    T av = ka.av[k], z = ka.z[k];
    // g's parameters for coordinate k, fetched once
    const T gl = (ka.g.kind == CIAO_PROX_L1) ? a.hat_gamma * ka.g.lam : T(0);
    T plo = -INFINITY, phi = INFINITY;
    if (ka.g.kind == CIAO_PROX_BOX) {
        plo = ka.g.lo_vec ? ka.g.lo_vec[k] : ka.g.lo;
        phi = ka.g.hi_vec ? ka.g.hi_vec[k] : ka.g.hi;
    }
#pragma unroll
    for (int u = 0; u < PD; ++u) hist[u * 256 + tid] = T(0);
    auto refill = [&](int u, int64_t r) {
        const unsigned char *src[3] = {reinterpret_cast<const unsigned char *>(a.Q + r * a.ld + k),
                                       reinterpret_cast<const unsigned char *>(a.q + r * a.ld + k),
                                       reinterpret_cast<const unsigned char *>(a.table + r * a.d + k)};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int w = 0; w < DW; ++w) glds4(src[c] + 4 * w, ring_off + (uint32_t)((((u * 3 + c) * DW + w) * NW + wib) * 256));
    };
    auto slot = [&](int u, int c) -> T {   // plain LDS reads of what this lane's own DMA brought in
        uint32_t wv[DW];
#pragma unroll
        for (int w = 0; w < DW; ++w) wv[w] = ring[(((u * 3 + c) * DW + w) * NW + wib) * 64 + lane];
        T out;
        if constexpr (DW == 2)
            out = __longlong_as_double((long long)(((uint64_t)wv[1] << 32) | wv[0]));
        else
            out = __int_as_float((int)wv[0]);
        return out;
    };
    int64_t inb = 0;
    for (int64_t base = 0; base < a.nvisits; base += CH) {
        const int nch = (int)((a.nvisits - base) < CH ? (a.nvisits - base) : CH);
        __syncthreads();
        int64_t hrow = -1;
        if (tid < PD && base > 0) hrow = s_row[CH + tid];
        __syncthreads();
        if (tid < PD) s_row[tid] = hrow;
        for (int e = tid; e < nch + PD; e += 256) {
            int64_t v = base + e;
            if (v > a.nvisits - 1) v = a.nvisits - 1;          // look-ahead past the end repeats the last agent (harmless loads)
            int64_t r = ka.idx[v];
            if ((uint64_t)r >= (uint64_t)ka.N) {
                *ka.errflag = 1;
                r = 0;
            }
            s_row[PD + e] = r;
            if (e < nch) s_gam[e] = ka.gam[r];
        }
        __syncthreads();
        for (int e = tid; e < nch; e += 256) {
            const int64_t r = s_row[PD + e];
            int back = 0;
#pragma unroll
            for (int j = PD; j >= 1; --j)
                if (s_row[PD + e - j] == r) back = j;          // the most recent occurrence wins (smallest j)
            s_back[e] = back;
        }
        __syncthreads();
        if (wave_dead) continue;   // (after the chunk's last barrier; the next chunk starts with one)
        if (base == 0) {
#pragma unroll
            for (int u = 0; u < PD; ++u) refill(u, uniform64(s_row[PD + u]));
        }
        wait_vmcnt<0>();          // ring fully landed: the counted waits below assume the steady-state op sequence
        drain_vmcnt_visible();    // ... and hipcc knows that the staging loads are retired too
        // one ring revolution; CHK = false when every visit of the group exists (all groups of a chunk but possibly the last)
        auto group = [&](auto chk_tag, const int s0) {
            constexpr bool CHK = decltype(chk_tag)::value;
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int s = s0 + u;
                if (CHK && s >= nch) return;
                wait_vmcnt<WAIT_N>();                                            // slot u's DMA (issued PD visits ago) has landed
                const int64_t row = uniform64(s_row[PD + s]);
                const int64_t row_n = uniform64(s_row[PD + s + PD]);
                const int back = s_back[s];
                const T Qv = slot(u, 0), qv = slot(u, 1);
                T sv = slot(u, 2);
                {
                    int hs = u - back;                                           // the slot of the visit `back` visits ago
                    hs += hs < 0 ? PD : 0;
                    const T hv = hist[hs * 256 + tid];                           // (back == 0: this slot's own old entry, ignored)
                    sv = back ? hv : sv;
                }
                const T gi = s_gam[s];
                av -= sv;                                                        // :111
                const T s2 = sv + gi * z;                                        // :112
                const T pr = s2 < a.lo ? a.lo : (s2 > a.hi ? a.hi : s2);
                T gt = (Qv * s2 + qv) + a.eta * (s2 - pr);                       // :113  Quadratic: Q x + q ; SqrDistL2: eta (x - proj)
                gt *= -(gi * a.invN);                                            // :114
                gt += s2;                                                        // :115
                av += gt;                                                        // :116
                // :117 -- the dead lanes of a partly live wave do not store (the instruction is issued all the same: the counted
                // waits assume one store per visit and wave)
                if (live) a.table[row * a.d + k] = gt;
                hist[u * 256 + tid] = gt;
                if (++inb == a.batch) {
                    inb = 0;
                    z = fmin2(fmax2(av - clamp_sym(av, gl), plo), phi);          // :119  prox_{hat_gamma g}(av), branch-free
                    z -= av;                                                     // :120
                    z /= a.hat_gamma;                                            // :121
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this lane's LDS reads of slot u are done before the DMA overwrites it
                refill(u, row_n);                                                // after this visit's store (program order)
            }
        };
        for (int s0 = 0; s0 < nch; s0 += PD) {
            if (s0 + PD <= nch)
                group(std::false_type{}, s0);
            else
                group(std::true_type{}, s0);
        }
    }
    wait_vmcnt<0>();   // nothing may still be writing LDS when the workgroup retires
    if (live) {
        ka.av[k] = av;
        ka.z[k] = z;
    }
}

// solution(state): s_i += γ_i z for every agent, in place (ProShI_basic.jl:127-132)
template <typename T>
__global__ void __launch_bounds__(256) proshi_solution_kernel(int64_t N, int64_t d, const T *gam, const T *z, T *table)
{
    const int64_t total = N * d;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / d, k = t - i * d;
        table[t] += gam[i] * z[k];
    }
}

}  // namespace ciao
