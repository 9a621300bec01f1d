// rows_kernels.h -- the batch-parallel "one wave per data row" kernels (SURVEY.md section 8a rows S2, S4, G2, F2, F3,
// F4): every row a_i of the row-major N x d matrix is read from HBM exactly once, its dot product(s) with the
// iterate(s) are reduced inside one 64-lane wave (DPP + readlane, no LDS, no barrier), the scalar link function gives
// the rank-1 coefficient, and the row -- still in registers -- is accumulated into per-wave register accumulators.
// The waves of a block are then summed through LDS in a fixed order, each block writes one partial d-vector, and
// finalize_kernel sums the partials in a fixed order (bitwise run-to-run determinism; no float atomics) with the
// algorithm's epilogue (scale / axpy / prox) fused in.
//
// Roofline: HBM-bound.  Algorithmic bytes per row: d*s + s (row + b_i) for the gradient modes, 3*d*s + 2*s + 8 for
// the Finito batch (row, table row read+write, b_i, gamma_i, index).  Arithmetic intensity 0.5-1 flop/B: far below
// the vector-FMA ridge, and there is no reuse of A, so MFMA does not apply (one right-hand side = GEMV shape).
#pragma once

#include "ciao_common.h"
#include "peer_kernels.h"

namespace ciao {

enum RowMode {
    RM_GRAD = 0,         // acc += coef(a'x1) * a                                   (+ extra = sum f_i(x1) if want_fval)
    RM_GRAD2 = 1,        // acc += (coef(a'x1) - coef(a'x2)) * a ; extra += hg/gam_i  (LFinito batch, Finito_LFinito.jl:93-98)
    RM_SAGA_INIT = 2,    // table_i = grad f_i(x1) ; acc += table_i                   (SAGA_basic.jl:42-47)
    RM_FINITO_INIT = 3,  // table_i = x1 - (gam_i/N) grad f_i(x1) ; acc += table_i/gam_i   (Finito_basic.jl:77-83)
    RM_FINITO_BATCH = 4, // t = x1 - (gam_i/N) grad f_i(x1) ; acc += (t - table_i)*(hg/gam_i) ; table_i = t  (:110-117)
    RM_AFINITO_INIT = 5  // adaptive Finito init (Finito_adaptive.jl:65-93): table_i = x1; Lipschitz probe at x1 .+ 1 ->
                         // gam_i; meta_i = {c_i, f_i(x1), gam_i, a_i'x1}; acc += x1/gam_i - (c_i/N) a_i; extra += 1/gam_i
};

template <typename T>
struct RowsArgs {
    const T *A;
    const T *b;
    int64_t ld, d;
    int loss;
    T lam;
    int64_t row0, nrows;   // rows row0 .. row0+nrows (idx == nullptr) or idx[0..nrows)
    const int64_t *idx;
    const T *x1, *x2;
    T *table;              // N x d, stride d
    const T *gam;          // per-sample stepsizes (or nullptr -> gam_uniform)
    T gam_uniform;
    T invN;                // 1 / N_total
    T hat_gamma;
    int want_fval;
    T *meta;               // AFINITO_INIT: N x 4 copies x 4 per-sample scalars {c_i, f_i, gam_i, a_i's_i}
    T alpha;               // AFINITO_INIT: the solver's α
    double Nd;             // AFINITO_INIT: N_total as a double (the reference divides a Float64 by the Int N, :88)
    T *rowdot_out;         // GRAD only: if non-null, rowdot_out[row] = a_row'x1 (feeds the SVRG chain, chain_common.h CA_SVRGC)
    T *partial;            // [gridDim.x][pstride]
    int64_t pstride;
    T *pextra;             // [gridDim.x]
    int64_t N;             // local rows (index validation)
    int *errflag;          // device word set to 1 on an out-of-range index
    int small_nb;          // rows_smallm_kernel: tile buffers per wave
};

// What a rows kernel's LOOP reads of its argument block, loaded once and held in scalar registers (sgpr_pin, ciao_common.h): read in
// place through CIAO_KERNARG0, hipcc re-loads the fields inside the loop, each behind a full scalar wait in front of a row's
// requests (profiles/r05_kernarg_ab.txt).  The iterate(s), the partials' addresses and the one-off fields are read where they are used
// and take no register in the loop; a field a kernel does not read costs one scalar load.
template <typename T>
struct RowsLoopArgs {
    const T *A, *b, *gam;
    T *table, *rowdot_out;
    const int64_t *idx;
    int *errflag;
    int64_t ld, d, N, row0, nrows;
    int loss, want_fval;
    T lam, gam_uniform, invN, hat_gamma;
};
template <typename T, typename KA>
__device__ __forceinline__ RowsLoopArgs<T> rows_loop_args(const KA &ka)
{
    return RowsLoopArgs<T>{sgpr_pin_global(ka.A), sgpr_pin_global(ka.b), sgpr_pin_global(ka.gam), sgpr_pin_global(ka.table),
                           sgpr_pin_global(ka.rowdot_out), sgpr_pin_global(ka.idx), sgpr_pin_global(ka.errflag), sgpr_pin(ka.ld), sgpr_pin(ka.d),
                           sgpr_pin(ka.N), sgpr_pin(ka.row0), sgpr_pin(ka.nrows), sgpr_pin(ka.loss), sgpr_pin(ka.want_fval), sgpr_pin(ka.lam),
                           sgpr_pin(ka.gam_uniform), sgpr_pin(ka.invN), sgpr_pin(ka.hat_gamma)};
}

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
};
template <>
struct VecOf<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int N = 2;
};

constexpr int ROWS_BLOCK = 256;
constexpr int ROWS_WAVES = ROWS_BLOCK / WAVE;

// ------------------------------------------------------------------------------------------------------------------
// Fast path: d == K * 64 * VEC exactly, rows 16-byte aligned.  Lane l owns the 16-byte chunks (k*64 + l), k < K, of
// every d-vector, so each wave-instruction moves 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------------------------
// Register budget -> waves per SIMD requested from the compiler (2nd __launch_bounds__ argument = min waves per SIMD):
// a row fragment is 4*K 32-bit registers; live at once are the accumulator, the row(s) in flight (+1 when pipelined,
// +1 for the Finito table row) and ~44 (f32) / ~76 (f64) registers of addressing / dot-product temporaries.
template <int ES, int K, int MODE, int PF>
struct RowsWaves {
    // (the adaptive init of 16 KiB fp64 rows keeps a second probe's worth of scalars: at two waves per SIMD it spilled 18 registers)
    static constexpr int budget = 4 * K * (2 + PF) + ((MODE == RM_FINITO_BATCH || MODE == RM_AFINITO_INIT) ? 2 * K + 16 : 0) + (ES == 8 ? 76 : 44) +
                                  ((MODE == RM_AFINITO_INIT && K == 16 && ES == 8) ? 24 : 0);
    static constexpr int raw = 512 / ((budget + 7) / 8 * 8);
    static constexpr int value = raw < 1 ? 1 : (raw > 8 ? 8 : raw);
};

// Store flavours (16-byte chunks).  Table rows are written once per visit and not re-read by this kernel; the per-block
// partials are read by the NEXT kernel (finalize), on other XCDs.  Flavours: 0 = non-temporal, 1 = default policy, 2 = sc1
// (write-through: the bytes leave the XCD's L2 during the kernel instead of at its end-of-kernel release, MI355X_MICROARCH.md
// "publish-large").  Measured neutral within 3 % for both uses (profiles/r02_batch_store_policy_ab.txt); table rows go out
// non-temporal, partials with the default policy.
template <int FLAVOUR, typename V>
__device__ __forceinline__ void store16(V val, V *ptr)
{
    if constexpr (sizeof(V) != 16) {   // element-wise chunks (rows with no 16-byte structure): non-temporal or default only
        if constexpr (FLAVOUR == 1)
            *ptr = val;
        else
            __builtin_nontemporal_store(val, ptr);
    } else if constexpr (FLAVOUR == 0) {
        __builtin_nontemporal_store(val, ptr);
    } else if constexpr (FLAVOUR == 1) {
        *ptr = val;
    } else {
        // hipcc does not count asm stores (nothing here waits on them; the hardware retires them before the kernel ends);
        // s_nop 1: the data registers may not be overwritten before the store has read them (cdna_hip_programming.md 5.7)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(val) : "memory");
    }
}
#define TSTORE(val, ptr) store16<0>((val), (ptr))   // table rows: non-temporal
#define PSTORE(val, ptr) store16<1>((val), (ptr))   // partials: default policy

template <typename T, int K, int MODE, int PF>
__global__ void __launch_bounds__(ROWS_BLOCK, (RowsWaves<sizeof(T), K, MODE, PF>::value)) rows_fast_kernel(RowsArgs<T> a)
{
    using V = typename VecOf<T>::type;
    constexpr int VEC = VecOf<T>::N;
    constexpr int D = K * WAVE * VEC;
    constexpr bool TWO = (MODE == RM_GRAD2);
    constexpr bool TABLE = (MODE == RM_SAGA_INIT || MODE == RM_FINITO_INIT || MODE == RM_FINITO_BATCH || MODE == RM_AFINITO_INIT);

    // LDS: the iterate(s) during the sweep, then the cross-wave reduction buffer.
    __shared__ __attribute__((aligned(16))) T lds[(TWO ? 2 : 1) * D];
    __shared__ T red_extra[ROWS_WAVES];

    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * ROWS_WAVES;
    int64_t q = (int64_t)blockIdx.x * ROWS_WAVES + wib;

    for (int e = threadIdx.x; e < D; e += ROWS_BLOCK) {
        lds[e] = a.x1[e];
        if (TWO) lds[D + e] = a.x2[e];
    }
    __syncthreads();

    V acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = V(T(0));
    T extra = T(0);

    auto row_of = [&](int64_t qq) -> int64_t {
        if (!a.idx) return a.row0 + qq;
        int64_t r = a.idx[qq];
        if ((uint64_t)r >= (uint64_t)a.N) {   // memory-safe: flag it, use row 0 (results are void once flagged)
            if (lane == 0) *a.errflag = 1;
            r = 0;
        }
        return r;
    };
    auto load_row = [&](V(&r)[K], int64_t row) {
        if (a.A) {
            const V *ap = reinterpret_cast<const V *>(a.A + row * a.ld);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // every row is read exactly once per sweep: non-temporal loads keep the stream out of L2/MALL and run
                // ~12 % faster than default-policy loads here (7.1 vs 6.3 TB/s at N=10M, d=1024, fp64)
                r[k] = __builtin_nontemporal_load(&ap[k * WAVE + lane]);
            }
        } else {   // F = fill(Zero(), N): there is no data matrix at all
#pragma unroll
            for (int k = 0; k < K; ++k) r[k] = V(T(0));
        }
    };

    // one row: dot(s) -> link function -> rank-1 accumulate (+ table update)
    auto process = [&](V(&cur)[K], int64_t row, T bi) {
        // The iterate is re-read from LDS for every row (keeps it out of the register budget); the opaque lane offset
        // stops the compiler from hoisting these loop-invariant reads back into 2*K*VEC registers.
        int xl = lane;
        asm volatile("" : "+v"(xl));
        const V *x1v = reinterpret_cast<const V *>(lds) + xl;
        const V *x2v = reinterpret_cast<const V *>(lds + D) + xl;

        T *tp = TABLE ? a.table + row * a.d : nullptr;

        T d1 = T(0), d2 = T(0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const V xv = x1v[k * WAVE];
#pragma unroll
            for (int v = 0; v < VEC; ++v) d1 += cur[k][v] * xv[v];
            if (TWO) {
                const V yv = x2v[k * WAVE];
#pragma unroll
                for (int v = 0; v < VEC; ++v) d2 += cur[k][v] * yv[v];
            }
        }
        d1 = wave_allsum(d1);
        if (TWO) d2 = wave_allsum(d2);

        const GradCoef<T> g1 = grad_coef(a.loss, d1, bi, a.lam);
        if (MODE == RM_GRAD) {
            const T c = g1.coef();
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += c * cur[k];
            if (a.want_fval) extra += loss_value(a.loss, d1, bi, a.lam);
            if (a.rowdot_out && lane == 0) a.rowdot_out[row] = d1;
        } else if (MODE == RM_GRAD2) {
            const GradCoef<T> g2 = grad_coef(a.loss, d2, bi, a.lam);
            const T c = g1.coef() - g2.coef();
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += c * cur[k];
            const T gi = a.gam ? a.gam[row] : a.gam_uniform;
            extra += a.hat_gamma / gi;
        } else if (MODE == RM_SAGA_INIT) {
            V *sp = reinterpret_cast<V *>(tp);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                V gv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) gv[v] = g1.elem(cur[k][v]);
                TSTORE(gv, &sp[k * WAVE + lane]);
                acc[k] += gv;
            }
        } else if (MODE == RM_AFINITO_INIT) {
            // both probe gradients are multiples of a_i:  ||grad f_i(x1 .+ 1) - grad f_i(x1)|| = |c(d1 + sum a) - c(d1)| ||a||
            T sa = T(0), n2 = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    sa += cur[k][v];
                    n2 += cur[k][v] * cur[k][v];
                }
            sa = wave_allsum(sa);
            n2 = wave_allsum(n2);
            const T c0 = g1.coef();
            const T c1 = grad_coef(a.loss, d1 + sa, bi, a.lam).coef();
            T nmg = fabs2(c1 - c0) * fsqrt(n2);
            // a.gam, when given, holds stepsizes resolved by the host (> 0: use it) -- the second pass after a re-probe
            const T gov = a.gam ? a.gam[row] : T(0);
            bool degenerate = false;
            if (!(gov > T(0)) && nmg < Eps<T>::value) {
                // grad f_i(x0 .+ 1) == grad f_i(x0) (a row whose entries sum to zero): the reference now probes at random points
                // x0 .+ rand(t*[-1,1]) from its RNG (:78-85).  Random draws are a host input on this path: flag the row (gamma_i =
                // -1 in meta, status 2), carry on with a finite placeholder, and let the host resolve it (ciao_afinito_probe)
                if (lane == 0) *a.errflag = 2;
                nmg = Eps<T>::value;
                degenerate = true;
            }
            // L_int = zeros(N) is a Float64 array whatever R (:73): L = nmg / (t sqrt(d)) / N and alpha / L are Float64,
            // gamma_i is that quotient rounded to R (:86-88)
            const T gi = gov > T(0) ? gov : (T)((double)a.alpha / (((double)nmg / sqrt((double)a.d)) / a.Nd));
            const T rinv = T(1) / gi;
            const T cn = c0 * a.invN;
            V *sp = reinterpret_cast<V *>(tp);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const V xv = x1v[k * WAVE];
                TSTORE(xv, &sp[k * WAVE + lane]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[k][v] += xv[v] * rinv - cn * cur[k][v];
            }
            extra += rinv;
            if (lane < 4) {   // four identical copies, one per wave of the step kernel (afinito_kernels.h)
                T *mp = a.meta + (row * 4 + lane) * 4;
                mp[0] = c0;
                mp[1] = loss_value(a.loss, d1, bi, a.lam);
                mp[2] = degenerate ? T(-1) : gi;
                mp[3] = d1;
            }
        } else {  // FINITO_INIT / FINITO_BATCH
            const T gi = a.gam ? a.gam[row] : a.gam_uniform;
            const T cg = gi * a.invN;   // gam_i / N
            const T rr = (MODE == RM_FINITO_INIT) ? T(1) / gi : a.hat_gamma / gi;
            V *sp = reinterpret_cast<V *>(tp);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const V xv = x1v[k * WAVE];
                V tv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) tv[v] = xv[v] - cg * g1.elem(cur[k][v]);
                if (MODE == RM_FINITO_INIT) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[k][v] += tv[v] * rr;   // s_i / gam_i
                } else {
                    const V sv = __builtin_nontemporal_load(&sp[k * WAVE + lane]);   // old table row, read just before it is replaced
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[k][v] += (tv[v] - sv[v]) * rr;
                }
                TSTORE(tv, &sp[k * WAVE + lane]);
                // keep the scheduler from hoisting every x / table fragment read to the top (register pressure)
                if (K >= 8 && (k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (PF) {
        // explicit two-deep software pipeline (ping-pong register buffers): the next row's loads are in flight while
        // the current row is reduced.
        V bufA[K], bufB[K];
        int64_t rowA = 0, rowB = 0;
        T bA = T(0), bB = T(0);
        if (q < a.nrows) {
            rowA = row_of(q);
            load_row(bufA, rowA);
            bA = a.b ? a.b[rowA] : T(0);
            while (true) {
                int64_t qn = q + nwaves;
                bool more = qn < a.nrows;
                if (more) {
                    rowB = row_of(qn);
                    load_row(bufB, rowB);
                    bB = a.b ? a.b[rowB] : T(0);
                }
                process(bufA, rowA, bA);
                if (!more) break;
                q = qn;
                qn = q + nwaves;
                more = qn < a.nrows;
                if (more) {
                    rowA = row_of(qn);
                    load_row(bufA, rowA);
                    bA = a.b ? a.b[rowA] : T(0);
                }
                process(bufB, rowB, bB);
                if (!more) break;
                q = qn;
            }
        }
    } else {
        // latency hidden by occupancy alone: one row in flight per wave, more waves per SIMD
        for (; q < a.nrows; q += nwaves) {
            V cur[K];
            const int64_t row = row_of(q);
            load_row(cur, row);
            const T bi = a.b ? a.b[row] : T(0);
            process(cur, row, bi);
        }
    }

    // cross-wave reduction through LDS in wave order (deterministic), then one partial per block
    __syncthreads();  // everyone is done reading the iterates
    V *red = reinterpret_cast<V *>(lds);
    for (int w = 0; w < ROWS_WAVES; ++w) {
        if (wib == w) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (w == 0)
                    red[k * WAVE + lane] = acc[k];
                else
                    red[k * WAVE + lane] += acc[k];
            }
        }
        __syncthreads();
    }
    if (lane == 0) red_extra[wib] = extra;   // extra is wave-uniform
    __syncthreads();
    T *pout = a.partial + (int64_t)blockIdx.x * a.pstride;
    for (int e = threadIdx.x; e < D; e += ROWS_BLOCK) pout[e] = lds[e];
    if (threadIdx.x == 0) {
        T ex = T(0);
        for (int w = 0; w < ROWS_WAVES; ++w) ex += red_extra[w];
        a.pextra[blockIdx.x] = ex;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ONE WORKGROUP per row (rows_split_kernel).  Two jobs:
//
// (1) Small and medium batches of the batch steps (FINITO_BATCH, and GRAD2 = the LFinito batch).  A batch of a few
//     hundred rows gives the wave-per-row kernel above one row per wave and nothing to hide its serial latencies behind
//     (row -> dot -> table row -> store: two dependent HBM round trips on 16 loads per lane).  Here the four waves of a
//     workgroup share a row -- thread t owns the 16-byte chunks t + 256*j, j < J, as in the chain kernels -- so the row
//     and its table row are requested together in J loads per lane, the dot product costs one 4-partial LDS exchange,
//     and the per-thread accumulators need no cross-wave combine: a block's partial is stored straight from registers.
//
// (2) Every row length the wave-per-row fast path does not cover (it needs d = 64*VEC*{1,2,4,8,16} exactly): with
//     MASKED the last chunk group is predicated, so any 16-byte aligned row of up to J*4096 bytes, J <= 16, runs here
//     (d = 1000, 1536, 3000 ...; also d = 4096 fp64) instead of on the scalar generic kernel.
// ------------------------------------------------------------------------------------------------------------------
// (3) VEC = 1 (one element per "chunk" instead of 16 bytes): rows with no alignment at all -- odd d, odd row strides --
//     of up to 4096 elements.  Narrower loads (256 / 512 B per wave-instruction) but still one pass over the row.
template <typename T, int N>
struct ChunkOf {
    typedef T type __attribute__((ext_vector_type(N)));
};

template <typename T, int J, int MODE, bool MASKED, int VEC>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_split_kernel(RowsArgs<T> a)
{
    using V = typename ChunkOf<T, VEC>::type;
    constexpr bool TWO = (MODE == RM_GRAD2);
    constexpr bool TABLE = (MODE == RM_SAGA_INIT || MODE == RM_FINITO_INIT || MODE == RM_FINITO_BATCH);
    constexpr bool TREAD = (MODE == RM_FINITO_BATCH);
    static_assert(MODE != RM_AFINITO_INIT, "the adaptive init keeps to the wave-per-row kernels");

    __shared__ T red[2][ROWS_WAVES][2];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nchunks = a.d / VEC;

    bool ok[J];
    V x1[J], x2[J], acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        ok[j] = !MASKED || (tid + j * ROWS_BLOCK < nchunks);
        x1[j] = ok[j] ? reinterpret_cast<const V *>(a.x1)[tid + j * ROWS_BLOCK] : V(T(0));
        x2[j] = (TWO && ok[j]) ? reinterpret_cast<const V *>(a.x2)[tid + j * ROWS_BLOCK] : V(T(0));
        acc[j] = V(T(0));
    }
    T extra = T(0);
    int par = 0;

    struct RowIn {
        V ar[J], sr[J];
        V *sp;
        T bi, gi;
        int64_t row;
    };
    // request everything row q needs at once: the row, its table row and its scalars
    auto issue = [&](RowIn &x, int64_t q) {
        int64_t row = a.idx ? a.idx[q] : a.row0 + q;
        if (a.idx && (uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        x.row = row;
        x.sp = TABLE ? reinterpret_cast<V *>(a.table + row * a.d) : nullptr;
        if (a.A) {
            const V *ap = reinterpret_cast<const V *>(a.A + row * a.ld);
#pragma unroll
            for (int j = 0; j < J; ++j) x.ar[j] = ok[j] ? __builtin_nontemporal_load(&ap[tid + j * ROWS_BLOCK]) : V(T(0));
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j) x.ar[j] = V(T(0));
        }
        if (TREAD) {
#pragma unroll
            for (int j = 0; j < J; ++j) x.sr[j] = ok[j] ? __builtin_nontemporal_load(&x.sp[tid + j * ROWS_BLOCK]) : V(T(0));
        }
        x.bi = a.b ? a.b[row] : T(0);
        x.gi = (MODE == RM_GRAD || MODE == RM_SAGA_INIT) ? T(1) : (a.gam ? a.gam[row] : a.gam_uniform);
    };
    auto process = [&](RowIn &x) {
        T d1 = T(0), d2 = T(0);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                d1 += x.ar[j][v] * x1[j][v];
                if (TWO) d2 += x.ar[j][v] * x2[j][v];
            }
        d1 = wave_sum_lane63(d1);
        if (TWO) d2 = wave_sum_lane63(d2);
        if (lane == WAVE - 1) {   // the lane that holds the wave's sum
            red[par][wib][0] = d1;
            if (TWO) red[par][wib][1] = d2;
        }
        // one raw barrier per row (LDS traffic only, no memory fence needed); the slots alternate, so a wave that is one
        // row ahead writes the other pair
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        d1 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
        if (TWO) d2 = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
        par ^= 1;

        const GradCoef<T> g1 = grad_coef(a.loss, d1, x.bi, a.lam);
        if (MODE == RM_GRAD) {                        // SVRG_basic.jl:58-63, :87-92
            const T c = g1.coef();
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += c * x.ar[j];
            if (a.want_fval) extra += loss_value(a.loss, d1, x.bi, a.lam);
            if (a.rowdot_out && tid == 0) a.rowdot_out[x.row] = d1;
        } else if (MODE == RM_GRAD2) {                // Finito_LFinito.jl:93-98
            const T c = g1.coef() - grad_coef(a.loss, d2, x.bi, a.lam).coef();
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += c * x.ar[j];
            extra += a.hat_gamma / x.gi;
        } else if (MODE == RM_SAGA_INIT) {            // SAGA_basic.jl:42-47
#pragma unroll
            for (int j = 0; j < J; ++j) {
                V gv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) gv[v] = g1.elem(x.ar[j][v]);
                acc[j] += gv;
                if (ok[j]) TSTORE(gv, &x.sp[tid + j * ROWS_BLOCK]);
            }
        } else {                                      // Finito_basic.jl:77-83 (init) / :110-117 (batch)
            const T cg = x.gi * a.invN;
            const T rr = (MODE == RM_FINITO_INIT) ? T(1) / x.gi : a.hat_gamma / x.gi;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                V tv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    tv[v] = x1[j][v] - cg * g1.elem(x.ar[j][v]);
                    if (MODE == RM_FINITO_INIT)
                        acc[j][v] += tv[v] * rr;
                    else
                        acc[j][v] += (tv[v] - x.sr[j][v]) * rr;
                }
                if (ok[j]) TSTORE(tv, &x.sp[tid + j * ROWS_BLOCK]);
            }
        }
    };

    // One row at a time per workgroup.  Measured and dropped, both bitwise neutral: a two-deep register pipeline (next row in
    // flight while this one is reduced: 29.1 vs 28.5 us at r = 2048, 5.1 vs 5.5 TB/s at r = 65536), and -- for short batches of 2-4
    // rows per workgroup, BASELINE config #5's 512-row share -- ALL of a workgroup's rows and table rows requested before the first
    // is reduced (round 4, profiles/r04_c5_pre_ab.txt: 13.71 -> 13.59 us at r = 512, 18.39 -> 18.23 at r = 1024).  Every further
    // row per workgroup costs 2.4 us either way: 256 rows x 48 KB in 2.4 us is 5.2 TB/s, the HBM's mixed read/write ceiling (one
    // CU streams ~20 GB/s of it), so the rows are bandwidth, not latency; what is left of a batch is its fixed part -- two kernel
    // boundaries and two dependent memory round trips (rows -> partials -> finalize), 8.9 us at d = 4096.
    RowIn cur;
    for (int64_t q = blockIdx.x; q < a.nrows; q += gridDim.x) {
        issue(cur, q);
        process(cur);
    }

    V *pout = reinterpret_cast<V *>(a.partial + (int64_t)blockIdx.x * a.pstride);
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) {
            if constexpr (VEC * sizeof(T) == 16)
                PSTORE(acc[j], &pout[tid + j * ROWS_BLOCK]);
            else
                pout[tid + j * ROWS_BLOCK] = acc[j];
        }
    if (tid == 0) a.pextra[blockIdx.x] = extra;   // extra is workgroup-uniform
}

// ------------------------------------------------------------------------------------------------------------------
// Rows shorter than a wave's reach (d <= 256 elements): SEVERAL ROWS PER WAVE (rows_small_kernel).
// The full sweeps over a dense matrix (no index list, ld == d) see G consecutive rows as one contiguous stretch of G*d
// elements; a wave loads it coalesced (lane l takes elements l, l+64, ...: SMALL_I of them), so element (l, i) always
// belongs to the same row-in-group and the same column, whatever the group.  Per group: products a*x go to the wave's
// private LDS area, Q lanes per row add up the row's d products and combine by shuffles, one of them evaluates the link function, and every element picks up its row's scalar again through LDS.  Accumulators
// stay in registers per (lane, i); columns are combined once, at the end.  Covers GRAD, SAGA_INIT and FINITO_INIT.
// ------------------------------------------------------------------------------------------------------------------
// SMALL_I elements per lane and wave-iteration (8 or 16): the smaller group halves the registers and doubles the waves per CU

template <typename T, int MODE, int SMALL_I, bool PADDED>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_small_kernel(RowsArgs<T> a_by_value)
{
    (void)a_by_value;
    CIAO_KERNARG0(RowsArgs<T>, ka);
    const RowsLoopArgs<T> a = rows_loop_args<T>(ka);
    constexpr int SMALL_GE = WAVE * SMALL_I;   // elements of one wave-iteration
    static_assert(MODE == RM_GRAD || MODE == RM_SAGA_INIT || MODE == RM_FINITO_INIT, "contiguous full sweeps only");
    extern __shared__ __attribute__((aligned(16))) unsigned char small_raw[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int d = (int)a.d;
    // per wave: prod[SMALL_GE] | s1[64] | gg[64];  after the sweep the first ROWS_WAVES*d elements of the block's area hold
    // the per-wave column sums
    T *base = reinterpret_cast<T *>(small_raw);
    T *prod = base + (size_t)wib * (SMALL_GE + 2 * WAVE);
    T *s1s = prod + SMALL_GE;
    T *ggs = s1s + WAVE;
    int G = SMALL_GE / d;
    if (G > WAVE) G = WAVE;
    const int used = G * d;
    const T s2 = (a.loss == CIAO_LOSS_LS) ? a.lam : (a.loss == CIAO_LOSS_LOGISTIC ? T(1) : T(0));
    // the dot product of row r is shared by Q = 2^qs adjacent lanes (r*Q .. r*Q+Q-1) when the group has few rows
    int qs = 0;
    while ((2 << qs) * G <= WAVE) ++qs;
    const int Q = 1 << qs;
    const int myrow = lane >> qs, myq = lane & (Q - 1);

    // PADDED (row stride ld > d): element (r, c) of the group sits r*ld + c elements after the group's first; with
    // ld == d that is the element's own number e, the loads are perfectly contiguous and no offset table is kept (its 16
    // registers cost the fp64 variant half its bandwidth: 4.5 -> 2.4 TB/s)
    const int ld = PADDED ? (int)a.ld : d;
    // An element beyond the group's G*d (a dead slot of the wave-iteration) carries the row number WAVE: "its row is one of this
    // group's nr <= G <= WAVE rows" is then the ONE test that also says the slot is live.  (As an array of booleans the liveness
    // was 2 * SMALL_I scalar registers of loop-invariant lane masks: with 16 slots per lane the kernel spilled 12-47 of them.)
    // s1s[WAVE] is ggs[0]: finite, and multiplied by the dead slot's zero.
    int rrow[SMALL_I], aoff[PADDED ? SMALL_I : 1];
    T xcol[SMALL_I], acc[SMALL_I];
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i) {
        const int e = lane + WAVE * i;
        const bool live = e < used;
        rrow[i] = live ? e / d : WAVE;
        const int c = live ? e - rrow[i] * d : 0;
        if (PADDED) aoff[i] = live ? rrow[i] * ld + c : 0;
        xcol[i] = live ? ka.x1[c] : T(0);
        acc[i] = T(0);
    }
    T ex = T(0);
    T gam_u = a.gam_uniform;   // as a VALUE (see fetch)
    asm volatile("" : "+v"(gam_u));
    s1s[lane] = T(0);   // rows of a short last group are never written: their (unused, times-zero) scalars must be finite
    ggs[lane] = T(1);
    const int64_t ngroups = (a.nrows + G - 1) / G;
    const int64_t nwaves = (int64_t)gridDim.x * ROWS_WAVES;
    // the elements of group g (rows g*G ...): one coalesced element-wise load per (lane, i), zero beyond the matrix
    // ... and the scalars (b_i, gamma_i) of the row a lane works on in the same breath: a load issued later, between the phases, would
    // make the compiler's in-order vmcnt wait drain the prefetch of the next group as well
    auto fetch = [&](T(&v)[SMALL_I], T &bi, T &gi, int64_t g) {
        const int64_t left = a.nrows - g * G;
        const int nrg = left < G ? (int)left : G;
        const int64_t rb = a.row0 + g * G;
        const T *gp = a.A ? a.A + rb * (int64_t)ld : nullptr;
        // branch-free: dead elements read the group's first element and are zeroed by a select (a predicated load per
        // element costs an exec-mask branch each; the loop was instruction-bound at ~1500 instructions per group)
#pragma unroll
        for (int i = 0; i < SMALL_I; ++i) {
            const bool on = rrow[i] < nrg;
            const int ec = on ? (PADDED ? aoff[i] : lane + WAVE * i) : 0;
            const T val = gp ? __builtin_nontemporal_load(&gp[ec]) : T(0);
            v[i] = on ? val : T(0);
        }
        // volatile: hipcc otherwise sinks these two loads down to their use, behind the prefetch.  Through GLOBAL-address-space
        // pointers, unconditionally per lane (a lane without a row reads the group's first row's scalar and drops it): a
        // volatile load through a generic pointer is a FLAT instruction, and a per-lane `cond ? *p : value` made the compiler
        // select between p and a scratch-memory copy of the value (8-16 bytes of scratch in every variant of this kernel)
        typedef const volatile __attribute__((address_space(1))) T *gvol;
        const int64_t rsel = rb + (myrow < nrg ? myrow : 0);
        T bv = T(0), gv = gam_u;
        if (a.b) bv = *reinterpret_cast<gvol>((uintptr_t)(a.b + rsel));
        if (MODE == RM_FINITO_INIT && a.gam) gv = *reinterpret_cast<gvol>((uintptr_t)(a.gam + rsel));
        bi = myrow < nrg ? bv : T(0);
        gi = myrow < nrg ? gv : gam_u;
    };
    T av[SMALL_I], avn[SMALL_I];
    T bcur = T(0), gcur = T(1), bnext = T(0), gnext = T(1);
    int64_t g = (int64_t)blockIdx.x * ROWS_WAVES + wib;
    if (g < ngroups) fetch(av, bcur, gcur, g);
    for (; g < ngroups; g += nwaves) {
        const int64_t row_b = a.row0 + g * G;
        const int64_t left = a.nrows - g * G;
        const int nr = left < G ? (int)left : G;
#pragma unroll
        for (int i = 0; i < SMALL_I; ++i) prod[lane + WAVE * i] = av[i] * xcol[i];   // unconditional: dead slots are never read
        // the next group's elements travel while this one is reduced (one group in flight per wave is too little to cover
        // the HBM latency at two waves per SIMD)
        const bool more = g + nwaves < ngroups;
        if (more) fetch(avn, bnext, gnext, g + nwaves);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed (no other wave touches them)
        {
            // lane (r, q) adds elements q, q+Q, ... of row r (two interleaved partial sums keep two LDS reads in flight),
            // then the Q partials are combined by xor-shuffles inside the aligned group of Q lanes
            const bool rowlive = myrow < nr;
            const T *pr = prod + (rowlive ? myrow : 0) * d;
            T d0 = T(0), d1 = T(0);
            int k = myq;
            for (; k + Q < d; k += 2 * Q) {
                d0 += pr[k];
                d1 += pr[k + Q];
            }
            if (k < d) d0 += pr[k];
            T dot = d0 + d1;
            for (int m = Q >> 1; m > 0; m >>= 1) dot += __shfl_xor(dot, m, WAVE);
            if (rowlive && myq == 0) {
                const int64_t row = row_b + myrow;
                const T bi = bcur;
                const GradCoef<T> gc = grad_coef(a.loss, dot, bi, a.lam);
                s1s[myrow] = gc.s1;
                if (MODE == RM_FINITO_INIT) ggs[myrow] = gcur;
                if (MODE == RM_GRAD) {
                    if (a.want_fval) ex += loss_value(a.loss, dot, bi, a.lam);
                    if (a.rowdot_out) a.rowdot_out[row] = dot;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T *tp = (MODE == RM_GRAD) ? nullptr : a.table + row_b * (int64_t)d;
#pragma unroll
        for (int i = 0; i < SMALL_I; ++i) {
            const int e = lane + WAVE * i;
            if (MODE == RM_GRAD) {
                // unconditional: dead elements carry a = 0 and a (finite) scalar of an earlier group, i.e. add zero
                acc[i] += (s1s[rrow[i]] * s2) * av[i];               // coef() * a, as the wave-per-row kernels
                continue;
            }
            if (!(rrow[i] < nr)) continue;
            const T s1 = s1s[rrow[i]];
            if (MODE == RM_SAGA_INIT) {
                const T gv = (av[i] * s1) * s2;                       // GradCoef::elem
                __builtin_nontemporal_store(gv, &tp[e]);
                acc[i] += gv;
            } else {
                const T gi = ggs[rrow[i]];
                const T tv = xcol[i] - (gi * a.invN) * ((av[i] * s1) * s2);
                __builtin_nontemporal_store(tv, &tp[e]);
                acc[i] += tv * (T(1) / gi);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // s1s / ggs are rewritten by the next group
        if (more) {
#pragma unroll
            for (int i = 0; i < SMALL_I; ++i) av[i] = avn[i];
            bcur = bnext;
            gcur = gnext;
        }
    }

    // columns: element (lane, i) -> prod[e]; lane c then adds rows 0..G-1 of column c in row order
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i)
        if (rrow[i] < WAVE) prod[lane + WAVE * i] = acc[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T colsum[(256 + WAVE - 1) / WAVE];
#pragma unroll
    for (int q = 0; q < (256 + WAVE - 1) / WAVE; ++q) {
        const int c = lane + WAVE * q;
        T sacc = T(0);
        if (c < d)
            for (int r = 0; r < G; ++r) sacc += prod[r * d + c];
        colsum[q] = sacc;
    }
    ex = wave_allsum(ex);
    __syncthreads();   // every wave is done with its private area; reuse the block's area for the per-wave column sums
    __shared__ T red_extra_small[ROWS_WAVES];
#pragma unroll
    for (int q = 0; q < (256 + WAVE - 1) / WAVE; ++q) {
        const int c = lane + WAVE * q;
        if (c < d) base[wib * d + c] = colsum[q];
    }
    if (lane == 0) red_extra_small[wib] = ex;
    __syncthreads();
    T *pout = ka.partial + (int64_t)blockIdx.x * ka.pstride;
    for (int c = threadIdx.x; c < d; c += ROWS_BLOCK) {
        T sacc = base[c];
        for (int w = 1; w < ROWS_WAVES; ++w) sacc += base[w * d + c];
        pout[c] = sacc;
    }
    if (threadIdx.x == 0) {
        T e2 = T(0);
        for (int w = 0; w < ROWS_WAVES; ++w) e2 += red_extra_small[w];
        ka.pextra[blockIdx.x] = e2;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The BATCH modes on rows shorter than a wave's reach (rows_smallb_kernel; round 4): Finito batches (FINITO_BATCH) and LFinito's
// batch sweep (GRAD2) over an index list or a row block of rows of up to 256 elements -- tabular-size d, the shape of the
// reference's own problems scaled in N (test/test_lasso.jl:15, test/test_logistic_l1.jl:12-26), which used to fall to the
// scalar generic kernel at 0.5-1.6 TB/s.  Same geometry as rows_small_kernel: a wave takes G rows per iteration, element
// (lane, i) of the G*d elements always belongs to the same row-in-group and column, products through the wave's private LDS
// area, Q = 64/G lanes add up a row's products and combine by shuffles, the row's scalars travel back through LDS, accumulators per
// (lane, i) in registers, columns combined once at the end.  What the batch modes add:
//   * the rows of a group may lie ANYWHERE (index list; padded rows): lane r < G resolves row r of the group -- index, bounds
//     check, element offsets of the data row and the table row, b_i, gamma_i -- and parks them in LDS, TWO groups ahead of their
//     use (three rotating buffers), so that neither the index load nor these scalars are ever waited for; an element's address
//     is its row's offset (one LDS read) plus its column;
//   * GRAD2: a second dot product per row (a second product area) and extra += hat_gamma / gamma_i per row;
//   * FINITO_BATCH: the table row travels with the data row (same element mapping), t = z - (gamma_i/N) grad, acc += (t - s_i)
//     hat_gamma / gamma_i, s_i = t (Finito_basic.jl:110-117).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int MODE, int SMALL_I, bool GATHER>
constexpr size_t smallb_wave_bytes()
{
    // (GATHER: offA[3][64] i64 | offT[3][64] i64 |) prod[GE] (| prod2[GE]) | s1[64] | cg[64] | rr[64] | bis[3][64] | gis[3][64]
    return (size_t)(MODE == RM_GRAD2 ? 2 : 1) * WAVE * SMALL_I * sizeof(T) + 3 * WAVE * sizeof(T) + (GATHER ? 2 * 3 * WAVE * sizeof(int64_t) : 0) +
           2 * 3 * WAVE * sizeof(T);
}

// GATHER = false: a dense row block (no index list, ld == d): the group's rows -- and table rows -- are ONE contiguous stretch, an
// element's address is the group's base plus its own number, and only b_i / gamma_i go through the staging.
template <typename T, int MODE, int SMALL_I, bool GATHER>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_smallb_kernel(RowsArgs<T> a_by_value)
{
    (void)a_by_value;
    // (read in place: holding the fields costs 6-16 spilled registers here, and index lists run on rows_wrow_kernel by default)
    CIAO_KERNARG0(RowsArgs<T>, a);
    static_assert(MODE == RM_GRAD2 || MODE == RM_FINITO_BATCH, "the batch modes");
    constexpr bool TWO = (MODE == RM_GRAD2), TABLE = (MODE == RM_FINITO_BATCH);
    constexpr int GE = WAVE * SMALL_I;
    extern __shared__ __attribute__((aligned(16))) unsigned char smallb_raw[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int d = (int)a.d;
    unsigned char *area = smallb_raw + (size_t)wib * smallb_wave_bytes<T, MODE, SMALL_I, GATHER>();
    int64_t *offA = reinterpret_cast<int64_t *>(area);                 // [3][64]: 8-byte aligned pieces first
    int64_t *offT = offA + (GATHER ? 3 * WAVE : 0);
    T *prod = reinterpret_cast<T *>(offT + (GATHER ? 3 * WAVE : 0));
    T *prod2 = prod + (TWO ? GE : 0);
    T *s1s = prod2 + GE;
    T *cgs = s1s + WAVE;
    T *rrs = cgs + WAVE;
    T *bis = rrs + WAVE;                                               // [3][64]
    T *gis = bis + 3 * WAVE;
    int G = GE / d;
    if (G > WAVE) G = WAVE;
    const int used = G * d;
    const T s2 = (a.loss == CIAO_LOSS_LS) ? a.lam : (a.loss == CIAO_LOSS_LOGISTIC ? T(1) : T(0));
    int qs = 0;
    while ((2 << qs) * G <= WAVE) ++qs;
    const int Q = 1 << qs;
    const int myrow = lane >> qs, myq = lane & (Q - 1);

    bool live[SMALL_I];
    int rrow[SMALL_I], col[SMALL_I];
    T xcol[SMALL_I], x2col[TWO ? SMALL_I : 1], acc[SMALL_I];
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i) {
        const int e = lane + WAVE * i;
        live[i] = e < used;
        rrow[i] = live[i] ? e / d : 0;
        col[i] = live[i] ? e - rrow[i] * d : 0;
        xcol[i] = live[i] ? a.x1[col[i]] : T(0);
        if (TWO) x2col[i] = live[i] ? a.x2[col[i]] : T(0);
        acc[i] = T(0);
    }
    T ex = T(0);
    T gam_u = a.gam_uniform;   // as a VALUE (rows_small_kernel: a select against a global load would park it in scratch)
    asm volatile("" : "+v"(gam_u));
    s1s[lane] = T(0);
    cgs[lane] = T(0);
    rrs[lane] = T(0);
    const int64_t ngroups = (a.nrows + G - 1) / G;
    const int64_t nwaves = (int64_t)gridDim.x * ROWS_WAVES;
    typedef const volatile __attribute__((address_space(1))) T *gvol;
    typedef const volatile __attribute__((address_space(1))) int64_t *gvol64;

    // Lane r < G resolves row r of a group in three steps that are spread over three iterations, so that none of their loads is
    // ever waited for: the row's INDEX (three groups ahead), then b_i and gamma_i at that row (two groups ahead), then everything
    // parked in an LDS buffer (three rotate: this group's, the next's, the one after).  Rows beyond the batch's end stand in for
    // their group's first row (their elements are masked out of everything).  Volatile global loads: they must be ISSUED where
    // they stand.
    auto load_row = [&](int64_t gg) -> int64_t {
        if (lane >= G || gg >= ngroups) return 0;
        int64_t q = gg * G + lane;
        if (q >= a.nrows) q = gg * G;
        return a.idx ? *reinterpret_cast<gvol64>((uintptr_t)(a.idx + q)) : a.row0 + q;
    };
    auto check_row = [&](int64_t row) -> int64_t {
        if (a.idx && (uint64_t)row >= (uint64_t)a.N) {
            *a.errflag = 1;
            row = 0;
        }
        return row;
    };
    auto load_scal = [&](int64_t row, T &bq, T &gq) {
        bq = T(0);
        gq = gam_u;
        if (lane < G) {
            if (a.b) bq = *reinterpret_cast<gvol>((uintptr_t)(a.b + row));
            if (a.gam) gq = *reinterpret_cast<gvol>((uintptr_t)(a.gam + row));
        }
    };
    auto park = [&](int buf, int64_t row, T bq, T gq) {
        if (lane < G) {
            if (GATHER) {
                offA[buf * WAVE + lane] = row * a.ld;
                offT[buf * WAVE + lane] = row * (int64_t)d;
            }
            bis[buf * WAVE + lane] = bq;
            gis[buf * WAVE + lane] = gq;
        }
    };
    // the elements of group g (staged in `buf`): data row and, for the Finito batch, table row, one element-wise load each
    auto fetch = [&](T(&v)[SMALL_I], T(&sv)[TABLE ? SMALL_I : 1], int64_t g, int buf) {
        const int64_t left = a.nrows - g * G;
        const int nrg = left < G ? (int)left : G;
        const int64_t gbase = (a.row0 + g * G) * (int64_t)d;   // dense: the group's first element (ld == d)
#pragma unroll
        for (int i = 0; i < SMALL_I; ++i) {
            const bool on = live[i] && rrow[i] < nrg;
            const int r = on ? rrow[i] : 0;
            const int c = on ? col[i] : 0;
            const int64_t ea = GATHER ? offA[buf * WAVE + r] + c : gbase + (on ? lane + WAVE * i : 0);
            const T val = a.A ? __builtin_nontemporal_load(a.A + ea) : T(0);
            v[i] = on ? val : T(0);
            if (TABLE) {
                const int64_t et = GATHER ? offT[buf * WAVE + r] + c : ea;
                const T tval = __builtin_nontemporal_load(a.table + et);
                sv[i] = on ? tval : T(0);
            }
        }
    };

    T av[SMALL_I], avn[SMALL_I], sv[TABLE ? SMALL_I : 1], svn[TABLE ? SMALL_I : 1];
    int64_t g = (int64_t)blockIdx.x * ROWS_WAVES + wib;
    int cur = 0;   // LDS buffer of group g; (cur + 1) % 3 = group g + nwaves, (cur + 2) % 3 = group g + 2 nwaves
    int64_t rowp = 0;   // lane r < G: row r of the group after next (its index has arrived; scalars and parking still to come)
    if (g < ngroups) {
        // the start of the pipeline with as few DEPENDENT round trips as the data allow (a wave of a single batch often has one
        // group in all, and this is then its whole memory latency): indices first; a dense block's elements need nothing else and
        // travel with the scalars, gathered rows need their offsets parked before their elements can be requested
        const int64_t r0 = check_row(load_row(g)), r1 = check_row(load_row(g + nwaves));
        rowp = check_row(load_row(g + 2 * nwaves));
        T b0, g0, b1, g1;
        if (!GATHER) fetch(av, sv, g, 0);
        load_scal(r0, b0, g0);
        load_scal(r1, b1, g1);
        if (GATHER) {
            park(0, r0, T(0), T(0));                         // offsets now (the indices are here), scalars when they arrive
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's own LDS writes (no other wave touches its area)
            fetch(av, sv, g, 0);
        }
        park(0, r0, b0, g0);
        park(1, r1, b1, g1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (; g < ngroups; g += nwaves) {
        const int64_t left = a.nrows - g * G;
        const int nr = left < G ? (int)left : G;
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        const int64_t rown = load_row(g + 3 * nwaves);   // index: three groups ahead
        T bq, gq;
        load_scal(rowp, bq, gq);                          // scalars: two groups ahead (that index came in an iteration ago)
#pragma unroll
        for (int i = 0; i < SMALL_I; ++i) {
            prod[lane + WAVE * i] = av[i] * xcol[i];
            if (TWO) prod2[lane + WAVE * i] = av[i] * x2col[i];
        }
        const bool more = g + nwaves < ngroups;
        if (more) fetch(avn, svn, g + nwaves, nxt);   // its descriptors were staged an iteration ago
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const bool rowlive = myrow < nr;
            const int mr = rowlive ? myrow : 0;
            const T *pr = prod + mr * d;
            const T *pr2 = prod2 + mr * d;
            T d0 = T(0), d1 = T(0), e0 = T(0), e1 = T(0);
            int k = myq;
            for (; k + Q < d; k += 2 * Q) {
                d0 += pr[k];
                d1 += pr[k + Q];
                if (TWO) {
                    e0 += pr2[k];
                    e1 += pr2[k + Q];
                }
            }
            if (k < d) {
                d0 += pr[k];
                if (TWO) e0 += pr2[k];
            }
            T dot = d0 + d1, dot2 = e0 + e1;
            for (int m = Q >> 1; m > 0; m >>= 1) {
                dot += __shfl_xor(dot, m, WAVE);
                if (TWO) dot2 += __shfl_xor(dot2, m, WAVE);
            }
            if (rowlive && myq == 0) {
                const T bi = bis[cur * WAVE + myrow], gi = gis[cur * WAVE + myrow];
                const GradCoef<T> gc = grad_coef(a.loss, dot, bi, a.lam);
                if (TWO) {                                    // Finito_LFinito.jl:93-98
                    s1s[myrow] = gc.coef() - grad_coef(a.loss, dot2, bi, a.lam).coef();
                    ex += a.hat_gamma / gi;
                } else {                                      // Finito_basic.jl:110-117
                    s1s[myrow] = gc.s1;
                    cgs[myrow] = gi * a.invN;
                    rrs[myrow] = a.hat_gamma / gi;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < SMALL_I; ++i) {
            if (TWO) {
                acc[i] += s1s[rrow[i]] * av[i];              // unconditional: dead elements carry a = 0 and a finite scalar
                continue;
            }
            if (!(live[i] && rrow[i] < nr)) continue;
            const T s1 = s1s[rrow[i]];
            const T tv = xcol[i] - cgs[rrow[i]] * ((av[i] * s1) * s2);   // GradCoef::elem, as the wave- / workgroup-per-row kernels
            acc[i] += (tv - sv[i]) * rrs[rrow[i]];
            const int64_t et = GATHER ? offT[cur * WAVE + rrow[i]] + col[i] : (a.row0 + g * G) * (int64_t)d + lane + WAVE * i;
            __builtin_nontemporal_store(tv, a.table + et);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row scalars and this group's descriptors are rewritten next
        park(nx2, rowp, bq, gq);                             // (buffer nx2 held the group before this one: nobody reads it any more)
        rowp = check_row(rown);
        if (more) {
#pragma unroll
            for (int i = 0; i < SMALL_I; ++i) {
                av[i] = avn[i];
                if (TABLE) sv[i] = svn[i];
            }
        }
        cur = nxt;
    }

    // columns: element (lane, i) -> prod[e]; lane c then adds rows 0..G-1 of column c in row order (as rows_small_kernel)
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i)
        if (live[i]) prod[lane + WAVE * i] = acc[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T colsum[(256 + WAVE - 1) / WAVE];
#pragma unroll
    for (int q = 0; q < (256 + WAVE - 1) / WAVE; ++q) {
        const int c = lane + WAVE * q;
        T sacc = T(0);
        if (c < d)
            for (int r = 0; r < G; ++r) sacc += prod[r * d + c];
        colsum[q] = sacc;
    }
    ex = wave_allsum(ex);
    __syncthreads();   // every wave is done with its private area; the block's area now holds the per-wave column sums
    __shared__ T red_extra_smallb[ROWS_WAVES];
    T *base = reinterpret_cast<T *>(smallb_raw);
#pragma unroll
    for (int q = 0; q < (256 + WAVE - 1) / WAVE; ++q) {
        const int c = lane + WAVE * q;
        if (c < d) base[wib * d + c] = colsum[q];
    }
    if (lane == 0) red_extra_smallb[wib] = ex;
    __syncthreads();
    T *pout = a.partial + (int64_t)blockIdx.x * a.pstride;
    for (int c = threadIdx.x; c < d; c += ROWS_BLOCK) {
        T sacc = base[c];
        for (int w = 1; w < ROWS_WAVES; ++w) sacc += base[w * d + c];
        pout[c] = sacc;
    }
    if (threadIdx.x == 0) {
        T e2 = T(0);
        for (int w = 0; w < ROWS_WAVES; ++w) e2 += red_extra_smallb[w];
        a.pextra[blockIdx.x] = e2;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Short-row variant of the gradient sweeps (GRAD / GRAD2): R rows per wave per iteration.
// The chip streams fastest with about 8 KiB of loads in flight per wave at one block per CU (tools/tune_sweep.py); a
// 4 KiB (d=1024 fp32) or shorter row leaves a wave with too little in flight and too much per-row latency (dot ->
// DPP chain -> link function) exposed.  Here a wave fetches R rows at once (R*K 16-byte loads per lane in flight) and
// runs their R reductions interleaved, so R*rowbytes = 8 KiB whatever the row size.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int K, int R, int MODE>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_multi_kernel(RowsArgs<T> a_by_value)
{
    (void)a_by_value;
    constexpr bool TWO_MODE = (MODE == RM_GRAD2);
    CIAO_KERNARG0(RowsArgs<T>, ka);
    // What the sweep loop reads, loaded ONCE into scalar registers (sgpr_pin).  Read in place, hipcc re-loaded the fields inside the loop,
    // each behind an lgkmcnt(0) -- the one in front of the second row's address also waited for the first row's loads: the fp32
    // d = 1024 sweep went from 5.90 to 7.15 ms (profiles/r05_kernarg_ab.txt).  Prologue and epilogue fields are read where they are used.
    const struct {
        const T *A, *b;
        int64_t ld, row0, nrows, N;
        const int64_t *idx;
        int loss, want_fval;
        T lam, gam_uniform, hat_gamma;
        const T *gam;
        T *rowdot_out;
        int *errflag;
    } a = {sgpr_pin_global(ka.A), sgpr_pin_global(ka.b), sgpr_pin(ka.ld), sgpr_pin(ka.row0), sgpr_pin(ka.nrows), sgpr_pin(ka.N),
           sgpr_pin_global(ka.idx), sgpr_pin(ka.loss), sgpr_pin(ka.want_fval), sgpr_pin(ka.lam),
           TWO_MODE ? sgpr_pin(ka.gam_uniform) : T(0), TWO_MODE ? sgpr_pin(ka.hat_gamma) : T(0),
           TWO_MODE ? sgpr_pin_global(ka.gam) : nullptr, TWO_MODE ? nullptr : sgpr_pin_global(ka.rowdot_out), sgpr_pin_global(ka.errflag)};
    using V = typename VecOf<T>::type;
    constexpr int VEC = VecOf<T>::N;
    constexpr int D = K * WAVE * VEC;
    constexpr bool TWO = (MODE == RM_GRAD2);
    static_assert(MODE == RM_GRAD || MODE == RM_GRAD2, "gradient sweeps only");

    __shared__ __attribute__((aligned(16))) T lds[(TWO ? 2 : 1) * D];
    __shared__ T red_extra[ROWS_WAVES];

    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * ROWS_WAVES;
    const int64_t wave = (int64_t)blockIdx.x * ROWS_WAVES + wib;

    for (int e = threadIdx.x; e < D; e += ROWS_BLOCK) {
        lds[e] = ka.x1[e];
        if (TWO) lds[D + e] = ka.x2[e];
    }
    __syncthreads();

    V acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = V(T(0));
    T extra = T(0);

    // group g of this wave = rows q0 + r, r < R, with q0 = (g*nwaves + wave)*R : R consecutive rows (one 8 KiB stretch)
    for (int64_t q0 = wave * R; q0 < a.nrows; q0 += nwaves * R) {
        V cur[R][K];
        int64_t row[R];
        T bi[R], gi[R];   // (gamma_i requested with the row: at the end of the group it was a dependent load, and kept row[] live)
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t q = q0 + r;
            live[r] = q < a.nrows;
            int64_t rr = live[r] ? (a.idx ? a.idx[q] : a.row0 + q) : 0;
            if (a.idx && live[r] && (uint64_t)rr >= (uint64_t)a.N) {
                if (lane == 0) *a.errflag = 1;
                rr = 0;
            }
            row[r] = rr;
            if (a.A) {
                const V *ap = reinterpret_cast<const V *>(a.A + rr * a.ld);
#pragma unroll
                for (int k = 0; k < K; ++k) cur[r][k] = __builtin_nontemporal_load(&ap[k * WAVE + lane]);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) cur[r][k] = V(T(0));
            }
            bi[r] = (a.b && live[r]) ? a.b[rr] : T(0);
            gi[r] = (TWO && a.gam && live[r]) ? a.gam[rr] : a.gam_uniform;
        }
        int xl = lane;
        asm volatile("" : "+v"(xl));
        const V *x1v = reinterpret_cast<const V *>(lds) + xl;
        const V *x2v = reinterpret_cast<const V *>(lds + D) + xl;
        T d1[R], d2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) d1[r] = d2[r] = T(0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const V xv = x1v[k * WAVE];
            V yv;
            if (TWO) yv = x2v[k * WAVE];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    d1[r] += cur[r][k][v] * xv[v];
                    if (TWO) d2[r] += cur[r][k][v] * yv[v];
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            d1[r] = wave_allsum(d1[r]);
            if (TWO) d2[r] = wave_allsum(d2[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const GradCoef<T> g1 = grad_coef(a.loss, d1[r], bi[r], a.lam);
            T c = g1.coef();
            if (TWO) c -= grad_coef(a.loss, d2[r], bi[r], a.lam).coef();
            if (!live[r]) c = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += c * cur[r][k];
            if (live[r]) {
                if (MODE == RM_GRAD) {
                    if (a.want_fval) extra += loss_value(a.loss, d1[r], bi[r], a.lam);
                    if (a.rowdot_out && lane == 0) a.rowdot_out[row[r]] = d1[r];
                } else {
                    extra += a.hat_gamma / gi[r];
                }
            }
        }
    }

    __syncthreads();
    V *red = reinterpret_cast<V *>(lds);
    for (int w = 0; w < ROWS_WAVES; ++w) {
        if (wib == w) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (w == 0)
                    red[k * WAVE + lane] = acc[k];
                else
                    red[k * WAVE + lane] += acc[k];
            }
        }
        __syncthreads();
    }
    if (lane == 0) red_extra[wib] = extra;
    __syncthreads();
    T *pout = ka.partial + (int64_t)blockIdx.x * ka.pstride;
    for (int e = threadIdx.x; e < D; e += ROWS_BLOCK) pout[e] = lds[e];
    if (threadIdx.x == 0) {
        T ex = T(0);
        for (int w = 0; w < ROWS_WAVES; ++w) ex += red_extra[w];
        ka.pextra[blockIdx.x] = ex;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Generic path: any d, any alignment.  Lane l owns elements l, l+64, ...; the per-wave accumulator lives in LDS
// (wave-private region, so no atomics and no barriers in the loop); the row is read twice (second read is an L1/L2
// hit).  Correctness path for small / odd shapes (e.g. the reference's own N=6,d=3 and N=8,d=5 tests).
// Dynamic LDS layout: x1[d] | x2[d] (if TWO) | acc[NW][d]
// ------------------------------------------------------------------------------------------------------------------
// GLOBAL_ACC (rows too long for LDS: beyond ~72 KiB with one iterate, e.g. d = 16384 fp64): the iterate(s) are read from
// global memory (cache-resident) and every WAVE accumulates into a partial d-vector of its own in the workspace -- one
// partial per wave instead of one per block, nothing staged in LDS.  Each element of a wave's partial is only ever touched by
// the lane that owns it (k = lane mod 64), so plain program order is all the ordering it needs.
template <typename T, int NW, int MODE, bool GLOBAL_ACC = false>
__global__ void __launch_bounds__(NW *WAVE) rows_generic_kernel(RowsArgs<T> a)
{
    constexpr bool TWO = (MODE == RM_GRAD2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    const int64_t d = a.d;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    const T *x1s, *x2s;
    T *acc;
    if constexpr (GLOBAL_ACC) {
        x1s = a.x1;
        x2s = a.x2;
        acc = a.partial + ((int64_t)blockIdx.x * NW + wib) * a.pstride;
        for (int64_t e = lane; e < d; e += WAVE) acc[e] = T(0);
    } else {
        T *x1w = smem;
        T *x2w = smem + d;
        T *accs = smem + (TWO ? 2 : 1) * d;
        acc = accs + (int64_t)wib * d;
        for (int64_t e = threadIdx.x; e < d; e += NW * WAVE) {
            x1w[e] = a.x1[e];
            if (TWO) x2w[e] = a.x2[e];
        }
        for (int64_t e = threadIdx.x; e < (int64_t)NW * d; e += NW * WAVE) accs[e] = T(0);
        __syncthreads();
        x1s = x1w;
        x2s = x2w;
    }

    T extra = T(0);
    for (int64_t q = (int64_t)blockIdx.x * NW + wib; q < a.nrows; q += nwaves) {
        int64_t row = a.idx ? a.idx[q] : a.row0 + q;
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (lane == 0) *a.errflag = 1;
            row = 0;
        }
        const T *ap = a.A ? a.A + row * a.ld : nullptr;
        const T bi = a.b ? a.b[row] : T(0);
        T d1 = T(0), d2 = T(0);
        for (int64_t e = lane; e < d; e += WAVE) {
            const T av = ap ? ap[e] : T(0);
            d1 += av * x1s[e];
            if (TWO) d2 += av * x2s[e];
        }
        d1 = wave_allsum(d1);
        if (TWO) d2 = wave_allsum(d2);
        const GradCoef<T> g1 = grad_coef(a.loss, d1, bi, a.lam);
        T *tp = a.table ? a.table + row * d : nullptr;
        if (MODE == RM_GRAD) {
            const T c = g1.coef();
            for (int64_t e = lane; e < d; e += WAVE) acc[e] += c * (ap ? ap[e] : T(0));
            if (a.want_fval) extra += loss_value(a.loss, d1, bi, a.lam);
            if (a.rowdot_out && lane == 0) a.rowdot_out[row] = d1;
        } else if (MODE == RM_GRAD2) {
            const GradCoef<T> g2 = grad_coef(a.loss, d2, bi, a.lam);
            const T c = g1.coef() - g2.coef();
            for (int64_t e = lane; e < d; e += WAVE) acc[e] += c * (ap ? ap[e] : T(0));
            const T gi = a.gam ? a.gam[row] : a.gam_uniform;
            extra += a.hat_gamma / gi;
        } else if (MODE == RM_SAGA_INIT) {
            for (int64_t e = lane; e < d; e += WAVE) {
                const T gv = g1.elem(ap ? ap[e] : T(0));
                tp[e] = gv;
                acc[e] += gv;
            }
        } else if (MODE == RM_AFINITO_INIT) {
            T sa = T(0), n2 = T(0);
            for (int64_t e = lane; e < d; e += WAVE) {
                const T av = ap ? ap[e] : T(0);
                sa += av;
                n2 += av * av;
            }
            sa = wave_allsum(sa);
            n2 = wave_allsum(n2);
            const T c0 = g1.coef();
            const T c1 = grad_coef(a.loss, d1 + sa, bi, a.lam).coef();
            T nmg = fabs2(c1 - c0) * fsqrt(n2);
            const T gov = a.gam ? a.gam[row] : T(0);   // host-resolved stepsize (second pass after a re-probe), see the fast kernel
            bool degenerate = false;
            if (!(gov > T(0)) && nmg < Eps<T>::value) {
                if (lane == 0) *a.errflag = 2;
                nmg = Eps<T>::value;
                degenerate = true;
            }
            const T gi = gov > T(0) ? gov : (T)((double)a.alpha / (((double)nmg / sqrt((double)d)) / a.Nd));   // Float64 as in the reference (:73, :86-88)
            const T rinv = T(1) / gi;
            const T cn = c0 * a.invN;
            for (int64_t e = lane; e < d; e += WAVE) {
                const T xv = x1s[e];
                tp[e] = xv;
                acc[e] += xv * rinv - cn * (ap ? ap[e] : T(0));
            }
            extra += rinv;
            if (lane < 4) {   // four identical copies, one per wave of the step kernel (afinito_kernels.h)
                T *mp = a.meta + (row * 4 + lane) * 4;
                mp[0] = c0;
                mp[1] = loss_value(a.loss, d1, bi, a.lam);
                mp[2] = degenerate ? T(-1) : gi;
                mp[3] = d1;
            }
        } else {
            const T gi = a.gam ? a.gam[row] : a.gam_uniform;
            const T cg = gi * a.invN;
            const T rr = (MODE == RM_FINITO_INIT) ? T(1) / gi : a.hat_gamma / gi;
            for (int64_t e = lane; e < d; e += WAVE) {
                const T tv = x1s[e] - cg * g1.elem(ap ? ap[e] : T(0));
                if (MODE == RM_FINITO_INIT)
                    acc[e] += tv * rr;
                else
                    acc[e] += (tv - tp[e]) * rr;
                tp[e] = tv;
            }
        }
    }

    if constexpr (GLOBAL_ACC) {
        if (lane == 0) a.pextra[(int64_t)blockIdx.x * NW + wib] = extra;   // one partial (and one extra) per wave
    } else {
        __shared__ T red_extra[NW];
        T *accs = smem + (TWO ? 2 : 1) * d;
        if (lane == 0) red_extra[wib] = extra;   // extra is wave-uniform
        __syncthreads();
        T *pout = a.partial + (int64_t)blockIdx.x * a.pstride;
        for (int64_t e = threadIdx.x; e < d; e += NW * WAVE) {
            T s = accs[e];
            for (int w = 1; w < NW; ++w) s += accs[(int64_t)w * d + e];
            pout[e] = s;
        }
        if (threadIdx.x == 0) {
            T ex = T(0);
            for (int w = 0; w < NW; ++w) ex += red_extra[w];
            a.pextra[blockIdx.x] = ex;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Complex T (CIAO_LOSS_LS_COMPLEX): every complex vector is (re, im) pairs of T.  Correctness path, one wave per row, per-wave
// accumulators in the workspace (as rows_generic_kernel<..., GLOBAL_ACC>), the iterate(s) read from global memory.
//   res = a_i . x - b_i  (complex, no conjugation: A*x);   grad f_i(x)_k = (conj(a_k) res) lam;   f_i = lam/2 |res|^2
// Modes GRAD, GRAD2, SAGA_INIT, FINITO_INIT, FINITO_BATCH, AFINITO_INIT with the formulas of the real kernels, pair by pair.
// AFINITO_INIT: the probe point is x0 .+ one(R) (Finito_adaptive.jl:74) -- a REAL one added to every complex entry, so
// a.(xeps - x0) = sum_k a_k; sqrt(length(x0)) counts complex entries (:86).  The per-sample scalars are complex now: the 16
// meta slots of a sample hold {Re c, f_i, gamma, Re a.x_i} in copies 0 and 2 and {Im c, f_i, gamma, Im a.x_i} in copies 1
// and 3 (c = lam res, grad f_i = conj(a_i) c); gamma stays at slot 2 of every copy, where the host reads it.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_cplx_kernel(RowsArgs<T> a)
{
    constexpr bool TWO = (MODE == RM_GRAD2);
    constexpr bool AF = (MODE == RM_AFINITO_INIT);
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t dc = a.d / 2;
    const int64_t nwaves = (int64_t)gridDim.x * ROWS_WAVES;
    T *acc = a.partial + ((int64_t)blockIdx.x * ROWS_WAVES + wib) * a.pstride;
    for (int64_t e = lane; e < a.d; e += WAVE) acc[e] = T(0);
    T extra = T(0);
    for (int64_t q = (int64_t)blockIdx.x * ROWS_WAVES + wib; q < a.nrows; q += nwaves) {
        int64_t row = a.idx ? a.idx[q] : a.row0 + q;
        if ((uint64_t)row >= (uint64_t)a.N) {
            if (lane == 0) *a.errflag = 1;
            row = 0;
        }
        const T *ap = a.A + row * a.ld;
        const T br = a.b[2 * row], bi = a.b[2 * row + 1];
        T s1r = T(0), s1i = T(0), s2r = T(0), s2i = T(0), n2 = T(0);
        for (int64_t e = lane; e < dc; e += WAVE) {
            const T ar = ap[2 * e], ai = ap[2 * e + 1];
            const T xr = a.x1[2 * e], xi = a.x1[2 * e + 1];
            s1r += ar * xr - ai * xi;
            s1i += ar * xi + ai * xr;
            if (TWO) {
                const T yr = a.x2[2 * e], yi = a.x2[2 * e + 1];
                s2r += ar * yr - ai * yi;
                s2i += ar * yi + ai * yr;
            }
            if (AF) {            // s2 = sum_k a_k = a.(xeps - x0);  n2 = ||a||^2
                s2r += ar;
                s2i += ai;
                n2 += ar * ar + ai * ai;
            }
        }
        s1r = wave_allsum(s1r);
        s1i = wave_allsum(s1i);
        if (TWO || AF) {
            s2r = wave_allsum(s2r);
            s2i = wave_allsum(s2i);
        }
        if (AF) n2 = wave_allsum(n2);
        const T r1r = s1r - br, r1i = s1i - bi;          // res = A x - b
        const T r2r = s2r - br, r2i = s2i - bi;
        T *tp = a.table ? a.table + row * a.d : nullptr;
        T gi = (MODE == RM_GRAD || MODE == RM_SAGA_INIT) ? T(1) : (a.gam ? a.gam[row] : a.gam_uniform);
        if (AF) {                                                             // Finito_adaptive.jl:73-88
            const T c0r = a.lam * r1r, c0i = a.lam * r1i;
            const T c1r = a.lam * ((s1r + s2r) - br), c1i = a.lam * ((s1i + s2i) - bi);
            T nmg = fhypot(c1r - c0r, c1i - c0i) * fsqrt(n2);                 // || conj(a) (c1 - c0) || = |c1 - c0| ||a||
            const T gov = a.gam ? a.gam[row] : T(0);                          // host-resolved stepsize (after a re-probe)
            bool degenerate = false;
            if (!(gov > T(0)) && nmg < Eps<T>::value) {
                if (lane == 0) *a.errflag = 2;
                nmg = Eps<T>::value;
                degenerate = true;
            }
            gi = gov > T(0) ? gov : (T)((double)a.alpha / (((double)nmg / sqrt((double)dc)) / a.Nd));
            const T rinv = T(1) / gi;
            const T fv = (a.lam / T(2)) * (r1r * r1r + r1i * r1i);
            for (int64_t e = lane; e < dc; e += WAVE) {
                const T ar = ap[2 * e], ai = ap[2 * e + 1];
                T gr, gim;
                cgrad_elem(ar, ai, r1r, r1i, a.lam, gr, gim);
                const T xr = a.x1[2 * e], xi = a.x1[2 * e + 1];
                tp[2 * e] = xr;                                                // :68  s_i = x0
                tp[2 * e + 1] = xi;
                acc[2 * e] += xr * rinv - gr * a.invN;                         // :92  sum(s ./ gamma) - sum(grad f) / N
                acc[2 * e + 1] += xi * rinv - gim * a.invN;
            }
            extra += rinv;
            if (lane < 4) {
                T *mp = a.meta + (row * 4 + lane) * 4;
                mp[0] = (lane & 1) ? c0i : c0r;
                mp[1] = fv;
                mp[2] = degenerate ? T(-1) : gi;
                mp[3] = (lane & 1) ? s1i : s1r;
            }
            continue;
        }
        if (MODE == RM_GRAD && a.want_fval) extra += (a.lam / T(2)) * (r1r * r1r + r1i * r1i);
        if (MODE == RM_GRAD2) extra += a.hat_gamma / gi;
        const T cg = gi * a.invN;
        const T rr = (MODE == RM_FINITO_INIT) ? T(1) / gi : a.hat_gamma / gi;
        for (int64_t e = lane; e < dc; e += WAVE) {
            const T ar = ap[2 * e], ai = ap[2 * e + 1];
            T gr, gim;
            cgrad_elem(ar, ai, r1r, r1i, a.lam, gr, gim);
            if (MODE == RM_GRAD) {
                acc[2 * e] += gr;
                acc[2 * e + 1] += gim;
            } else if (MODE == RM_GRAD2) {
                T hr, hi;
                cgrad_elem(ar, ai, r2r, r2i, a.lam, hr, hi);
                acc[2 * e] += gr - hr;
                acc[2 * e + 1] += gim - hi;
            } else if (MODE == RM_SAGA_INIT) {
                tp[2 * e] = gr;
                tp[2 * e + 1] = gim;
                acc[2 * e] += gr;
                acc[2 * e + 1] += gim;
            } else {
                const T tr = a.x1[2 * e] - cg * gr, ti = a.x1[2 * e + 1] - cg * gim;
                if (MODE == RM_FINITO_INIT) {
                    acc[2 * e] += tr * rr;
                    acc[2 * e + 1] += ti * rr;
                } else {
                    acc[2 * e] += (tr - tp[2 * e]) * rr;
                    acc[2 * e + 1] += (ti - tp[2 * e + 1]) * rr;
                }
                tp[2 * e] = tr;
                tp[2 * e + 1] = ti;
            }
        }
    }
    if (lane == 0) a.pextra[(int64_t)blockIdx.x * ROWS_WAVES + wib] = extra;
}

// ------------------------------------------------------------------------------------------------------------------
// Complex T, the streaming kernel: rows_split_kernel's layout (one workgroup per row, thread t owns the 16-byte chunks
// t + 256 j of every vector, accumulators and the iterate in registers, one raw barrier per row) with complex arithmetic
// on the chunk's (re, im) pairs -- one complex per chunk for fp64, two for fp32.  Rows of whole 16-byte chunks up to 64 KiB;
// dead chunks of the last group are masked.  Everything else complex (odd complex counts in fp32, longer rows, the adaptive
// init) stays on rows_cplx_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int J, int MODE>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_csplit_kernel(RowsArgs<T> a)
{
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NC = VEC / 2;                       // complex numbers per chunk
    using V = typename ChunkOf<T, VEC>::type;
    constexpr bool TWO = (MODE == RM_GRAD2);
    constexpr bool TABLE = (MODE == RM_SAGA_INIT || MODE == RM_FINITO_INIT || MODE == RM_FINITO_BATCH);
    constexpr bool TREAD = (MODE == RM_FINITO_BATCH);
    static_assert(MODE != RM_AFINITO_INIT, "the adaptive init keeps to rows_cplx_kernel");

    __shared__ T red[2][ROWS_WAVES][4];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nchunks = a.d / VEC;

    bool ok[J];
    V x1[J], x2[J], acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        ok[j] = tid + j * ROWS_BLOCK < nchunks;
        x1[j] = ok[j] ? reinterpret_cast<const V *>(a.x1)[tid + j * ROWS_BLOCK] : V(T(0));
        x2[j] = (TWO && ok[j]) ? reinterpret_cast<const V *>(a.x2)[tid + j * ROWS_BLOCK] : V(T(0));
        acc[j] = V(T(0));
    }
    T extra = T(0);
    int par = 0;
    for (int64_t q = blockIdx.x; q < a.nrows; q += gridDim.x) {
        int64_t row = a.idx ? a.idx[q] : a.row0 + q;
        if (a.idx && (uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        V *sp = TABLE ? reinterpret_cast<V *>(a.table + row * a.d) : nullptr;
        const V *ap = reinterpret_cast<const V *>(a.A + row * a.ld);
        V ar[J], sr[J];
#pragma unroll
        for (int j = 0; j < J; ++j) ar[j] = ok[j] ? __builtin_nontemporal_load(&ap[tid + j * ROWS_BLOCK]) : V(T(0));
        if (TREAD) {
#pragma unroll
            for (int j = 0; j < J; ++j) sr[j] = ok[j] ? __builtin_nontemporal_load(&sp[tid + j * ROWS_BLOCK]) : V(T(0));
        }
        const T br = a.b[2 * row], bi = a.b[2 * row + 1];
        const T gi = (MODE == RM_GRAD || MODE == RM_SAGA_INIT) ? T(1) : (a.gam ? a.gam[row] : a.gam_uniform);
        T s1r = T(0), s1i = T(0), s2r = T(0), s2i = T(0);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const T pr = ar[j][2 * c], pi = ar[j][2 * c + 1];
                s1r += pr * x1[j][2 * c] - pi * x1[j][2 * c + 1];
                s1i += pr * x1[j][2 * c + 1] + pi * x1[j][2 * c];
                if (TWO) {
                    s2r += pr * x2[j][2 * c] - pi * x2[j][2 * c + 1];
                    s2i += pr * x2[j][2 * c + 1] + pi * x2[j][2 * c];
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
        __builtin_amdgcn_s_barrier();
        s1r = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
        s1i = (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]);
        if (TWO) {
            s2r = (red[par][0][2] + red[par][1][2]) + (red[par][2][2] + red[par][3][2]);
            s2i = (red[par][0][3] + red[par][1][3]) + (red[par][2][3] + red[par][3][3]);
        }
        par ^= 1;
        const T r1r = s1r - br, r1i = s1i - bi;          // res = a.x - b
        const T r2r = s2r - br, r2i = s2i - bi;
        if (MODE == RM_GRAD && a.want_fval) extra += (a.lam / T(2)) * (r1r * r1r + r1i * r1i);
        if (MODE == RM_GRAD2) extra += a.hat_gamma / gi;
        const T cg = gi * a.invN;
        const T rr = (MODE == RM_FINITO_INIT) ? T(1) / gi : a.hat_gamma / gi;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            V tv;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                T gr, gim;
                cgrad_elem(ar[j][2 * c], ar[j][2 * c + 1], r1r, r1i, a.lam, gr, gim);
                if (MODE == RM_GRAD) {
                    acc[j][2 * c] += gr;
                    acc[j][2 * c + 1] += gim;
                } else if (MODE == RM_GRAD2) {
                    T hr, hi;
                    cgrad_elem(ar[j][2 * c], ar[j][2 * c + 1], r2r, r2i, a.lam, hr, hi);
                    acc[j][2 * c] += gr - hr;
                    acc[j][2 * c + 1] += gim - hi;
                } else if (MODE == RM_SAGA_INIT) {
                    tv[2 * c] = gr;
                    tv[2 * c + 1] = gim;
                    acc[j][2 * c] += gr;
                    acc[j][2 * c + 1] += gim;
                } else {
                    const T tr = x1[j][2 * c] - cg * gr, ti = x1[j][2 * c + 1] - cg * gim;
                    if (MODE == RM_FINITO_INIT) {
                        acc[j][2 * c] += tr * rr;
                        acc[j][2 * c + 1] += ti * rr;
                    } else {
                        acc[j][2 * c] += (tr - sr[j][2 * c]) * rr;
                        acc[j][2 * c + 1] += (ti - sr[j][2 * c + 1]) * rr;
                    }
                    tv[2 * c] = tr;
                    tv[2 * c + 1] = ti;
                }
            }
            if (TABLE && ok[j]) TSTORE(tv, &sp[tid + j * ROWS_BLOCK]);
        }
    }
    V *pout = reinterpret_cast<V *>(a.partial + (int64_t)blockIdx.x * a.pstride);
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) PSTORE(acc[j], &pout[tid + j * ROWS_BLOCK]);
    if (tid == 0) a.pextra[blockIdx.x] = extra;
}

// ------------------------------------------------------------------------------------------------------------------
// finalize: sum the per-block partials in a fixed order and apply the epilogue.
// A batch step is rows kernel -> finalize -> next rows kernel, so for batches of a few hundred rows this kernel is half of
// the step (rocprofv3, r = 256, d = 4096 fp32: rows 6.4 us, finalize 5.0 us with the first version of this kernel, which
// made two dependent rounds of 4-byte loads and fetched the epilogue's operands only after the reduction).  Now:
//   - a block covers one 128-byte line of every partial row (8 lanes x 16 B) with 32 slices of partial rows, so with up to
//     256 partials every thread has ALL its loads (<= 8 x 16 B) in flight at once: one memory round trip;
//   - the epilogue's own operands (acc_in, u, v, pw) are requested before the partials, not after the reduction;
//   - the order of the additions is fixed (per thread: pairwise over its loads; then 4 groups of 8 slices; then the 4 group
//     sums pairwise), so results are bitwise reproducible run to run.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FIN_THREADS = 256;
constexpr int FIN_BYTES = 128;   // bytes of one partial row that a finalize block covers (one L2 line)

// NT = FIN_THREADS (up to 256 partials in one round) or FIN_THREADS_MANY = 1024 (up to 1024 per round: the kernels that fill the chip with
// several small workgroups per CU write 512-2048 partials, and every further round is a memory round trip -- 1536 partials of 50 Float64
// took 13 us on 256 threads, profiles/r05_wrow_blocks.txt).  Same tree, four times as wide; the launcher picks by the number of partials.
constexpr int FIN_THREADS_MANY = 1024;

template <typename T, int NT = FIN_THREADS>
__global__ void __launch_bounds__(NT)
    finalize_kernel(const T *__restrict__ partial, int64_t pstride, int nparts, const T *__restrict__ pextra,
                    int64_t d, T *raw_out, Epilogue<T> ep, PeerDev peers)
{
    constexpr int VEC = 16 / sizeof(T);
    using V = typename ChunkOf<T, VEC>::type;
    constexpr int LANES = FIN_BYTES / 16;         // 16-byte chunks of the line
    constexpr int SLICES = NT / LANES;                 // 32 (128)
    constexpr int GROUPS = SLICES / 8;                 // 4 (16)
    constexpr int COLS = FIN_BYTES / sizeof(T);
    constexpr int U = 8;                               // loads in flight per thread and round
    __shared__ V lds[SLICES][LANES];
    __shared__ V lds2[GROUPS][LANES];
    __shared__ T lds_extra;
    const int tx = threadIdx.x % LANES;
    const int ty = threadIdx.x / LANES;
    const int64_t col = (int64_t)blockIdx.x * COLS + tx * VEC;   // this thread's first column (rows are padded to pstride)

    // the epilogue's operands for the VEC columns this thread may own at the end: requested first
    T pa[VEC], pu[VEC], pv[VEC], pp[VEC];
    if (ty == 0 && !raw_out) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const bool in = col + v < d;
            pa[v] = (ep.acc_in && in) ? ep.acc_in[col + v] : T(0);
            pu[v] = (ep.u && in) ? ep.u[col + v] : T(0);
            pv[v] = (ep.v && in) ? ep.v[col + v] : T(0);
            pp[v] = (ep.pw && in) ? ep.pw[col + v] : T(0);
        }
    }

    V s = V(T(0));
    if (col < pstride) {
        const V *pp0 = reinterpret_cast<const V *>(partial + col);
        const int64_t vstride = pstride / VEC;   // pstride is a multiple of 64 elements
        for (int p0 = ty; p0 < nparts; p0 += U * SLICES) {
            V v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * SLICES;
                v[u] = p < nparts ? pp0[(int64_t)p * vstride] : V(T(0));
            }
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    lds[ty][tx] = s;
    if (threadIdx.x >= NT - WAVE) {   // last wave: the extra scalar (all 64 lanes active)
        const int l = threadIdx.x & (WAVE - 1);
        T ex = T(0);
        for (int p = l; p < nparts; p += WAVE) ex += pextra[p];
        ex = wave_allsum(ex);
        if (l == 0) lds_extra = ex;
    }
    __syncthreads();
    if (ty < GROUPS) {
        V g = lds[ty * 8][tx];
#pragma unroll
        for (int j = 1; j < 8; ++j) g += lds[ty * 8 + j][tx];
        lds2[ty][tx] = g;
    }
    __syncthreads();
    if (ty == 0) {
        V tot = (lds2[0][tx] + lds2[1][tx]) + (lds2[2][tx] + lds2[3][tx]);
        if constexpr (GROUPS == 16)
            tot = (tot + ((lds2[4][tx] + lds2[5][tx]) + (lds2[6][tx] + lds2[7][tx]))) +
                  (((lds2[8][tx] + lds2[9][tx]) + (lds2[10][tx] + lds2[11][tx])) + ((lds2[12][tx] + lds2[13][tx]) + (lds2[14][tx] + lds2[15][tx])));
        const T extra = lds_extra;
        const bool cpairs = !raw_out && ep.z_out && ep.g.kind == CIAO_PROX_L1_COMPLEX;   // (re, im) pairs: d even, col even
#pragma unroll
        for (int v = 0; v < VEC; v += 2) {
            if (cpairs) {
                if (col + v + 1 >= d) continue;
                const T t0 = epilogue_pre(ep, col + v, tot[v], extra, pa[v], pu[v], pv[v], pp[v]);
                const T t1 = epilogue_pre(ep, col + v + 1, tot[v + 1], extra, pa[v + 1], pu[v + 1], pv[v + 1], pp[v + 1]);
                const T tau = epilogue_tau(ep, extra);
                T y0, y1;
                prox_cpair(tau * ep.g.lam, t0, t1, y0, y1);
                epilogue_post(ep, col + v, t0, tau, y0);
                epilogue_post(ep, col + v + 1, t1, tau, y1);
                continue;
            }
#pragma unroll
            for (int w = v; w < v + 2; ++w) {
                if (col + w >= d) continue;
                if (raw_out) {
                    raw_out[col + w] = tot[w];
                    if (col + w == 0) raw_out[d] = extra;
                    if (peers.world > 0) {   // the rank's raw sum straight into every rank's mailbox (peer_kernels.h)
                        peer_put(peers, col + w, tot[w]);
                        if (col + w == 0) peer_put(peers, d, extra);
                    }
                } else {
                    epilogue_apply_pre(ep, col + w, tot[w], extra, pa[w], pu[w], pv[w], pp[w]);
                }
            }
        }
    }
    if (peers.world > 0) peer_publish(peers);   // the last workgroup to get here releases this rank's flag in every mailbox
}

// epilogue on an already reduced (and all-reduced) raw sum: raw[0..d) + extra at raw[d]
template <typename T>
__global__ void __launch_bounds__(256) epilogue_kernel(const T *__restrict__ raw, int64_t d, Epilogue<T> ep)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ep.z_out && ep.g.kind == CIAO_PROX_L1_COMPLEX) {   // (re, im) pairs: thread k takes coordinates 2k and 2k+1
        if (2 * k + 1 < d) epilogue_apply2(ep, 2 * k, raw[2 * k], raw[2 * k + 1], raw[d]);
        return;
    }
    if (k < d) epilogue_apply(ep, k, raw[k], raw[d]);
}

}  // namespace ciao
