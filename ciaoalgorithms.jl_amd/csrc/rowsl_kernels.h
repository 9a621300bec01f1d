// rowsl_kernels.h -- the sweeps and batch steps over rows LONGER than one workgroup's registers hold (more than 64 KiB: beyond 8192
// fp64 / 16 384 fp32 elements): a CLUSTER of S workgroups per row (rows_long_kernel).
//
// Why.  rows_split_kernel (rows_kernels.h) ends at 4096 chunks of 16 bytes per row -- sixteen per thread, the row, the iterate and the
// accumulators in registers.  Beyond that the scalar generic kernel took over with its accumulators in global memory: 0.5-1.6 TB/s,
// while the chains on such rows (chain_wide_kernels.h) had already moved to several workgroups.  A row's work is a dot product over
// the row and an element-wise accumulation: both split by COLUMNS.  Workgroup (c, s) -- segment s of cluster c -- owns the chunks
// [s J 256, (s + 1) J 256) of every vector: its slice of the iterate(s) and of the accumulators lives in registers for the whole
// launch, the next row's slice is in flight while this one is used, and all the S workgroups of a cluster exchange per row is their
// partial dot product(s), through the mailbox of chain_wide_kernels.h (64-bit words of payload + sequence number, relaxed device-scope
// atomics, no fence; two parities; spins bounded in wall-clock time).  Unlike a chain, the rows of a sweep are independent: the
// clusters run side by side and several workgroups per CU hide the exchange's round trip, so the kernel streams.
//
// Results: rows_split_kernel's arithmetic element for element; the dot products are added in another order (slices, then segments in
// one fixed tree), and a cluster writes ONE partial d-vector (every workgroup its own columns), so finalize sums C partials.
// Residency: every workgroup of a cluster must be on the chip at once -- the grid never exceeds what the occupancy calculator says is
// resident (rowsl_launch.inc), and a workgroup that never sees another's word gives up after 4 s (error word 6).
#pragma once

#include "chain_wide_kernels.h"
#include "rows_kernels.h"

namespace ciao {

constexpr int LONG_SMAX = 64;   // segments per row (the exchange's fixed tree is one wave wide)
constexpr int LONG_NBUF = 4;    // mailbox buffers (a row's total is collected one row late: rowsl_kernels.h, reduce_publish)

struct LongArgs {
    unsigned long long *box;   // [LONG_NBUF][C][S][words]
    int S, C;
};

template <typename T, int J, int MODE>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_long_kernel(RowsArgs<T> a_by_value, LongArgs la)
{
    (void)a_by_value;
    CIAO_KERNARG0(RowsArgs<T>, ka);
    using V = typename VecOf<T>::type;
    using W = WideWord<T>;
    constexpr int VEC = VecOf<T>::N;
    constexpr bool TWO = (MODE == RM_GRAD2);
    constexpr bool TABLE = (MODE == RM_SAGA_INIT || MODE == RM_FINITO_INIT || MODE == RM_FINITO_BATCH);
    constexpr bool TREAD = (MODE == RM_FINITO_BATCH);
    constexpr bool LATE = (J == 4);   // the row's total one iteration late (below)
    constexpr bool GAMS = !(MODE == RM_GRAD || MODE == RM_SAGA_INIT);
    // what the row loop reads, loaded once and held in scalar registers (read in place hipcc re-loads fields inside the loop, each
    // load a scalar-cache round trip in front of the row's requests: d = 32 768 fp64 sweep 4.83 -> 4.27 TB/s, profiles/r05_kernarg_ab.txt);
    // the iterate, the partials and the other modes' fields are read where they are used and hold no register in the loop
    const struct {
        const T *A, *b;
        int64_t ld, d, row0, nrows, N;
        const int64_t *idx;
        int loss, want_fval;
        T lam, gam_uniform, invN, hat_gamma;
        const T *gam;
        T *table, *rowdot_out;
        int *errflag;
    } a = {sgpr_pin_global(ka.A), sgpr_pin_global(ka.b), sgpr_pin(ka.ld), sgpr_pin(ka.d), sgpr_pin(ka.row0), sgpr_pin(ka.nrows), sgpr_pin(ka.N),
           sgpr_pin_global(ka.idx), sgpr_pin(ka.loss), MODE == RM_GRAD ? sgpr_pin(ka.want_fval) : 0, sgpr_pin(ka.lam),
           GAMS ? sgpr_pin(ka.gam_uniform) : T(0), (TABLE && MODE != RM_SAGA_INIT) ? sgpr_pin(ka.invN) : T(0),
           (TWO || TREAD) ? sgpr_pin(ka.hat_gamma) : T(0), GAMS ? sgpr_pin_global(ka.gam) : nullptr,
           TABLE ? sgpr_pin_global(ka.table) : nullptr, MODE == RM_GRAD ? sgpr_pin_global(ka.rowdot_out) : nullptr, sgpr_pin_global(ka.errflag)};
    constexpr int NWORD = (TWO ? 2 : 1) * W::N;
    static_assert(MODE != RM_AFINITO_INIT, "the adaptive init keeps to the wave-per-row kernels");

    __shared__ T red[2][ROWS_WAVES][2];   // by row parity: iteration 0 has no second barrier
    __shared__ T tot[2];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = la.S, C = la.C;
    const int c = (int)blockIdx.x / S, s = (int)blockIdx.x - c * S;
    const int64_t nchunks = a.d / VEC;
    const int64_t coff = (int64_t)s * J * ROWS_BLOCK + tid;   // this thread's first chunk
    if (tid == 0) s_fail = 0;
    const bool full = ((int64_t)(s + 1) * J * ROWS_BLOCK <= nchunks);   // (workgroup-uniform)

    bool ok[J];
    V x1[J], x2[J], acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        ok[j] = coff + j * ROWS_BLOCK < nchunks;
        x1[j] = ok[j] ? reinterpret_cast<const V *>(ka.x1)[coff + j * ROWS_BLOCK] : V(T(0));
        x2[j] = (TWO && ok[j]) ? reinterpret_cast<const V *>(ka.x2)[coff + j * ROWS_BLOCK] : V(T(0));
        acc[j] = V(T(0));
    }
    T extra = T(0);
    __syncthreads();

    struct RowIn {
        V ar[J], sr[J];
        V *sp;
        T bi, gi;
        int64_t row;
    };
    // request everything this workgroup needs of row q at once: its slice of the row and of the table row, the row's scalars
    auto issue = [&](RowIn &x, int64_t q) {
        int64_t row = a.idx ? a.idx[q] : a.row0 + q;
        if (a.idx && (uint64_t)row >= (uint64_t)a.N) {
            if (tid == 0) *a.errflag = 1;
            row = 0;
        }
        x.row = row;
        x.sp = TABLE ? reinterpret_cast<V *>(a.table + row * a.d) + coff : nullptr;
        if (!a.A) {
#pragma unroll
            for (int j = 0; j < J; ++j) x.ar[j] = V(T(0));
        } else if (full) {   // every thread's J chunks lie inside the row (all segments but possibly the last): no predication
            const V *ap = reinterpret_cast<const V *>(a.A + row * a.ld) + coff;
#pragma unroll
            for (int j = 0; j < J; ++j) x.ar[j] = __builtin_nontemporal_load(&ap[j * ROWS_BLOCK]);
        } else {
            const V *ap = reinterpret_cast<const V *>(a.A + row * a.ld) + coff;
#pragma unroll
            for (int j = 0; j < J; ++j) x.ar[j] = ok[j] ? __builtin_nontemporal_load(&ap[j * ROWS_BLOCK]) : V(T(0));
        }
        if (TREAD) {
            if (full) {
#pragma unroll
                for (int j = 0; j < J; ++j) x.sr[j] = __builtin_nontemporal_load(&x.sp[j * ROWS_BLOCK]);
            } else {
#pragma unroll
                for (int j = 0; j < J; ++j) x.sr[j] = ok[j] ? __builtin_nontemporal_load(&x.sp[j * ROWS_BLOCK]) : V(T(0));
            }
        }
        x.bi = a.b ? a.b[row] : T(0);
        x.gi = (MODE == RM_GRAD || MODE == RM_SAGA_INIT) ? T(1) : (a.gam ? a.gam[row] : a.gam_uniform);
    };
    // LATE (the 16 KiB segments, J = 4 -- the Finito batches): row t's partial dot product(s) leave through the mailbox in iteration t and
    // its total is collected in iteration t + 1, behind row t + 1's dot product and publication: the mailbox round trip -- a microsecond
    // across the chip's dies -- is off the path of an iteration.  Three register sets: row t (sum outstanding), row t + 1 (being
    // reduced), row t + 2 (in flight).  Measured (profiles/r05_long_rows_late.txt, d = 32 768 fp64): Finito batches of 256 / 4096 rows
    // 59.9 -> 52.3 / 660 -> 554 us, the J = 4 sweep 3.53 -> 4.22 TB/s.  NOT for the 32 KiB segments (J = 8): a third set of 32 KiB rows
    // leaves one workgroup per CU instead of two (326 registers; forced under 256 it spills 300 B per lane) and the sweep falls from
    // 5.4 to 3.75 TB/s -- there the total is collected in the same iteration, as before.  The mailbox has LONG_NBUF = 4 buffers: a
    // workgroup publishes row t + 4 only after it has collected row t + 2, which every sibling published after collecting row t -- so
    // nobody still polls a word when it is reused.
    unsigned int seq = 0;   // rows published so far (the next row's sequence number is seq + 1)
    auto reduce_publish = [&](RowIn &x) {
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
        const int par = (int)(seq & 1u);
        if (lane == WAVE - 1) {
            red[par][wib][0] = d1;
            if (TWO) red[par][wib][1] = d2;
        }
        __syncthreads();
        ++seq;
        if (wib == 0 && lane == 0) {
            const T p1 = (red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0]);
            unsigned long long *slot = la.box + ((((size_t)(seq & (LONG_NBUF - 1))) * C + c) * S + s) * NWORD;
            W::put(slot, p1, seq);
            if (TWO) W::put(slot + W::N, (red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1]), seq);
        }
    };
    // false: the cluster's exchange timed out.  `want` = the sequence number of the row whose total is collected (the row of x)
    auto collect_apply = [&](RowIn &x, unsigned int want) -> bool {
        if (wib == 0) {
            // lane q polls segment q's word(s); the S values are added in one fixed tree (lanes beyond S hold zeros)
            const unsigned long long *slot = la.box + (((size_t)(want & (LONG_NBUF - 1))) * C + c) * S * NWORD;
            T g1 = T(0), g2 = T(0);
            if (lane < S) {
                unsigned int pw[NWORD];
                if (!wide_get_run(slot + (size_t)lane * NWORD, want, pw)) {
#pragma unroll
                    for (int i = 0; i < NWORD; ++i) pw[i] = 0;
                    s_fail = 1;
                }
                g1 = W::decode(pw);
                if (TWO) g2 = W::decode(pw + (TWO ? W::N : 0));
            }
            g1 = wave_sum_lane63(g1);
            if (TWO) g2 = wave_sum_lane63(g2);
            if (lane == WAVE - 1) {
                tot[0] = g1;
                if (TWO) tot[1] = g2;
            }
        }
        __syncthreads();
        if (s_fail) {
            if (tid == 0) *a.errflag = 6;
            return false;
        }
        const T d1 = tot[0];
        const T d2 = TWO ? tot[1] : T(0);

        const GradCoef<T> g1 = grad_coef(a.loss, d1, x.bi, a.lam);
        if (MODE == RM_GRAD) {                        // SVRG_basic.jl:58-63, :87-92
            const T cf = g1.coef();
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += cf * x.ar[j];
            if (a.want_fval) extra += loss_value(a.loss, d1, x.bi, a.lam);
            if (a.rowdot_out && s == 0 && tid == 0) a.rowdot_out[x.row] = d1;
        } else if (MODE == RM_GRAD2) {                // Finito_LFinito.jl:93-98
            const T cf = g1.coef() - grad_coef(a.loss, d2, x.bi, a.lam).coef();
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += cf * x.ar[j];
            extra += a.hat_gamma / x.gi;
        } else if (MODE == RM_SAGA_INIT) {            // SAGA_basic.jl:42-47
#pragma unroll
            for (int j = 0; j < J; ++j) {
                V gv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) gv[v] = g1.elem(x.ar[j][v]);
                acc[j] += gv;
                if (ok[j]) TSTORE(gv, &x.sp[j * ROWS_BLOCK]);
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
                if (ok[j]) TSTORE(tv, &x.sp[j * ROWS_BLOCK]);
            }
        }
        return true;
    };

    // cluster c takes rows c, c + C, ...: n of them
    const int64_t n = a.nrows > c ? (a.nrows - c + C - 1) / C : 0;
    int64_t k = 0;
    if constexpr (LATE) {
        // iteration k: request row k + 1, reduce and publish row k, collect and apply row k - 1 (three register sets, rotating)
        RowIn r0, r1, r2;
        if (n > 0) issue(r0, c);
#define CIAO_LONG_ITER(CUR, PREV, NEXT)                              \
    {                                                                \
        if (k + 1 < n) issue(NEXT, c + (k + 1) * C);                 \
        reduce_publish(CUR);                                         \
        if (k > 0 && !collect_apply(PREV, seq - 1)) return;          \
        ++k;                                                         \
        if (k >= n) break;                                           \
    }
        int last = -1;   // which set holds the last row (its total is still outstanding)
        while (k < n) {
            last = 0;
            CIAO_LONG_ITER(r0, r2, r1)
            last = 1;
            CIAO_LONG_ITER(r1, r0, r2)
            last = 2;
            CIAO_LONG_ITER(r2, r1, r0)
        }
#undef CIAO_LONG_ITER
        if (n > 0) {
            const bool okk = last == 0 ? collect_apply(r0, seq) : (last == 1 ? collect_apply(r1, seq) : collect_apply(r2, seq));
            if (!okk) return;
        }
    } else {
        // the next row's slice is requested before this one is reduced; its total is collected at once (two register sets)
        RowIn r0, r1;
        if (n > 0) issue(r0, c);
        while (k < n) {
            if (k + 1 < n) issue(r1, c + (k + 1) * C);
            reduce_publish(r0);
            if (!collect_apply(r0, seq)) return;
            if (++k >= n) break;
            if (k + 1 < n) issue(r0, c + (k + 1) * C);
            reduce_publish(r1);
            if (!collect_apply(r1, seq)) return;
            ++k;
        }
    }

    V *pout = reinterpret_cast<V *>(ka.partial + (int64_t)c * ka.pstride) + coff;
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) PSTORE(acc[j], &pout[j * ROWS_BLOCK]);
    if (s == 0 && tid == 0) ka.pextra[c] = extra;   // extra is the same in every workgroup of the cluster
}

}  // namespace ciao
