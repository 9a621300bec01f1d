// rowsw_kernels.h -- batch steps over rows of tabular size (d <= 256) given as an INDEX LIST, one WAVE per row (rows_wrow_kernel; round 5).
//
// Why.  The reference's default Finito draws every batch as a list of random rows (sweeping = 1, Finito.jl:46:
// sample(1:N, r, replace=false), Finito_basic.jl:97).  On rows of a few hundred bytes such a batch is r random DRAM pages for the
// data rows, r for the table rows, r cache lines each for b_i and gamma_i: what it costs is set by how many of those requests the
// chip has in flight, not by instructions.  tools/micro/gather_ceiling.hip (profiles/r05_small_gather_tried.txt): one wave per row,
// everything of a row requested at once, as many waves as the chip holds moves a batch of 65 536 rows of 50 Float64 (read row, read and
// write table row) in 25.6 us; the several-rows-per-wave kernel (rows_smallb_kernel: a dozen LDS operations per element, 4-8 waves
// per CU) needs 46 us for the same batch, and the matrix-core tiles gathered row by row (tried, removed) 45 us: both keep too
// few rows in flight.  So: the micro-benchmark's shape with the arithmetic in it.
//
// A wave owns row q, q + #waves, ... of the list.  Lane l holds chunk l (+ 64 k) of the row, of the table row, of the iterate and of
// the wave's accumulator -- a chunk is 16 bytes where rows are whole 16-byte chunks at aligned addresses, one element otherwise
// (d = 50 fp32: 200-byte rows); lanes beyond the row are idle, which costs nothing here.  The list entry, b_i and gamma_i are SCALAR
// loads (the row is wave-uniform): no vector register, no vmcnt.  Two rows are in flight per wave (ping-pong register sets), eight
// waves per SIMD: 16 rows per SIMD, 16 K rows chip-wide.  Dot product: masked partial sums, one DPP wave sum.  The wave's accumulator is
// combined across the block's waves through LDS in wave order; a block writes one partial d-vector (fixed order: bitwise reproducible).
// Modes: RM_FINITO_BATCH (Finito_basic.jl:110-117) and RM_GRAD2 (LFinito's batch sweep, Finito_LFinito.jl:93-98: two dot products per row, no
// table).  Row blocks (idx == nullptr) run too -- the shapes the matrix-core kernel does not hold.
#pragma once

#include "rows_kernels.h"

namespace ciao {

// registers -> waves per SIMD asked of the compiler: iterate, accumulator and two (row, table row) sets are 6 K chunks of VEC elements,
// plus addressing / reduction temporaries (the fp64 wave sum alone takes a dozen)
template <int ES, int VEC, int K>
struct WrowWaves {
    static constexpr int budget = 6 * K * (VEC * ES / 4) + (ES == 8 ? 56 : 40);
    static constexpr int raw = 512 / ((budget + 7) / 8 * 8);
    static constexpr int value = raw < 1 ? 1 : (raw > 8 ? 8 : raw);
    // waves of a workgroup: TWO workgroups fill a CU's four SIMDs at `value` waves each -- at most 2 x 256 partial d-vectors for
    // finalize to add, whose time is set by their number (with 256-thread workgroups, 6-8 per CU, 1536-2048 partials: finalize 13 us
    // of a 42 us batch of 65 536 rows of 50 Float64; now 512: profiles/r05_wrow_blocks.txt)
    static constexpr int block_waves = 2 * value;
};

template <typename T, int VEC, int K, int MODE>
__global__ void __launch_bounds__((WAVE * WrowWaves<sizeof(T), VEC, K>::block_waves), (WrowWaves<sizeof(T), VEC, K>::value))
    rows_wrow_kernel(RowsArgs<T> a_by_value)
{
    constexpr int NWAVES = WrowWaves<sizeof(T), VEC, K>::block_waves;   // waves of this workgroup
    constexpr int BLOCK = NWAVES * WAVE;
    (void)a_by_value;
    CIAO_KERNARG0(RowsArgs<T>, ka);
    // what the row loop reads: loaded once, held in scalar registers (read in place hipcc re-loaded a.idx, a.N, a.A, a.ld ... in front of
    // every row's requests -- three scalar-cache round trips on the path list entry -> row address -> loads)
    const struct {
        const T *A, *b, *gam;
        T *table;
        const int64_t *idx;
        int64_t ld, d, N, row0, nrows;
        int *errflag;
        T gam_uniform, invN, hat_gamma, lam;
        int loss;
    } a = {sgpr_pin_global(ka.A), sgpr_pin_global(ka.b), sgpr_pin_global(ka.gam), sgpr_pin_global(ka.table), sgpr_pin_global(ka.idx),
           sgpr_pin(ka.ld), sgpr_pin(ka.d), sgpr_pin(ka.N), sgpr_pin(ka.row0), sgpr_pin(ka.nrows), sgpr_pin_global(ka.errflag),
           sgpr_pin(ka.gam_uniform), sgpr_pin(ka.invN), sgpr_pin(ka.hat_gamma), sgpr_pin(ka.lam), sgpr_pin(ka.loss)};
    static_assert(MODE == RM_FINITO_BATCH || MODE == RM_GRAD2, "the batch modes");
    constexpr bool TWO = (MODE == RM_GRAD2);
    using V = typename ChunkOf<T, VEC>::type;
    constexpr int D = K * WAVE * VEC;   // elements a wave's lanes reach

    __shared__ __attribute__((aligned(16))) T red[NWAVES][D];   // (at most 16 waves x 256 elements x 8 bytes = 32 KiB)
    __shared__ T red_extra[NWAVES];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * NWAVES;
    const int64_t d = a.d;
    const int64_t nchunks = d / VEC;

    bool ok[K];
    V xv[K], x2v[K], acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ok[k] = (k * WAVE + lane) < nchunks;
        xv[k] = ok[k] ? reinterpret_cast<const V *>(ka.x1)[k * WAVE + lane] : V(T(0));
        x2v[k] = (TWO && ok[k]) ? reinterpret_cast<const V *>(ka.x2)[k * WAVE + lane] : V(T(0));
        acc[k] = V(T(0));
    }
    T extra = T(0);
    // read-only for the kernel's lifetime and addressed wave-uniformly: constant address space, i.e. scalar loads
    typedef const __attribute__((address_space(4))) int64_t *cidx_t;
    typedef const __attribute__((address_space(4))) T *cval_t;
    const cidx_t idxc = (cidx_t)(uintptr_t)a.idx;
    const cval_t bc = (cval_t)(uintptr_t)a.b;
    const cval_t gc = (cval_t)(uintptr_t)a.gam;
    const T gam_u = a.gam_uniform, invN = a.invN, hat_gamma = a.hat_gamma, lam = a.lam;
    const int loss = a.loss;

    struct RowIn {
        V ar[K], sr[K];
        int64_t row;
        T bi, gi;
    };
    // everything row q needs, requested at once
    auto issue = [&](RowIn &x, int64_t q) __attribute__((always_inline)) {
        int64_t row = idxc ? idxc[q] : a.row0 + q;
        if (idxc && (uint64_t)row >= (uint64_t)a.N) {   // memory-safe: flag it, take row 0 (results are void once flagged)
            if (lane == 0) *a.errflag = 1;
            row = 0;
        }
        x.row = row;
        const V *ap = reinterpret_cast<const V *>(a.A + row * a.ld);
        const V *sp = TWO ? nullptr : reinterpret_cast<const V *>(a.table + row * d);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            x.ar[k] = ok[k] ? __builtin_nontemporal_load(&ap[k * WAVE + lane]) : V(T(0));
            if (!TWO) x.sr[k] = ok[k] ? __builtin_nontemporal_load(&sp[k * WAVE + lane]) : V(T(0));
        }
        x.bi = bc ? bc[row] : T(0);
        x.gi = gc ? gc[row] : gam_u;
    };
    auto process = [&](RowIn &x) __attribute__((always_inline)) {
        T d1 = T(0);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) d1 += x.ar[k][v] * xv[k][v];
        d1 = wave_allsum(d1);
        const GradCoef<T> g1 = grad_coef(loss, d1, x.bi, lam);
        if constexpr (TWO) {                 // Finito_LFinito.jl:93-98: acc += (c(a'z_full) - c(a'z)) a;  extra += hat_gamma / gamma_i
            T d2 = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int v = 0; v < VEC; ++v) d2 += x.ar[k][v] * x2v[k][v];
            d2 = wave_allsum(d2);
            const T c = g1.coef() - grad_coef(loss, d2, x.bi, lam).coef();
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += c * x.ar[k];
            extra += hat_gamma / x.gi;
            return;
        }
        const T cg = x.gi * invN;            // gamma_i / N
        const T rr = hat_gamma / x.gi;
        V *sp = reinterpret_cast<V *>(a.table + x.row * d);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            V tv;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                tv[v] = xv[k][v] - cg * g1.elem(x.ar[k][v]);          // t = z - (gamma_i / N) grad f_i(z)      Finito_basic.jl:112-114
                acc[k][v] += (tv[v] - x.sr[k][v]) * rr;                // av += (t - s_i) hat_gamma / gamma_i   :115
            }
            if (ok[k]) __builtin_nontemporal_store(tv, &sp[k * WAVE + lane]);   // s_i = t                      :116
        }
    };

    int64_t q = (int64_t)blockIdx.x * NWAVES + wib;
    if (q < a.nrows) {
        RowIn A0, B0;
        issue(A0, q);
        while (true) {
            int64_t qn = q + nwaves;
            bool more = qn < a.nrows;
            if (more) issue(B0, qn);
            process(A0);
            if (!more) break;
            q = qn;
            qn = q + nwaves;
            more = qn < a.nrows;
            if (more) issue(A0, qn);
            process(B0);
            if (!more) break;
            q = qn;
        }
    }

    // every wave's accumulator into its own LDS row, then element e summed over the waves in wave order: one partial per block
#pragma unroll
    for (int k = 0; k < K; ++k) reinterpret_cast<V *>(red[wib])[k * WAVE + lane] = acc[k];
    if (lane == 0) red_extra[wib] = extra;   // (wave-uniform)
    __syncthreads();
    T *pout = ka.partial + (int64_t)blockIdx.x * ka.pstride;
    for (int e = threadIdx.x; e < d; e += BLOCK) {
        T t = red[0][e];
#pragma unroll
        for (int w = 1; w < NWAVES; ++w) t += red[w][e];
        pout[e] = t;
    }
    if (threadIdx.x == 0) {
        T ex = T(0);
        for (int w = 0; w < NWAVES; ++w) ex += red_extra[w];
        ka.pextra[blockIdx.x] = ex;
    }
}

}  // namespace ciao
