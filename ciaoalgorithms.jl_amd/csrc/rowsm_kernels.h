// rowsm_kernels.h -- full-gradient sweeps over rows of tabular size (17 .. 256 elements) with the row dots and the rank-1
// accumulation on the matrix cores (rows_smallm_kernel; VERDICT r3 item 5: "the one formulation north_star allows MFMA for").
//
// Why.  rows_small_kernel (rows_kernels.h) is instruction-bound: a dozen vector / LDS instructions per 64 elements for the
// segmented sums of several short rows sharing a wave (fp32 3.8-4.5 TB/s, fp64 5.2-5.5).  A tile of 16 consecutive rows of a
// dense matrix (ld == d) is ONE contiguous, 16-byte aligned stretch of 16 d elements; with the tile in LDS
//      D = A x          16 rows x d  times  d x 16 (the iterate in every column)       d/4 MFMAs of 16x16x4: the segmented sums for free
//      c = link(D, b)   per row                                                        (SVRG_basic.jl:58-63, :87-92)
//      acc += c_r a_r   lane (row r, slot h) keeps A[r][4 j + h] of step j in its register: one FMA per step, no second read
// costs one LDS read, one MFMA and one FMA per 64 elements; the sum over the 16 rows of a column waits for the end of the sweep
// (a butterfly over the 16 lanes of a slot).  Fifteen of the sixteen MFMA columns compute the same dot again -- the matrix pipe is
// idle otherwise: d/4 MFMAs per tile are 8 bytes per cycle and SIMD in either precision, twice the HBM rate.
//
// Structure.  A wave owns tiles g, g + #waves, ...; a ring of nb LDS buffers per wave.  The next nb - 1 tiles travel by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction straight into LDS, no registers, no LDS stores) while this one is
// multiplied; every vector-memory instruction of the loop is such a load, so the wait for the current tile is a counted vmcnt
// (a wave needs 2-3 tiles in flight: with one, 13 MB chip-wide, the sweep was latency-bound at 4.5 TB/s).  The iterate is
// operand B of GEMM 1, in registers: lane (column c, slot h) step j = x[4 j + h].
// Operand A, lane (row r, slot h) step j = tile[r][4 j + h].  Indexes beyond the row (the last steps of a row-length class) are
// clamped to the row's last element and meet a zero of the iterate / a column that is dropped.
// The matrix's last tile may hold fewer than 16 rows: it is brought in dword-wise (no 16-byte access may cross the end of A),
// missing rows are zero.  The b_i of a tile travel the same way (16 values into a 128-byte buffer beside the tile's).  Fixed order everywhere: bitwise reproducible.  Objective monitor (sum of f_i) and the cached row dots
// of the SVRG chain (rowdot_out) as in the other sweeps.
#pragma once

#include "chain_kernels.h"   // glds16s: the LDS-DMA load
#include "rows_kernels.h"

namespace ciao {

// LDS bytes of a workgroup: the iterate (zero-padded, a whole KiB), then per wave `nb` tile buffers (16 d elements each, back to
// back) and `nb` 128-byte buffers for the tiles' b_i; the per-wave column sums reuse the tile buffers at the end
template <typename T>
inline size_t smallm_x_bytes(int64_t d) { return (size_t)(((d + 3) / 4 * 4 * sizeof(T)) + 1023) / 1024 * 1024; }   // one iterate; GRAD2 keeps two
template <typename T>
inline size_t smallm_wave_bytes(int64_t d, int nb, bool table_in) { return (size_t)(nb + (table_in ? 1 : 0)) * ((size_t)16 * d * sizeof(T)) + (size_t)nb * 256 + 256; }
template <typename T>
inline size_t smallm_lds_bytes(int64_t d, int nb, bool table_in)
{
    return 2 * smallm_x_bytes<T>(d) + (size_t)ROWS_WAVES * smallm_wave_bytes<T>(d, nb, table_in);
}

// s_waitcnt vmcnt(k) for a wave-uniform k known only at run time (the instruction takes an immediate): 0 .. 63
__device__ __forceinline__ void wait_vmcnt_uniform(int k)
{
#define CIAO_W1(n) case n: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); break;
#define CIAO_W4(n) CIAO_W1(n) CIAO_W1(n + 1) CIAO_W1(n + 2) CIAO_W1(n + 3)
#define CIAO_W16(n) CIAO_W4(n) CIAO_W4(n + 4) CIAO_W4(n + 8) CIAO_W4(n + 12)
    switch (k) {
        CIAO_W16(0) CIAO_W16(16) CIAO_W16(32) CIAO_W16(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef CIAO_W16
#undef CIAO_W4
#undef CIAO_W1
}

// NC2: the row length class, d in (32 (NC2 - 1), 32 NC2]: 8 NC2 MFMA steps, all unrolled -- the steps that a shorter row of the class
// does not have multiply clamped elements by zeros of the iterate and land in columns that are dropped.  Only the last 8 steps can
// be such: the others address LDS with immediates.
// MODE (rows_kernels.h): RM_GRAD; RM_GRAD2 (LFinito's batch sweep over a row block, Finito_LFinito.jl:93-98): both dots of a row come
// out of the ONE MFMA pass -- x1 is operand B in output columns 0..7, x2 in columns 8..15 -- and the row's coefficient is
// c(a'x1) - c(a'x2); and the TABLE modes over a dense block of rows and table rows: RM_SAGA_INIT (SAGA_basic.jl:42-47), RM_FINITO_INIT
// (Finito_basic.jl:77-83), RM_FINITO_BATCH (:110-117).  There the lane that holds A[r][4 j + h] forms the new table element
// t = x - (gamma_r / N) c_r a (or c_r a) on the spot, adds its share to the aggregate, and parks t in LDS -- over the row tile itself, or,
// when the old table row is needed (FINITO_BATCH), over the table tile that came by a third LDS-DMA stream -- from where the tile
// leaves as it came: 16 bytes per lane, 1 KiB per wave-instruction.
template <typename T, int NC2, int MODE>
__global__ void __launch_bounds__(ROWS_BLOCK) rows_smallm_kernel(RowsArgs<T> a_by_value)
{
    (void)a_by_value;
    // (read in place: the tile loop is long, and holding the fields costs 15-21 spilled registers for nothing measurable --
    // profiles/r05_kernarg_ab.txt, the small-row lines)
    CIAO_KERNARG0(RowsArgs<T>, a);
    constexpr bool TWO = (MODE == RM_GRAD2);
    constexpr bool TABLE = (MODE == RM_SAGA_INIT || MODE == RM_FINITO_INIT || MODE == RM_FINITO_BATCH);
    constexpr bool SIN = (MODE == RM_FINITO_BATCH);                       // the old table tile comes in
    constexpr bool GAM = TWO || MODE == RM_FINITO_INIT || MODE == RM_FINITO_BATCH;   // per-row stepsizes are used
    static_assert(MODE == RM_GRAD || TWO || TABLE, "mode");
    constexpr int SLMAX = 8 * NC2, SAFE = 8 * (NC2 - 1);
    using M = MfmaOf<T>;
    using Acc = typename M::acc;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smm_raw[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, h = lane >> 4;
    const int d = (int)a.d;
    const int xp = (d + 3) & ~3;
    const uint32_t tb = 16u * (uint32_t)d * (uint32_t)sizeof(T);   // bytes of a whole tile = the buffer pitch
    const uint32_t pitch = tb;
    const int nb = a.small_nb;                                      // tile buffers per wave: nb - 1 tiles in flight behind the current one
    const uint32_t xbytes = 2u * (((uint32_t)(xp * sizeof(T)) + 1023u) & ~1023u);
    T *xl = reinterpret_cast<T *>(smm_raw);
    T *xl2 = reinterpret_cast<T *>(smm_raw + xbytes / 2);
    unsigned char *wbuf = smm_raw + xbytes + (size_t)wib * ((size_t)(nb + (SIN ? 1 : 0)) * pitch + (size_t)nb * 256 + 256);
    const uint32_t wbuf_off = (uint32_t)(uintptr_t)wbuf;
    const T *bl = reinterpret_cast<const T *>(wbuf + (size_t)nb * pitch);              // [nb][16]  b_i
    const T *gl = reinterpret_cast<const T *>(wbuf + (size_t)nb * (pitch + 128));      // [nb][16]  gamma_i (TWO)
    T *dl = reinterpret_cast<T *>(wbuf + (size_t)nb * (pitch + 256));                  // [2][16]: a tile's row dots, from the accumulator layout to the lanes of each row
    unsigned char *stb = wbuf + (size_t)nb * (pitch + 256) + 256;                      // SIN: the table tile (old rows in, new rows out)

    for (int c = threadIdx.x; c < xp; c += ROWS_BLOCK) {
        xl[c] = c < d ? a.x1[c] : T(0);
        if (TWO) xl2[c] = c < d ? a.x2[c] : T(0);
    }
    __syncthreads();
    // operand B of GEMM 1 stays in registers for the whole sweep: lane (column c, slot h), step j: x[4 j + h], zero beyond the row
    T xr[SLMAX];
    {
        const T *xs = (TWO && r >= 8) ? xl2 : xl;
#pragma unroll
        for (int j = 0; j < SLMAX; ++j) xr[j] = 4 * j + h < xp ? xs[4 * j + h] : T(0);
    }

    const int64_t ntiles = (a.nrows + 15) >> 4;
    const int64_t nwaves = (int64_t)gridDim.x * ROWS_WAVES;
    const T s2 = (a.loss == CIAO_LOSS_LS) ? a.lam : (a.loss == CIAO_LOSS_LOGISTIC ? T(1) : T(0));

    // tile g -> LDS buffer `buf`, with the b_i of its rows.  Everything by LDS-DMA: a load the compiler counts would have it wait,
    // before the value's first use, with a vmcnt that knows nothing of the DMA in flight behind it -- i.e. for the NEXT tile too
    // (the first version of this kernel read b_i through volatile loads: every tile waited out the latency of the one after)
    auto fetch = [&](int64_t g, int buf) {
        const int64_t row_b = g << 4;
        const int64_t left = a.nrows - row_b;
        const int nr = left < 16 ? (int)left : 16;
        const T *gp = a.A + (a.row0 + row_b) * (int64_t)d;
        const uint32_t dst = wbuf_off + (uint32_t)buf * pitch;
        if (nr == 16) {
            for (uint32_t off = 0; off < tb; off += 1024u) {
                const uint32_t vo = off + (uint32_t)lane * 16u;
                if (vo < tb) glds16s(gp, vo, dst + off);
            }
        } else {
            // the matrix's last, short tile: dword by dword (no 16-byte access may cross the end of A); the rows beyond it are zeroed
            // (their coefficient is zero, but whatever the buffer held -- possibly nothing yet -- times zero need not be)
            const uint32_t have = (uint32_t)nr * (uint32_t)d * (uint32_t)sizeof(T);
            for (uint32_t off = 0; off < have; off += 256u) {
                const uint32_t vo = off + (uint32_t)lane * 4u;
                if (vo < have) glds4(reinterpret_cast<const unsigned char *>(gp) + vo, dst + off);
            }
            uint32_t *zb = reinterpret_cast<uint32_t *>(wbuf + (size_t)buf * pitch);
            for (uint32_t e = have / 4u + (uint32_t)lane; e < tb / 4u; e += WAVE) zb[e] = 0u;
        }
        if (a.b) {
            const uint32_t vo = (uint32_t)lane * 4u;
            if (vo < (uint32_t)nr * (uint32_t)sizeof(T))
                glds4(reinterpret_cast<const unsigned char *>(a.b + a.row0 + row_b) + vo, wbuf_off + (uint32_t)nb * pitch + (uint32_t)buf * 128u);
        }
        if (GAM && a.gam) {
            const uint32_t vo = (uint32_t)lane * 4u;
            if (vo < (uint32_t)nr * (uint32_t)sizeof(T))
                glds4(reinterpret_cast<const unsigned char *>(a.gam + a.row0 + row_b) + vo, wbuf_off + (uint32_t)nb * (pitch + 128u) + (uint32_t)buf * 128u);
        }
    };

    // FINITO_BATCH: the table rows of tile g -> the wave's table buffer (the same two forms; the short last tile zero-filled)
    auto fetch_table = [&](int64_t g) {
        const int64_t row_b = g << 4;
        const int64_t left = a.nrows - row_b;
        const int nr = left < 16 ? (int)left : 16;
        const T *gp = a.table + (a.row0 + row_b) * (int64_t)d;
        const uint32_t dst = (uint32_t)(uintptr_t)stb;
        if (nr == 16) {
            for (uint32_t off = 0; off < tb; off += 1024u) {
                const uint32_t vo = off + (uint32_t)lane * 16u;
                if (vo < tb) glds16s(gp, vo, dst + off);
            }
        } else {
            const uint32_t have = (uint32_t)nr * (uint32_t)d * (uint32_t)sizeof(T);
            for (uint32_t off = 0; off < have; off += 256u) {
                const uint32_t vo = off + (uint32_t)lane * 4u;
                if (vo < have) glds4(reinterpret_cast<const unsigned char *>(gp) + vo, dst + off);
            }
            uint32_t *zb = reinterpret_cast<uint32_t *>(stb);
            for (uint32_t e = have / 4u + (uint32_t)lane; e < tb / 4u; e += WAVE) zb[e] = 0u;
        }
    };
    // the new table rows of tile g leave LDS (from `src`) as the tile came: 16 bytes per lane; the short last tile dword by dword
    auto store_table = [&](int64_t g, const unsigned char *src) {
        typedef uint32_t V4 __attribute__((ext_vector_type(4)));
        const int64_t row_b = g << 4;
        const int64_t left = a.nrows - row_b;
        const int nr = left < 16 ? (int)left : 16;
        unsigned char *gp = reinterpret_cast<unsigned char *>(a.table + (a.row0 + row_b) * (int64_t)d);
        if (nr == 16) {
            for (uint32_t off = 0; off < tb; off += 1024u) {
                const uint32_t vo = off + (uint32_t)lane * 16u;
                if (vo < tb) __builtin_nontemporal_store(*reinterpret_cast<const V4 *>(src + vo), reinterpret_cast<V4 *>(gp + vo));
            }
        } else {
            const uint32_t have = (uint32_t)nr * (uint32_t)d * (uint32_t)sizeof(T);
            for (uint32_t vo = (uint32_t)lane * 4u; vo < have; vo += 256u)
                *reinterpret_cast<uint32_t *>(gp + vo) = *reinterpret_cast<const uint32_t *>(src + vo);
        }
    };

    T acc[SLMAX];
#pragma unroll
    for (int j = 0; j < SLMAX; ++j) acc[j] = T(0);
    T ex = T(0);
    const bool extras = TWO || a.want_fval || a.rowdot_out != nullptr;
    // LDS-DMA instructions of one whole tile (what stays in flight behind the tile being waited for is a multiple of it)
    const int per_tile = (int)((tb + 1023u) >> 10) + (a.b ? 1 : 0) + (GAM && a.gam ? 1 : 0);
    int64_t g = (int64_t)blockIdx.x * ROWS_WAVES + wib;
    int buf = 0;
    for (int k = 0; k < nb - 1; ++k)
        if (g + k * nwaves < ntiles) fetch(g + k * nwaves, k);
    if (SIN && g < ntiles) fetch_table(g);
    const int n_kib = (int)((tb + 1023u) >> 10);
    bool first_tile = true;
    for (; g < ntiles; g += nwaves) {
        const int64_t row_b = g << 4;
        const int64_t left = a.nrows - row_b;
        const int nr = left < 16 ? (int)left : 16;
        // this tile has landed when no more than the nb - 2 tiles requested after it are outstanding -- counted only while all of
        // those exist and are whole tiles (the matrix's short last tile issues another number of instructions): else everything
        const int64_t g_last = g + (int64_t)(nb - 2) * nwaves;      // the youngest tile in flight
        if (!TABLE && nb > 2 && ((g_last + 1) << 4) <= a.nrows)
            wait_vmcnt_uniform((nb - 2) * per_tile);
        else if (SIN && nb == 2 && nr == 16)
            // FINITO_BATCH: behind this tile's rows, in issue order, are the previous tile's table stores (none before the first tile)
            // and this tile's table rows -- one instruction per KiB each, this tile being whole; only the rows are needed for the dots
            wait_vmcnt_uniform((first_tile ? 1 : 2) * n_kib);
        else
            wait_vmcnt_uniform(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) (the short last tile's zero fill), vmcnt / expcnt at their maximum
        asm volatile("" ::: "memory");
        {
            const int64_t gn = g + (int64_t)(nb - 1) * nwaves;       // into the buffer the previous tile has just left
            int bn = buf + nb - 1;
            bn = bn >= nb ? bn - nb : bn;
            if (gn < ntiles) fetch(gn, bn);
        }
        const T *tl = reinterpret_cast<const T *>(wbuf + (size_t)buf * pitch);
        // ---- the 16 row dots on the matrix cores (every column of D the same), two accumulators in turn.  The A operand of step j --
        // lane (row r, slot h): tile[r][4 j + h] -- STAYS in its register: it is this lane's share of the rank-1 accumulation below
        Acc D = Acc(T(0)), D1 = Acc(T(0));
        T av[SLMAX];
        {
            const T *ta = tl + r * d + h;
            const int last = d - 1 - h;                     // (index of the row's last element relative to ta)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SLMAX; ++j) {
                av[j] = j < SAFE ? ta[4 * j] : ta[4 * j < last ? 4 * j : last];
                if (j & 1)
                    D1 = M::mma(av[j], xr[j], D1);
                else
                    D = M::mma(av[j], xr[j], D);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int j = 0; j < SLMAX - 4; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        D += D1;
        // ---- row r's dot to the lanes of row r: the accumulator holds rows row(h, q) in lane (c, h); through the wave's LDS words
        if (r == 0 || (TWO && r == 8)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dl[(r >> 3) * 16 + M::row(h, q)] = D[q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const T dot = dl[r];
        const T bi = (a.b && r < nr) ? bl[buf * (128 / (int)sizeof(T)) + r] : T(0);
        // ---- the link function, once per lane: its row's coefficient; rows beyond the matrix get zero
        T coef, coef1;
        if (a.loss == CIAO_LOSS_LS)
            coef1 = grad_coef(CIAO_LOSS_LS, dot, bi, a.lam).s1;
        else if (a.loss == CIAO_LOSS_LOGISTIC)
            coef1 = grad_coef(CIAO_LOSS_LOGISTIC, dot, bi, a.lam).s1;
        else
            coef1 = T(0);
        coef = coef1 * s2;
        if (TWO) {
            const T dot2 = dl[16 + r];
            T c2;
            if (a.loss == CIAO_LOSS_LS)
                c2 = grad_coef(CIAO_LOSS_LS, dot2, bi, a.lam).s1 * s2;
            else if (a.loss == CIAO_LOSS_LOGISTIC)
                c2 = grad_coef(CIAO_LOSS_LOGISTIC, dot2, bi, a.lam).s1 * s2;
            else
                c2 = T(0);
            coef = coef - c2;
        }
        coef = r < nr ? coef : T(0);
        if (extras && h == 0 && r < nr) {
            if (TWO) {
                const T gi = a.gam ? gl[buf * (128 / (int)sizeof(T)) + r] : a.gam_uniform;
                ex += a.hat_gamma / gi;
            } else {
                if (a.want_fval) ex += loss_value(a.loss, dot, bi, a.lam);
                if (a.rowdot_out) a.rowdot_out[a.row0 + row_b + r] = dot;
            }
        }
        if (!TABLE) {
            // ---- the rank-1 accumulation: acc[j] += c_r tile[r][4 j + h], from the registers GEMM 1 read (the first form of this kernel
            // ran it as a second MFMA product: fifteen sixteenths of twice the matrix work for nothing, the matrix pipe 63 % busy and
            // the bound).  The sum over the 16 rows of a column is taken once, at the end of the sweep.
#pragma unroll
            for (int j = 0; j < SLMAX; ++j) acc[j] = __builtin_fma(coef, av[j], acc[j]);
        } else {
            // ---- the new table element of (row r, column 4 j + h), its share of the aggregate, and the element parked in LDS.
            // s1 / s2 as GradCoef::elem applies them: grad f_i = (a s1) s2
            const T s1 = r < nr ? coef1 : T(0);
            const T gi = (GAM && a.gam && r < nr) ? gl[buf * (128 / (int)sizeof(T)) + r] : a.gam_uniform;
            const T cg = gi * a.invN;
            const T rr = (MODE == RM_FINITO_INIT) ? T(1) / gi : a.hat_gamma / gi;
            if (SIN) {
                // the old table rows: everything up to them has landed when only the NEXT tile's rows (requested at the top of this
                // iteration: per_tile instructions if that tile is whole) are still outstanding
                const int64_t gn = g + nwaves;
                wait_vmcnt_uniform((nb == 2 && gn < ntiles && ((gn + 1) << 4) <= a.nrows) ? per_tile : 0);
                asm volatile("" ::: "memory");
            }
            T *tout = SIN ? reinterpret_cast<T *>(stb) : const_cast<T *>(tl);
            T *to = tout + r * d + h;
            const bool rowlive = r < nr;
#pragma unroll
            for (int j = 0; j < SLMAX; ++j) {
                const bool live = rowlive && (j < SAFE || 4 * j + h < d);
                const T gv = (av[j] * s1) * s2;
                if (MODE == RM_SAGA_INIT) {
                    acc[j] += gv;
                    if (live) to[4 * j] = gv;
                } else {
                    const T tv = xr[j] - cg * gv;
                    if (MODE == RM_FINITO_INIT) {
                        acc[j] += rowlive ? tv * rr : T(0);
                    } else {
                        const T sv = to[j < SAFE ? 4 * j : (4 * j + h < d ? 4 * j : d - 1 - h)];
                        acc[j] += rowlive ? (tv - sv) * rr : T(0);
                    }
                    if (live) to[4 * j] = tv;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            store_table(g, reinterpret_cast<const unsigned char *>(tout));
            if (SIN && g + nwaves < ntiles) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile has left the buffer
                fetch_table(g + nwaves);
            }
        }
        buf = buf + 1 == nb ? 0 : buf + 1;
        first_tile = false;
    }

    // ---- per-wave column sums: column 4 j + h is spread over the 16 lanes (rows) of slot h -> summed in a fixed butterfly, written by
    // the lane of row 0 -> the block's partial
    wait_vmcnt_uniform(0);
    __syncthreads();   // every wave is done with its buffers and with the iterate
    T *cs = reinterpret_cast<T *>(smm_raw + xbytes) + (size_t)wib * (4 * SLMAX);   // [wave][column] (over the tile buffers)
#pragma unroll
    for (int j = 0; j < SLMAX; ++j) {
        T v = acc[j];
        v += __shfl_xor(v, 1, WAVE);
        v += __shfl_xor(v, 2, WAVE);
        v += __shfl_xor(v, 4, WAVE);
        v += __shfl_xor(v, 8, WAVE);
        if (r == 0 && 4 * j + h < d) cs[4 * j + h] = v;
    }
    ex = wave_allsum(ex);
    __shared__ T red_extra_smallm[ROWS_WAVES];
    if (lane == 0) red_extra_smallm[wib] = ex;
    __syncthreads();
    const T *cs0 = reinterpret_cast<const T *>(smm_raw + xbytes);
    T *pout = a.partial + (int64_t)blockIdx.x * a.pstride;
    for (int c = threadIdx.x; c < d; c += ROWS_BLOCK) {
        T sacc = cs0[c];
        for (int w = 1; w < ROWS_WAVES; ++w) sacc += cs0[w * (4 * SLMAX) + c];
        pout[c] = sacc;
    }
    if (threadIdx.x == 0) {
        T e2 = T(0);
        for (int w = 0; w < ROWS_WAVES; ++w) e2 += red_extra_smallm[w];
        a.pextra[blockIdx.x] = e2;
    }
}

}  // namespace ciao
