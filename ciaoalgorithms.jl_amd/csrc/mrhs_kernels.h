// mrhs_kernels.h -- the full-gradient pass for K iterates at once:  AV_k = (1/N) sum_i grad f_i(x_k),  k < K, in ONE pass over A.
//
// Why it exists.  One solve's full pass (SVRG_basic.jl:87-92) is a GEMV-shaped contraction with a single right-hand side: no reuse
// of A, HBM-bound, MFMA does not apply (rows_kernels.h).  K independent solves over the same rows (a regularisation path,
// cross-validation folds, restarts: solvers.solve_together, ciao_ctx_chain_batch_begin) are K right-hand sides:
//      D = A Z          (N x d)(d x K): every row's K dot products          Z = [x_1 ... x_K]
//      C = link(D, b)   the scalar link function, elementwise               (lambda (a'x - b)  or  -y / (1 + e^{y a'x}))
//      G = A' C         (d x N)(N x K): the K rank-1 accumulations
// -- the one place on this path where the f_i gradients ARE a dense A.x contraction (north_star), so this kernel is matrix-core
// work: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, 4 N d K flops on 82 GB of rows read once per block of 16 solves instead
// of once per solve.  fp64 at N = 10M, d = 1024, K = 256: 1.05e13 flops (0.13 s at the 78.6 TFLOP/s fp64 peak) against 256 sweeps of
// 11.5 ms = 2.9 s.
//
// Structure (a fused pair of GEMMs that share the A tile, as an attention kernel shares K/V tiles):
//   grid = (K / 16) column blocks x P row partitions.  A workgroup = 4 waves owns 16 solves and a contiguous range of rows, and
//   walks it in tiles of 16 rows.  The d columns are split four ways over the waves: wave w owns columns [w d/4, (w+1) d/4).
//   Per tile and wave:
//     1. its 16 x d/4 piece of the tile comes from global memory with coalesced 16-byte loads, a tile ahead, and is parked row-major
//        in the wave's own LDS area, strip by strip (256 bytes of every row) behind GEMM 2 of the tile before;
//     2. GEMM 1:  D_w(16 rows x 16 solves) += a[j] (x) z[j],  j < SL = d/16: lane (row r = l & 15, k-slot h = l >> 4) reads
//        a[j] = tile[r][h SL + j] from LDS (the four k-slots of an MFMA may be ANY four columns as long as both operands agree), and
//        z[j] = x_solve[that column] sits in REGISTERS for the whole kernel (the iterates are constant over the pass);
//     3. the four partial D_w are added through LDS in wave order (one barrier per tile, slots alternate);
//     4. C = link(D, b): in the accumulator layout lane (c = l & 15, h) holds D[row(h, reg)][solve c] for reg < 4 -- and that IS the
//        B operand of GEMM 2's step t = reg (k-slot h <-> row(h, t)): no data movement between the two products;
//     5. GEMM 2:  G_w[chunk](16 columns x 16 solves) += A'(16 columns x 4 rows) C(4 rows x 16 solves), 4 steps per chunk of 16
//        columns, d/64 chunks per wave; the A operand is the tile TRANSPOSED (lane i = column, slot h = row), read back from the
//        same LDS copy of the wave's piece;
//        G_w (d/4 x 16 per wave: SL accumulators per lane) stays in registers for the whole kernel.
//   At the end every workgroup writes its d x 16 partial; mrhs_finalize_kernel adds the P partials of a column block in order and
//   scales by 1/N.  Fixed order everywhere: bitwise reproducible run to run.  All column blocks of a row partition run on the SAME
//   XCD (blockIdx -> (partition, block) map below), so the tile they all read comes from HBM once and from that XCD's L2 for the rest.
//
// Register budget per lane (fp64, d = 1024): z 64 + the staged next tile 64 + G 64 doubles = 384 VGPRs + D/C 8 + addressing: one wave per SIMD
// (__launch_bounds__(256), 512 registers with the accumulator half).  Shapes: d = 16 SL with SL in {16, 32, 64} (d = 256, 512, 1024);
// K a multiple of 16 (the host pads with copies of the last iterate).  Everything else falls back to K single sweeps.
#pragma once

#include <type_traits>

#include "ciao_common.h"

namespace ciao {

template <typename T>
struct MrhsArgs {
    const T *A;
    const T *b;
    int64_t ld, N, row0;          // rows row0 .. row0 + N of A (local rows)
    int loss;
    T lam;
    const T *const *x;            // K device pointers (K a multiple of 16: padded by the host)
    T *partial;                   // [K / 16][P][d][16]
    int P;                        // row partitions
};

constexpr int MRHS_TILE = 16;    // rows per tile = MFMA rows
constexpr int MRHS_KB = 16;      // solves per workgroup = MFMA columns

template <typename T, int SL>
constexpr size_t mrhs_lds_bytes()
{
    // per wave: its 16 x (4 SL) piece of the tile with padded rows; then the exchange area: 2 parities x 4 waves x 64 lanes x 4 values
    return (size_t)4 * MRHS_TILE * (4 * SL + 16 / sizeof(T)) * sizeof(T) + (size_t)2 * 4 * WAVE * 4 * sizeof(T);
}

template <typename T, int SL, int LOSS>
__global__ void __launch_bounds__(256) mrhs_kernel(MrhsArgs<T> a)
{
    using M = MfmaOf<T>;
    using Acc = typename M::acc;
    constexpr int VEC = 16 / sizeof(T);
    typedef T Vld __attribute__((ext_vector_type(VEC)));
    constexpr int DW = 4 * SL;                 // columns per wave
    constexpr int PITCH = DW + VEC;            // padded LDS row (elements)
    constexpr int NCH = DW / 16;               // 16-column chunks per wave
    // A STRIP is 256 bytes of every row of the wave's piece (32 fp64 / 64 fp32 columns = CPS chunks): the unit in which the next
    // tile replaces this one in LDS, behind GEMM 2's progress.  One 16-byte load per lane covers 4 rows of a strip (16 lanes a row).
    constexpr int SC = 256 / sizeof(T);        // columns per strip
    constexpr int NS = DW / SC;                // strips per wave (1 .. 8)
    constexpr int CPS = SC / 16;               // chunks per strip
    constexpr int RPL = 4;                     // rows per load instruction
    constexpr int NQ = MRHS_TILE / RPL;        // loads per strip
    static_assert(NS >= 1 && NS * SC == DW, "strip layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char mrhs_raw[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, h = lane >> 4;
    T *tile = reinterpret_cast<T *>(mrhs_raw) + (size_t)w * MRHS_TILE * PITCH;
    T *xch = reinterpret_cast<T *>(mrhs_raw) + (size_t)4 * MRHS_TILE * PITCH;     // [2][4 waves][4 regs][64 lanes]

    // blockIdx -> (row partition, column block): the column blocks of one partition are consecutive slots of ONE XCD
    // (workgroups go to XCDs round-robin by blockIdx), so they read the same tiles at about the same time through one L2
    const int nkb = gridDim.x / a.P;
    int part, kb;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        if ((a.P & 7) == 0) {
            part = xcd + 8 * (slot / nkb);
            kb = slot % nkb;
        } else {
            part = b / nkb;
            kb = b % nkb;
        }
    }
    const int64_t per = ((a.N + a.P - 1) / a.P + MRHS_TILE - 1) / MRHS_TILE * MRHS_TILE;   // rows per partition, whole tiles
    const int64_t r_lo = (int64_t)part * per;
    const int64_t r_hi = r_lo + per < a.N ? r_lo + per : a.N;

    // the iterates of this block's 16 solves, as the B operand of GEMM 1: lane (solve c = l & 15, slot h), step j: x_c[w DW + h SL + j]
    T z[SL];
    {
        // a pointer read from memory is generic to the compiler: flat loads, which count on both memory counters and may return out
        // of order -- the first MFMA of the tile loop then waited for EVERY load in flight, tile after tile; it is global memory
        const T *xc = a.x[kb * MRHS_KB + r];
        xc = (const T *)(const __attribute__((address_space(1))) T *)(uintptr_t)xc;
#pragma unroll
        for (int j = 0; j < SL; j += VEC) {
            const Vld v = *reinterpret_cast<const Vld *>(xc + w * DW + h * SL + j);
#pragma unroll
            for (int e = 0; e < VEC; ++e) z[j + e] = v[e];
        }
        // have them HERE: left to their first use the wait lands inside the tile loop, where it also drains, tile after tile, the
        // loads of the next tile that GEMM 1 has no need of   (vmcnt(0), the other counters at their maximum)
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    Acc G[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) G[c] = Acc(T(0));

    // The wave's 16 x DW piece of a tile travels COALESCED (a wave-instruction reads four 256-byte runs; a lane-owns-its-row-segment
    // layout reads 64 different cache lines per instruction and sixteen bytes of each: measured 5x slower, the L2 moved eight times
    // the tile), is staged in registers a whole tile ahead, and replaces the current tile in the wave's LDS area strip by strip as
    // GEMM 2 finishes with each strip -- so the parking stores and the loads of the tile after next are spread between the MFMAs
    // instead of standing between two tiles with the matrix pipe idle (first version of this kernel: 0.44 of the fp64 peak).
    // A partition's last tile may be short: its missing rows are read from the partition's last row (in bounds, finite wherever the
    // data is) and their link coefficient is zero, so they add nothing.
    Vld st[NS * NQ];
    const int64_t voff = (int64_t)h * a.ld + r * VEC;          // this lane's place in a 4-row x 256-byte load
    auto load_strip = [&](int s, int64_t t0) {
        if (t0 + MRHS_TILE <= r_hi) {
            const T *sb = a.A + (a.row0 + t0) * a.ld + (w * DW + s * SC);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                st[s * NQ + q] = __builtin_nontemporal_load(reinterpret_cast<const Vld *>(sb + (int64_t)(RPL * q) * a.ld + voff));
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                int64_t row = t0 + RPL * q + h;
                row = row < r_hi ? row : r_hi - 1;
                st[s * NQ + q] = __builtin_nontemporal_load(reinterpret_cast<const Vld *>(a.A + (a.row0 + row) * a.ld + (w * DW + s * SC + r * VEC)));
            }
        }
    };
    auto park_strip = [&](int s) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) *reinterpret_cast<Vld *>(tile + (RPL * q + h) * PITCH + s * SC + r * VEC) = st[s * NQ + q];
    };

    // b_i of the four rows whose dots this lane holds after GEMM 1 (rows row(h, q) of a tile), a tile ahead like the rows themselves
    auto load_b = [&](T(&bo)[4], int64_t t0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int64_t rq = t0 + M::row(h, q);
            rq = rq < r_hi ? rq : r_hi - 1;
            bo[q] = a.b ? a.b[a.row0 + rq] : T(0);
        }
    };
    constexpr int SGB_MFMA = 0x008, SGB_VMEM_RD = 0x020, SGB_DS_RD = 0x100, SGB_DS_WR = 0x200;
    int par = 0;
    T bq[4] = {T(0), T(0), T(0), T(0)}, bn[4] = {T(0), T(0), T(0), T(0)};

    // One tile.  STEADY: the tile after next exists and is whole (and b is there), so nothing in the step is conditional -- the
    // compiler's wait counts then leave the loads of the next tiles in flight across the whole step (with the conditional form it
    // drained them at the top of every tile) -- and the order of the matrix, LDS and memory instructions is pinned with
    // sched_group_barrier: LDS operand reads run a few MFMAs ahead of their use, the parking stores and the loads of the tile
    // after next are dealt one per pair of MFMAs.
    auto tile_step = [&](auto steady_tag, int64_t t0) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const bool next1 = STEADY || t0 + MRHS_TILE < r_hi, next2 = STEADY || t0 + 2 * MRHS_TILE < r_hi;
        if (STEADY) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bn[q] = a.b[a.row0 + t0 + MRHS_TILE + M::row(h, q)];
        } else if (next1) {
            load_b(bn, t0 + MRHS_TILE);
        }
        // ---- GEMM 1: this wave's share of the 16 x 16 row dots; A operand lane (row r, slot h), step j: tile[r][h SL + j], sixteen
        // bytes per LDS read (two accumulators: consecutive MFMAs do not depend on each other)
        Acc D = Acc(T(0));
        {
            const T *trow = tile + r * PITCH + h * SL;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SL; j += VEC) {
                const Vld av = *reinterpret_cast<const Vld *>(trow + j);
#pragma unroll
                for (int e = 0; e < VEC; ++e) D = M::mma(av[e], z[j + e], D);   // one accumulator: the matrix pipe forwards it
            }
            constexpr int NR = SL / VEC;       // LDS reads, VEC MFMAs each; two reads ahead
            __builtin_amdgcn_sched_group_barrier(SGB_DS_RD, 2, 0);
#pragma unroll
            for (int i = 0; i < NR - 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(SGB_MFMA, VEC, 0);
                __builtin_amdgcn_sched_group_barrier(SGB_DS_RD, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 2 * VEC, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the four partial D through LDS, added in wave order by every wave (the same sum everywhere)
        T *slot = xch + (size_t)par * 4 * 4 * WAVE;
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[(w * 4 + q) * WAVE + lane] = D[q];
        __syncthreads();
        Acc C;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            C[q] = (slot[(0 * 4 + q) * WAVE + lane] + slot[(1 * 4 + q) * WAVE + lane]) + (slot[(2 * 4 + q) * WAVE + lane] + slot[(3 * 4 + q) * WAVE + lane]);
        par ^= 1;
        // ---- the link function, in place: lane (solve c, slot h) holds rows row(h, reg); rows beyond the range get coefficient zero
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const T cf = grad_coef(LOSS, C[q], bq[q], a.lam).coef();
            C[q] = (STEADY || t0 + M::row(h, q) < r_hi) ? cf : T(0);
        }
        // ---- GEMM 2, strip by strip: G[chunk] += A'(16 columns x 4 rows) C(4 rows x 16 solves); A operand lane (column i = r, row
        // slot h) at step t (t outside, the strip's chunks inside: consecutive MFMAs go to different accumulators).  Behind each
        // strip the next tile's strip (in registers since the previous iteration) takes its place in LDS -- the area is this wave's
        // own and a wave's LDS operations execute in order: no barrier -- and the strip of the tile after next is requested.  The
        // operands of strip s + 1 are read BEFORE strip s is overwritten in program order (to the compiler the stores may alias
        // them), so they are in flight while strip s multiplies.
        __builtin_amdgcn_sched_barrier(0);
        T op[4 * CPS], on[4 * CPS];
        auto read_ops = [&](int s, T(&o)[4 * CPS]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int cc = 0; cc < CPS; ++cc) o[t * CPS + cc] = tile[M::row(h, t) * PITCH + 16 * (s * CPS + cc) + r];
        };
        read_ops(0, op);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // half-way through the strip's MFMAs the next strip's operands are requested (into the registers of the steps done)
                if (t == 2 && s + 1 < NS) read_ops(s + 1, on);
#pragma unroll
                for (int cc = 0; cc < CPS; ++cc) {
                    const int c = s * CPS + cc;
                    G[c] = M::mma(op[t * CPS + cc], C[t], G[c]);
                }
            }
            if (next1) {
                park_strip(s);
                if (next2) load_strip(s, t0 + 2 * MRHS_TILE);
            }
#pragma unroll
            for (int i = 0; i < 4 * CPS; ++i) op[i] = on[i];
        }
        if (STEADY) {
            // per strip: 4 CPS MFMAs, up to 4 CPS LDS reads of the next strip after the first half of them (the compiler pairs the
            // reads: never more), then NQ parking stores and NQ loads dealt among the MFMAs of the second half and of the next strip
            __builtin_amdgcn_sched_group_barrier(SGB_DS_RD, 4 * CPS, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // first half of the strip's MFMAs with the parking stores and loads of the strip BEFORE dealt between them, then
                // the next strip's operand reads (in program order they follow those stores, which may alias them), then the rest
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, CPS, 0);
                    if (s > 0) {
                        __builtin_amdgcn_sched_group_barrier(SGB_DS_WR, NQ / 2, 0);
                        __builtin_amdgcn_sched_group_barrier(SGB_VMEM_RD, NQ / 2, 0);
                    }
                }
                if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(SGB_DS_RD, 4 * CPS, 0);
                __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 2 * CPS, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SGB_DS_WR, NQ, 0);      // the last strip's
            __builtin_amdgcn_sched_group_barrier(SGB_VMEM_RD, NQ, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = bn[q];
    };

    if (r_lo < r_hi) {
#pragma unroll
        for (int s = 0; s < NS; ++s) load_strip(s, r_lo);
        load_b(bq, r_lo);
#pragma unroll
        for (int s = 0; s < NS; ++s) park_strip(s);
        if (r_lo + MRHS_TILE < r_hi) {
#pragma unroll
            for (int s = 0; s < NS; ++s) load_strip(s, r_lo + MRHS_TILE);
        }
    }
    int64_t t0 = r_lo;
    if (a.b)
        for (; t0 + 3 * MRHS_TILE <= r_hi; t0 += MRHS_TILE) tile_step(std::true_type{}, t0);
    for (; t0 < r_hi; t0 += MRHS_TILE) tile_step(std::false_type{}, t0);

    // ---- the block's partial: [kb][part][column][solve], lane (solve c = r, slot h) holds column 16 ch + row(h, reg) of its wave's quarter
    T *out = a.partial + ((size_t)kb * a.P + part) * (size_t)(16 * SL) * MRHS_KB;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(size_t)(w * DW + 16 * c + M::row(h, q)) * MRHS_KB + r] = G[c][q];
}

// AV_k[col] = (1/N_total) sum over the P partials of column block k / 16, in partition order
template <typename T>
__global__ void __launch_bounds__(256) mrhs_finalize_kernel(const T *__restrict__ partial, int P, int64_t d, int K, T invN, T *const *av)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (kb, column, solve-in-block)
    if (e >= (int64_t)((K + MRHS_KB - 1) / MRHS_KB) * d * MRHS_KB) return;
    const int c = (int)(e % MRHS_KB);
    const int64_t col = (e / MRHS_KB) % d;
    const int kb = (int)(e / (MRHS_KB * d));
    if (kb * MRHS_KB + c >= K) return;   // a padded solve of the last column block
    const T *p = partial + (size_t)kb * P * d * MRHS_KB + (size_t)col * MRHS_KB + c;
    T s = T(0);
    for (int q = 0; q < P; ++q) s += p[(size_t)q * d * MRHS_KB];
    av[kb * MRHS_KB + c][col] = s * invN;
}

}  // namespace ciao
