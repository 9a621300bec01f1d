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
//        in the wave's own LDS area;
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

template <typename T>
struct MfmaOf;
template <>
struct MfmaOf<double> {
    typedef double acc __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D: col = lane & 15, row = (lane >> 4) + 4 reg
    static __device__ __forceinline__ int row(int h, int reg) { return h + 4 * reg; }
};
template <>
struct MfmaOf<float> {
    typedef float acc __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
    static __device__ __forceinline__ int row(int h, int reg) { return 4 * h + reg; }
};

constexpr int MRHS_TILE = 16;    // rows per tile = MFMA rows
constexpr int MRHS_KB = 16;      // solves per workgroup = MFMA columns

template <typename T, int SL>
constexpr size_t mrhs_lds_bytes()
{
    // per wave: its 16 x (4 SL) piece of the tile with padded rows; then the exchange area: 2 parities x 4 waves x 64 lanes x 4 values
    return (size_t)4 * MRHS_TILE * (4 * SL + 16 / sizeof(T)) * sizeof(T) + (size_t)2 * 4 * WAVE * 4 * sizeof(T);
}

template <typename T, int SL>
__global__ void __launch_bounds__(256) mrhs_kernel(MrhsArgs<T> a)
{
    using M = MfmaOf<T>;
    using Acc = typename M::acc;
    constexpr int VEC = 16 / sizeof(T);
    typedef T Vld __attribute__((ext_vector_type(VEC)));
    constexpr int DW = 4 * SL;                 // columns per wave
    constexpr int PITCH = DW + VEC;            // padded LDS row (elements)
    constexpr int NCH = DW / 16;               // 16-column chunks per wave
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
        const T *xc = a.x[kb * MRHS_KB + r];
#pragma unroll
        for (int j = 0; j < SL; j += VEC) {
            const Vld v = *reinterpret_cast<const Vld *>(xc + w * DW + h * SL + j);
#pragma unroll
            for (int e = 0; e < VEC; ++e) z[j + e] = v[e];
        }
    }
    Acc G[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) G[c] = Acc(T(0));

    // The wave's 16 x DW piece of a tile travels COALESCED: element e = (i 64 + lane) VEC of the piece linearised as [row][DW], so a
    // wave-instruction reads whole 512 B - 1 KiB runs of a row (a lane-owns-its-row-segment layout reads 64 different cache lines per
    // instruction and sixteen bytes of each: measured 5x slower, the L2 moved eight times the tile).  It is staged in registers a
    // whole tile ahead, parked in the wave's LDS area (row-major, padded rows) when the previous tile is done with it, and BOTH
    // products take their A operand from there: GEMM 1 as lane (row r, slot h) -> tile[r][h SL + j], GEMM 2 transposed.
    constexpr int NLD = MRHS_TILE * DW / (WAVE * VEC);   // 16-byte loads per lane and tile
    Vld st[NLD];
    T bq[4], bn[4];
    auto load_tile = [&](int64_t t0, T(&bo)[4]) {   // rows beyond the range read as zero; b_i of the four rows whose dots this lane will hold
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = (i * WAVE + lane) * VEC;
            const int rr = e / DW, cc = e % DW;
            const int64_t row = t0 + rr;
            const bool on = row < r_hi;
            const Vld v = __builtin_nontemporal_load(reinterpret_cast<const Vld *>(a.A + (a.row0 + (on ? row : r_lo)) * a.ld + w * DW + cc));
            st[i] = on ? v : Vld(T(0));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t rq = t0 + M::row(h, q);
            bo[q] = (a.b && rq < r_hi) ? a.b[a.row0 + rq] : T(0);
        }
    };
    auto park_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = (i * WAVE + lane) * VEC;
            *reinterpret_cast<Vld *>(tile + (e / DW) * PITCH + (e % DW)) = st[i];
        }
    };

    int par = 0;
    if (r_lo < r_hi) {
        load_tile(r_lo, bq);
        park_tile();
        if (r_lo + MRHS_TILE < r_hi) load_tile(r_lo + MRHS_TILE, bn);
    }
    for (int64_t t0 = r_lo; t0 < r_hi; t0 += MRHS_TILE) {
        // ---- GEMM 1: this wave's share of the 16 x 16 row dots (two accumulators: consecutive MFMAs do not depend on each other)
        Acc D = Acc(T(0)), D1 = Acc(T(0));
#pragma unroll
        for (int j = 0; j < SL; j += 2) {
            D = M::mma(tile[r * PITCH + h * SL + j], z[j], D);
            D1 = M::mma(tile[r * PITCH + h * SL + j + 1], z[j + 1], D1);
        }
        D += D1;
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
        // ---- the link function, in place: lane (solve c, slot h) holds rows row(h, reg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = t0 + M::row(h, q);
            C[q] = (row < r_hi) ? grad_coef(a.loss, C[q], bq[q], a.lam).coef() : T(0);
        }
        // ---- GEMM 2: G[chunk] += A'(16 columns x 4 rows) C(4 rows x 16 solves); A operand lane (column i = r, row slot h) at step t
        // (t outside, chunks inside: consecutive MFMAs go to different accumulators)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) G[c] = M::mma(tile[M::row(h, t) * PITCH + 16 * c + r], C[t], G[c]);
        }
        // ---- the next tile (in registers since the previous iteration) takes this one's place in LDS (the area is this wave's own:
        // no barrier; the exchange slots alternate, so the one barrier above is the only one per tile), and the one after is requested
        if (t0 + MRHS_TILE < r_hi) {
            park_tile();
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = bn[q];
            if (t0 + 2 * MRHS_TILE < r_hi) load_tile(t0 + 2 * MRHS_TILE, bn);
        }
    }

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
