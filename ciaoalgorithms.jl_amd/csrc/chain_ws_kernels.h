// chain_ws_kernels.h -- the sequential inner loops as a WAVE-SPECIALISED workgroup.
//
// chain_dma_kernel (chain_dma_kernels.h) runs a step on four waves that each do everything: address arithmetic and LDS-DMA issue
// for the rows DEPTH steps ahead, the arithmetic of the step, and a cross-wave exchange of the four partial dot products made
// of two serialized LDS round trips around an s_barrier (write, lgkmcnt(0), barrier, read).  One wave per SIMD executes that
// as one dependent instruction stream, so every instruction that is not arithmetic is time (profiles/r02_chain_pmc_instructions).
// Here the roles are separate waves of one workgroup:
//
//   waves 0..3  CONSUMERS  the dependent chain and nothing else.  State in registers exactly as in chain_dma_kernel (thread t
//                          owns the 16-byte chunks t + 256 j), rows read from the LDS ring, per-step scalars from an LDS record;
//                          no vector-memory instruction in a table-free step, only the table-row stores in a SAGA step.
//   wave  4     STAGER     index -> row address, b_i, per-sample scalar (SVRG with cached row dots: the link coefficient at
//                          a_i'z_full), table-row address, table-row hazard flag; 64 steps at a time into an LDS record ring,
//                          hundreds of steps ahead, with ordinary loads (its latency is nobody's).
//   waves 5..   ISSUERS    the LDS-DMA (global_load_lds_dwordx4) of every row and table row into the LDS rings, R steps ahead
//                          of the consumers, with hand-counted vmcnt waits; they publish how many steps have LANDED.
//
// The cross-wave exchange of the partial dot products has no barrier (an s_barrier would involve the producer waves): lane
// 63 of each consumer wave writes its partial and then adds 1 to a counter of the step's parity (LDS operations of one wave
// are performed in order); every consumer reads counter and partials in ONE batch of LDS reads, counter first, and repeats
// until the counter shows all four arrivals -- values read after a counter that shows four arrivals are those four waves'
// values.  One LDS round trip instead of two plus a barrier.  Two parities suffice: a wave can only write its partial of
// step s+2 after it has read every partial of step s+1, which every wave wrote after reading those of step s.
//
// Flow control, all through monotonic LDS words: the consumers' two parity counters give the number of COMPLETE steps (all
// four partials written; steps complete in order), which is what stager and issuers wait on before they overwrite a record or
// a ring slot; `staged` (stager -> issuers), `landed[q]` (issuers -> consumers, checked twice per ring revolution).
// Every spin waits on a wave of the same workgroup that never waits on the spinner's own progress beyond what it has already
// been given: a blocked issuer first drains its queue and publishes everything it has issued.
//
// SAGA table rows: the consumer that owns a chunk stores it; the issuers fetch table rows like data rows.  A consumer's
// `s_waitcnt vmcnt(K*J)` after its stores means its stores of K steps ago have been written (stores are its only vector-memory
// operations, so the counter is theirs), hence "step s complete" implies every table store of steps <= s-2-K is in memory, and
// a table row DMA issued when C steps were complete is fresh unless the row was a sample of steps [C-1-K, t).  The stager
// flags every step whose row occurs among the previous R+K+3 steps (a superset: +1 for the deferred store, +2 of margin); a flagged step ignores its ring slot and
// re-reads the row: each thread reads back the bytes it stored itself, in program order (as chain_dma_kernel does).
// The arithmetic of a step is chain_dma_kernel's, operation for operation: results are bitwise identical (tests; tools/ws_soak.py:
// 2 x 10^7 steps, the whole 8 GB table).  What a step leaves behind that the next dot product does not need -- the table row's
// store, SVRG's z += w -- is issued in the NEXT step's exchange, between the poll's issue and its wait.
// Dispatch (chain_launch.inc): SAGA / SAG on rows of 2-8 KiB.  The kernel also instantiates for SVRG (it was measured: 10 % slower
// than chain_dma_kernel there, profiles/r03_chain_step_instruction_classes.md), which is why those units are not built.
#pragma once

#include "chain_kernels.h"

namespace ciao {

constexpr int WS_NCW = 4;          // consumer waves
constexpr int WS_RR = 256;         // record ring entries (steps staged ahead of the consumers: up to WS_RR - 64)
constexpr int WS_STORE_LAG = 6;    // K: a consumer's table stores are known to be written K steps after issue

template <typename T>
struct WsRec {     // per-step scalars, read by every consumer lane from one address
    T bi, gi;
    T *tptr;       // table row of the step's sample (SAGA)
    int stale;     // the table row may have been rewritten after its DMA was issued: re-read it
    int pad;
};
struct WsIss {     // per-step addresses for the issuers
    const unsigned char *aptr;
    const unsigned char *tptr;
};
struct WsCtl {
    unsigned int cnt[2][16];     // parity counters of the exchange (64 bytes apart)
    unsigned int staged[16];
    unsigned int landed[2][16];
};

template <int J, bool TABLE>
struct WsRing {   // ring slots: a power of two, 128 KiB of LDS for the ring(s) at most, 16 slots at most
    static constexpr int bytes = J * 4096 * (TABLE ? 2 : 1);
    static constexpr int value = (128 * 1024 / bytes) >= 16 ? 16 : ((128 * 1024 / bytes) >= 8 ? 8 : ((128 * 1024 / bytes) >= 4 ? 4 : 2));
};

template <typename T, int J, int ALG>
struct WsLayout {
    static constexpr bool TABLE = (ALG == CA_SAGA);
    static constexpr bool TWO = (ALG == CA_SVRG);
    static constexpr int R = WsRing<J, TABLE>::value;
    static constexpr int ROW_BYTES = J * 4096;
    static constexpr size_t ringA = 0;
    static constexpr size_t ringT = ringA + (size_t)R * ROW_BYTES;
    static constexpr size_t rec = ringT + (TABLE ? (size_t)R * ROW_BYTES : 0);
    static constexpr size_t iss = rec + WS_RR * sizeof(WsRec<T>);
    static constexpr size_t rows = iss + WS_RR * sizeof(WsIss);
    static constexpr size_t red = rows + WS_RR * sizeof(int64_t);          // T red[2][4][2], 128-byte slots per parity
    static constexpr size_t ctl = red + 2 * 128;
    static constexpr size_t shards = ctl + sizeof(WsCtl);                  // the shard table of a row-sharded problem (stager only)
    static constexpr size_t total = shards + SHARD_QW * sizeof(int64_t);
};

// control words are read and written through LDS-address-space pointers (a volatile access through a generic pointer becomes
// a FLAT instruction with a full vmcnt/lgkmcnt drain); the result is made wave-uniform so that every spin is a scalar branch
typedef __attribute__((address_space(3))) unsigned int ws_lds_u32;
__device__ __forceinline__ unsigned int lds_peek(const unsigned int *p)
{
    const unsigned int v = *reinterpret_cast<const volatile ws_lds_u32 *>((uint32_t)(uintptr_t)p);
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void lds_poke(unsigned int *p, unsigned int v) { *reinterpret_cast<volatile ws_lds_u32 *>((uint32_t)(uintptr_t)p) = v; }

// Every spin is bounded: a wave that has retried WS_SPIN_LIMIT times (tenths of a second; a legitimate wait is microseconds)
// sets the sticky error word to 3 and ends, and the waves waiting on it do the same in turn -- a protocol bug ends as an error
// from ciao_ctx_synchronize, never as a hung GPU.  The count costs nothing on the path where the first look succeeds.
constexpr unsigned int WS_SPIN_LIMIT = 1u << 21;
__device__ __forceinline__ void ws_spin(unsigned int &spins, int *errflag)
{
    if (++spins > WS_SPIN_LIMIT) {
        *errflag = 3;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // an issuer's LDS-DMA in flight must have landed before its wave is gone
        __builtin_amdgcn_endpgm();
    }
}

// The exchange, in inline asm so that it is exactly these instructions.  Arrival (lane 63 only): the wave's partial(s), then +1 on
// the parity's counter -- one wave's LDS operations are performed in order.  Poll: counter FIRST, then the partials, one wait.
template <int OFF_VAL, int OFF_CNT>
__device__ __forceinline__ void ws_arrive(uint32_t red_w, uint32_t red0, float d1)
{
    asm volatile("ds_write_b32 %0, %1 offset:%4\n\tds_add_u32 %2, %3 offset:%5" ::"v"(red_w), "v"(d1), "v"(red0), "v"(1u), "n"(OFF_VAL), "n"(OFF_CNT) : "memory");
}
template <int OFF_VAL, int OFF_CNT>
__device__ __forceinline__ void ws_arrive(uint32_t red_w, uint32_t red0, double d1)
{
    asm volatile("ds_write_b64 %0, %1 offset:%4\n\tds_add_u32 %2, %3 offset:%5" ::"v"(red_w), "v"(d1), "v"(red0), "v"(1u), "n"(OFF_VAL), "n"(OFF_CNT) : "memory");
}
template <int OFF_VAL, int OFF_CNT>
__device__ __forceinline__ void ws_arrive2(uint32_t red_w, uint32_t red0, float d1, float d2)
{
    typedef float F2 __attribute__((ext_vector_type(2)));
    F2 pr;
    pr.x = d1;
    pr.y = d2;
    asm volatile("ds_write_b64 %0, %1 offset:%4\n\tds_add_u32 %2, %3 offset:%5" ::"v"(red_w), "v"(pr), "v"(red0), "v"(1u), "n"(OFF_VAL), "n"(OFF_CNT) : "memory");
}
template <int OFF_VAL, int OFF_CNT>
__device__ __forceinline__ void ws_arrive2(uint32_t red_w, uint32_t red0, double d1, double d2)
{
    typedef double D2 __attribute__((ext_vector_type(2)));
    D2 pr;
    pr.x = d1;
    pr.y = d2;
    asm volatile("ds_write_b128 %0, %1 offset:%4\n\tds_add_u32 %2, %3 offset:%5" ::"v"(red_w), "v"(pr), "v"(red0), "v"(1u), "n"(OFF_VAL), "n"(OFF_CNT) : "memory");
}
// Poll in two halves, so that work can sit between issue and wait: NRV 16-byte reads of the partials behind one read of the
// counter (issue); s_waitcnt for all of them, counter returned wave-uniform (wait).
template <int OFF_VAL, int OFF_CNT, typename V>
__device__ __forceinline__ void ws_poll_issue(uint32_t red0, unsigned int &c, V (&rv)[1])
{
    asm volatile("ds_read_b32 %0, %2 offset:%4\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(c), "=&v"(rv[0]) : "v"(red0), "n"(OFF_VAL), "n"(OFF_CNT) : "memory");
}
template <int OFF_VAL, int OFF_CNT, typename V>
__device__ __forceinline__ void ws_poll_issue(uint32_t red0, unsigned int &c, V (&rv)[2])
{
    asm volatile("ds_read_b32 %0, %3 offset:%6\n\tds_read_b128 %1, %3 offset:%4\n\tds_read_b128 %2, %3 offset:%5"
                 : "=&v"(c), "=&v"(rv[0]), "=&v"(rv[1]) : "v"(red0), "n"(OFF_VAL), "n"(OFF_VAL + 16), "n"(OFF_CNT) : "memory");
}
template <int OFF_VAL, int OFF_CNT, typename V>
__device__ __forceinline__ void ws_poll_issue(uint32_t red0, unsigned int &c, V (&rv)[4])
{
    asm volatile("ds_read_b32 %0, %5 offset:%10\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %5 offset:%7\n\t"
                 "ds_read_b128 %3, %5 offset:%8\n\tds_read_b128 %4, %5 offset:%9"
                 : "=&v"(c), "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3])
                 : "v"(red0), "n"(OFF_VAL), "n"(OFF_VAL + 16), "n"(OFF_VAL + 32), "n"(OFF_VAL + 48), "n"(OFF_CNT) : "memory");
}
template <typename V>
__device__ __forceinline__ unsigned int ws_poll_wait(unsigned int &c, V (&rv)[1])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(rv[0])::"memory");
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)c);
}
template <typename V>
__device__ __forceinline__ unsigned int ws_poll_wait(unsigned int &c, V (&rv)[2])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(rv[0]), "+v"(rv[1])::"memory");
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)c);
}
template <typename V>
__device__ __forceinline__ unsigned int ws_poll_wait(unsigned int &c, V (&rv)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3])::"memory");
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)c);
}

// the table-row DMA with agent scope (sc1): never served from a line the CU's vector L1 cached before a consumer's store.
// (The memory instruction reads a scalar COPY of the base made inside the asm: glds16s in chain_dma_kernels.h, and why.)
__device__ __forceinline__ void glds16s_sc1(const void *sbase, uint32_t voff, uint32_t lds_dst)
{
    uint64_t base_copy;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %3\n\ts_mov_b64 %0, %2\n\tglobal_load_lds_dwordx4 %1, %0 sc1"
                 : "=&s"(base_copy) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}

// 16-byte store to (uniform base in SGPRs) + (32-bit lane offset): the consumers' only vector-memory instruction, in inline asm
// so that the hand-counted vmcnt wait is the only wait it ever gets.  THE BASE IS COPIED BY A SCALAR INSTRUCTION FIRST: the base is
// the table row's address of the PREVIOUS step (`tprev`), live across a whole step -- the first thing hipcc spills when scalar
// registers are short -- and a spilled register comes back by v_readlane_b32 directly in front of its use.  "VALU writes SGPR ->
// VMEM reads that SGPR" needs five wait states that no one inserts for an inline asm: the store then goes to whatever the register
// pair held before (measured: 92 % of the time with no wait state in between, profiles/r05_sgpr_hazard_lab.txt) -- round 4's
// "Memory access fault by GPU" in the variant of this kernel that spilled 142 scalar registers (CHANGELOG round 5).
template <typename V>
__device__ __forceinline__ void gstore16s(void *sbase, uint32_t voff, V data)
{
    uint64_t base_copy;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0" : "=&s"(base_copy) : "v"(voff), "v"(data), "s"(sbase) : "memory");
}


template <typename T, int J, int ALG, int LOSS, bool MASKED, int NISS>
__global__ void __launch_bounds__((WS_NCW + 1 + NISS) * WAVE) chain_ws_kernel(ChainArgs<T> a_by_value)
{
    // The arguments are read THROUGH THE KERNEL-ARGUMENT SEGMENT, field by field where they are used, not from the by-value
    // parameter: hipcc loads every field of a by-value argument into scalar registers in the entry block, and the fields only the
    // stager or the final stores need then stay live through the consumers' loop -- 10-14 scalar registers spilled to VGPR lanes
    // and re-read every step (0.369 us per SAGA update against 0.361).  This way: 89-94 SGPRs, no spill.
    typedef const __attribute__((address_space(4))) ChainArgs<T> KernArgs;
    KernArgs &a = *(KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_by_value;

    using V = typename VecOfC<T>::type;
    using LY = WsLayout<T, J, ALG>;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NW = WS_NCW;
    constexpr bool HAS_TABLE = LY::TABLE;
    constexpr bool TWO = LY::TWO;
    constexpr bool SVRG_ANY = (ALG == CA_SVRG || ALG == CA_SVRGC);
    constexpr int R = LY::R;
    constexpr int HALF = R / 2;
    constexpr int ROW_BYTES = LY::ROW_BYTES;
    constexpr int NPIECE = ROW_BYTES / 1024;                      // 1 KiB wave-instructions per row
    constexpr int PPW = (NPIECE / NISS) * (HAS_TABLE ? 2 : 1);   // DMA instructions per step and issuer wave
    constexpr int LAGMAX = 63 / PPW;
    constexpr int LAG = (R - HALF - 1) < LAGMAX ? ((R - HALF - 1) < 1 ? 1 : (R - HALF - 1)) : LAGMAX;   // steps in flight behind `landed`
    constexpr int STALE_W = R + WS_STORE_LAG + 3;   // + 1 for the deferred store, + 2 of margin
    static_assert(ALG == CA_SVRG || ALG == CA_SVRGC || ALG == CA_SAGA, "SVRG and SAGA chains");
    static_assert(NPIECE % NISS == 0 && PPW <= 63 && R >= 2 && (R % 2) == 0, "ring / issuer geometry");
    static_assert(WS_RR % R == 0 && STALE_W < WS_RR - 64, "record ring geometry");
    static_assert(WS_STORE_LAG * J <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *ringA = dsm + LY::ringA;
    unsigned char *ringT = dsm + LY::ringT;
    WsRec<T> *rec = reinterpret_cast<WsRec<T> *>(dsm + LY::rec);
    WsIss *iss = reinterpret_cast<WsIss *>(dsm + LY::iss);
    int64_t *srows = reinterpret_cast<int64_t *>(dsm + LY::rows);
    unsigned char *red = dsm + LY::red;
    WsCtl *ctl = reinterpret_cast<WsCtl *>(dsm + LY::ctl);

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nsteps = a.nsteps;

    // ---- common init ----
    int64_t *s_sh = reinterpret_cast<int64_t *>(dsm + LY::shards);
    if (a.nshards > 0) shard_table_to_lds<T>(s_sh, tid);
    for (int e = tid; e < WS_RR; e += blockDim.x) srows[e] = -1;
    if (tid < (int)(sizeof(WsCtl) / 4)) reinterpret_cast<unsigned int *>(ctl)[tid] = 0;
    __syncthreads();   // the only barrier of the kernel: every wave is still here

    // number of complete steps (all four partials written), from the two parity counters; a lower bound by construction
    auto complete = [&]() -> int64_t {
        const unsigned int c0 = lds_peek(&ctl->cnt[0][0]), c1 = lds_peek(&ctl->cnt[1][0]);
        return (int64_t)(c0 >> 2) + (int64_t)(c1 >> 2);
    };

    if (wave == WS_NCW) {
        // =================================================== STAGER ===================================================
        for (int64_t S = 0; S < nsteps; S += WAVE) {
            // the records of steps S .. S+63 replace those of steps S-RR ..: all of them must be complete
            for (unsigned int spins = 0; complete() < S + WAVE - WS_RR; ws_spin(spins, a.errflag)) __builtin_amdgcn_s_sleep(8);
            asm volatile("" ::: "memory");
            int64_t st = S + lane;
            if (st > nsteps - 1) st = nsteps - 1;
            int64_t r = a.idx[st];
            if ((uint64_t)r >= (uint64_t)a.N) {
                *a.errflag = 1;
                r = 0;
            }
            const T *arow, *bp;
            T *trow = nullptr;
            if (a.nshards > 0) {   // global row -> its shard's memory (which may be another GPU's)
                const ShardRow<T> sr = shard_resolve<T>(s_sh, a.nshards, r, a.ld, a.d);
                arow = sr.arow;
                bp = sr.bp;
                if (HAS_TABLE) trow = sr.trow;
            } else {
                arow = a.A + r * a.ld;
                bp = a.b ? a.b + r : nullptr;
                if (HAS_TABLE) trow = a.table + r * a.d;
            }
            const T bi = bp ? *bp : T(0);
            T gi = T(0);
            if (ALG == CA_SVRGC) {
                // what the step needs of a_i'z_full is the link coefficient at it (chain_dma_kernel stages the same value)
                const T gv = a.gam ? a.gam[r] : a.gam_uniform;
                gi = grad_coef_t<T, LOSS>(gv, bi, a.lam).coef();
            }
            const int e = (int)((S + lane) & (WS_RR - 1));
            int stale = 0;
            if (HAS_TABLE) {
                srows[e] = r;   // this wave's own LDS operations are performed in order: the reads below see all 64 writes
#pragma unroll 4
                for (int k = 1; k <= STALE_W; ++k) stale |= (srows[(e - k) & (WS_RR - 1)] == r) ? 1 : 0;
            }
            WsRec<T> rc;
            rc.bi = bi;
            rc.gi = gi;
            rc.tptr = trow;
            rc.stale = stale;
            rc.pad = 0;
            rec[e] = rc;
            WsIss is;
            is.aptr = reinterpret_cast<const unsigned char *>(arow);
            is.tptr = reinterpret_cast<const unsigned char *>(trow);
            iss[e] = is;
            const int64_t upto = (S + WAVE < nsteps) ? S + WAVE : nsteps;
            if (lane == 0) lds_poke(&ctl->staged[0], (unsigned int)upto);   // after the records, in this wave's LDS order
        }
        return;
    }

    if (wave > WS_NCW) {
        // =================================================== ISSUER ===================================================
        const int q = wave - (WS_NCW + 1);
        const uint32_t ringA_off = (uint32_t)(uintptr_t)ringA, ringT_off = (uint32_t)(uintptr_t)ringT;
        // this wave's pieces of a row: k = q, q + NISS, ...; lane offsets, dead chunks (beyond the row) redirected to chunk 0
        uint32_t voff[NPIECE / NISS];
        const int64_t rowb = a.d * (int64_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < NPIECE / NISS; ++i) {
            const uint32_t o = (uint32_t)((q + i * NISS) * 1024 + lane * 16);
            voff[i] = (!MASKED || (int64_t)o < rowb) ? o : 0u;
        }
        int64_t done = 0;   // complete steps, as last seen
        for (int64_t S = 0; S < nsteps; S += WAVE) {
            const int nb = (int)((nsteps - S) < WAVE ? (nsteps - S) : WAVE);
            for (unsigned int spins = 0; (int64_t)lds_peek(&ctl->staged[0]) < S + nb; ws_spin(spins, a.errflag)) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
            const WsIss mine = iss[(S + lane) & (WS_RR - 1)];   // lane l: the addresses of step S + l
            const int64_t ap_v = (int64_t)(uintptr_t)mine.aptr, tp_v = (int64_t)(uintptr_t)mine.tptr;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int l = 0; l < nb; ++l) {
                const int64_t s = S + l;
                if (s - R + 1 > done) {   // slot s mod R still holds the row of step s - R: that step must be complete
                    done = complete();
                    if (s - R + 1 > done) {
                        // blocked: nothing further will be issued for a while, so retire and publish everything issued so
                        // far -- the consumers may be waiting for exactly those rows
                        wait_vmcnt<0>();
                        if (lane == 0) lds_poke(&ctl->landed[q][0], (unsigned int)s);
                        unsigned int spins = 0;
                        do {
                            __builtin_amdgcn_s_sleep(1);
                            ws_spin(spins, a.errflag);
                            done = complete();
                        } while (s - R + 1 > done);
                    }
                }
                const uint32_t slot = (uint32_t)(s & (R - 1)) * ROW_BYTES;
                const unsigned char *ap = reinterpret_cast<const unsigned char *>((uintptr_t)(
                    ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)ap_v >> 32), l) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ap_v, l)));
#pragma unroll
                for (int i = 0; i < NPIECE / NISS; ++i) glds16s(ap, voff[i], ringA_off + slot + (uint32_t)((q + i * NISS) * 1024));
                if (HAS_TABLE) {
                    const unsigned char *tp = reinterpret_cast<const unsigned char *>((uintptr_t)(
                        ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)tp_v >> 32), l) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tp_v, l)));
#pragma unroll
                    for (int i = 0; i < NPIECE / NISS; ++i) glds16s_sc1(tp, voff[i], ringT_off + slot + (uint32_t)((q + i * NISS) * 1024));
                }
                wait_vmcnt<PPW * LAG>();   // the rows of steps <= s - LAG have landed
                if (s + 1 - LAG > 0 && lane == 0) lds_poke(&ctl->landed[q][0], (unsigned int)(s + 1 - LAG));
            }
        }
        wait_vmcnt<0>();
        if (lane == 0) lds_poke(&ctl->landed[q][0], (unsigned int)nsteps);
        return;
    }

    // ====================================================== CONSUMERS ======================================================
    const int wib = wave;
    constexpr int CNT_OFF = (int)(LY::ctl - LY::red);                       // cnt[0] relative to red; cnt[1] 64 bytes on
    uint32_t red0 = (uint32_t)(uintptr_t)red;                                // LDS byte addresses of the exchange area,
    uint32_t red_w = red0 + (uint32_t)wib * (TWO ? 2 : 1) * (uint32_t)sizeof(T);   // held in two VGPRs for the whole chain
    asm volatile("" : "+v"(red0), "+v"(red_w));
    // chunk ownership: thread t owns 16-byte chunks t + 256*j; with MASKED those at or beyond the row's end are dead (their
    // state stays zero, what the ring holds for them is discarded, their stores are predicated off)
    const int64_t nchunks = a.d / VEC;
    bool ok[J];
    int64_t cl[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t c = tid + (int64_t)j * 256;
        ok[j] = !MASKED || c < nchunks;
        cl[j] = ok[j] ? c : 0;
    }
    V av[J], p[J], zf[J], zs[J], plo[J], phi[J];
    const T plam = (a.g.kind == CIAO_PROX_L1) ? a.g.lam : T(0);
    const bool hasbox = (a.g.kind == CIAO_PROX_BOX);
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
            if (a.g.kind == CIAO_PROX_BOX && ok[j]) {
                plo[j][v] = a.g.lo_vec ? a.g.lo_vec[c * VEC + v] : a.g.lo;
                phi[j][v] = a.g.hi_vec ? a.g.hi_vec[c * VEC + v] : a.g.hi;
            }
        }
    }
    V gav[J];   // SVRG: av is constant over the inner cycle
#pragma unroll
    for (int j = 0; j < J; ++j) gav[j] = a.gamma * av[j];
    drain_vmcnt_visible();   // the state loads are retired: from here on the only vector-memory operations are asm stores

    struct StepIn {
        V ar[J], sr[J];
        T bi, gi;
        T *tptr;
        int stale;
    };
    StepIn in[2];
    auto fetch = [&](StepIn &x, int slot, int64_t s) {   // plain LDS reads of step s: its ring slot and its record
#pragma unroll
        for (int j = 0; j < J; ++j) {
            x.ar[j] = *reinterpret_cast<const V *>(ringA + slot * ROW_BYTES + ((j * NW + wib) * 64 + lane) * 16);
            if (HAS_TABLE) x.sr[j] = *reinterpret_cast<const V *>(ringT + slot * ROW_BYTES + ((j * NW + wib) * 64 + lane) * 16);
            // (MASKED: the dead chunks are zeroed when the step that uses them begins, mask_dead() -- here the wave would have to
            // wait for the reads it has just issued)
        }
        const WsRec<T> *rc = &rec[s & (WS_RR - 1)];
        x.bi = rc->bi;
        x.gi = (ALG == CA_SVRGC) ? rc->gi : T(1);
        x.tptr = HAS_TABLE ? rc->tptr : nullptr;
        x.stale = HAS_TABLE ? rc->stale : 0;
    };
    auto mask_dead = [&](StepIn &x) {   // rows shorter than the consumers' reach: what the ring holds for the dead chunks is discarded
        if constexpr (MASKED) {
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (!ok[j]) {
                    x.ar[j] = V(T(0));
                    if (HAS_TABLE) x.sr[j] = V(T(0));
                }
        }
    };
    auto wait_landed = [&](int64_t need) {   // rows of steps < need are in the ring (and their records staged before them)
        if (need > nsteps) need = nsteps;
        for (unsigned int spins = 0;; ws_spin(spins, a.errflag)) {
            unsigned int l = lds_peek(&ctl->landed[0][0]);
            if (NISS > 1) {
                const unsigned int l1 = lds_peek(&ctl->landed[1][0]);
                l = l1 < l ? l1 : l;
            }
            if ((int64_t)l >= need) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");   // nothing that reads the ring or the records moves above the spin
    };

    // What a step leaves for the NEXT step's exchange to hide: SVRG's `z += w` (SVRG_basic.jl:81) of the iterate it produced, SAGA's
    // store of the table row it produced (SAGA_basic.jl:65).  Neither is needed by the dot product that follows.
    V gprev[J];          // SAGA: the new table row of the previous step, not yet stored
    T *tprev = nullptr;  // ... and where it goes
#pragma unroll
    for (int j = 0; j < J; ++j) gprev[j] = V(T(0));
    auto store_prev = [&]() {
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (!MASKED || ok[j]) gstore16s(tprev, (uint32_t)cl[j] * 16u, gprev[j]);
    };

    // one ring revolution (R steps), in versions selected once per revolution: with / without the IndBox clamp (HB), with /
    // without end-of-chain checks (CHK), SAG / SAGA.
    // The step is laid out around the exchange's two shadows (tools/micro/xchg_lab.hip: with the counter poll, two dozen
    // independent instructions placed there cost nothing, behind an s_barrier they cost most of their issue time):
    //   dot, 6 DPP stages, ARRIVE | next step's LDS reads, q1, q2 | POLL ISSUE | z += w of the previous step / previous table
    //   row's store | POLL WAIT | sum of the partials, link function, update + prox
    auto group = [&](auto hb_tag, auto chk_tag, auto sag_tag, const int64_t s0) {
        constexpr bool HB = decltype(hb_tag)::value;
        constexpr bool CHK = decltype(chk_tag)::value;
        constexpr bool SAG = decltype(sag_tag)::value;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t s = s0 + u;
            if (CHK && s >= nsteps) return;
            const int par = u & 1;   // s0 is a multiple of R (even): the step's parity is u's
            if (u == 0) wait_landed(s0 + HALF + 1);          // the prefetches of the first half reach step s0 + HALF
            if (u == HALF) wait_landed(s0 + R + 1);          // ... of the second half, step s0 + R
            StepIn &x = in[u & 1];
            mask_dead(x);   // read in the previous step's first shadow: long here
            const T bi = x.bi;
            T *const tptr = HAS_TABLE ? reinterpret_cast<T *>((uintptr_t)uniform64((int64_t)(uintptr_t)x.tptr)) : nullptr;
            const bool has_prev = (u > 0) || (s0 > 0);   // compile-time true except in the first step of a revolution

            if (HAS_TABLE && __builtin_amdgcn_readfirstlane(x.stale)) {
                // the row may have been rewritten after its DMA was issued: this very thread stored (or is about to store) these
                // bytes, so program order makes them visible to its own load.  The previous step's row goes out first (it is
                // stored again in this step's shadow: the same bytes).
                if (has_prev) store_prev();
                const V *sp = reinterpret_cast<const V *>(tptr);
#pragma unroll
                for (int j = 0; j < J; ++j) x.sr[j] = ok[j] ? sp[cl[j]] : V(T(0));
                drain_vmcnt_visible();
            }

            T d1 = T(0), d2 = T(0);
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    d1 = fmad(x.ar[j][v], p[j][v], d1);
                    if (TWO) d2 = fmad(x.ar[j][v], zf[j][v], d2);
                }
            // the six DPP stages of the wave sum (wave_sum_lane63, stage for stage), with SVRG's q1 = gamma * a_i -- which needs
            // nothing of this step -- issued in the wait states between them instead of s_nops
            V q1[J];
            {
                int nq = 0;   // elements of q1 placed so far (compile-time after unrolling)
                auto fill = [&]() {
                    if (SVRG_ANY && nq < J * VEC) {
                        asm volatile("" : "+v"(x.ar[nq / VEC][nq % VEC]));   // not before this point
                        q1[nq / VEC][nq % VEC] = a.gamma * x.ar[nq / VEC][nq % VEC];
                        asm volatile("" : "+v"(q1[nq / VEC][nq % VEC]));
                        ++nq;
                    }
                };
                auto stage = [&](auto f) {
                    d1 = f(d1);
                    if (TWO) d2 = f(d2);
                    asm volatile("" : "+v"(d1));
                    if (TWO) asm volatile("" : "+v"(d2));
                    fill();
                };
                asm volatile("" : "+v"(d1));
                stage([](T v) { return v + dpp_mov<0xB1>(v); });
                stage([](T v) { return v + dpp_mov<0x4E>(v); });
                stage([](T v) { return v + dpp_mov<0x141>(v); });
                stage([](T v) { return v + dpp_mov<0x140>(v); });
                stage([](T v) { return v + dpp_rows<0x142, 0xA>(v); });
                d1 = d1 + dpp_rows<0x143, 0xC>(d1);
                if (TWO) d2 = d2 + dpp_rows<0x143, 0xC>(d2);
                if (SVRG_ANY) {   // what did not fit between the stages
                    for (; nq < J * VEC; ++nq) q1[nq / VEC][nq % VEC] = a.gamma * x.ar[nq / VEC][nq % VEC];
                }
            }
            // ---- ARRIVE.  lane 63: the wave's partial(s), then one arrival on the parity's counter (performed in this order)
            if (lane == WAVE - 1) {
                if (par == 0) {
                    if constexpr (TWO) ws_arrive2<0, CNT_OFF>(red_w, red0, d1, d2); else ws_arrive<0, CNT_OFF>(red_w, red0, d1);
                } else {
                    if constexpr (TWO) ws_arrive2<128, CNT_OFF + 64>(red_w, red0, d1, d2); else ws_arrive<128, CNT_OFF + 64>(red_w, red0, d1);
                }
            }
            // ---- first shadow (the partials travel, the other waves arrive).  The empty asm statements are volatile, so they
            // stay between ARRIVE and POLL ISSUE; what they (re)define cannot be computed before / used after that point.
#pragma unroll
            for (int j = 0; j < J; ++j) asm volatile("" : "+v"(p[j]));
            if (!CHK || s + 1 < nsteps) fetch(in[(u + 1) & 1], (u + 1) % R, s + 1);   // next step's inputs: read, do not wait
            V q2[J];
            if (SVRG_ANY) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    q2[j] = p[j] - gav[j];
                    asm volatile("" : "+v"(q1[j]), "+v"(q2[j]));
                }
            }
            // ---- POLL ISSUE: counter first, then the four partials, one batch of LDS reads
            const unsigned int expect = (unsigned int)(2 * s + 4 - 2 * par);   // 4 * (s / 2 + 1)
            constexpr int NRV = 4 * (TWO ? 2 : 1) * (int)sizeof(T) / 16;   // 16-byte reads that hold the partials
            V rv[NRV];
            unsigned int cv;
            if (par == 0) ws_poll_issue<0, CNT_OFF>(red0, cv, rv); else ws_poll_issue<128, CNT_OFF + 64>(red0, cv, rv);
            // ---- second shadow (the reads travel)
            if (SVRG_ANY) {
                if (has_prev) {
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        asm volatile("" : "+v"(p[j]));
                        zs[j] += p[j];                     // SVRG_basic.jl:81 for the iterate of the previous step
                        asm volatile("" : "+v"(zs[j]));
                    }
                }
            } else {
                if (has_prev) {
                    store_prev();                          // SAGA_basic.jl:65 for the previous step
                    // this wave's table stores of WS_STORE_LAG steps ago are written (stores are its only vector-memory
                    // operations; a masked wave may have issued fewer, which only makes the wait stricter)
                    wait_vmcnt<WS_STORE_LAG * J>();
                }
            }
            // ---- POLL WAIT: until the counter read BEFORE the partials shows this step's four arrivals (the first look almost
            // always does: the shadows gave the other waves time to arrive)
            if (__builtin_expect(ws_poll_wait(cv, rv) != expect, 0)) {
                unsigned int spins = 0;
#pragma clang loop unroll(disable)
                do {
                    ws_spin(spins, a.errflag);
                    if (par == 0) ws_poll_issue<0, CNT_OFF>(red0, cv, rv); else ws_poll_issue<128, CNT_OFF + 64>(red0, cv, rv);
                } while (ws_poll_wait(cv, rv) != expect);
            }
            auto val = [&](int k) -> T { return rv[k / VEC][k % VEC]; };
            constexpr int ST = TWO ? 2 : 1;
            const T r0 = val(0), r1 = val(ST), r2 = val(2 * ST), r3 = val(3 * ST);
            {
                T lo = r0 + r1, hi = r2 + r3;
                if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(lo), "+v"(hi));
                d1 = lo + hi;
            }
            if (TWO) d2 = (val(1) + val(3)) + (val(5) + val(7));

            {
                const GradCoef<T> gp = grad_coef_t<T, LOSS>(d1, bi, a.lam);
                if (SVRG_ANY) {                                                  // SVRG_basic.jl:74-80
                    const T cz = (ALG == CA_SVRGC) ? x.gi : grad_coef_t<T, LOSS>(d2, bi, a.lam).coef();
                    const T gl = a.gamma * plam;
                    const T dc = cz - gp.coef();
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const T t = fmad(q1[j][v], dc, q2[j][v]);
                            p[j][v] = HB ? prox_bf(t, gl, plo[j][v], phi[j][v]) : prox_l1(t, gl);
                        }
                } else {                                                         // SAGA_basic.jl:56-64
                    const T gl = a.gamma * plam;
                    const T cp = gp.coef();
                    const T ngam = -a.gamma;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const T gn = x.ar[j][v] * cp;
                            const T del = gn - x.sr[j][v];
                            const T avn = fmad(del, a.invN, av[j][v]);
                            const T wv = fmad(ngam, SAG ? avn : del + av[j][v], p[j][v]);
                            av[j][v] = avn;
                            p[j][v] = HB ? prox_bf(wv, gl, plo[j][v], phi[j][v]) : prox_l1(wv, gl);
                            gprev[j][v] = gn;
                        }
                    }
                    tprev = tptr;
                }
            }
        }
    };
    auto pick_sag = [&](auto hb_tag, auto chk_tag, const int64_t s0) {
        if constexpr (ALG == CA_SAGA) {
            if (a.sag)
                group(hb_tag, chk_tag, std::true_type{}, s0);
            else
                group(hb_tag, chk_tag, std::false_type{}, s0);
        } else {
            group(hb_tag, chk_tag, std::false_type{}, s0);
        }
    };
    wait_landed(1);
    fetch(in[0], 0, 0);
    for (int64_t s0 = 0; s0 < nsteps; s0 += R) {
        if (s0 + R < nsteps) {
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
    // what the last step left behind (nsteps >= 1 here)
    if (SVRG_ANY) {
#pragma unroll
        for (int j = 0; j < J; ++j) zs[j] += p[j];
    } else {
        store_prev();
    }
    wait_vmcnt<0>();

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

}  // namespace ciao
