// chain_dma_kernels.h -- the LDS-DMA chains: chain_dma_kernel (every real chain on rows of whole 16-byte chunks up to 32 KiB) and
// chain_cdma_kernel (complex T), with the LDS-DMA load / counted-wait helpers the other LDS-DMA kernels share.  Split out of
// chain_kernels.h in round 5.
#pragma once

#include "chain_common.h"
#include "chain_reg_kernels.h"

namespace ciao {

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

}  // namespace ciao
