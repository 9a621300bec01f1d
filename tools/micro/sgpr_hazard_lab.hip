// sgpr_hazard_lab.hip -- does gfx950 hardware interlock "VALU writes SGPR -> VMEM reads that SGPR", or does the vector-memory
// instruction read the STALE register pair when fewer than the documented 5 wait states separate them?
//
// hipcc inserts `s_nop 4` between a v_readfirstlane_b32 and a global load that uses the register as its scalar base (its hazard
// recognizer knows the rule for the instructions it emits); the operands of an INLINE-ASM global_load_lds / global_store are opaque
// to it.  Round 4's "Memory access fault by GPU" came from a chain_ws_kernel build with 142 scalar registers spilled to VGPR lanes:
// a spilled base is restored by v_readlane_b32 (a VALU write of an SGPR) right in front of the asm that uses it.
//
// The lab: s[20:21] holds the address of buffer A (every word 1: the STALE base).  Then v_readfirstlane_b32 / v_readlane_b32
// write the address of buffer B (every word 2: the FRESH base) into s[20:21], W wait states pass (W = 0 .. 6, `s_nop W-1`), and
// `global_load_dword v, voff, s[20:21]` loads.  A returned 1 is a read through the stale base.  Both bases are valid memory, so
// nothing can fault here.  Prints, per W and per way of writing the SGPRs, how many of the loads went through the stale base.
//   hipcc --offload-arch=gfx950 -O3 sgpr_hazard_lab.hip -o sgpr_hazard_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define NOP_0 ""
#define NOP_1 "s_nop 0\n\t"
#define NOP_2 "s_nop 1\n\t"
#define NOP_3 "s_nop 2\n\t"
#define NOP_4 "s_nop 3\n\t"
#define NOP_5 "s_nop 4\n\t"
#define NOP_6 "s_nop 5\n\t"

// MODE 0: v_readfirstlane_b32 (a wave-uniform pointer that was read from LDS); MODE 1: v_readlane_b32 lane 5 (a spill restore);
// MODE 2: as the chain kernels' asm until round 4: s_mov_b32 m0 + s_nop 0 between the write and the load (2 wait states built in) + W more;
// MODE 3: the round-5 form: the base COPIED by s_mov_b64 right behind its VALU write, the load reads the copy (+ W s_nops behind the copy)
#define BODY(MODE, W)                                                                                                             \
    for (int it = 0; it < iters; ++it) {                                                                                          \
        unsigned r;                                                                                                               \
        if (MODE == 0)                                                                                                            \
            asm volatile("s_mov_b64 s[20:21], %[pa]\n\ts_nop 7\n\t"                                                               \
                         "v_readfirstlane_b32 s20, %[lo]\n\tv_readfirstlane_b32 s21, %[hi]\n\t" NOP_##W                           \
                         "global_load_dword %[r], %[off], s[20:21]\n\ts_waitcnt vmcnt(0)"                                         \
                         : [r] "=&v"(r) : [pa] "s"(pa), [lo] "v"(lo), [hi] "v"(hi), [off] "v"(off) : "s20", "s21", "memory");     \
        else if (MODE == 1)                                                                                                       \
            asm volatile("s_mov_b64 s[20:21], %[pa]\n\ts_nop 7\n\t"                                                               \
                         "v_readlane_b32 s20, %[lo], 5\n\tv_readlane_b32 s21, %[hi], 5\n\t" NOP_##W                               \
                         "global_load_dword %[r], %[off], s[20:21]\n\ts_waitcnt vmcnt(0)"                                         \
                         : [r] "=&v"(r) : [pa] "s"(pa), [lo] "v"(lo), [hi] "v"(hi), [off] "v"(off) : "s20", "s21", "memory");     \
        else if (MODE == 3)                                                                                                       \
            asm volatile("s_mov_b64 s[20:21], %[pa]\n\ts_mov_b64 s[22:23], %[pa]\n\ts_nop 7\n\t"                                  \
                         "v_readlane_b32 s21, %[hi], 5\n\tv_readlane_b32 s20, %[lo], 5\n\t"                                       \
                         "s_mov_b64 s[22:23], s[20:21]\n\t" NOP_##W                                                                \
                         "global_load_dword %[r], %[off], s[22:23]\n\ts_waitcnt vmcnt(0)"                                         \
                         : [r] "=&v"(r) : [pa] "s"(pa), [lo] "v"(lo), [hi] "v"(hi), [off] "v"(off) : "s20", "s21", "s22", "s23", "memory"); \
        else                                                                                                                      \
            asm volatile("s_mov_b64 s[20:21], %[pa]\n\ts_nop 7\n\t"                                                               \
                         "v_readlane_b32 s21, %[hi], 5\n\tv_readlane_b32 s20, %[lo], 5\n\t"                                       \
                         "s_mov_b32 m0, %[mv]\n\ts_nop 0\n\t" NOP_##W                                                              \
                         "global_load_dword %[r], %[off], s[20:21]\n\ts_waitcnt vmcnt(0)"                                         \
                         : [r] "=&v"(r) : [pa] "s"(pa), [lo] "v"(lo), [hi] "v"(hi), [off] "v"(off), [mv] "s"(mv) : "s20", "s21", "m0", "memory"); \
        stale += (r == 1u) ? 1 : 0;                                                                                               \
        other += (r != 1u && r != 2u) ? 1 : 0;                                                                                    \
    }

template <int MODE, int W>
__global__ void lab(const unsigned *A, const unsigned *B, int iters, unsigned long long *count)
{
    const unsigned long long pa = (unsigned long long)A;
    const unsigned long long pb = (unsigned long long)B;
    unsigned lo = (unsigned)pb, hi = (unsigned)(pb >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));   // the fresh base lives in VGPRs
    const unsigned off = (threadIdx.x & 63) * 4;
    const unsigned mv = (unsigned)iters;   // (any scalar: m0 is only written, as in the chain kernels' asm)
    unsigned long long stale = 0, other = 0;
    if (W == 0) { BODY(MODE, 0) }
    if (W == 1) { BODY(MODE, 1) }
    if (W == 2) { BODY(MODE, 2) }
    if (W == 3) { BODY(MODE, 3) }
    if (W == 4) { BODY(MODE, 4) }
    if (W == 5) { BODY(MODE, 5) }
    if (W == 6) { BODY(MODE, 6) }
    atomicAdd(&count[0], stale);
    atomicAdd(&count[1], other);
}

template <int MODE, int W>
int run(const unsigned *A, const unsigned *B, unsigned long long *cnt, const char *what)
{
    const int iters = 2000, grid = 1024, block = 256;
    CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL((lab<MODE, W>), dim3(grid), dim3(block), 0, 0, A, B, iters, cnt);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    const unsigned long long total = (unsigned long long)iters * grid * block;
    printf("%-46s extra wait states %d: %12llu of %llu loads through the STALE base (%.4f %%), %llu neither\n", what, W, h[0], total,
           100.0 * (double)h[0] / (double)total, h[1]);
    return 0;
}

int main()
{
    unsigned *A, *B;
    unsigned long long *cnt;
    CK(hipMalloc(&A, 4096));
    CK(hipMalloc(&B, 4096));
    CK(hipMalloc(&cnt, 16));
    unsigned h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1;
    CK(hipMemcpy(A, h, 4096, hipMemcpyHostToDevice));
    for (int i = 0; i < 1024; ++i) h[i] = 2;
    CK(hipMemcpy(B, h, 4096, hipMemcpyHostToDevice));
#define ALLW(MODE, what)                                                                                             \
    if (run<MODE, 0>(A, B, cnt, what) || run<MODE, 1>(A, B, cnt, what) || run<MODE, 2>(A, B, cnt, what) ||           \
        run<MODE, 3>(A, B, cnt, what) || run<MODE, 4>(A, B, cnt, what) || run<MODE, 5>(A, B, cnt, what) ||           \
        run<MODE, 6>(A, B, cnt, what))                                                                                 \
        return 1;
    ALLW(0, "v_readfirstlane_b32 -> global_load saddr")
    ALLW(1, "v_readlane_b32 -> global_load saddr")
    ALLW(2, "v_readlane_b32, s_mov m0, s_nop 0 -> load (+2)")
    ALLW(3, "v_readlane_b32, s_mov_b64 copy -> load copy")
    return 0;
}
