// xchg_lab.hip -- what does one all-to-all exchange of a scalar among the four waves of a workgroup cost on gfx950?
// A dependent chain: every iteration each wave reduces a double over its lanes (6 DPP stages), the four wave sums are exchanged
// through LDS, and the next iteration's value depends on the total.  Variants of the exchange:
//   0  ds_write (lane 63) -> s_waitcnt lgkmcnt(0) -> s_barrier -> ds_read x2 -> add            (chain_dma_kernel)
//   1  ds_write + ds_add_u32 counter (lane 63) -> poll {counter, 4 partials} until counter == 4k (chain_ws_kernel)
//   2  ds_write_b128 {value, sequence} (lane 63) -> poll 4 x b128 until the four sequences match (no atomic)
//   3  as 1, but s_sleep 0 between failed polls
// EXTRA waves (poll variants only) sit in the workgroup sleeping, as the producer waves of chain_ws_kernel do.
// Prints cycles per iteration (s_memtime) and ns per iteration (events).   hipcc --offload-arch=gfx950 -O3 xchg_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_rows(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROWS, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROWS, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    v += dpp_rows<0x142, 0xA>(v);
    v += dpp_rows<0x143, 0xC>(v);
    return v;
}

// RED = 1: the wave sum by two v_mfma_f64_16x16x4_f64 against a matrix of ones (A[i][k] = lane i + 16k: the first sums the
// four rows of 16 lanes column by column, three adds combine a lane's four column sums, the second sums those over the rows):
// the total in every lane, 5 instructions instead of 18 + s_nops
typedef double D4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double wave_sum_mfma(double v)
{
    D4 z = {0.0, 0.0, 0.0, 0.0};
    D4 c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
    const double s = (c.x + c.y) + (c.z + c.w);
    D4 t = __builtin_amdgcn_mfma_f64_16x16x4f64(s, 1.0, z, 0, 0, 0);
    return t.x;
}

// MODE 4: no exchange (the wave's own sum stands in for the total); 5: barrier WITHOUT the lgkmcnt(0) in front of it (not a
// correct program: for information); 6: as 0 with EXTRA waves that only execute the barriers; 7: as 0 but no in-wave reduction
template <int MODE, int WORK, int RED = 0>
__global__ void __launch_bounds__(512) k_xchg(int iters, double *out, long long *cyc, int extra)
{
    __shared__ __attribute__((aligned(16))) double red[2][8];      // [parity][wave] (+ sequence words in mode 2: 16 B per wave)
    __shared__ unsigned int cnt[2][16];
    __shared__ unsigned int done;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 32) reinterpret_cast<unsigned int *>(red)[threadIdx.x] = 0;
    if (threadIdx.x < 32) reinterpret_cast<unsigned int *>(cnt)[threadIdx.x] = 0;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave >= 4 && MODE == 6) {   // EXTRA waves that take part in every barrier and do nothing else
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
        return;
    }
    if (wave >= 4) {   // EXTRA waves: sleep until the chain is over
        while (__builtin_amdgcn_readfirstlane(*(volatile unsigned int *)&done) == 0) __builtin_amdgcn_s_sleep(8);
        return;
    }
    uint32_t red0 = (uint32_t)(uintptr_t)&red[0][0], cnt0 = (uint32_t)(uintptr_t)&cnt[0][0];
    asm volatile("" : "+v"(red0), "+v"(cnt0));
    double x = 1.0 + 1e-3 * threadIdx.x, acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        double d = x;
#pragma unroll
        for (int w = 0; w < WORK; ++w) d = __builtin_fma(d, 0.999, 1e-6);   // dependent arithmetic standing in for the step
        if (MODE != 7) d = RED ? wave_sum_mfma(d) : wave_sum_lane63(d);
        double tot;
        if (MODE == 4) {
            tot = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(d) >> 32), 63) << 32) |
                                       (unsigned int)__builtin_amdgcn_readlane((int)__double_as_longlong(d), 63));
        } else if (MODE == 0 || MODE == 5 || MODE == 6 || MODE == 7) {
            if (lane == 63) red[par][wave] = d;
            if (MODE != 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tot = (red[par][0] + red[par][1]) + (red[par][2] + red[par][3]);
        } else if (MODE == 1 || MODE == 3) {
            if (lane == 63)
                asm volatile("ds_write_b64 %0, %1\n\tds_add_u32 %2, %3" ::"v"(red0 + (uint32_t)(par * 64 + wave * 8)), "v"(d), "v"(cnt0 + (uint32_t)(par * 64)), "v"(1u) : "memory");
            const unsigned int expect = 4u * (unsigned int)(it / 2 + 1);
            typedef double D2 __attribute__((ext_vector_type(2)));
            D2 a, b;
            for (;;) {
                unsigned int c;
                asm volatile("ds_read_b32 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:16\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(c), "=&v"(a), "=&v"(b) : "v"(cnt0 + (uint32_t)(par * 64)), "v"(red0 + (uint32_t)(par * 64)) : "memory");
                if ((unsigned int)__builtin_amdgcn_readfirstlane((int)c) == expect) break;
                if (MODE == 3) __builtin_amdgcn_s_sleep(0);
            }
            tot = (a.x + a.y) + (b.x + b.y);
        } else {
            // slot = {value, sequence}: one 16-byte write per wave, four 16-byte reads per poll; red is 2 x 8 doubles = 2 x 4 slots
            typedef double D2 __attribute__((ext_vector_type(2)));
            const double seq = (double)(it + 1);
            if (lane == 63) {
                D2 sl;
                sl.x = d;
                sl.y = seq;
                asm volatile("ds_write_b128 %0, %1" ::"v"(red0 + (uint32_t)(par * 64 + wave * 16)), "v"(sl) : "memory");
            }
            D2 s0, s1, s2, s3;
            for (;;) {
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(red0 + (uint32_t)(par * 64)) : "memory");
                const bool ok = (s0.y == seq) & (s1.y == seq) & (s2.y == seq) & (s3.y == seq);
                if (__builtin_amdgcn_readfirstlane((int)ok)) break;
            }
            tot = (s0.x + s1.x) + (s2.x + s3.x);
        }
        x = x * 0.5 + tot * 1e-9;
        acc += x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[wave] = (long long)(t1 - t0);
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) *(volatile unsigned int *)&done = 1;
}

// Independent work (W1 + W2 instructions that do not depend on the total: the q1/q2, zs, refill, prefetch of a chain step) placed
// either AFTER the exchange (SH = 0) or in its shadows (SH = 1): W1 between the partial's write and the wait / first poll, W2
// between issuing the reads of the partials and waiting for them.  XM = 0 barrier, 1 counter poll.
template <int XM, int SH, int W1, int W2>
__global__ void __launch_bounds__(256) k_shadow(int iters, double *out, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double red[2][8];
    __shared__ unsigned int cnt[2][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 32) reinterpret_cast<unsigned int *>(red)[threadIdx.x] = 0;
    if (threadIdx.x < 32) reinterpret_cast<unsigned int *>(cnt)[threadIdx.x] = 0;
    __syncthreads();
    uint32_t red0 = (uint32_t)(uintptr_t)&red[0][0], cnt0 = (uint32_t)(uintptr_t)&cnt[0][0];
    asm volatile("" : "+v"(red0), "+v"(cnt0));
    double x = 1.0 + 1e-3 * threadIdx.x, acc = 0.0;
    double f1[W1 > 0 ? W1 : 1], f2[W2 > 0 ? W2 : 1];
    for (int i = 0; i < W1; ++i) f1[i] = 1.0 + i;
    for (int i = 0; i < W2; ++i) f2[i] = 2.0 + i;
    typedef double D2 __attribute__((ext_vector_type(2)));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        double d = wave_sum_lane63(x);
        const uint32_t slot = red0 + (uint32_t)(par * 64), cs = cnt0 + (uint32_t)(par * 64);
        D2 a, b;
        unsigned int c = 0;
        if (lane == 63) {
            if (XM == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(slot + (uint32_t)(wave * 8)), "v"(d) : "memory");
            else asm volatile("ds_write_b64 %0, %1\n\tds_add_u32 %2, %3" ::"v"(slot + (uint32_t)(wave * 8)), "v"(d), "v"(cs), "v"(1u) : "memory");
        }
        if (SH) {
#pragma unroll
            for (int i = 0; i < W1; ++i) { f1[i] = __builtin_fma(f1[i], 0.999, 1e-6); asm volatile("" : "+v"(f1[i])); }
        }
        if (XM == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(a), "=&v"(b) : "v"(slot) : "memory");
            if (SH) {
#pragma unroll
                for (int i = 0; i < W2; ++i) { f2[i] = __builtin_fma(f2[i], 0.999, 1e-6); asm volatile("" : "+v"(f2[i])); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
        } else {
            const unsigned int expect = 4u * (unsigned int)(it / 2 + 1);
            bool first = true;
            for (;;) {
                asm volatile("ds_read_b32 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:16" : "=&v"(c), "=&v"(a), "=&v"(b) : "v"(cs), "v"(slot) : "memory");
                if (SH && first) {
#pragma unroll
                    for (int i = 0; i < W2; ++i) { f2[i] = __builtin_fma(f2[i], 0.999, 1e-6); asm volatile("" : "+v"(f2[i])); }
                    first = false;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
                if ((unsigned int)__builtin_amdgcn_readfirstlane((int)c) == expect) break;
            }
        }
        const double tot = (a.x + a.y) + (b.x + b.y);
        if (!SH) {
#pragma unroll
            for (int i = 0; i < W1; ++i) { f1[i] = __builtin_fma(f1[i], 0.999, 1e-6); asm volatile("" : "+v"(f1[i])); }
#pragma unroll
            for (int i = 0; i < W2; ++i) { f2[i] = __builtin_fma(f2[i], 0.999, 1e-6); asm volatile("" : "+v"(f2[i])); }
        }
        x = x * 0.5 + tot * 1e-9;
        acc += x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < W1; ++i) acc += f1[i];
    for (int i = 0; i < W2; ++i) acc += f2[i];
    if (lane == 0) cyc[wave] = (long long)(t1 - t0);
    out[threadIdx.x] = acc;
}

template <int XM, int SH, int W1, int W2>
int run_shadow(const char *name)
{
    double *out;
    long long *cyc;
    CK(hipMalloc(&out, 512 * 8));
    CK(hipMalloc(&cyc, 8 * 8));
    const int iters = 200000;
    hipLaunchKernelGGL((k_shadow<XM, SH, W1, W2>), dim3(1), dim3(256), 0, 0, 2000, out, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k_shadow<XM, SH, W1, W2>), dim3(1), dim3(256), 0, 0, iters, out, cyc);
    CK(hipDeviceSynchronize());
    long long h[4];
    CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost));
    printf("%-28s independent work %2d + %2d, %s: %7.1f cycles/iter\n", name, W1, W2, SH ? "in the shadows  " : "after the exchange", (double)h[0] / iters);
    CK(hipFree(out));
    CK(hipFree(cyc));
    return 0;
}

template <int MODE, int WORK, int RED = 0>
int run(const char *name, int extra)
{
    double *out;
    long long *cyc;
    CK(hipMalloc(&out, 512 * 8));
    CK(hipMalloc(&cyc, 8 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 200000;
    const int threads = 256 + 64 * extra;
    hipLaunchKernelGGL((k_xchg<MODE, WORK, RED>), dim3(1), dim3(threads), 0, 0, 2000, out, cyc, extra);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_xchg<MODE, WORK, RED>), dim3(1), dim3(threads), 0, 0, iters, out, cyc, extra);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[4];
    CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost));
    printf("%-44s work=%2d extra=%d: %7.1f cycles/iter  %7.1f ns/iter\n", name, WORK, extra, (double)h[0] / iters, ms * 1e6 / iters);
    CK(hipFree(out));
    CK(hipFree(cyc));
    return 0;
}

int main()
{
    if (run_shadow<0, 0, 0, 0>("barrier")) return 1;
    if (run_shadow<1, 0, 0, 0>("counter poll")) return 1;
    if (run_shadow<0, 0, 8, 16>("barrier")) return 1;
    if (run_shadow<0, 1, 8, 16>("barrier")) return 1;
    if (run_shadow<1, 0, 8, 16>("counter poll")) return 1;
    if (run_shadow<1, 1, 8, 16>("counter poll")) return 1;
    if (run_shadow<0, 1, 12, 12>("barrier")) return 1;
    if (run_shadow<1, 1, 12, 12>("counter poll")) return 1;
    if (run_shadow<0, 1, 16, 8>("barrier")) return 1;
    if (run_shadow<1, 1, 16, 8>("counter poll")) return 1;
    if (run_shadow<0, 1, 0, 24>("barrier")) return 1;
    if (run_shadow<1, 1, 0, 24>("counter poll")) return 1;
    if (run_shadow<1, 1, 24, 0>("counter poll")) return 1;
    if (run_shadow<0, 1, 24, 0>("barrier")) return 1;
    if (run_shadow<0, 0, 16, 32>("barrier")) return 1;
    if (run_shadow<0, 1, 16, 32>("barrier")) return 1;
    if (run_shadow<1, 1, 16, 32>("counter poll")) return 1;
    if (run_shadow<1, 1, 32, 16>("counter poll")) return 1;
    if (run<0, 0>("0 barrier", 0)) return 1;
    if (run<1, 0>("1 counter poll", 0)) return 1;
    if (run<3, 0>("3 counter poll + s_sleep 0", 0)) return 1;
    if (run<2, 0>("2 sequence-tagged slots", 0)) return 1;
    if (run<1, 0>("1 counter poll", 3)) return 1;
    if (run<2, 0>("2 sequence-tagged slots", 3)) return 1;
    if (run<4, 0>("4 no exchange, DPP reduce", 0)) return 1;
    if (run<4, 0, 1>("4 no exchange, MFMA reduce", 0)) return 1;
    if (run<0, 0, 1>("0 barrier, MFMA reduce", 0)) return 1;
    if (run<7, 0>("7 barrier, no in-wave reduce", 0)) return 1;
    if (run<5, 0>("5 barrier without lgkmcnt(0) [unsafe]", 0)) return 1;
    if (run<6, 0>("6 barrier, extra waves in the barrier", 3)) return 1;
    if (run<6, 0, 1>("6 barrier + MFMA, extra waves in barrier", 3)) return 1;
    if (run<0, 16>("0 barrier", 0)) return 1;
    if (run<0, 16, 1>("0 barrier, MFMA reduce", 0)) return 1;
    if (run<1, 16>("1 counter poll", 0)) return 1;
    if (run<2, 16>("2 sequence-tagged slots", 0)) return 1;
    if (run<1, 16>("1 counter poll", 3)) return 1;
    if (run<2, 16>("2 sequence-tagged slots", 3)) return 1;
    return 0;
}
