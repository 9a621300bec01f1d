// stream_copy.hip -- what the HBM of THIS box gives to plain streaming kernels with the read:write mixes of the table modes:
//   read     (1R)      sum reduction of a buffer                      (the GRAD sweep's mix)
//   copy     (1R:1W)   dst = src                                      (SAGA / Finito init: read A, write table)
//   update   (2R:1W)   dst = a + 0.5*dst                              (Finito batch: read A, read+write table)
//   triad    (3R:1W)   dst = a + b*dst + c ... (ProShI step: Q, q, table read; table written)
// 16 bytes per lane, 4 chunks in flight per lane, persistent grid; loads/stores non-temporal or default.  GB/s counts R+W.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float V __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ void __launch_bounds__(256) k(const V *__restrict__ a, const V *__restrict__ b, const V *__restrict__ c, V *dst, size_t n, float *sink)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    V acc = V(0.f);
    for (size_t i0 = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i0 < n; i0 += stride) {
        V va[4], vb[4], vc[4], vd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            if (i < n) {
                va[u] = NT ? __builtin_nontemporal_load(&a[i]) : a[i];
                if (MODE >= 2) vd[u] = NT ? __builtin_nontemporal_load(&dst[i]) : dst[i];
                if (MODE >= 3) { vb[u] = NT ? __builtin_nontemporal_load(&b[i]) : b[i]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            if (i < n) {
                V r = va[u];
                if (MODE == 0) { acc += r; continue; }
                if (MODE >= 2) r += 0.5f * vd[u];
                if (MODE >= 3) r += vb[u] * 0.25f;
                if (NT) __builtin_nontemporal_store(r, &dst[i]); else dst[i] = r;
            }
        }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.f;
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)8 << 30;   // 8 GiB per buffer: far beyond the 256 MiB Infinity Cache
    const size_t n = bytes / 16;
    V *a, *b, *dst; float *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(dst, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"read   1R   ", "copy   1R:1W", "update 2R:1W", "triad  3R:1W"};
    const double traffic[4] = {1, 2, 3, 4};
    for (int mode = 0; mode < 4; ++mode)
        for (int nt = 0; nt < 2; ++nt)
            for (int grid : {256, 512, 1024, 2048, 4096}) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(e0));
#define L(M, N) hipLaunchKernelGGL((k<M, N>), dim3(grid), dim3(256), 0, 0, a, b, b, dst, n, sink)
                    if (mode == 0) { if (nt) L(0, true); else L(0, false); }
                    if (mode == 1) { if (nt) L(1, true); else L(1, false); }
                    if (mode == 2) { if (nt) L(2, true); else L(2, false); }
                    if (mode == 3) { if (nt) L(3, true); else L(3, false); }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("%s %s grid=%4d  %7.1f GB/s\n", names[mode], nt ? "nt     " : "default", grid, traffic[mode] * bytes / (best * 1e-3) / 1e9);
            }
    return 0;
}
