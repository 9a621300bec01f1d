// gather_ceiling.hip -- what does the memory system give a batch of r RANDOM rows of d elements (one read of the data row, one read
// and one write of the table row: the traffic of a Finito batch over an index list, Finito_basic.jl:110-117) when nothing but the
// traffic is done?  One wave per row, 16 bytes per lane, every load of a row issued at once, r / 4 workgroups: as much memory-level
// parallelism as the chip takes.  Rows of 50 / 100 / 255 elements, fp64 and fp32, r = 4096 and 65536; index lists: random, the same
// random rows sorted, and a contiguous block.  Prints us per batch and TB/s of the 3 d s bytes per row.
//   hipcc --offload-arch=gfx950 -O3 gather_ceiling.hip -o gather_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int G>   // bytes per lane: 16 or 4
__global__ void __launch_bounds__(256) batch_rw(const unsigned char *A, unsigned char *Tb, const int64_t *idx, int r, int rowb)
{
    const int w = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (w >= r) return;
    const int64_t row = idx[w];
    const unsigned char *a = A + row * rowb;
    unsigned char *t = Tb + row * rowb;
    if (G == 16) {
        typedef uint32_t V4 __attribute__((ext_vector_type(4)));
        V4 x[2], y[2];
        for (int k = 0; k < 2; ++k) {
            const int o = (k * 64 + lane) * 16;
            if (o < rowb) { x[k] = __builtin_nontemporal_load((const V4 *)(a + o)); y[k] = *(const V4 *)(t + o); }
        }
        for (int k = 0; k < 2; ++k) {
            const int o = (k * 64 + lane) * 16;
            if (o < rowb) { y[k] += x[k]; __builtin_nontemporal_store(y[k], (V4 *)(t + o)); }
        }
    } else {
        uint32_t x[8], y[8];
        for (int k = 0; k < 8; ++k) {
            const int o = (k * 64 + lane) * 4;
            if (o < rowb) { x[k] = __builtin_nontemporal_load((const uint32_t *)(a + o)); y[k] = *(const uint32_t *)(t + o); }
        }
        for (int k = 0; k < 8; ++k) {
            const int o = (k * 64 + lane) * 4;
            if (o < rowb) { y[k] += x[k]; __builtin_nontemporal_store(y[k], (uint32_t *)(t + o)); }
        }
    }
}

int main()
{
    const int64_t N = 4000000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int es : {8, 4})
        for (int d : {50, 100, 255}) {
            const int rowb = d * es;
            unsigned char *A, *Tb;
            CK(hipMalloc(&A, (size_t)N * rowb));
            CK(hipMalloc(&Tb, (size_t)N * rowb));
            CK(hipMemset(A, 1, (size_t)N * rowb));
            CK(hipMemset(Tb, 1, (size_t)N * rowb));
            for (int r : {4096, 65536}) {
                const int nb = 24;
                std::vector<int64_t> h((size_t)nb * r);
                int64_t *idx;
                CK(hipMalloc(&idx, h.size() * 8));
                for (int kind = 0; kind < 3; ++kind) {
                    uint64_t s = 88172645463325252ull;
                    for (int b = 0; b < nb; ++b) {
                        int64_t *p = h.data() + (size_t)b * r;
                        if (kind == 2) {
                            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                            const int64_t r0 = (int64_t)(s % (uint64_t)(N - r));
                            for (int i = 0; i < r; ++i) p[i] = r0 + i;
                        } else {
                            for (int i = 0; i < r; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; p[i] = (int64_t)(s % (uint64_t)N); }
                            std::sort(p, p + r);
                            p[0] += 0;
                            int64_t *e = std::unique(p, p + r);
                            for (int64_t *q = e; q < p + r; ++q) *q = (q[-1] + 1) % N;   // (distinct rows)
                            if (kind == 0) for (int i = r - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(p[i], p[s % (uint64_t)(i + 1)]); }
                        }
                    }
                    CK(hipMemcpy(idx, h.data(), h.size() * 8, hipMemcpyHostToDevice));
                    const bool g16 = rowb % 16 == 0;
                    auto launch = [&](int b) {
                        if (g16) hipLaunchKernelGGL(batch_rw<16>, dim3((r + 3) / 4), dim3(256), 0, 0, A, Tb, idx + (size_t)b * r, r, rowb);
                        else hipLaunchKernelGGL(batch_rw<4>, dim3((r + 3) / 4), dim3(256), 0, 0, A, Tb, idx + (size_t)b * r, r, rowb);
                    };
                    for (int b = 0; b < 4; ++b) launch(b);
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int b = 4; b < nb; ++b) launch(b);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    const double us = ms * 1e3 / (nb - 4);
                    printf("d=%3d %s r=%5d %-10s %7.1f us per batch (launch to launch)  %5.2f TB/s of 3 d s per row  [%d bytes per lane]\n", d, es == 8 ? "f64" : "f32", r,
                           kind == 0 ? "random" : (kind == 1 ? "sorted" : "contiguous"), us, 3.0 * rowb * r / us * 1e-6, g16 ? 16 : 4);
                }
                CK(hipFree(idx));
            }
            CK(hipFree(A));
            CK(hipFree(Tb));
        }
    return 0;
}
