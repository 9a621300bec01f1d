// launch_floor.hip -- calibration: what do dependent kernel launches cost on this box, eager and replayed from a graph?
// (a) trivial kernels of G workgroups; (b) the rows -> finalize shape of a Finito batch: G writers of 16 KiB partials, then
// 128 reducers that read every partial.  Prints microseconds per kernel.   hipcc --offload-arch=gfx950 -O3 launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_trivial(float *p) { p[blockIdx.x * 256 + threadIdx.x] += 1.0f; }

// writer: block b writes a 16 KiB partial (4096 floats) derived from x (16 KiB, read by everyone)
__global__ void __launch_bounds__(256) k_writer(const float *x, float *partial)
{
    const float4 *xv = reinterpret_cast<const float4 *>(x);
    float4 *pv = reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * 4096);
    for (int j = 0; j < 4; ++j) {
        float4 v = xv[threadIdx.x + 256 * j];
        v.x += 1.f; v.y += 2.f; v.z += 3.f; v.w += 4.f;
        pv[threadIdx.x + 256 * j] = v;
    }
}
// reducer: block c sums columns [32c, 32c+32) over all nparts partials, writes x
__global__ void __launch_bounds__(256) k_reducer(const float *partial, int nparts, float *x)
{
    __shared__ float4 lds[32][8];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const float4 *pp = reinterpret_cast<const float4 *>(partial + blockIdx.x * 32 + tx * 4);
    float4 s = make_float4(0, 0, 0, 0);
    for (int p0 = ty; p0 < nparts; p0 += 256) {
        float4 v[8];
        for (int u = 0; u < 8; ++u) { int p = p0 + 32 * u; v[u] = p < nparts ? pp[(size_t)p * 1024] : make_float4(0, 0, 0, 0); }
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    lds[ty][tx] = s;
    __syncthreads();
    if (ty == 0) {
        float4 t = lds[0][tx];
        for (int j = 1; j < 32; ++j) { t.x += lds[j][tx].x; t.y += lds[j][tx].y; t.z += lds[j][tx].z; t.w += lds[j][tx].w; }
        float4 *xo = reinterpret_cast<float4 *>(x + blockIdx.x * 32 + tx * 4);
        t.x *= 1e-3f; t.y *= 1e-3f; t.z *= 1e-3f; t.w *= 1e-3f;
        *xo = t;
    }
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float *p, *x, *partial;
    CK(hipMalloc(&p, 256 * 256 * 4));
    CK(hipMalloc(&x, 4096 * 4));
    CK(hipMalloc(&partial, (size_t)256 * 4096 * 4));
    CK(hipMemset(p, 0, 256 * 256 * 4));
    CK(hipMemset(x, 0, 4096 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int K = 2000;
    auto run = [&](const char *name, auto enqueue, int per) -> int {
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0;
            if (mode == 0) {
                for (int i = 0; i < 50; ++i) enqueue();
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < K; ++i) enqueue();
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
            } else {
                hipGraph_t g; hipGraphExec_t ex;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < K; ++i) enqueue();
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ex, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ex, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipGraphExecDestroy(ex));
                CK(hipGraphDestroy(g));
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %-6s %8.3f us per kernel\n", name, mode ? "graph" : "eager", ms * 1e3 / (K * per));
        }
        return 0;
    };
    for (int G : {1, 16, 64, 256}) {
        char nm[64];
        snprintf(nm, sizeof nm, "trivial kernel, %d workgroups", G);
        if (run(nm, [&] { hipLaunchKernelGGL(k_trivial, dim3(G), dim3(256), 0, st, p); }, 1)) return 1;
    }
    for (int G : {16, 64, 256}) {
        char nm[64];
        snprintf(nm, sizeof nm, "writer(%d x 16 KiB) -> reducer(128)", G);
        if (run(nm, [&] {
                hipLaunchKernelGGL(k_writer, dim3(G), dim3(256), 0, st, x, partial);
                hipLaunchKernelGGL(k_reducer, dim3(128), dim3(256), 0, st, partial, G, x);
            }, 2)) return 1;
    }
    printf("done\n");
    return 0;
}
