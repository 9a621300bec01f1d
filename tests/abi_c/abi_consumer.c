/*
 * abi_consumer.c -- a plain C program that drives libciao_hip.so through include/ciao_hip.h with nothing but the HIP
 * runtime for device memory: no Python, no torch.  It is what a non-Python host (the Julia ccall wrapper, a C++ service)
 * sees of the drop-in boundary.  Built by __graft_entry__.build(); run by tests/test_gpu_abi_c.py on the GPU box.
 *
 * Checks, against loops written out here in C (double precision, the reference's formulas):
 *   ciao_full_gradient   av = (1/N) sum_i lam (a_i'x - b_i) a_i                      SVRG_basic.jl:58-63
 *   ciao_proxgrad_step   y = soft_threshold(x - gamma av, gamma lam_g)                SVRG_basic.jl:80 form
 *   ciao_saga_init + ciao_saga_steps with a given index stream                        SAGA_basic.jl:41-48, :53-68
 * and the error convention (negative status + ciao_last_error text, nothing thrown).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ciao_hip.h"

#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)
#define CIAOCHK(x)                                                                             \
    do {                                                                                       \
        int32_t s_ = (x);                                                                      \
        if (s_ != CIAO_OK) {                                                                   \
            fprintf(stderr, "ciao error %d (%s) at %s:%d\n", s_, ciao_last_error(), __FILE__, __LINE__); \
            return 3;                                                                          \
        }                                                                                      \
    } while (0)

static double soft(double v, double t) { return v > t ? v - t : (v < -t ? v + t : 0.0); }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void)
{   /* splitmix64 -> [0,1) */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}

static double maxabsdiff(const double *a, const double *b, int n)
{
    double m = 0;
    for (int i = 0; i < n; ++i) {
        const double e = fabs(a[i] - b[i]);
        if (e > m) m = e;
    }
    return m;
}

int main(void)
{
    const int N = 300, d = 1024, K = 500;   /* d*8 = 8 KiB rows: the wave-per-row sweep and the LDS-DMA chain */
    const double lam_f = (double)N, lam_g = 0.01, gamma = 1.0 / (3.0 * 1.5 * lam_f);
    if (ciao_abi_version() != CIAO_ABI_VERSION) {
        fprintf(stderr, "ABI version mismatch\n");
        return 1;
    }
    double *A = malloc(sizeof(double) * N * d), *b = malloc(sizeof(double) * N), *x = malloc(sizeof(double) * d);
    for (int i = 0; i < N * d; ++i) A[i] = (urand() - 0.5) * 2.0 / sqrt((double)d) * 1.7;
    for (int i = 0; i < N; ++i) b[i] = urand() - 0.5;
    for (int k = 0; k < d; ++k) x[k] = 0.3 * (urand() - 0.5);
    int64_t *idx = malloc(sizeof(int64_t) * K);
    for (int s = 0; s < K; ++s) idx[s] = (int64_t)(urand() * N) % N;

    double *dA, *db, *dx, *dav, *dy, *dz, *dtable;
    int64_t *didx;
    HIPCHK(hipMalloc((void **)&dA, sizeof(double) * N * d));
    HIPCHK(hipMalloc((void **)&db, sizeof(double) * N));
    HIPCHK(hipMalloc((void **)&dx, sizeof(double) * d));
    HIPCHK(hipMalloc((void **)&dav, sizeof(double) * d));
    HIPCHK(hipMalloc((void **)&dy, sizeof(double) * d));
    HIPCHK(hipMalloc((void **)&dz, sizeof(double) * d));
    HIPCHK(hipMalloc((void **)&dtable, sizeof(double) * N * d));
    HIPCHK(hipMalloc((void **)&didx, sizeof(int64_t) * K));
    HIPCHK(hipMemcpy(dA, A, sizeof(double) * N * d, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, sizeof(double) * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dx, x, sizeof(double) * d, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(didx, idx, sizeof(int64_t) * K, hipMemcpyHostToDevice));

    ciao_ctx *ctx = NULL;
    CIAOCHK(ciao_ctx_create(0, NULL, &ctx));
    ciao_problem p = {CIAO_LOSS_LS, CIAO_F64, N, d, d, N, dA, db, lam_f};
    ciao_prox_desc g = {CIAO_PROX_L1, 0, lam_g, -INFINITY, INFINITY, NULL, NULL};

    /* ---- full gradient + fused prox step ---------------------------------------------------------------------------- */
    double *av_ref = calloc(d, sizeof(double)), *y_ref = malloc(sizeof(double) * d);
    for (int i = 0; i < N; ++i) {
        double dot = 0;
        for (int k = 0; k < d; ++k) dot += A[i * d + k] * x[k];
        const double c = lam_f * (dot - b[i]) / N;
        for (int k = 0; k < d; ++k) av_ref[k] += c * A[i * d + k];
    }
    for (int k = 0; k < d; ++k) y_ref[k] = soft(x[k] - gamma * av_ref[k], gamma * lam_g);
    double *av = malloc(sizeof(double) * d), *y = malloc(sizeof(double) * d);
    CIAOCHK(ciao_full_gradient(ctx, &p, dx, dav));
    CIAOCHK(ciao_ctx_synchronize(ctx));
    HIPCHK(hipMemcpy(av, dav, sizeof(double) * d, hipMemcpyDeviceToHost));
    const double e1 = maxabsdiff(av, av_ref, d);
    CIAOCHK(ciao_proxgrad_step(ctx, &p, &g, gamma, dx, dav, dy));
    CIAOCHK(ciao_ctx_synchronize(ctx));
    HIPCHK(hipMemcpy(y, dy, sizeof(double) * d, hipMemcpyDeviceToHost));
    const double e2 = maxabsdiff(y, y_ref, d);
    printf("full_gradient: max|err| = %.3e   proxgrad_step: max|err| = %.3e   kernel: %s\n", e1, e2, ciao_ctx_last_kernel(ctx));

    /* ---- SAGA: init + K steps with the given draws (SAGA_basic.jl:41-48, :53-68) --------------------------------------- */
    double *table = malloc(sizeof(double) * N * d), *avs = calloc(d, sizeof(double)), *z = malloc(sizeof(double) * d);
    for (int i = 0; i < N; ++i) {
        double dot = 0;
        for (int k = 0; k < d; ++k) dot += A[i * d + k] * x[k];
        const double c = lam_f * (dot - b[i]);
        for (int k = 0; k < d; ++k) {
            table[i * d + k] = c * A[i * d + k];
            avs[k] += table[i * d + k];
        }
    }
    for (int k = 0; k < d; ++k) {
        avs[k] /= N;
        z[k] = soft((1.0 - gamma) * x[k], gamma * lam_g);   /* the reference's z0 quirk, :48 */
    }
    for (int s = 0; s < K; ++s) {
        const int i = (int)idx[s];
        double dot = 0;
        for (int k = 0; k < d; ++k) dot += A[i * d + k] * z[k];
        const double c = lam_f * (dot - b[i]);
        for (int k = 0; k < d; ++k) {
            const double gnew = c * A[i * d + k], del = gnew - table[i * d + k];
            const double w = z[k] - gamma * (del + avs[k]);   /* :61 */
            avs[k] += del / N;                                 /* :62 */
            z[k] = soft(w, gamma * lam_g);                     /* :64 */
            table[i * d + k] = gnew;                           /* :65 */
        }
    }
    CIAOCHK(ciao_saga_init(ctx, &p, &g, gamma, dx, dtable, dav, dz));
    CIAOCHK(ciao_saga_steps(ctx, &p, &g, gamma, 0, K, didx, dtable, dav, dz));
    CIAOCHK(ciao_ctx_synchronize(ctx));
    double *zd = malloc(sizeof(double) * d), *avd = malloc(sizeof(double) * d);
    HIPCHK(hipMemcpy(zd, dz, sizeof(double) * d, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(avd, dav, sizeof(double) * d, hipMemcpyDeviceToHost));
    const double e3 = maxabsdiff(zd, z, d), e4 = maxabsdiff(avd, avs, d);
    printf("saga %d steps: max|err| z = %.3e  av = %.3e   kernel: %s\n", K, e3, e4, ciao_ctx_last_kernel(ctx));

    /* ---- error convention: a NULL vector is a status, not a crash --------------------------------------------------------- */
    const int32_t st = ciao_full_gradient(ctx, &p, NULL, dav);
    const int err_ok = (st == CIAO_ERR_ARG) && strlen(ciao_last_error()) > 0;
    printf("NULL argument -> status %d (\"%s\")\n", st, ciao_last_error());

    CIAOCHK(ciao_ctx_destroy(ctx));
    const double tol = 1e-10;
    const int ok = e1 < tol && e2 < tol && e3 < 1e-8 && e4 < 1e-8 && err_ok;
    printf("%s\n", ok ? "ABI_C_OK" : "ABI_C_FAIL");
    return ok ? 0 : 4;
}
