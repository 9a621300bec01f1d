/*
 * ciao_oracle.c -- CPU oracle (TEST INFRASTRUCTURE; see ciao_oracle.h for the rules and the parity status).
 *
 * Build:  make -C oracle        (gcc -O2 -fno-fast-math, no -ffp-contract: FMA contraction is OFF so the
 *                                rounding sequence is the one Julia's scalar loops / 1 x d BLAS calls give)
 */
#include "ciao_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

/* ---- fp64 instance ---- */
#define R double
#define SFX(n) n##_f64
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#define R_FABS fabs
#define R_HYPOT hypot
#define R_EPS DBL_EPSILON
#include "ciao_oracle_impl.inc"
#undef R
#undef SFX
#undef R_EXP
#undef R_LOG
#undef R_SQRT
#undef R_FABS
#undef R_HYPOT
#undef R_EPS

/* ---- fp32 instance (the reference keeps Float32 problems in Float32: test_lasso.jl:74) ---- */
#define R float
#define SFX(n) n##_f32
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_HYPOT hypotf
#define R_EPS FLT_EPSILON
#include "ciao_oracle_impl.inc"
#undef R
#undef SFX
#undef R_EXP
#undef R_LOG
#undef R_SQRT
#undef R_FABS
#undef R_HYPOT
#undef R_EPS

/*
 * Best-case CPU full-gradient sweep on all host cores -- NOT a restatement of the reference (which is
 * single-threaded); used only by bench.py to report a "CPU-all" number beside the reference-shaped
 * single-thread one.  Each thread sums a contiguous block of rows into a private
 * d-vector; the privates are added in thread order.
 */
#define OMP_SWEEP(R, S, EXPF)                                                                                  \
    EXPORT int orc_full_pass_omp_##S(const orc_problem *p, const R *x, R *av)                                  \
    {                                                                                                          \
        const int64_t N = p->N, d = p->d;                                                                      \
        const R *A = (const R *)p->A, *b = (const R *)p->b;                                                    \
        const R lam = (R)p->lam;                                                                               \
        int nt = 1;                                                                                            \
        _Pragma("omp parallel")                                                                                \
        {                                                                                                      \
            _Pragma("omp single") nt = omp_get_num_threads();                                                  \
        }                                                                                                      \
        R *priv = (R *)calloc((size_t)nt * (size_t)d, sizeof(R));                                              \
        _Pragma("omp parallel num_threads(nt)")                                                                \
        {                                                                                                      \
            int t = omp_get_thread_num();                                                                      \
            R *acc = priv + (size_t)t * (size_t)d;                                                             \
            int64_t lo = N * t / nt, hi = N * (t + 1) / nt;                                                    \
            for (int64_t i = lo; i < hi; ++i) {                                                                \
                const R *a = A + i * d;                                                                        \
                R dot = (R)0;                                                                                  \
                for (int64_t k = 0; k < d; ++k) dot += a[k] * x[k];                                            \
                R c;                                                                                           \
                if (p->loss == ORC_LOSS_LS)                                                                    \
                    c = lam * (dot - b[i]);                                                                    \
                else if (p->loss == ORC_LOSS_LOGISTIC)                                                         \
                    c = -b[i] / ((R)1 + EXPF(b[i] * dot));                                                     \
                else                                                                                           \
                    c = (R)0;                                                                                  \
                for (int64_t k = 0; k < d; ++k) acc[k] += c * a[k];                                            \
            }                                                                                                  \
        }                                                                                                      \
        for (int64_t k = 0; k < d; ++k) {                                                                      \
            R s = (R)0;                                                                                        \
            for (int t = 0; t < nt; ++t) s += priv[(size_t)t * (size_t)d + (size_t)k];                         \
            av[k] = s / (R)N;                                                                                  \
        }                                                                                                      \
        free(priv);                                                                                            \
        return nt;                                                                                             \
    }

OMP_SWEEP(double, f64, exp)
OMP_SWEEP(float, f32, expf)

/* Raw sums of a row shard on all cores (see ciao_oracle.h): a checker for the sharded configs at their real size.  The
 * dot products and the products c_i a_ik are formed in double whatever R is, the sums are kept in long double (x87 80-bit):
 * the value is the exact-arithmetic sum to well below one eps(R) of its own magnitude even after 10^7 cancelling terms. */
#define OMP_SHARD(R, S, EXPF)                                                                                  \
    EXPORT int orc_shard_sums_omp_##S(const orc_problem *p, int64_t rows, const R *x, long double *acc)        \
    {                                                                                                          \
        const int64_t d = p->d;                                                                                \
        const R *A = (const R *)p->A, *b = (const R *)p->b;                                                    \
        const double lam = p->lam;                                                                             \
        int nt = 1;                                                                                            \
        _Pragma("omp parallel")                                                                                \
        {                                                                                                      \
            _Pragma("omp single") nt = omp_get_num_threads();                                                  \
        }                                                                                                      \
        long double *priv = (long double *)calloc((size_t)nt * (size_t)d, sizeof(long double));                \
        _Pragma("omp parallel num_threads(nt)")                                                                \
        {                                                                                                      \
            int t = omp_get_thread_num();                                                                      \
            long double *a2 = priv + (size_t)t * (size_t)d;                                                        \
            int64_t lo = rows * t / nt, hi = rows * (t + 1) / nt;                                              \
            for (int64_t i = lo; i < hi; ++i) {                                                                \
                const R *a = A + i * d;                                                                        \
                double dot = 0.0;                                                                              \
                for (int64_t k = 0; k < d; ++k) dot += (double)a[k] * (double)x[k];                            \
                double c;                                                                                      \
                if (p->loss == ORC_LOSS_LS)                                                                    \
                    c = lam * (dot - (double)b[i]);                                                            \
                else if (p->loss == ORC_LOSS_LOGISTIC)                                                         \
                    c = -(double)b[i] / (1.0 + exp((double)b[i] * dot));                                       \
                else                                                                                           \
                    c = 0.0;                                                                                   \
                for (int64_t k = 0; k < d; ++k) a2[k] += (long double)(c * (double)a[k]);                      \
            }                                                                                                  \
        }                                                                                                      \
        for (int64_t k = 0; k < d; ++k)                                                                        \
            for (int t = 0; t < nt; ++t) acc[k] += priv[(size_t)t * (size_t)d + (size_t)k];                    \
        free(priv);                                                                                            \
        return nt;                                                                                             \
    }

OMP_SHARD(double, f64, exp)
OMP_SHARD(float, f32, expf)
