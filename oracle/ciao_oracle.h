/*
 * ciao_oracle.h -- CPU oracle for the finite-sum hot path of CIAOAlgorithms.jl.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ciaoalgorithms.jl_amd/ (the product) may include, link, import or
 * execute anything in this directory; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / the CPU baseline.
 *
 * Parity status: PINNED by the reference's own known-answer tests (tests/test_oracle_pins.py):
 *   - literal l1-logistic fixture + hard-coded x_star     (/root/reference/test/test_logistic_l1.jl:12-29)
 *   - lasso known-answer generator                        (/root/reference/test/test_lasso.jl:15-47)
 *   - structural pins: maxit=1 returns the init state     (test_lasso.jl:188-192, :224-228)
 * What kind of pin that is: a CONVERGENCE pin.  The reference holds known ANSWERS (fixed points of the restated
 * operators), no per-step vectors, and cannot run here (no julia binary): there is no oracle/_ref build and no golden
 * vector produced by the reference itself.  So the fixed point of every restated algorithm is pinned by reference-held
 * data; the transient-only details (SAGA's z0 = prox((1-γ)x0), cyclic Finito starting at batch 2, the order of the
 * additions) are pinned by the structural tests and by reading the source line by line, i.e. by THIS restatement.
 * Per-step vectors under tests/golden/ are produced by this restatement and are labelled so.
 */
#ifndef CIAO_ORACLE_H
#define CIAO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ORC_LOSS_LS_COMPLEX / ORC_PROX_L1_COMPLEX: complex T (CIAOAlgorithms.jl:3) with every complex vector stored as interleaved
 * (re, im) pairs of R, i.e. reinterpret(R, ::Vector{Complex{R}}): d counts reals (even), b holds N pairs. */
enum { ORC_LOSS_LS = 0, ORC_LOSS_LOGISTIC = 1, ORC_LOSS_ZERO = 2, ORC_LOSS_LS_COMPLEX = 3 };
enum { ORC_PROX_ZERO = 0, ORC_PROX_L1 = 1, ORC_PROX_BOX = 2, ORC_PROX_L1_COMPLEX = 3 };

/* F = [f_1..f_N] packed: row-major A (N x d), b (N) = targets (LS) or labels (logistic), lam = LeastSquares λ. */
typedef struct {
    int32_t loss;
    int32_t _pad;
    int64_t N, d;
    const void *A;
    const void *b;
    double lam;
} orc_problem;

/* g */
typedef struct {
    int32_t kind;
    int32_t _pad;
    double lam;          /* NormL1(lam)                     */
    double lo, hi;       /* IndBox scalar bounds            */
    const void *lo_vec;  /* IndBox per-coordinate (or NULL) */
    const void *hi_vec;
} orc_prox_desc;

#define ORC_DECL(R, S)                                                                                          \
    R orc_gradient_##S(int loss, int64_t d, const R *a, const R *bp, R lam, const R *x, R *y);                 \
    void orc_prox_##S(const orc_prox_desc *g, int64_t d, const R *x, R gamma, R *y);                           \
    void orc_full_pass_##S(const orc_problem *p, const R *x, R *av, R *tmp);                                   \
    void orc_shard_pass_##S(const orc_problem *p, int64_t rows, const R *x, R *av, R *tmp);                    \
    void orc_svrg_init_##S(const orc_problem *p, const R *x0, R *av, R *z, R *z_full, R *w);                   \
    void orc_svrg_inner_##S(const orc_problem *p, const orc_prox_desc *g, R gamma, int64_t m,                  \
                            const int64_t *idx, const R *av, R *z, const R *z_full, R *w);                     \
    void orc_svrg_iterate_##S(const orc_problem *p, const orc_prox_desc *g, R gamma, int64_t m,                \
                              const int64_t *idx, int plus, R *av, R *z, R *z_full, R *w);                     \
    void orc_saga_init_##S(const orc_problem *p, const orc_prox_desc *g, R gamma, const R *x0, R *table,       \
                           R *av, R *z);                                                                       \
    void orc_saga_steps_##S(const orc_problem *p, const orc_prox_desc *g, R gamma, int sag, int64_t nsteps,    \
                            const int64_t *idx, R *table, R *av, R *z);                                        \
    void orc_finito_init_##S(const orc_problem *p, const orc_prox_desc *g, const R *gam, const R *x0,          \
                             R *table, R *av, R *z, R *hat_gamma);                                             \
    void orc_finito_steps_##S(const orc_problem *p, const orc_prox_desc *g, const R *gam, R hat_gamma,         \
                              int64_t nit, const int64_t *bptr, const int64_t *bidx, R *table, R *av, R *z);   \
    void orc_lfinito_init_##S(const orc_problem *p, const R *gam, const R *x0, R *av, R *z, R *z_full,         \
                              R *hat_gamma);                                                                   \
    void orc_lfinito_iterate_##S(const orc_problem *p, const orc_prox_desc *g, const R *gam, R hat_gamma,      \
                                 int64_t nb, const int64_t *bptr, const int64_t *bidx, R *av, R *z,            \
                                 R *z_full);                                                                   \
    int orc_afinito_init_##S(const orc_problem *p, const orc_prox_desc *g, R alpha, const R *x0, R *table,      \
                             R *gtable, R *gam, R *fi_x, R *av, R *z, R *hat_gamma, const R *retry_signs,     \
                             int64_t *nretry);                                                                 \
    int64_t orc_afinito_steps_##S(const orc_problem *p, const orc_prox_desc *g, R alpha, R tol_b,              \
                                  int64_t nsteps, const int64_t *idx, R *table, R *gtable, R *gam, R *fi_x,   \
                                  R *hat_gamma, R *av, R *z, int64_t *ntrials);                                \
    void orc_proshi_init_##S(int64_t N, int64_t d, int32_t dense, const R *Q, const R *q, R eta, R lo, R hi,                   \
                             const orc_prox_desc *g, const R *gam, const R *x0, R *table, R *av, R *z,          \
                             R *hat_gamma);                                                                    \
    void orc_proshi_steps_##S(int64_t N, int64_t d, int32_t dense, const R *Q, const R *q, R eta, R lo, R hi,                  \
                              const orc_prox_desc *g, const R *gam, R hat_gamma, int64_t nit,                   \
                              const int64_t *bptr, const int64_t *bidx, R *table, R *av, R *z);                 \
    void orc_proshi_solution_##S(int64_t N, int64_t d, const R *gam, const R *z, R *table);                     \
    double orc_objective_##S(const orc_problem *p, const orc_prox_desc *g, const R *x);                        \
    /* Julia's sum(A): Base.mapreduce_impl, pairwise above 1024 elements (the six init sums use these) */      \
    void orc_julia_sum_vec_##S(int64_t N, int64_t d, const R *rows, const R *div, R *out);                     \
    R orc_julia_sum_scalar_##S(int64_t N, const R *x, int inv);

ORC_DECL(double, f64)
ORC_DECL(float, f32)
#undef ORC_DECL

/* Multi-threaded best-case CPU sweep (OpenMP) used ONLY as bench.py's "all host cores" baseline:
 * av = (1/N) sum_i grad f_i(x);  returns the number of threads used. */
int orc_full_pass_omp_f64(const orc_problem *p, const double *x, double *av);
int orc_full_pass_omp_f32(const orc_problem *p, const float *x, float *av);

/* All-cores raw sums of a row shard, for checks at sizes one core cannot sweep in seconds:  acc[k] += sum_i c_i(x) a_ik over
 * the `rows` rows stored in p->A (NOT divided by N; acc is long double and is continued, so slabs can be chained).  Every thread
 * sums a contiguous block of rows into a private d-vector of R; the privates are added into acc in thread order.  Not the
 * reference's summation order -- a high-accuracy value of the same sum; returns the number of threads used. */
int orc_shard_sums_omp_f64(const orc_problem *p, int64_t rows, const double *x, long double *acc);
int orc_shard_sums_omp_f32(const orc_problem *p, int64_t rows, const float *x, long double *acc);

#ifdef __cplusplus
}
#endif
#endif
