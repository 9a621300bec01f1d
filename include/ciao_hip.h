/*
 * ciao_hip.h -- C ABI of libciao_hip.so: the finite-sum hot path of CIAOAlgorithms.jl (SVRG / SAGA+SAG / Finito /
 * LFinito iterate updates) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * This is the drop-in boundary.  The reference (pure Julia) has no FFI of its own; the boundary is inserted exactly
 * where its iterables (L2) call the ProximalOperators.jl plugin API (L1):
 *      gradient!(y, f_i, x)      prox!(y, g, x, gamma)
 * and where the solver functors (L3) call Base.iterate on those iterables.  Each entry point below names the
 * reference lines (relative to /root/reference/src/algorithms/) whose work it performs.  INTEGRATION.md shows the
 * Julia `ccall` stubs that bind these symbols.
 *
 * Conventions
 *   - extern "C"; plain pointers and sizes; no C++/torch types; no exceptions cross the ABI.
 *   - every call returns int32 status: 0 = ok, negative = error (ciao_last_error() has the text).
 *   - all vectors/matrices are DEVICE pointers unless the parameter name ends in `_host`.  The caller owns every
 *     buffer (AMDGPU.jl ROCArray / torch tensor); the library never frees or reallocates them and only owns the
 *     private workspace inside ciao_ctx.
 *   - work is enqueued on the ctx's HIP stream; calls are asynchronous unless they return host data.
 *   - one ctx = one device + one stream; not thread-safe (the reference is single-threaded with a global RNG).
 *   - sample indices are int64, 0-BASED (the Julia wrapper subtracts 1).  The reference's RNG draws are an INPUT
 *     (index arrays), because Julia's stream is not reproducible outside Julia.
 *   - real type `R` of the reference = `dtype` here (f32 or f64); nothing is silently promoted
 *     (test/test_lasso.jl:74 asserts eltype).  Scalars cross the ABI as double and are rounded to R once.
 *   - data layout: A is row-major N x d with row stride `ld` elements (a Julia d x N column-major Matrix is exactly
 *     this with ld = d); the SAGA/Finito table is row-major N x d, stride d.
 *   - multi-GPU: rows are sharded; every entry point that sums over samples calls the ctx's all-reduce hook (if set)
 *     on the raw d-vector sum before its epilogue (see ciao_ctx_set_allreduce).
 */
#ifndef CIAO_HIP_H
#define CIAO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIAO_ABI_VERSION 3   /* 3: ciao_shard_table grew by meta[8] (round 4 changed the layout under version 2: ADVICE r4) */

#if defined(__GNUC__)
#define CIAO_API __attribute__((visibility("default")))
#else
#define CIAO_API
#endif

enum {
    CIAO_OK = 0,
    CIAO_ERR_ARG = -1,         /* bad argument (null pointer, negative size, misaligned, index out of range) */
    CIAO_ERR_HIP = -2,         /* a HIP runtime call failed                                                */
    CIAO_ERR_UNSUPPORTED = -3, /* shape / operator family outside what the kernels cover                   */
    CIAO_ERR_ALLOC = -4,       /* workspace allocation failed                                              */
    CIAO_ERR_HOOK = -5         /* the all-reduce hook returned non-zero                                    */
};

enum { CIAO_F32 = 0, CIAO_F64 = 1 };

/* f_i families */
enum {
    CIAO_LOSS_LS = 0,       /* LeastSquares(a_i' (1 x d), [b_i], lam): f_i(x) = lam/2 (a_i'x - b_i)^2   test/test_lasso.jl:54 */
    CIAO_LOSS_LOGISTIC = 1, /* Precompose(LogisticLoss([y_i],1), a_i', 1): log(1+exp(-y_i a_i'x))        test/test_logistic_l1.jl:36 */
    CIAO_LOSS_ZERO = 2,     /* Zero(): the default F = fill(Zero(), N)                                   SVRG/SVRG.jl:58 */
    /* Complex T (src/CIAOAlgorithms.jl:3 RealOrComplex; test/test_lasso.jl:3 runs ComplexF32/64): LeastSquares with a complex
     * row and target, f_i(x) = lam/2 |a_i.x - b_i|^2, grad = lam conj(a_i) (a_i.x - b_i).  Every complex vector is stored as
     * interleaved (re, im) pairs of the REAL dtype -- what reinterpret(R, ::Vector{Complex{R}}) gives: d and ld count reals
     * (d = 2 x the complex length, even), A is N x ld, b holds N pairs, tables N x d.  Correctness path: one wave per row, no
     * tuning. */
    CIAO_LOSS_LS_COMPLEX = 3
};

/* g families (rows O3, O4) */
enum {
    CIAO_PROX_ZERO = 0, /* Zero(): prox = identity                       SVRG/SVRG.jl:49             */
    CIAO_PROX_L1 = 1,   /* NormL1(lam): soft threshold at gamma*lam      test/test_lasso.jl:59       */
    CIAO_PROX_BOX = 2,  /* IndBox(lo, hi): clamp, scalar or per-coordinate  test/test_sharing.jl:16   */
    CIAO_PROX_L1_COMPLEX = 3   /* NormL1(lam) on complex coordinates stored as (re, im) pairs: y = sign(x) max(|x| - gamma lam, 0)
                                  with the complex modulus and sign; d must be even */
};

/* Packed F = [f_1 .. f_N] (replaces the reference's Vector of N one-row operator objects, test/test_lasso.jl:50-58).
 * N is the number of rows RESIDENT on this device (the local shard); N_total the global count used for the 1/N
 * factors (== N on one GPU).  Sample indices, `gam` and the SAGA/Finito tables are indexed by LOCAL row. */
typedef struct {
    int32_t loss;      /* CIAO_LOSS_*                        */
    int32_t dtype;     /* CIAO_F32 / CIAO_F64                */
    int64_t N;         /* local rows                         */
    int64_t d;         /* features                           */
    int64_t ld;        /* row stride of A in elements (>= d) */
    int64_t N_total;   /* global N (for the 1/N factors)     */
    const void *A;     /* device, N x ld                     */
    const void *b;     /* device, N: targets b_i (LS) or labels y_i in {-1,+1} (logistic); NULL for ZERO */
    double lam;        /* LeastSquares lambda (ignored otherwise) */
} ciao_problem;

typedef struct {
    int32_t kind;       /* CIAO_PROX_*                                  */
    int32_t _pad;
    double lam;         /* NormL1 lambda                                */
    double lo, hi;      /* IndBox scalar bounds                         */
    const void *lo_vec; /* device d-vector of lower bounds, or NULL     */
    const void *hi_vec; /* device d-vector of upper bounds, or NULL     */
} ciao_prox_desc;

typedef struct ciao_ctx ciao_ctx; /* opaque: device id, stream, private workspace, all-reduce hook */

/* All-reduce hook: sum `count` elements of dtype at device pointer `buf` IN PLACE across all ranks, enqueued on
 * `stream` (a hipStream_t).  Return 0 on success.  With RCCL: ncclAllReduce(buf, buf, count, type, ncclSum, comm,
 * stream).  The Python host passes a closure around torch.distributed.all_reduce. */
typedef int32_t (*ciao_allreduce_fn)(void *user, void *buf, int64_t count, int32_t dtype, void *stream);

/* ---- library / context ------------------------------------------------------------------------------------- */
CIAO_API int32_t ciao_abi_version(void);
CIAO_API const char *ciao_last_error(void);
/* The extra compiler flags this library was built with: "" for the product build.  Experiment builds (timing macros, some
 * of which give wrong results; tools/exp_build.sh) report theirs here and are never written over the product library. */
CIAO_API const char *ciao_build_flags(void);
/* device >= 0; stream = hipStream_t to enqueue on (NULL = the device's default stream). */
CIAO_API int32_t ciao_ctx_create(int32_t device, void *stream, ciao_ctx **out);
CIAO_API int32_t ciao_ctx_destroy(ciao_ctx *ctx);
CIAO_API int32_t ciao_ctx_set_stream(ciao_ctx *ctx, void *stream);
CIAO_API int32_t ciao_ctx_synchronize(ciao_ctx *ctx);
CIAO_API int32_t ciao_ctx_set_allreduce(ciao_ctx *ctx, ciao_allreduce_fn fn, void *user); /* fn = NULL: single GPU */
/* The same without a host callback: `comm` is an ncclComm_t of RCCL spanning the ranks that share the rows; the library
 * then calls ncclAllReduce(buf, buf, count, ncclFloat|ncclDouble, ncclSum, comm, stream) itself wherever the hook would
 * be called.  RCCL is resolved at run time from `librccl_path` (NULL = "librccl.so"), so single-GPU hosts need no RCCL;
 * pass the library the communicator was created with.  comm = NULL removes it.  Replaces any hook set before. */
CIAO_API int32_t ciao_ctx_set_rccl(ciao_ctx *ctx, void *comm, const char *librccl_path);
/* Objective monitor (the step after the path; the reference has stop(state) = false, SVRG/SVRG.jl:55,70,
 * and computes the cost outside, test/test_lasso.jl:45-47).  obj_dev = device double[3], or NULL to switch it off.  While
 * set, every full pass over the rows -- ciao_full_gradient, ciao_proxgrad_step, ciao_svrg_init, the tail pass of
 * ciao_svrg_iterate (at the new z_full = the solution), ciao_lfinito_init, the full pass of ciao_lfinito_iterate (at
 * z_full) -- also leaves, for the point x the pass was taken at,
 *     obj_dev[1] = (1/N_total) sum_i f_i(x)    obj_dev[2] = g(x)    obj_dev[0] = obj_dev[1] + obj_dev[2]
 * where the f_i(x) are the values `gradient!` returns and the reference discards: they ride on the same sweep (no extra
 * pass over A; all-reduced with the d-vector on a row-sharded problem).  g (may be NULL = Zero) is copied. */
CIAO_API int32_t ciao_ctx_set_monitor(ciao_ctx *ctx, const ciao_prox_desc *g, double *obj_dev);
/* Tuning knobs (performance only, never results-changing beyond summation order); returns CIAO_ERR_ARG for unknown keys:
 *   "sweep_blocks_per_cu", "sweep_grid", "sweep_prefetch", "sweep_multi", "force_generic"   grid / variant of the rows kernels
 *   "chain_max_batch"      Finito / LFinito batches up to this size run as one sequential chain (-1 = measured crossover)
 *   "split_max_rows"       batches up to this size run one workgroup per row instead of one wave per row (-1 = 16384)
 *   "split_blocks_per_cu"  grid cap of that kernel (0 = automatic)
 *   "split_all"            experiment: that kernel for every mode and size (tools/tune_split.py)
 *   "small_i"              rows_small_kernel (rows under 1 KiB): elements per lane and iteration, 8 or 16 (0 = automatic)
 *   "small_mfma"           0: dense rows of 17 .. 256 elements do not take the matrix-core tile kernel (rows_smallm_kernel)
 *   "small_mfma_table"     0: ... its table modes (SAGA / Finito init, Finito batches over row blocks) are off, the sweeps stay
 *   "small_nb"             ... tile buffers per wave, 2 .. 4 (0 = automatic: 2)
 *   "small_wrow"           0: Finito / LFinito batches on rows of up to 256 elements do not take the one-wave-per-row kernel
 *                          (rows_wrow_kernel: index lists, and row blocks of up to 8192 rows) but rows_smallb_kernel / the tiles
 *   "wrow_rows_per_wave"   rows_wrow_kernel: rows a wave takes in a small batch before more workgroups are used (0 = 4)
 *   "multi_rhs_off"        1: ciao_full_gradient_multi / ciao_svrg_epoch_tail_multi run K single sweeps (testing)
 *   "peer_timeout_s"       wall-clock bound of a peer-mailbox wait, seconds (default 30)
 *   "chain_no_dma"         the register-ring chain instead of the LDS-DMA one;  "chain_big": the any-length kernels at any d
 *   "chain_four_waves"     rows of up to 2 KiB (fp64: 4 KiB) on the four-wave chain instead of the single-wave one
 *   "proshi_chain_max_batch"  ProShI batches up to this size run as one coordinate-parallel chain launch (-1 = measured crossover)
 *   "chain_no_ws"          SAGA / SAG chains on chain_dma_kernel instead of the wave-specialised chain_ws_kernel
 *   "chain_no_wide"        SVRG / SAGA chains on rows beyond 8192 elements on the one-workgroup kernel instead of chain_wide_kernel
 *   "chain_ws_issuers"     issuer waves of chain_ws_kernel, 1 or 2 (0 = automatic)
 *   "long_rows"            0: sweeps / batches on rows beyond 64 KiB on the generic kernel instead of rows_long_kernel (a cluster of
 *                          workgroups per row);  "long_j": its 16-byte chunks per thread, 4 or 8 (0 = automatic)
 *   "svrg_cache_rowdots"   0: the SVRG full pass does not store a_i'z_full at all (see ciao_svrg_iterate) */
CIAO_API int32_t ciao_ctx_set_option(ciao_ctx *ctx, const char *key, int64_t value);
/* Kernel timing for bench.py's roofline line: when enabled, every launch of the dominant streaming kernel of an entry
 * point (rows_fast_kernel / rows_generic_kernel) is bracketed by HIP events on the ctx's stream.  _read synchronises,
 * returns the summed device time of those launches and their count since the last read, and resets the counters. */
CIAO_API int32_t ciao_ctx_timing_enable(ciao_ctx *ctx, int32_t enable);
CIAO_API int32_t ciao_ctx_timing_read(ciao_ctx *ctx, double *total_ms_host, int64_t *launches_host);
/* Name + grid of the dominant kernel of the last entry point called (for bench.py / rocprof correlation). */
CIAO_API const char *ciao_ctx_last_kernel(ciao_ctx *ctx);

/* ---- L1 plugin API (ProximalOperators.jl calling convention) ---------------------------------------------- */
/* gradient!(y, F[i], x): y = grad f_i(x); if fval != NULL also *fval = f_i(x) (device scalar of dtype).
 * Call sites: SVRG/SVRG_basic.jl:60,74,75,89; SAGA_SAG/SAGA_basic.jl:43,56; Finito/Finito_basic.jl:78,112. */
CIAO_API int32_t ciao_gradient(ciao_ctx *ctx, const ciao_problem *p, int64_t i, const void *x, void *y, void *fval);
/* prox!(y, g, x, gamma): y = prox_{gamma g}(x); y may alias x.
 * Call sites: SVRG_basic.jl:80; SAGA_basic.jl:48,64; Finito_basic.jl:84,118; Finito_LFinito.jl:83,92. */
CIAO_API int32_t ciao_prox(ciao_ctx *ctx, int32_t dtype, int64_t d, const ciao_prox_desc *g, const void *x, double gamma,
                  void *y);

/* ---- the roofline sweep ------------------------------------------------------------------------------------ */
/* av = (1/N_total) sum_i grad f_i(x)       SVRG_basic.jl:58-63 (init) and :87-92 (epoch tail).
 * Reads every row of A exactly once. */
CIAO_API int32_t ciao_full_gradient(ciao_ctx *ctx, const ciao_problem *p, const void *x, void *av);
/* The same pass for K iterates over the same rows -- av[k] = (1/N_total) sum_i grad f_i(x[k]), k < K -- in ONE pass over A: with
 * K right-hand sides the contraction is GEMM-shaped (A Z, the link function, A' C) and runs on the matrix cores
 * (csrc/mrhs_kernels.h; north_star: "MFMA ... if the f_i gradients are expressed as a dense A.x contraction").  x and av are HOST
 * arrays of K device pointers (16-byte aligned d-vectors).  Not in the reference, which solves one problem per call
 * (SVRG_basic.jl:87-92 is the single pass): an extension for hosts that advance several solves together (a regularisation path,
 * folds: ciao_ctx_chain_batch_begin).  Shapes the kernel does not take (d other than 256 / 512 / 768 / 1024 -- and 128 in fp64 --, unaligned rows, a
 * row-sharded problem, the objective monitor) run as K single passes inside the call.  Each av[k] agrees with its own
 * ciao_full_gradient to rounding (another summation order), not bitwise. */
CIAO_API int32_t ciao_full_gradient_multi(ciao_ctx *ctx, const ciao_problem *p, int32_t K, const void *const *x, void *const *av);
/* Fused proximal-gradient step on the full gradient (the "full-gradient + prox sweep" of the north star; it is also
 * LFinito's `prox!` + full pass, Finito_LFinito.jl:83-88, in x-coordinates):
 *     av = (1/N_total) sum_i grad f_i(x);   y = prox_{gamma g}(x - gamma * av)          (y may alias x) */
CIAO_API int32_t ciao_proxgrad_step(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma,
                           const void *x, void *av, void *y);
/* (1/N_total) sum_i f_i(x) + g(x) on the same sweep (the value `gradient!` returns and the reference discards;
 * test/test_lasso.jl:45 computes it outside).  *obj_host receives the value (synchronises). */
CIAO_API int32_t ciao_objective(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *x,
                       double *obj_host);

/* ---- SVRG / SVRG++  (SVRG/SVRG_basic.jl) -------------------------------------------------------------------- */
/* Base.iterate(iter), :57-66: av = full gradient at x0; z_full = x0; z = 0; w = x0. */
CIAO_API int32_t ciao_svrg_init(ciao_ctx *ctx, const ciao_problem *p, const void *x0, void *av, void *z, void *z_full,
                       void *w);
/* The inner cycle :73-82 for the m draws idx[0..m) (device int64, 0-based LOCAL rows):
 *     temp = gamma*(grad f_i(z_full) - grad f_i(w) - av) + w ; w = prox_{gamma g}(temp) ; z += w.
 * Strictly sequential in w: runs as one persistent workgroup (replicas-only across GPUs). */
CIAO_API int32_t ciao_svrg_inner(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int64_t m,
                        const int64_t *idx, const void *av, void *z, const void *z_full, void *w);
/* Base.iterate(iter, state), :71-96 = inner cycle + tail (z_full = z/m; basic: w = z_full; z = 0) + full pass.
 * (`state.m *= 2` of SVRG++ is host bookkeeping; pass the current m.)
 * The full pass also stores a_i'z_full per row in the ctx workspace (N scalars).  With reuse_rowdots != 0 the CALLER
 * vouches that A and z_full still hold exactly what the previous ciao_svrg_init / ciao_svrg_iterate on this ctx left in
 * them (nothing wrote to either from outside, not even in place); the inner cycle then reads those N scalars instead of
 * recomputing the second dot product of every step (0.39 vs 0.50 us per update at d = 1024 fp64).
 * The library additionally requires that the previous call on this ctx was one of those two (or ciao_svrg_inner) with
 * the same A, z_full and N; anything else recomputes.  With reuse_rowdots == 0 nothing cached is ever read: the safe
 * choice for a host that hands the state vectors to user code between calls.  The host mirrors track this themselves
 * (solvers.py: torch's in-place version counters of z_full, A and b). */
CIAO_API int32_t ciao_svrg_iterate(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int64_t m,
                          const int64_t *idx, int32_t plus, int32_t reuse_rowdots, void *av, void *z, void *z_full, void *w);
/* The second half of ciao_svrg_iterate on its own, :84-92: z_full = z / m; basic (plus == 0): w = z_full; z = 0; then the full
 * pass av = (1/N) sum_i grad f_i(z_full).  ciao_svrg_inner followed by this IS ciao_svrg_iterate with reuse_rowdots == 0, bit for
 * bit; it exists for hosts that run the inner cycles of several solves as one chain batch (ciao_ctx_chain_batch_begin) and
 * finish each solve's outer iteration afterwards. */
CIAO_API int32_t ciao_svrg_epoch_tail(ciao_ctx *ctx, const ciao_problem *p, int64_t m, int32_t plus, void *av, void *z,
                             void *z_full, void *w);
/* ... for K solves over the same rows at once (host arrays of K device pointers): every solve's tail, then the K full passes as
 * ONE pass over A (ciao_full_gradient_multi).  K calls of ciao_svrg_epoch_tail to rounding. */
CIAO_API int32_t ciao_svrg_epoch_tail_multi(ciao_ctx *ctx, const ciao_problem *p, int32_t K, int64_t m, int32_t plus, void *const *av,
                                   void *const *z, void *const *z_full, void *const *w);

/* ---- the sequential chains on a row-sharded problem ("one chain on one GPU pulling remote rows over xGMI") --
 * The inner loops of SVRG and SAGA are one dependent chain and do not shard: on a problem whose rows do not fit one GPU
 * (BASELINE config #4: 80M x 1024) exactly ONE rank, the chain owner, runs the chain and reads the rows of the other shards
 * through peer-mapped pointers (hipIpcOpenMemHandle / peer access: the loads travel over xGMI; all indices are known up front,
 * SVRG_basic.jl:73, so the rows are prefetched through the same LDS-DMA ring as local ones).  Same arithmetic, same order:
 * the chain's results are bitwise those of the unsharded chain on the concatenated rows.
 *   shard k = GLOBAL rows [row0[k], row0[k+1]) (contiguous block partition, row0[0] = 0, row0[nshards] = N_total);
 *   A[k], b[k] (and table[k] for SAGA): device pointers valid on THIS device for shard k's rows (its own shard: the
 *   local pointers); every A[k] has the row stride of the local problem.  Non-owners pass the same nshards / row0 and
 *   may leave the pointers NULL. */
#define CIAO_MAX_SHARDS 8
typedef struct {
    int32_t nshards;
    int32_t owner;                          /* != 0 on the one rank that runs the chains */
    int64_t row0[CIAO_MAX_SHARDS + 1];
    const void *A[CIAO_MAX_SHARDS];
    const void *b[CIAO_MAX_SHARDS];
    void *table[CIAO_MAX_SHARDS];           /* SAGA gradient table / adaptive Finito s-table shards (NULL when only SVRG is run) */
    void *meta[CIAO_MAX_SHARDS];            /* adaptive Finito: the per-sample scalars of the shard's rows (n x 4 x 4; NULL otherwise) */
} ciao_shard_table;
/* Installs (copies) the shard table; NULL removes it.  While one is set AND an all-reduce hook is installed,
 * ciao_svrg_inner / ciao_svrg_iterate / ciao_saga_steps / ciao_afinito_init / ciao_afinito_steps are valid on the row-sharded
 * problem (adaptive Finito: every rank initialises its own rows' table and scalars in one all-reduced sweep; the owner's chain
 * then reads and writes the other shards' table rows and scalars, and av, z, hat_gamma and the two counters reach every rank;
 * rows must be whole 16-byte chunks of at most 32 KiB): `idx` then holds GLOBAL rows,
 * the owner runs the chain, every other rank skips it, and the iterates the chain produced (z, w; for SAGA z, av) reach all
 * ranks through one all-reduce of 2d scalars in which the non-owners contribute zeros; the tail and the full pass of
 * ciao_svrg_iterate then run sharded as usual.  (The a_i'z_full cache is not used across shards.) */
CIAO_API int32_t ciao_ctx_set_shards(ciao_ctx *ctx, const ciao_shard_table *shards);
/* ---- one-shot peer all-reduce (SURVEY.md sections 5 / 8e, the "tuned alternative" to ncclAllReduce; no reference counterpart) --
 * The all-reduce of the d+1 scalars of a sweep / batch is latency, not bandwidth (4-16 KB over 7 direct xGMI links).  With
 * peers set, the kernel that produces a rank's raw sum writes it straight into a slot of EVERY rank's mailbox and releases a
 * flag, and the kernel that applies the epilogue waits for the world's flags and adds the slots in rank order (bitwise the
 * same sum on every rank): no collective call and no extra launch per reduction (csrc/peer_kernels.h).
 *   _mailbox_create: this rank's mailbox, fine-grained device memory for reductions of up to max_elems elements (d + 1, or
 *                    2 d for the sharded chains' hand-over); share it with ciao_ipc_export / _open.  _destroy frees it.
 *   ciao_ctx_set_peers: mailboxes[r] = rank r's mailbox as mapped on THIS device (mailboxes[rank] = its own); world <= 8.  Every
 *                    rank calls it with freshly created mailboxes at the same point of its program and from then on makes
 *                    the same reductions in the same order.  Replaces any all-reduce hook / RCCL communicator; NULL or world 0
 *                    turns it off.  A rank whose flag never arrives is reported by ciao_ctx_synchronize (CIAO_ERR_HOOK). */
CIAO_API int32_t ciao_peer_mailbox_create(ciao_ctx *ctx, int64_t max_elems, void **mailbox_out, int64_t *bytes_out);
CIAO_API int32_t ciao_peer_mailbox_destroy(ciao_ctx *ctx, void *mailbox);
CIAO_API int32_t ciao_ctx_set_peers(ciao_ctx *ctx, int32_t rank, int32_t world, void *const *mailboxes, int64_t max_elems);
/* Sharing a device allocation with the chain owner's process (plain HIP IPC, no torch / AMDGPU.jl needed):
 * _export: handle_out = 64 bytes describing the allocation that contains dev_ptr, *offset_out = dev_ptr's offset in it;
 * _open (in the other process, on the device that will read it): *dev_ptr_out = the mapped pointer + offset;
 * _close: unmaps what _open returned (pass the same dev_ptr and offset). */
CIAO_API int32_t ciao_ipc_export(const void *dev_ptr, void *handle_out, int64_t *offset_out);
CIAO_API int32_t ciao_ipc_open(const void *handle, int64_t offset, void **dev_ptr_out);
CIAO_API int32_t ciao_ipc_close(void *dev_ptr, int64_t offset);

/* ---- SAGA / SAG  (SAGA_SAG/SAGA_basic.jl) ------------------------------------------------------------------- */
/* Base.iterate(iter), :41-48: table[i] = grad f_i(x0); av = sum/N; z = prox_{gamma g}((1-gamma) x0). */
CIAO_API int32_t ciao_saga_init(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma,
                       const void *x0, void *table, void *av, void *z);
/* nsteps consecutive Base.iterate(iter,state), :53-68, with the draws idx[0..nsteps) (device int64). */
CIAO_API int32_t ciao_saga_steps(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double gamma, int32_t sag,
                        int64_t nsteps, const int64_t *idx, void *table, void *av, void *z);

/* ---- several independent chains at once ----------------------------------------------------------------------- */
/* A chain (the SVRG inner cycle SVRG_basic.jl:73-82, the SAGA / SAG steps SAGA_basic.jl:53-68) is one workgroup on one of the
 * GPU's 256 compute units: sequential by definition.  A host that runs SEVERAL solves over the same rows -- a
 * regularisation path (g = NormL1(lambda_k)), cross-validation folds (index streams over different row subsets), restarts --
 * has that many independent chains.  Between _begin and _end the calls of ciao_svrg_inner, ciao_saga_steps and ciao_finito_steps
 * (index-list form, every iteration's batch of the same size, small enough to run as a sequential chain: option "chain_max_batch";
 * the default Finito has batches of one sample, Finito.jl:50) on this ctx are
 * RECORDED instead of launched (every other entry point returns CIAO_ERR_ARG meanwhile); _end(launch = 1) checks that no chain
 * writes state another one touches (SVRG writes w, z; SAGA and Finito z, av and their table; CIAO_ERR_ARG otherwise), and launches them as one
 * grid per kernel variant with one workgroup per chain, on the ctx's stream; _end(launch = 0) drops the records.  Each chain's
 * results are bitwise those of the same call made alone.  Takes the chains of the LDS-DMA kernels: real scalars, rows of whole
 * 16-byte chunks of at most 32 KiB, 16-byte aligned, unsharded (CIAO_ERR_UNSUPPORTED from the recording call otherwise: the batch
 * stays open and that call is not part of it).  No counterpart in the reference, which solves one problem per call. */
CIAO_API int32_t ciao_ctx_chain_batch_begin(ciao_ctx *ctx);
CIAO_API int32_t ciao_ctx_chain_batch_end(ciao_ctx *ctx, int32_t launch);

/* ---- Finito / MISO  (Finito/Finito_basic.jl) ---------------------------------------------------------------- */
/* hat_gamma = 1 / sum_i (1/gam_i)  over the LOCAL rows then all-reduced (Finito_basic.jl:82). Synchronises. */
CIAO_API int32_t ciao_hat_gamma(ciao_ctx *ctx, int32_t dtype, int64_t N, const void *gam, double *hat_gamma_host);
/* Base.iterate(iter), :76-84: table[i] = x0 - (gam_i/N) grad f_i(x0); av = hat_gamma sum_i table[i]/gam_i;
 * z = prox_{hat_gamma g}(av).  gam: device N-vector of per-sample stepsizes (built by the host from :61-74). */
CIAO_API int32_t ciao_finito_init(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam,
                         double hat_gamma, const void *x0, void *table, void *av, void *z);
/* nit consecutive Base.iterate(iter,state), :109-118.  Iteration t uses the samples
 * bidx[bptr_host[t] .. bptr_host[t+1]) (bidx: device int64, LOCAL rows; bptr_host: HOST int64[nit+1]); the batch
 * choice (:95-108) is host logic.  Small batches run as one persistent chain; large ones batch-parallel.
 * The samples of one batch must be distinct, as every batch the reference can build is (sample(1:N, r, replace=false),
 * :97, or a static block, :49-59): the members of a batch are updated concurrently. */
CIAO_API int32_t ciao_finito_steps(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam,
                          double hat_gamma, int64_t nit, const int64_t *bptr_host, const int64_t *bidx,
                          void *table, void *av, void *z);

/* The same for batches that are contiguous blocks of LOCAL rows -- every static batch of sweeping 2 and 3 is one
 * (Finito_basic.jl:52-58), on a row-sharded problem too (the local members of a contiguous global block are a contiguous
 * local block under both ownership rules of the host mirror): iteration t updates rows first_host[t] .. first_host[t] +
 * len_host[t] - 1 (HOST int64[nit] each).  No index array exists: nothing is built or uploaded per batch (8 bytes per
 * sample and iteration otherwise -- for LFinito a whole N-index array per iteration) and a sharded host needs no per-batch
 * membership test.  The kernels are the same and so is their speed (measured: 12.0 vs 12.0 us per batch at r = 256,
 * 45.1 vs 46.2 at r = 4096, d = 4096 fp32); results are bitwise those of ciao_finito_steps on arange(first, first+len). */
CIAO_API int32_t ciao_finito_steps_blocks(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam,
                                 double hat_gamma, int64_t nit, const int64_t *first_host, const int64_t *len_host,
                                 void *table, void *av, void *z);

/* ---- LFinito  (Finito/Finito_LFinito.jl) -------------------------------------------------------------------- */
/* Base.iterate(iter), :66-72: av = x0 - (hat_gamma/N) sum_i grad f_i(x0); z = z_full = av (state ctor :27-37). */
CIAO_API int32_t ciao_lfinito_init(ciao_ctx *ctx, const ciao_problem *p, double hat_gamma, const void *x0, void *av,
                          void *z, void *z_full);
/* One Base.iterate(iter,state), :82-100: z_full = prox(av); av = z_full - (hat_gamma/N) sum grad f_i(z_full);
 * then for each of the nb batches, in the order given: z = prox(av); av += sum_{i in batch} [ (hat_gamma/N)
 * (grad f_i(z_full) - grad f_i(z)) + (hat_gamma/gam_i)(z - z_full) ]. */
CIAO_API int32_t ciao_lfinito_iterate(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam,
                             double hat_gamma, int64_t nb, const int64_t *bptr_host, const int64_t *bidx,
                             void *av, void *z, void *z_full);

/* The same with the batches given as contiguous row blocks (LFinito's batches are ALWAYS the static blocks of
 * Finito_LFinito.jl:44-49; sweeping 3 only permutes their order, :89): see ciao_finito_steps_blocks. */
CIAO_API int32_t ciao_lfinito_iterate_blocks(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, const void *gam,
                                    double hat_gamma, int64_t nb, const int64_t *first_host, const int64_t *len_host,
                                    void *av, void *z, void *z_full);

/* ---- adaptive Finito  (Finito/Finito_adaptive.jl) -------------------------------- */
/* For the row-structured f_i here grad f_i = c_i a_i, so the reference's N x d gradient table is N scalars: `meta` is a
 * device N x 4 x 4 array of R: per sample FOUR identical copies of {c_i, f_i(x_i), gamma_i, a_i'x_i} (one per wave of
 * the step kernel, which makes every read follow its write in one wave's program order); `table` is the N x d table of points
 * x_i; `hat_gamma_dev` is a DEVICE scalar of R (it changes during backtracking).
 * Base.iterate(iter), :59-98: x_i = x0; gamma_i = alpha / L_i with L_i = ||grad f_i(x0 .+ 1) - grad f_i(x0)|| / (sqrt(d) N);
 * hat_gamma = 1/sum 1/gamma_i; av = hat_gamma (sum x_i/gamma_i - sum grad f_i / N); z = prox_{hat_gamma g}(av).
 * The reference's random re-probe (:78-85: when grad f_i(x0 .+ 1) == grad f_i(x0), i.e. for a row whose entries sum to zero,
 * it probes at x0 .+ rand(t*[-1,1]) with t = 1, 2, 4, ...) needs RNG draws, which are a HOST input on this path: such
 * samples get gamma_i = -1 in meta and the next ciao_ctx_synchronize returns CIAO_ERR_UNSUPPORTED.  The host then resolves
 * them in increasing i with ciao_afinito_probe and its own +-1 draws (gamma_i = alpha / (nmg / (t sqrt(d)) / N) with the t
 * AFTER its doubling, as the reference has it, :82-88) and repeats this call with `gam_override`: a device N-vector of R
 * whose entries > 0 are taken as gamma_i (the others are computed as before); NULL = none. */
CIAO_API int32_t ciao_afinito_init(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double alpha,
                                   const void *x0, void *table, void *meta, void *av, void *z, void *hat_gamma_dev,
                                   const void *gam_override);
/* One retry of that probe for sample i: *nmg_host = || grad f_i(x0 + t*signs) - grad f_i(x0) || (signs: device d-vector of
 * +-1 in R; for CIAO_LOSS_LS_COMPLEX d/2 real +-1 entries, added to the real parts as rand(t*[-1,1], size(x0)) does, and
 * sqrt(length(x0)) in the host's formula counts the d/2 complex entries).  Synchronises. */
CIAO_API int32_t ciao_afinito_probe(ciao_ctx *ctx, const ciao_problem *p, int64_t i, const void *x0, const void *signs, double t,
                                    double *nmg_host);
/* nsteps consecutive Base.iterate(iter,state), :120-152, for the samples idx[0..nsteps) (the selection :105-118 is host
 * logic).  Synchronises; *done_host = steps completed (< nsteps iff a stepsize fell below tol_b/N: the reference then
 * warns and ends the iteration, :123-126), *trials_host = backtracking trials taken.  Julia's promotions are kept for
 * R = Float32: the model value `0.5 * iter.N * iter.α / γ ...` and the comparison are Float64, `γ *= 0.8` is a Float64
 * product rounded back (:128-136).  Rows of any length and complex T run afinito_big_kernel (state in the caller's vectors);
 * meta for complex T: copies 0/2 hold {Re c, f_i, gamma_i, Re a_i.x_i}, copies 1/3 the imaginary parts (c = lam res). */
CIAO_API int32_t ciao_afinito_steps(ciao_ctx *ctx, const ciao_problem *p, const ciao_prox_desc *g, double alpha,
                                    double tol_b, int64_t nsteps, const int64_t *idx, void *table, void *meta, void *av,
                                    void *z, void *hat_gamma_dev, int64_t *done_host, int64_t *trials_host);

/* ---- ProShI  (ProShI/ProShI_basic.jl) ------------------------------------------------ */
/* minimize (1/N) sum_i f_i(x_i) + g(sum_i x_i): the solution is the whole N x d table.  Operator family of the reference's
 * own test (test/test_sharing.jl:16-25): f_i = Sum(Quadratic(Q_i, q_i), SqrDistL2(IndBox(lo, hi), eta)).
 *   dense = 0: Q_i = diagm(Q[i,:]) as that test builds it; grad f_i(x)_k = Q_ik x_k + q_ik + eta (x_k - clamp(x_k, lo, hi)),
 *              element-wise in each agent's own row.
 *   dense = 1: a full d x d matrix per agent (ProShI_basic.jl:113 calls the operator's generic gradient!, which for Quadratic
 *              is Q_i x + q_i): Q holds N row-major blocks, row k of agent i at Q + (i*d + k)*ld; needs d*sizeof(T) <= 64 KiB. */
typedef struct {
    int32_t dtype;     /* CIAO_F32 / CIAO_F64                                  */
    int32_t dense;     /* 0: Q holds the N diagonals; 1: N dense d x d blocks  */
    int64_t N;         /* local agents (rows)                                  */
    int64_t d;
    int64_t ld;        /* row stride of Q and q in elements (>= d)             */
    int64_t N_total;   /* global number of agents (the 1/N factor)             */
    const void *Q;     /* device N x ld diagonals, or (dense) N*d x ld rows    */
    const void *q;     /* device N x ld: their linear terms                    */
    double eta, lo, hi; /* SqrDistL2(IndBox(lo, hi), eta); eta = 0 drops it    */
} ciao_sepquad;
/* Base.iterate(iter), :76-87: table_i = x0 - (gam_i/N) grad f_i(x0); hat_gamma = sum gam_i (written to the device scalar
 * hat_gamma_dev); av = sum_i table_i; z = (prox_{hat_gamma g}(av) - av) / hat_gamma. */
CIAO_API int32_t ciao_proshi_init(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam,
                                  const void *x0, void *table, void *av, void *z, void *hat_gamma_dev);
/* nit consecutive Base.iterate(iter,state), :109-121; iteration t updates the agents bidx[bptr_host[t]..bptr_host[t+1])
 * (device int64, LOCAL rows): av -= s_i; s_i += gam_i z; s_i -= (gam_i/N) grad f_i(s_i); av += s_i; then the z update. */
CIAO_API int32_t ciao_proshi_steps(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam,
                                   double hat_gamma, int64_t nit, const int64_t *bptr_host, const int64_t *bidx,
                                   void *table, void *av, void *z);
/* The same for batches that are contiguous blocks of local agents (sweeping 2 and 3, ProShI_basic.jl:50-57). */
CIAO_API int32_t ciao_proshi_steps_blocks(ciao_ctx *ctx, const ciao_sepquad *f, const ciao_prox_desc *g, const void *gam,
                                 double hat_gamma, int64_t nit, const int64_t *first_host, const int64_t *len_host,
                                 void *table, void *av, void *z);
/* solution(state), :127-132: table_i += gam_i z for every agent, IN PLACE (as the reference does). */
CIAO_API int32_t ciao_proshi_solution(ciao_ctx *ctx, const ciao_sepquad *f, const void *gam, const void *z, void *table);

/* ==================================================================================================================
 * Everything above is the drop-in surface: the entry points a reference-side binding needs for this path.
 * The declarations below are NOT part of it.  They are helpers the benchmark, the tests and the Python host mirror use
 * (synthetic data generated on the device; the host-side batch sampler of the injected index stream) and are only
 * visible with CIAO_BENCH_API defined.
 * ================================================================================================================== */
#ifdef CIAO_BENCH_API
/* ---- synthetic data (bench / tests): counter-based generator, reproducible per (seed, row, col) ------------- */
/* out[i*ld + k] = scale * N(0,1) for rows row0 .. row0+nrows, keyed by the GLOBAL (row, col). */
CIAO_API int32_t ciao_synth_normal(ciao_ctx *ctx, int32_t dtype, void *out, int64_t nrows, int64_t d, int64_t ld,
                          int64_t row0, uint64_t seed, double scale);
/* b = A x_true + noise*N(0,1) (LS targets) or sign(...) (logistic labels when `labels` != 0). */
CIAO_API int32_t ciao_synth_targets(ciao_ctx *ctx, const ciao_problem *p, const void *x_true, double noise, int32_t labels,
                           int64_t row0, uint64_t seed, void *b_out);

/* ---- host-side helper: the batch draws of Finito / ProShI with sweeping = 1 ------------------------------------ */
/* n consecutive `sample(1:N, r, replace=false)` (Finito_basic.jl:97, ProShI_basic.jl:97) from the injected
 * counter-based index stream (splitmix64(seed), outputs pos, pos+1, ...: DESIGN.md "index streams"; Julia's own RNG is
 * not reproducible outside Julia), written 0-based to the HOST array out_host[n*r]; *pos_out_host = the stream
 * position after the draws.  Rule per batch: draw as many uniform candidates as are still missing, append, drop
 * repeats keeping first occurrences, until r distinct indices are held.  Needs 2r <= N (the wrapper shuffles
 * otherwise).  Pure host code, no ctx: it exists because a device batch takes 10-50 us and an interpreted host cannot
 * draw batches at that rate. */
CIAO_API int32_t ciao_sample_batches(uint64_t seed, uint64_t pos, int64_t N, int64_t r, int64_t n, int64_t *out_host,
                            uint64_t *pos_out_host);
/* m i.i.d. uniform draws from 0..N-1 of the same stream (outputs pos .. pos+m-1; the rule of `rand(state.ind, m)`,
 * SVRG_basic.jl:73, and `rand(1:N)`, SAGA_basic.jl:55, on the injected stream: ((splitmix64 >> 32) * N) >> 32), written to the
 * DEVICE array out_dev[m] on the ctx's stream: an epoch's 10^7 indices cost the host a second and 80 MB of PCIe, the device
 * microseconds.  Needs 0 < N < 2^32. */
CIAO_API int32_t ciao_sample_uniform(ciao_ctx *ctx, uint64_t seed, uint64_t pos, int64_t N, int64_t m, int64_t *out_dev);
/* The peer all-reduce (ciao_ctx_set_peers) on a device buffer of its own, as two kernels (send, wait + add): what bench.py
 * times as allreduce_us_per_step for that collective, and what tests hold against a host-staged sum. */
CIAO_API int32_t ciao_peer_allreduce(ciao_ctx *ctx, int32_t dtype, int64_t count, void *buf);

#endif /* CIAO_BENCH_API */

#ifdef __cplusplus
}
#endif
#endif /* CIAO_HIP_H */
