// launch.h -- host dispatch entry points shared by api.hip and the per-type instantiation units.
#pragma once

#include "chain_kernels.h"
#include "ciao_ctx.h"
#include "proshi_kernels.h"
#include "rows_kernels.h"

namespace ciao {

// the next reduction's view of the peer mailboxes (advances the sequence number: every rank makes the same reductions in the
// same order)
inline PeerDev peer_dev(ciao_ctx *ctx)
{
    PeerDev p{};
    p.world = ctx->peer_world;
    p.rank = ctx->peer_rank;
    p.seq = ++ctx->peer_seq;
    p.slot_bytes = ctx->peer_slot_bytes;
    for (int r = 0; r < PEER_MAX; ++r) p.mail[r] = ctx->peer_mail[r];
    p.counter = ctx->peer_counter;
    p.errflag = ctx->errflag;
    // the chain owner's broadcast sets peer_wait_s_once for the one reduction that has to outwait a whole chain kernel
    const int64_t wait_s = ctx->peer_wait_s_once > ctx->peer_timeout_s ? ctx->peer_wait_s_once : ctx->peer_timeout_s;
    p.wait_ticks = (unsigned long long)wait_s * PEER_TICKS_PER_S;
    return p;
}

// An open chain batch (ciao_ctx_chain_batch_begin): the launch of a chain kernel is RECORDED -- kernel, block, LDS, argument block,
// the byte ranges of its state and which of them it writes -- and made by ciao_ctx_chain_batch_end, one launch per kernel with one
// workgroup per chain.  Only the chains without a Finito batch structure are taken (SVRG inner cycles, SAGA / SAG steps).
template <typename T, int ALG>
inline int32_t chain_batch_record(ciao_ctx *ctx, const void *kern, unsigned block, size_t lds, const ChainArgs<T> &a)
{
    constexpr bool svrg = (ALG == CA_SVRG || ALG == CA_SVRGC);
    if (!(svrg || ALG == CA_SAGA || ALG == CA_FINITO) || a.nshards > 0) {
        set_error("a chain batch takes SVRG inner cycles, SAGA / SAG steps and small-batch Finito steps of unsharded problems only");
        return CIAO_ERR_UNSUPPORTED;
    }
    ciao_chain_rec r;
    r.kern = kern;
    r.block = block;
    r.lds = lds;
    r.args.assign(reinterpret_cast<const unsigned char *>(&a), reinterpret_cast<const unsigned char *>(&a) + sizeof a);
    const size_t vb = (size_t)a.d * sizeof(T);
    auto range = [&](int k, const void *p, size_t bytes, bool w) {
        r.lo[k] = static_cast<const unsigned char *>(p);
        r.hi[k] = p ? r.lo[k] + bytes : r.lo[k];
        r.wr[k] = w;
    };
    range(0, a.av, vb, !svrg);
    range(1, a.z, vb, true);
    range(2, a.zf, vb, false);
    range(3, a.w, vb, svrg);
    range(4, svrg ? nullptr : a.table, (size_t)a.N * vb, true);
    ctx->batch.push_back(std::move(r));
    return CIAO_OK;
}

// rows kernel + finalize (+ all-reduce hook) + epilogue.  Specialised in rows_f32.hip / rows_f64.hip.
template <typename T>
int32_t launch_rows(ciao_ctx *ctx, int mode, RowsArgs<T> &a, const Epilogue<T> &ep);
// same, but leaves the reduced sum in ctx->sumbuf[0..d) and the extra scalar in ctx->sumbuf[d]; no epilogue.
template <typename T>
int32_t launch_rows_raw(ciao_ctx *ctx, int mode, RowsArgs<T> &a);
// persistent single-workgroup chain.  Specialised in chain_f32.hip / chain_f64.hip.
template <typename T>
int32_t launch_chain(ciao_ctx *ctx, int alg, ChainArgs<T> &a);

// the LDS-DMA fast chain for one (algorithm, loss); J256 = row bytes / 4096 rounded up to a power of two.  Defined in
// chain_dma_launch.inc, instantiated in chain_dma{0..4}_f32/f64.hip.
template <typename T, int ALG, int LOSS>
int32_t launch_dma(ciao_ctx *ctx, int J256, bool masked, ChainArgs<T> &a);

// the wave-specialised chain (chain_ws_kernels.h: consumer / stager / issuer waves, barrier-free exchange) for SAGA / SAG on rows
// of up to 4 KiB.  Defined in chain_ws_launch.inc, instantiated in chain_ws1_f32/f64.hip.
template <typename T, int ALG, int LOSS>
int32_t launch_ws(ciao_ctx *ctx, int J256, bool masked, ChainArgs<T> &a);

// complex chains on the LDS-DMA ring (chain_cdma_kernel): rows of whole 16-byte chunks up to 16 KiB.  Specialised in
// chain_cdma_f32.hip / chain_cdma_f64.hip.
template <typename T>
int32_t launch_cdma(ciao_ctx *ctx, int alg, int J, bool masked, ChainArgs<T> &a);

// the full-gradient pass for K iterates in one pass over A (mrhs_kernels.h: MFMA).  Specialised in mrhs_f32.hip / mrhs_f64.hip.
template <typename T>
bool mrhs_supported(const ciao_ctx *ctx, const ciao_problem *p, int K, const void *const *x, void *const *av);
template <typename T>
int32_t launch_mrhs(ciao_ctx *ctx, const ciao_problem *p, int K, const void *const *x, void *const *av);

// full-gradient sweeps over rows of 17 .. 256 elements on the matrix cores (rowsm_kernels.h).  Specialised in rowsm_f32.hip / rowsm_f64.hip.
template <typename T>
size_t smallm_lds(int64_t d, int nb, bool table_in);   // table_in: RM_FINITO_BATCH (a third tile stream: the old table rows)
template <typename T>
int32_t launch_smallm(ciao_ctx *ctx, int mode, int grid, size_t lds, RowsArgs<T> &a);

// Finito batches over an index list (or a row block) of rows of tabular size, one wave per row (rowsw_kernels.h).  Specialised in
// rowsw_f32.hip / rowsw_f64.hip.  wrow_plan: false when the shape / mode is not for this kernel.
template <typename T>
bool wrow_plan(int mode, const RowsArgs<T> &a, int *vec, int *k);
template <typename T>
int32_t launch_wrow(ciao_ctx *ctx, int mode, int vec, int k, int grid, RowsArgs<T> &a);
template <typename T>
int wrow_block_waves(int vec, int k);

// sweeps / batch steps over rows beyond 64 KiB: a cluster of S workgroups per row (rowsl_kernels.h).  Specialised in rowsl_f32.hip /
// rowsl_f64.hip.  long_plan: CIAO_ERR_UNSUPPORTED (no error text) when the shape is not for this kernel.
constexpr int LONG_J_DEFAULT = 8;   // 16-byte chunks per thread: a workgroup's segment is J * 4 KiB of the row (Finito batches: 4)
constexpr int LONG_BPC_J4 = 4, LONG_BPC_J8 = 2;   // workgroups per CU (never more than the occupancy calculator allows)
template <typename T>
int32_t long_plan(ciao_ctx *ctx, int mode, int64_t d, int64_t nrows, int *J, int *S, int *C);
template <typename T>
int32_t launch_long(ciao_ctx *ctx, int mode, int J, int S, int C, RowsArgs<T> &a);

// ProShI agent rows (init or one batch) + finalize + epilogue.  Specialised in rows_f32.hip / rows_f64.hip.
template <typename T>
int32_t launch_proshi(ciao_ctx *ctx, bool init, ProshiArgs<T> &a, const Epilogue<T> &ep);
// adaptive Finito chain (one sample per step, backtracking).  Specialised in chain_f32.hip / chain_f64.hip.
template <typename T>
int32_t launch_afinito(ciao_ctx *ctx, int loss, AFinitoArgs<T> &a);

}  // namespace ciao
