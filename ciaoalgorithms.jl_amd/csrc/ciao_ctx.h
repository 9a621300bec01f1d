// ciao_ctx.h -- host-side context of libciao_hip.so (private; the public view is include/ciao_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#define CIAO_BENCH_API 1
#include "../../include/ciao_hip.h"

// one recorded chain of an open batch (ciao_ctx_chain_batch_begin): the launch it would have been
struct ciao_chain_rec {
    const void *kern = nullptr;          // the kernel (host function pointer)
    unsigned block = 0;
    size_t lds = 0;
    std::vector<unsigned char> args;     // its ChainArgs<T>
    const unsigned char *lo[5] = {}, *hi[5] = {};   // byte ranges of av, z, z_full, w, table ...
    bool wr[5] = {};                                //   ... and whether the chain writes them
    std::string name;
};

struct ciao_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cu = 256;

    // all-reduce hook (multi-GPU); nullptr = single device
    ciao_allreduce_fn hook = nullptr;
    void *hook_user = nullptr;

    // row-sharded problem for the sequential chains (ciao_ctx_set_shards); nshards = 0: none
    ciao_shard_table shards{};

    // private workspace (grown on demand, never inside a timed region after the first call of a given shape)
    void *partial = nullptr;   // per-block partial d-vectors
    void *mrhs_ptrs = nullptr;  // multi-right-hand-side pass: device tables of iterate / output pointers
    size_t mrhs_ptrs_bytes = 0;
    std::vector<void *> mrhs_host;   // their host staging copy
    int64_t mrhs_off = 0;       // option multi_rhs_off: K iterates as K single sweeps (testing)
    size_t partial_bytes = 0;
    void *pextra = nullptr;    // per-block extra scalars
    size_t pextra_bytes = 0;
    void *sumbuf = nullptr;    // raw reduced sum (d + 1 elements) handed to the all-reduce hook
    size_t sumbuf_bytes = 0;
    void *rowdot = nullptr;    // a_i'z_full per local row, written by the SVRG full pass, read by the SVRG chain
    size_t rowdot_bytes = 0;
    const void *rowdot_A = nullptr, *rowdot_x = nullptr;   // which (data matrix, z_full vector) the cache belongs to
    int64_t rowdot_N = -1;
    double *monitor = nullptr;         // ciao_ctx_set_monitor: device double[3] {F, (1/N) sum f_i, g} of the last full pass, or NULL
    bool monitor_has_g = false;
    ciao_prox_desc monitor_g{};
    void *monx = nullptr;              // copy of x for the monitor when a step overwrites x in place
    size_t monx_bytes = 0;
    void *idxbuf = nullptr;            // spelled-out indices of row-block batches that run as a sequential chain
    size_t idxbuf_bytes = 0;
    double *scal = nullptr;    // small device scratch for scalar reductions (4096 doubles)
    int *errflag = nullptr;    // sticky device error word (out-of-range index)

    // native RCCL all-reduce (ciao_ctx_set_rccl): resolved with dlopen so that nothing links against RCCL
    void *rccl_comm = nullptr;
    void *rccl_lib = nullptr;
    int (*rccl_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*rccl_errstr)(int) = nullptr;

    // one-shot peer all-reduce (ciao_ctx_set_peers; peer_kernels.h): world == 0 = off.  While set, ctx->hook is the library's own
    // peer hook, so everything that asks "is this a row-sharded run?" keeps asking ctx->hook.
    int peer_world = 0, peer_rank = 0;
    unsigned int peer_seq = 0;                 // sequence number of the last reduction (the same on every rank)
    int64_t peer_slot_bytes = 0, peer_max_elems = 0;
    unsigned char *peer_mail[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned int *peer_counter = nullptr;      // device word: workgroups of the sending kernel that are done
    int64_t peer_timeout_s = 30;               // option peer_timeout_s: how long a reduction waits for a rank (wall clock)
    int64_t peer_wait_s_once = 0;              // set around ONE reduction that has to wait longer (the chain owner's broadcast)
    // what ciao_ctx_set_peers displaced, put back when the peers are turned off
    ciao_allreduce_fn peer_saved_hook = nullptr;
    void *peer_saved_user = nullptr;
    bool peer_saved = false;

    // tuning
    int64_t sweep_blocks_per_cu = 0;   // 0 = choose from the row size (rows_launch.inc)
    int64_t sweep_multi = 1;           // short rows (<= 4 KiB): several rows per wave per iteration (rows_multi_kernel)
    int64_t split_max_rows = -1;       // batches up to this many rows run one workgroup per row (rows_split_kernel); -1 = automatic
    int64_t wrow_rows_per_wave = 0;    // ... rows per wave a small batch is cut into (0: four)
    int64_t small_wrow = -1;           // Finito batches over index lists on rows of tabular size one wave per row (rows_wrow_kernel): 0 = off (rows_smallb_kernel)
    int64_t small_mfma_table = -1;     // ... also the table modes (SAGA / Finito init, Finito batches over row blocks): 0 = off
    int64_t small_nb = 0;              // rows_smallm_kernel: tile buffers per wave, 2 .. 4 (0 = automatic)
    int64_t small_mfma = -1;           // rows_smallm_kernel (GRAD sweeps on dense rows of 17 .. 256 elements on the matrix cores): 0 = off
    int64_t small_i = 0;               // rows_small_kernel: elements per lane and iteration, 8 or 16 (0 = automatic)
    int64_t split_all = 0;             // tuning experiment: workgroup-per-row kernel for every mode and size
    int64_t split_blocks_per_cu = 0;   // its grid cap in blocks per CU (0 = automatic)
    int64_t long_rows = 1;             // rows beyond 64 KiB on the cluster kernel (rows_long_kernel); 0 = the generic kernel (testing)
    int64_t long_j = 0;                // its chunks per thread, 4 or 8 (0 = automatic)
    int64_t sweep_grid = 0;            // testing: absolute grid override for the rows kernels (0 = automatic)
    int64_t sweep_prefetch = -1;    // gradient sweeps: 1 = two-deep register pipeline, 0 = occupancy only, -1 = by row size
    int64_t chain_max_batch = -1;   // Finito/LFinito batches up to this size run as a sequential chain (-1 = automatic)
    int64_t svrg_cache_rowdots = 1; // reuse a_i'z_full from the full pass inside the SVRG inner cycle (ciao_svrg_iterate)
    int64_t proshi_chain_max_batch = -1;   // ProShI batches up to this size run as one coordinate-parallel chain launch (-1 = automatic)
    int64_t chain_four_waves = 0;   // testing: short rows (<= 2 KiB) on the four-wave chain instead of the single-wave one
    int64_t chain_no_wide = 0;      // testing: rows beyond 8192 elements on chain_big_kernel instead of the several-workgroup chain_wide_kernel
    void *wide_box = nullptr;       // chain_wide_kernel: the workgroups' mailbox
    size_t wide_box_bytes = 0;
    int64_t chain_big = 0;          // testing: route chains through chain_big_kernel (the any-d kernel) whatever d
    int64_t chain_no_dma = 0;       // testing: route chains through the register-ring kernel instead of the LDS-DMA one
    int64_t chain_no_ws = 0;        // testing: SVRG / SAGA chains on chain_dma_kernel instead of the wave-specialised chain_ws_kernel
    int64_t chain_ws_issuers = 0;   // tuning: issuer waves of chain_ws_kernel, 1 or 2 (0 = automatic)
    int chain_last_ws = 0;          // issuer waves of the wave-specialised chain the last launch took, 0 = another kernel
    int chain_last_one_wave = 0;    // E of the single-wave register-ring chain the last launch took, 0 = four waves
    int chain_last_dma = 0;
    int long_occ[2][8] = {};        // rows_long_kernel: workgroups per CU by (J == 8, mode), asked of the runtime once (0 = not yet)
    int chain_last_block = 0;   // threads of the chain_dma_kernel launch that was made (the launcher records it: the name reports what ran)
    bool chain_last_masked = false;
    int64_t force_generic = 0;      // testing: route every rows launch through the generic kernel

    std::string last_kernel;

    // a batch of independent chains for one launch per kernel (ciao_ctx_chain_batch_begin / _end): while open, the chain entry
    // points record instead of launching
    int batch_open = 0;
    std::vector<ciao_chain_rec> batch;
    void *batch_dev = nullptr;           // the launched batch's argument blocks (device) ...
    size_t batch_dev_bytes = 0;
    void *batch_host = nullptr;          // ... and their pinned staging copy
    size_t batch_host_bytes = 0;
    hipEvent_t batch_ev = nullptr;       // the staging copy of the previous batch has been read

    // optional HIP-event timing of the dominant kernel (bench.py roofline)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;   // pairs (start, stop), reused after every read
    size_t ev_used = 0;
};

namespace ciao {

void set_error(const char *fmt, ...);
int32_t hip_fail(hipError_t e, const char *what);

#define CIAO_HIP(call)                                                  \
    do {                                                                \
        hipError_t _e = (call);                                         \
        if (_e != hipSuccess) return ::ciao::hip_fail(_e, #call);       \
    } while (0)

#define CIAO_REQUIRE(cond, ...)                                         \
    do {                                                                \
        if (!(cond)) {                                                  \
            ::ciao::set_error(__VA_ARGS__);                             \
            return CIAO_ERR_ARG;                                        \
        }                                                               \
    } while (0)

#define CIAO_TRY(expr)                                                  \
    do {                                                                \
        int32_t _s = (expr);                                            \
        if (_s != CIAO_OK) return _s;                                   \
    } while (0)

int32_t ensure(ciao_ctx *ctx, void **buf, size_t *have, size_t need);

// Every extern "C" entry point runs with the ctx's device current and restores the caller's device on the way out: a host
// that holds several contexts (one per GPU) and switches devices between calls must not get workspace allocations or
// launches on the wrong GPU.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// (while a chain batch is open only the entry points that record into it -- CIAO_ENTER_BATCHABLE -- may be called)
#define CIAO_ENTER_BATCHABLE(ctx)           \
    CIAO_REQUIRE((ctx), "ctx is NULL");     \
    ::ciao::DeviceGuard _ciao_dg((ctx)->device)
#define CIAO_ENTER(ctx)                                                                                                           \
    CIAO_REQUIRE((ctx), "ctx is NULL");                                                                                           \
    CIAO_REQUIRE(!(ctx)->batch_open, "a chain batch is open on this ctx (ciao_ctx_chain_batch_begin): only ciao_svrg_inner, "    \
                                     "ciao_saga_steps and ciao_finito_steps may be called before ciao_ctx_chain_batch_end");      \
    ::ciao::DeviceGuard _ciao_dg((ctx)->device)

}  // namespace ciao
