// peer_kernels.h -- the one-shot peer all-reduce of the d+1 scalars of a sweep / batch ("tuned
// alternative"; no counterpart in the reference, which has no collective at all).
//
// The message is 4-16 KB and the 8 GPUs of a node are fully connected (7 xGMI links per GPU): a ring is the wrong shape, the wire
// time is a fraction of a microsecond and everything that matters is latency -- an RCCL call per reduction costs more than the
// reduction.  So every rank owns a MAILBOX in its own HBM (fine-grained memory allocated by the library, mapped into the other
// ranks by HIP IPC): two parities x one slot per rank.  A reduction with sequence number q:
//   * the kernel that produced the rank's raw sum (finalize_kernel) stores it into slot [q & 1][rank] of EVERY rank's mailbox -- one
//     direct write per peer, one xGMI link each -- and the LAST of its workgroups to finish (device-scope counter) then releases
//     flag [q & 1][rank] = q in every mailbox (system scope);
//   * the kernel that consumes the sum (the epilogue) waits in its own mailbox for the world's flags to read q (acquire, system
//     scope), adds the slots IN RANK ORDER -- every rank forms the bitwise-identical sum -- and applies the epilogue.
// No collective call, no extra launch: the two kernels exist anyway.  Two parities suffice: a rank can only write reduction q+2
// after its epilogue of q+1 has seen every rank's flag q+1, which each rank set after ITS epilogue of q had read the slots of q.
// The wait is bounded in wall-clock time (errflag 4, then ciao_ctx_synchronize reports it): a rank that never arrives is an error, not
// a hung GPU.
#pragma once

#include "ciao_common.h"

namespace ciao {

constexpr int PEER_MAX = 8;
constexpr int64_t PEER_HDR = 1024;        // flags: [2 parities][8 ranks] x 64 bytes
constexpr unsigned long long PEER_TICKS_PER_S = 100000000ull;   // s_memrealtime: the 100 MHz constant clock
constexpr int64_t PEER_WAIT_S_DEFAULT = 30;   // a reduction between ranks that run the same kernels: far more than any skew
                                              // (option peer_timeout_s; the chain owner's broadcast waits for a whole chain, launch.h)

struct PeerDev {
    int world, rank;                   // world == 0: no peers (single device, or the hook path)
    unsigned int seq;                  // this reduction's sequence number (>= 1; the same on every rank)
    int64_t slot_bytes;
    unsigned char *mail[PEER_MAX];     // mail[r] = rank r's mailbox (mail[rank] = own; the others IPC-mapped)
    unsigned int *counter;             // device-scope "workgroups done" counter of the sending kernel (this rank's own memory)
    int *errflag;
    unsigned long long wait_ticks;     // how long peer_wait() waits for a rank before it gives up (realtime ticks)
};

__device__ __forceinline__ unsigned int *peer_flag(const PeerDev &p, int owner, int from)
{
    return reinterpret_cast<unsigned int *>(p.mail[owner] + (int64_t)(((p.seq & 1u) * PEER_MAX + (unsigned)from) * 64));
}
template <typename T>
__device__ __forceinline__ T *peer_slot(const PeerDev &p, int owner, int from)
{
    return reinterpret_cast<T *>(p.mail[owner] + PEER_HDR + (int64_t)((p.seq & 1u) * PEER_MAX + (unsigned)from) * p.slot_bytes);
}

// element k of this rank's contribution -> every rank's mailbox
template <typename T>
__device__ __forceinline__ void peer_put(const PeerDev &p, int64_t k, T v)
{
    for (int r = 0; r < p.world; ++r) peer_slot<T>(p, r, p.rank)[k] = v;
}

// End of the sending kernel, called by EVERY thread of EVERY workgroup after its peer_put()s: the last workgroup to get here
// publishes the flags.  (fence: this thread's stores are performed system-wide; barrier: the workgroup's are; the counter orders
// the workgroups; the fence behind it orders the flags after everything the counter has seen.)
__device__ __forceinline__ void peer_publish(const PeerDev &p)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = gridDim.x * gridDim.y;
        const unsigned int old = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == total - 1) {
            __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next reduction
            __threadfence_system();
            for (int r = 0; r < p.world; ++r) __hip_atomic_store(peer_flag(p, r, p.rank), p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Start of the consuming kernel, called by every thread of a workgroup (of at least one full wave): returns false if a rank never
// arrived.  Lane r of the first wave polls rank r's flag, so the world's flags are ONE memory round trip per poll, not one per rank
// (eight ranks polled one after the other are eight dependent round trips before the first slot is read); the polls are relaxed and
// the acquire is made once, behind the last of them.
__device__ __forceinline__ bool peer_wait(const PeerDev &p)
{
    __shared__ int peer_ok;
    if (threadIdx.x < WAVE) {
        const int r = (int)threadIdx.x;
        const bool mine = r < p.world;
        const unsigned int *flag = peer_flag(p, p.rank, mine ? r : 0);
        int ok = 1;
        // bounded by WALL-CLOCK time, not by a poll count (a non-owner of a sharded chain waits here for the
        // owner's whole chain kernel -- seconds at 10^7 steps -- and a count of ~1 us polls gave up on a correct run)
        unsigned long long t0 = 0;
        while (true) {
            const unsigned int v = mine ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : p.seq;
            if (__all(v == p.seq)) break;
            __builtin_amdgcn_s_sleep(32);
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now ? now : 1ull;
            if (now - t0 > p.wait_ticks) {
                if (threadIdx.x == 0) *p.errflag = 4;
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: once, behind the flags
        if (threadIdx.x == 0) peer_ok = ok;
    }
    __syncthreads();
    return peer_ok != 0;
}

// element k of the reduced sum: the world's slots of this rank's own mailbox, added in rank order (system-scope loads: the
// slots were written by other GPUs)
__device__ __forceinline__ double peer_load(const double *q)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ float peer_load(const float *q)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// (all the world's loads requested before the first addition: a loop over a run-time world waits for each before it asks for the next)
template <typename T>
__device__ __forceinline__ T peer_sum(const PeerDev &p, int64_t k)
{
    T v[PEER_MAX];
#pragma unroll
    for (int r = 0; r < PEER_MAX; ++r) v[r] = r < p.world ? peer_load(peer_slot<T>(p, p.rank, r) + k) : T(0);
    T s = v[0];
#pragma unroll
    for (int r = 1; r < PEER_MAX; ++r)
        if (r < p.world) s += v[r];
    return s;
}

// the sending half on its own (the library's all-reduce hook on an arbitrary buffer: chain-owner broadcasts, bench): buf[0..count)
template <typename T>
__global__ void __launch_bounds__(256) peer_send_kernel(const T *__restrict__ buf, int64_t count, PeerDev p)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) peer_put(p, k, buf[k]);
    peer_publish(p);
}

// the consuming half: the reduced sum into raw_out[0..count) (hook form), or the epilogue on it (fused form: count = d + 1,
// the extra scalar at [d])
template <typename T>
__global__ void __launch_bounds__(256) peer_recv_kernel(T *__restrict__ raw_out, int64_t count, PeerDev p)
{
    if (!peer_wait(p)) return;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) raw_out[k] = peer_sum<T>(p, k);
}
template <typename T>
__global__ void __launch_bounds__(256) peer_epilogue_kernel(int64_t d, Epilogue<T> ep, PeerDev p)
{
    if (!peer_wait(p)) return;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ep.z_out && ep.g.kind == CIAO_PROX_L1_COMPLEX) {   // (re, im) pairs: thread k takes coordinates 2k and 2k+1
        if (2 * k + 1 < d) epilogue_apply2(ep, 2 * k, peer_sum<T>(p, 2 * k), peer_sum<T>(p, 2 * k + 1), peer_sum<T>(p, d));
        return;
    }
    if (k < d) epilogue_apply(ep, k, peer_sum<T>(p, k), peer_sum<T>(p, d));
}

}  // namespace ciao
