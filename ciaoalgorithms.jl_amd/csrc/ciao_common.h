// ciao_common.h -- shared device helpers for the gfx950 kernels of libciao_hip.so.
//
// CDNA4 facts this code is written against (MI355X_MICROARCH.md): 64-lane waves on SIMD-32 units, 256 CUs in 8 XCDs,
// 160 KiB LDS per CU, 16 B per lane = 1 KiB per wave-instruction is the widest coalesced access, HBM3E 8 TB/s spec
// (about 6.3 TB/s measured for streaming), per-XCD L2s that are not coherent with each other (so every cross-workgroup
// hand-off here is a kernel boundary, never an in-kernel flag).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CIAO_BENCH_API 1
#include "../../include/ciao_hip.h"

namespace ciao {

constexpr int WAVE = 64;

// The FIRST kernel argument (a struct passed by value), read through the kernel-argument segment, field by field WHERE IT IS USED:
// hipcc loads every field of a by-value argument into scalar registers in the entry block, and the fields that only a prologue, a
// staging phase or the final stores need then stay live through the hot loop -- 10 to 100 scalar registers pushed out to VGPR lanes
// in the chain / long-row / small-row families (tools/kernel_meta.py).  The block is constant memory: scalar loads.  A field a hot
// LOOP reads is loaded once into a local through sgpr_pin (below): read in place hipcc re-loads it inside the loop, a scalar-cache
// round trip behind a full wait on every iteration (profiles/r05_kernarg_ab.txt: up to 21 % of a sweep, 19 % of a chain step).
#ifdef CIAO_KERNARG_BYVALUE   // same-box A/B only (tools/exp/kernarg_ab.sh): the by-value argument as it is
#define CIAO_KERNARG0(Type, name) const Type &name = a_by_value
#else
#define CIAO_KERNARG0(Type, name)                                   \
    typedef const __attribute__((address_space(4))) Type name##_kernarg_t; \
    name##_kernarg_t &name = *(name##_kernarg_t *)__builtin_amdgcn_kernarg_segment_ptr()
#endif

// A field of the argument block that the STEP LOOP reads: loaded once, and made opaque in its scalar register(s) so that hipcc
// neither re-loads it from the block inside the loop (a scalar-cache round trip on the dependent path of every step -- what it did
// with a.gamma, a.invN, a.lam once the block was read through a pointer) nor keeps it anywhere but in SGPRs.
template <typename X>
__device__ __forceinline__ X sgpr_pin(X v)
{
    asm volatile("" : "+s"(v));
    return v;
}
// ... a value COMPUTED from such fields (the vector pipeline did the arithmetic): through v_readfirstlane, then pinned
__device__ __forceinline__ float sgpr_pin_computed(float v)
{
    return sgpr_pin(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))));
}
__device__ __forceinline__ double sgpr_pin_computed(double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return sgpr_pin(__builtin_bit_cast(double, ((uint64_t)hi << 32) | lo));
}
// ... a pointer field: the asm hides that it came from the argument block, so say again that it is global memory (generic
// pointers make FLAT loads / stores, which count on both memory counters and break the hand-counted waits)
template <typename P>
__device__ __forceinline__ P *sgpr_pin_global(P *p)
{
    asm volatile("" : "+s"(p));
    return (P *)(__attribute__((address_space(1))) P *)(uintptr_t)p;
}


// ------------------------------------------------------------------------------------------------------------------
// DPP cross-lane moves (32-bit halves; 64-bit values move as two halves).  Controls used:
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
// After the four steps every lane of a 16-lane row holds the row's sum; the four row sums are then combined through
// v_readlane (SGPRs), in a fixed order, so all 64 lanes end with the bitwise-identical total.
// ------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float readlane(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double readlane(double v, int l)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), l);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes of a wave; result identical (bitwise) in every lane.  Requires all 64 lanes active.
template <typename T>
__device__ __forceinline__ T wave_allsum(T v)
{
    v += dpp_mov<0xB1>(v);   // xor 1
    v += dpp_mov<0x4E>(v);   // xor 2
    v += dpp_mov<0x141>(v);  // row_half_mirror: quads 0<->1, 2<->3 within a row of 16
    v += dpp_mov<0x140>(v);  // row_mirror: halves of the row
    T r0 = readlane(v, 0), r1 = readlane(v, 16), r2 = readlane(v, 32), r3 = readlane(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// The same sum, delivered to LANE 63 ONLY (the other lanes hold partial sums): for the chains' cross-wave exchange, where one
// lane stores the wave's partial.  After the four in-row stages every lane holds its row-of-16 sum r0..r3; row_bcast:15 adds
// lane 15 of the previous row into rows 1 and 3, row_bcast:31 adds lane 31 into rows 2 and 3, so lane 63 ends with
// (r3 + r2) + (r1 + r0) -- bitwise the value wave_allsum returns, (r0 + r1) + (r2 + r3), without the eight v_readlane and the
// scalar round trip.
// (the rows outside ROWS receive an UNDEFINED value -- no zero-fill moves: only lanes 31 and 63 of the result are ever used)
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_rows(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, ROWS, 0xF, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_rows(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROWS, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROWS, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum_lane63(T v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    v += dpp_rows<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_rows<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// ------------------------------------------------------------------------------------------------------------------
// Operators (ProximalOperators.jl 0.14 formulas)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fmad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmin2(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double fmin2(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fmax2(double a, double b) { return __builtin_fmax(a, b); }
// clamp(v, -t, t) for t >= 0: one v_med3_f32 in single precision, max/min in double
__device__ __forceinline__ float clamp_sym(float v, float t) { return __builtin_amdgcn_fmed3f(v, -t, t); }
__device__ __forceinline__ double clamp_sym(double v, double t) { return __builtin_fmin(__builtin_fmax(v, -t), t); }
__device__ __forceinline__ float fsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float fabs2(float x) { return fabsf(x); }
__device__ __forceinline__ double fabs2(double x) { return fabs(x); }
template <typename T>
struct Eps;
template <>
struct Eps<float> {
    static constexpr float value = 1.1920928955078125e-07f;
};
template <>
struct Eps<double> {
    static constexpr double value = 2.220446049250313e-16;
};
__device__ __forceinline__ float fexp(float x) { return expf(x); }
__device__ __forceinline__ double fexp(double x) { return exp(x); }
__device__ __forceinline__ float flog(float x) { return logf(x); }
__device__ __forceinline__ double flog(double x) { return log(x); }

// The per-sample gradient is grad f_i(x)_k = (a_k * s1) * s2 with
//   LS:        s1 = a'x - b_i,                 s2 = lam        (mul!(y, A', res); y .*= lam)
//   logistic:  s1 = -y_i / (1 + exp(y_i a'x)), s2 = 1
//   zero:      s1 = 0,                         s2 = 0
// so that one code path serves the three families.  `coef` = s1*s2 is the rank-1 coefficient of the row.
template <typename T>
struct GradCoef {
    T s1, s2;
    __device__ __forceinline__ T coef() const { return s1 * s2; }
    __device__ __forceinline__ T elem(T a) const { return (a * s1) * s2; }
};

template <typename T>
__device__ __forceinline__ GradCoef<T> grad_coef(int loss, T dot, T bi, T lam)
{
    GradCoef<T> g;
    if (loss == CIAO_LOSS_LS) {
        g.s1 = dot - bi;
        g.s2 = lam;
    } else if (loss == CIAO_LOSS_LOGISTIC) {
        g.s1 = -bi / (T(1) + fexp(bi * dot));
        g.s2 = T(1);
    } else {
        g.s1 = T(0);
        g.s2 = T(0);
    }
    return g;
}

__device__ __forceinline__ float fhypot(float a, float b) { return hypotf(a, b); }
__device__ __forceinline__ double fhypot(double a, double b) { return hypot(a, b); }

// ---- complex T as interleaved (re, im) pairs (CIAO_LOSS_LS_COMPLEX / CIAO_PROX_L1_COMPLEX) --------------------------------
// prox_{gl |.|}(v) for one complex coordinate: sign(v) * max(|v| - gl, 0) with the complex modulus and sign = v / |v|
// (ProximalOperators.jl NormL1 on a complex array), in that operation order.
template <typename T>
__device__ __forceinline__ void prox_cpair(T gl, T vr, T vi, T &yr, T &yi)
{
    const T ax = fhypot(vr, vi);
    if (ax > gl) {
        const T m = ax - gl;
        yr = (vr / ax) * m;
        yi = (vi / ax) * m;
    } else {
        yr = T(0);
        yi = T(0);
    }
}
// The same operator for the sequential chains, where it sits on the dependent path of every step (the form above is a hypot and two
// divisions per coordinate: ~60 fp64 instructions, half of a complex chain step):  sign(v) max(|v| - gl, 0) = v (1 - gl / |v|)
// for |v| > gl, with 1 / |v| = rsqrt(re^2 + im^2) -- about a dozen instructions.  Same value to rounding (absolute error below
// 2 eps |v|); re^2 + im^2 needs |v| within 1e-154 .. 1e154 (fp32: 1e-19 .. 1e19): beyond, inf gives y = v (off by gl / |v| < 1e-154),
// 0 gives y = 0 (off by less than 1e-154).
__device__ __forceinline__ float frsqrt(float x) { return rsqrtf(x); }
__device__ __forceinline__ double frsqrt(double x) { return rsqrt(x); }
template <typename T>
__device__ __forceinline__ void prox_cpair_chain(T gl, T vr, T vi, T &yr, T &yi)
{
    const T n2 = fmad(vr, vr, vi * vi);
    // (the floor: with gl = 0 or tiny a SUBNORMAL n2 would pass the test, v_rsq of a flushed subnormal is inf, and gl * inf = NaN;
    // below the smallest normal number the coordinate is 0 to the stated error.  gl is a per-chain constant: the max is hoisted.)
    constexpr T tiny = sizeof(T) == 8 ? T(2.2250738585072014e-308) : T(1.17549435e-38f);
    const T thr = gl * gl > tiny ? gl * gl : tiny;
    if (n2 > thr) {
        const T s = T(1) - gl * frsqrt(n2);
        yr = vr * s;
        yi = vi * s;
    } else {
        yr = T(0);
        yi = T(0);
    }
}
// grad f_i(x)_k = (conj(a_k) * res) * lam for the complex residual res = a_i.x - b_i   (mul!(y, A', res); y .*= lam)
template <typename T>
__device__ __forceinline__ void cgrad_elem(T ar, T ai, T rr, T ri, T lam, T &gr, T &gi)
{
    gr = (ar * rr + ai * ri) * lam;
    gi = (ar * ri - ai * rr) * lam;
}

// f_i(x) given the same scalars (the value gradient! returns).
template <typename T>
__device__ __forceinline__ T loss_value(int loss, T dot, T bi, T lam)
{
    if (loss == CIAO_LOSS_LS) {
        T r = dot - bi;
        return (lam / T(2)) * r * r;
    } else if (loss == CIAO_LOSS_LOGISTIC) {
        // log(1 + exp(-y t)), evaluated stably on either side
        T u = -bi * dot;
        return u > T(0) ? u + flog(T(1) + fexp(-u)) : flog(T(1) + fexp(u));
    }
    return T(0);
}

template <typename T>
struct ProxD {
    int kind;
    T lam, lo, hi;
    const T *lo_vec, *hi_vec;
};

template <typename T>
__host__ inline ProxD<T> make_prox(const ciao_prox_desc *g)
{
    ProxD<T> p;
    if (!g) {
        p.kind = CIAO_PROX_ZERO;
        p.lam = p.lo = p.hi = T(0);
        p.lo_vec = p.hi_vec = nullptr;
        return p;
    }
    p.kind = g->kind;
    p.lam = (T)g->lam;
    p.lo = (T)g->lo;
    p.hi = (T)g->hi;
    p.lo_vec = (const T *)g->lo_vec;
    p.hi_vec = (const T *)g->hi_vec;
    return p;
}

// prox_{gamma g}(v) for coordinate k.
template <typename T>
__device__ __forceinline__ T prox_elem(const ProxD<T> &g, T v, T gamma, int64_t k)
{
    if (g.kind == CIAO_PROX_L1) {
        T gl = gamma * g.lam;
        return v + (v <= -gl ? gl : (v >= gl ? -gl : -v));
    } else if (g.kind == CIAO_PROX_BOX) {
        T l = g.lo_vec ? g.lo_vec[k] : g.lo, h = g.hi_vec ? g.hi_vec[k] : g.hi;
        return v < l ? l : (v > h ? h : v);
    }
    return v;
}

// g(x) contribution of coordinate k (0 for Zero and for a feasible Box point).
template <typename T>
__device__ __forceinline__ T prox_value_elem(const ProxD<T> &g, T v)
{
    return g.kind == CIAO_PROX_L1 ? g.lam * (v < T(0) ? -v : v) : T(0);
}

// prox of the two consecutive coordinates (k, k+1), k even: a complex pair for CIAO_PROX_L1_COMPLEX, two scalars otherwise
template <typename T>
__device__ __forceinline__ void prox_pair(const ProxD<T> &g, T v0, T v1, T gamma, int64_t k, T &y0, T &y1)
{
    if (g.kind == CIAO_PROX_L1_COMPLEX) {
        prox_cpair(gamma * g.lam, v0, v1, y0, y1);
    } else {
        y0 = prox_elem(g, v0, gamma, k);
        y1 = prox_elem(g, v1, gamma, k + 1);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Epilogue applied to a reduced d-vector `sum` (+ one extra reduced scalar), per coordinate k:
//     a = c_acc*acc_in[k] + c_sum*sum[k] + cu*u[k] + cv*v[k]           (cu,cv = c_u,c_v, or +/- extra if uv_extra)
//     av_out[k] = a
//     z_out[k]  = prox_{tau g}(p0*a + p1*pw[k])                         (if z_out)
// Covers: SVRG av = sum/N; LFinito av = z_full - (hg/N) sum; Finito av += sum, z = prox(av); SAGA/Finito init;
// proximal-gradient step y = prox(x - gamma*av).
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
struct Epilogue {
    T c_acc, c_sum, c_u, c_v;
    const T *acc_in, *u, *v;
    T *av_out;
    int uv_extra;  // cu = +extra*c_u, cv = -extra*c_u ... see apply()
    T *z_out;
    T tau, p0, p1;
    const T *pw;
    ProxD<T> g;
    int inv_extra;   // 1 (adaptive Finito init): hat_gamma = 1/extra scales c_sum and is the prox parameter;
                     // 2 (ProShI init): hat_gamma = extra is the prox parameter;  both: *hg_out = hat_gamma
    T *hg_out;
    int zmode;       // 0: z = prox(..) ; 1 (ProShI, ProShI_basic.jl:84-86, :119-121): z = (prox_{tau g}(a) - a) / tau
    double *obj_out; // objective monitor (full passes with want_fval): obj_out[1] = extra * obj_scale = (1/N) sum_i f_i(x)
    double obj_scale;
};

// The epilogue in two halves, so that callers owning two consecutive coordinates can apply a PAIR prox (complex NormL1):
// epilogue_pre  -> a (written to av_out) and the prox argument t = p0*a + p1*pw[k]; operands already in registers
// epilogue_post -> z_out[k] from the prox value
template <typename T>
__device__ __forceinline__ T epilogue_pre(const Epilogue<T> &e, int64_t k, T sum, T extra, T acc_k, T u_k, T v_k, T pw_k)
{
    const T hgx = e.inv_extra == 1 ? T(1) / extra : (e.inv_extra == 2 ? extra : T(1));
    if (e.inv_extra && e.hg_out && k == 0) *e.hg_out = hgx;
    if (e.obj_out && k == 0) e.obj_out[1] = (double)extra * e.obj_scale;
    T a = (e.inv_extra == 1 ? e.c_sum * hgx : e.c_sum) * sum;
    if (e.acc_in) a += e.c_acc * acc_k;
    T cu = e.uv_extra ? e.c_u * extra : e.c_u;
    T cv = e.uv_extra ? e.c_v * extra : e.c_v;
    if (e.u) a += cu * u_k;
    if (e.v) a += cv * v_k;
    if (e.av_out) e.av_out[k] = a;
    T t = e.p0 * a;
    if (e.pw) t += e.p1 * pw_k;
    return t;
}
template <typename T>
__device__ __forceinline__ T epilogue_tau(const Epilogue<T> &e, T extra)
{
    return e.inv_extra == 1 ? T(1) / extra : (e.inv_extra == 2 ? extra : e.tau);
}
template <typename T>
__device__ __forceinline__ void epilogue_post(const Epilogue<T> &e, int64_t k, T t, T tau, T pz)
{
    e.z_out[k] = e.zmode == 1 ? (pz - t) / tau : pz;
}

// one coordinate, operands already in registers (the caller requested them before its reduction)
template <typename T>
__device__ __forceinline__ void epilogue_apply_pre(const Epilogue<T> &e, int64_t k, T sum, T extra, T acc_k, T u_k, T v_k, T pw_k)
{
    const T t = epilogue_pre(e, k, sum, extra, acc_k, u_k, v_k, pw_k);
    if (e.z_out) {
        const T tau = epilogue_tau(e, extra);
        epilogue_post(e, k, t, tau, prox_elem(e.g, t, tau, k));
    }
}

template <typename T>
__device__ __forceinline__ void epilogue_apply(const Epilogue<T> &e, int64_t k, T sum, T extra)
{
    epilogue_apply_pre(e, k, sum, extra, e.acc_in ? e.acc_in[k] : T(0), e.u ? e.u[k] : T(0), e.v ? e.v[k] : T(0),
                       e.pw ? e.pw[k] : T(0));
}

// two consecutive coordinates (k even): pair prox when g is the complex NormL1
template <typename T>
__device__ __forceinline__ void epilogue_apply2(const Epilogue<T> &e, int64_t k, T sum0, T sum1, T extra)
{
    const T t0 = epilogue_pre(e, k, sum0, extra, e.acc_in ? e.acc_in[k] : T(0), e.u ? e.u[k] : T(0), e.v ? e.v[k] : T(0),
                              e.pw ? e.pw[k] : T(0));
    const T t1 = epilogue_pre(e, k + 1, sum1, extra, e.acc_in ? e.acc_in[k + 1] : T(0), e.u ? e.u[k + 1] : T(0),
                              e.v ? e.v[k + 1] : T(0), e.pw ? e.pw[k + 1] : T(0));
    if (e.z_out) {
        const T tau = epilogue_tau(e, extra);
        T y0, y1;
        prox_pair(e.g, t0, t1, tau, k, y0, y1);
        epilogue_post(e, k, t0, tau, y0);
        epilogue_post(e, k + 1, t1, tau, y1);
    }
}

// ---- the matrix cores: v_mfma_{f64,f32}_16x16x4 (16 x 16 output, four k-slots, one element of each operand per lane: A lane (row l & 15,
// slot l >> 4), B lane (column l & 15, slot l >> 4); accumulator lane (column l & 15, l >> 4) holds four rows, row(h, reg))
template <typename T>
struct MfmaOf;
template <>
struct MfmaOf<double> {
    typedef double acc __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D: col = lane & 15, row = (lane >> 4) + 4 reg
    static __device__ __forceinline__ int row(int h, int reg) { return h + 4 * reg; }
};
template <>
struct MfmaOf<float> {
    typedef float acc __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
    static __device__ __forceinline__ int row(int h, int reg) { return 4 * h + reg; }
};


}  // namespace ciao
