// Shared host/device helpers for libsmt_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include "../../include/smt.h"

#define SMT_API extern "C" __attribute__((visibility("default")))

extern thread_local int g_smt_last_hip;

#define SMT_HIP(expr)                                      \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            g_smt_last_hip = (int)_e;                      \
            return SMT_ERR_HIP;                            \
        }                                                  \
    } while (0)

#define SMT_LAUNCH_CHECK()                                 \
    do {                                                   \
        hipError_t _e = hipGetLastError();                 \
        if (_e != hipSuccess) {                            \
            g_smt_last_hip = (int)_e;                      \
            return SMT_ERR_HIP;                            \
        }                                                  \
    } while (0)

static inline hipStream_t smt_stream(void *s) { return (hipStream_t)s; }

// Every handle remembers the device it was created on; its entry points make that device current for
// the duration of the call and restore the caller's afterwards (one host thread may drive several GPUs).
struct smt_dev_guard {
    int prev = -1;
    explicit smt_dev_guard(int dev)
    {
        int cur = -1;
        if (dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~smt_dev_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
    smt_dev_guard(const smt_dev_guard &) = delete;
    smt_dev_guard &operator=(const smt_dev_guard &) = delete;
};
static inline int smt_current_device() { int d = -1; return hipGetDevice(&d) == hipSuccess ? d : -1; }

// Stream-ordered scratch for smt_asw and smt_ncc (tables that live for one call): csrc/scratch.hip owns the one pool
// set of the library.  Free with smt_scratch_free on the same stream.
hipError_t smt_scratch_alloc(void **p, size_t bytes, hipStream_t st);
void smt_scratch_free(void *p, hipStream_t st);

#ifdef __HIPCC__
constexpr int WAVE = 64;

// std::min / std::max semantics of the reference (returns the FIRST argument on ties and
// for the +0/-0 pair), kept instead of fminf so that signs of zero cannot diverge.
__device__ __forceinline__ float ref_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float ref_max(float a, float b) { return (a < b) ? b : a; }

// Wave-wide minimum of non-NaN floats (all 64 lanes active).
__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, WAVE));
    return v;
}
// ---- DPP wave reductions (gfx9 row_bcast forms; all 64 lanes must be active) ------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    // old = identity of umin so the DPP combiner can fold the move into v_min_u32_dpp
    return (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// minimum over the wave of unsigned keys, returned wave-uniform (SGPR).
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
    v = min(v, dpp_u32<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
    v = min(v, dpp_u32<0x141, 0xF>(v));   // row_half_mirror
    v = min(v, dpp_u32<0x140, 0xF>(v));   // row_mirror      -> every lane holds its row's min
    v = min(v, dpp_u32<0x142, 0xA>(v));   // row_bcast15 into rows 1,3
    v = min(v, dpp_u32<0x143, 0xC>(v));   // row_bcast31 into rows 2,3 -> lane 63 holds the min
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// order-preserving map float -> uint32 (handles negatives; NaNs are not ordered -- wave_wta gives them
// the reference's semantics)
__device__ __forceinline__ unsigned f32_key(float f)
{
    const unsigned b = __float_as_uint(f + 0.0f);          // -0 -> +0 so that equal zeros tie
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// WTA of one pixel held by a wave (lane l owns the C hypotheses dl = l*C .. l*C+C-1; elements at or past
// D are ignored), with the reference's exact rule (CrossArm.cpp:44-52, ScanlineOptimizer.h:51-59,
// AD-Census.h:355-373): `cost = vol[0]; for d >= 1: if (cost > vol[d]) { cost = vol[d]; best = d; }`.
// A NaN never satisfies `cost > value`, so a NaN entry never wins, and a NaN at d = 0 freezes the
// result at 0 (every later comparison against it is false).  NaN keys sort above +inf; -0 ties +0.
template <int C, bool FULL>
__device__ __forceinline__ int wave_wta(const float (&v)[C], int dl, int D)
{
    unsigned best = 0xFFFFFFFFu; int bk = 0;
#pragma unroll
    for (int k = 0; k < C; k++)
        if (FULL || dl + k < D) {
            const unsigned key = (v[k] != v[k]) ? 0xFFFFFFFFu : f32_key(v[k]);
            if (k == 0 || key < best) { best = key; bk = k; }
        }
    const unsigned m = wave_min_u32(best);
    const unsigned long long b = __ballot(best == m);
    const int first = __builtin_ctzll(b);
    const int wd = __builtin_amdgcn_readlane(dl + bk, first);
    const int nan0 = __builtin_amdgcn_readlane((int)(v[0] != v[0]), 0);   // lane 0 owns d = 0
    return nan0 ? 0 : wd;
}

// Correctly rounded a[k] / b for a wave-uniform integer-valued divisor 1 <= b <= 65535 (a rectangle area, a support
// count) in 3 vector instructions per value instead of the 11 of the IEEE division sequence.  With r = v_rcp_f32(b)
// (error <= 1 ulp; 2 would do), q0 = RN(a r), rem = fma(-q0, b, a), q = fma(rem, r, q0):
//  * |q0 - a/b| < 3 ulp, so rem = a - q0 b is a multiple of ulp(q0) and fewer than 2^19 of them: exact in the fma;
//  * q0 + rem r = a/b + (a/b - q0) e1 with |e1| <= 2^-23: within 2^-21 ulp of the true quotient;
//  * a/b is never closer than 2^-17 ulp to a rounding boundary (for a midpoint m, a - b m is a non-zero multiple of half
//    an ulp of the quotient -- b m has 25 significant bits or more, a has 24 -- and b < 2^16),
// so the final rounding is the rounding of a/b.  That holds while nothing leaves the normal range: every value the wave
// holds for the pixel must lie in [2^-60, 2^61) (zeros, denormals, infinities, NaNs and larger divisors take the
// division), checked with one ballot.  Elements at or past D do not count.  All 64 lanes must be active.
template <int C, bool FULL>
__device__ __forceinline__ void wave_quotient(const float (&a)[C], float b, int dl, int D, float (&q)[C])
{
    bool fast = false;
    if (b >= 1.0f && b <= 65535.0f) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const unsigned t = (__float_as_uint(a[k]) << 1) - (67u << 24);       // biased exponent - 67, sign dropped
            ok = ok && ((!FULL && dl + k >= D) || t < (121u << 24));
        }
        fast = __ballot(!ok) == 0;
    }
    if (fast) {
        float r = __builtin_amdgcn_rcpf(b), bv = b;
        asm volatile("" : "+v"(r), "+v"(bv));              // vector operands: an SGPR multiplier halves the FMA rate
#pragma unroll
        for (int k = 0; k < C; k++) {
            const float q0 = a[k] * r;
            const float rem = __builtin_fmaf(-q0, bv, a[k]);
            q[k] = __builtin_fmaf(rem, r, q0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < C; k++) q[k] = a[k] / b;
    }
}

// First-strict-minimum WTA across a wave whose lanes hold candidates in increasing-d
// order: (v, d) = this lane's first local minimum (v = +inf for lanes with no candidate).
// Returns the winning d, wave-uniform.  All 64 lanes must be active.
__device__ __forceinline__ int wave_argmin_first(float v, int d)
{
    const unsigned key = f32_key(v);
    const unsigned m = wave_min_u32(key);
    const unsigned long long b = __ballot(key == m);
    const int first = __builtin_ctzll(b);
    return __builtin_amdgcn_readlane(d, first);
}
#endif
