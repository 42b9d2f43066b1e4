// Cross-based cost aggregation -- replaces class CrossAggregator
// (CBLSM/cross_aggregator.{h,cpp}, vendored from ethan-li-coding/AD-Census).
//
// Arms (FindHorizontalArm / FindVerticalArm, cross_aggregator.cpp:135-269) and support
// counts (ComputeSupPixelCount, :271-325) are per-pixel work.  AggregateInArms (:327-394)
// is, per disparity plane, two separable passes of SEQUENTIAL float sums along the arms;
// the planes are independent, so one wavefront takes one pixel with the disparity axis on
// its lanes (coalesced 64*C*4-byte taps) and walks the arm in the reference's order
// (t = -arm .. +arm).  Pass 1 goes cur -> tmp, pass 2 tmp -> cur with the division by the
// uint16 support count, exactly like vec_cost_tmp_[0/1] and cost_aggr_ per plane.
#include "smt_common.h"
#include <type_traits>
#include <new>

namespace {

constexpr int NT = 256;

// One pixel's B, G, R as the low three bytes of an aligned word (k_ca_pack, once per image): the arm walk then
// costs one coalesced dword load per step.  Three byte loads per step, or one unaligned dword at a 3-byte stride,
// keep the kernel on the texture addresser (0.12-0.15 ms at 1280x720 against 0.0x with the packed copy).
__global__ void __launch_bounds__(NT) k_ca_pack(const uint8_t *__restrict__ img, int n, uint32_t *__restrict__ pix)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= n) return;
    const uint8_t *q = img + (size_t)p * 3;
    pix[p] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
}

__device__ __forceinline__ int col_dist(uint32_t a, uint32_t b)
{
    const int d0 = abs((int)(a & 255u) - (int)(b & 255u)), d1 = abs((int)((a >> 8) & 255u) - (int)((b >> 8) & 255u)),
              d2 = abs((int)(a >> 16) - (int)(b >> 16));
    return max(d0, max(d1, d2));                                          // ColorDist, h:78-80
}

// The walk's tests depend on the pixel n steps out, the one before it and the anchor -- never on an earlier
// test -- so the pixels are fetched eight steps at a time with independent loads and the sequential rule is then
// applied in registers (a load per step with the break deciding the next one costs a memory latency per step:
// 120-150 us at 1280x720 against the few microseconds the loads themselves need).
__device__ int ca_arm(const uint32_t *pix, int W, int H, int x, int y, int dx, int dy, int L1, int L2,
                      int t1, int t2)
{
    const uint32_t c0 = pix[(size_t)y * W + x];
    uint32_t prev = c0;
    const int lim = L1 < 255 ? L1 : 255;                                  // MAX_ARM_LENGTH
    int len = 0;
    for (int n0 = 0; n0 < lim; n0 += 8) {
        uint32_t c[8];
        bool in[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int xn = x + dx * (n0 + k + 1), yn = y + dy * (n0 + k + 1);
            in[k] = n0 + k < lim && xn >= 0 && xn < W && yn >= 0 && yn < H;          // :154-163
            c[k] = in[k] ? pix[(size_t)yn * W + xn] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int n = n0 + k;
            if (!in[k]) return len;
            const int d1 = col_dist(c[k], c0);
            if (d1 >= t1) return len;                                     // :169-172
            if (n > 0 && col_dist(c[k], prev) >= t1) return len;          // :175-180
            if (n + 1 > L2 && d1 >= t2) return len;                       // :183-187
            len++;
            prev = c[k];
        }
    }
    return len;
}

__global__ void __launch_bounds__(NT) k_ca_arms(const uint32_t *__restrict__ pix, int W, int H, int L1, int L2,
                                                int t1, int t2, uint8_t *__restrict__ arms)
{
    // blockIdx.y = direction: four times the threads, each with one walk of dependent byte loads instead of four
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= W * H) return;
    const int y = p / W, x = p - y * W;
    const int dir = blockIdx.y;                             // 0 left, 1 right, 2 up, 3 down: the byte order of the map
    const int dx = dir == 0 ? -1 : (dir == 1 ? 1 : 0), dy = dir == 2 ? -1 : (dir == 3 ? 1 : 0);
    arms[(size_t)p * 4 + dir] = (uint8_t)ca_arm(pix, W, H, x, y, dx, dy, L1, L2, t1, t2);
}

// cnt[0]: horizontal first (pass-1 = L+R+1, pass-2 sums those along the vertical arm);
// cnt[1]: vertical first.  Stored as uint16 like the reference's vectors.
__global__ void __launch_bounds__(NT) k_ca_counts(const uint8_t *__restrict__ arms, int W, int H,
                                                  uint16_t *__restrict__ cnt0, uint16_t *__restrict__ cnt1)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= W * H) return;
    const int y = p / W, x = p - y * W;
    const uchar4 a = reinterpret_cast<const uchar4 *>(arms)[p];
    int c0 = 0, c1 = 0;
    for (int t = -(int)a.z; t <= (int)a.w; t++) {
        const uchar4 q = reinterpret_cast<const uchar4 *>(arms)[(y + t) * W + x];
        c0 += (uint16_t)((int)q.x + (int)q.y + 1);
    }
    for (int t = -(int)a.x; t <= (int)a.y; t++) {
        const uchar4 q = reinterpret_cast<const uchar4 *>(arms)[y * W + x + t];
        c1 += (uint16_t)((int)q.z + (int)q.w + 1);
    }
    cnt0[p] = (uint16_t)c0;
    cnt1[p] = (uint16_t)c1;
}

// one wave per pixel; HORIZ: taps along x with (left,right) else along y with (top,bottom)
template <int C, bool HORIZ, bool FINAL>
__global__ void __launch_bounds__(NT) k_ca_pass(const float *__restrict__ src, float *__restrict__ dst, int W, int H,
                                                int D, const uint8_t *__restrict__ arms,
                                                const uint16_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * (NT / 64) + wv;
    if (p >= W * H) return;
    const uchar4 a = reinterpret_cast<const uchar4 *>(arms)[p];
    const int lo = HORIZ ? -(int)a.x : -(int)a.z;
    const int hi = HORIZ ? (int)a.y : (int)a.w;
    const long stride = HORIZ ? 1 : W;
    const int dl = lane * C;
    float acc[C];
#pragma unroll
    for (int k = 0; k < C; k++) acc[k] = 0.0f;
    for (int t = lo; t <= hi; t++) {
        const float *s = src + ((long)p + t * stride) * D + dl;
#pragma unroll
        for (int k = 0; k < C; k++)
            if (dl + k < D) acc[k] += s[k];
    }
    float *o = dst + (long)p * D + dl;
    const float n = FINAL ? (float)cnt[p] : 1.0f;
#pragma unroll
    for (int k = 0; k < C; k++)
        if (dl + k < D) o[k] = FINAL ? acc[k] / n : acc[k];
}

// ---- shared-tap pass (default) --------------------------------------------------------------------------
// k_ca_pass above loads every tap of every pixel (arm sum ~20 taps x 64*C*4 bytes through the L1 per pixel and
// pass: 1.4-2.1 ms per pass at 1280x720x128).  The sums are one-dimensional, so the register-sharing scheme of
// the rectangle aggregation (crossarm.hip, k_aggregate_multi) applies in its simplest form: one wave owns 16
// consecutive pixels ALONG the pass axis, walks the union of their tap intervals [q - a_q, q + b_q] once in
// increasing order -- the reference's order for every pixel (t = -arm .. +arm, cross_aggregator.cpp:342-392) --
// and adds each tap to all 16 accumulators as fma(x, f, acc), f = 1.0f for the pixels whose interval holds it
// and 0.0f for the others (fma(x, 1, acc) is the reference's acc + x, fma(x, 0, acc) is acc bit for bit for
// finite x; a pixel whose result is NaN is recomputed by the plain walk, which is the reference's answer in every
// case).  The membership masks of 64 positions are built at once (lane = position); a tap's 16 flags are two rows
// of the 256 x 8 table, fetched one tap ahead.  16 + a + b taps instead of 16 x (a + b + 1).
// Measured at 1280x720x128 (profiles/r2f): non-final passes 0.19 ms = the time of reading and writing the volume
// once; final passes (IEEE division per output) 0.25 ms.
constexpr int CAP = 16;
typedef float caf2 __attribute__((ext_vector_type(2)));
typedef int cai2 __attribute__((ext_vector_type(2)));
typedef int cai3 __attribute__((ext_vector_type(3)));
typedef int cai4 __attribute__((ext_vector_type(4)));

template <int C, bool HORIZ, bool FINAL, bool FULL>
__global__ void __launch_bounds__(NT) k_ca_pass2(const float *__restrict__ src, float *__restrict__ dst, int W, int H,
                                                 int D, const uint8_t *__restrict__ arms, const uint16_t *__restrict__ cnt,
                                                 const float *__restrict__ member)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // HORIZ: workgroups are dealt to the 8 XCDs round-robin, and the tiles of neighbouring workgroups share their
    // halo taps; giving every XCD one contiguous range of workgroups (rows) keeps that sharing inside one L2
    // (measured: 1.8x the input fetched across the fabric with the plain order).  The vertical pass already has
    // its vertical neighbours on one XCD (W / 4 workgroups per band, a multiple of 8 for the usual widths).
    long wg = blockIdx.x;
    if (HORIZ) {
        const long per = gridDim.x / 8;                     // gridDim.x is a multiple of 8 (host); surplus workgroups leave at y0 >= H
        wg = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    const long wid = wg * (NT / 64) + wv;
    int x0, y0, npx;
    if (HORIZ) {
        const int nxb = (W + CAP - 1) / CAP;
        y0 = (int)(wid / nxb); x0 = (int)(wid % nxb) * CAP;
        if (y0 >= H) return;
        npx = min(CAP, W - x0);
    } else {
        const int nyb = (H + CAP - 1) / CAP;               // consecutive waves = consecutive columns of one row band
        const long yb = wid / W;
        x0 = (int)(wid % W); y0 = (int)yb * CAP;
        if (yb >= nyb) return;
        npx = min(CAP, H - y0);
    }
    const long stride = HORIZ ? 1 : W;
    const long p0 = (long)y0 * W + x0;
    // lane q: tap interval of pixel q relative to the first pixel; empty past the tile
    int lo = 1, hi = 0;
    float myn = 1.0f;                                       // FINAL: lane q fetches pixel q's support count here, long before it is used
    if (lane < npx) {
        const uchar4 a = reinterpret_cast<const uchar4 *>(arms)[p0 + lane * stride];
        lo = lane - (HORIZ ? (int)a.x : (int)a.z);
        hi = lane + (HORIZ ? (int)a.y : (int)a.w);
        if (FINAL) myn = (float)cnt[p0 + lane * stride];
    }
    int umin = 0, umax = npx - 1;                           // every pixel holds its own position
#pragma unroll
    for (int q = 0; q < CAP; q++) {
        if (q < npx) {
            umin = min(umin, __builtin_amdgcn_readlane(lo, q));
            umax = max(umax, __builtin_amdgcn_readlane(hi, q));
        }
    }
    const int dl = lane * C;
    caf2 acc[CAP / 2][C];
#pragma unroll
    for (int j = 0; j < CAP / 2; j++)
#pragma unroll
        for (int k = 0; k < C; k++) acc[j][k] = caf2{0.0f, 0.0f};
    auto ld = [&](long pix, float (&x)[C]) {
        const float *sp = src + pix * D + dl;
        if (FULL) {
            if (C == 1) x[0] = sp[0];
            else if (C == 2) { const cai2 v = *reinterpret_cast<const cai2 *>(sp); x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y); }
            else if (C == 3) {
                x[0] = sp[0]; x[C > 1 ? 1 : 0] = sp[C > 1 ? 1 : 0]; x[C > 2 ? 2 : 0] = sp[C > 2 ? 2 : 0];
            } else {
                const cai4 v = *reinterpret_cast<const cai4 *>(sp);
                x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y);
                x[C > 2 ? 2 : 0] = __int_as_float(v.z); x[C > 3 ? 3 : 0] = __int_as_float(v.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) x[k] = (dl + k < D) ? sp[k] : 0.0f;
        }
    };
    // Tap loop.  Taps are umin .. umax in increasing order; the tap pointer is wave-uniform and advances by one
    // pixel step per tap (two scalar adds -- no per-tap 64-bit multiply); the 16 flags of a tap (two rows of the
    // table) are fetched one tap AHEAD of the FMAs that use them into the other half of a two-deep SGPR buffer,
    // so the scalar-cache latency sits behind the previous tap's FMAs (k_aggregate_multi, SKIP == 2).
    typedef float caf8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) caf8 *cflag_p;
    const cflag_p mtab = (cflag_p)(member);
    auto load_flags = [&](unsigned m, caf8 (&F)[2]) {
        F[0] = mtab[m & 255u];
        F[1] = mtab[(m >> 8) & 255u];
    };
    auto fma_flagged = [&](unsigned m, const caf8 (&F)[2], const float (&x)[C]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            auto pair = [&](int j) {
                const caf2 fl = caf2{F[h][2 * j], F[h][2 * j + 1]};
#pragma unroll
                for (int c = 0; c < C; c++)
                    acc[4 * h + j][c] = __builtin_elementwise_fma(caf2{x[c], x[c]}, fl, acc[4 * h + j][c]);
            };
            pair(0); pair(1); pair(2); pair(3);            // no group skip here: measured equal, and the skip's compares and branches load the scalar issue port
        }
    };
    const long step = stride * D;                           // floats between two taps
    typedef const __attribute__((address_space(1))) float *gfloat_p;
    gfloat_p tp;                                            // wave-uniform tap pointer (global address space: no flat loads)
    {
        const uint64_t v = (uint64_t)(src + (p0 + (long)umin * stride) * D);
        tp = (gfloat_p)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    }
    // membership masks of the positions umin + lane and umin + 64 + lane (the union spans at most 16 + 2 * 34 + ...
    // positions; longer arms take further chunks on the fly): bit q = pixel q's interval holds the position
    auto chunk_masks = [&](int first) {
        const int pos = first + lane;
        unsigned mk = 0;
#pragma unroll
        for (int q = 0; q < CAP; q++) {
            const int lq = __builtin_amdgcn_readlane(lo, q), hq = __builtin_amdgcn_readlane(hi, q);
            mk |= (lq <= pos && pos <= hq) ? (1u << q) : 0u;
        }
        return mk;
    };
    unsigned maskv = chunk_masks(umin);
    int mbase = umin;                                       // position of lane 0 of maskv
    auto ldp = [&](gfloat_p base, float (&x)[C]) {
        gfloat_p sp = base + dl;
        if (FULL) {
            if (C == 1) x[0] = sp[0];
            else if (C == 2) { const cai2 v = *(const __attribute__((address_space(1))) cai2 *)(sp); x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y); }
            else if (C == 3) {
                x[0] = sp[0]; x[C > 1 ? 1 : 0] = sp[C > 1 ? 1 : 0]; x[C > 2 ? 2 : 0] = sp[C > 2 ? 2 : 0];
            } else {
                const cai4 v = *(const __attribute__((address_space(1))) cai4 *)(sp);
                x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y);
                x[C > 2 ? 2 : 0] = __int_as_float(v.z); x[C > 3 ? 3 : 0] = __int_as_float(v.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) x[k] = (dl + k < D) ? sp[k] : 0.0f;
        }
    };
    auto group = [&](auto gtag, int u0) {
        constexpr int GG = decltype(gtag)::value;
        unsigned m[GG];
        float x[GG][C];
#pragma unroll
        for (int k = 0; k < GG; k++) {
            m[k] = (unsigned)__builtin_amdgcn_readlane((int)maskv, u0 + k - mbase);
            ldp(tp, x[k]);
            tp += step;
        }
        caf8 F[2][2];
        load_flags(m[0], F[0]);
#pragma unroll
        for (int k = 0; k < GG; k++) {
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): this tap's flags are in
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < GG) load_flags(m[k + 1], F[(k + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            fma_flagged(m[k], F[k & 1], x[k]);
        }
    };
    constexpr int G = C <= 2 ? 8 : 4;                       // taps loaded ahead of their adds (64 % G == 0: a group never straddles a mask chunk)
    int u0 = umin;
    while (u0 <= umax) {
        const int uend = min(umax + 1, mbase + 64);         // taps of this mask chunk
        for (; u0 + G <= uend; u0 += G) group(std::integral_constant<int, G>{}, u0);
        if (G == 8 && u0 + 4 <= uend) { group(std::integral_constant<int, 4>{}, u0); u0 += 4; }
        if (u0 + 2 <= uend) { group(std::integral_constant<int, 2>{}, u0); u0 += 2; }
        if (u0 < uend) { group(std::integral_constant<int, 1>{}, u0); u0 += 1; }
        if (u0 <= umax) { mbase += 64; maskv = chunk_masks(mbase); }
    }
    // finish: plain walk for pixels whose sum is NaN (a non-finite tap times a zero flag, or a genuine NaN)
#pragma unroll
    for (int q = 0; q < CAP; q++) {
        if (q >= npx) continue;
        float a[C];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < C; k++) {
            a[k] = (q & 1) ? acc[q / 2][k].y : acc[q / 2][k].x;
            bad = bad || (a[k] != a[k]);
        }
        const long p = p0 + q * stride;
        if (__ballot(bad)) {
            const int qlo = __builtin_amdgcn_readlane(lo, q), qhi = __builtin_amdgcn_readlane(hi, q);
#pragma unroll
            for (int k = 0; k < C; k++) a[k] = 0.0f;
            for (int u = qlo; u <= qhi; u++) {
                float x[C];
                ld(p0 + u * stride, x);
#pragma unroll
                for (int k = 0; k < C; k++) a[k] = a[k] + x[k];
            }
        }
        float *o = dst + p * D + dl;
        if constexpr (FINAL) {
            // cost / count (cross_aggregator.cpp:389, float over uint16): the correctly rounded quotient without the division sequence
            // (wave_quotient, smt_common.h; k_ca_pass keeps the IEEE division as the independent formulation)
            float qv[C];
            wave_quotient<C, FULL>(a, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myn), q)), dl, D, qv);
#pragma unroll
            for (int k = 0; k < C; k++)
                if (FULL || dl + k < D) o[k] = qv[k];
        } else {
#pragma unroll
            for (int k = 0; k < C; k++)
                if (FULL || dl + k < D) o[k] = a[k];
        }
    }
}

}  // namespace

struct smt_crossagg {
    int device;
    int W, H, D;
    int L1, L2, t1, t2;
    hipStream_t stream;
    float *cur, *tmp;
    uint8_t *arms;
    uint32_t *pix;       // [H][W] packed B | G << 8 | R << 16 of the current image
    uint16_t *cnt[2];
    float *member;       // 256 x 8 membership flags (row m: 1.0f where bit q of m is set)
    int impl;            // 2: shared-tap passes (default), 1: one pixel per wave (first formulation)
};

SMT_API int smt_crossagg_create(int W, int H, int D, smt_crossagg **out)
{
    if (!out) return SMT_ERR_ARG;
    if ((long)W * H <= 0 || W <= 0 || H <= 0 || D <= 0 || D > 256) return SMT_ERR_ARG;   // Initialize returns false (:28-31)
    smt_crossagg *h = new (std::nothrow) smt_crossagg();
    if (!h) return SMT_ERR_ALLOC;
    h->device = smt_current_device();
    h->W = W; h->H = H; h->D = D;
    h->L1 = 34; h->L2 = 17; h->t1 = 20; h->t2 = 6;                       // adcensus_types.h:69-70
    const size_t N = (size_t)W * H;
    int rc = smt_malloc((void **)&h->cur, N * D * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->tmp, N * D * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->arms, N * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->pix, N * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->cnt[0], N * 2);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->cnt[1], N * 2);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->member, 256 * 8 * 4);
    if (rc != SMT_OK) { smt_crossagg_destroy(h); return rc; }
    {
        float tab[256 * 8];
        for (int m = 0; m < 256; m++)
            for (int q = 0; q < 8; q++) tab[m * 8 + q] = ((m >> q) & 1) ? 1.0f : 0.0f;
        if (hipMemcpy(h->member, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) { smt_crossagg_destroy(h); return SMT_ERR_HIP; }
    }
    h->impl = 2;
    *out = h;
    return SMT_OK;
}

SMT_API int smt_crossagg_create_on(int device, int W, int H, int D, smt_crossagg **out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(device);
    return smt_crossagg_create(W, H, D, out);
}

SMT_API int smt_crossagg_destroy(smt_crossagg *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    (void)hipFree(h->cur); (void)hipFree(h->tmp); (void)hipFree(h->arms); (void)hipFree(h->pix);
    (void)hipFree(h->cnt[0]); (void)hipFree(h->cnt[1]);
    (void)hipFree(h->member);
    delete h;
    return SMT_OK;
}

SMT_API int smt_crossagg_set_stream(smt_crossagg *h, void *s)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->stream = smt_stream(s);
    return SMT_OK;
}

SMT_API int smt_crossagg_set_params(smt_crossagg *h, int L1, int L2, int t1, int t2)
{
    if (!h || L1 < 0) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->L1 = L1; h->L2 = L2; h->t1 = t1; h->t2 = t2;
    return SMT_OK;
}

template <int C, bool FULL>
static void ca_iter2(smt_crossagg *h, bool hfirst, const float *in)
{
    const long nh = (long)h->H * ((h->W + CAP - 1) / CAP), nv = (long)h->W * ((h->H + CAP - 1) / CAP);   // waves per pass
    dim3 gh((unsigned)(((nh + 3) / 4 + 7) / 8 * 8)), gv((unsigned)((nv + 3) / 4));   // gh: whole rounds of the 8 XCDs (k_ca_pass2's order)
    const uint16_t *cnt = h->cnt[hfirst ? 0 : 1];
    if (hfirst) {
        hipLaunchKernelGGL((k_ca_pass2<C, true, false, FULL>), gh, dim3(NT), 0, h->stream, in, h->tmp, h->W, h->H, h->D, h->arms, cnt, h->member);
        hipLaunchKernelGGL((k_ca_pass2<C, false, true, FULL>), gv, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H, h->D, h->arms, cnt, h->member);
    } else {
        hipLaunchKernelGGL((k_ca_pass2<C, false, false, FULL>), gv, dim3(NT), 0, h->stream, in, h->tmp, h->W, h->H, h->D, h->arms, cnt, h->member);
        hipLaunchKernelGGL((k_ca_pass2<C, true, true, FULL>), gh, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H, h->D, h->arms, cnt, h->member);
    }
}

template <int C>
// in: the volume the iteration starts from -- cost_init for the first one (cross_aggregator.cpp:108 copies it into
// the working volume first; reading it in place saves that pass), h->cur afterwards
static void ca_iter(smt_crossagg *h, bool hfirst, const float *in)
{
    if (h->impl == 2) {
        if (h->D == 64 * C) ca_iter2<C, true>(h, hfirst, in); else ca_iter2<C, false>(h, hfirst, in);
        return;
    }
    const int N = h->W * h->H;
    dim3 grid((N + 3) / 4);
    const uint16_t *cnt = h->cnt[hfirst ? 0 : 1];
    if (hfirst) {
        hipLaunchKernelGGL((k_ca_pass<C, true, false>), grid, dim3(NT), 0, h->stream, in, h->tmp, h->W, h->H,
                           h->D, h->arms, cnt);
        hipLaunchKernelGGL((k_ca_pass<C, false, true>), grid, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H,
                           h->D, h->arms, cnt);
    } else {
        hipLaunchKernelGGL((k_ca_pass<C, false, false>), grid, dim3(NT), 0, h->stream, in, h->tmp, h->W, h->H,
                           h->D, h->arms, cnt);
        hipLaunchKernelGGL((k_ca_pass<C, true, true>), grid, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H,
                           h->D, h->arms, cnt);
    }
}

SMT_API int smt_crossagg_aggregate(smt_crossagg *h, const uint8_t *img, const float *cost_init, int iters)
{
    if (!h || !img || !cost_init || iters < 0) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const int N = h->W * h->H;
    hipLaunchKernelGGL(k_ca_pack, dim3((N + NT - 1) / NT), dim3(NT), 0, h->stream, img, N, h->pix);
    hipLaunchKernelGGL(k_ca_arms, dim3((N + NT - 1) / NT, 4), dim3(NT), 0, h->stream, h->pix, h->W, h->H, h->L1, h->L2,
                       h->t1, h->t2, h->arms);                            // BuildArms :76-86
    hipLaunchKernelGGL(k_ca_counts, dim3((N + NT - 1) / NT), dim3(NT), 0, h->stream, h->arms, h->W, h->H,
                       h->cnt[0], h->cnt[1]);                             // ComputeSupPixelCount
    if (iters == 0 && cost_init != h->cur)
        SMT_HIP(hipMemcpyAsync(h->cur, cost_init, (size_t)N * h->D * 4, hipMemcpyDeviceToDevice, h->stream)); // :108
    bool hfirst = true;
    for (int k = 0; k < iters; k++) {                                     // :111-117
        const float *in = k == 0 ? cost_init : h->cur;                    // first iteration: cost_init read in place
        switch ((h->D + 63) / 64) {
        case 1: ca_iter<1>(h, hfirst, in); break;
        case 2: ca_iter<2>(h, hfirst, in); break;
        case 3: ca_iter<3>(h, hfirst, in); break;
        default: ca_iter<4>(h, hfirst, in); break;
        }
        hfirst = !hfirst;
    }
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_crossagg_set_impl(smt_crossagg *h, int impl)
{
    if (!h || (impl != 1 && impl != 2)) return SMT_ERR_ARG;
    h->impl = impl;
    return SMT_OK;
}

SMT_API int smt_crossagg_cost(smt_crossagg *h, float **cost)
{
    if (!h || !cost) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    *cost = h->cur;
    return SMT_OK;
}

SMT_API int smt_crossagg_arms(smt_crossagg *h, uint8_t **arms)
{
    if (!h || !arms) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    *arms = h->arms;
    return SMT_OK;
}

// ---- ADCensusOption (CBLSM/adcensus_types.h:45-75) and the one caller shape the reference holds ------------
SMT_API void smt_adcensus_option_default(smt_adcensus_option *o)
{
    if (!o) return;
    o->min_disparity = 0; o->max_disparity = 64;          // adcensus_types.h:69
    o->lambda_ad = 10; o->lambda_census = 30;
    o->cross_L1 = 34; o->cross_L2 = 17; o->cross_t1 = 20; o->cross_t2 = 6;
    o->so_p1 = 1.0f; o->so_p2 = 3.0f; o->so_tso = 15;
    o->irv_ts = 20; o->irv_th = 0.4f;
    o->lrcheck_thres = 1.0f;
    o->do_lr_check = 1; o->do_filling = 1; o->do_discontinuity_adjustment = 0;
}

SMT_API int smt_adcensus_option_aggregate(const smt_adcensus_option *o, const uint8_t *bytes_left, const float *cost_init,
                                          int W, int H, int num_iters, float *cost_out, float *disp, void *stream)
{
    if (!o || !bytes_left || !cost_init || !cost_out || num_iters < 0) return SMT_ERR_ARG;
    const int D = o->max_disparity - o->min_disparity;
    smt_crossagg *h = nullptr;
    int rc = smt_crossagg_create(W, H, D, &h);                                           // Initialize(col, row, 0, dispRange), CBLSM.cpp:139
    if (rc != SMT_OK) return rc;
    rc = smt_crossagg_set_stream(h, stream);
    if (rc == SMT_OK) rc = smt_crossagg_set_params(h, o->cross_L1, o->cross_L2, o->cross_t1, o->cross_t2);   // :141
    if (rc == SMT_OK) rc = smt_crossagg_aggregate(h, bytes_left, cost_init, num_iters);  // SetData + Aggregate, :140, :142
    if (rc == SMT_OK && hipMemcpyAsync(cost_out, h->cur, (size_t)W * H * D * 4, hipMemcpyDeviceToDevice, smt_stream(stream)) != hipSuccess)
        rc = SMT_ERR_HIP;                                                                // get_cost_ptr, :143
    if (rc == SMT_OK && disp) rc = smt_wta(cost_out, H, W, D, disp, stream);             // ComputeDispOringin, :152
    if (hipStreamSynchronize(smt_stream(stream)) != hipSuccess && rc == SMT_OK) rc = SMT_ERR_HIP;
    smt_crossagg_destroy(h);
    return rc;
}
