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
#include <new>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ int col_dist(const uint8_t *a, const uint8_t *b)
{
    const int d0 = abs((int)a[0] - (int)b[0]), d1 = abs((int)a[1] - (int)b[1]), d2 = abs((int)a[2] - (int)b[2]);
    return max(d0, max(d1, d2));                                          // ColorDist, h:78-80
}

__device__ int ca_arm(const uint8_t *img, int W, int H, int x, int y, int dx, int dy, int L1, int L2,
                      int t1, int t2)
{
    const uint8_t *c0 = img + ((size_t)y * W + x) * 3;
    const uint8_t *prev = c0;
    const int lim = L1 < 255 ? L1 : 255;                                  // MAX_ARM_LENGTH
    int xn = x + dx, yn = y + dy, len = 0;
    for (int n = 0; n < lim; n++) {
        if (xn < 0 || xn == W || yn < 0 || yn == H) break;                // :154-163
        const uint8_t *c = img + ((size_t)yn * W + xn) * 3;
        const int d1 = col_dist(c, c0);
        if (d1 >= t1) break;                                              // :169-172
        if (n > 0 && col_dist(c, prev) >= t1) break;                      // :175-180
        if (n + 1 > L2 && d1 >= t2) break;                                // :183-187
        len++;
        prev = c; xn += dx; yn += dy;
    }
    return len;
}

__global__ void __launch_bounds__(NT) k_ca_arms(const uint8_t *__restrict__ img, int W, int H, int L1, int L2,
                                                int t1, int t2, uint8_t *__restrict__ arms)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= W * H) return;
    const int y = p / W, x = p - y * W;
    uchar4 a;
    a.x = (uint8_t)ca_arm(img, W, H, x, y, -1, 0, L1, L2, t1, t2);
    a.y = (uint8_t)ca_arm(img, W, H, x, y, +1, 0, L1, L2, t1, t2);
    a.z = (uint8_t)ca_arm(img, W, H, x, y, 0, -1, L1, L2, t1, t2);
    a.w = (uint8_t)ca_arm(img, W, H, x, y, 0, +1, L1, L2, t1, t2);
    reinterpret_cast<uchar4 *>(arms)[p] = a;
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
// case).  The membership of a tap is one ballot; its 16 flags are two rows of the 256 x 8 table; 4-pixel groups
// without a member are skipped.  16 + a + b taps instead of 16 x (a + b + 1).
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
    const long wid = (long)blockIdx.x * (NT / 64) + wv;
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
    if (lane < npx) {
        const uchar4 a = reinterpret_cast<const uchar4 *>(arms)[p0 + lane * stride];
        lo = lane - (HORIZ ? (int)a.x : (int)a.z);
        hi = lane + (HORIZ ? (int)a.y : (int)a.w);
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
    auto add_flagged = [&](unsigned m, const float (&x)[C]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned mb = (m >> (8 * h)) & 255u;
            const float *f = member + mb * 8u;
            auto pair = [&](int j) {
                const caf2 fl = caf2{f[2 * j], f[2 * j + 1]};
#pragma unroll
                for (int c = 0; c < C; c++)
                    acc[4 * h + j][c] = __builtin_elementwise_fma(caf2{x[c], x[c]}, fl, acc[4 * h + j][c]);
            };
            if (mb & 0x0fu) { pair(0); pair(1); }
            if (mb & 0xf0u) { pair(2); pair(3); }
        }
    };
    constexpr int G = 4;                                    // taps loaded ahead of their adds
    for (int u0 = umin; u0 <= umax; u0 += G) {
        unsigned m[G];
        float x[G][C];
#pragma unroll
        for (int k = 0; k < G; k++) {
            const int u = min(u0 + k, umax);                // a short last group repeats its last tap with no member
            m[k] = (u0 + k <= umax) ? (unsigned)__ballot(lo <= u && u <= hi) : 0u;
            ld(p0 + u * stride, x[k]);
        }
#pragma unroll
        for (int k = 0; k < G; k++) add_flagged(m[k], x[k]);
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
        const float n = FINAL ? (float)cnt[p] : 1.0f;
#pragma unroll
        for (int k = 0; k < C; k++)
            if (FULL || dl + k < D) o[k] = FINAL ? a[k] / n : a[k];
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
    (void)hipFree(h->cur); (void)hipFree(h->tmp); (void)hipFree(h->arms);
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
static void ca_iter2(smt_crossagg *h, bool hfirst)
{
    const long nh = (long)h->H * ((h->W + CAP - 1) / CAP), nv = (long)h->W * ((h->H + CAP - 1) / CAP);   // waves per pass
    dim3 gh((unsigned)((nh + 3) / 4)), gv((unsigned)((nv + 3) / 4));
    const uint16_t *cnt = h->cnt[hfirst ? 0 : 1];
    if (hfirst) {
        hipLaunchKernelGGL((k_ca_pass2<C, true, false, FULL>), gh, dim3(NT), 0, h->stream, h->cur, h->tmp, h->W, h->H, h->D, h->arms, cnt, h->member);
        hipLaunchKernelGGL((k_ca_pass2<C, false, true, FULL>), gv, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H, h->D, h->arms, cnt, h->member);
    } else {
        hipLaunchKernelGGL((k_ca_pass2<C, false, false, FULL>), gv, dim3(NT), 0, h->stream, h->cur, h->tmp, h->W, h->H, h->D, h->arms, cnt, h->member);
        hipLaunchKernelGGL((k_ca_pass2<C, true, true, FULL>), gh, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H, h->D, h->arms, cnt, h->member);
    }
}

template <int C>
static void ca_iter(smt_crossagg *h, bool hfirst)
{
    if (h->impl == 2) {
        if (h->D == 64 * C) ca_iter2<C, true>(h, hfirst); else ca_iter2<C, false>(h, hfirst);
        return;
    }
    const int N = h->W * h->H;
    dim3 grid((N + 3) / 4);
    const uint16_t *cnt = h->cnt[hfirst ? 0 : 1];
    if (hfirst) {
        hipLaunchKernelGGL((k_ca_pass<C, true, false>), grid, dim3(NT), 0, h->stream, h->cur, h->tmp, h->W, h->H,
                           h->D, h->arms, cnt);
        hipLaunchKernelGGL((k_ca_pass<C, false, true>), grid, dim3(NT), 0, h->stream, h->tmp, h->cur, h->W, h->H,
                           h->D, h->arms, cnt);
    } else {
        hipLaunchKernelGGL((k_ca_pass<C, false, false>), grid, dim3(NT), 0, h->stream, h->cur, h->tmp, h->W, h->H,
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
    hipLaunchKernelGGL(k_ca_arms, dim3((N + NT - 1) / NT), dim3(NT), 0, h->stream, img, h->W, h->H, h->L1, h->L2,
                       h->t1, h->t2, h->arms);                            // BuildArms :76-86
    hipLaunchKernelGGL(k_ca_counts, dim3((N + NT - 1) / NT), dim3(NT), 0, h->stream, h->arms, h->W, h->H,
                       h->cnt[0], h->cnt[1]);                             // ComputeSupPixelCount
    SMT_HIP(hipMemcpyAsync(h->cur, cost_init, (size_t)N * h->D * 4, hipMemcpyDeviceToDevice, h->stream)); // :108
    bool hfirst = true;
    for (int k = 0; k < iters; k++) {                                     // :111-117
        switch ((h->D + 63) / 64) {
        case 1: ca_iter<1>(h, hfirst); break;
        case 2: ca_iter<2>(h, hfirst); break;
        case 3: ca_iter<3>(h, hfirst); break;
        default: ca_iter<4>(h, hfirst); break;
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
