// 4-direction scanline optimiser -- replaces class ScanlineOptimizer
// (AD-CensusV1/ScanlineOptimizer.h).
//
// Every scanline (a row for the left/right passes, a column for the up/down passes) is an
// independent recurrence over pixels; inside a pixel the D hypotheses are independent
// except for the minimum of the previous path vector.  One wavefront owns one scanline:
// lane l keeps C = ceil(D/64) consecutive entries of the previous path vector in
// registers, gets last[d-1] / last[d+1] from its neighbour lanes with wave_shr / wave_shl
// DPP moves (pad value 65535 at the ends, ScanlineOptimizer.h:151) and reduces the new
// minimum with DPP; cost rows are prefetched PF steps ahead so the sequential chain only
// sees register data.  The reference's quirks are kept: the up/down passes use last[d]
// for the "d-1" term (:238), step the gray pointer by one element instead of one row
// (:221, :250) and never update grayLast (:210).
//
// The reference stores four path volumes and then adds them ((left+right)+up)+down
// (:124).  Here the left and right passes run CONCURRENTLY in one launch (2H scanlines in
// flight instead of H) into the output volume and one scratch volume; the up pass then
// writes (left + right) + up and the down pass adds itself and can fuse
// ScanlineOptimizer::WTA (:40-64).  Same association order, 44 bytes of HBM traffic per
// hypothesis in total (8 + 8 + 16 + 12).
#include "smt_common.h"
#include <new>

namespace {

#ifndef SMT_SCAN_PF
#define SMT_SCAN_PF 8
#endif
#ifndef SMT_SCAN_PF_H
#define SMT_SCAN_PF_H SMT_SCAN_PF
#endif
constexpr int PF_V = SMT_SCAN_PF, PF_H = SMT_SCAN_PF_H;   // prefetch depth (scan steps) of the vertical / horizontal passes: loads are issued PF steps ahead so the
                                  // in-order vmcnt wait never lands behind the step's own stores
constexpr float PAD = 65535.0f;   // 0xffff as float (:151, :162, :169)

template <int CTRL>
__device__ __forceinline__ float dpp_shift_f32(float v, float fill)
{
    // lanes with no source keep `fill`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL,
                                                      0xF, 0xF, false));
}

// wave-wide float minimum via DPP; ordinary-number inputs (no NaN).
__device__ __forceinline__ float wave_min_f32_dpp(float v)
{
    const float big = INFINITY;
    v = fminf(v, dpp_shift_f32<0xB1>(v, big));
    v = fminf(v, dpp_shift_f32<0x4E>(v, big));
    v = fminf(v, dpp_shift_f32<0x141>(v, big));
    v = fminf(v, dpp_shift_f32<0x140>(v, big));
    float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(big), __float_as_int(v), 0x142, 0xA, 0xF, false));
    v = fminf(v, t);
    t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(big), __float_as_int(v), 0x143, 0xC, 0xF, false));
    v = fminf(v, t);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct ScanArgs {
    const float *cost;   // [H][W][D]
    const float *gray;   // [H][W]
    float *out;          // [H][W][D]
    const float *add2;   // MODE 3: second addend volume (right path), else unused
    float *out_b;        // k_scan_lr: destination of the right->left pass
    float *disp;         // fused WTA of the accumulated volume (last pass) or null
    int H, W, D;
    float p1, p2;
};

template <int C> struct vecf { float v[C]; };          // C = 5..8 (D > 256)
template <> struct vecf<1> { float v[1]; };
template <> struct __attribute__((aligned(8))) vecf<2> { float v[2]; };
template <> struct vecf<3> { float v[3]; };
template <> struct __attribute__((aligned(16))) vecf<4> { float v[4]; };

// Cache policy of the volume traffic (compile-time, A/B builds): SMT_SCAN_NT & 1 = non-temporal loads, & 2 =
// non-temporal stores.  Every byte a pass touches is used once per pass, so nothing is lost by not keeping it.
// Measured at 1080p x 192 (tools/scan_time.py, alternating processes on one box): 3.38 -> 3.23, 3.42 -> 3.27,
// 3.35 -> 3.34 ms for the three passes together with both: the default.
#ifndef SMT_SCAN_NT
#define SMT_SCAN_NT 3
#endif
// row of C consecutive floats per lane; FULL (D == 64*C): one unpredicated vector access
template <int C, bool FULL>
__device__ __forceinline__ void ld_row(const float *p, int dl, int D, float fill, float (&dst)[C])
{
    if (FULL) {
        if (SMT_SCAN_NT & 1) {
#pragma unroll
            for (int k = 0; k < C; k++) dst[k] = __builtin_nontemporal_load(p + k);
            return;
        }
        const vecf<C> x = *reinterpret_cast<const vecf<C> *>(p);
#pragma unroll
        for (int k = 0; k < C; k++) dst[k] = x.v[k];
    } else {
#pragma unroll
        for (int k = 0; k < C; k++) dst[k] = (dl + k < D) ? p[k] : fill;
    }
}
template <int C, bool FULL>
__device__ __forceinline__ void st_row(float *p, int dl, int D, const float (&src)[C])
{
    if (FULL) {
        if (SMT_SCAN_NT & 2) {
#pragma unroll
            for (int k = 0; k < C; k++) __builtin_nontemporal_store(src[k], p + k);
            return;
        }
        vecf<C> x;
#pragma unroll
        for (int k = 0; k < C; k++) x.v[k] = src[k];
        *reinterpret_cast<vecf<C> *>(p) = x;
    } else {
#pragma unroll
        for (int k = 0; k < C; k++)
            if (dl + k < D) p[k] = src[k];
    }
}

// PASS: 0 left->right, 1 right->left, 2 top->bottom, 3 bottom->top
// MODE: 0 out = path ; 1 out = out + path ; 2 out = out + path and fused WTA of the sum ;
//       3 out = (out + add2) + path
template <int C, int PASS, int MODE, bool FULL>
__device__ __forceinline__ void scan_body(const ScanArgs &a, float *out, int line)
{
    constexpr int PF = (PASS < 2) ? PF_H : PF_V;
    constexpr bool ACC = (MODE >= 1);
    constexpr bool ACC2 = (MODE == 3);
    constexpr bool WTA = (MODE == 2);
    const int lane = threadIdx.x & 63;
    constexpr bool HORIZ = (PASS < 2);
    const int H = a.H, W = a.W, D = a.D;
    const int nlines = HORIZ ? H : W;
    if (line >= nlines) return;
    const int nsteps = HORIZ ? W : H;                 // pixels on the line
    constexpr int dirn = (PASS == 0 || PASS == 2) ? 1 : -1;

    // pixel index (flat, in pixels) of step s:  start + s*pstride
    long start, pstride;
    if (HORIZ) { start = (long)line * W + (dirn > 0 ? 0 : W - 1); pstride = dirn; }
    else       { start = (long)(dirn > 0 ? 0 : H - 1) * W + line; pstride = (long)dirn * W; }
    // gray pointer: horizontal passes follow the pixel; vertical passes step by ONE element (:221,:250)
    const float *gp = a.gray + start;
    const int dl = lane * C;
    const float *cp = a.cost + dl;
    float *op = out + dl;
    const float *qp = a.add2 + dl;
    const float p1 = a.p1;

    auto wta_store = [&](const float (&res)[C], long pix) {
        const int wd = wave_wta<C, FULL>(res, dl, D);
        if (lane == 0) a.disp[pix] = (float)wd;
    };

    float last[C];
    // first pixel: path = cost (:153-155)
    ld_row<C, FULL>(cp + start * D, dl, D, PAD, last);
    {
        float res[C];
        if (ACC) {
            float o[C];
            ld_row<C, FULL>(op + start * D, dl, D, 0.0f, o);
            if (ACC2) {
                float q[C];
                ld_row<C, FULL>(qp + start * D, dl, D, 0.0f, q);
#pragma unroll
                for (int k = 0; k < C; k++) o[k] = o[k] + q[k];
            }
#pragma unroll
            for (int k = 0; k < C; k++) res[k] = o[k] + last[k];
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) res[k] = last[k];
        }
        st_row<C, FULL>(op + start * D, dl, D, res);
        if (WTA) wta_store(res, start);
    }
    // minLastPath = min over the padded vector (:162-166); pads are 65535
    float lm = PAD;
#pragma unroll
    for (int k = 0; k < C; k++) lm = ref_min(last[k], lm);
    float minLast = wave_min_f32_dpp(lm);
    float lastgray = gp[0];

    // prefetch rings: cost row, guidance gray and (when accumulating) the running sum.
    // Loads run PF steps ahead; indices past the end are clamped (harmless re-read) so the
    // steady-state step has no branch and its vmcnt wait is an exact count.
    float cbuf[PF][C];
    float obuf[ACC ? PF : 1][C];
    float qbuf[ACC2 ? PF : 1][C];
    float gbuf[PF];
    const int slast = nsteps - 1;
#pragma unroll
    for (int u = 0; u < PF; u++) {
        const int sp = min(1 + u, slast);
        gbuf[u] = gp[(long)sp * dirn];
        ld_row<C, FULL>(cp + (start + (long)sp * pstride) * D, dl, D, PAD, cbuf[u]);
        if (ACC) ld_row<C, FULL>(op + (start + (long)sp * pstride) * D, dl, D, 0.0f, obuf[ACC ? u : 0]);
        if (ACC2) ld_row<C, FULL>(qp + (start + (long)sp * pstride) * D, dl, D, 0.0f, qbuf[ACC2 ? u : 0]);
    }

#define SMT_SCAN_STEP(u, s)                                                                     \
    {                                                                                           \
        const long pix = start + (long)(s) * pstride;                                           \
        const float g = gbuf[u];                                                                \
        const float p2 = ref_max(p1, a.p2 / (fabsf(g - lastgray) + 1.0f)); /* :171 / :232 */    \
        if (HORIZ) lastgray = g;                                           /* :172 only */      \
        const float up_in = dpp_shift_f32<0x138>(last[C - 1], PAD); /* wave_shr:1 */            \
        const float dn_in = dpp_shift_f32<0x130>(last[0], PAD);     /* wave_shl:1 */            \
        const float l4 = minLast + p2;                                                          \
        float cur[C];                                                                           \
        float cm = PAD;                                                                         \
        _Pragma("unroll") for (int k = 0; k < C; k++) {                                         \
            const float lprev = (k == 0) ? up_in : last[k == 0 ? 0 : k - 1];                    \
            float lnext = (k == C - 1) ? dn_in : last[k == C - 1 ? k : k + 1];                  \
            if (dl + k + 1 >= D) lnext = PAD; /* entries past D-1 are the pad */                \
            const float l1 = last[k];                                                           \
            const float l2 = (HORIZ ? lprev : last[k]) + p1; /* :177 vs :238 (sic) */           \
            const float l3 = lnext + p1;                                                        \
            const float m = ref_min(ref_min(l1, l2), ref_min(l3, l4));                          \
            const float cs = cbuf[u][k] + m - minLast; /* :180 */                               \
            cur[k] = cs;                                                                        \
            if (FULL || dl + k < D) cm = ref_min(cm, cs);                                       \
        }                                                                                       \
        float res[C];                                                                           \
        _Pragma("unroll") for (int k = 0; k < C; k++)                                           \
            res[k] = ACC2 ? (obuf[ACC ? u : 0][k] + qbuf[ACC2 ? u : 0][k]) + cur[k]             \
                          : (ACC ? obuf[ACC ? u : 0][k] + cur[k] : cur[k]);                     \
        st_row<C, FULL>(op + pix * D, dl, D, res);                                              \
        if (WTA) wta_store(res, pix);                                                           \
        minLast = wave_min_f32_dpp(cm);                                                         \
        _Pragma("unroll") for (int k = 0; k < C; k++) last[k] = (FULL || dl + k < D) ? cur[k] : PAD; \
        const int sp = min((s) + PF, slast);                                                    \
        gbuf[u] = gp[(long)sp * dirn];                                                          \
        ld_row<C, FULL>(cp + (start + (long)sp * pstride) * D, dl, D, PAD, cbuf[u]);            \
        if (ACC) ld_row<C, FULL>(op + (start + (long)sp * pstride) * D, dl, D, 0.0f, obuf[ACC ? u : 0]); \
        if (ACC2) ld_row<C, FULL>(qp + (start + (long)sp * pstride) * D, dl, D, 0.0f, qbuf[ACC2 ? u : 0]); \
    }

    int s0 = 1;
    for (; s0 + PF <= nsteps; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) SMT_SCAN_STEP(u, s0 + u)
    }
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (s0 + u < nsteps) SMT_SCAN_STEP(u, s0 + u)
#undef SMT_SCAN_STEP
}

// Threads per workgroup.  Every scanline of a pass is resident at once, so the pass ends with the CU that got the most
// waves.  Rows (horizontal passes) have nothing in common: one wave per workgroup spreads H or 2 H of them evenly
// (1080p: 270 or 540 workgroups of four waves leave some CUs with 8 or 12 waves and the rest with 4 or 8 -- the
// left / right launch took 1.38 ms that way and takes 1.05 ms now).  Columns (vertical passes) are neighbours in memory:
// four adjacent columns per workgroup read 3 KB runs, which measures 2 % better than one column per workgroup.
#ifndef SMT_SCAN_NTLR
#define SMT_SCAN_NTLR 64
#endif
#ifndef SMT_SCAN_NTS
#define SMT_SCAN_NTS 256
#endif
constexpr int NTLR = SMT_SCAN_NTLR, NTS = SMT_SCAN_NTS;
template <int PASS> constexpr int scan_nt() { return PASS < 2 ? NTLR : NTS; }
template <int C, int PASS, int MODE, bool FULL>
__global__ void __launch_bounds__(scan_nt<PASS>()) k_scan(ScanArgs a)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    scan_body<C, PASS, MODE, FULL>(a, a.out, blockIdx.x * (scan_nt<PASS>() / 64) + wv);
}

// left->right into a.out and right->left into a.out_b, concurrently (blockIdx.y picks the pass)
template <int C, bool FULL>
__global__ void __launch_bounds__(NTLR) k_scan_lr(ScanArgs a)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int line = blockIdx.x * (NTLR / 64) + wv;
    if (blockIdx.y == 0) scan_body<C, 0, 0, FULL>(a, a.out, line);
    else scan_body<C, 1, 0, FULL>(a, a.out_b, line);
}

}  // namespace

struct smt_scanline {
    int device;
    int H, W, D, p1, p2;
    hipStream_t stream;
    float *scratch;      // one [H][W][D] volume: the right path until the up pass consumes it
};

SMT_API int smt_scanline_create(int H, int W, int D, int p1, int p2, smt_scanline **out)
{
    if (!out || H <= 0 || W <= 0 || D <= 0 || D > SMT_MAX_DISPARITY) return SMT_ERR_ARG;
    smt_scanline *h = new (std::nothrow) smt_scanline();
    if (!h) return SMT_ERR_ALLOC;
    h->device = smt_current_device();
    h->H = H; h->W = W; h->D = D; h->p1 = p1; h->p2 = p2; h->stream = nullptr; h->scratch = nullptr;
    *out = h;
    return SMT_OK;
}
SMT_API int smt_scanline_create_on(int device, int H, int W, int D, int p1, int p2, smt_scanline **out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(device);
    return smt_scanline_create(H, W, D, p1, p2, out);
}
SMT_API int smt_scanline_destroy(smt_scanline *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (h->scratch) (void)hipFree(h->scratch);
    delete h;
    return SMT_OK;
}
SMT_API int smt_scanline_set_stream(smt_scanline *h, void *s)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->stream = smt_stream(s);
    return SMT_OK;
}

template <int C, int MODE, bool FULL>
static void launch_scan2(smt_scanline *h, int pass, const ScanArgs &a)
{
    const int nlines = pass < 2 ? h->H : h->W;
    const int nt = pass < 2 ? NTLR : NTS;
    dim3 grid((nlines + nt / 64 - 1) / (nt / 64));
    switch (pass) {
    case 0: hipLaunchKernelGGL((k_scan<C, 0, MODE, FULL>), grid, dim3(nt), 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((k_scan<C, 1, MODE, FULL>), grid, dim3(nt), 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((k_scan<C, 2, MODE, FULL>), grid, dim3(nt), 0, h->stream, a); break;
    default: hipLaunchKernelGGL((k_scan<C, 3, MODE, FULL>), grid, dim3(nt), 0, h->stream, a); break;
    }
}

template <int C>
static void launch_scan(smt_scanline *h, int pass, const ScanArgs &a, int mode)
{
    const bool full = (h->D == 64 * C) && C <= 4;         // 5..8 hypotheses per lane (D > 256): the predicated form only
    if (full) {
        constexpr int CF = C <= 4 ? C : 1;                // the vector form is never instantiated for C > 4
        if (mode == 0) launch_scan2<CF, 0, true>(h, pass, a);
        else if (mode == 1) launch_scan2<CF, 1, true>(h, pass, a);
        else if (mode == 2) launch_scan2<CF, 2, true>(h, pass, a);
        else launch_scan2<CF, 3, true>(h, pass, a);
    } else {
        if (mode == 0) launch_scan2<C, 0, false>(h, pass, a);
        else if (mode == 1) launch_scan2<C, 1, false>(h, pass, a);
        else if (mode == 2) launch_scan2<C, 2, false>(h, pass, a);
        else launch_scan2<C, 3, false>(h, pass, a);
    }
}

template <int C>
static void launch_lr(smt_scanline *h, const ScanArgs &a)
{
    dim3 grid((h->H + NTLR / 64 - 1) / (NTLR / 64), 2);
    if (C <= 4 && h->D == 64 * C) hipLaunchKernelGGL((k_scan_lr<(C <= 4 ? C : 1), true>), grid, dim3(NTLR), 0, h->stream, a);
    else hipLaunchKernelGGL((k_scan_lr<C, false>), grid, dim3(NTLR), 0, h->stream, a);
}

static int scan_pass(smt_scanline *h, const ScanArgs &a, int pass, int mode)
{
    switch ((h->D + 63) / 64) {
    case 1: launch_scan<1>(h, pass, a, mode); break;
    case 2: launch_scan<2>(h, pass, a, mode); break;
    case 3: launch_scan<3>(h, pass, a, mode); break;
    case 4: launch_scan<4>(h, pass, a, mode); break;
    case 5: launch_scan<5>(h, pass, a, mode); break;
    case 6: launch_scan<6>(h, pass, a, mode); break;
    case 7: launch_scan<7>(h, pass, a, mode); break;
    default: launch_scan<8>(h, pass, a, mode); break;
    }
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_scanline_pass(smt_scanline *h, const float *vin, const float *gray, int pass, float *vout)
{
    if (!h || !vin || !gray || !vout || vin == vout || pass < 0 || pass > 3) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    ScanArgs a{vin, gray, vout, nullptr, nullptr, nullptr, h->H, h->W, h->D, (float)h->p1, (float)h->p2};
    return scan_pass(h, a, pass, 0);
}

SMT_API int smt_scanline_run(smt_scanline *h, const float *vin, const float *gray, float *vout, float *disp)
{
    if (!h || !vin || !gray || !vout || vin == vout) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (!h->scratch) {
        int rc = smt_malloc((void **)&h->scratch, (size_t)h->H * h->W * h->D * 4);
        if (rc != SMT_OK) return rc;
    }
    ScanArgs a{vin, gray, vout, h->scratch, h->scratch, nullptr, h->H, h->W, h->D, (float)h->p1, (float)h->p2};
    switch ((h->D + 63) / 64) {                                            // left -> vout, right -> scratch
    case 1: launch_lr<1>(h, a); break;
    case 2: launch_lr<2>(h, a); break;
    case 3: launch_lr<3>(h, a); break;
    case 4: launch_lr<4>(h, a); break;
    case 5: launch_lr<5>(h, a); break;
    case 6: launch_lr<6>(h, a); break;
    case 7: launch_lr<7>(h, a); break;
    default: launch_lr<8>(h, a); break;
    }
    SMT_LAUNCH_CHECK();
    int rc = scan_pass(h, a, 2, 3);                                        // vout = (left + right) + up
    if (rc != SMT_OK) return rc;
    a.disp = disp;
    return scan_pass(h, a, 3, disp ? 2 : 1);                               // vout = (..) + down [+ WTA]
}
