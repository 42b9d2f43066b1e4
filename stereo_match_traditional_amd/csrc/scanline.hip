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
// (:124).  Here pass 0 writes the output volume and passes 1..3 accumulate into it in
// that same association order; the last pass can fuse ScanlineOptimizer::WTA (:40-64).
#include "smt_common.h"
#include <new>

namespace {

constexpr int NT = 256;
constexpr int PF = 4;             // prefetch depth (scan steps)
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
    float *disp;         // fused WTA of the accumulated volume (last pass) or null
    int H, W, D;
    float p1, p2;
    int accumulate;      // out = out + path instead of out = path
};

// PASS: 0 left->right, 1 right->left, 2 top->bottom, 3 bottom->top
template <int C, int PASS>
__global__ void __launch_bounds__(NT) k_scan(ScanArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int line = blockIdx.x * (NT / 64) + wv;
    constexpr bool HORIZ = (PASS < 2);
    const int H = a.H, W = a.W, D = a.D;
    const int nlines = HORIZ ? H : W;
    if (line >= nlines) return;
    const int nsteps = HORIZ ? W : H;                 // pixels on the line
    const int dirn = (PASS == 0 || PASS == 2) ? 1 : -1;

    // pixel index (flat, in pixels) of step s:  start + s*pstride
    long start;
    long pstride;
    if (HORIZ) { start = (long)line * W + (dirn > 0 ? 0 : W - 1); pstride = dirn; }
    else       { start = (long)(dirn > 0 ? 0 : H - 1) * W + line; pstride = (long)dirn * W; }
    // gray pointer: horizontal passes follow the pixel; vertical passes step by ONE element (:221,:250)
    const long gstart = start;
    const long gstride = dirn;

    const int dl = lane * C;
    bool act[C];
#pragma unroll
    for (int k = 0; k < C; k++) act[k] = (dl + k < D);

    auto load_cost = [&](long pix, float (&dst)[C]) {
        const float *src = a.cost + pix * D + dl;
#pragma unroll
        for (int k = 0; k < C; k++) dst[k] = act[k] ? src[k] : PAD;
    };

    float last[C];
    // first pixel: path = cost (:153-155)
    load_cost(start, last);
    {
        float *dst = a.out + start * D + dl;
        float res[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
            res[k] = last[k];
            if (a.accumulate && act[k]) res[k] = dst[k] + last[k];
            if (act[k]) dst[k] = res[k];
        }
        if (a.disp) {
            float best = INFINITY; int bk = 0;
#pragma unroll
            for (int k = 0; k < C; k++)
                if (act[k] && (k == 0 || best > res[k])) { best = res[k]; bk = k; }
            if (dl >= D) best = INFINITY;
            const int wd = wave_argmin_first(best, dl + bk);
            if (lane == 0) a.disp[start] = (float)wd;
        }
    }
    // minLastPath = min over the padded vector (:162-166); pads are 65535
    float lm = PAD;
#pragma unroll
    for (int k = 0; k < C; k++) lm = ref_min(last[k], lm);
    float minLast = wave_min_f32_dpp(lm);

    float lastgray = a.gray[gstart];

    // prefetch ring
    float cbuf[PF][C];
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (1 + u < nsteps) load_cost(start + (long)(1 + u) * pstride, cbuf[u]);

    for (int s0 = 1; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int s = s0 + u;
            if (s < nsteps) {
                const long pix = start + (long)s * pstride;
                const float g = a.gray[gstart + (long)s * gstride];
                const float p2 = ref_max(a.p1, a.p2 / (fabsf(g - lastgray) + 1.0f));   // :171 / :232
                if (HORIZ) lastgray = g;                                                 // :172 (not in up/down)
                // neighbours of the previous path vector
                const float up_in = dpp_shift_f32<0x138>(last[C - 1], PAD);   // wave_shr:1 -> from lane-1
                const float dn_in = dpp_shift_f32<0x130>(last[0], PAD);       // wave_shl:1 -> from lane+1
                const float l4 = minLast + p2;
                float cur[C];
                float cm = PAD;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    const float lprev = (k == 0) ? up_in : last[k - 1];
                    float lnext = (k == C - 1) ? dn_in : last[k + 1];
                    // entries past D-1 are the pad value, whichever lane holds them
                    if (dl + k + 1 >= D) lnext = PAD;
                    const float l1 = last[k];
                    const float l2 = (HORIZ ? lprev : last[k]) + a.p1;        // :177 vs :238 (sic)
                    const float l3 = lnext + a.p1;
                    const float m = ref_min(ref_min(l1, l2), ref_min(l3, l4));
                    const float cs = cbuf[u][k] + m - minLast;               // :180
                    cur[k] = cs;
                    if (act[k]) cm = ref_min(cm, cs);
                }
                // write (or accumulate) the path value
                float *dst = a.out + pix * D + dl;
                float res[C];
#pragma unroll
                for (int k = 0; k < C; k++) {
                    res[k] = cur[k];
                    if (a.accumulate && act[k]) res[k] = dst[k] + cur[k];
                    if (act[k]) dst[k] = res[k];
                }
                if (a.disp) {
                    float best = INFINITY; int bk = 0;
#pragma unroll
                    for (int k = 0; k < C; k++)
                        if (act[k] && (k == 0 || best > res[k])) { best = res[k]; bk = k; }
                    if (dl >= D) best = INFINITY;
                    const int wd = wave_argmin_first(best, dl + bk);
                    if (lane == 0) a.disp[pix] = (float)wd;
                }
                minLast = wave_min_f32_dpp(cm);
#pragma unroll
                for (int k = 0; k < C; k++) last[k] = act[k] ? cur[k] : PAD;
                // refill this ring slot
                if (s + PF < nsteps) load_cost(start + (long)(s + PF) * pstride, cbuf[u]);
            }
        }
    }
}

}  // namespace

struct smt_scanline {
    int H, W, D, p1, p2;
    hipStream_t stream;
};

SMT_API int smt_scanline_create(int H, int W, int D, int p1, int p2, smt_scanline **out)
{
    if (!out || H <= 0 || W <= 0 || D <= 0 || D > 256) return SMT_ERR_ARG;
    smt_scanline *h = new (std::nothrow) smt_scanline();
    if (!h) return SMT_ERR_ALLOC;
    h->H = H; h->W = W; h->D = D; h->p1 = p1; h->p2 = p2; h->stream = nullptr;
    *out = h;
    return SMT_OK;
}
SMT_API int smt_scanline_destroy(smt_scanline *h)
{
    if (!h) return SMT_ERR_ARG;
    delete h;
    return SMT_OK;
}
SMT_API int smt_scanline_set_stream(smt_scanline *h, void *s)
{
    if (!h) return SMT_ERR_ARG;
    h->stream = smt_stream(s);
    return SMT_OK;
}

template <int C>
static void launch_scan(smt_scanline *h, int pass, const ScanArgs &a)
{
    const int nlines = pass < 2 ? h->H : h->W;
    dim3 grid((nlines + 3) / 4);
    switch (pass) {
    case 0: hipLaunchKernelGGL((k_scan<C, 0>), grid, dim3(NT), 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((k_scan<C, 1>), grid, dim3(NT), 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((k_scan<C, 2>), grid, dim3(NT), 0, h->stream, a); break;
    default: hipLaunchKernelGGL((k_scan<C, 3>), grid, dim3(NT), 0, h->stream, a); break;
    }
}

static int scan_pass(smt_scanline *h, const float *vin, const float *gray, int pass, float *vout,
                     int accumulate, float *disp)
{
    ScanArgs a{vin, gray, vout, disp, h->H, h->W, h->D, (float)h->p1, (float)h->p2, accumulate};
    switch ((h->D + 63) / 64) {
    case 1: launch_scan<1>(h, pass, a); break;
    case 2: launch_scan<2>(h, pass, a); break;
    case 3: launch_scan<3>(h, pass, a); break;
    default: launch_scan<4>(h, pass, a); break;
    }
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_scanline_pass(smt_scanline *h, const float *vin, const float *gray, int pass, float *vout)
{
    if (!h || !vin || !gray || !vout || vin == vout || pass < 0 || pass > 3) return SMT_ERR_ARG;
    return scan_pass(h, vin, gray, pass, vout, 0, nullptr);
}

SMT_API int smt_scanline_run(smt_scanline *h, const float *vin, const float *gray, float *vout, float *disp)
{
    if (!h || !vin || !gray || !vout || vin == vout) return SMT_ERR_ARG;
    int rc = scan_pass(h, vin, gray, 0, vout, 0, nullptr);                 // left
    if (rc == SMT_OK) rc = scan_pass(h, vin, gray, 1, vout, 1, nullptr);   // (left + right)
    if (rc == SMT_OK) rc = scan_pass(h, vin, gray, 2, vout, 1, nullptr);   // (..) + up
    if (rc == SMT_OK) rc = scan_pass(h, vin, gray, 3, vout, 1, disp);      // (..) + down, fused WTA
    return rc;
}
