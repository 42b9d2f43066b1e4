// Fixed-window matchers -- replace SAD/Sad.h (GetPointDepthLeft/Right, OptimalDisparity,
// GetMinSadIndex), NCC/NCC.h (ComputeCost, WinTakeAll, NCC_algorithem) and ASW/ASW.h
// (bilateralfiterWight, AdaptiveSupportWeight(Right), WinTakeAll).
//
// All three share one mapping: one wavefront per output pixel, the disparity axis strided
// over the lanes (d = lane + 64*k), so that for every window tap the lanes read 64
// consecutive bytes of the other image.  The per-pixel cost vectors are never written to
// HBM unless the caller asks for them (the reference keeps them in a per-pixel
// std::vector too).
//
// The reference's "copy the previous cost" branches (Sad.h:125-129,167-171; ASW.h:369-372,
// 422-425) make cost[d] = cost[dmax] for d beyond the last in-range disparity dmax, i.e.
// the hypothesis is evaluated at min(d, dmax).
#include "smt_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int NT = 256;
constexpr int KMAX = 4;        // D <= 256

// wave minimum of non-negative floats (SAD sums, 65535, +inf): for v >= +0 the bit patterns order like
// the values, so this is the DPP integer reduction (no LDS-crossbar shuffles)
__device__ __forceinline__ float wave_min_nonneg(float v) { return __uint_as_float(wave_min_u32(__float_as_uint(v))); }

// ---------------------------------------------------------------------------------- SAD
// OptimalDisparity (Sad.h:40-85) over the wave-distributed vector sad[d], d = lane+64k.
__device__ int sad_optimal(const float (&sad)[KMAX], int D, int lane)
{
    // min over d >= 1, first strict (:46-53), starting from 0xffff
    float lm = 65535.0f; int ld = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int d = lane + 64 * k;
        if (d >= 1 && d < D && lm > sad[k]) { lm = sad[k]; ld = d; }
    }
    const float minv = wave_min_nonneg(lm);
    // first d >= 1 whose value equals the minimum (if any value beat 65535)
    int cand = (lm == minv && ld != 0x7fffffff) ? ld : 0x7fffffff;
    cand = (int)wave_min_u32((unsigned)cand);
    const float best = (cand == 0x7fffffff) ? 65535.0f : (float)cand;
    // second minimum: starts at sad[0]; every entry equal to minv is skipped (:55-64)
    float ls = INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int d = lane + 64 * k;
        if (d < D && !(minv == sad[k])) ls = fminf(ls, sad[k]);
    }
    const float s0 = __shfl(sad[0], 0, WAVE);
    const float sec = fminf(s0, wave_min_nonneg(ls));
    if ((double)(sec - minv) <= 0.01) return 0;                           // :66
    if (best == 0.0f || best == (float)(D - 1)) return 0;                 // :71
    return (int)best;                                                     // :84
}

// grid: one wave per ORIGINAL pixel (io, jo); Lp/Rp padded by w = winsize+1
__global__ void __launch_bounds__(NT) k_sad(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp, int H,
                                            int W, int D, int w, int view, int32_t *__restrict__ disp)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * (NT / 64) + wv;
    if (p >= H * W) return;
    const int io = p / W, jo = p - io * W;
    const int Wp = W + 2 * w, side = 2 * w + 1;
    if (view == 1 && (io >= H - 1 || jo >= W - 1)) {                      // never written (:157,:160)
        if (lane == 0) disp[p] = 0;
        return;
    }
    const int dmax = (view == 0) ? jo : (W - 1 - jo);                     // last in-range disparity
    float sad[KMAX];
    // top-left corners of the two windows in padded coordinates; hypothesis d is evaluated at min(d, dmax)
    const uint8_t *a = (view == 0 ? Lp : Rp) + (size_t)io * Wp + jo;
    const uint8_t *b[KMAX];
    unsigned acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int d = lane + 64 * k;
        const int dd = d < dmax ? d : dmax;
        b[k] = (view == 0) ? Rp + (size_t)io * Wp + jo - dd : Lp + (size_t)io * Wp + jo + dd;
        acc[k] = 0;
    }
    const int nk = (D + 63) >> 6;                                         // live 64-disparity slots (uniform)
    if (side >= 4) {
        // four taps per instruction: unaligned dword fetches + v_sad_u8 (sum of four |a - b| + accumulator).
        // A row is side bytes: full dwords, then one dword ending at the row's last byte with the
        // bytes already counted masked out of both operands (so nothing is read past the window).
        const int nfull = side >> 2, rem = side & 3;
        const unsigned tmask = rem ? (0xffffffffu << (8 * (4 - rem))) : 0u;
        for (int r = 0; r < side; r++) {
            for (int g = 0; g <= nfull; g++) {
                if (g == nfull && !rem) break;
                const int c0 = (g < nfull) ? 4 * g : side - 4;
                const unsigned m = (g < nfull) ? 0xffffffffu : tmask;
                unsigned a4;
                __builtin_memcpy(&a4, a + r * Wp + c0, 4);
                a4 &= m;
#pragma unroll
                for (int k = 0; k < KMAX; k++)
                    if (k < nk) {
                        unsigned b4;
                        __builtin_memcpy(&b4, b[k] + r * Wp + c0, 4);
                        acc[k] = __builtin_amdgcn_sad_u8(a4, b4 & m, acc[k]);
                    }
            }
        }
    } else {
        for (int r = 0; r < side; r++)
            for (int c = 0; c < side; c++)
#pragma unroll
                for (int k = 0; k < KMAX; k++)
                    if (k < nk) acc[k] += (unsigned)abs((int)a[r * Wp + c] - (int)b[k][r * Wp + c]);
    }
#pragma unroll
    for (int k = 0; k < KMAX; k++) sad[k] = (lane + 64 * k < D) ? (float)acc[k] : 0.0f;   // sadvalue :15-20
    int out;
    if (view == 0) out = sad_optimal(sad, D, lane);
    else {                                                                // GetMinSadIndex :22-38
        float lm = INFINITY; int ld = 0;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int d = lane + 64 * k;
            if (d < D && sad[k] < lm) { lm = sad[k]; ld = d; }
        }
        const float m = wave_min_nonneg(lm);
        int cand = (lm == m) ? ld : 0x7fffffff;
        out = (int)wave_min_u32((unsigned)cand);
    }
    if (lane == 0) disp[p] = out;
}

// Second formulation (default for windows of side >= 4): k_sad above is bound by its own global-load latency (one
// wave per pixel, ten unaligned dword loads per lane and pixel at 5x5; 0.8 % of the v_sad_u8 rate at config 1).
// Here a workgroup owns STP consecutive pixels of a row and stages, once, the `side` rows of the anchor image over
// its pixels and of the other image over every window position its hypotheses can touch -- each row expanded to one
// dword per BYTE offset (E[A] = bytes A .. A + 3: the four byte-shifted copies of the row interleaved dword by dword),
// so that the four bytes at any byte address A are the aligned dword E[A].  The tap loop is then LDS reads + v_sad_u8
// only: per window row and group of four columns one uniform (broadcast) read of the anchor dword and one read per
// hypothesis slot; lanes with consecutive d read consecutive dwords, conflict-free whatever the pixel.  (With the four
// copies kept apart, 8 banks from each other, a lane whose byte offset crosses a dword boundary reads one word lower
// than its neighbours and shares a bank with another lane for three pixels in four -- the NCC kernel had the same flaw.)
// Sums, masks and the selection rules are those of k_sad; the two are compared bit for bit in the tests.
constexpr int STP = 32;                                   // pixels per workgroup (8 per wave)

template <int K>
__global__ void __launch_bounds__(NT) k_sad2(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp, int H, int W,
                                             int D, int w, int view, int32_t *__restrict__ disp, int ALW, int OLW)
{
    extern __shared__ __attribute__((aligned(16))) unsigned s_sad[];
    const int side = 2 * w + 1, Wp = W + 2 * w;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int io = blockIdx.y, jo0 = blockIdx.x * STP;
    // ALW / OLW: dword columns staged per row of the anchor / other image (4 expanded dwords each)
    unsigned *s_a = s_sad;                                 // [side][4 * ALW]
    unsigned *s_o = s_sad + (size_t)side * 4 * ALW;        // [side][4 * OLW]
    const uint8_t *Aimg = (view == 0 ? Lp : Rp) + (size_t)io * Wp;
    const uint8_t *Bimg = (view == 0 ? Rp : Lp) + (size_t)io * Wp;
    // first byte column staged of the other image: view 0: jo0 - (64 K - 1) .. ; view 1: jo0 ..
    const int xbase = (view == 0) ? jo0 - (64 * K - 1) : jo0;
    auto stage = [&](unsigned *dst, const uint8_t *img, int x0, int LW) {
        // one thread per dword column j of a row: bytes x0 + 4j .. x0 + 4j + 7 as two (unaligned) dword loads, the four
        // shifted copies from them by v_alignbyte; columns that touch the image edge are assembled byte by byte with
        // the column clamped (bytes no hypothesis uses: any in-image value)
        for (int e = threadIdx.x; e < side * LW; e += NT) {
            const int r = e / LW, j = e - r * LW;
            const uint8_t *row = img + (size_t)r * Wp;
            const int x = x0 + 4 * j;
            unsigned lo, hi;
            if (x >= 0 && x + 7 <= Wp - 1) {
                __builtin_memcpy(&lo, row + x, 4);
                __builtin_memcpy(&hi, row + x + 4, 4);
            } else {
                lo = hi = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    int xa = x + b, xb = x + 4 + b;
                    xa = xa < 0 ? 0 : (xa > Wp - 1 ? Wp - 1 : xa);
                    xb = xb < 0 ? 0 : (xb > Wp - 1 ? Wp - 1 : xb);
                    lo |= (unsigned)row[xa] << (8 * b);
                    hi |= (unsigned)row[xb] << (8 * b);
                }
            }
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4v *>(dst + (size_t)r * 4 * LW + 4 * j) =     // one 16-byte store: E[4j .. 4j + 3]
                u4v{lo, __builtin_amdgcn_alignbyte(hi, lo, 1), __builtin_amdgcn_alignbyte(hi, lo, 2), __builtin_amdgcn_alignbyte(hi, lo, 3)};
        }
    };
    stage(s_a, Aimg, jo0, ALW);
    stage(s_o, Bimg, xbase, OLW);
    __syncthreads();

    const int nfull = side >> 2, rem = side & 3;
    const unsigned tmask = rem ? (0xffffffffu << (8 * (4 - rem))) : 0u;
    for (int p = wv; p < STP; p += NT / 64) {
        const int jo = jo0 + p;
        if (jo >= W) break;
        const int pix = io * W + jo;
        if (view == 1 && (io >= H - 1 || jo >= W - 1)) {                  // never written (:157,:160)
            if (lane == 0) disp[pix] = 0;
            continue;
        }
        const int dmax = (view == 0) ? jo : (W - 1 - jo);                 // last in-range disparity
        // dword index, inside a row's expanded block, of the window's first full group and of its tail group
        // (columns side - 4 .. side - 1): a full group g is then just "+ 4 g", so the tap loop has no address arithmetic
        int bfull[K], btail[K];
        unsigned acc[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int d = lane + 64 * k;
            const int dd = d < dmax ? d : dmax;
            const int xs = ((view == 0) ? jo - dd : jo + dd) - xbase, xt = xs + side - 4;
            bfull[k] = xs;
            btail[k] = xt;
            acc[k] = 0;
        }
        const int afull = p, atail = p + side - 4;
        for (int r = 0; r < side; r++) {
            const unsigned *ra = s_a + (size_t)r * 4 * ALW, *ro = s_o + (size_t)r * 4 * OLW;
            for (int g = 0; g < nfull; g++) {
                const unsigned a4 = ra[afull + 4 * g];
#pragma unroll
                for (int k = 0; k < K; k++) acc[k] = __builtin_amdgcn_sad_u8(a4, ro[bfull[k] + 4 * g], acc[k]);
            }
            if (rem) {
                // the last dword ends at the row's last byte; the bytes already counted are masked out of both operands
                const unsigned a4 = ra[atail] & tmask;
#pragma unroll
                for (int k = 0; k < K; k++) acc[k] = __builtin_amdgcn_sad_u8(a4, ro[btail[k]] & tmask, acc[k]);
            }
        }
        float sad[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) sad[k] = (k < K && lane + 64 * k < D) ? (float)acc[k < K ? k : 0] : 0.0f;   // sadvalue :15-20
        int out;
        if (view == 0) out = sad_optimal(sad, D, lane);
        else {                                                            // GetMinSadIndex :22-38
            float lm = INFINITY; int ld = 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int d = lane + 64 * k;
                if (d < D && sad[k] < lm) { lm = sad[k]; ld = d; }
            }
            const float mn = wave_min_nonneg(lm);
            int cand = (lm == mn) ? ld : 0x7fffffff;
            out = (int)wave_min_u32((unsigned)cand);
        }
        if (lane == 0) disp[pix] = out;
    }
}

// ---------------------------------------------------------------------------------- NCC
__device__ double ncc_cost(const uint8_t *a, const uint8_t *b, int W, int side)   // NCC.h:15-49
{
    double lm = 0, rm = 0, ls = 0, rs = 0, num = 0;
    const int n = side * side;
    for (int i = 0; i < side; i++)
        for (int j = 0; j < side; j++) { lm += a[i * W + j]; rm += b[i * W + j]; }   // exact integer sums
    lm /= n; rm /= n;
    for (int i = 0; i < side; i++)
        for (int j = 0; j < side; j++) {
            const double x = a[i * W + j] - lm, y = b[i * W + j] - rm;
            ls += x * x; rs += y * y; num += x * y;
        }
    return num / (sqrt(ls) * sqrt(rs));
}

__global__ void __launch_bounds__(NT) k_ncc(const uint8_t *__restrict__ L, const uint8_t *__restrict__ R, int H, int W,
                                            int D, int win, int32_t *__restrict__ disp, double *__restrict__ cost_out)
{
    __shared__ double s_cost[NT / 64][256];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * (NT / 64) + wv;
    if (p >= H * W) return;
    const int i = p / W, j = p - i * W;
    if (i < win || i >= H - win || j < win || j >= W - win) {             // border untouched (zeros)
        if (lane == 0) disp[p] = 0;
        return;
    }
    const int side = 2 * win + 1;
    for (int d = lane; d < D; d += 64) {
        double c;
        if (j - win - d >= 0)
            c = ncc_cost(L + (size_t)(i - win) * W + (j - win), R + (size_t)(i - win) * W + (j - win - d), W, side);
        else c = 255.0;                                                   // `invalid` 0xff, NCC.h:88
        s_cost[wv][d] = c;
        if (cost_out) cost_out[(size_t)p * D + d] = c;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {                                                      // WinTakeAll :53-67 (sequential, float-narrowed)
        int best = 0;
        float m = (float)s_cost[wv][0];
        for (int d = 1; d < D; d++)
            if ((double)m < s_cost[wv][d]) { best = d; m = (float)s_cost[wv][d]; }
        disp[p] = best;
    }
}

// ---- NCC, second formulation (default) --------------------------------------------------------------------
// k_ncc above is the reference's loop nest, one lane per hypothesis, two passes of byte loads over both windows:
// 2.4 ms at 450x375, D=64, 21x21.  All three sums of ComputeCost are integer-valued polynomials of the bytes:
// with n = side^2, Sa = sum a, Saa = sum a^2, Sb, Sbb likewise and Sab = sum a*b,
//     sum (a-lm)^2 = (n*Saa - Sa^2)/n,  sum (b-rm)^2 = (n*Sbb - Sb^2)/n,  sum (a-lm)(b-rm) = (n*Sab - Sa*Sb)/n,
// so  cost = (n*Sab - Sa*Sb) / (sqrt(n*Saa - Sa^2) * sqrt(n*Sbb - Sb^2))   (the 1/n cancel),
// every integer exact in float64 (< 2^53).  Sa, Saa depend on the left position only and Sb, Sbb on the right
// position only (k_ncc_stats, separable window sums, once per image); what is left per hypothesis is Sab, taken
// four taps per v_dot4_u32_u8.  The reference accumulates the same quantities in float64 with a rounding per
// tap; the two agree to ~1e-14 relative, the test tolerance is 1e-4 (north_star), and the cases the reference
// turns into NaN (a flat window: 0/0) are exact zeros here too.  WinTakeAll's float-narrowed running maximum
// (NCC.h:53-67) is evaluated in parallel: m before step d equals the maximum of (float)c[e] over e < d (NaN
// entries skipped, a NaN at d = 0 poisons everything), so d wins iff c[d] > that prefix maximum, and the answer
// is the last winner.
constexpr int NCP = 16;                                   // pixels (waves) per workgroup
constexpr int NCT = 16;                                   // rows per k_ncc_stats tile

__global__ void __launch_bounds__(256) k_ncc_stats(const uint8_t *__restrict__ L, const uint8_t *__restrict__ R, int H, int W,
                                                   int win, int *__restrict__ sumL, double *__restrict__ rootL,
                                                   int *__restrict__ sumR, double *__restrict__ rootR)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int side = 2 * win + 1, RC = 64 + 2 * win, RR = NCT + 2 * win;
    uint8_t *raw = smem;                                  // [RR][RC]
    int *hs = (int *)(smem + (((size_t)RR * RC + 15) & ~(size_t)15));   // [RR][64] row-window sums
    int *hq = hs + RR * 64;                               // [RR][64] row-window sums of squares
    const uint8_t *img = blockIdx.z == 0 ? L : R;
    int *osum = blockIdx.z == 0 ? sumL : sumR;
    double *oroot = blockIdx.z == 0 ? rootL : rootR;
    const int y0 = win + blockIdx.y * NCT, x0 = win + blockIdx.x * 64;   // first output of the tile
    for (int e = threadIdx.x; e < RR * RC; e += 256) {
        const int r = e / RC, c = e - r * RC;
        const int yy = min(y0 - win + r, H - 1), xx = min(x0 - win + c, W - 1);   // >= 0 by construction
        raw[e] = img[(size_t)yy * W + xx];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RR * 64; e += 256) {
        const int r = e >> 6, x = e & 63;
        int s1 = 0, s2 = 0;
        for (int c = 0; c < side; c++) { const int v = raw[r * RC + x + c]; s1 += v; s2 += v * v; }
        hs[e] = s1; hq[e] = s2;
    }
    __syncthreads();
    const double n = (double)(side * side);
    for (int e = threadIdx.x; e < NCT * 64; e += 256) {
        const int y = e >> 6, x = e & 63;
        if (y0 + y >= H - win || x0 + x >= W - win) continue;
        int s1 = 0, s2 = 0;
        for (int r = 0; r < side; r++) { s1 += hs[(y + r) * 64 + x]; s2 += hq[(y + r) * 64 + x]; }
        const size_t p = (size_t)(y0 + y) * W + x0 + x;
        osum[p] = s1;
        oroot[p] = sqrt(n * (double)s2 - (double)s1 * (double)s1);   // both products < 2^53: exact, and so is the difference
    }
}

// One ds_read_b32 that stays one (see lds_f64 below in the ASW section: merged wide reads of 4-byte-aligned
// addresses are serialised by the LDS -- 65 LDS cycles per instruction measured here).
__device__ __forceinline__ uint32_t lds_u32(const uint32_t *p)
{
    return *(const volatile __attribute__((address_space(3))) uint32_t *)p;
}

// Offset (in dwords) of byte-shifted copy s of the right rows.  The 32 lanes of a half-wave read 8 groups x 4
// copies; consecutive groups sit K dwords apart, so the copies are placed on the banks that spacing leaves free
// (K = 3 keeps one 2-way overlap).  CS is a multiple of 32.
template <int K>
__device__ __forceinline__ int ncc_copy_off(int s, int CS)
{
    const int bank = K == 1 ? 8 * s : (K == 2 ? (s & 1) + 16 * (s >> 1) : s);
    return s * CS + bank;
}

// inclusive prefix maximum over the lanes of a wave (f32; NaN-free input)
__device__ __forceinline__ float wave_prefix_max_f32(float v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(v, off, WAVE);
        if (lane >= off) v = fmaxf(v, o);
    }
    return v;
}

template <int K, int G>
__global__ void __launch_bounds__(NCP * 64) k_ncc2(const uint8_t *__restrict__ L, const uint8_t *__restrict__ R, int H, int W,
                                                   int D, int win, const int *__restrict__ sumL,
                                                   const double *__restrict__ rootL, const int *__restrict__ sumR,
                                                   const double *__restrict__ rootR, int RW, int CS,
                                                   int32_t *__restrict__ disp, double *__restrict__ cost_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int side = 2 * win + 1;
    // LDS: the right image's rows under the workgroup as 4 byte-shifted dword copies (copy s, dword w = bytes
    // xlo + 4w + s .. + 3: any unaligned dword of a row is an aligned dword of one copy), the raw rows they are
    // built from, and each pixel's left window as dwords of 4 taps (zero past the window's last column)
    uint32_t *s_R = (uint32_t *)smem;                     // [4][CS], row r of copy s at ncc_copy_off<K>(s, CS) + r * RW
    uint32_t *s_raw = s_R + 4 * CS;                       // [side][RW + 1]
    uint32_t *s_A = s_raw + side * (RW + 1);              // [NCP][side * G]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = win + blockIdx.y, j0 = win + blockIdx.x * NCP;
    const int xlo = ((j0 - win - (64 * K - 1)) >> 2) << 2;               // first staged column (multiple of 4, may be < 0)
    for (int e = threadIdx.x; e < side * (RW + 1); e += NCP * 64) {
        const int r = e / (RW + 1), w = e - r * (RW + 1);
        const uint8_t *row = R + (size_t)(i - win + r) * W;
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int x = xlo + 4 * w + b;
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);      // columns outside the image only feed sentinel hypotheses
            v |= (uint32_t)row[x] << (8 * b);
        }
        s_raw[e] = v;
    }
    const int j = j0 + wv;
    const bool live = j < W - win;
    {
        const int jc = live ? j : W - win - 1;
        for (int t = lane; t < side * G; t += 64) {
            const int r = t / G, g = t - r * G;
            const uint8_t *row = L + (size_t)(i - win + r) * W + (jc - win) + 4 * g;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4 * g + b < side) v |= (uint32_t)row[b] << (8 * b);
            s_A[wv * side * G + t] = v;
        }
    }
    __syncthreads();
    // K == 1 (D <= 64): the four copies interleaved dword by dword, E[r][4 w + s] = the dword at byte offset 4 w + s of
    // row r.  A half-wave's lanes then read the dwords at 32 consecutive byte offsets o = c - lane: 32 consecutive LDS
    // dwords, conflict-free for every pixel.  With the copies apart (below) a lane whose byte offset crosses a dword
    // boundary of the row reads one word lower than its neighbours, which puts two lanes of a half-wave on one bank for
    // three pixels in four (PMC: 26 % of the LDS cycles at D = 64 were conflict cycles).  For K > 1 the groups sit 4K
    // bytes apart and the interleaved form would fold them onto each other: the separate copies stay.
    constexpr bool EL = (K == 1);
    if constexpr (EL) {
        for (int e = threadIdx.x; e < 4 * side * RW; e += NCP * 64) {
            const int sft = e & 3, rw = e >> 2, r = rw / RW, w = rw - r * RW;
            s_R[e] = __builtin_amdgcn_alignbyte(s_raw[r * (RW + 1) + w + 1], s_raw[r * (RW + 1) + w], (unsigned)sft);
        }
    } else {
        for (int e = threadIdx.x; e < 4 * side * RW; e += NCP * 64) {
            const int sft = e / (side * RW), rem = e - sft * (side * RW), r = rem / RW, w = rem - r * RW;
            s_R[ncc_copy_off<K>(sft, CS) + r * RW + w] = __builtin_amdgcn_alignbyte(s_raw[r * (RW + 1) + w + 1], s_raw[r * (RW + 1) + w], (unsigned)sft);
        }
    }
    __syncthreads();
    if (!live) return;
    const int p = i * W + j;
    // Hypotheses of a lane: d = 4K * (lane / 4) + 4k + lane % 4, k = 0..K-1.  Slot k at tap group g then needs the
    // dword 4k bytes before slot 0's, i.e. the dword slot 0 needs at group g - k: one LDS read per row position
    // serves every slot (G + K - 1 reads per window row instead of G * K).
    // K > 1 and D <= 64 K - 3: the hypotheses of the wave start at d = ((j - win - xlo) & 3) - 3 (-3 .. 0) instead of 0, so
    // that the four lanes of a group read the four copies at ONE word index: with the set starting at 0 a lane whose
    // byte offset crosses a dword boundary reads one word lower than its neighbours and lands on the bank of another
    // group's lane (28 % of the LDS cycles at D = 200 were conflict cycles).  Lanes with d < 0 are idle.
    const int grp = lane >> 2, rr = lane & 3;
    const int sh = (K > 1 && D <= 64 * K - 3) ? ((j - win - xlo) & 3) - 3 : 0;
    const int d0 = 4 * K * grp + rr + sh;
    unsigned sab[K];
#pragma unroll
    for (int k = 0; k < K; k++) sab[k] = 0u;
    const int o = (j - win - d0) - xlo;                   // >= 4K - 4: xlo <= j0 - win - (64K - 1), d0 <= 60K + 3; at most 64K + 20
    const uint32_t *pb = EL ? s_R + o : s_R + ncc_copy_off<K>(o & 3, CS) + (o >> 2) - (K - 1);
    const uint32_t *ap = s_A + wv * side * G;
    for (int r = 0; r < side; r++) {
        uint32_t bv[G + K - 1];
#pragma unroll
        for (int t = 0; t < G + K - 1; t++) bv[t] = lds_u32(pb + (EL ? 4 * t : t));
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t a = lds_u32(ap + g);           // one address for the whole wave: a broadcast read
#pragma unroll
            for (int k = 0; k < K; k++) sab[k] = __builtin_amdgcn_udot4(a, bv[g - k + K - 1], sab[k], false);
        }
        ap += G;
        pb += EL ? 4 * RW : RW;
    }
    // cost per hypothesis (float64)
    const double n = (double)(side * side);
    const double sa = (double)sumL[p], ra = rootL[p];
    double c[K];
    float v[K];
    bool poison = false;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int d = d0 + 4 * k;
        const bool act = d >= 0 && d < D;
        if (act && j - win - d >= 0) {
            const size_t q = (size_t)p - d;
            const double num = n * (double)sab[k] - sa * (double)sumR[q];
            c[k] = num / (ra * rootR[q]);
        } else c[k] = 255.0;                              // `invalid` 0xff, NCC.h:88
        if (cost_out && act) cost_out[(size_t)p * D + d] = c[k];
        const bool isn = c[k] != c[k];
        if (k == 0) poison = __ballot(isn && d == 0) != 0;   // d = 0 sits in slot 0 of one of the first four lanes
        v[k] = (act && !isn) ? (float)c[k] : -INFINITY;
    }
    // WinTakeAll.  m before step d = max over e < d of (float)c[e] with NaNs skipped (a NaN at d = 0 makes every test
    // false), in d order = (group of 4 lanes, slot, lane in the group):
    //   before (grp, k, rr) = max( all of the groups < grp,  slots < k of this group,  lanes < rr of slot k )
    float exq[K];                                         // the last two terms
    float run = -INFINITY;                                // maximum of the slots < k of this group (same in its 4 lanes)
#pragma unroll
    for (int k = 0; k < K; k++) {
        float inc = v[k];
        float t1 = __shfl_up(inc, 1, 4);
        inc = fmaxf(inc, rr >= 1 ? t1 : -INFINITY);
        float t2 = __shfl_up(inc, 2, 4);
        inc = fmaxf(inc, rr >= 2 ? t2 : -INFINITY);      // inclusive over the lanes of the group
        const float below = __shfl_up(inc, 1, 4);
        exq[k] = fmaxf(run, rr >= 1 ? below : -INFINITY);
        run = fmaxf(run, __shfl(inc, 3, 4));
    }
    const float pref = wave_prefix_max_f32(run, lane);    // run = the group's maximum: inclusive over the groups
    const float prev = __shfl(pref, (4 * grp - 1) & 63, WAVE);
    const float before = grp >= 1 ? prev : -INFINITY;     // all of the groups < grp
    unsigned key = 0;                                     // 1 + the largest winning d of the lane (0: none)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int d = d0 + 4 * k;
        const float m = fmaxf(before, exq[k]);
        if (d >= 0 && d < D && (double)m < c[k]) key = (unsigned)d + 1u;    // d grows with k
    }
    const unsigned kmax = ~wave_min_u32(~key);            // the last winner
    if (lane == 0) disp[p] = (poison || kmax == 0u) ? 0 : (int)(kmax - 1u);
}

// ---------------------------------------------------------------------------------- ASW
// Per hypothesis: cost = sum_q w0(q) w1(q) e(q) / sum_q w0(q) w1(q) over the (2*wins+1)^2 window, with
// w0 = color[|A(q)-A(c)|]*space(q) (anchor window, independent of d), w1 = color[|B(q)-B(c)|]*space(q)
// (other image at the hypothesis' offset), e = min(|A(q)-B(q)|, T)                (ASW.h:210-257).
//
// One wave per pixel, d = lane + 64k.  The anchor weights are the same for all 64 lanes, so the
// lanes compute them 64 taps at a time (one tap per lane) and the tap loop broadcasts
// {w0*space, anchor byte} with v_readlane -- no per-tap uniform LDS traffic.  The only per-lane
// table read is color[|B(q)-B(c)|]; the table is replicated [256][32] in LDS so that lane l
// always hits bank l%32 (conflict-free random access; float32 copy, relative error 6e-8).  Bytes of the other image are
// fetched as (unaligned) dwords, 4 taps per load.
// Arithmetic: m2 = (w0*space)*w1' with w1' = color*space folded as (w0*space*space)*color -- a
// re-association of the reference's (color*space)*(color*space); float64 throughout, well
// inside the 1e-4 tolerance on the float32 cost the reference itself narrows to (:255-256).
constexpr int ANT = 1024;      // 16 pixels (waves) per workgroup share one replicated table

// K = number of live 64-disparity slots per lane (ceil(D/64)), a template parameter so that the
// tap loop is branch-free: columns are processed in dwords, the last (partial) dword is taken
// from columns side-4..side-1 with the weights of already-counted columns zeroed.
template <int K>
__global__ void __launch_bounds__(ANT) k_asw(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp, int H,
                                             int W, int D, int wins, const double *__restrict__ space,
                                             const double *__restrict__ color, int T, int view,
                                             float *__restrict__ disp, float *__restrict__ cost_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int side = 2 * wins + 1;
    double *s_color = (double *)smem;                 // [256][32] replica: entry k of lane l at k*32 + l%32
    double *s_sp2 = s_color + 256 * 32;               // space(q)^2, side*side
    for (int e = threadIdx.x; e < 256 * 32; e += ANT) s_color[e] = color[e >> 5];
    for (int e = threadIdx.x; e < side * side; e += ANT) { const double v = space[e]; s_sp2[e] = v * v; }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * (ANT / 64) + wv;
    if (p >= H * W) return;
    const int io = p / W, jo = p - io * W;
    const int Wp = W + 2 * wins;
    // anchor window: left image for view 0, right image for view 1 (ASW.h:342 / :395)
    const uint8_t *A = (view == 0 ? Lp : Rp) + (size_t)io * Wp + jo;
    const uint8_t *B = (view == 0 ? Rp : Lp) + (size_t)io * Wp + jo;
    // last in-range disparity: left j-wins-d >= 0 (:348); right j+wins+d+1 < Wp-wins (:401)
    const int dmax = (view == 0) ? jo : (W - wins - 2 - jo);
    const int ca = A[wins * Wp + wins];
    const double *lut = s_color + (lane & 31);

    double sw[K], sv[K];
    const uint8_t *bk[K];
    int cb[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int d = lane + 64 * k;
        const int dd = d < dmax ? d : (dmax < 0 ? 0 : dmax);    // d >= D lanes compute a harmless duplicate
        bk[k] = (view == 0) ? B - dd : B + dd;
        cb[k] = bk[k][wins * Wp + wins];
        sw[k] = 0.0; sv[k] = 0.0;
    }
    const int nfull = side >> 2;                       // full dwords per row
    const int ctail = side - 4;                        // start of the tail dword (side >= 5 always: wins >= 2)
    const bool has_tail = (side & 3) != 0;

    auto taps4 = [&](const unsigned (&word)[K], int c0, int whi, int wlo, int pa_l) {
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const double w = __hiloint2double(__builtin_amdgcn_readlane(whi, c0 + cc),
                                              __builtin_amdgcn_readlane(wlo, c0 + cc));
            const int pa = __builtin_amdgcn_readlane(pa_l, c0 + cc);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const unsigned pb = (word[k] >> (8 * cc)) & 0xffu;
                // |x - y| in one v_sad_u16 (operands < 256)
                const double c1 = lut[__builtin_amdgcn_sad_u16(pb, (unsigned)cb[k], 0u) * 32];
                unsigned e = __builtin_amdgcn_sad_u16((unsigned)pa, pb, 0u);
                e = e > (unsigned)T ? (unsigned)T : e;
                const double m2 = w * c1;
                sw[k] = __builtin_fma(w, c1, sw[k]);
                sv[k] = __builtin_fma(m2, (double)e, sv[k]);
            }
        }
    };

    for (int r = 0; r < side; r++) {
        // this row's anchor weights, one column per lane: w = color[|A-ca|] * space^2
        int pa_l = 0; double w_l = 0.0;
        if (lane < side) {
            pa_l = A[r * Wp + lane];
            w_l = s_color[abs(pa_l - ca) * 32 + (lane & 31)] * s_sp2[r * side + lane];
        }
        const int wlo = __double2loint(w_l), whi = __double2hiint(w_l);
        // tail copy: columns already covered by the full dwords get weight 0
        const double w_t = (lane >= 4 * nfull) ? w_l : 0.0;
        const int tlo = __double2loint(w_t), thi = __double2hiint(w_t);
        for (int g = 0; g < nfull; g++) {
            unsigned word[K];
#pragma unroll
            for (int k = 0; k < K; k++) __builtin_memcpy(&word[k], bk[k] + r * Wp + 4 * g, 4);   // unaligned dword
            taps4(word, 4 * g, whi, wlo, pa_l);
        }
        if (has_tail) {
            unsigned word[K];
#pragma unroll
            for (int k = 0; k < K; k++) __builtin_memcpy(&word[k], bk[k] + r * Wp + ctail, 4);
            taps4(word, ctail, thi, tlo, pa_l);
        }
    }
    float cv[K];
#pragma unroll
    for (int k = 0; k < K; k++) cv[k] = (float)(sv[k] / sw[k]);

    // WinTakeAll: first strict minimum (:193-208).  dmax < 0 (right view, last columns): the
    // reference chains every cost to an out-of-bounds read -> all equal -> 0.
    float lm = INFINITY; int ld = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int d = lane + 64 * k;
        if (d < D && lm > cv[k]) { lm = cv[k]; ld = d; }
    }
    const float m = wave_min_f32(lm);
    int cand = (lm == m) ? ld : 0x7fffffff;
    for (int off = 32; off >= 1; off >>= 1) cand = min(cand, __shfl_xor(cand, off, WAVE));
    if (lane == 0) disp[p] = (dmax < 0) ? 0.0f : (float)cand;
    if (cost_out) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int d = lane + 64 * k;
            if (d < D) cost_out[(size_t)p * D + d] = (dmax < 0) ? NAN : cv[k];
        }
    }
}

// ---- ASW, second formulation (default) ---------------------------------------------------------------
// k_asw above is VALU-issue bound (rocprofv3: 98.8 % of the SIMD cycles issue a vector instruction, 11.4 per
// tap and hypothesis slot).  Two of its per-tap costs do not depend on the hypothesis the lane owns:
//   * the anchor weight w0(q)*space(q)^2 and the anchor byte are wave-uniform, yet reach the FMAs through
//     three v_readlane per tap.  k_asw_anchor writes them once per pixel to a table in HBM
//     ([pixel][tap] float64, 5 GB at 960x540 / 35x35, a fraction of a millisecond to write) and a
//     dword-per-pixel copy of the anchor image; the main kernel reads both through the SCALAR cache
//     (uniform addresses, s_load), which costs no vector issue slot at all;
//   * the other image's weight color[|B(q) - B(centre)|] depends only on the window position in the other
//     image, xs = j -+ d, not on (j, d) separately: the 32 pixels of a workgroup (16 waves x 2 pixels) x D
//     hypotheses touch only 32 + D - 1 window positions per row.  Per window row the workgroup builds that
//     table once in LDS (T[xs][column], float64, odd row stride) and every lane reads its entry with one
//     conflict-free ds_read_b64 -- no |difference|, no table-address arithmetic, no random LDS access in the
//     tap loop.  The reads go through lds_f64(): left alone the compiler pairs neighbouring columns into
//     ds_read2_b64, which the LDS serves at half the bytes per clock.
//   * the other image's byte behind a tap depends on xs + column only: the row is staged once per window row as
//     floats in LDS (P[xs + c]); the truncated error min(|pa - pb|, T) is then two f32 instructions on exact
//     small integers (subtract, min with |.| modifier) and one f32 -> f64 conversion -- no byte extraction and
//     no global loads in the tap loop.
// What is left per tap and slot: f32 subtract, f32 min |.|, f32 -> f64, one multiply, two FMAs -- 6 vector
// instructions, 7.0 measured with the table build and the address updates (rocprofv3 SQ_INSTS_VALU), on a
// VALU that is busy 93 % of the kernel's cycles (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)).
// Issue rates measured with tools/valu_rate.hip (16 waves per SIMD, ns per wave64 instruction per SIMD): every
// f64 instruction used here (fma, mul, add, min, cvt_f64_f32) 1.71-1.74 ns = 4 cycles; v_sub_f32 0.90 ns = 2 cycles;
// v_min_f32 with |.| (VOP3) 4 cycles.  Forming the error in f64 from a float64 copy of the row (add + min, one
// instruction less) was tried and measured equal (17.97 against 17.74 ms): more LDS bytes per tap, nothing gained.
// Arithmetic and summation order are those of k_asw (and the results identical bit for bit): the table
// holds the same float64 products w0*space^2, and sw / sv accumulate the taps in the same order.
constexpr int A3P = 16;                                   // pixels (waves) per workgroup

// the anchor image as one dword (float bits) per pixel: scalar / uniform loads want 4-byte elements, and the truncated
// error is formed in f32 (exact small integers)
__global__ void __launch_bounds__(256) k_asw_a32(const uint8_t *__restrict__ Ap, size_t n, unsigned *__restrict__ a32)
{
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        a32[k] = __float_as_uint((float)Ap[k]);
}

// anchor weights of the rows [i0, i0 + nrows) of the image: table entry ((io - i0) * W + jo) * side^2 + tap
__global__ void __launch_bounds__(256) k_asw_anchor(const uint8_t *__restrict__ Ap, int i0, int nrows, int W, int wins,
                                                    const double *__restrict__ space, const double *__restrict__ color,
                                                    double *__restrict__ w0)
{
    // one wave per pixel, 4 pixels per workgroup; the wave walks the pixel's side*side taps 64 at a time, so its
    // stores are whole 512-byte pieces of the table
    const int side = 2 * wins + 1, Wp = W + 2 * wins;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 4 + wv;           // pixel of the band
    if (p >= (size_t)nrows * W) return;
    const int io = i0 + (int)(p / W), jo = (int)(p % W);
    const uint8_t *A = Ap + (size_t)io * Wp + jo;
    const int ca = A[wins * Wp + wins];
    const int ntap = side * side;
    const float rs = 1.0f / (float)side;
    double *out = w0 + p * ntap;
    for (int t = lane; t < ntap; t += 64) {
        int r = (int)((float)t * rs);                      // t / side, corrected below
        int c = t - r * side;
        if (c < 0) { r--; c += side; }
        if (c >= side) { r++; c -= side; }
        const int pa = A[r * Wp + c];
        const double sp = space[t];
        out[t] = color[abs(pa - ca)] * (sp * sp);          // the two products of k_asw, same order
    }
}

// Diagnostic (SMT_ASW_VERIFY=1): recomputes every entry of the two anchor tables with the arithmetic of k_asw_anchor
// and compares it with what memory holds, read with ordinary vector loads.  counts[0] = w0 entries that differ,
// [1] = of those, entries that read as +0.0 (a cleared page), [2] = entries that read as NaN, [3] = a32 entries
// that differ, [4] = first differing w0 index + 1 (atomicMin over a start value of ~0), [5] = last differing index + 1.
__global__ void __launch_bounds__(256) k_asw_anchor_check(const uint8_t *__restrict__ Ap, int H, int W, int wins,
                                                          const double *__restrict__ space, const double *__restrict__ color,
                                                          const double *w0, const unsigned *a32, unsigned long long *counts)
{
    const int side = 2 * wins + 1, Wp = W + 2 * wins, Hp = H + 2 * wins;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < (size_t)Hp * Wp; k += (size_t)gridDim.x * 256)
        if (a32[k] != __float_as_uint((float)Ap[k])) atomicAdd(&counts[3], 1ull);
    const size_t p = (size_t)blockIdx.x * 4 + wv;
    if (p >= (size_t)H * W) return;
    const int io = (int)(p / W), jo = (int)(p % W);
    const uint8_t *A = Ap + (size_t)io * Wp + jo;
    const int ca = A[wins * Wp + wins];
    const int ntap = side * side;
    const double *in = w0 + p * ntap;
    for (int t = lane; t < ntap; t += 64) {
        const int r = t / side, c = t - r * side;
        const int pa = A[r * Wp + c];
        const double sp = space[t];
        const double want = color[abs(pa - ca)] * (sp * sp);
        const double got = in[t];
        if (__double_as_longlong(got) != __double_as_longlong(want)) {
            atomicAdd(&counts[0], 1ull);
            if (__double_as_longlong(got) == 0) atomicAdd(&counts[1], 1ull);
            if (got != got) atomicAdd(&counts[2], 1ull);
            const unsigned long long idx = (unsigned long long)(p * ntap + t) + 1;
            atomicMin(&counts[4], idx);
            atomicMax(&counts[5], idx);
        }
    }
}

// One ds_read_b64 that stays one: the compiler would pair neighbouring columns into ds_read2_b64, which the LDS
// serves at half the bytes per clock of ds_read_b64 (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ double lds_f64(const double *p)
{
    return *(const volatile __attribute__((address_space(3))) double *)p;
}

// A pointer the compiler must treat as wave-uniform (its loads then go through the scalar cache).
template <class P>
__device__ __forceinline__ P uniform_ptr(P p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (P)(((uint64_t)hi << 32) | lo);
}

// A3Q = pixels per wave: the per-row tables are built once for A3P * A3Q pixels
// SLOAD = false (smt_asw_set_impl(5)) reads the two anchor operands with ordinary vector loads from the same
// wave-uniform addresses instead of s_load: same arithmetic, same results, one more formulation for bisecting.
// One tile = NPX consecutive pixels of image row io, starting at column jo0.  OWN = false: w0 is the anchor table of
// the rows [i0, ...) written by k_asw_anchor.  OWN = true (k_asw4): w0 is this workgroup's private slot
// ([A3P waves][A3Q pixels][side^2] doubles) and every wave first writes the anchor weights of its own pixels there --
// the arithmetic of k_asw_anchor -- then reads them back through the scalar cache: vector stores, s_waitcnt vmcnt(0)
// (the stores have reached L2), s_dcache_inv (the scalar cache may hold the previous tile's slot contents), scalar loads.
// No other wave ever touches a wave's part of the slot, so no barrier is involved.
template <int K, int A3Q, bool SLOAD, bool OWN>
__device__ __forceinline__ void asw3_tile(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp, int H,
                                          int W, int D, int wins, const double *__restrict__ space,
                                          double *w0, const unsigned *__restrict__ a32, int T,
                                          int view, float *__restrict__ disp, float *__restrict__ cost_out, int i0, int io, int jo0,
                                          unsigned char *smem)
{
    constexpr int NPX = A3P * A3Q;                         // pixels per workgroup
    constexpr int NXP = NPX + 64 * K;                      // window positions of the other image per row (padded)
    const int side = 2 * wins + 1, Wp = W + 2 * wins;
    double *s_T = (double *)smem;                          // [NXP][side]
    double *s_color = s_T + (size_t)side * NXP;            // [256], filled by the caller
    float *s_P = (float *)(s_color + 256);                 // [NXP + side]: the other image's current window row as floats
    // OWN only: squared spatial mask [side^2] (filled by the caller) and the tile's anchor window [side][SA] bytes
    double *s_sp2 = (double *)(s_P + ((NXP + side + 1) & ~1));
    uint8_t *s_A = (uint8_t *)(s_sp2 + (OWN ? side * side : 0));
    const int SA = NPX + side - 1;

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (OWN) {
        // anchor window of the tile: rows io .. io + side - 1, padded columns jo0 .. jo0 + NPX + side - 2 (clamped)
        const uint8_t *Aimg = (view == 0 ? Lp : Rp) + (size_t)io * Wp;
        for (int e = threadIdx.x; e < side * SA; e += A3P * 64) {
            const int r = e / SA, xx = e - r * SA;
            const int x = jo0 + xx;
            s_A[e] = Aimg[(size_t)r * Wp + (x < Wp ? x : Wp - 1)];
        }
        __syncthreads();
    }
    // the other image and the first window position the workgroup can touch:
    //   view 0: xs = jo - dd in [jo0 - (D-1), jo0 + NPX-1];   view 1: xs = jo + dd in [jo0, jo0 + NPX-1 + D-1]
    const uint8_t *Bimg = (view == 0 ? Rp : Lp) + (size_t)io * Wp;
    const int xbase = (view == 0) ? jo0 - (64 * K - 1) : jo0;

    // the wave's A3Q pixels: jo0 + wv and jo0 + A3P + wv (pixels past the row end still help to build the tables)
    bool live[A3Q];
    int jo[A3Q], dmax[A3Q];
    typedef const __attribute__((address_space(4))) double *cdouble_p;     // constant address space: uniform loads become s_load
    typedef const __attribute__((address_space(4))) unsigned *cunsigned_p;
    cdouble_p wrow[A3Q];
    cunsigned_p arow[A3Q];
    double sw[A3Q][K], sv[A3Q][K];
    const double *tk[A3Q][K];
    const float *pk[A3Q][K];
#pragma unroll
    for (int q = 0; q < A3Q; q++) {
        jo[q] = jo0 + q * A3P + wv;
        live[q] = jo[q] < W;
        const int jc = live[q] ? jo[q] : W - 1;
        dmax[q] = (view == 0) ? jc : (W - wins - 2 - jc);  // last in-range disparity (ASW.h:348 / :401)
        double *wtab = OWN ? w0 + ((size_t)wv * A3Q + q) * side * side : w0 + ((size_t)(io - i0) * W + jc) * side * side;
        wrow[q] = (cdouble_p)wtab;
        if (OWN) {
            // this pixel's anchor weights: color[|A(q) - A(centre)|] * space(q)^2, the two products of k_asw in its
            // order, from the tile's anchor window and the squared spatial mask staged in LDS by the caller
            const int xx0 = jc - jo0;
            const int ca = s_A[wins * SA + xx0 + wins];
            const int ntap = side * side;
            const float rs = 1.0f / (float)side;
#pragma unroll 4
            for (int t = lane; t < ntap; t += 64) {
                int r = (int)((float)t * rs);              // t / side, corrected below
                int c = t - r * side;
                if (c < 0) { r--; c += side; }
                if (c >= side) { r++; c -= side; }
                wtab[t] = s_color[abs((int)s_A[r * SA + xx0 + c] - ca)] * s_sp2[t];
            }
        }
        arow[q] = (cunsigned_p)(a32 + (size_t)io * Wp + jc);
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int d = lane + 64 * k;
            const int dd = d < dmax[q] ? d : (dmax[q] < 0 ? 0 : dmax[q]);   // d >= D lanes compute a harmless duplicate
            const int xs = (view == 0) ? jc - dd : jc + dd;
            tk[q][k] = s_T + (size_t)(xs - xbase) * side;  // row e of T[e][c]: the row stride (side, odd) keeps b64 reads conflict-free
            pk[q][k] = s_P + (xs - xbase);
            sw[q][k] = 0.0; sv[q][k] = 0.0;
        }
    }
    if (OWN) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the table stores have completed (L2 has them)
        __builtin_amdgcn_s_dcache_inv();                   // the scalar cache may still hold this slot's previous contents
        // the loads below go through constant-address-space pointers, which the compiler may treat as never written:
        // make the pointers opaque here so that no load of the table can be scheduled above this point
#pragma unroll
        for (int q = 0; q < A3Q; q++) {
            uint64_t v = (uint64_t)wrow[q];
            asm volatile("" : "+v"(v) : : "memory");
            wrow[q] = (cdouble_p)v;
        }
    }
    const float Tf = (float)T;
    // table builders: thread t owns the window position e = t % NXP and the columns t / NXP, + ngrp, ...; its
    // centre byte does not depend on the window row
    constexpr int NGRP = (A3P * 64) / NXP;
    const int te = (int)threadIdx.x % NXP, tg = (int)threadIdx.x / NXP;
    int tx = xbase + te;
    tx = tx < 0 ? 0 : (tx > Wp - side ? Wp - side : tx);   // positions no hypothesis uses: any in-image value
    const unsigned tcb = Bimg[(size_t)wins * Wp + tx + wins];

    for (int r = 0; r < side; r++) {
        __syncthreads();                                   // previous row's tables are no longer read (and s_color is in)
        // T[e][c] = color[|B[r][x + c] - B[wins][x + wins]|] for the window position x = xbase + e;
        // P[xx] = (float)B[r][xbase + xx]: the byte behind tap column c of window position e sits at xx = e + c
        if (tg < NGRP) {
            const uint8_t *brow = Bimg + (size_t)r * Wp + tx;
            for (int c = tg; c < side; c += NGRP)
                s_T[te * side + c] = s_color[__builtin_amdgcn_sad_u16((unsigned)brow[c], tcb, 0u)];
        }
        if ((int)threadIdx.x < NXP + side) {
            int x = xbase + (int)threadIdx.x;
            x = x < 0 ? 0 : (x > Wp - 1 ? Wp - 1 : x);
            s_P[threadIdx.x] = (float)Bimg[(size_t)r * Wp + x];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < A3Q; q++) {
            cdouble_p wr = uniform_ptr(wrow[q] + r * side);   // wave-uniform operands through the scalar cache
            cunsigned_p ar = uniform_ptr(arow[q] + (size_t)r * Wp);
            const double *wrv = (OWN ? w0 + ((size_t)wv * A3Q + q) * side * side
                                     : w0 + ((size_t)(io - i0) * W + (live[q] ? jo[q] : W - 1)) * side * side) + (size_t)r * side;
            const unsigned *arv = a32 + ((size_t)io + r) * Wp + (live[q] ? jo[q] : W - 1);
            auto tap = [&](int c) {
                const double w = SLOAD ? wr[c] : __builtin_nontemporal_load(wrv + c);
                const float pa = __uint_as_float(SLOAD ? ar[c] : __builtin_nontemporal_load(arv + c));
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const double c1 = lds_f64(tk[q][k] + c);
                    const float e = fminf(fabsf(pa - pk[q][k][c]), Tf);   // min(|pa - pb|, T), exact in f32
                    const double m2 = w * c1;
                    sw[q][k] = __builtin_fma(w, c1, sw[q][k]);
                    sv[q][k] = __builtin_fma(m2, (double)e, sv[q][k]);
                }
            };
            int c = 0;
            for (; c + 8 <= side; c += 8) {
#pragma unroll
                for (int cc = 0; cc < 8; cc++) tap(c + cc);
            }
            for (; c < side; c++) tap(c);
        }
    }
#pragma unroll
    for (int q = 0; q < A3Q; q++) {
        if (!live[q]) continue;
        float cv[K];
#pragma unroll
        for (int k = 0; k < K; k++) cv[k] = (float)(sv[q][k] / sw[q][k]);
        // WinTakeAll: first strict minimum (:193-208); dmax < 0: see k_asw
        float lm = INFINITY; int ld = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int d = lane + 64 * k;
            if (d < D && lm > cv[k]) { lm = cv[k]; ld = d; }
        }
        const float m = wave_min_f32(lm);
        int cand = (lm == m) ? ld : 0x7fffffff;
        for (int off = 32; off >= 1; off >>= 1) cand = min(cand, __shfl_xor(cand, off, WAVE));
        const size_t p = (size_t)io * W + jo[q];
        if (lane == 0) disp[p] = (dmax[q] < 0) ? 0.0f : (float)cand;
        if (cost_out) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int d = lane + 64 * k;
                if (d < D) cost_out[p * D + d] = (dmax[q] < 0) ? NAN : cv[k];
            }
        }
    }
}

template <int K, int A3Q, bool SLOAD = true>
__global__ void __launch_bounds__(A3P * 64, (K * A3Q <= 4 ? 8 : 4)) k_asw3(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp, int H,
                                                   int W, int D, int wins, const double *__restrict__ color,
                                                   const double *__restrict__ w0, const unsigned *__restrict__ a32, int T,
                                                   int view, float *__restrict__ disp, float *__restrict__ cost_out, int i0)
{
    // i0: first image row of this launch's band; w0 holds the anchor weights of the band's rows only
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPX = A3P * A3Q, NXP = NPX + 64 * K;
    double *s_color = (double *)smem + (size_t)(2 * wins + 1) * NXP;
    for (int e = threadIdx.x; e < 256; e += A3P * 64) s_color[e] = color[e];
    asw3_tile<K, A3Q, SLOAD, false>(Lp, Rp, H, W, D, wins, nullptr, const_cast<double *>(w0), a32, T, view, disp, cost_out, i0,
                                    i0 + (int)blockIdx.y, (int)blockIdx.x * NPX, smem);
}

// The same tiles from a fixed number of workgroups that each own one slot of anchor-weight scratch (fourth
// formulation, default): workgroup b takes tiles b, b + gridDim.x, ... of the row-major tile order and rebuilds its
// slot for every tile, so the anchor table is O(workgroups in flight) -- 160 MB at 35 x 35 whatever the image size --
// instead of O(image) (5 GB at 960 x 540), and no separate table kernel runs.  Results are those of k_asw3, bit for bit.
template <int K, int A3Q>
__global__ void __launch_bounds__(A3P * 64, (K * A3Q <= 4 ? 8 : 4)) k_asw4(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp, int H,
                                                   int W, int D, int wins, const double *__restrict__ space, const double *__restrict__ color,
                                                   double *__restrict__ slots, const unsigned *__restrict__ a32, int T,
                                                   int view, float *__restrict__ disp, float *__restrict__ cost_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPX = A3P * A3Q, NXP = NPX + 64 * K;
    const int side = 2 * wins + 1;
    double *s_color = (double *)smem + (size_t)side * NXP;
    float *s_P = (float *)(s_color + 256);
    double *s_sp2 = (double *)(s_P + ((NXP + side + 1) & ~1));
    for (int e = threadIdx.x; e < 256; e += A3P * 64) s_color[e] = color[e];
    for (int e = threadIdx.x; e < side * side; e += A3P * 64) { const double sp = space[e]; s_sp2[e] = sp * sp; }
    // (the anchor phase of a tile reads both after the barrier that follows its window staging)
    const int tpr = (W + NPX - 1) / NPX, ntiles = tpr * H;
    double *slot = slots + (size_t)blockIdx.x * NPX * side * side;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asw3_tile<K, A3Q, true, true>(Lp, Rp, H, W, D, wins, space, slot, a32, T, view, disp, cost_out, 0, tile / tpr,
                                      (tile % tpr) * NPX, smem);
        __syncthreads();                                   // the tile's LDS tables are no longer read
    }
}


}  // namespace

// 0 (default): 3 while the whole-image anchor table stays under kAswTableMax bytes, 6 beyond; 3: k_asw_anchor + k_asw3 over
// a whole-image table, two pixels per wave; 6: k_asw4, anchor weights in per-workgroup slots; 4: 3 with one pixel per wave;
// 5: 3 with vector loads of the anchor operands; 1: k_asw (first formulation)
static int g_asw_impl = 0;
SMT_API int smt_asw_set_impl(int impl)
{
    if (impl != 0 && impl != 1 && impl != 3 && impl != 4 && impl != 5 && impl != 6) return SMT_ERR_ARG;
    g_asw_impl = impl;
    return SMT_OK;
}

static int g_sad_impl = 2;   // 2: k_sad2 (rows staged in LDS as byte-shifted dword copies; default), 1: k_sad (one wave per pixel from global memory)
SMT_API int smt_sad_set_impl(int impl)
{
    if (impl != 1 && impl != 2) return SMT_ERR_ARG;
    g_sad_impl = impl;
    return SMT_OK;
}

SMT_API int smt_sad(const uint8_t *Lp, const uint8_t *Rp, int H, int W, int D, int winsize, int view,
                    int32_t *disp, void *stream)
{
    if (!Lp || !Rp || !disp || H <= 0 || W <= 0 || D <= 0 || D > 256 || winsize < 0 ||
        (view != SMT_VIEW_LEFT && view != SMT_VIEW_RIGHT))
        return SMT_ERR_ARG;
    const int N = H * W;
    const int w = winsize + 1, side = 2 * w + 1, K = (D + 63) / 64, v = view == SMT_VIEW_LEFT ? 0 : 1;
    // dword columns staged per row (rounded up to 8 mod 32, a leftover of the separate-copies layout): anchor bytes jo0 .. jo0 + STP + side - 2, other image
    // 64 K - 1 more
    auto copy_words = [](int bytes) { int n = (bytes + 3) / 4 + 1; return n + ((8 - n % 32) % 32 + 32) % 32; };
    const int ALW = copy_words(STP + side - 1), OLW = copy_words(STP + side - 1 + 64 * K - 1);
    const size_t shm = (size_t)side * 4 * (ALW + OLW) * 4;
    if (g_sad_impl == 2 && side >= 4 && shm <= 64 * 1024) {
        const dim3 grid((W + STP - 1) / STP, H);
        switch (K) {
        case 1: hipLaunchKernelGGL(k_sad2<1>, grid, dim3(NT), shm, smt_stream(stream), Lp, Rp, H, W, D, w, v, disp, ALW, OLW); break;
        case 2: hipLaunchKernelGGL(k_sad2<2>, grid, dim3(NT), shm, smt_stream(stream), Lp, Rp, H, W, D, w, v, disp, ALW, OLW); break;
        case 3: hipLaunchKernelGGL(k_sad2<3>, grid, dim3(NT), shm, smt_stream(stream), Lp, Rp, H, W, D, w, v, disp, ALW, OLW); break;
        default: hipLaunchKernelGGL(k_sad2<4>, grid, dim3(NT), shm, smt_stream(stream), Lp, Rp, H, W, D, w, v, disp, ALW, OLW); break;
        }
        SMT_LAUNCH_CHECK();
        return SMT_OK;
    }
    hipLaunchKernelGGL(k_sad, dim3((N + 3) / 4), dim3(NT), 0, smt_stream(stream), Lp, Rp, H, W, D, w, v, disp);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

static int g_ncc_impl = 2;   // 2: k_ncc_stats + k_ncc2 (default); 1: k_ncc (the reference's loop nest, also the fallback for windows wider than 31)
SMT_API int smt_ncc_set_impl(int impl)
{
    if (impl != 1 && impl != 2) return SMT_ERR_ARG;
    g_ncc_impl = impl;
    return SMT_OK;
}

template <int K>
static int launch_ncc2(int G, dim3 grid, size_t shm, hipStream_t st, const uint8_t *L, const uint8_t *R, int H, int W, int D,
                       int win, const int *sumL, const double *rootL, const int *sumR, const double *rootR, int RW, int CS,
                       int32_t *disp, double *cost)
{
#define SMT_NCC2(GG)                                                                                          \
    case GG:                                                                                                  \
        SMT_HIP(hipFuncSetAttribute((const void *)k_ncc2<K, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((k_ncc2<K, GG>), grid, dim3(NCP * 64), shm, st, L, R, H, W, D, win, sumL, rootL, sumR, rootR, RW, CS, disp, cost); \
        break
    switch (G) {
        SMT_NCC2(1); SMT_NCC2(2); SMT_NCC2(3); SMT_NCC2(4); SMT_NCC2(5); SMT_NCC2(6); SMT_NCC2(7); SMT_NCC2(8);
    default: return SMT_ERR_ARG;
    }
#undef SMT_NCC2
    return SMT_OK;
}

SMT_API int smt_ncc(const uint8_t *L, const uint8_t *R, int H, int W, int D, int winSize, int32_t *disp,
                    double *cost, void *stream)
{
    if (!L || !R || !disp || H <= 0 || W <= 0 || D <= 0 || D > 256 || winSize < 0) return SMT_ERR_ARG;
    const int N = H * W;
    hipStream_t st = smt_stream(stream);
    const int side = 2 * winSize + 1, Hi = H - 2 * winSize, Wi = W - 2 * winSize;
    static const int env_impl = [] { const char *e = getenv("SMT_NCC_IMPL"); return e ? atoi(e) : 0; }();   // debugging aid: overrides smt_ncc_set_impl
    if ((env_impl ? env_impl : g_ncc_impl) == 2 && side <= 31) {
        SMT_HIP(hipMemsetAsync(disp, 0, (size_t)N * 4, st));             // border pixels: 0, like k_ncc
        if (Hi <= 0 || Wi <= 0) return SMT_OK;
        int *sums = nullptr;
        double *roots = nullptr;
        if (smt_scratch_alloc((void **)&roots, (size_t)N * 8 * 2, st) == hipSuccess) {
            if (smt_scratch_alloc((void **)&sums, (size_t)N * 4 * 2, st) != hipSuccess) { smt_scratch_free(roots, st); return SMT_ERR_ALLOC; }
            const int RR = NCT + 2 * winSize, RC = 64 + 2 * winSize;
            const size_t shm1 = (((size_t)RR * RC + 15) & ~(size_t)15) + (size_t)RR * 64 * 8;
            hipLaunchKernelGGL(k_ncc_stats, dim3((Wi + 63) / 64, (Hi + NCT - 1) / NCT, 2), dim3(256), shm1, st, L, R, H, W, winSize,
                               sums, roots, sums + N, roots + N);
            const int K = (D + 63) / 64, G = (side + 3) / 4;
            const int RW = 16 * K + G + 5, CS = (side * RW + 31) / 32 * 32 + 32;   // + 32: room for the copies' bank offsets
            const size_t shm2 = ((size_t)4 * CS + (size_t)side * (RW + 1) + (size_t)NCP * side * G) * 4;
            const dim3 grid((Wi + NCP - 1) / NCP, Hi);
            int rc;
            switch (K) {
            case 1: rc = launch_ncc2<1>(G, grid, shm2, st, L, R, H, W, D, winSize, sums, roots, sums + N, roots + N, RW, CS, disp, cost); break;
            case 2: rc = launch_ncc2<2>(G, grid, shm2, st, L, R, H, W, D, winSize, sums, roots, sums + N, roots + N, RW, CS, disp, cost); break;
            case 3: rc = launch_ncc2<3>(G, grid, shm2, st, L, R, H, W, D, winSize, sums, roots, sums + N, roots + N, RW, CS, disp, cost); break;
            default: rc = launch_ncc2<4>(G, grid, shm2, st, L, R, H, W, D, winSize, sums, roots, sums + N, roots + N, RW, CS, disp, cost); break;
            }
            smt_scratch_free(sums, st);
            smt_scratch_free(roots, st);
            if (rc != SMT_OK) return rc;
            SMT_LAUNCH_CHECK();
            return SMT_OK;
        }
        // no scratch memory: the first formulation below
    }
    hipLaunchKernelGGL(k_ncc, dim3((N + 3) / 4), dim3(NT), 0, st, L, R, H, W, D, winSize, disp, cost);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_asw_masks(int winSize, double sigma_s, double sigma_c, double *space, double *color)
{
    if (!space || !color || winSize < 0) return SMT_ERR_ARG;
    const int side = 2 * winSize + 3, c = (side - 1) / 2;
    for (int i = 0; i < side; i++) {
        const double y = (double)((i - c) * (i - c));                     // pow(i - center_h, 2), ASW.h:26
        for (int j = 0; j < side; j++) {
            const double x = (double)((j - c) * (j - c));
            space[i * side + j] = exp(-(x + y) / (2 * sigma_s * sigma_s)); // :30
        }
    }
    for (int i = 0; i < 256; i++) color[i] = exp(-(i * i) / (2 * sigma_c * sigma_c)); // :44
    return SMT_OK;
}

SMT_API int smt_asw(const uint8_t *Lp, const uint8_t *Rp, int H, int W, int D, int winSize, const double *space,
                    const double *color, int T, int view, float *disp, float *cost, void *stream)
{
    if (!Lp || !Rp || !space || !color || !disp || H <= 0 || W <= 0 || D <= 0 || D > 256 || winSize < 0 ||
        (view != SMT_VIEW_LEFT && view != SMT_VIEW_RIGHT))
        return SMT_ERR_ARG;
    const int wins = winSize + 1, side = 2 * wins + 1;
    if (side > 64) return SMT_ERR_ARG;                   // one window row per wave pass
    const int N = H * W;
    const int v = view == SMT_VIEW_LEFT ? 0 : 1;
    static const int env_impl = [] { const char *e = getenv("SMT_ASW_IMPL"); return e ? atoi(e) : 0; }();   // debugging aid: overrides smt_asw_set_impl
    int impl = env_impl ? env_impl : g_asw_impl;
    if (impl == 0) {
        // the table kernels are 7 % faster at config 4 (17.6 against 18.9 ms), the slots need 160 MB whatever the image:
        // table up to SMT_ASW_TABLE_MAX_MB (default 6 144: config 4 takes 5 085), slots beyond (1080p at 35x35: 22.7 GB)
        static const size_t kAswTableMax = [] { const char *e = getenv("SMT_ASW_TABLE_MAX_MB"); return (size_t)(e ? atoi(e) : 6144) << 20; }();
        impl = (size_t)N * side * side * 8 <= kAswTableMax ? 3 : 6;
    }
    if (impl >= 3 && side >= 5) {
        const int A3Q = impl == 4 ? 1 : 2;
        const int K = (D + 63) / 64, NPX = A3P * A3Q, NXP = NPX + 64 * K;
        const size_t shm3 = ((size_t)side * NXP + 256) * 8 + (size_t)(NXP + side) * 4;
        const size_t ntap = (size_t)side * side, na = (size_t)(H + 2 * wins) * (W + 2 * wins);
        // k_asw4 adds the squared spatial mask and the tile's anchor window to the LDS image of k_asw3
        const size_t shm4 = ((size_t)side * NXP + 256) * 8 + (size_t)((NXP + side + 1) & ~1) * 4 + ntap * 8 +
                            (((size_t)side * (NPX + side - 1) + 15) & ~(size_t)15);
        hipStream_t st = smt_stream(stream);
        static const bool verify = [] { const char *e = getenv("SMT_ASW_VERIFY"); return e && atoi(e) != 0; }();
        const bool slots = impl == 6 && !verify;          // the check kernel compares a whole-image table
        // impl 6: one slot of anchor weights per workgroup in flight (k_asw4); impl 3 / 4 / 5: the whole-image table
        const int tpr = (W + NPX - 1) / NPX;
        int nwg = tpr * H;
        if (slots) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            const int per_cu = (K * A3Q <= 4) ? 2 : 1;   // what the kernel's launch bounds and its LDS tables admit
            if (nwg > cus * per_cu) nwg = cus * per_cu;
        }
        const size_t tab_bytes = slots ? (size_t)nwg * NPX * ntap * 8 : (size_t)N * ntap * 8;
        double *w0 = nullptr;
        unsigned *a32 = nullptr;
        bool have = (slots ? shm4 : shm3) <= 160 * 1024 && smt_scratch_alloc((void **)&w0, tab_bytes, st) == hipSuccess;
        if (have && smt_scratch_alloc((void **)&a32, na * 4, st) != hipSuccess) { smt_scratch_free(w0, st); w0 = nullptr; have = false; }
        if (have) {
            int rc = SMT_OK;                              // one exit path: the scratch is freed whatever happens
            const uint8_t *Ap = v == 0 ? Lp : Rp;
            unsigned long long *counts = nullptr;
            auto check_tables = [&](const char *when) {
                // diagnostic only: a synchronising read-back of the comparison of both tables with their definition
                unsigned long long h[6] = {0, 0, 0, 0, ~0ull, 0};
                const size_t nt = (size_t)N * ntap;
                if (hipMemcpyAsync(counts, h, sizeof h, hipMemcpyHostToDevice, st) != hipSuccess) return;
                hipLaunchKernelGGL(k_asw_anchor_check, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, Ap, H, W, wins, space, color, w0, a32, counts);
                if (hipMemcpyAsync(h, counts, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
                fprintf(stderr, "SMT_ASW_VERIFY %s: w0=%p..+%zu a32=%p view=%d  w0_bad=%llu (zero=%llu nan=%llu first=%lld last=%lld of %zu) a32_bad=%llu\n",
                        when, (void *)w0, nt * 8, (void *)a32, v, h[0], h[1], h[2], h[0] ? (long long)h[4] - 1 : -1ll,
                        h[0] ? (long long)h[5] - 1 : -1ll, nt, h[3]);
            };
            if (verify && hipMalloc((void **)&counts, 64) != hipSuccess) counts = nullptr;
            hipLaunchKernelGGL(k_asw_a32, dim3((unsigned)((na + 1023) / 1024 < 4096 ? (na + 1023) / 1024 : 4096)), dim3(256), 0, st, Ap, na, a32);
            if (!slots) {
                hipLaunchKernelGGL(k_asw_anchor, dim3((unsigned)(((size_t)N + 3) / 4)), dim3(256), 0, st, Ap, 0, H, W, wins, space, color, w0);
                if (counts) check_tables("after k_asw_anchor");
            }
            const dim3 grid(tpr, H);
#define SMT_ASW3(KK, QQ, SL)                                                                                 \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute((const void *)k_asw3<KK, QQ, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm3); \
        if (e_ != hipSuccess) { g_smt_last_hip = (int)e_; rc = SMT_ERR_HIP; break; }                         \
        hipLaunchKernelGGL((k_asw3<KK, QQ, SL>), grid, dim3(A3P * 64), shm3, st, Lp, Rp, H, W, D, wins, color, w0, a32, T, v, disp, cost, 0); \
    } while (0)
#define SMT_ASW4(KK)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute((const void *)k_asw4<KK, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm4); \
        if (e_ != hipSuccess) { g_smt_last_hip = (int)e_; rc = SMT_ERR_HIP; break; }                         \
        hipLaunchKernelGGL((k_asw4<KK, 2>), dim3((unsigned)nwg), dim3(A3P * 64), shm4, st, Lp, Rp, H, W, D, wins, space, color, w0, a32, T, v, disp, cost); \
    } while (0)
#define SMT_ASW3K(KK)                                                                                        \
    do {                                                                                                     \
        if (slots) SMT_ASW4(KK);                                                                             \
        else if (impl == 5) SMT_ASW3(KK, 2, false);                                                          \
        else if (A3Q == 2) SMT_ASW3(KK, 2, true);                                                            \
        else SMT_ASW3(KK, 1, true);                                                                          \
    } while (0)
            switch (K) {
            case 1: SMT_ASW3K(1); break;
            case 2: SMT_ASW3K(2); break;
            case 3: SMT_ASW3K(3); break;
            default: SMT_ASW3K(4); break;
            }
#undef SMT_ASW3K
#undef SMT_ASW4
#undef SMT_ASW3
            if (counts) { check_tables("after k_asw3"); (void)hipFree(counts); }
            smt_scratch_free(w0, st);
            smt_scratch_free(a32, st);
            if (rc != SMT_OK) return rc;
            SMT_LAUNCH_CHECK();
            return SMT_OK;
        }
        // table does not fit (LDS or device memory): the first formulation below
    }
    const size_t shm = (size_t)(256 * 32 + side * side) * 8;
#define SMT_ASW(KK)                                                                                          \
    do {                                                                                                     \
        SMT_HIP(hipFuncSetAttribute((const void *)k_asw<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL(k_asw<KK>, dim3((N + 15) / 16), dim3(ANT), shm, smt_stream(stream), Lp, Rp, H, W, D, wins, \
                           space, color, T, v, disp, cost);                                                  \
    } while (0)
    switch ((D + 63) / 64) {
    case 1: SMT_ASW(1); break;
    case 2: SMT_ASW(2); break;
    case 3: SMT_ASW(3); break;
    default: SMT_ASW(4); break;
    }
#undef SMT_ASW
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

// ---- batch variants: the single-pair entry points pair by pair on the caller's stream --------------
SMT_API int smt_sad_batch(const uint8_t *Lp, const uint8_t *Rp, int pairs, size_t img_stride, int H, int W, int D,
                          int winsize, int view, int32_t *disp, size_t disp_stride, void *stream)
{
    if (!Lp || !Rp || !disp || pairs <= 0 || H <= 0 || W <= 0 || winsize < 0) return SMT_ERR_ARG;
    const size_t w = (size_t)winsize + 1;
    const size_t is = img_stride ? img_stride : ((size_t)H + 2 * w) * ((size_t)W + 2 * w), ds = disp_stride ? disp_stride : (size_t)H * W;
    for (int b = 0; b < pairs; b++) {
        const int rc = smt_sad(Lp + b * is, Rp + b * is, H, W, D, winsize, view, disp + b * ds, stream);
        if (rc != SMT_OK) return rc;
    }
    return SMT_OK;
}

SMT_API int smt_ncc_batch(const uint8_t *L, const uint8_t *R, int pairs, size_t img_stride, int H, int W, int D, int winSize,
                          int32_t *disp, size_t disp_stride, void *stream)
{
    if (!L || !R || !disp || pairs <= 0 || H <= 0 || W <= 0) return SMT_ERR_ARG;
    const size_t is = img_stride ? img_stride : (size_t)H * W, ds = disp_stride ? disp_stride : (size_t)H * W;
    for (int b = 0; b < pairs; b++) {
        const int rc = smt_ncc(L + b * is, R + b * is, H, W, D, winSize, disp + b * ds, nullptr, stream);
        if (rc != SMT_OK) return rc;
    }
    return SMT_OK;
}

SMT_API int smt_asw_batch(const uint8_t *Lp, const uint8_t *Rp, int pairs, size_t img_stride, int H, int W, int D, int winSize,
                          const double *space, const double *color, int T, int view, float *disp, size_t disp_stride,
                          void *stream)
{
    if (!Lp || !Rp || !disp || pairs <= 0 || H <= 0 || W <= 0 || winSize < 0) return SMT_ERR_ARG;
    const size_t w = (size_t)winSize + 1;
    const size_t is = img_stride ? img_stride : ((size_t)H + 2 * w) * ((size_t)W + 2 * w), ds = disp_stride ? disp_stride : (size_t)H * W;
    for (int b = 0; b < pairs; b++) {
        const int rc = smt_asw(Lp + b * is, Rp + b * is, H, W, D, winSize, space, color, T, view, disp + b * ds, nullptr, stream);
        if (rc != SMT_OK) return rc;
    }
    return SMT_OK;
}
