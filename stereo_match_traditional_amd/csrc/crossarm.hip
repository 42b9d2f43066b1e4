// Cross-arm lengths + rectangle-mean aggregation -- replaces class CrossArmAggregation
// (AD-CensusV1/CrossArm.{h,cpp}) and CBLSM.h ArmLength{L,R,Up,Down} / costAggregationV5 /
// ComputeAD(Right).
//
// Arms.  The reference walks pixels row-major with ONE mutable threshold (`_tao`,
// CrossArm.h:34): it drops from tau to tau_low the first time any arm enters iteration
// sec+1 and never rises again -- across pixels and (member state) across the four
// direction calls.  Parallel form, two kernels:
//   k_arm_flip : per direction, F[dir] = min row-major index of a pixel whose neighbours
//                1..sec are all inside the image and all within the INITIAL tau (that is
//                exactly "enters iteration sec+1 while the threshold is still tau":
//                the flip at CrossArm.cpp:223-225 precedes the bounds test of neighbour
//                sec+1).
//   k_arms     : threshold entering direction dir = tau_low if chained and any earlier
//                direction has F < INF, else tau; pixels before F use it throughout, pixel
//                F switches to tau_low from k = sec+1, pixels after F use tau_low.
// ComputeRightArmLength's `col = _row` (CrossArm.cpp:265) is reproduced: iteration and
// bounds over j < H, store stride H, image reads with the true width.
//
// Aggregation.  Sequential float adds in the reference's own order (columns outer / rows
// inner for AggregationVertical, CrossArm.cpp:88-95; rows outer for costAggregationV5,
// CBLSM.h:1210-1216), because integral images would change the rounding.  One wavefront
// spans the disparity axis of a pixel (lane l owns C consecutive d), so every tap is one
// coalesced 64*C*4-byte read that neighbouring pixels' waves re-read from L2.
#include "smt_common.h"
#include <stdlib.h>
#include <limits.h>
#include <new>
#include <type_traits>

namespace {

constexpr int NT = 256;

struct ArmCfg {
    int H, W, ch;
    int tau, tau_low, sec, maxlen, chain, fix_right;
};

__device__ __forceinline__ int pix_diff(const uint8_t *img, int ch, int a, int b)
{
    if (ch == 1) return abs((int)img[a] - (int)img[b]);
    int m = 0;
    for (int c = 0; c < ch; c++) {
        const int v = abs((int)img[a * ch + c] - (int)img[b * ch + c]);
        m = v > m ? v : m;
    }
    return m;
}

// direction geometry: returns whether neighbour k of (i,j) is inside, and its coordinates
__device__ __forceinline__ bool arm_nb(int dir, int i, int j, int k, int H, int colR, int &ni, int &nj)
{
    ni = i; nj = j;
    switch (dir) {
    case 0: nj = j - k; return nj >= 0;
    case 1: nj = j + k; return nj < colR;
    case 2: ni = i - k; return ni >= 0;
    default: ni = i + k; return ni < H;
    }
}

// tau_state: the threshold as the previous call left it (`_tao`, CrossArm.h:34).  The fused four-direction
// call resets it first and derives the chaining from flip[0..dir-1] (prev_flips = 1); the one-direction
// calls (dir0 = that direction, gridDim.y = 1, prev_flips = 0) read it here and k_tau_update advances it.
__global__ void __launch_bounds__(NT) k_arm_flip(const uint8_t *__restrict__ img, ArmCfg c, int *flip, int dir0,
                                                 const int *__restrict__ tau_state)
{
    const int dir = dir0 + blockIdx.y;
    if (*tau_state != c.tau) return;                    // already lowered: no flip left to find
    const int colR = (dir == 1 && !c.fix_right) ? c.H : c.W;
    const int idx = blockIdx.x * NT + threadIdx.x;
    // only the smallest qualifying index matters: workgroups behind a candidate already found have
    // nothing to add (workgroups are dispatched in index order, so on images that flip early almost
    // all of them leave here)
    if ((int)(blockIdx.x * NT) > *(volatile int *)&flip[dir]) return;
    bool ok = idx < c.H * colR;
    const int i = ok ? idx / colR : 0, j = ok ? idx - i * colR : 0;
    for (int k = 1; k <= c.sec && ok; k++) {
        int ni, nj;
        ok = arm_nb(dir, i, j, k, c.H, colR, ni, nj);
        if (ok) ok = pix_diff(img, c.ch, i * c.W + j, ni * c.W + nj) <= c.tau;
    }
    // lanes are in index order: the wave's candidate is its first set lane
    const unsigned long long b = __ballot(ok);
    if (b && (threadIdx.x & 63) == 0) {
        const int cand = idx + __builtin_ctzll(b);
        if (cand < *(volatile int *)&flip[dir]) atomicMin(&flip[dir], cand);
    }
}

__global__ void __launch_bounds__(NT) k_arms(const uint8_t *__restrict__ img, ArmCfg c,
                                             const int *__restrict__ flip, int *armL, int *armR,
                                             int *armT, int *armB, int dir0, int prev_flips,
                                             const int *__restrict__ tau_state)
{
    const int dir = dir0 + blockIdx.y;
    const int colR = (dir == 1 && !c.fix_right) ? c.H : c.W;
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= c.H * colR) return;
    const int i = idx / colR, j = idx - i * colR;

    int tau_in = *tau_state;
    if (c.chain && prev_flips)
        for (int e = 0; e < dir; e++)
            if (flip[e] != INT_MAX) tau_in = c.tau_low;
    const int F = flip[dir];
    const int tauA = (tau_in == c.tau && idx <= F) ? c.tau : c.tau_low;   // k <= sec
    const int tauB = c.tau_low;                                           // k  > sec

    bool far;
    switch (dir) {
    case 0: far = (j - 1 >= 1); break;
    case 1: far = (j + 1 < colR - 1); break;
    case 2: far = (i - 1 >= 1); break;
    default: far = (i + 1 < c.H - 1); break;
    }
    int saved = 0;
    for (int k = 1;; k++) {
        saved = k - 1;
        if (k > c.sec && k > c.maxlen) break;
        int ni, nj;
        if (!arm_nb(dir, i, j, k, c.H, colR, ni, nj)) break;
        const int tau = (k > c.sec) ? tauB : tauA;
        if (pix_diff(img, c.ch, i * c.W + j, ni * c.W + nj) > tau) {
            if (far && saved < 1) saved = 1;
            break;
        }
    }
    int *out = dir == 0 ? armL : dir == 1 ? armR : dir == 2 ? armT : armB;
    out[(size_t)i * colR + j] = saved;
}

// ---- arms without the dependent-load chain ---------------------------------------------------------
// k_arm_flip / k_arms above walk an arm neighbour by neighbour and stop at the first failure: up to 34
// DEPENDENT byte loads per thread (0.28 ms per 1080p image for the two kernels).  For max(max_length,
// sec_length) <= 63 the walk is replaced by bit masks: every neighbour 1..K of the pixel is loaded
// unconditionally (independent loads), bit k-1 of m_hi / m_lo says |I(p) - I(p + k*delta)| > tau / tau_low,
// the image border is one more stop bit, and the reference's result for either threshold state is a
// count-trailing-zeros away (A.4 of SURVEY.md):
//   phase A, k <= sec, threshold tA (tau, or tau_low once lowered): first stop bit f -> saved = f
//            (forced to 1 when the stop is a difference, saved < 1 and the pixel is 2 or more from that border);
//   phase B, sec < k <= K = max(maxlen, sec), threshold tau_low: first stop bit, else saved = K.
// A pixel flips the sticky threshold iff phase A with tau has no stop bit (it enters iteration sec+1).
// k_arm_cand writes both candidates (threshold still tau / already tau_low) and finds the first flipping
// pixel per direction; k_arm_pick selects with the same rule as k_arms.  Same index spaces as above
// (direction 1 with the stride bug runs over j < H and stores with stride H).
template <bool GRAY>
__global__ void __launch_bounds__(NT) k_arm_cand(const uint8_t *__restrict__ img, ArmCfg c, int *flip, int dir0,
                                                 const int *__restrict__ tau_state, uint16_t *__restrict__ cand)
{
    const int dir = dir0 + blockIdx.y;
    const int colR = (dir == 1 && !c.fix_right) ? c.H : c.W;
    const int idx = blockIdx.x * NT + threadIdx.x;
    const bool live = idx < c.H * colR;
    const int i = live ? idx / colR : 0, j = live ? idx - i * colR : 0;
    int kmax, step;
    bool far;
    switch (dir) {
    case 0: kmax = j; step = -1; far = (j - 1 >= 1); break;
    case 1: kmax = colR - 1 - j; step = 1; far = (j + 1 < colR - 1); break;
    case 2: kmax = i; step = -c.W; far = (i - 1 >= 1); break;
    default: kmax = c.H - 1 - i; step = c.W; far = (i + 1 < c.H - 1); break;
    }
    const int K = max(c.maxlen, c.sec);
    const int nk = min(kmax, K);
    const int base = i * c.W + j;
    // masks are built far-to-near, m = 2m + flag: one compare + one add-with-carry per threshold and
    // neighbour, as two 32-bit halves (bits 0..31 for k = 1..32, the upper word for k = 33..63).  A lane
    // whose arm cannot reach neighbour k (border) re-reads its own pixel instead: difference 0, no flag
    // (thresholds are >= 0 on this path).
    unsigned h_hi = 0, h_lo = 0, l_hi = 0, l_lo = 0;
    const int self = GRAY ? (int)img[base] : 0;
    auto sweep = [&](int kfrom, int kto, unsigned &m_hi, unsigned &m_lo) {     // k = kfrom down to kto
        constexpr int U = 8;                                                   // neighbours loaded per batch
        for (int k0 = kfrom; k0 >= kto; k0 -= U) {
            int d[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 - u;
                const unsigned nb = (k >= kto && k <= nk) ? (unsigned)(base + k * step) : (unsigned)base;
                d[u] = GRAY ? (int)__builtin_amdgcn_sad_u8((unsigned)self, (unsigned)img[nb], 0u) : pix_diff(img, c.ch, base, (int)nb);
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (k0 - u >= kto) {                                           // wave-uniform
                    asm("v_cmp_lt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m_hi) : "s"(c.tau), "v"(d[u]) : "vcc");
                    asm("v_cmp_lt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m_lo) : "s"(c.tau_low), "v"(d[u]) : "vcc");
                }
        }
    };
    if (K > 32) sweep(K, 33, h_hi, h_lo);
    sweep(min(K, 32), 1, l_hi, l_lo);
    const unsigned long long mhi = ((unsigned long long)h_hi << 32) | l_hi, mlo = ((unsigned long long)h_lo << 32) | l_lo;
    const unsigned long long oob = kmax < 64 ? 1ull << kmax : 0ull;      // neighbour kmax+1 is outside
    const unsigned long long lowmask = c.sec >= 64 ? ~0ull : (1ull << c.sec) - 1;
    const unsigned long long allmask = K >= 64 ? ~0ull : (1ull << K) - 1;
    auto walk = [&](unsigned long long mA) {
        unsigned long long st = (mA | oob) & lowmask;
        if (!st) st = (mlo | oob) & allmask & ~lowmask;
        if (!st) return K;
        int saved = __builtin_ctzll(st);
        if (saved != kmax && saved < 1 && far) saved = 1;                // stopped by a difference, not by the border
        return saved;
    };
    if (live) cand[(size_t)dir * c.H * c.W + idx] = (uint16_t)(walk(mhi) | (walk(mlo) << 8));
    // first pixel (row-major in this direction's index space) that passes sec neighbours under tau
    const bool qual = live && ((mhi | oob) & lowmask) == 0;
    if (*tau_state != c.tau) return;
    const unsigned long long b = __ballot(qual);
    if (b && (threadIdx.x & 63) == 0) {
        const int first = idx + __builtin_ctzll(b);
        if (first < *(volatile int *)&flip[dir]) atomicMin(&flip[dir], first);
    }
}

__global__ void __launch_bounds__(NT) k_arm_pick(ArmCfg c, const int *__restrict__ flip, const uint16_t *__restrict__ cand,
                                                 int *armL, int *armR, int *armT, int *armB, int dir0, int prev_flips,
                                                 const int *__restrict__ tau_state)
{
    const int dir = dir0 + blockIdx.y;
    const int colR = (dir == 1 && !c.fix_right) ? c.H : c.W;
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= c.H * colR) return;
    int tau_in = *tau_state;
    if (c.chain && prev_flips)
        for (int e = 0; e < dir; e++)
            if (flip[e] != INT_MAX) tau_in = c.tau_low;
    const unsigned v = cand[(size_t)dir * c.H * c.W + idx];
    const bool hi = (tau_in == c.tau) && idx <= flip[dir];
    int *out = dir == 0 ? armL : dir == 1 ? armR : dir == 2 ? armT : armB;
    out[idx] = hi ? (int)(v & 255u) : (int)(v >> 8);
}

// after the direction(s) dir0 .. dir0+ndir-1: a member threshold (chain) that flipped stays lowered
__global__ void k_tau_update(const int *__restrict__ flip, int dir0, int ndir, int chain, int tau_low, int *tau_state)
{
    if (!chain) return;
    for (int e = dir0; e < dir0 + ndir; e++)
        if (flip[e] != INT_MAX) *tau_state = tau_low;
}

template <int C> struct vecf { float v[C]; };          // C = 5..8 (D > 256)
template <> struct vecf<1> { float v[1]; };
template <> struct __attribute__((aligned(8))) vecf<2> { float v[2]; };
template <> struct vecf<3> { float v[3]; };
template <> struct __attribute__((aligned(16))) vecf<4> { float v[4]; };

// one wave per pixel; lane owns d = lane*C .. lane*C+C-1 (masked when >= D)
template <int C, int ORDER>
__global__ void __launch_bounds__(NT) k_aggregate(const float *__restrict__ vin, float *__restrict__ vout,
                                                  int H, int W, int D, const int *__restrict__ armL,
                                                  const int *__restrict__ armR, const int *__restrict__ armT,
                                                  const int *__restrict__ armB, float *__restrict__ disp,
                                                  int *ub_flag)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = H * W;
    const int p = blockIdx.x * (NT / 64) + wv;
    if (p >= N) return;
    const int i = p / W, j = p - i * W;
    const int Ll = armL[p], Rr = armR[p], up = armT[p], dn = armB[p];
    const int dl = lane * C;
    const bool full = (dl + C <= D);
    float acc[C];
#pragma unroll
    for (int k = 0; k < C; k++) acc[k] = 0.0f;
    bool ub = false;

    auto tap = [&](int t, int l) {
        const int idx = (i + t) * W + j + l;            // flat, wraps across row ends like the reference
        if (idx < 0 || idx >= N) { ub = true; return; } // reference reads outside the plane: UB -> 0
        const float *src = vin + (size_t)idx * D + dl;
        if (full) {
            const vecf<C> x = *reinterpret_cast<const vecf<C> *>(src);
#pragma unroll
            for (int k = 0; k < C; k++) acc[k] = acc[k] + x.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < C; k++)
                if (dl + k < D) acc[k] = acc[k] + src[k];
        }
    };
    if (ORDER == 0) {
        for (int l = -Ll; l <= Rr; l++)
            for (int t = -up; t <= dn; t++) tap(t, l);
    } else if (ORDER == 1) {
        for (int t = -up; t <= dn; t++)
            for (int l = -Ll; l <= Rr; l++) tap(t, l);
    } else {
        // CrossArmAggregation::Aggregation (CrossArm.cpp:104-145, no call site): rows outer, EXCLUSIVE
        // upper bounds (:130-132); a pixel with Ll+Rr == 0 or up+dn == 0 divides 0 by 0 (:138) -> NaN,
        // reported as reference-undefined
        for (int t = -up; t < dn; t++)
            for (int l = -Ll; l < Rr; l++) tap(t, l);
    }
    const float cnt = (ORDER == 2) ? (float)((Ll + Rr) * (up + dn)) : (float)((Ll + Rr + 1) * (up + dn + 1));
    if (ORDER == 2 && cnt == 0.0f) ub = true;
    float *dst = vout + (size_t)p * D + dl;
#pragma unroll
    for (int k = 0; k < C; k++) {
        acc[k] = acc[k] / cnt;
        if (dl + k < D) dst[k] = acc[k];
    }
    if (ub && lane == 0) atomicOr(ub_flag, 1);
    if (disp) {
        const int wd = wave_wta<C, false>(acc, dl, D);
        if (lane == 0) disp[p] = (float)wd;
    }
}

// ---- pipelined direct aggregation --------------------------------------------------------
// One wave per pixel as above, but the rectangle walk is wave-uniform (scalar counters), so
// the loads of the next U taps are issued before the U in-order adds: U*64*C*4 bytes in
// flight per wave instead of one tap.  Out-of-plane taps (reference UB) contribute nothing
// and raise the flag.
constexpr int AU = 8;

template <int C, int ORDER, bool FULL>
__global__ void __launch_bounds__(NT) k_aggregate_pipe(const float *__restrict__ vin, float *__restrict__ vout,
                                                       int H, int W, int D, const int *__restrict__ armL,
                                                       const int *__restrict__ armR, const int *__restrict__ armT,
                                                       const int *__restrict__ armB, float *__restrict__ disp,
                                                       int *ub_flag, int SW)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = H * W;
    // XCD-aware pixel order (speed only; any placement is correct): blocks b and b+8 share an XCD
    // (MI355X_MICROARCH.md), so XCD x = b%8 sweeps the column strips x, x+8, x+16, ... of width SW
    // row by row.  The pixels in flight on one XCD then cover a few rows of one narrow strip and
    // their rectangles' union stays inside that XCD's 4 MB L2.
    int p;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int gpr = SW / (NT / 64);                  // 4-pixel groups per strip row
        const int gps = gpr * H;                         // groups per strip
        const int strip = xcd + 8 * (slot / gps);
        const int g = slot % gps;
        const int row = g / gpr, col = strip * SW + (g % gpr) * (NT / 64) + wv;
        if (col >= W || strip * SW >= W) return;
        p = row * W + col;
    }
    const int Ll = __builtin_amdgcn_readfirstlane(armL[p]), Rr = __builtin_amdgcn_readfirstlane(armR[p]);
    const int up = __builtin_amdgcn_readfirstlane(armT[p]), dn = __builtin_amdgcn_readfirstlane(armB[p]);
    const int dl = lane * C;
    // inner/outer extents and flat strides of the walk
    const int nI = (ORDER == 0) ? (up + dn + 1) : (Ll + Rr + 1);
    const int nO = (ORDER == 0) ? (Ll + Rr + 1) : (up + dn + 1);
    const int sI = (ORDER == 0) ? W : 1, sO = (ORDER == 0) ? 1 : W;
    const int total = nI * nO;
    const int first = p - Ll - up * W;                   // top-left corner (first tap in both orders)
    const long lastp = (long)p + Rr + (long)dn * W;      // bottom-right corner (last tap)
    const bool ub = (first < 0) || (lastp >= N);         // flat indices are monotone in both axes
    float acc[C];
#pragma unroll
    for (int k = 0; k < C; k++) acc[k] = 0.0f;
    const char *base = (const char *)(vin + dl);         // + tap byte offset (< 2^32: V*4 bytes <= 1.6e9... checked by host)

    auto ld = [&](unsigned off, float (&x)[C]) {
        const float *src = (const float *)(base + off);
        if (FULL) {
            const vecf<C> v = *reinterpret_cast<const vecf<C> *>(src);
#pragma unroll
            for (int k = 0; k < C; k++) x[k] = v.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) x[k] = (dl + k < D) ? src[k] : 0.0f;
        }
    };

    if (!ub) {
        // fast path: every tap is inside the plane.  Lanes compute the byte offsets of the next 64
        // taps in parallel (one division per 64 taps); the tap loop is readlane + load + adds.
        const float rI = 1.0f / (float)nI;
        for (int n0 = 0; n0 < total; n0 += 64) {
            const int n = n0 + lane;
            int o = (int)((float)n * rI);                // n < 4761, nI <= 69: off by at most one, fixed below
            int t = n - o * nI;
            if (t < 0) { o--; t += nI; }
            if (t >= nI) { o++; t -= nI; }
            const unsigned offs = (unsigned)(first + o * sO + t * sI) * (unsigned)(D * 4);
            const int cnt = min(64, total - n0);         // scalar
            int u = 0;
            for (; u + AU <= cnt; u += AU) {
                float x[AU][C];
#pragma unroll
                for (int k = 0; k < AU; k++) ld((unsigned)__builtin_amdgcn_readlane((int)offs, u + k), x[k]);
#pragma unroll
                for (int k = 0; k < AU; k++) {
#pragma unroll
                    for (int c = 0; c < C; c++) acc[c] = acc[c] + x[k][c];
                }
            }
            for (; u < cnt; u++) {
                float x[C];
                ld((unsigned)__builtin_amdgcn_readlane((int)offs, u), x);
#pragma unroll
                for (int c = 0; c < C; c++) acc[c] = acc[c] + x[c];
            }
        }
    } else {
        // reference UB (rectangle leaves the plane): out-of-plane taps contribute nothing
        int idx = first, inner = 0;
        const int wrap = sO - nI * sI;
        for (int n = 0; n < total; n++) {
            if (idx >= 0 && idx < N) {
                float x[C];
                ld((unsigned)idx * (unsigned)(D * 4), x);
#pragma unroll
                for (int c = 0; c < C; c++) acc[c] = acc[c] + x[c];
            }
            inner++; idx += sI;
            if (inner == nI) { inner = 0; idx += wrap; }
        }
    }
    const float cnt = (float)total;
    float *dst = vout + (size_t)p * D + dl;
#pragma unroll
    for (int k = 0; k < C; k++) {
        acc[k] = acc[k] / cnt;
        if (FULL || dl + k < D) dst[k] = acc[k];
    }
    if (ub && lane == 0) atomicOr(ub_flag, 1);
    if (disp) {
        const int wd = wave_wta<C, FULL>(acc, dl, D);
        if (lane == 0) disp[p] = (float)wd;
    }
}

// ---- register-level tap sharing: QP adjacent pixels per wave ----------------------------------
// The wave walks the bounding box of its QP rectangles in the reference's outer/inner order, 64 box
// positions at a time: every lane classifies one position (which of the QP rectangles contain it)
// and computes its byte offset; a ballot keeps only positions that belong to at least one rectangle,
// so each tap of the UNION is loaded exactly once and added -- in order -- to every pixel whose
// rectangle contains it (wave-uniform branches).  For neighbouring pixels union/sum of areas is
// ~0.5 (4 pixels), which is the saving on the texture path that bounds this stage.
template <int C, int ORDER, bool FULL, int QP>
__global__ void __launch_bounds__(NT) k_aggregate_quad(const float *__restrict__ vin, float *__restrict__ vout,
                                                       int H, int W, int D, const int *__restrict__ armL,
                                                       const int *__restrict__ armR, const int *__restrict__ armT,
                                                       const int *__restrict__ armB, float *__restrict__ disp,
                                                       int *ub_flag, int SW)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = H * W;
    int p0, nlive;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int gpr = SW / (QP * (NT / 64));           // blocks per strip row
        const int gps = gpr * H;
        const int strip = xcd + 8 * (slot / gps);
        const int g = slot % gps;
        const int row = g / gpr, col = strip * SW + ((g % gpr) * (NT / 64) + wv) * QP;
        if (col >= W || strip * SW >= W) return;
        p0 = row * W + col;
        nlive = min(QP, W - col);
    }
    int oa[QP], ob[QP], ia[QP], ib[QP], cnt[QP];
    int omin = INT_MAX, omax = INT_MIN, imin = INT_MAX, imax = INT_MIN;
    bool ub = false;
#pragma unroll
    for (int q = 0; q < QP; q++) {
        oa[q] = 1; ob[q] = 0; ia[q] = 1; ib[q] = 0; cnt[q] = 1;
        if (q < nlive) {
            const int p = p0 + q;
            const int Ll = __builtin_amdgcn_readfirstlane(armL[p]), Rr = __builtin_amdgcn_readfirstlane(armR[p]);
            const int up = __builtin_amdgcn_readfirstlane(armT[p]), dn = __builtin_amdgcn_readfirstlane(armB[p]);
            cnt[q] = (Ll + Rr + 1) * (up + dn + 1);
            if (ORDER == 0) { oa[q] = q - Ll; ob[q] = q + Rr; ia[q] = -up; ib[q] = dn; }
            else            { oa[q] = -up;    ob[q] = dn;     ia[q] = q - Ll; ib[q] = q + Rr; }
            omin = min(omin, oa[q]); omax = max(omax, ob[q]);
            imin = min(imin, ia[q]); imax = max(imax, ib[q]);
            ub = ub || (p - Ll - up * W < 0) || ((long)p + Rr + (long)dn * W >= N);
        }
    }
    const int so = (ORDER == 0) ? 1 : W, si = (ORDER == 0) ? W : 1;
    const int dl = lane * C;
    float acc[QP][C];
#pragma unroll
    for (int q = 0; q < QP; q++)
#pragma unroll
        for (int k = 0; k < C; k++) acc[q][k] = 0.0f;
    const char *base = (const char *)(vin + dl);

    auto ld = [&](unsigned off, float (&x)[C]) {
        const float *src = (const float *)(base + off);
        if (FULL) {
            const vecf<C> v = *reinterpret_cast<const vecf<C> *>(src);
#pragma unroll
            for (int k = 0; k < C; k++) x[k] = v.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) x[k] = (dl + k < D) ? src[k] : 0.0f;
        }
    };

    // acc[q] += x for every pixel q in the wave-uniform membership mask m: one 16-way switch instead
    // of QP bit tests and branches per tap
    auto add_bits = [&](auto mtag, auto basetag, const float (&x)[C]) {
        constexpr unsigned M = decltype(mtag)::value;
        constexpr int B = decltype(basetag)::value;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if ((M & (1u << q)) && B + q < QP) {
#pragma unroll
                for (int c = 0; c < C; c++) acc[B + q < QP ? B + q : 0][c] = acc[B + q < QP ? B + q : 0][c] + x[c];
            }
    };
    auto add_nibble = [&](unsigned m, auto basetag, const float (&x)[C]) {
        switch (m) {
#define SMT_CASE(V) case V: add_bits(std::integral_constant<unsigned, V>{}, basetag, x); break;
            SMT_CASE(1) SMT_CASE(2) SMT_CASE(3) SMT_CASE(4) SMT_CASE(5) SMT_CASE(6) SMT_CASE(7) SMT_CASE(8)
            SMT_CASE(9) SMT_CASE(10) SMT_CASE(11) SMT_CASE(12) SMT_CASE(13) SMT_CASE(14) SMT_CASE(15)
#undef SMT_CASE
        default: break;
        }
    };
    auto add_masked = [&](unsigned m, const float (&x)[C]) {
        add_nibble(m & 15u, std::integral_constant<int, 0>{}, x);
        if (QP > 4) add_nibble(m >> 4, std::integral_constant<int, 4>{}, x);
    };

    if (!ub) {
        const int nIb = imax - imin + 1;
        const int total = (omax - omin + 1) * nIb;
        const float rI = 1.0f / (float)nIb;
        for (int n0 = 0; n0 < total; n0 += 64) {
            const int n = n0 + lane;
            int o = (int)((float)n * rI);
            int t = n - o * nIb;
            if (t < 0) { o--; t += nIb; }
            if (t >= nIb) { o++; t -= nIb; }
            o += omin; t += imin;
            unsigned mask = 0;
            if (n < total) {
#pragma unroll
                for (int q = 0; q < QP; q++)
                    mask |= (unsigned)(o >= oa[q] && o <= ob[q] && t >= ia[q] && t <= ib[q]) << q;
            }
            const unsigned offs = (unsigned)(p0 + o * so + t * si) * (unsigned)(D * 4);
            unsigned long long live = __ballot(mask != 0);
            // full groups of AU union taps: indices, then all loads, then the in-order adds
            while (__builtin_popcountll(live) >= AU) {
                int u[AU];
                float x[AU][C];
#pragma unroll
                for (int k = 0; k < AU; k++) { u[k] = __builtin_ctzll(live); live &= live - 1; }
#pragma unroll
                for (int k = 0; k < AU; k++) ld((unsigned)__builtin_amdgcn_readlane((int)offs, u[k]), x[k]);
#pragma unroll
                for (int k = 0; k < AU; k++) add_masked((unsigned)__builtin_amdgcn_readlane((int)mask, u[k]), x[k]);
            }
            while (live) {
                const int u = __builtin_ctzll(live);
                live &= live - 1;
                float x[C];
                ld((unsigned)__builtin_amdgcn_readlane((int)offs, u), x);
                add_masked((unsigned)__builtin_amdgcn_readlane((int)mask, u), x);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < QP; q++) {
            if (q >= nlive) continue;
            const int nO = ob[q] - oa[q] + 1, nI = ib[q] - ia[q] + 1;
            for (int o = 0; o < nO; o++)
                for (int n = 0; n < nI; n++) {
                    const long idx = (long)p0 + (long)(oa[q] + o) * so + (long)(ia[q] + n) * si;
                    if (idx >= 0 && idx < N) {
                        float x[C];
                        ld((unsigned)idx * (unsigned)(D * 4), x);
#pragma unroll
                        for (int c = 0; c < C; c++) acc[q][c] = acc[q][c] + x[c];
                    }
                }
        }
        if (lane == 0) atomicOr(ub_flag, 1);
    }

#pragma unroll
    for (int q = 0; q < QP; q++) {
        if (q >= nlive) continue;
        const float fc = (float)cnt[q];
        float *dst = vout + (size_t)(p0 + q) * D + dl;
        float mean[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
            mean[k] = acc[q][k] / fc;
            if (FULL || dl + k < D) dst[k] = mean[k];
        }
        if (disp) {
            const int wd = wave_wta<C, FULL>(mean, dl, D);
            if (lane == 0) disp[p0 + q] = (float)wd;
        }
    }
}

// ---- 8 or 16 pixels per wave, branch-free membership -----------------------------------------------
// Same union walk as k_aggregate_quad with twice the sharing (union/sum of areas ~0.27 for 8 adjacent
// pixels vs ~0.40 for 4).  The per-tap switch is what made 8 pixels slower there; here every union
// tap is added to ALL 8 accumulators as fma(x, f, acc) with f = 1.0f for member pixels and 0.0f for
// the others:  fma(x, 1, acc) is the reference's acc + x (one rounding of the exact sum) and
// fma(x, 0, acc) == acc bit for bit for every finite x (acc starts at +0 and a sum that began at +0
// is never -0, so adding +-0 leaves it unchanged).  The flags of a tap are one row of a 256 x 8
// table in constant memory, fetched with the wave-uniform membership mask as one s_load_dwordx8, and
// two pixels share each v_pk_fma_f32: 12 packed FMAs per tap for D = 192, below the texture path's
// ~16 cycles per tap.  A non-finite x would turn fma(x, 0, acc) into NaN for non-members; every
// pixel whose result holds a NaN is therefore recomputed with the plain in-order walk, which is the
// reference's answer in every case (including a genuine NaN).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef int i3v __attribute__((ext_vector_type(3)));
typedef int i4v __attribute__((ext_vector_type(4)));
constexpr int MEMBER_TAB_FLOATS = 256 * 8;   // row m: 8 floats, 1.0f where bit q of m is set

// Buffer loads (SGPR resource + VGPR lane offset + SGPR tap offset).  Elements are converted with
// __int_as_float: __builtin_bit_cast(float, v.y) on a vector element reads element 0 with this clang.
__device__ __forceinline__ int buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0);
}
__device__ __forceinline__ i2v buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
{
    return __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0);
}
__device__ __forceinline__ i3v buf_ld3(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
{
    return __builtin_amdgcn_raw_buffer_load_b96(r, vo, so, 0);
}
__device__ __forceinline__ i4v buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
}

// One wave owns a block of QR rows x 8 columns of pixels (QR = 1: 8 pixels, QR = 2: 16 pixels; pixel
// q = r*8 + c).  Lane q keeps pixel q's rectangle (box coordinates relative to the block's top-left
// pixel, 16-bit fields); the classification loop broadcasts them with v_readlane once per batch of 64
// box positions, so no rectangle lives in SGPRs across the tap loop.
template <int C, int ORDER, bool FULL, int QR, int SKIP, int TWS = 3>
__global__ void __launch_bounds__(NT, (SKIP == 6 ? (C <= 2 ? 5 : C == 3 ? 4 : 3) : (SKIP == 4 || SKIP == 5) ? (C <= 3 ? 5 : 3) : (C <= 3 ? 6 : 4))) k_aggregate_multi(const float *__restrict__ vin, float *__restrict__ vout,
                                                        int H, int W, int D, const int *__restrict__ armL,
                                                        const int *__restrict__ armR, const int *__restrict__ armT,
                                                        const int *__restrict__ armB, float *__restrict__ disp,
                                                        int *ub_flag, int SW, const float *__restrict__ member, int sweep)
{
    // tile geometry: pixel q of the wave sits at row q >> TWS, column q & (QC - 1) of a TH x QC tile.  TWS = 3: QR x 8
    // (q's bit position in the membership mask and its place in the tile coincide); TWS = 2 with QR = 2: 4 x 4 pixels --
    // the most compact 16-pixel tile, 7-8 % fewer union taps on the benchmark pair (233 against 252 per tile) -- with
    // mask bits, flag rows and skip groups unchanged (a group of four = one row of the tile).
    constexpr int NPIX = 8 * QR, QC = 1 << TWS, TH = NPIX >> TWS;
    constexpr int G = (QR == 1) ? 8 : 4;                 // union taps loaded per group
    constexpr int BIAS = 16384;
    // SKIP == 3 ("lock-step"): SKIP == 2 plus one s_barrier per batch of 64 box positions, with all four waves
    // of the workgroup enumerating the positions of their COMMON bounding box.  The waves own neighbouring
    // tiles whose unions overlap; an XCD's 4 MB L2 is refilled every ~5 us at this kernel's fetch rate, so a
    // line fetched for one tile is gone before a free-running neighbour asks for it (TCC: ~10 fabric fetches
    // per input line).  Walking the same columns at the same time turns those re-fetches into L2 hits.
    // SKIP == 4 / 5 ("matrix"): the lock-step kernel with the flagged accumulate on the matrix pipe.  For a group of
    // four pixels v_mfma_f32_4x4x1_16b_f32 computes, in every 4-lane block, D[i][j] = A[i] * B[j] + C[i][j]: with A =
    // the four membership flags (lane l carries the flag of pixel l % 4 of the group) and B = the tap's row (lane l
    // carries its hypotheses as before), accumulator register i of lane l becomes acc_i[l] + flag_i * x[l] -- the same
    // one-rounding fma per element as v_pk_fma_f32 (tools/mfma_flag_probe.hip: bit-identical to the v_fma chain and
    // to plain adds on 4 096 taps of normal, denormal, negative and huge values), in the accumulator layout the
    // kernel already has, at the same 256 FMA per 8 cycles per SIMD -- on the other pipe, with no flag rows to fetch
    // through the scalar cache (the flags are two VALU operations on the membership mask).  4 = every group, no branch
    // in the tap loop; 5 = live groups only.
    // SKIP == 6 ("four taps per instruction", free-running): v_mfma_f32_16x16x4_f32 takes A = 16 pixels x 4 taps of
    // membership flags and B = 4 taps x 16 hypotheses and accumulates k = 0..3 in order as an exact fma chain, i.e. four
    // in-order flagged adds for 16 pixels x 16 hypotheses per instruction.  Per group of four union taps: C 16-byte gather
    // loads (lane l reads hypotheses 4n .. 4n + 3 of chunk j of tap l >> 4, n = l & 15), one A register built from the
    // four membership masks, 4 * C matrix instructions, two groups in flight.  The accumulators come out in the
    // instruction's layout (register v of lane l = pixel 4 (l >> 4) + v, hypothesis 64 j + 4 n + r for accumulator
    // (j, r)), so this form has its own mean / store / WTA epilogue; D must be a multiple of 64.
    // All three are measured equal to or slower than SKIP == 3: kept as independent formulations.
    // SKIP == 7 ("scalar word"): SKIP == 3 with the scalar side of a tap prepared by the vector classification of its
    // batch: every position's lane packs, next to the membership mask, one word holding the byte offsets of the tap's
    // two flag rows and one "group has a member" bit per group of four pixels.  The tap loop then reads that word with
    // one v_readlane and spends s_and + s_lshr on the two row addresses and s_bitcmp1 + branch per group, instead of
    // shift / mask / bit-field extract per row address and s_and + s_cmp + branch per group.
    constexpr bool MM = (SKIP == 6);
    constexpr bool MFMA = (SKIP == 4 || SKIP == 5);
    constexpr bool MSKIP = (SKIP == 5);
    constexpr bool SWORD = (SKIP == 7);
    static_assert(SKIP <= 7 && (!MM || (QR == 2 && FULL)) && (!SWORD || QR == 2), "");
    constexpr bool SYNC = (SKIP >= 3 && SKIP <= 5) || SWORD;   // 3 .. 5 and 7 walk in lock-step
    constexpr bool PREF = (SKIP >= 2);                   // flag rows fetched one tap ahead + per-axis tables
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = H * W;
    int p0, ncol, nrow;
    int ooff = 0, ioff = 0;                              // this wave's tile origin inside the workgroup's block
    {
        // XCD x = blockIdx%8 owns the column strips x, x+8, ... of width SW; a workgroup covers
        // wx*8 columns by wy*QR rows (wx*wy = 4 waves), so narrow strips stack the waves vertically.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int wx = min(NT / 64, SW / QC), wy = (NT / 64) / wx;
        const int gpr = SW / (QC * wx);
        const int nband = (H + wy * TH - 1) / (wy * TH);
        int row, col;
        if (sweep == 0) {
            // strips interleaved over the XCDs: XCD x sweeps strips x, x+8, ... top to bottom
            const int gps = gpr * nband;
            const int strip = xcd + 8 * (slot / gps);
            const int g = slot % gps;
            row = ((g / gpr) * wy + wv / wx) * TH;
            col = strip * SW + ((g % gpr) * wx + (wv % wx)) * QC;
        } else {
            // XCD x owns the x-th contiguous band of rows and sweeps it strip by strip: the column halo
            // between neighbouring strips stays inside one XCD's L2 instead of being fetched by two
            const int nbx = (nband + 7) >> 3;              // workgroup rows per XCD
            const int gps = gpr * nbx;
            const int strip = slot / gps;
            const int g = slot % gps;
            row = ((xcd * nbx + g / gpr) * wy + wv / wx) * TH;
            col = strip * SW + ((g % gpr) * wx + (wv % wx)) * QC;
        }
        if (SYNC) {
            const int roff = (wv / wx) * TH, coff = (wv % wx) * QC;
            ooff = (ORDER == 0) ? coff : roff;
            ioff = (ORDER == 0) ? roff : coff;
        }
        if (row >= H || col >= W) {
            if (!SYNC) return;
            row = 0; col = 0;                              // stays for the barriers; owns no pixel
            ncol = 0; nrow = 0;
        } else {
            ncol = min(QC, W - col);
            nrow = min(TH, H - row);
        }
        p0 = row * W + col;
    }
    const int so = (ORDER == 0) ? 1 : W, si = (ORDER == 0) ? W : 1;
    // lane q: rectangle of pixel q as outer [oa, ob] x inner [ia, ib]; empty for pixels off the image
    int my_oa = 1, my_ob = 0, my_ia = 1, my_ib = 0;
    bool my_ub = false;
    {
        const int r = lane >> TWS, c = lane & (QC - 1);
        if (lane < NPIX && r < nrow && c < ncol) {
            const int p = p0 + r * W + c;
            const int Ll = armL[p], Rr = armR[p], up = armT[p], dn = armB[p];
            if (ORDER == 0) { my_oa = c - Ll; my_ob = c + Rr; my_ia = r - up; my_ib = r + dn; }
            else            { my_oa = r - up; my_ob = r + dn; my_ia = c - Ll; my_ib = c + Rr; }
            my_ub = (p - Ll - up * W < 0) || ((long)p + Rr + (long)dn * W >= N);
        }
    }
    const bool ub = __ballot(my_ub) != 0;
    int omin = INT_MAX, omax = INT_MIN, imin = INT_MAX, imax = INT_MIN;
#pragma unroll
    for (int q = 0; q < NPIX; q++) {
        if ((q >> TWS) < nrow && (q & (QC - 1)) < ncol) {
            omin = min(omin, __builtin_amdgcn_readlane(my_oa, q)); omax = max(omax, __builtin_amdgcn_readlane(my_ob, q));
            imin = min(imin, __builtin_amdgcn_readlane(my_ia, q)); imax = max(imax, __builtin_amdgcn_readlane(my_ib, q));
        }
    }
    unsigned pk_o = (unsigned)(my_oa + BIAS) | ((unsigned)(my_ob + BIAS) << 16);
    unsigned pk_i = (unsigned)(my_ia + BIAS) | ((unsigned)(my_ib + BIAS) << 16);

    const int dl = lane * C;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 acc[MFMA ? 1 : NPIX / 2][C];
    f4 macc[MFMA ? NPIX / 4 : 1][C];                       // matrix formulation: group g = pixels 4g .. 4g + 3, register i = pixel 4g + i
#pragma unroll
    for (int j = 0; j < (MFMA ? 1 : NPIX / 2); j++)
#pragma unroll
        for (int k = 0; k < C; k++) acc[j][k] = f2{0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < (MFMA ? NPIX / 4 : 1); j++)
#pragma unroll
        for (int k = 0; k < C; k++) macc[j][k] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    f4 mm[MM ? 4 * C : 1];                                 // accumulator (j, r) = mm[4 j + r]
#pragma unroll
    for (int j = 0; j < (MM ? 4 * C : 1); j++) mm[j] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const int mm_k = lane >> 4, mm_n = lane & 15;          // tap of the group / hypothesis quad this lane feeds
    const int mm_shift = mm_n + 16 * (mm_k & 1);           // its flag bit in the packed mask pair
    // bit of the membership mask that lane l turns into its A operand for group g: 4g + (l & 3)
    int mshift[MFMA ? NPIX / 4 : 1];
#pragma unroll
    for (int g = 0; g < (MFMA ? NPIX / 4 : 1); g++) mshift[g] = 4 * g + (lane & 3);
    auto mfma_flagged = [&](unsigned m, const float (&x)[C]) {
        if constexpr (!MFMA) return;
#pragma unroll
        for (int g = 0; g < NPIX / 4; g++) {
            auto body = [&]() {
                // 0.0f or 1.0f: sign-extended 1-bit field (0 / -1) ANDed with the bits of 1.0f
                const float A = __int_as_float(__builtin_amdgcn_sbfe(m, mshift[MFMA ? g : 0], 1) & 0x3f800000);
#pragma unroll
                for (int c = 0; c < C; c++)
                    macc[MFMA ? g : 0][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, x[c], macc[MFMA ? g : 0][c], 0, 0, 0);
            };
            if constexpr (!MSKIP) body();
            else if ((m >> (4 * g)) & 15u) body();
        }
    };
    const unsigned lane_off = (unsigned)dl * 4u;
    // taps are fetched as buffer loads: wave-uniform byte offset of the tap in an SGPR, the lane's
    // disparity offset in a VGPR, no per-tap address arithmetic (volume < 4 GiB, checked by the host)
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)vin, 0, (int)((unsigned)N * (unsigned)(D * 4)), 0x00020000);
    auto ld = [&](unsigned off, float (&x)[C]) {
#if defined(SMT_AGG_KNOCKOUT) && SMT_AGG_KNOCKOUT == 2
        // diagnostic build: no tap is loaded (values made from the offset): the issue side of the kernel alone
#pragma unroll
        for (int k = 0; k < C; k++) x[k] = __int_as_float((int)(off >> 8) + k + (int)lane_off);
        return;
#endif
        if (FULL) {
            if (C == 1) x[0] = __int_as_float(buf_ld1(rsrc, lane_off, off));
            else if (C == 2) {
                const i2v v = buf_ld2(rsrc, lane_off, off);
                x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y);
            } else if (C == 3) {
                const i3v v = buf_ld3(rsrc, lane_off, off);
                x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y);
                x[C > 2 ? 2 : 0] = __int_as_float(v.z);
            } else {
                const i4v v = buf_ld4(rsrc, lane_off, off);
                x[0] = __int_as_float(v.x); x[C > 1 ? 1 : 0] = __int_as_float(v.y);
                x[C > 2 ? 2 : 0] = __int_as_float(v.z); x[C > 3 ? 3 : 0] = __int_as_float(v.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; k++)
                x[k] = (dl + k < D) ? __int_as_float(buf_ld1(rsrc, lane_off + 4u * k, off)) : 0.0f;
        }
    };
    // acc[q] = fma(x, member(q) ? 1 : 0, acc[q]), two pixels per v_pk_fma_f32.  SKIP: a group of 4
    // pixels none of which holds the tap is skipped by a wave-uniform branch (2.4 of 4 groups are live on
    // average with 2x8 pixels; testing pairs instead costs more scalar work than it saves).
    auto add_flagged = [&](unsigned m, const float (&x)[C]) {
#pragma unroll
        for (int h = 0; h < QR; h++) {
            const unsigned mb = (m >> (8 * h)) & 255u;
            const float *f = member + mb * 8u;
            auto pair = [&](int j) {
                const f2 fl = f2{f[2 * j], f[2 * j + 1]};
#pragma unroll
                for (int c = 0; c < C; c++)
                    acc[MFMA ? 0 : 4 * h + j][c] = __builtin_elementwise_fma(f2{x[c], x[c]}, fl, acc[MFMA ? 0 : 4 * h + j][c]);
            };
            if (SKIP == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) pair(j);
            } else {
                if (mb & 0x0fu) { pair(0); pair(1); }
                if (mb & 0xf0u) { pair(2); pair(3); }
            }
        }
    };

    // SKIP == 2: the flag rows of a tap (QR x 8 floats, one s_load_dwordx8 per row of pixels) are fetched
    // one tap AHEAD of the FMAs that use them, into the other half of a two-deep SGPR buffer, so the scalar
    // cache latency hides behind the previous tap's FMAs instead of sitting in front of every live group
    // (SKIP == 1 loads four flags per live group and waits for them on the spot).
    typedef float f8 __attribute__((ext_vector_type(8)));
    auto load_flags = [&](unsigned m, f8 (&F)[QR]) {
        if constexpr (SWORD) {
            // m = scalar word: bits 5..12 = 32 * (mask & 255), bits 21..28 = 32 * (mask >> 8): byte offsets of the rows
            const char *mb8 = reinterpret_cast<const char *>(member);
            F[0] = *reinterpret_cast<const f8 *>(mb8 + (m & 0x1fe0u));
            F[QR - 1] = *reinterpret_cast<const f8 *>(mb8 + (m >> 16));
            return;
        }
#pragma unroll
        for (int h = 0; h < QR; h++) F[h] = *reinterpret_cast<const f8 *>(member + ((m >> (8 * h)) & 255u) * 8u);
    };
    auto fma_flagged = [&](unsigned m, const f8 (&F)[QR], const float (&x)[C]) {
#if defined(SMT_AGG_KNOCKOUT) && SMT_AGG_KNOCKOUT == 1
        // diagnostic build: every tap is still loaded and waited for, its flags still fetched, one packed FMA per
        // tap keeps them live -- what is left is the memory side + the scalar bookkeeping of the kernel
#pragma unroll
        for (int c = 0; c < C; c++)
            acc[0][c] = __builtin_elementwise_fma(f2{x[c], x[c]}, f2{F[0][0], F[QR - 1][1]}, acc[0][c]);
        return;
#endif
#pragma unroll
        for (int h = 0; h < QR; h++) {
            const unsigned mb = (m >> (8 * h)) & 255u;
            auto pair = [&](int j) {
                const f2 fl = f2{F[h][2 * j], F[h][2 * j + 1]};
#pragma unroll
                for (int c = 0; c < C; c++)
                    acc[MFMA ? 0 : 4 * h + j][c] = __builtin_elementwise_fma(f2{x[c], x[c]}, fl, acc[MFMA ? 0 : 4 * h + j][c]);
            };
            if constexpr (SWORD) {
                // bits 0..3 of the scalar word: group g = pixels 4g .. 4g + 3 has a member (one s_bitcmp1 each)
                if (__builtin_expect((m & (1u << (2 * h))) != 0, 1)) { pair(0); pair(1); }   // live bodies stay in line
                if (__builtin_expect((m & (2u << (2 * h))) != 0, 1)) { pair(2); pair(3); }
            } else {
                if (mb & 0x0fu) { pair(0); pair(1); }
                if (mb & 0xf0u) { pair(2); pair(3); }
            }
        }
    };

    // box whose positions are enumerated: the wave's own bounding box, or (SYNC) the common one of the four
    // waves in workgroup coordinates (wave-relative position = workgroup-relative - (ooff, ioff))
    int bo0 = omin, bo1 = omax, bi0 = imin, bi1 = imax;
    if (SYNC) {
        __shared__ int s_bb[NT / 64][4];
        if (lane == 0) {
            const bool any = omax >= omin;
            s_bb[wv][0] = any ? omin + ooff : INT_MAX; s_bb[wv][1] = any ? omax + ooff : INT_MIN;
            s_bb[wv][2] = any ? imin + ioff : INT_MAX; s_bb[wv][3] = any ? imax + ioff : INT_MIN;
        }
        __syncthreads();
        bo0 = INT_MAX; bo1 = INT_MIN; bi0 = INT_MAX; bi1 = INT_MIN;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) {
            bo0 = min(bo0, s_bb[w][0]); bo1 = max(bo1, s_bb[w][1]);
            bi0 = min(bi0, s_bb[w][2]); bi1 = max(bi1, s_bb[w][3]);
        }
        if (bo1 >= bo0) { bo0 -= ooff; bo1 -= ooff; bi0 -= ioff; bi1 -= ioff; }   // back to this wave's coordinates
        if (ub && lane == 0) atomicOr(ub_flag, 1);
    }
    if (SYNC || !ub) {
        const int nIb = (bo1 >= bo0) ? bi1 - bi0 + 1 : 1;
        const int total = (bo1 >= bo0) ? (bo1 - bo0 + 1) * nIb : 0;
        const float rI = 1.0f / (float)nIb;
        // per-axis membership tables (SKIP == 2 only; bounding boxes wider or taller than 64 -- arms of 28 and
        // more on both sides -- classify pixel by pixel as before): lane l holds, as one bit per pixel of the
        // tile, which pixels' outer ranges contain omin + l (tab_o) and which inner ranges contain imin + l (tab_i)
        const bool axis_tables = PREF && (omax >= omin) && (omax - omin + 1 <= 64) && (imax - imin + 1 <= 64);
        unsigned tab_o = 0, tab_i = 0;
        if (axis_tables) {
            const unsigned ob16 = (unsigned)(lane + omin + BIAS), tb16 = (unsigned)(lane + imin + BIAS);
#pragma unroll
            for (int q = 0; q < NPIX; q++) {
                const unsigned bo = (unsigned)__builtin_amdgcn_readlane((int)pk_o, q);
                const unsigned bi = (unsigned)__builtin_amdgcn_readlane((int)pk_i, q);
                tab_o |= (unsigned)(ob16 >= (bo & 0xffffu) && ob16 <= (bo >> 16)) << q;
                tab_i |= (unsigned)(tb16 >= (bi & 0xffffu) && tb16 <= (bi >> 16)) << q;
            }
        }
        for (int n0 = 0; n0 < total; n0 += 64) {
            const int n = n0 + lane;
            int o = (int)((float)n * rI);
            int t = n - o * nIb;
            if (t < 0) { o--; t += nIb; }
            if (t >= nIb) { o++; t -= nIb; }
            o += bo0; t += bi0;
            unsigned mask = 0;
            if (axis_tables) {
                // membership of position (o, t) = (pixels whose outer range holds o) & (pixels whose inner range
                // holds t): two per-axis bit tables built once per tile, fetched from the lane that owns the index
                const unsigned mo = (unsigned)__builtin_amdgcn_ds_bpermute((o - omin) << 2, (int)tab_o);
                const unsigned mi = (unsigned)__builtin_amdgcn_ds_bpermute((t - imin) << 2, (int)tab_i);
                const bool own = !SYNC || (o >= omin && o <= omax && t >= imin && t <= imax);   // inside this wave's own box
                mask = (n < total && own) ? (mo & mi) : 0u;
            } else {
                asm volatile("" : "+v"(pk_o), "+v"(pk_i));   // keep the broadcasts below inside the loop
                const unsigned ob16 = (unsigned)(o + BIAS), tb16 = (unsigned)(t + BIAS);
#pragma unroll
                for (int q = 0; q < NPIX; q++) {
                    const unsigned bo = (unsigned)__builtin_amdgcn_readlane((int)pk_o, q);
                    const unsigned bi = (unsigned)__builtin_amdgcn_readlane((int)pk_i, q);
                    const bool in = ob16 >= (bo & 0xffffu) && ob16 <= (bo >> 16) && tb16 >= (bi & 0xffffu) && tb16 <= (bi >> 16);
                    mask |= (unsigned)in << q;
                }
                if (n >= total) mask = 0;
            }
            if (SYNC && ub) mask = 0;                    // reference-undefined tile: every pixel goes to the plain walk below
            unsigned sword = 0;
            if constexpr (SWORD)
                sword = (unsigned)((mask & 0x000fu) != 0) | ((unsigned)((mask & 0x00f0u) != 0) << 1) |
                        ((unsigned)((mask & 0x0f00u) != 0) << 2) | ((unsigned)((mask & 0xf000u) != 0) << 3) |
                        ((mask & 255u) << 5) | ((mask >> 8) << 21);
            const unsigned offs = (unsigned)(p0 + o * so + t * si) * (unsigned)(D * 4);
            unsigned long long live = __ballot(mask != 0);
            // groups of union taps: indices, then all loads, then the in-order adds.  A short last
            // group is padded with repeats of its last tap under an all-zero membership mask.
            auto next = [&]() {
                const int u = __builtin_ctzll(live);
                asm("s_bitset0_b64 %0, %1" : "+s"(live) : "s"(u));
                return u;
            };
            auto group = [&](auto gtag, auto padtag) {
                constexpr int GG = decltype(gtag)::value;
                constexpr bool PAD = decltype(padtag)::value;
                int u[GG];
                unsigned m[GG];
                float x[GG][C];
                int last = 0;
#pragma unroll
                for (int k = 0; k < GG; k++) {
                    const bool valid = !PAD || live != 0;
                    if (valid) last = next();
                    u[k] = last;
                    m[k] = valid ? (unsigned)__builtin_amdgcn_readlane((int)(SWORD ? sword : mask), last) : 0u;
                }
#pragma unroll
                for (int k = 0; k < GG; k++) ld((unsigned)__builtin_amdgcn_readlane((int)offs, u[k]), x[k]);
                if constexpr (MFMA) {
                    __builtin_amdgcn_sched_barrier(0);               // every load of the group is issued before its first use
#pragma unroll
                    for (int k = 0; k < GG; k++) mfma_flagged(m[k], x[k]);
                } else if constexpr (PREF) {
                    f8 F[2][QR];
                    load_flags(m[0], F[0]);
#pragma unroll
                    for (int k = 0; k < GG; k++) {
                        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this tap's flags are in
                        __builtin_amdgcn_sched_barrier(0);
                        if (k + 1 < GG) load_flags(m[k + 1], F[(k + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        fma_flagged(m[k], F[k & 1], x[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < GG; k++) add_flagged(m[k], x[k]);
                }
            };
            int cnt = __builtin_popcountll(live);
            if constexpr (MM) {
                // groups of four union taps (the last one padded with repeats of its last tap under an all-zero mask),
                // two in flight: the gather loads of group g + 1 are issued before the matrix instructions of group g
                auto prep = [&](i4v (&xb)[C], float &A) {
                    unsigned o[4], m[4];
                    int last = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const bool valid = live != 0;
                        if (valid) last = next();
                        o[k] = (unsigned)__builtin_amdgcn_readlane((int)offs, last);
                        m[k] = valid ? (unsigned)__builtin_amdgcn_readlane((int)mask, last) : 0u;
                    }
                    // lane l gathers from tap l >> 4: its 16 bytes of each 256-byte chunk of that tap's row
                    unsigned vo = o[0];
                    vo = mm_k >= 1 ? o[1] : vo;
                    vo = mm_k >= 2 ? o[2] : vo;
                    vo = mm_k >= 3 ? o[3] : vo;
                    vo += (unsigned)mm_n * 16u;
#pragma unroll
                    for (int j = 0; j < C; j++) xb[j] = buf_ld4(rsrc, vo, 256u * j);
                    const unsigned m01 = m[0] | (m[1] << 16), m23 = m[2] | (m[3] << 16);
                    A = __int_as_float(__builtin_amdgcn_sbfe((mm_k < 2) ? m01 : m23, mm_shift, 1) & 0x3f800000);
                };
                auto fire = [&](const i4v (&xb)[C], float A) {
#pragma unroll
                    for (int j = 0; j < C; j++) {
                        mm[MM ? 4 * j + 0 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, __int_as_float(xb[j].x), mm[MM ? 4 * j + 0 : 0], 0, 0, 0);
                        mm[MM ? 4 * j + 1 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, __int_as_float(xb[j].y), mm[MM ? 4 * j + 1 : 0], 0, 0, 0);
                        mm[MM ? 4 * j + 2 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, __int_as_float(xb[j].z), mm[MM ? 4 * j + 2 : 0], 0, 0, 0);
                        mm[MM ? 4 * j + 3 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, __int_as_float(xb[j].w), mm[MM ? 4 * j + 3 : 0], 0, 0, 0);
                    }
                };
                int ng = (cnt + 3) >> 2;
                if (ng > 0) {
                    i4v xa[C], xc[C];
                    float Aa, Ac;
                    prep(xa, Aa);
                    ng--;
                    while (ng >= 2) {
                        prep(xc, Ac);
                        __builtin_amdgcn_sched_barrier(0);
                        fire(xa, Aa);
                        __builtin_amdgcn_sched_barrier(0);
                        prep(xa, Aa);
                        __builtin_amdgcn_sched_barrier(0);
                        fire(xc, Ac);
                        __builtin_amdgcn_sched_barrier(0);
                        ng -= 2;
                    }
                    if (ng == 1) {
                        prep(xc, Ac);
                        __builtin_amdgcn_sched_barrier(0);
                        fire(xa, Aa);
                        fire(xc, Ac);
                    } else {
                        fire(xa, Aa);
                    }
                }
            } else {
                for (; cnt >= G; cnt -= G) group(std::integral_constant<int, G>{}, std::false_type{});
                if (G == 8 && cnt > 4) group(std::integral_constant<int, G>{}, std::true_type{});
                else if (cnt > 0) group(std::integral_constant<int, 4>{}, std::true_type{});
            }
            if (SYNC) __builtin_amdgcn_s_barrier();      // time alignment only: nothing is exchanged through memory
        }
    } else if (lane == 0) atomicOr(ub_flag, 1);

    // mean, store and fused WTA of one pixel
    auto finish = [&](int q, const float (&a)[C]) {
        const unsigned bo = (unsigned)__builtin_amdgcn_readlane((int)pk_o, q), bi = (unsigned)__builtin_amdgcn_readlane((int)pk_i, q);
        const float fc = (float)(((int)(bo >> 16) - (int)(bo & 0xffffu) + 1) * ((int)(bi >> 16) - (int)(bi & 0xffffu) + 1));
        const int p = p0 + (q >> TWS) * W + (q & (QC - 1));
        float *dst = vout + (size_t)p * D + dl;
        float mean[C];
        // SKIP == 7: the correctly rounded quotient in 3 vector instructions per value (wave_quotient, smt_common.h) where
        // the other variants run the IEEE division sequence -- 48 divisions per wave are 8 % of this kernel's vector
        // instructions
        if constexpr (SWORD) wave_quotient<C, FULL>(a, fc, dl, D, mean);
        else {
#pragma unroll
            for (int k = 0; k < C; k++) mean[k] = a[k] / fc;
        }
#pragma unroll
        for (int k = 0; k < C; k++) {
            // streaming store: the output is not read by this kernel and would otherwise evict input
            // taps from L2
            if (FULL || dl + k < D) __builtin_nontemporal_store(mean[k], dst + k);
        }
        if (disp) {
            const int wd = wave_wta<C, FULL>(mean, dl, D);
            if (lane == 0) disp[p] = (float)wd;
        }
    };
    // accumulators are read with static register numbers; pixels that need the plain walk (reference UB,
    // or a NaN the flag arithmetic may have produced) are collected and handled by one rolled loop
    unsigned redo = 0;
    if constexpr (MM) {
        // lane l holds, for each of its four pixels q = 4 (l >> 4) + v, the hypotheses 64 j + 4 n + r (n = l & 15)
        const int myc = ((int)(pk_o >> 16) - (int)(pk_o & 0xffffu) + 1) * ((int)(pk_i >> 16) - (int)(pk_i & 0xffffu) + 1);   // lane q: taps of pixel q
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int q = 4 * mm_k + v;
            const bool have = (q >> TWS) < nrow && (q & (QC - 1)) < ncol;
            const float fc = (float)__builtin_amdgcn_ds_bpermute(q << 2, myc);
            const int p = p0 + (q >> TWS) * W + (q & (QC - 1));
            f4 mean[C];
            bool bad = ub;
#pragma unroll
            for (int j = 0; j < C; j++) {
                const f4 a = f4{mm[MM ? 4 * j : 0][v], mm[MM ? 4 * j + 1 : 0][v], mm[MM ? 4 * j + 2 : 0][v], mm[MM ? 4 * j + 3 : 0][v]};
                bad = bad || (a.x != a.x) || (a.y != a.y) || (a.z != a.z) || (a.w != a.w);
                mean[j] = f4{a.x / fc, a.y / fc, a.z / fc, a.w / fc};
            }
            // a pixel with a NaN anywhere in its row (or a reference-undefined tile) goes to the plain walk below
            const unsigned long long bal = __ballot(bad && have);
#pragma unroll
            for (int g = 0; g < 4; g++)
                if ((bal >> (16 * g)) & 0xffffull) redo |= 1u << (4 * g + v);
            const bool mine_bad = ((bal >> (16 * mm_k)) & 0xffffull) != 0;
            if (have && !mine_bad) {
                float *dst = vout + (size_t)p * D + 4 * mm_n;
#pragma unroll
                for (int j = 0; j < C; j++) __builtin_nontemporal_store(mean[j], reinterpret_cast<f4 *>(dst + 64 * j));
            }
            if (disp) {
                // first strict minimum over d (CrossArm.cpp:44-52): the lane's own candidates in increasing d, then the
                // lexicographic minimum of (key, d) over the 16 lanes of the row; NaN keys sort last, a NaN at d = 0 wins
                unsigned bk = 0xFFFFFFFFu; int bd = 0;
#pragma unroll
                for (int j = 0; j < C; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float x = mean[j][r];
                        const unsigned key = (x != x) ? 0xFFFFFFFFu : f32_key(x);
                        const int d = 64 * j + 4 * mm_n + r;
                        if ((j == 0 && r == 0) || key < bk) { bk = key; bd = d; }
                    }
                auto step = [&](auto ctrl) {
                    constexpr int CTRL = decltype(ctrl)::value;
                    const unsigned ok = (unsigned)__builtin_amdgcn_update_dpp((int)bk, (int)bk, CTRL, 0xF, 0xF, false);
                    const int od = __builtin_amdgcn_update_dpp(bd, bd, CTRL, 0xF, 0xF, false);
                    if (ok < bk || (ok == bk && od < bd)) { bk = ok; bd = od; }
                };
                step(std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
                step(std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
                step(std::integral_constant<int, 0x141>{});    // row_half_mirror
                step(std::integral_constant<int, 0x140>{});    // row_mirror: every lane of the row holds the row's minimum
                const float m0 = mean[0][0];
                const int nan0 = __builtin_amdgcn_ds_bpermute((16 * mm_k) << 2, (int)(m0 != m0));   // lane n = 0 of the row owns d = 0
                if (have && !mine_bad && mm_n == 0) disp[p] = (float)(nan0 ? 0 : bd);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < (MM ? 0 : NPIX); q++) {
        if ((q >> TWS) >= nrow || (q & (QC - 1)) >= ncol) continue;
        float a[C];
        bool bad = ub;
#pragma unroll
        for (int k = 0; k < C; k++) {
            a[k] = MFMA ? macc[MFMA ? q / 4 : 0][k][q & 3] : ((q & 1) ? acc[MFMA ? 0 : q / 2][k].y : acc[MFMA ? 0 : q / 2][k].x);
            bad = bad || (a[k] != a[k]);
        }
        if (__ballot(bad)) { redo |= 1u << q; continue; }
        finish(q, a);
    }
    while (redo) {
        const int q = __builtin_ctz(redo);
        redo &= redo - 1;
        const unsigned bo = (unsigned)__builtin_amdgcn_readlane((int)pk_o, q), bi = (unsigned)__builtin_amdgcn_readlane((int)pk_i, q);
        const int oa = (int)(bo & 0xffffu) - BIAS, ob = (int)(bo >> 16) - BIAS;
        const int ia = (int)(bi & 0xffffu) - BIAS, ib = (int)(bi >> 16) - BIAS;
        // the reference's own walk for this pixel; out-of-plane taps contribute nothing
        float a[C];
#pragma unroll
        for (int k = 0; k < C; k++) a[k] = 0.0f;
        for (int o = oa; o <= ob; o++)
            for (int t = ia; t <= ib; t++) {
                const long idx = (long)p0 + (long)o * so + (long)t * si;
                if (idx >= 0 && idx < N) {
                    float x[C];
                    ld((unsigned)idx * (unsigned)(D * 4), x);
#pragma unroll
                    for (int k = 0; k < C; k++) a[k] = a[k] + x[k];
                }
            }
        finish(q, a);
    }
}

__global__ void __launch_bounds__(NT) k_cblsm_ad(const uint8_t *__restrict__ L, const uint8_t *__restrict__ R,
                                                 int H, int W, int D, int view, float *__restrict__ vol)
{
    const size_t k = (size_t)blockIdx.x * NT + threadIdx.x;
    const size_t V = (size_t)H * W * D;
    if (k >= V) return;
    const int d = (int)(k % D);
    const size_t p = k / D;
    const int j = (int)(p % W);
    const size_t row = p - j;
    int a, b;
    if (view == 0) {                 // CBLSM.h:340-349: copy d-1 == clamp j-d at 0
        const int x = j - d < 0 ? 0 : j - d;
        a = L[p]; b = R[row + x];
    } else {                         // CBLSM.h:368-377
        const int x = j + d >= W ? W - 1 : j + d;
        a = L[row + x]; b = R[p];
    }
    vol[k] = (float)abs(a - b);
}

// CBLSM.h:65-236 chooseArmLength{Left,Right,Up,Down}: per-hypothesis arm lengths from the two
// views' arm maps (experiments whose call sites are commented out, CBLSM.cpp:108-111).  One thread
// per (i, j, d); the loops are the reference's, conditions in the reference's order.
__global__ void __launch_bounds__(NT) k_choose_arm(int dir, const int *__restrict__ own, const int *__restrict__ vert,
                                                   const int *__restrict__ RL, const int *__restrict__ RR, int H, int W,
                                                   int D, int *__restrict__ out)
{
    const size_t k = (size_t)blockIdx.x * NT + threadIdx.x;
    const size_t V = (size_t)H * W * D;
    if (k >= V) return;
    const int d = (int)(k % D);
    const size_t p = k / D;
    const int j = (int)(p % W), i = (int)(p / W);
    int save = 0;
    if (dir == 0) {                                            // :65-102
        const int LL = own[p], rl = RL[p], rr = RR[p];
        if (!((j - d < j - rl) || (j + d > j + rr)))
            for (int a = 1; a <= LL; a++) {
                if (((j - a - d) >= (j - rl)) && ((j - a - d) <= (j + rr))) save++;
                else break;
            }
    } else if (dir == 1) {                                     // :104-147
        const int LR = own[p], rl = RL[p], rr = RR[p];
        if (!((j - d < j - rl) || (j - d > j + rr)))
            for (int a = 1; a <= LR; a++) {
                if ((j + a - d >= j - rl) && (j + a - d < j + rr)) save++;
                else break;
            }
    } else if (dir == 2) {                                     // :151-192
        const int LUp = own[p], RUp = vert[p];
        for (int up = 1; up <= LUp; up++) {
            const int pr = i - up;
            const int pl = RL[(size_t)pr * W + j], prr = RR[(size_t)pr * W + j];
            if (pr >= i - RUp) {
                if (j - d < 0) break;
                if (((j - d) < (j + prr)) && ((j - d) > (j - pl))) save++;
            } else { save = 0; break; }
        }
    } else {                                                   // :195-236
        const int LDown = own[p], RDown = vert[p];
        for (int dn = 1; dn <= LDown; dn++) {
            const int pr = i + dn;
            const int pl = RL[(size_t)pr * W + j], prr = RR[(size_t)pr * W + j];
            if (pr <= i + RDown) {
                if (j - d < 0) { save = 0; break; }
                if ((j - d <= j + prr) && (j - d >= j - pl)) save++;
            } else break;
        }
    }
    out[k] = save;
}

// CBLSM.h:969-1045 ComputeLocalValue: mean-like value over a region whose rows carry their own
// per-hypothesis horizontal arms (see smt.h for the count quirk).  `value` is a float that absorbs each row's
// exact integer sum through a double add, as `value = value + sum(image)[0]` does.
__device__ __forceinline__ float cblsm_local_value(const uint8_t *__restrict__ img, int Hp, int Wp, int i, int j, int Up,
                                                   int Down, int w, const int *__restrict__ LArm,
                                                   const int *__restrict__ RArm, int H, int W, int D, int d)
{
    int count = 0;
    float value = 0.0f;
    const int ptrj = j - w;
    for (int r = -Up; r <= Down; r++) {
        const int ptr_i = i - w + r;
        if (ptr_i < 0 || ptr_i >= H || i + r < 0 || i + r >= Hp) continue;
        const int L = LArm[((size_t)ptr_i * W + ptrj) * D + d], R = RArm[((size_t)ptr_i * W + ptrj) * D + d];
        int c0, c1;
        if (d > 0) {
            if (j - L - d < 0) {
                if (j + R - d <= 0) { c0 = 0; c1 = 1; count += 1; }
                else { c0 = 0; c1 = j + R - d; count += R + 1; }
            } else { c0 = j - L - d; c1 = j + R - d; count += L + R + 1; }
        } else {
            if (L == 0 && R == 0) { c0 = j; c1 = j + 1; }
            else { c0 = j - L; c1 = j + R; }
            count += L + R + 1;
        }
        c0 = max(c0, 0); c1 = min(c1, Wp);
        int sum = 0;                                   // <= 4096 pixels of 255: exact
        const uint8_t *row = img + (size_t)(i + r) * Wp;
        for (int c = c0; c < c1; c++) sum += row[c];
        value = (float)((double)value + (double)sum);
    }
    return value / (float)count;
}

// costAggregationNew (CBLSM.h:1087-1126): one thread per (pixel, d)
__global__ void __launch_bounds__(NT) k_cblsm_cost_agg_new(const uint8_t *__restrict__ Lp, const uint8_t *__restrict__ Rp,
                                                           int Hp, int Wp, int w, const int *__restrict__ armL,
                                                           const int *__restrict__ armR, const int *__restrict__ armUp,
                                                           const int *__restrict__ armDown, int D, float *__restrict__ cost)
{
    const int H = Hp - 2 * w, W = Wp - 2 * w;
    const size_t k = (size_t)blockIdx.x * NT + threadIdx.x;
    if (k >= (size_t)H * W * D) return;
    const int d = (int)(k % D);
    const size_t p = k / D;
    const int j = (int)(p % W) + w, i = (int)(p / W) + w;
    const int Up = armUp[k], Down = armDown[k];
    const float lv = cblsm_local_value(Lp, Hp, Wp, i, j, Up, Down, w, armL, armR, H, W, D, 0);
    const float rv = cblsm_local_value(Rp, Hp, Wp, i, j, Up, Down, w, armL, armR, H, W, D, d);
    cost[k] = fabsf(lv - rv);
}

}  // namespace

struct smt_crossarm {
    int device;
    int H, W, D;
    smt_crossarm_params P;
    hipStream_t stream;
    int *arm[4];
    int *flip;     // 4 flip indices + 1 UB flag + the sticky threshold (`_tao`) as the last call left it
    bool have_arms;
    int variant;         // aggregation kernel variant (test / tuning hook)
    int strip_w;         // column-strip width of the XCD-aware pixel order (variants 0 and 2)
    int strip_w8;        // the same for variant 3 (8 pixels per wave)
    int sweep;           // 0: strips interleaved over XCDs, 1: each XCD owns a band of rows (variants 3-5)
    float *member;       // 256 x 8 membership flags for variant 3
    uint16_t *cand;      // [4][H][W] arm candidates {threshold still tau, already tau_low} (mask-based arm kernels)
    bool arm_walk;       // test hook: use the neighbour-by-neighbour kernels even when the masks apply
    int occ_lds;         // dynamic LDS bytes per aggregation workgroup, used as an occupancy limiter (see smt_crossarm_set_occupancy)
};

SMT_API void smt_crossarm_default_params(smt_crossarm_params *p)
{
    if (!p) return;
    p->tau = 30; p->tau_low = 6; p->sec_length = 17; p->max_length = 34; p->chain_tau = 1; p->quirks = 0;
}
SMT_API void smt_crossarm_cblsm_params(smt_crossarm_params *p)
{
    if (!p) return;
    p->tau = 25; p->tau_low = 6; p->sec_length = 17; p->max_length = 34; p->chain_tau = 0;
    p->quirks = SMT_QUIRK_FIX_RIGHT_ARM_STRIDE;
}

SMT_API int smt_crossarm_create(int H, int W, int D, const smt_crossarm_params *p, smt_crossarm **out)
{
    if (!out || H <= 0 || W <= 0 || D <= 0 || D > SMT_MAX_DISPARITY) return SMT_ERR_ARG;
    smt_crossarm *h = new (std::nothrow) smt_crossarm();
    if (!h) return SMT_ERR_ALLOC;
    h->device = smt_current_device();
    h->H = H; h->W = W; h->D = D; h->strip_w = 16; h->strip_w8 = 8; h->variant = 13;
    {
        // default: no limit (6 waves per SIMD from the register count), see smt_crossarm_set_occupancy; SMT_AGG_WAVES overrides
        static const int env_waves = [] { const char *e = getenv("SMT_AGG_WAVES"); return e ? atoi(e) : -1; }();
        const int waves = env_waves >= 0 ? env_waves : 0;
        h->occ_lds = (waves >= 3 && waves <= 5) ? (160 * 1024 / waves - 512) & ~255 : 0;   // <= 64 KB: no attribute needed
    }
    if (p) h->P = *p; else smt_crossarm_default_params(&h->P);
    if (h->P.sec_length < 0 || h->P.max_length < 0 || h->P.max_length > 4096) { delete h; return SMT_ERR_ARG; }
    int rc = SMT_OK;
    for (int k = 0; k < 4 && rc == SMT_OK; k++) rc = smt_malloc((void **)&h->arm[k], (size_t)H * W * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->flip, 8 * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->member, MEMBER_TAB_FLOATS * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->cand, (size_t)H * W * 4 * sizeof(uint16_t));
    if (rc != SMT_OK) { smt_crossarm_destroy(h); return rc; }
    if (hipMemset(h->flip, 0, 32) != hipSuccess) { smt_crossarm_destroy(h); return SMT_ERR_HIP; }
    // Initialize (CrossArm.cpp:6-18): `_tao = tao` and four value-initialised (zero) maps
    for (int k = 0; k < 4; k++)
        if (hipMemset(h->arm[k], 0, (size_t)H * W * 4) != hipSuccess) { smt_crossarm_destroy(h); return SMT_ERR_HIP; }
    if (hipMemcpy(h->flip + 5, &h->P.tau, 4, hipMemcpyHostToDevice) != hipSuccess) { smt_crossarm_destroy(h); return SMT_ERR_HIP; }
    {
        float tab[MEMBER_TAB_FLOATS];
        for (int m = 0; m < 256; m++)
            for (int q = 0; q < 8; q++) tab[m * 8 + q] = ((m >> q) & 1) ? 1.0f : 0.0f;
        if (hipMemcpy(h->member, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) { smt_crossarm_destroy(h); return SMT_ERR_HIP; }
    }
    *out = h;
    return SMT_OK;
}

SMT_API int smt_crossarm_create_on(int device, int H, int W, int D, const smt_crossarm_params *p, smt_crossarm **out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(device);
    return smt_crossarm_create(H, W, D, p, out);
}

SMT_API int smt_crossarm_destroy(smt_crossarm *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    for (int k = 0; k < 4; k++) (void)hipFree(h->arm[k]);
    (void)hipFree(h->flip);
    (void)hipFree(h->member);
    (void)hipFree(h->cand);
    delete h;
    return SMT_OK;
}

SMT_API int smt_crossarm_set_stream(smt_crossarm *h, void *s)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->stream = smt_stream(s);
    return SMT_OK;
}

static const int kFlipInit[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};

// flip search + arm lengths of the directions dir0 .. dir0 + grid.y - 1
static void launch_arms(smt_crossarm *h, const uint8_t *img, const ArmCfg &c, dim3 grid, int dir0, int prev_flips)
{
    const bool masks = !h->arm_walk && c.sec <= 63 && c.maxlen <= 63 && c.tau >= 0 && c.tau_low >= 0;
    if (masks) {
        if (c.ch == 1) hipLaunchKernelGGL(k_arm_cand<true>, grid, dim3(NT), 0, h->stream, img, c, h->flip, dir0, h->flip + 5, h->cand);
        else hipLaunchKernelGGL(k_arm_cand<false>, grid, dim3(NT), 0, h->stream, img, c, h->flip, dir0, h->flip + 5, h->cand);
        hipLaunchKernelGGL(k_arm_pick, grid, dim3(NT), 0, h->stream, c, h->flip, h->cand, h->arm[0], h->arm[1],
                           h->arm[2], h->arm[3], dir0, prev_flips, h->flip + 5);
    } else {
        hipLaunchKernelGGL(k_arm_flip, grid, dim3(NT), 0, h->stream, img, c, h->flip, dir0, h->flip + 5);
        hipLaunchKernelGGL(k_arms, grid, dim3(NT), 0, h->stream, img, c, h->flip, h->arm[0], h->arm[1],
                           h->arm[2], h->arm[3], dir0, prev_flips, h->flip + 5);
    }
}

SMT_API int smt_crossarm_reset(smt_crossarm *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const size_t N = (size_t)h->H * h->W;
    // `new int[col*row]()` zero-initialises every map on each Initialize (CrossArm.cpp:14-17);
    // with the stride bug most of rightLength stays 0.
    for (int k = 0; k < 4; k++) SMT_HIP(hipMemsetAsync(h->arm[k], 0, N * 4, h->stream));
    SMT_HIP(hipMemcpyAsync(h->flip + 5, &h->P.tau, 4, hipMemcpyHostToDevice, h->stream));   // `_tao = tao` (:13)
    h->have_arms = true;
    return SMT_OK;
}

SMT_API int smt_crossarm_arms(smt_crossarm *h, const uint8_t *img, int channels)
{
    if (!h || !img || (channels != 1 && channels != 3)) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const bool fix = (h->P.quirks & SMT_QUIRK_FIX_RIGHT_ARM_STRIDE) != 0;
    // with the stride bug the reference reads image columns up to H-1+max_length-ish of a
    // W-wide row: undefined for portrait images
    if (!fix && h->H > h->W) return SMT_ERR_REF_UB;
    ArmCfg c{h->H, h->W, channels, h->P.tau, h->P.tau_low, h->P.sec_length, h->P.max_length,
             h->P.chain_tau, fix ? 1 : 0};
    const size_t N = (size_t)h->H * h->W;
    int rc = smt_crossarm_reset(h);
    if (rc != SMT_OK) return rc;
    SMT_HIP(hipMemcpyAsync(h->flip, kFlipInit, 16, hipMemcpyHostToDevice, h->stream));
    dim3 grid((unsigned)((N + NT - 1) / NT), 4);
    launch_arms(h, img, c, grid, 0, 1);
    hipLaunchKernelGGL(k_tau_update, dim3(1), dim3(1), 0, h->stream, h->flip, 0, 4, h->P.chain_tau, h->P.tau_low,
                       h->flip + 5);
    SMT_LAUNCH_CHECK();
    h->have_arms = true;
    return SMT_OK;
}

SMT_API int smt_crossarm_arm_dir(smt_crossarm *h, const uint8_t *img, int channels, int dir)
{
    if (!h || !img || (channels != 1 && channels != 3) || dir < 0 || dir > 3) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const bool fix = (h->P.quirks & SMT_QUIRK_FIX_RIGHT_ARM_STRIDE) != 0;
    if (dir == 1 && !fix && h->H > h->W) return SMT_ERR_REF_UB;
    ArmCfg c{h->H, h->W, channels, h->P.tau, h->P.tau_low, h->P.sec_length, h->P.max_length,
             h->P.chain_tau, fix ? 1 : 0};
    const size_t N = (size_t)h->H * h->W;
    SMT_HIP(hipMemcpyAsync(h->flip + dir, kFlipInit, 4, hipMemcpyHostToDevice, h->stream));
    dim3 grid((unsigned)((N + NT - 1) / NT), 1);
    launch_arms(h, img, c, grid, dir, 0);
    hipLaunchKernelGGL(k_tau_update, dim3(1), dim3(1), 0, h->stream, h->flip, dir, 1, h->P.chain_tau, h->P.tau_low,
                       h->flip + 5);
    SMT_LAUNCH_CHECK();
    h->have_arms = true;
    return SMT_OK;
}

SMT_API int smt_crossarm_tau(smt_crossarm *h, int *tau)
{
    if (!h || !tau) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    SMT_HIP(hipMemcpyAsync(tau, h->flip + 5, 4, hipMemcpyDeviceToHost, h->stream));
    SMT_HIP(hipStreamSynchronize(h->stream));
    return SMT_OK;
}

SMT_API int smt_crossarm_arm_maps(smt_crossarm *h, int **l, int **r, int **t, int **b)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (l) *l = h->arm[0];
    if (r) *r = h->arm[1];
    if (t) *t = h->arm[2];
    if (b) *b = h->arm[3];
    return SMT_OK;
}

template <int ORDER, int QP>
static void launch_agg_quad(smt_crossarm *h, const float *vin, float *vout, float *disp)
{
    int SW = h->strip_w;
    const int bw = QP * (NT / 64);
    SW = ((SW + bw - 1) / bw) * bw;
    const int nstrips = (h->W + SW - 1) / SW;
    const int per_xcd = (nstrips + 7) / 8;
    dim3 grid((unsigned)(8 * per_xcd * (SW / bw) * h->H));
    const int C = (h->D + 63) / 64;
    const bool full = (h->D == 64 * C);
    int *ub = h->flip + 4;
#define SMT_AGGQ(CC, FF)                                                                                  \
    hipLaunchKernelGGL((k_aggregate_quad<CC, ORDER, FF, QP>), grid, dim3(NT), 0, h->stream, vin, vout, h->H, h->W, \
                       h->D, h->arm[0], h->arm[1], h->arm[2], h->arm[3], disp, ub, SW)
    switch (C * 2 + (full ? 1 : 0)) {
    case 2: SMT_AGGQ(1, false); break;
    case 3: SMT_AGGQ(1, true); break;
    case 4: SMT_AGGQ(2, false); break;
    case 5: SMT_AGGQ(2, true); break;
    case 6: SMT_AGGQ(3, false); break;
    case 7: SMT_AGGQ(3, true); break;
    case 8: SMT_AGGQ(4, false); break;
    default: SMT_AGGQ(4, true); break;
    }
#undef SMT_AGGQ
}

template <int ORDER, int QR, int SKIP, int TWS = 3>
static void launch_agg_multi(smt_crossarm *h, const float *vin, float *vout, float *disp)
{
    // strip width 8 / 16 / multiple of 32: the 4 waves of a workgroup sit 1x4, 2x2 or 4x1 (tiles 8 wide; 4-wide
    // tiles: 2x2 at strip width 8, 4x1 from 16 on)
    constexpr int TW = 1 << TWS, TH = (8 * QR) >> TWS;
    int SW = h->strip_w8;
    SW = SW <= 8 ? 8 : SW <= 16 ? 16 : ((SW + 31) / 32) * 32;
    const int wx = SW / TW < 4 ? SW / TW : 4, wy = 4 / wx;
    const int nstrips = (h->W + SW - 1) / SW;
    const int per_xcd = (nstrips + 7) / 8;
    const int nband = (h->H + wy * TH - 1) / (wy * TH);
    dim3 grid(h->sweep == 0 ? (unsigned)(8 * per_xcd * (SW / (TW * wx)) * nband)
                            : (unsigned)(8 * nstrips * (SW / (TW * wx)) * ((nband + 7) / 8)));
    const int C = (h->D + 63) / 64;
    const bool full = (h->D == 64 * C);
    int *ub = h->flip + 4;
#define SMT_AGGM(CC, FF)                                                                                  \
    hipLaunchKernelGGL((k_aggregate_multi<CC, ORDER, FF, QR, SKIP, TWS>), grid, dim3(NT), (size_t)h->occ_lds, h->stream, vin, vout, h->H, h->W, \
                       h->D, h->arm[0], h->arm[1], h->arm[2], h->arm[3], disp, ub, SW, h->member, h->sweep)
    if constexpr (SKIP == 6) {
        switch (C) {                                       // D is a multiple of 64 here (smt_crossarm_aggregate)
        case 1: SMT_AGGM(1, true); break;
        case 2: SMT_AGGM(2, true); break;
        case 3: SMT_AGGM(3, true); break;
        default: SMT_AGGM(4, true); break;
        }
    } else {
        switch (C * 2 + (full ? 1 : 0)) {
        case 2: SMT_AGGM(1, false); break;
        case 3: SMT_AGGM(1, true); break;
        case 4: SMT_AGGM(2, false); break;
        case 5: SMT_AGGM(2, true); break;
        case 6: SMT_AGGM(3, false); break;
        case 7: SMT_AGGM(3, true); break;
        case 8: SMT_AGGM(4, false); break;
        default: SMT_AGGM(4, true); break;
        }
    }
#undef SMT_AGGM
}

template <int ORDER>
static void launch_agg_pipe(smt_crossarm *h, const float *vin, float *vout, float *disp)
{
    const int SW = h->strip_w;                           // strip width (multiple of 4)
    const int nstrips = (h->W + SW - 1) / SW;
    const int per_xcd = (nstrips + 7) / 8;               // strips per XCD
    dim3 grid((unsigned)(8 * per_xcd * (SW / 4) * h->H));
    const int C = (h->D + 63) / 64;
    const bool full = (h->D == 64 * C);
    int *ub = h->flip + 4;
#define SMT_AGGP(CC, FF)                                                                                  \
    hipLaunchKernelGGL((k_aggregate_pipe<CC, ORDER, FF>), grid, dim3(NT), 0, h->stream, vin, vout, h->H, h->W, \
                       h->D, h->arm[0], h->arm[1], h->arm[2], h->arm[3], disp, ub, SW)
    switch (C * 2 + (full ? 1 : 0)) {
    case 2: SMT_AGGP(1, false); break;
    case 3: SMT_AGGP(1, true); break;
    case 4: SMT_AGGP(2, false); break;
    case 5: SMT_AGGP(2, true); break;
    case 6: SMT_AGGP(3, false); break;
    case 7: SMT_AGGP(3, true); break;
    case 8: SMT_AGGP(4, false); break;
    default: SMT_AGGP(4, true); break;
    }
#undef SMT_AGGP
}

template <int ORDER>
static void launch_agg(smt_crossarm *h, const float *vin, float *vout, float *disp)
{
    const int N = h->H * h->W;
    dim3 grid((N + 3) / 4);
    const int C = (h->D + 63) / 64;
    int *ub = h->flip + 4;
#define SMT_AGG(CC)                                                                                   \
    hipLaunchKernelGGL((k_aggregate<CC, ORDER>), grid, dim3(NT), 0, h->stream, vin, vout, h->H, h->W, \
                       h->D, h->arm[0], h->arm[1], h->arm[2], h->arm[3], disp, ub)
    switch (C) {
    case 1: SMT_AGG(1); break;
    case 2: SMT_AGG(2); break;
    case 3: SMT_AGG(3); break;
    case 4: SMT_AGG(4); break;
    case 5: SMT_AGG(5); break;
    case 6: SMT_AGG(6); break;
    case 7: SMT_AGG(7); break;
    default: SMT_AGG(8); break;
    }
#undef SMT_AGG
}

// arm maps computed elsewhere (costAggregationV5 and AggregationVertical take them as plain int arrays):
// copied into the handle; a length outside 0..8191 is not something either reference loop survives (its
// rectangle leaves the plane or is empty: 0/0) -- clamped, and reported by smt_crossarm_status as SMT_ERR_REF_UB
__global__ void __launch_bounds__(NT) k_load_arms(const int *__restrict__ l, const int *__restrict__ r, const int *__restrict__ t,
                                                  const int *__restrict__ b, int *__restrict__ ol, int *__restrict__ orr,
                                                  int *__restrict__ ot, int *__restrict__ ob, int n, int *ub_flag)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= n) return;
    const int v[4] = {l[p], r[p], t[p], b[p]};
    bool bad = false;
    int c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { c[k] = min(max(v[k], 0), 8191); bad = bad || c[k] != v[k]; }
    ol[p] = c[0]; orr[p] = c[1]; ot[p] = c[2]; ob[p] = c[3];
    if (bad) atomicOr(ub_flag, 1);
}

SMT_API int smt_crossarm_load_arm_maps(smt_crossarm *h, const int *left, const int *right, const int *top, const int *bottom)
{
    if (!h || !left || !right || !top || !bottom) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const int n = h->H * h->W;
    hipLaunchKernelGGL(k_load_arms, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, h->stream, left, right, top, bottom,
                       h->arm[0], h->arm[1], h->arm[2], h->arm[3], n, h->flip + 4);
    SMT_LAUNCH_CHECK();
    h->have_arms = true;
    return SMT_OK;
}

SMT_API int smt_crossarm_aggregate(smt_crossarm *h, const float *vin, float *vout, int order, float *disp)
{
    if (!h || !vin || !vout || vin == vout || order < 0 || order > 2) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (!h->have_arms) return SMT_ERR_STATE;
    // variant: 12 = 4x4 pixels per wave sharing the taps of the union of their rectangles, membership flags,
    // groups of 4 pixels (= tile rows) without a member skipped, flag rows prefetched, per-axis membership tables, the
    // four waves of a workgroup (8 x 8 pixels) in lock-step; 13 = 12 with the tap's flag-row offsets and group bits
    // packed into one word by the batch classification (default); 7 = 12 with 2x8 tiles (round 2's default);
    // 6 = 7 free-running; 4 = 6 with flags per live group and pixel-by-pixel classification; 5 = 4 without the skip;
    // 3 = 1x8 pixels, no skip; 8 / 9 = 7 with the flagged accumulate on the matrix pipe (v_mfma_f32_4x4x1: every
    // group / live groups only), 11 = 9 with 4x4 tiles, 10 = four taps per v_mfma_f32_16x16x4 (4x4 tiles, free-running);
    // 0 = 4 adjacent pixels per wave with a 16-way switch on the membership mask; 1 = plain one-pixel-per-wave walk
    // (also the form used for volumes >= 4 GiB, for D > 256 and for order 2); 2 = pipelined one-pixel-per-wave walk.
    // All but 1 address taps with 32-bit byte offsets.
    int variant = h->variant;
    if (variant != 1 && (size_t)h->H * h->W * h->D * 4 >= ((size_t)1 << 32)) variant = 1;
    if (h->D > 256) variant = 1;                          // 5..8 hypotheses per lane: the plain walk only
    if (variant == 10 && (h->D % 64 != 0 || ((uintptr_t)vin & 15) != 0)) variant = 12;   // the four-taps form gathers whole 16-byte quads
    if (order == 2) { launch_agg<2>(h, vin, vout, disp); SMT_LAUNCH_CHECK(); return SMT_OK; }   // inactive sibling: plain walk only
    if (variant == 0) { if (order == 0) launch_agg_quad<0, 4>(h, vin, vout, disp); else launch_agg_quad<1, 4>(h, vin, vout, disp); }
    else if (variant == 1) { if (order == 0) launch_agg<0>(h, vin, vout, disp); else launch_agg<1>(h, vin, vout, disp); }
    else if (variant == 3) { if (order == 0) launch_agg_multi<0, 1, 0>(h, vin, vout, disp); else launch_agg_multi<1, 1, 0>(h, vin, vout, disp); }
    else if (variant == 4) { if (order == 0) launch_agg_multi<0, 2, 1>(h, vin, vout, disp); else launch_agg_multi<1, 2, 1>(h, vin, vout, disp); }
    else if (variant == 5) { if (order == 0) launch_agg_multi<0, 2, 0>(h, vin, vout, disp); else launch_agg_multi<1, 2, 0>(h, vin, vout, disp); }
    else if (variant == 6) { if (order == 0) launch_agg_multi<0, 2, 2>(h, vin, vout, disp); else launch_agg_multi<1, 2, 2>(h, vin, vout, disp); }
    else if (variant == 7) { if (order == 0) launch_agg_multi<0, 2, 3>(h, vin, vout, disp); else launch_agg_multi<1, 2, 3>(h, vin, vout, disp); }
    else if (variant == 8) { if (order == 0) launch_agg_multi<0, 2, 4>(h, vin, vout, disp); else launch_agg_multi<1, 2, 4>(h, vin, vout, disp); }
    else if (variant == 9) { if (order == 0) launch_agg_multi<0, 2, 5>(h, vin, vout, disp); else launch_agg_multi<1, 2, 5>(h, vin, vout, disp); }
    else if (variant == 10) { if (order == 0) launch_agg_multi<0, 2, 6, 2>(h, vin, vout, disp); else launch_agg_multi<1, 2, 6, 2>(h, vin, vout, disp); }
    else if (variant == 11) { if (order == 0) launch_agg_multi<0, 2, 5, 2>(h, vin, vout, disp); else launch_agg_multi<1, 2, 5, 2>(h, vin, vout, disp); }
    else if (variant == 12) { if (order == 0) launch_agg_multi<0, 2, 3, 2>(h, vin, vout, disp); else launch_agg_multi<1, 2, 3, 2>(h, vin, vout, disp); }
    else if (variant == 13) { if (order == 0) launch_agg_multi<0, 2, 7, 2>(h, vin, vout, disp); else launch_agg_multi<1, 2, 7, 2>(h, vin, vout, disp); }
    else { if (order == 0) launch_agg_pipe<0>(h, vin, vout, disp); else launch_agg_pipe<1>(h, vin, vout, disp); }
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_crossarm_set_variant(smt_crossarm *h, int variant)
{
    if (!h || variant < 0 || variant > 13) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->variant = variant;
    // the lock-step kernel wants its four waves stacked vertically (8 columns x 8 rows per workgroup: the
    // waves then walk the same columns); the free-running ones measure best with 16-column strips
    h->strip_w8 = variant >= 7 ? 8 : 16;
    return SMT_OK;
}

SMT_API int smt_crossarm_set_strip_width(smt_crossarm *h, int w)
{
    if (!h || w < 4 || (w & 3)) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->strip_w = w; h->strip_w8 = w;
    return SMT_OK;
}

SMT_API int smt_crossarm_set_arm_walk(smt_crossarm *h, int on)
{
    if (!h) return SMT_ERR_ARG;
    h->arm_walk = on != 0;
    return SMT_OK;
}

// Aggregation workgroups per CU (= waves per SIMD: a workgroup is four waves, one per SIMD), enforced through an LDS
// claim the kernel does not use.  The kernel needs 80 VGPRs; left alone it runs 6 waves per SIMD = 480 of the 512
// VGPRs, so a kernel on another stream finds no room until the aggregation's grid is exhausted.  Round 3 measured what
// making room buys: nothing -- scanline passes resident beside the aggregation slow down with it,
// both wait on the same memory system -- while the aggregation alone loses 15 % at 4 or 5 waves per SIMD.  The default
// is therefore no limit; the knob stays for experiments.
SMT_API int smt_crossarm_set_occupancy(smt_crossarm *h, int waves_per_simd)
{
    if (!h || !(waves_per_simd == 0 || (waves_per_simd >= 3 && waves_per_simd <= 5))) return SMT_ERR_ARG;
    h->occ_lds = (waves_per_simd >= 3 && waves_per_simd <= 5) ? (160 * 1024 / waves_per_simd - 512) & ~255 : 0;
    return SMT_OK;
}

SMT_API int smt_crossarm_set_sweep(smt_crossarm *h, int sweep)
{
    if (!h || sweep < 0 || sweep > 1) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->sweep = sweep;
    return SMT_OK;
}

SMT_API int smt_crossarm_status(smt_crossarm *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    int f = 0;
    SMT_HIP(hipMemcpyAsync(&f, h->flip + 4, 4, hipMemcpyDeviceToHost, h->stream));
    SMT_HIP(hipMemsetAsync(h->flip + 4, 0, 4, h->stream));          // read-and-clear: reports what happened since the last call
    SMT_HIP(hipStreamSynchronize(h->stream));
    return f ? SMT_ERR_REF_UB : SMT_OK;
}

SMT_API int smt_cblsm_ad(const uint8_t *L, const uint8_t *R, int H, int W, int D, int view, float *vol,
                         void *stream)
{
    if (!L || !R || !vol || H <= 0 || W <= 0 || D <= 0 || (view != SMT_VIEW_LEFT && view != SMT_VIEW_RIGHT))
        return SMT_ERR_ARG;
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_cblsm_ad, dim3((unsigned)((V + NT - 1) / NT)), dim3(NT), 0, smt_stream(stream), L, R,
                       H, W, D, view == SMT_VIEW_LEFT ? 0 : 1, vol);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_cblsm_choose_arm_length(int dir, const int *own_arm, const int *other_vertical_arm,
                                        const int *armRL, const int *armRR, int H, int W, int D, int *arm_volume,
                                        void *stream)
{
    if (dir < 0 || dir > 3 || !own_arm || !armRL || !armRR || !arm_volume || H <= 0 || W <= 0 || D <= 0 ||
        (dir >= 2 && !other_vertical_arm))
        return SMT_ERR_ARG;
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_choose_arm, dim3((unsigned)((V + NT - 1) / NT)), dim3(NT), 0, smt_stream(stream), dir, own_arm,
                       other_vertical_arm, armRL, armRR, H, W, D, arm_volume);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_cblsm_cost_aggregation_new(const uint8_t *Lp, const uint8_t *Rp, int H, int W, int D, int winSize,
                                           const int *armvolL, const int *armvolR, const int *armvolUp,
                                           const int *armvolDown, float *cost, void *stream)
{
    if (!Lp || !Rp || !armvolL || !armvolR || !armvolUp || !armvolDown || !cost || H <= 0 || W <= 0 || D <= 0 ||
        winSize < 0)
        return SMT_ERR_ARG;
    const int w = winSize + 1;
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_cblsm_cost_agg_new, dim3((unsigned)((V + NT - 1) / NT)), dim3(NT), 0, smt_stream(stream), Lp, Rp,
                       H + 2 * w, W + 2 * w, w, armvolL, armvolR, armvolUp, armvolDown, D, cost);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}
